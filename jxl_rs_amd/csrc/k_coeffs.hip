// Sparse coefficient transport: on-device zero-fill + scatter of the entropy decoder's
// (position, value) stream into the dense per-group slabs K1 reads.
//
// The reference keeps coeffs[3][65536] i32 dense per group (frame/group.rs:437-440, with the TODO
// "use 16 bits if possible", :53-55, :428) and fills it with `coeffs[c][offset + order[k]] += v`
// while decoding (:560-575); at d1 about 90 % of the slab stays zero.  Shipping the dense slab over
// PCIe (12 B/px) is the end-to-end limiter, so the host sends only the
// non-zero entries as 4-byte (u16 position, i16 value) pairs per channel -- values outside i16
// travel in a side list of 8-byte pairs -- and this kernel rebuilds the slab in HBM:
//   one 1024-thread workgroup per quarter of a (group, channel) slab: the 64 KB quarter is built in
//   LDS (zero, then ds_add scatter of the channel's pairs that fall into it -- wrapping i32 `+=`:
//   duplicate positions accumulate like the reference's multi-pass accumulation, and the result
//   is order independent) and leaves as 16-byte coalesced stores, so HBM sees exactly one dense
//   write of the slab and one read of the pairs per quarter (the pair runs are tiny and L2-hot).
#include "jxlh_internal.h"

namespace jxlh {
namespace {

constexpr int kExpandThreads = 1024;
constexpr int kQuarter = kGroupArea / 4;  // 16384 coefficients = 64 KB of LDS

__global__ __launch_bounds__(kExpandThreads) void k_expand_sparse(int32_t* __restrict__ coeffs,
                                                                  const uint32_t* __restrict__ pairs,
                                                                  const SparseGroup* __restrict__ groups,
                                                                  const uint8_t* __restrict__ only_flagged) {
  __shared__ __attribute__((aligned(16))) int32_t s_q[kQuarter];
  const int q = blockIdx.x & 3, c = (blockIdx.x >> 2) % 3, tid = threadIdx.x;
  const SparseGroup sg = groups[blockIdx.x / 12];
  if (only_flagged && !only_flagged[sg.group]) return;  // K1 reads this group's pairs directly
  int4* s4 = reinterpret_cast<int4*>(s_q);
  int4* d4 = reinterpret_cast<int4*>(coeffs + ((size_t)sg.group * 3 + c) * kGroupArea + q * kQuarter);
  // a later pass of a progressive frame adds to what the earlier passes left (frame/group.rs:572 `+=` on the
  // frame's hf_coefficients, frame/decode.rs:547-558); a first / only submission starts from zero
  const bool accumulate = (sg.flags & 1u) != 0;
#pragma unroll
  for (int i = 0; i < kQuarter / 4 / kExpandThreads; i++)
    s4[i * kExpandThreads + tid] = accumulate ? d4[i * kExpandThreads + tid] : make_int4(0, 0, 0, 0);
  __syncthreads();
  const uint32_t n = sg.n[c];
  const uint32_t first = sg.offset + (c > 0 ? sg.n[0] : 0u) + (c > 1 ? sg.n[1] : 0u);
  const uint32_t* __restrict__ src = pairs + first;
  // 8 independent loads in flight per thread: the scan is latency bound otherwise
  for (uint32_t base = 0; base < n; base += 8 * kExpandThreads) {
    uint32_t v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const uint32_t i = base + j * kExpandThreads + tid;
      v[j] = i < n ? src[i] : 0xffffffffu;
    }
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const uint32_t i = base + j * kExpandThreads + tid;
      const uint32_t pos = v[j] & 0xffffu;  // little endian {u16 pos; i16 val}
      if (i < n && (int)(pos >> 14) == q) atomicAdd(&s_q[pos & (kQuarter - 1)], (int32_t)(int16_t)(v[j] >> 16));
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < kQuarter / 4 / kExpandThreads; i++) d4[i * kExpandThreads + tid] = s4[i * kExpandThreads + tid];
}

// Buckets one (group, channel) run of pairs by 64-coefficient slot (= the unit varblock
// coefficient offsets are counted in, group.rs:612): sorted[] receives the run reordered so that
// a varblock's pairs are contiguous, slot_start[] the first index of every slot.  The order of
// pairs inside a slot is arbitrary (K1 adds them up before dequantising).
__global__ __launch_bounds__(kExpandThreads) void k_sort_sparse(const uint32_t* __restrict__ pairs,
                                                                const SparseGroup* __restrict__ groups,
                                                                uint32_t* __restrict__ sorted,
                                                                uint32_t* __restrict__ slot_start) {
  __shared__ uint32_t s_count[1024], s_cursor[1024], s_wsum[kExpandThreads / 64];
  const SparseGroup sg = groups[blockIdx.x / 3];
  const int c = blockIdx.x % 3, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t n = sg.n[c];
  const uint32_t first = sg.offset + (c > 0 ? sg.n[0] : 0u) + (c > 1 ? sg.n[1] : 0u);
  const uint32_t* __restrict__ src = pairs + first;
  s_count[tid] = 0;
  __syncthreads();
  for (uint32_t i = tid; i < n; i += kExpandThreads) atomicAdd(&s_count[(src[i] & 0xffffu) >> 6], 1u);
  __syncthreads();
  // exclusive scan of the 1024 counts (one per thread)
  const uint32_t mine = s_count[tid];
  uint32_t incl = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t v = __shfl_up(incl, d, 64);
    if (lane >= d) incl += v;
  }
  if (lane == 63) s_wsum[wave] = incl;
  __syncthreads();
  uint32_t base = 0;
  for (int w = 0; w < wave; w++) base += s_wsum[w];
  const uint32_t excl = base + incl - mine;
  s_cursor[tid] = excl;
  uint32_t* table = slot_start + ((size_t)sg.group * 3 + c) * kSlotTable;
  table[tid] = first + excl;
  if (tid == 0) table[1024] = first + n;
  __syncthreads();
  for (uint32_t i = tid; i < n; i += kExpandThreads) {
    const uint32_t p = src[i];
    const uint32_t at = atomicAdd(&s_cursor[(p & 0xffffu) >> 6], 1u);
    sorted[first + at] = p;
  }
}

// Dense slab of a group from its bucketed pairs (no descriptors needed: the slot table delimits every
// quarter of every (group, channel) run).  flags: expand group g only if flags[g] != 0.
__global__ __launch_bounds__(kExpandThreads) void k_expand_sorted(int32_t* __restrict__ coeffs,
                                                                  const uint32_t* __restrict__ sorted,
                                                                  const uint32_t* __restrict__ slot_start,
                                                                  const uint8_t* __restrict__ flags) {
  __shared__ __attribute__((aligned(16))) int32_t s_q[kQuarter];
  const int q = blockIdx.x & 3, c = (blockIdx.x >> 2) % 3, group = blockIdx.x / 12, tid = threadIdx.x;
  if (!flags[group]) return;
  int4* s4 = reinterpret_cast<int4*>(s_q);
#pragma unroll
  for (int i = 0; i < kQuarter / 4 / kExpandThreads; i++) s4[i * kExpandThreads + tid] = make_int4(0, 0, 0, 0);
  __syncthreads();
  const uint32_t* table = slot_start + ((size_t)group * 3 + c) * kSlotTable + q * (kQuarter / 64);
  const uint32_t i0 = table[0], i1 = table[kQuarter / 64];
  for (uint32_t i = i0 + tid; i < i1; i += kExpandThreads) {
    const uint32_t p = sorted[i];
    atomicAdd(&s_q[p & (kQuarter - 1)], (int32_t)(int16_t)(p >> 16));
  }
  __syncthreads();
  int4* d4 = reinterpret_cast<int4*>(coeffs + ((size_t)group * 3 + c) * kGroupArea + q * kQuarter);
#pragma unroll
  for (int i = 0; i < kQuarter / 4 / kExpandThreads; i++) d4[i * kExpandThreads + tid] = s4[i * kExpandThreads + tid];
}

// values outside i16: pos = (group * 3 + channel) * 65536 + position
__global__ void k_expand_wide(int32_t* __restrict__ coeffs, const uint2* __restrict__ wide, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) atomicAdd(&coeffs[wide[i].x], (int32_t)wide[i].y);
}

}  // namespace

void launch_sort_sparse(hipStream_t s, const uint32_t* pairs, const SparseGroup* groups, int n_groups,
                        uint32_t* sorted, uint32_t* slot_start) {
  if (n_groups > 0)
    hipLaunchKernelGGL(k_sort_sparse, dim3(n_groups * 3), dim3(kExpandThreads), 0, s, pairs, groups, sorted,
                       slot_start);
}

void launch_expand_sorted(hipStream_t s, int32_t* coeffs, const uint32_t* sorted, const uint32_t* slot_start,
                          const uint8_t* flags, int n_groups) {
  if (n_groups > 0)
    hipLaunchKernelGGL(k_expand_sorted, dim3(n_groups * 12), dim3(kExpandThreads), 0, s, coeffs, sorted, slot_start,
                       flags);
}

void launch_expand_sparse(hipStream_t s, int32_t* coeffs, const uint32_t* pairs, const SparseGroup* groups,
                          int n_groups, const uint2* wide, uint32_t n_wide, const uint8_t* only_flagged) {
  if (n_groups > 0)
    hipLaunchKernelGGL(k_expand_sparse, dim3(n_groups * 12), dim3(kExpandThreads), 0, s, coeffs, pairs, groups,
                       only_flagged);
  if (n_wide > 0)
    hipLaunchKernelGGL(k_expand_wide, dim3((n_wide + 255) / 256), dim3(256), 0, s, coeffs, wide, n_wide);
}

namespace {
// {u16 pos} + {i8 val} -> the {u16 pos; i16 val} pair word of the sparse pipeline
__global__ __launch_bounds__(256) void k_pack_pairs8(const uint16_t* __restrict__ pos, const int8_t* __restrict__ val,
                                                     size_t n, uint32_t* __restrict__ pairs) {
  const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i + 4 <= n && ((reinterpret_cast<uintptr_t>(pos + i) & 7) == 0) && ((reinterpret_cast<uintptr_t>(val + i) & 3) == 0) &&
      ((reinterpret_cast<uintptr_t>(pairs + i) & 15) == 0)) {
    const uint2 p = *reinterpret_cast<const uint2*>(pos + i);
    const uint32_t v = *reinterpret_cast<const uint32_t*>(val + i);
    auto mk = [](uint32_t ps, uint32_t vb) { return ps | ((uint32_t)(uint16_t)(int16_t)(int8_t)vb << 16); };
    *reinterpret_cast<uint4*>(pairs + i) = make_uint4(mk(p.x & 0xffffu, v & 0xffu), mk(p.x >> 16, (v >> 8) & 0xffu),
                                                      mk(p.y & 0xffffu, (v >> 16) & 0xffu), mk(p.y >> 16, v >> 24));
  } else {
    for (size_t k = i; k < n && k < i + 4; k++) pairs[k] = (uint32_t)pos[k] | ((uint32_t)(uint16_t)(int16_t)val[k] << 16);
  }
}
// the 2-byte form: one workgroup per (group of the batch, channel).  desc[4 * (3 i + c) + {0, 1, 2, 3}] = first entry of
// the run in `entries`, first overflow update in pos8 / val8, number of overflow updates, first pair of the run in
// `pairs`; seg_counts as in the ABI.  Output: the run's pair words, nibble entries first (segment order), then the
// overflow updates.
__global__ __launch_bounds__(256) void k_pack_pairs4(const uint16_t* __restrict__ entries, const uint16_t* __restrict__ seg_counts,
                                                     const uint16_t* __restrict__ pos8, const int8_t* __restrict__ val8,
                                                     const uint32_t* __restrict__ desc, uint32_t* __restrict__ pairs) {
  __shared__ uint32_t s_start[17];
  const int run = blockIdx.x;
  if (threadIdx.x == 0) {
    uint32_t acc = 0;
    for (int s = 0; s < 16; s++) {
      s_start[s] = acc;
      acc += seg_counts[run * 16 + s];
    }
    s_start[16] = acc;
  }
  __syncthreads();
  const uint32_t e0 = desc[4 * run], o0 = desc[4 * run + 1], no = desc[4 * run + 2], p0 = desc[4 * run + 3];
  const uint32_t n4 = s_start[16];
  for (uint32_t j = threadIdx.x; j < n4; j += 256) {
    int seg = 0;  // largest s with s_start[s] <= j (a 16-way select chain on LDS broadcasts)
#pragma unroll
    for (int s = 1; s < 16; s++) seg += j >= s_start[s] ? 1 : 0;
    const uint32_t e = entries[e0 + j];
    const int32_t v = (int32_t)(e << 16) >> 28;  // sign-extended nibble
    pairs[p0 + j] = ((uint32_t)seg << 12 | (e & 0xfffu)) | ((uint32_t)(uint16_t)(int16_t)v << 16);
  }
  for (uint32_t j = threadIdx.x; j < no; j += 256)
    pairs[p0 + n4 + j] = (uint32_t)pos8[o0 + j] | ((uint32_t)(uint16_t)(int16_t)val8[o0 + j] << 16);
}
// ---- the slot-bucketed form (jxlh_submit_groups_slots), kept on the device exactly as uploaded: u16 entries in
// (group, channel, slot) order, one u8 count per slot, {first entry, entries} per (group, channel) run.  The transforms
// read it in place (k1_scan derives the per-varblock ranges); the kernels below serve the other routes.
__device__ __forceinline__ uint32_t block_excl_scan_1024(uint32_t mine, uint32_t* s_wsum, int tid) {
  const int lane = tid & 63, wave = tid >> 6;
  uint32_t incl = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t v = __shfl_up(incl, d, 64);
    if (lane >= d) incl += v;
  }
  if (lane == 63) s_wsum[wave] = incl;
  __syncthreads();
  uint32_t base = 0;
  for (int w = 0; w < wave; w++) base += s_wsum[w];
  return base + incl - mine;
}
__device__ __forceinline__ uint32_t entry_pair_word(uint32_t e, uint32_t slot) {
  const int32_t v = (int32_t)(e << 16) >> 22;  // sign-extended 10 bits
  return (slot << 6 | (e & 63u)) | ((uint32_t)(uint16_t)(int16_t)v << 16);
}

// 12-bit entries, two per three bytes (JXLH_GROUP_ENTRIES12): every (group, channel) run holds an even number of them,
// so a whole batch is one stream of 3-byte pairs -> two u16 entries (6-bit value sign-extended to 10)
__global__ __launch_bounds__(256) void k_unpack_entries12(const uint8_t* __restrict__ bytes, size_t n_pairs,
                                                          uint16_t* __restrict__ entries) {
  const size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= n_pairs) return;
  const uint8_t* b = bytes + k * 3;
  const uint32_t e0 = (uint32_t)b[0] | ((uint32_t)b[1] & 15u) << 8, e1 = (uint32_t)b[1] >> 4 | (uint32_t)b[2] << 4;
  auto widen = [](uint32_t e) {
    const int32_t v = (int32_t)(e << 20) >> 26;  // sign-extended 6 bits
    return (uint32_t)(e & 63u) | ((uint32_t)v & 1023u) << 6;
  };
  *reinterpret_cast<uint32_t*>(entries + 2 * k) = widen(e0) | widen(e1) << 16;  // runs start on even entries: aligned
}

// entries of the flagged groups -> pair words at the same indices (thread = slot).  A count table that claims more
// than the run holds is cut at the run's end; entries the table does not account for become zero updates.
__global__ __launch_bounds__(kExpandThreads) void k_entries_to_pairs(const uint16_t* __restrict__ entries,
                                                                    const uint8_t* __restrict__ counts,
                                                                    const uint2* __restrict__ runs,
                                                                    const uint8_t* __restrict__ flags,
                                                                    uint32_t* __restrict__ pairs) {
  __shared__ uint32_t s_wsum[kExpandThreads / 64];
  const int gc = blockIdx.x, tid = threadIdx.x;
  if (!flags[gc / 3]) return;
  const uint2 run = runs[gc];
  const uint32_t mine = counts[(size_t)gc * kSlotsPerRun + tid];
  const uint32_t excl = block_excl_scan_1024(mine, s_wsum, tid);
  const uint32_t lo = min(excl, run.y), hi = min(excl + mine, run.y);
  for (uint32_t j = lo; j < hi; j++) pairs[run.x + j] = entry_pair_word(entries[run.x + j], (uint32_t)tid);
  if (tid == kExpandThreads - 1)
    for (uint32_t j = hi; j < run.y; j++) pairs[run.x + j] = (uint32_t)tid << 6;
}

// dense slab quarter of a flagged group from its entries (the role k_expand_sorted has for the pair form)
__global__ __launch_bounds__(kExpandThreads) void k_expand_entries(int32_t* __restrict__ coeffs,
                                                                  const uint16_t* __restrict__ entries,
                                                                  const uint8_t* __restrict__ counts,
                                                                  const uint2* __restrict__ runs,
                                                                  const uint8_t* __restrict__ flags) {
  __shared__ __attribute__((aligned(16))) int32_t s_q[kQuarter];
  __shared__ uint32_t s_wsum[kExpandThreads / 64];
  const int q = blockIdx.x & 3, c = (blockIdx.x >> 2) % 3, group = blockIdx.x / 12, tid = threadIdx.x;
  if (!flags[group]) return;
  int4* s4 = reinterpret_cast<int4*>(s_q);
#pragma unroll
  for (int i = 0; i < kQuarter / 4 / kExpandThreads; i++) s4[i * kExpandThreads + tid] = make_int4(0, 0, 0, 0);
  const int gc = group * 3 + c;
  const uint2 run = runs[gc];
  const uint32_t mine = counts[(size_t)gc * kSlotsPerRun + tid];
  const uint32_t excl = block_excl_scan_1024(mine, s_wsum, tid);  // (its barrier also publishes the zero fill)
  if ((tid >> 8) == q) {
    const uint32_t lo = min(excl, run.y), hi = min(excl + mine, run.y);
    for (uint32_t j = lo; j < hi; j++) {
      const uint32_t e = entries[run.x + j];
      atomicAdd(&s_q[(tid & 255) * 64 + (int)(e & 63u)], (int32_t)(e << 16) >> 22);
    }
  }
  __syncthreads();
  int4* d4 = reinterpret_cast<int4*>(coeffs + ((size_t)group * 3 + c) * kGroupArea + q * kQuarter);
#pragma unroll
  for (int i = 0; i < kQuarter / 4 / kExpandThreads; i++) d4[i * kExpandThreads + tid] = s4[i * kExpandThreads + tid];
}
}  // namespace

void launch_unpack_entries12(hipStream_t s, const uint8_t* bytes, size_t n_pairs, uint16_t* entries) {
  if (n_pairs == 0) return;
  hipLaunchKernelGGL(k_unpack_entries12, dim3((unsigned)((n_pairs + 255) / 256)), dim3(256), 0, s, bytes, n_pairs, entries);
}

void launch_entries_to_pairs(hipStream_t s, const uint16_t* entries, const uint8_t* counts, const uint2* runs,
                             const uint8_t* flags, int n_groups, uint32_t* pairs) {
  if (n_groups > 0)
    hipLaunchKernelGGL(k_entries_to_pairs, dim3(n_groups * 3), dim3(kExpandThreads), 0, s, entries, counts, runs, flags, pairs);
}

void launch_expand_entries(hipStream_t s, int32_t* coeffs, const uint16_t* entries, const uint8_t* counts,
                           const uint2* runs, const uint8_t* flags, int n_groups) {
  if (n_groups > 0)
    hipLaunchKernelGGL(k_expand_entries, dim3(n_groups * 12), dim3(kExpandThreads), 0, s, coeffs, entries, counts, runs, flags);
}

void launch_pack_pairs4(hipStream_t s, const uint16_t* entries, const uint16_t* seg_counts, const uint16_t* pos8,
                        const int8_t* val8, const uint32_t* desc, int n_runs, uint32_t* pairs) {
  if (n_runs <= 0) return;
  hipLaunchKernelGGL(k_pack_pairs4, dim3(n_runs), dim3(256), 0, s, entries, seg_counts, pos8, val8, desc, pairs);
}

void launch_pack_pairs8(hipStream_t s, const uint16_t* pos, const int8_t* val, size_t n, uint32_t* pairs) {
  if (n == 0) return;
  hipLaunchKernelGGL(k_pack_pairs8, dim3((unsigned)((n / 4 + 256) / 256)), dim3(256), 0, s, pos, val, n, pairs);
}

}  // namespace jxlh
