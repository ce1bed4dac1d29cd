// Host side of the slot-bucketed coefficient transport (jxlh_submit_groups_slots): plain CPU code, no device, no
// context -- what a decoder thread runs between the entropy loop and the upload.
//
//   jxlh_host_pack_slots     one group's dense slab (the reference's VarDctBuffers::coeffs_storage, frame/group.rs:27-67,
//                            :437-440) -> entries + slot counts.  For callers that keep the reference's dense slabs, and
//                            the measured cost of producing the form from them (bench.py: host_pack_ms_per_frame).
//   jxlh_slot_writer_*       the same form produced by the entropy loop itself: `coeffs[c][offset + order[k]] += v`
//                            (frame/group.rs:557-575) becomes jxlh_slot_writer_add, a varblock's end (`coeffs_offset +=
//                            cx * cy * 64`, :612) flushes its slot buckets.
//
// Values outside the entries' range are SPLIT into repeated in-range entries at the same position: they add up as
// integers on the device before the dequantisation, like `coeffs[i] += v` over several passes (jxl_hip.h), so the frame
// stays in the form the transforms read in place.  Only what no slot can hold (a slot's count is a u8) goes to `wide`,
// which routes that one GROUP through its dense slab.
#include <immintrin.h>  // SSE2 is the baseline (the library also runs on the GPU box's host, whatever CPU that has);
                        // the AVX2 form of the zero scan is picked at run time (__builtin_cpu_supports)

#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/jxl_hip.h"

namespace {

constexpr int kSlots = 1024, kSlotCoeffs = 64, kArea = kSlots * kSlotCoeffs;
constexpr int kMaxSplit = 96;  // entries one coefficient may be split into (10 bits: |v| <= 49 056)

struct Range {
  int lo, hi, mask, shift;
};
inline Range range_of(uint32_t flags) {
  return (flags & JXLH_GROUP_ENTRIES12) ? Range{-32, 31, 63, 6} : Range{-512, 511, 1023, 6};
}
// entries value v takes at position pos (appended at e), 0 if it has to go to `wide`
inline int split_value(int32_t v, uint32_t pos, const Range& r, int room, uint16_t* e) {
  if (v >= r.lo && v <= r.hi) {
    if (room < 1) return 0;
    e[0] = (uint16_t)((pos & 63u) | ((uint32_t)v & (uint32_t)r.mask) << r.shift);
    return 1;
  }
  const int64_t a = v < 0 ? -(int64_t)v : (int64_t)v;
  const int step = v < 0 ? -r.lo : r.hi;  // the larger magnitude on the negative side
  const int64_t k = (a + step - 1) / step;
  if (k > kMaxSplit || k > room) return 0;
  int64_t left = v;
  for (int i = 0; i < (int)k; i++) {
    const int32_t part = (int32_t)(v < 0 ? (left < -step ? -step : left) : (left > step ? step : left));
    e[i] = (uint16_t)((pos & 63u) | ((uint32_t)part & (uint32_t)r.mask) << r.shift);
    left -= part;
  }
  return (int)k;
}

// 12-bit entries, two per three bytes; `e` holds an even number of 16-bit-staged entries
inline void pack12(const uint16_t* e, size_t n, uint8_t* out) {
  for (size_t i = 0; i < n; i += 2) {
    const uint32_t e0 = e[i], e1 = e[i + 1];
    out[0] = (uint8_t)(e0 & 255u);
    out[1] = (uint8_t)((e0 >> 8) | ((e1 & 15u) << 4));
    out[2] = (uint8_t)(e1 >> 4);
    out += 3;
  }
}

// 64-bit map of a slot's non-zero coefficients
inline uint64_t nonzero_map_sse2(const int32_t* p) {
  const __m128i zero = _mm_setzero_si128();
  uint64_t nz = 0;
  for (int q = 0; q < 16; q++) {
    const __m128i v = _mm_loadu_si128(reinterpret_cast<const __m128i*>(p + 4 * q));
    const int m = _mm_movemask_ps(_mm_castsi128_ps(_mm_cmpeq_epi32(v, zero)));
    nz |= (uint64_t)(~m & 15) << (4 * q);
  }
  return nz;
}
__attribute__((target("avx2"))) inline uint64_t nonzero_map_avx2(const int32_t* p) {
  const __m256i zero = _mm256_setzero_si256();
  uint64_t nz = 0;
  for (int q = 0; q < 8; q++) {
    const __m256i v = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(p + 8 * q));
    const int m = _mm256_movemask_ps(_mm256_castsi256_ps(_mm256_cmpeq_epi32(v, zero)));
    nz |= (uint64_t)(~m & 255) << (8 * q);
  }
  return nz;
}
// a channel's 1024 maps in one go (the AVX2 body must not be inlined into baseline code: one call per channel)
void nonzero_maps_sse2(const int32_t* ch, uint64_t* maps) {
  for (int s = 0; s < kSlots; s++) maps[s] = nonzero_map_sse2(ch + s * kSlotCoeffs);
}
__attribute__((target("avx2"))) void nonzero_maps_avx2(const int32_t* ch, uint64_t* maps) {
  for (int s = 0; s < kSlots; s++) maps[s] = nonzero_map_avx2(ch + s * kSlotCoeffs);
}
using MapsFn = void (*)(const int32_t*, uint64_t*);
inline MapsFn pick_maps() {
  // (JXLH_HOST_PACK_NO_AVX2=1: the baseline form on any host -- tests/test_host_pack_cpu.py runs both)
  static const MapsFn fn = [] {
    const char* e = getenv("JXLH_HOST_PACK_NO_AVX2");
    return (!(e && *e && *e != '0') && __builtin_cpu_supports("avx2")) ? nonzero_maps_avx2 : nonzero_maps_sse2;
  }();
  return fn;
}

}  // namespace

struct jxlh_slot_writer {
  Range r;
  uint32_t flags = 0;
  void* entries = nullptr;
  size_t capacity = 0;  // in entries
  uint8_t* counts = nullptr;
  jxlh_coeff32* wide = nullptr;
  uint32_t wide_capacity = 0, n_wide = 0, group_id = 0;
  bool in_group = false, overflow = false;
  // per channel: the run so far, and the pending varblock's updates (several slots: bucketed when it ends)
  std::vector<uint16_t> run[3];
  std::vector<uint32_t> vb_pos[3];
  std::vector<int32_t> vb_val[3];
  uint32_t first_slot = 0, num_slots = 0;
};

namespace {

inline void writer_emit(jxlh_slot_writer* w, int c, uint32_t slot, uint32_t pos, int32_t v) {
  uint16_t tmp[kMaxSplit];
  uint8_t& cnt = w->counts[c * kSlots + slot];
  const int k = split_value(v, pos, w->r, 255 - (int)cnt, tmp);
  if (k == 0) {
    if (w->n_wide < w->wide_capacity) {
      w->wide[w->n_wide].pos = (w->group_id * 3u + (uint32_t)c) * kArea + slot * kSlotCoeffs + (pos & 63u);
      w->wide[w->n_wide].val = v;
      w->n_wide++;
    } else {
      w->overflow = true;
    }
    return;
  }
  cnt = (uint8_t)(cnt + k);
  w->run[c].insert(w->run[c].end(), tmp, tmp + k);
}

// the pending varblock's updates of channel c, slot by slot (a stable counting sort over its slots)
void writer_flush_varblock(jxlh_slot_writer* w) {
  if (w->num_slots <= 1) return;  // one-slot varblocks are emitted as they arrive
  for (int c = 0; c < 3; c++) {
    std::vector<uint32_t>& P = w->vb_pos[c];
    std::vector<int32_t>& V = w->vb_val[c];
    if (P.empty()) continue;
    // varblocks hold 2 .. 1024 slots: bucket heads by counting
    std::vector<uint32_t> head(w->num_slots + 1, 0);
    for (uint32_t p : P) head[(p >> 6) + 1]++;
    for (uint32_t s = 0; s < w->num_slots; s++) head[s + 1] += head[s];
    std::vector<uint32_t> order(P.size());
    {
      std::vector<uint32_t> cur(head.begin(), head.end() - 1);
      for (uint32_t i = 0; i < P.size(); i++) order[cur[P[i] >> 6]++] = i;
    }
    for (uint32_t i : order) writer_emit(w, c, w->first_slot + (P[i] >> 6), P[i], V[i]);
    P.clear();
    V.clear();
  }
}

}  // namespace

extern "C" {

jxlh_status jxlh_host_pack_slots(const int32_t* coeffs, uint32_t group_id, uint32_t flags, void* entries,
                                 size_t entries_capacity, uint8_t* slot_counts, uint32_t n[3], jxlh_coeff32* wide,
                                 uint32_t wide_capacity, uint32_t* n_wide) {
  if (!coeffs || !entries || !slot_counts || !n || (wide_capacity && !wide) || (flags & ~(uint32_t)JXLH_GROUP_ENTRIES12))
    return JXLH_ERR_INVALID_ARGUMENT;
  const Range r = range_of(flags);
  const bool e12 = (flags & JXLH_GROUP_ENTRIES12) != 0;
  uint32_t nw = 0;
  size_t used = 0;  // entries written so far (all channels)
  uint16_t* out16 = static_cast<uint16_t*>(entries);
  std::vector<uint16_t> stage;  // 12-bit form: a channel's run is staged as 16-bit entries, then packed
  if (e12) stage.reserve(kArea / 4);
  const MapsFn maps_of = pick_maps();
  uint64_t maps[kSlots];
  for (int c = 0; c < 3; c++) {
    const int32_t* ch = coeffs + (size_t)c * kArea;
    uint8_t* cnt = slot_counts + c * kSlots;
    size_t run = 0;
    if (e12) stage.clear();
    maps_of(ch, maps);
    for (int s = 0; s < kSlots; s++) {
      const int32_t* p = ch + s * kSlotCoeffs;
      uint64_t nz = maps[s];
      int count = 0;
      while (nz) {
        const int k = __builtin_ctzll(nz);
        nz &= nz - 1;
        const int32_t v = p[k];
        if (!e12 && v >= r.lo && v <= r.hi && count < 255) {  // the usual coefficient: one entry, no staging
          if (used + run + 1 > entries_capacity) return JXLH_ERR_INVALID_ARGUMENT;
          out16[used + run] = (uint16_t)((uint32_t)k | ((uint32_t)v & (uint32_t)r.mask) << r.shift);
          run++;
          count++;
          continue;
        }
        uint16_t tmp[kMaxSplit];
        const int got = split_value(v, (uint32_t)k, r, 255 - count, tmp);
        if (got == 0) {
          if (nw >= wide_capacity) return JXLH_ERR_INVALID_ARGUMENT;
          wide[nw].pos = (group_id * 3u + (uint32_t)c) * kArea + (uint32_t)s * kSlotCoeffs + (uint32_t)k;
          wide[nw].val = p[k];
          nw++;
          continue;
        }
        if (e12) {
          stage.insert(stage.end(), tmp, tmp + got);
        } else {
          if (used + run + got > entries_capacity) return JXLH_ERR_INVALID_ARGUMENT;
          memcpy(out16 + used + run, tmp, (size_t)got * sizeof(uint16_t));
        }
        run += got;
        count += got;
      }
      cnt[s] = (uint8_t)count;
    }
    if (e12) {
      if (run & 1) {  // a run is closed to an even number of entries with a zero update, counted in its last slot
        int s = kSlots - 1;
        while (s > 0 && cnt[s] == 255) s--;
        if (cnt[s] == 255) return JXLH_ERR_INVALID_ARGUMENT;  // (every slot full: 261 120 entries in a 65 536-entry run)
        // the zero update must sit INSIDE slot s's stretch of the run: behind everything of the slots up to s
        size_t at = 0;
        for (int t = 0; t <= s; t++) at += cnt[t];
        stage.insert(stage.begin() + (ptrdiff_t)at, (uint16_t)0);
        cnt[s]++;
        run++;
      }
      if (used + run > entries_capacity) return JXLH_ERR_INVALID_ARGUMENT;
      pack12(stage.data(), run, static_cast<uint8_t*>(entries) + used / 2 * 3);
    }
    n[c] = (uint32_t)run;
    used += run;
  }
  if (n_wide) *n_wide = nw;
  else if (nw) return JXLH_ERR_INVALID_ARGUMENT;
  return JXLH_OK;
}

jxlh_status jxlh_host_pack_slots_many(const int32_t* const* group_coeffs, const uint32_t* group_ids, uint32_t n_groups,
                                      uint32_t flags, void* entries, size_t entries_capacity, uint8_t* slot_counts,
                                      uint32_t* n, jxlh_coeff32* wide, uint32_t wide_capacity, uint32_t* n_wide,
                                      size_t* entries_used) {
  if (!group_coeffs || !group_ids || !entries || !slot_counts || !n || (wide_capacity && !wide) ||
      (flags & ~(uint32_t)JXLH_GROUP_ENTRIES12))
    return JXLH_ERR_INVALID_ARGUMENT;
  const bool e12 = (flags & JXLH_GROUP_ENTRIES12) != 0;
  size_t used = 0;  // entries (every run of the 12-bit form is even: a group starts on a byte)
  uint32_t nw = 0;
  for (uint32_t g = 0; g < n_groups; g++) {
    uint32_t got_wide = 0;
    void* out = e12 ? static_cast<void*>(static_cast<uint8_t*>(entries) + used / 2 * 3)
                    : static_cast<void*>(static_cast<uint16_t*>(entries) + used);
    const jxlh_status st =
        jxlh_host_pack_slots(group_coeffs[g], group_ids[g], flags, out, entries_capacity - used, slot_counts + (size_t)g * 3 * kSlots,
                             n + (size_t)g * 3, wide ? wide + nw : nullptr, wide_capacity - nw, &got_wide);
    if (st != JXLH_OK) return st;
    nw += got_wide;
    used += (size_t)n[g * 3] + n[g * 3 + 1] + n[g * 3 + 2];
  }
  if (n_wide) *n_wide = nw;
  else if (nw) return JXLH_ERR_INVALID_ARGUMENT;
  if (entries_used) *entries_used = used;
  return JXLH_OK;
}

jxlh_status jxlh_slot_writer_create(jxlh_slot_writer** out) {
  if (!out) return JXLH_ERR_INVALID_ARGUMENT;
  jxlh_slot_writer* w = new (std::nothrow) jxlh_slot_writer;
  if (!w) return JXLH_ERR_OUT_OF_MEMORY;
  try {
    for (int c = 0; c < 3; c++) {
      w->run[c].reserve(kArea / 4);
      w->vb_pos[c].reserve(4096);
      w->vb_val[c].reserve(4096);
    }
  } catch (...) {
    delete w;
    return JXLH_ERR_OUT_OF_MEMORY;
  }
  *out = w;
  return JXLH_OK;
}

void jxlh_slot_writer_destroy(jxlh_slot_writer* w) { delete w; }

jxlh_status jxlh_slot_writer_begin_group(jxlh_slot_writer* w, uint32_t group_id, uint32_t flags, void* entries,
                                         size_t entries_capacity, uint8_t* slot_counts, jxlh_coeff32* wide,
                                         uint32_t wide_capacity) {
  if (!w || !entries || !slot_counts || (wide_capacity && !wide) || (flags & ~(uint32_t)JXLH_GROUP_ENTRIES12))
    return JXLH_ERR_INVALID_ARGUMENT;
  w->r = range_of(flags);
  w->flags = flags;
  w->group_id = group_id;
  w->entries = entries;
  w->capacity = entries_capacity;
  w->counts = slot_counts;
  w->wide = wide;
  w->wide_capacity = wide_capacity;
  w->n_wide = 0;
  w->overflow = false;
  w->first_slot = w->num_slots = 0;
  memset(slot_counts, 0, 3 * kSlots);
  for (int c = 0; c < 3; c++) {
    w->run[c].clear();
    w->vb_pos[c].clear();
    w->vb_val[c].clear();
  }
  w->in_group = true;
  return JXLH_OK;
}

jxlh_status jxlh_slot_writer_begin_varblock(jxlh_slot_writer* w, uint32_t first_slot, uint32_t num_slots) {
  if (!w || !w->in_group) return JXLH_ERR_BAD_STATE;
  if (num_slots == 0 || first_slot + num_slots > (uint32_t)kSlots) return JXLH_ERR_INVALID_ARGUMENT;
  try {
    writer_flush_varblock(w);
  } catch (...) {
    return JXLH_ERR_OUT_OF_MEMORY;
  }
  // varblocks arrive in the order of their coefficient offsets (raster order of their first block, group.rs:612): every
  // channel's run then is in slot order by construction
  if (first_slot < w->first_slot + w->num_slots) return JXLH_ERR_INVALID_ARGUMENT;
  w->first_slot = first_slot;
  w->num_slots = num_slots;
  return JXLH_OK;
}

jxlh_status jxlh_slot_writer_add(jxlh_slot_writer* w, uint32_t channel, uint32_t pos, int32_t value) {
  if (!w || !w->in_group || w->num_slots == 0) return JXLH_ERR_BAD_STATE;
  if (channel > 2 || pos >= w->num_slots * (uint32_t)kSlotCoeffs) return JXLH_ERR_INVALID_ARGUMENT;
  if (value == 0) return JXLH_OK;
  try {
    if (w->num_slots == 1) {
      writer_emit(w, (int)channel, w->first_slot, pos, value);
    } else {
      w->vb_pos[channel].push_back(pos);
      w->vb_val[channel].push_back(value);
    }
  } catch (...) {
    return JXLH_ERR_OUT_OF_MEMORY;
  }
  return JXLH_OK;
}

jxlh_status jxlh_slot_writer_add_many(jxlh_slot_writer* w, uint32_t channel, const uint32_t* pos, const int32_t* value,
                                      size_t count) {
  if (count && (!pos || !value)) return JXLH_ERR_INVALID_ARGUMENT;
  for (size_t i = 0; i < count; i++)
    if (jxlh_status st = jxlh_slot_writer_add(w, channel, pos[i], value[i])) return st;
  return JXLH_OK;
}

jxlh_status jxlh_slot_writer_end_group(jxlh_slot_writer* w, uint32_t n[3], uint32_t* n_wide) {
  if (!w || !w->in_group) return JXLH_ERR_BAD_STATE;
  if (!n) return JXLH_ERR_INVALID_ARGUMENT;
  try {
    writer_flush_varblock(w);
  } catch (...) {
    return JXLH_ERR_OUT_OF_MEMORY;
  }
  w->in_group = false;
  w->num_slots = 0;
  if (w->overflow) return JXLH_ERR_INVALID_ARGUMENT;  // more wide values than the caller's list holds
  const bool e12 = (w->flags & JXLH_GROUP_ENTRIES12) != 0;
  size_t used = 0;
  for (int c = 0; c < 3; c++) {
    std::vector<uint16_t>& run = w->run[c];
    if (e12 && (run.size() & 1)) {
      uint8_t* cnt = w->counts + c * kSlots;
      int s = kSlots - 1;
      while (s > 0 && cnt[s] == 255) s--;
      if (cnt[s] == 255) return JXLH_ERR_INVALID_ARGUMENT;
      size_t at = 0;
      for (int t = 0; t <= s; t++) at += cnt[t];
      try {
        run.insert(run.begin() + (ptrdiff_t)at, (uint16_t)0);
      } catch (...) {
        return JXLH_ERR_OUT_OF_MEMORY;
      }
      cnt[s]++;
    }
    if (used + run.size() > w->capacity) return JXLH_ERR_INVALID_ARGUMENT;
    if (e12) pack12(run.data(), run.size(), static_cast<uint8_t*>(w->entries) + used / 2 * 3);
    else if (!run.empty()) memcpy(static_cast<uint16_t*>(w->entries) + used, run.data(), run.size() * sizeof(uint16_t));
    n[c] = (uint32_t)run.size();
    used += run.size();
  }
  if (n_wide) *n_wide = w->n_wide;
  else if (w->n_wide) return JXLH_ERR_INVALID_ARGUMENT;
  return JXLH_OK;
}

}  // extern "C"
