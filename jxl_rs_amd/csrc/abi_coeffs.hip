// C ABI, coefficient transport: dense group slabs (jxlh_submit_group) and the sparse (position, value) forms
//  -- asynchronous H2D on the caller's slot stream, multi-pass accumulation.
#include <algorithm>
#include <vector>

#include "jxlh_ctx.h"

extern "C" {

jxlh_status jxlh_submit_group(jxlh_ctx* ctx, int32_t slot, uint32_t group_id, const int32_t* coeffs, uint32_t flags) {
  JXLH_ON_DEVICE(ctx);
  if (!ctx || !coeffs || slot < 0 || (size_t)slot >= ctx->slots.size()) return JXLH_ERR_INVALID_ARGUMENT;
  if (!ctx->in_frame) return JXLH_ERR_BAD_STATE;
  if (group_id >= ctx->ngroups) return JXLH_ERR_INVALID_ARGUMENT;
  // JXLH_GROUP_COMPLETE is the caller's bookkeeping (set_buffer_for_group's `complete`): a slab always REPLACES the
  // group's coefficients, so a progressive decoder submits what it has accumulated so far (the reference keeps that
  // in Frame::hf_coefficients, frame/decode.rs:547-558) and re-renders the group when a later pass changes it
  if (flags & JXLH_GROUP_ACCUMULATE) return JXLH_ERR_INVALID_ARGUMENT;  // device-side accumulation: sparse form only
  Slot& s = ctx->slots[slot];
  {
    std::lock_guard<std::mutex> lock(ctx->sp_mutex);
    if (ctx->touched[group_id] == 2) {
      // submitted as pairs earlier in this epoch: the dense slab replaces that submission
      for (size_t i = 0; i < ctx->sp_pending.size();) {
        if (ctx->sp_pending[i].group == group_id) ctx->sp_pending.erase(ctx->sp_pending.begin() + i);
        else i++;
      }
    }
    ctx->touched[group_id] = 1;
    if (group_id < ctx->bucketed.size()) ctx->bucketed[group_id] = 0;
    ctx->epoch_dirty = true;
  }
  int32_t* dst = ctx->coeffs.p + (size_t)group_id * 3 * kGroupArea;
  // the previous jxlh_frame_run's transforms may still be reading the slab (callers that use the *_async reads
  // do not wait between frames)
  if (ctx->k1_done_valid) HIPCHK(ctx, hipStreamWaitEvent(s.stream, ctx->k1_done, 0));
  if (dst != coeffs) {
    HIPCHK(ctx, hipMemcpyAsync(dst, coeffs, (size_t)3 * kGroupArea * sizeof(int32_t), hipMemcpyDefault, s.stream));
  }
  s.copied_valid = false;
  HIPCHK(ctx, hipEventRecord(s.done, s.stream));
  s.used = true;
  return JXLH_OK;
}

namespace {
// bookkeeping shared by the sparse submission forms: validates, reserves `total` pairs in the frame's pair buffer
// (offset returned) and records the groups / wide entries for the next jxlh_frame_run
jxlh_status sparse_reserve(jxlh_ctx* ctx, int32_t slot, uint32_t count, const uint32_t* group_ids, const uint32_t* n,
                           const jxlh_coeff32* wide, uint32_t n_wide, uint32_t flags, size_t* offset_out,
                           size_t* total_out) {
  if (!ctx || slot < 0 || (size_t)slot >= ctx->slots.size() || !group_ids || !n || (n_wide && !wide))
    return JXLH_ERR_INVALID_ARGUMENT;
  if (!ctx->in_frame) return JXLH_ERR_BAD_STATE;
  size_t total = 0;
  for (uint32_t i = 0; i < count; i++) {
    if (group_ids[i] >= ctx->ngroups) return JXLH_ERR_INVALID_ARGUMENT;
    total += (size_t)n[3 * i] + n[3 * i + 1] + n[3 * i + 2];
  }
  const size_t wide_limit = ctx->ngroups * 3 * (size_t)kGroupArea;
  for (uint32_t i = 0; i < n_wide; i++)
    if (wide[i].pos >= wide_limit) return JXLH_ERR_INVALID_ARGUMENT;
  std::lock_guard<std::mutex> lock(ctx->sp_mutex);
  const size_t capacity = ctx->ngroups * 3 * (size_t)kGroupArea;  // one pair per coefficient
  if (jxlh_status st = ensure(ctx, ctx->sp_pairs, capacity)) return st;
  if (!ctx->sp_expanded) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->sp_expanded, hipEventDisableTiming));
  if (ctx->sp_used + total > capacity) return JXLH_ERR_INVALID_ARGUMENT;  // more pairs than coefficients
  for (uint32_t i = 0; i < count; i++) {  // one sparse submission per group between two runs (its list may
    if (ctx->touched[group_ids[i]] == 2) return JXLH_ERR_BAD_STATE;  // hold several passes' updates)
    for (uint32_t k = 0; k < i; k++)  // ... and not twice inside this batch either
      if (group_ids[k] == group_ids[i]) return JXLH_ERR_BAD_STATE;
  }
  const size_t offset = ctx->sp_used;
  ctx->sp_used += total;
  size_t o = offset;
  for (uint32_t i = 0; i < count; i++) {
    SparseGroup g;
    g.group = group_ids[i];
    g.offset = (uint32_t)o;
    for (int c = 0; c < 3; c++) {
      g.n[c] = n[3 * i + c];
      o += g.n[c];
    }
    g.flags = (flags & JXLH_GROUP_ACCUMULATE) ? 1u : 0u;
    ctx->sp_pending.push_back(g);
    ctx->touched[g.group] = 2;
    if (g.group < ctx->bucketed.size()) ctx->bucketed[g.group] = 0;  // only jxlh_submit_groups_slots sets it (again)
  }
  ctx->epoch_dirty = true;
  for (uint32_t i = 0; i < n_wide; i++) ctx->sp_wide.push_back(make_uint2(wide[i].pos, (uint32_t)wide[i].val));
  *offset_out = offset;
  *total_out = total;
  return JXLH_OK;
}
}  // namespace

jxlh_status jxlh_submit_groups_sparse(jxlh_ctx* ctx, int32_t slot, uint32_t count, const uint32_t* group_ids,
                                      const jxlh_coeff16* pairs, const uint32_t* n, const jxlh_coeff32* wide,
                                      uint32_t n_wide, uint32_t flags) {
  JXLH_ON_DEVICE(ctx);
  if (count == 0 && ctx && ctx->in_frame) return JXLH_OK;
  size_t offset = 0, total = 0;
  if (!pairs && n && count)  // checked before anything is reserved: a failed call leaves the epoch as it was
    for (size_t r = 0; r < (size_t)count * 3; r++)
      if (n[r]) return JXLH_ERR_INVALID_ARGUMENT;
  if (jxlh_status st = sparse_reserve(ctx, slot, count, group_ids, n, wide, n_wide, flags, &offset, &total)) return st;
  Slot& s = ctx->slots[slot];
  // the pair buffer is recycled per frame: the previous frame's expansion must have read it
  if (ctx->sp_expanded_valid) HIPCHK(ctx, hipStreamWaitEvent(s.stream, ctx->sp_expanded, 0));
  if (total)
    HIPCHK(ctx, hipMemcpyAsync(ctx->sp_pairs.p + offset, pairs, total * sizeof(uint32_t), hipMemcpyDefault, s.stream));
  s.copied_valid = false;
  HIPCHK(ctx, hipEventRecord(s.done, s.stream));
  s.used = true;
  return JXLH_OK;
}

// 3 bytes per coefficient update on the bus: positions and values as separate arrays (u16 / i8), widened into the
// pair buffer by a small kernel on the slot's stream
jxlh_status jxlh_submit_groups_sparse8(jxlh_ctx* ctx, int32_t slot, uint32_t count, const uint32_t* group_ids,
                                       const uint16_t* pos, const int8_t* val, const uint32_t* n,
                                       const jxlh_coeff32* wide, uint32_t n_wide, uint32_t flags) {
  JXLH_ON_DEVICE(ctx);
  if (count == 0 && ctx && ctx->in_frame) return JXLH_OK;
  size_t offset = 0, total = 0;
  if ((!pos || !val) && n && count)  // checked before anything is reserved
    for (size_t r = 0; r < (size_t)count * 3; r++)
      if (n[r]) return JXLH_ERR_INVALID_ARGUMENT;
  if (jxlh_status st = sparse_reserve(ctx, slot, count, group_ids, n, wide, n_wide, flags, &offset, &total)) return st;
  Slot& s = ctx->slots[slot];
  if (ctx->sp_expanded_valid) HIPCHK(ctx, hipStreamWaitEvent(s.stream, ctx->sp_expanded, 0));
  if (total) {
    // staging: [positions | values], reused by the slot (stream-ordered)
    const size_t pos_bytes = (total * sizeof(uint16_t) + 15) & ~(size_t)15;
    if (s.stage8_cap < pos_bytes + total) {
      HIPCHK(ctx, hipStreamSynchronize(s.stream));  // the old staging may still be read by a queued kernel
      if (s.stage8) (void)hipFree(s.stage8);
      s.stage8 = nullptr;
      s.stage8_cap = 0;
      const size_t cap = (pos_bytes + total) * 5 / 4 + 4096;
      if (hipMalloc(reinterpret_cast<void**>(&s.stage8), cap) != hipSuccess) return JXLH_ERR_OUT_OF_MEMORY;
      s.stage8_cap = cap;
    }
    HIPCHK(ctx, hipMemcpyAsync(s.stage8, pos, total * sizeof(uint16_t), hipMemcpyDefault, s.stream));
    HIPCHK(ctx, hipMemcpyAsync(s.stage8 + pos_bytes, val, total, hipMemcpyDefault, s.stream));
    launch_pack_pairs8(s.stream, reinterpret_cast<const uint16_t*>(s.stage8),
                       reinterpret_cast<const int8_t*>(s.stage8 + pos_bytes), total, ctx->sp_pairs.p + offset);
    HIPCHK(ctx, hipGetLastError());
  }
  s.copied_valid = false;
  HIPCHK(ctx, hipEventRecord(s.done, s.stream));
  s.used = true;
  return JXLH_OK;
}

// 2 bytes per coefficient update on the bus: u16 entries = position inside a 4096-coefficient segment | value nibble,
// per-segment counts, and a 3-byte overflow list for the values the nibble does not hold; widened into the pair
// buffer by one workgroup per (group, channel) on the slot's stream
jxlh_status jxlh_submit_groups_sparse4(jxlh_ctx* ctx, int32_t slot, uint32_t count, const uint32_t* group_ids,
                                       const uint16_t* entries, const uint16_t* seg_counts, const uint16_t* pos8,
                                       const int8_t* val8, const uint32_t* n8, const jxlh_coeff32* wide,
                                       uint32_t n_wide, uint32_t flags) {
  JXLH_ON_DEVICE(ctx);
  if (count == 0 && ctx && ctx->in_frame) return JXLH_OK;
  if (!ctx || !seg_counts) return JXLH_ERR_INVALID_ARGUMENT;
  // per (group, channel): entries of the 2-byte form, overflow updates; their sum is what the pair buffer receives
  const size_t runs = (size_t)count * 3;
  std::vector<uint32_t> n(runs), desc(4 * runs);
  size_t tot4 = 0, tot8 = 0;
  for (size_t r = 0; r < runs; r++) {
    uint32_t n4 = 0;
    for (int sgm = 0; sgm < 16; sgm++) n4 += seg_counts[r * 16 + sgm];
    const uint32_t no = n8 ? n8[r] : 0u;
    if (n4 > (uint32_t)kGroupArea || no > (uint32_t)kGroupArea) return JXLH_ERR_INVALID_ARGUMENT;
    desc[4 * r] = (uint32_t)tot4;
    desc[4 * r + 1] = (uint32_t)tot8;
    desc[4 * r + 2] = no;
    n[r] = n4 + no;
    tot4 += n4;
    tot8 += no;
  }
  if ((tot4 && !entries) || (tot8 && (!pos8 || !val8))) return JXLH_ERR_INVALID_ARGUMENT;
  size_t offset = 0, total = 0;
  if (jxlh_status st = sparse_reserve(ctx, slot, count, group_ids, n.data(), wide, n_wide, flags, &offset, &total)) return st;
  {
    size_t o = 0;
    for (size_t r = 0; r < runs; r++) {
      desc[4 * r + 3] = (uint32_t)o;  // relative to the batch's first pair
      o += n[r];
    }
  }
  Slot& s = ctx->slots[slot];
  if (ctx->sp_expanded_valid) HIPCHK(ctx, hipStreamWaitEvent(s.stream, ctx->sp_expanded, 0));
  if (total) {
    // staging: [entries | seg counts | overflow positions | overflow values | run descriptors], reused by the slot
    auto up = [](size_t v) { return (v + 15) & ~(size_t)15; };
    const size_t b_ent = up(tot4 * 2), b_cnt = up(runs * 16 * 2), b_pos = up(tot8 * 2), b_val = up(tot8), b_desc = up(runs * 16);
    const size_t need = b_ent + b_cnt + b_pos + b_val + b_desc;
    if (s.stage8_cap < need) {
      HIPCHK(ctx, hipStreamSynchronize(s.stream));  // the old staging may still be read by a queued kernel
      if (s.stage8) (void)hipFree(s.stage8);
      s.stage8 = nullptr;
      s.stage8_cap = 0;
      const size_t cap = need * 5 / 4 + 4096;
      if (hipMalloc(reinterpret_cast<void**>(&s.stage8), cap) != hipSuccess) return JXLH_ERR_OUT_OF_MEMORY;
      s.stage8_cap = cap;
    }
    uint8_t* d_ent = s.stage8, *d_cnt = d_ent + b_ent, *d_pos = d_cnt + b_cnt, *d_val = d_pos + b_pos, *d_desc = d_val + b_val;
    if (tot4) HIPCHK(ctx, hipMemcpyAsync(d_ent, entries, tot4 * 2, hipMemcpyDefault, s.stream));
    HIPCHK(ctx, hipMemcpyAsync(d_cnt, seg_counts, runs * 16 * 2, hipMemcpyDefault, s.stream));
    if (tot8) {
      HIPCHK(ctx, hipMemcpyAsync(d_pos, pos8, tot8 * 2, hipMemcpyDefault, s.stream));
      HIPCHK(ctx, hipMemcpyAsync(d_val, val8, tot8, hipMemcpyDefault, s.stream));
    }
    // the descriptors are built here: a pageable source is staged by the runtime before the call returns
    HIPCHK(ctx, hipMemcpyAsync(d_desc, desc.data(), runs * 16, hipMemcpyHostToDevice, s.stream));
    launch_pack_pairs4(s.stream, reinterpret_cast<const uint16_t*>(d_ent), reinterpret_cast<const uint16_t*>(d_cnt),
                       reinterpret_cast<const uint16_t*>(d_pos), reinterpret_cast<const int8_t*>(d_val),
                       reinterpret_cast<const uint32_t*>(d_desc), (int)runs, ctx->sp_pairs.p + offset);
    HIPCHK(ctx, hipGetLastError());
  }
  s.copied_valid = false;
  HIPCHK(ctx, hipEventRecord(s.done, s.stream));
  s.used = true;
  return JXLH_OK;
}

// slot-bucketed form: entries, slot counts and run descriptors go to the context's PENDING set as they are (no unpack
// pass, round 5); jxlh_frame_run decides what reads them (run_prologue)
jxlh_status jxlh_submit_groups_slots(jxlh_ctx* ctx, int32_t slot, uint32_t count, const uint32_t* group_ids,
                                     const uint16_t* entries, const uint8_t* slot_counts, const uint32_t* n,
                                     const jxlh_coeff32* wide, uint32_t n_wide, uint32_t flags) {
  JXLH_ON_DEVICE(ctx);
  if (count == 0 && ctx && ctx->in_frame) return JXLH_OK;
  // every argument is checked BEFORE anything is reserved: a failed call leaves the epoch as it was
  if (!ctx || !slot_counts || !n || !group_ids || slot < 0 || (size_t)slot >= ctx->slots.size()) return JXLH_ERR_INVALID_ARGUMENT;
  const size_t runs = (size_t)count * 3;
  const bool e12 = (flags & JXLH_GROUP_ENTRIES12) != 0;
  size_t total_check = 0;
  for (size_t r = 0; r < runs; r++) {
    if (n[r] > (uint32_t)kGroupArea) return JXLH_ERR_INVALID_ARGUMENT;
    if (e12 && (n[r] & 1u)) return JXLH_ERR_INVALID_ARGUMENT;  // 12-bit runs are closed to an even number of entries
    total_check += n[r];
  }
  if (total_check && !entries) return JXLH_ERR_INVALID_ARGUMENT;
  size_t offset = 0, total = 0;
  if (jxlh_status st = sparse_reserve(ctx, slot, count, group_ids, n, wide, n_wide, flags, &offset, &total)) return st;
  int pend;
  {
    std::lock_guard<std::mutex> lock(ctx->sp_mutex);
    pend = ctx->se_live ^ 1;
    // (+ 64 entries: the transforms request a varblock's first entries before they look at its count)
    if (jxlh_status st = ensure(ctx, ctx->se_entries[pend], ctx->ngroups * 3 * (size_t)kGroupArea + 64)) return st;
    if (jxlh_status st = ensure(ctx, ctx->se_counts[pend], ctx->ngroups * 3 * (size_t)kSlotsPerRun)) return st;
    if (jxlh_status st = ensure(ctx, ctx->se_runs[pend], ctx->ngroups * 3)) return st;
    if (ctx->bucketed.size() != ctx->ngroups) ctx->bucketed.assign(ctx->ngroups, 0);
    for (uint32_t i = 0; i < count; i++) ctx->bucketed[group_ids[i]] = 1;
  }
  Slot& s = ctx->slots[slot];
  // the pending set was last read two frames ago (by the transforms of the frame that made it live, or by the previous
  // epoch's widening into the pair buffer): nothing here waits for the frame that is running now
  if (ctx->se_read_valid[pend]) HIPCHK(ctx, hipStreamWaitEvent(s.stream, ctx->se_read[pend], 0));
  if (ctx->sp_expanded_valid) HIPCHK(ctx, hipStreamWaitEvent(s.stream, ctx->sp_expanded, 0));
  uint16_t* d_ent = ctx->se_entries[pend].p + offset;
  if (total) {
    if (!e12) {
      HIPCHK(ctx, hipMemcpyAsync(d_ent, entries, total * sizeof(uint16_t), hipMemcpyDefault, s.stream));
    } else {
      const size_t bytes = total / 2 * 3;
      if (s.stage8_cap < bytes) {
        HIPCHK(ctx, hipStreamSynchronize(s.stream));  // the old staging may still be read by a queued kernel
        if (s.stage8) (void)hipFree(s.stage8);
        s.stage8 = nullptr;
        s.stage8_cap = 0;
        const size_t cap = bytes * 5 / 4 + 4096;
        if (hipMalloc(reinterpret_cast<void**>(&s.stage8), cap) != hipSuccess) return JXLH_ERR_OUT_OF_MEMORY;
        s.stage8_cap = cap;
      }
      HIPCHK(ctx, hipMemcpyAsync(s.stage8, entries, bytes, hipMemcpyDefault, s.stream));
      // (the unpack kernel goes behind the other copies, below: what jxlh_slot_wait / jxlh_slot_after wait for is the
      // copies -- a kernel queued behind another context's transforms would hold the next upload, and the bus, back)
    }
  }
  // counts and run descriptors: one copy per stretch of consecutive group ids (a decoder thread's batch is usually one)
  std::vector<uint2> desc(runs);
  {
    size_t o = offset;
    for (size_t r = 0; r < runs; r++) {
      desc[r] = make_uint2((uint32_t)o, n[r]);
      o += n[r];
    }
  }
  for (uint32_t i0 = 0; i0 < count;) {
    uint32_t i1 = i0 + 1;
    while (i1 < count && group_ids[i1] == group_ids[i1 - 1] + 1) i1++;
    const size_t g0 = group_ids[i0], len = i1 - i0;
    HIPCHK(ctx, hipMemcpyAsync(ctx->se_counts[pend].p + g0 * 3 * kSlotsPerRun, slot_counts + (size_t)i0 * 3 * kSlotsPerRun,
                               len * 3 * kSlotsPerRun, hipMemcpyDefault, s.stream));
    // (built here: a pageable source is staged by the runtime before the call returns)
    HIPCHK(ctx, hipMemcpyAsync(ctx->se_runs[pend].p + g0 * 3, desc.data() + (size_t)i0 * 3, len * 3 * sizeof(uint2),
                               hipMemcpyHostToDevice, s.stream));
    i0 = i1;
  }
  if (total && e12) {
    HIPCHK(ctx, hipEventRecord(s.copied, s.stream));
    launch_unpack_entries12(s.stream, s.stage8, total / 2, d_ent);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipEventRecord(s.done, s.stream));
    s.copied_valid = true;
  } else {
    s.copied_valid = false;
    HIPCHK(ctx, hipEventRecord(s.done, s.stream));
  }
  s.used = true;
  return JXLH_OK;
}

jxlh_status jxlh_submit_group_sparse(jxlh_ctx* ctx, int32_t slot, uint32_t group_id, const jxlh_coeff16* pairs,
                                     const uint32_t n[3], const jxlh_coeff32* wide, uint32_t n_wide,
                                     uint32_t flags) {
  JXLH_ON_DEVICE(ctx);
  if (!ctx || !n) return JXLH_ERR_INVALID_ARGUMENT;
  if (group_id >= ctx->ngroups) return JXLH_ERR_INVALID_ARGUMENT;
  // the single-group form addresses wide entries relative to the group
  std::vector<jxlh_coeff32> w;
  if (n_wide) {
    if (!wide) return JXLH_ERR_INVALID_ARGUMENT;
    w.assign(wide, wide + n_wide);
    for (auto& e : w) {
      if (e.pos >= 3u * kGroupArea) return JXLH_ERR_INVALID_ARGUMENT;
      e.pos += group_id * 3u * kGroupArea;
    }
  }
  return jxlh_submit_groups_sparse(ctx, slot, 1, &group_id, pairs, n, w.data(), n_wide, flags);
}

jxlh_status jxlh_slot_wait(jxlh_ctx* ctx, int32_t slot) {
  JXLH_ON_DEVICE(ctx);
  if (!ctx || slot < 0 || (size_t)slot >= ctx->slots.size()) return JXLH_ERR_INVALID_ARGUMENT;
  Slot& s = ctx->slots[slot];
  if (s.copied_valid) HIPCHK(ctx, hipEventSynchronize(s.copied));  // device work behind the copies is the frame's business
  else HIPCHK(ctx, hipStreamSynchronize(s.stream));
  return JXLH_OK;
}

jxlh_status jxlh_slot_after(jxlh_ctx* ctx, int32_t slot, jxlh_ctx* after_ctx, int32_t after_slot) {
  JXLH_ON_DEVICE(ctx);
  if (!ctx || !after_ctx || slot < 0 || (size_t)slot >= ctx->slots.size() || after_slot < 0 ||
      (size_t)after_slot >= after_ctx->slots.size() || ctx->device != after_ctx->device)
    return JXLH_ERR_INVALID_ARGUMENT;
  const Slot& a = after_ctx->slots[after_slot];
  // (`done` is re-recorded by every submission on that slot: this waits for the latest one recorded so far)
  if (a.used) HIPCHK(ctx, hipStreamWaitEvent(ctx->slots[slot].stream, a.copied_valid ? a.copied : a.done, 0));
  return JXLH_OK;
}

jxlh_status jxlh_frame_coeff_buffer(jxlh_ctx* ctx, int32_t** device_ptr, size_t* n_int32) {
  JXLH_ON_DEVICE(ctx);
  if (!ctx || !device_ptr) return JXLH_ERR_INVALID_ARGUMENT;
  if (!ctx->in_frame) return JXLH_ERR_BAD_STATE;
  *device_ptr = ctx->coeffs.p;
  if (n_int32) *n_int32 = ctx->ngroups * 3 * kGroupArea;
  return JXLH_OK;
}

}  // extern "C"
