// C ABI of the MI355X JPEG XL reconstruction path (include/jxl_hip.h): context, device
// memory, streams, uploads and kernel sequencing.  No compute happens on the host here and
// there is no CPU fallback: every entry point either launches HIP kernels or fails.
//
// Sequencing mirrors the reference's frame flow: decode_lf_group ->
// set_lf* / set_hf_meta; decode_hf_global -> set_dequant_tables; decode_hf_group ->
// submit_group (async H2D on the caller's slot stream, overlapping the host's entropy decode
// of the next group); finalize_lf + render -> frame_run.
#include <algorithm>

#include "jxlh_ctx.h"

namespace jxlh_host {
// the restoration filter's header fields in the form the kernels take them
void set_filter_params(FrameDev& f, const jxlh_frame_params& p) {
  for (int c = 0; c < 3; c++) {  // GaborishStage::new, gaborish.rs:20-27
    const float total = 1.0f + p.gab_w1[c] * 4.0f + p.gab_w2[c] * 4.0f;
    f.gab_k[c][0] = 1.0f / total;
    f.gab_k[c][1] = p.gab_w1[c] / total;
    f.gab_k[c][2] = p.gab_w2[c] / total;
    f.epf_channel_scale[c] = p.epf_channel_scale[c];
  }
  const float sigma_scale[3] = {p.epf_pass0_sigma_scale, 1.0f, p.epf_pass2_sigma_scale};  // render.rs:599-621
  for (int s = 0; s < 3; s++) {
    f.epf_sm[s] = sigma_scale[s] * 1.65f;  // epf1.rs:67-68
    f.epf_bsm[s] = f.epf_sm[s] * p.epf_border_sad_mul;
  }
  f.epf_iters = (int)p.epf_iters;
  f.gab = (int)p.gab;
}
}  // namespace jxlh_host

extern "C" {

uint32_t jxlh_abi_version(void) { return JXLH_ABI_VERSION; }
int32_t jxlh_covered_blocks_x(int32_t t) { return (t >= 0 && t < 27) ? covered_x(t) : -1; }
int32_t jxlh_covered_blocks_y(int32_t t) { return (t >= 0 && t < 27) ? covered_y(t) : -1; }
int32_t jxlh_quant_table_for_type(int32_t t) { return (t >= 0 && t < 27) ? quant_table_for_type(t) : -1; }
int32_t jxlh_quant_table_size(int32_t q) { return (q >= 0 && q < 17) ? quant_table_size(q) : -1; }

const char* jxlh_status_string(jxlh_status s) {
  switch (s) {
    case JXLH_OK: return "ok";
    case JXLH_ERR_INVALID_ARGUMENT: return "invalid argument";
    case JXLH_ERR_OUT_OF_MEMORY: return "out of device memory";
    case JXLH_ERR_DEVICE: return "HIP runtime error";
    case JXLH_ERR_BAD_STATE: return "call order violated";
    case JXLH_ERR_INVALID_TRANSFORM: return "invalid VarDCT transform id";
    case JXLH_ERR_UNSUPPORTED: return "unsupported on the device path";
    case JXLH_ERR_INVALID_BLOCK_SIZE: return "varblock larger than 8x8 in a chroma-subsampled frame";
    case JXLH_ERR_BLOCK_OUT_OF_BOUNDS: return "varblock crosses its group or the frame edge";
    default: return "unknown status";
  }
}

const char* jxlh_last_error(const jxlh_ctx* ctx) { return ctx ? ctx->last_error.c_str() : ""; }

jxlh_status jxlh_default_frame_params(jxlh_frame_params* p, uint32_t xsize, uint32_t ysize) {
  if (!p) return JXLH_ERR_INVALID_ARGUMENT;
  std::memset(p, 0, sizeof *p);
  p->abi_version = JXLH_ABI_VERSION;
  p->xsize = xsize;
  p->ysize = ysize;
  p->global_scale = 21845;  // not a header default; a typical d1 value
  p->quant_lf = 16;         // QuantizerParams::read default branch (quantizer.rs:67)
  p->lf_quant_factors[0] = 1.0f / 4096.0f;  // quant_weights.rs:24-30
  p->lf_quant_factors[1] = 1.0f / 512.0f;
  p->lf_quant_factors[2] = 1.0f / 256.0f;
  p->quant_biases[0] = 1.0f - 0.05465007330715401f;  // headers/transform_data.rs:30-31
  p->quant_biases[1] = 1.0f - 0.07005449891748593f;
  p->quant_biases[2] = 1.0f - 0.049935103337343655f;
  p->quant_biases[3] = 0.145f;
  p->x_qm_scale = 3;  // frame_header.rs:308-315
  p->b_qm_scale = 2;
  p->color_factor = 84;  // color_correlation_map.rs:18, :31-40
  p->base_correlation_x = 0.0f;
  p->base_correlation_b = 1.0f;
  p->gab = 1;  // frame_header.rs:150-177
  for (int c = 0; c < 3; c++) {
    p->gab_w1[c] = 0.115169525f;
    p->gab_w2[c] = 0.061248592f;
  }
  p->epf_iters = 2;
  for (int i = 0; i < 8; i++) p->epf_sharp_lut[i] = (float)i / 7.0f;
  p->epf_sharp_lut[7] = 1.0f;
  p->epf_channel_scale[0] = 40.0f;
  p->epf_channel_scale[1] = 5.0f;
  p->epf_channel_scale[2] = 3.5f;
  p->epf_quant_mul = 0.46f;
  p->epf_pass0_sigma_scale = 0.9f;
  p->epf_pass2_sigma_scale = 6.5f;
  p->epf_border_sad_mul = 2.0f / 3.0f;
  p->do_lf_smoothing = 1;
  p->epf_sigma_for_modular = 1.0f;  // RestorationFilter default (headers/frame_header.rs:229-231)
  return JXLH_OK;
}

jxlh_status jxlh_ctx_create(int32_t device_ordinal, int32_t n_slots, jxlh_ctx** out) {
  if (!out || n_slots < 1 || n_slots > 1024) return JXLH_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device_ordinal < 0 || device_ordinal >= ndev)
    return JXLH_ERR_DEVICE;
  jxlh_ctx* ctx = new (std::nothrow) jxlh_ctx();
  if (!ctx) return JXLH_ERR_OUT_OF_MEMORY;
  ctx->device = device_ordinal;
  // Stream priorities: all streams in the default class.  The runtime maps the streams of one class onto a handful of
  // hardware queues (GPU_MAX_HW_QUEUES, 4), least-used first, and two streams on one queue execute in order: with two
  // contexts of THREE slot streams each the two main streams land on one queue and the frames no longer overlap (8K d1,
  // two frames in flight: 0.74 ms per frame instead of 0.65-0.67; one or two slot streams per context do not collide).
  // Putting the main streams into the high class (or the slot streams) removes that collision (0.67) but makes the
  // PCIe-inclusive legs 1.2-1.7x SLOWER (the unpack kernels on the other class's queues are starved or pre-empt the
  // transforms): measured round 5, profiles/r05_a_stream_queues.txt.  JXLH_STREAM_PRIORITY="<main>,<slot>" selects other
  // classes for A/B runs (hipDeviceGetStreamPriorityRange: -1 high, 0 default, 1 low).
  static const struct Prio { int main_p, slot_p; } prio = [] {
    Prio p{0, 0};
    if (const char* e = getenv("JXLH_STREAM_PRIORITY")) (void)sscanf(e, "%d,%d", &p.main_p, &p.slot_p);
    return p;
  }();
  // JXLH_STREAM_CU_MASK="0x....,0x....,...": test hook -- the context's MAIN stream is created with that CU mask
  // (hipExtStreamCreateWithCUMask; 32 CUs per word): soaks of the dataflow squeeze launch run with fewer CU slots than
  // the launch has tickets, and with its workgroups confined to some XCDs
  std::vector<uint32_t> cu_mask;
  if (const char* e = getenv("JXLH_STREAM_CU_MASK")) {
    for (const char* q = e; *q;) {
      char* end = nullptr;
      cu_mask.push_back((uint32_t)strtoul(q, &end, 0));
      if (end == q) {
        cu_mask.clear();
        break;
      }
      q = *end == ',' ? end + 1 : end;
    }
  }
  if (hipSetDevice(device_ordinal) != hipSuccess ||
      (cu_mask.empty() ? hipStreamCreateWithPriority(&ctx->stream, hipStreamNonBlocking, prio.main_p)
                       : hipExtStreamCreateWithCUMask(&ctx->stream, (uint32_t)cu_mask.size(), cu_mask.data())) != hipSuccess ||
      hipEventCreate(&ctx->t0) != hipSuccess || hipEventCreate(&ctx->t1) != hipSuccess) {
    delete ctx;
    return JXLH_ERR_DEVICE;
  }
  ctx->slots.resize(n_slots);
  for (auto& s : ctx->slots) {
    if (hipStreamCreateWithPriority(&s.stream, hipStreamNonBlocking, prio.slot_p) != hipSuccess ||
        hipEventCreateWithFlags(&s.done, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&s.copied, hipEventDisableTiming) != hipSuccess) {
      jxlh_ctx_destroy(ctx);
      return JXLH_ERR_DEVICE;
    }
  }
  *out = ctx;
  return JXLH_OK;
}

void jxlh_ctx_destroy(jxlh_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipDeviceSynchronize();
  drain_timers(ctx);
  if (ctx->handover) (void)hipEventDestroy(ctx->handover);
  for (auto& s : ctx->slots) {
    if (s.done) (void)hipEventDestroy(s.done);
    if (s.copied) (void)hipEventDestroy(s.copied);
    if (s.stream) (void)hipStreamDestroy(s.stream);
    if (s.stage8) (void)hipFree(s.stage8);
  }
  for (int c = 0; c < 3; c++) {
    release(ctx->planes[c]);
    release(ctx->tmp[c]);
    release(ctx->lf_raw[c]);
    release(ctx->lf_sm[c]);
  }
  release(ctx->sigma);
  release(ctx->tables);
  release(ctx->coeffs);
  release(ctx->sp_pairs);
  release(ctx->sp_groups_dev);
  release(ctx->sp_wide_dev);
  release(ctx->sp_sorted);
  release(ctx->sp_slot_start);
  release(ctx->bucketed_dev);
  release(ctx->route_dev);
  for (int i = 0; i < 2; i++) {
    release(ctx->se_entries[i]);
    release(ctx->se_counts[i]);
    release(ctx->se_runs[i]);
    if (ctx->se_read[i]) (void)hipEventDestroy(ctx->se_read[i]);
  }
  release(ctx->group_dense);
  for (auto& e : ctx->marks)
    if (e) (void)hipEventDestroy(e);
  if (ctx->sp_expanded) (void)hipEventDestroy(ctx->sp_expanded);
  if (ctx->k1_done) (void)hipEventDestroy(ctx->k1_done);
  comm_release(ctx);
  release(ctx->raw_quant);
  release(ctx->lfq);
  release(ctx->transform_map);
  release(ctx->epf_map);
  release(ctx->ytox);
  release(ctx->ytob);
  release(ctx->error_flag);
  release(ctx->tables_ok);
  release(ctx->rgb8);
  for (auto& b : ctx->ups) release(b);
  for (auto& b : ctx->noise) release(b);
  release(ctx->xs_jump);
  for (auto& b : ctx->ups_kernels_n) release(b);
  if (ctx->host_flag) (void)hipHostFree(ctx->host_flag);
  if (ctx->host_flow_flag) (void)hipHostFree(ctx->host_flow_flag);
  release(ctx->flow_words);
  release(ctx->flow_prof);
  release(ctx->worklist);
  for (auto& e : ctx->extra) {
    release(e.raw);
    release(e.f32);
    release(e.out);
  }
  release(ctx->strip_desc);
  release(ctx->strip_mode);
  release(ctx->strip_xchg);
  release(ctx->strip_flags);
  for (auto& b : ctx->hook_f) release(b);
  for (auto& b : ctx->hook_i) release(b);
  if (ctx->t0) (void)hipEventDestroy(ctx->t0);
  if (ctx->t1) (void)hipEventDestroy(ctx->t1);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

jxlh_status jxlh_alloc_pinned(jxlh_ctx* ctx, size_t bytes, void** out) {
  JXLH_ON_DEVICE(ctx);
  if (!ctx || !out) return JXLH_ERR_INVALID_ARGUMENT;
  HIPCHK(ctx, hipHostMalloc(out, bytes, hipHostMallocDefault));
  return JXLH_OK;
}

jxlh_status jxlh_free_pinned(jxlh_ctx* ctx, void* p) {
  JXLH_ON_DEVICE(ctx);
  if (!ctx) return JXLH_ERR_INVALID_ARGUMENT;
  if (p) HIPCHK(ctx, hipHostFree(p));
  return JXLH_OK;
}

// Where the driver places a context's large buffers decides how fast the transforms and the filters run on them: the
// same kernels on the same data are up to 10 % apart between two contexts of one process, persistently, while copies
// between the buffers run alike (profiles/r06_q_context_placement.txt).  With jxlh_ctx_tune_placement(ctx, n) the first
// allocation of {three planes, three filter planes, coefficient buffer} becomes a pick among n candidate sets, rated by
// two byte movers with the streams of the 8x8 transform class and of the filters (k_probe.hip); the candidates are held
// until the pick (so that they ARE different placements), then all but the best are freed.  Setup cost: n x ~2.6 GB of
// transient device memory at 8192^2 and a few ms per candidate; a candidate that cannot be allocated ends the trials.
static jxlh_status choose_placement(jxlh_ctx* ctx, size_t plane_n, size_t tmp_n, size_t coeff_n, size_t probe_elems) {
  struct Cand {
    float* planes[3] = {nullptr, nullptr, nullptr};
    float* tmp[3] = {nullptr, nullptr, nullptr};
    int32_t* coeffs = nullptr;
    float k1 = 0.f, filt = 0.f;
  };
  auto drop = [](Cand& c) {
    for (int i = 0; i < 3; i++) {
      if (c.planes[i]) (void)hipFree(c.planes[i]);
      if (c.tmp[i]) (void)hipFree(c.tmp[i]);
    }
    if (c.coeffs) (void)hipFree(c.coeffs);
    c = Cand();
  };
  std::vector<Cand> cands;
  ctx->placement_report.clear();
  // `trials` candidates, and up to as many again while none of them stands out (the k1-like ratings of the two classes
  // are ~10 % apart and the slow one is ~6 % wide: a best one within 7 % of the worst means every candidate so far is slow)
  const int max_trials = std::min(64, 2 * ctx->placement_trials);
  for (int t = 0; t < max_trials; t++) {
    if (t >= ctx->placement_trials) {
      float lo = cands[0].k1, hi = cands[0].k1;
      for (const Cand& o : cands) lo = std::min(lo, o.k1), hi = std::max(hi, o.k1);
      if (lo <= 0.93f * hi) break;
    }
    Cand c;
    bool ok = true;
    for (int i = 0; i < 3 && ok; i++) {  // (the order of the plain path below)
      ok = hipMalloc(reinterpret_cast<void**>(&c.planes[i]), plane_n * sizeof(float)) == hipSuccess &&
           hipMalloc(reinterpret_cast<void**>(&c.tmp[i]), tmp_n * sizeof(float)) == hipSuccess;
    }
    ok = ok && hipMalloc(reinterpret_cast<void**>(&c.coeffs), coeff_n * sizeof(int32_t)) == hipSuccess;
    if (!ok) {
      (void)hipGetLastError();  // out of memory: the candidates so far are the choice
      drop(c);
      break;
    }
    const jxlh_status st = probe_placement(ctx, c.coeffs, ctx->ngroups, c.planes, c.tmp, probe_elems & ~(size_t)511, &c.k1, &c.filt);
    if (st != JXLH_OK) {
      drop(c);
      for (Cand& o : cands) drop(o);
      return st;
    }
    ctx->placement_report.push_back(c.k1);
    ctx->placement_report.push_back(c.filt);
    cands.push_back(c);
  }
  if (cands.empty()) return JXLH_OK;  // (the plain path reports the allocation failure)
  // the real K1 follows its mover closely (0.283 -> 0.351 ms, 0.303-0.311 -> 0.380-0.387), the filters theirs loosely
  // (slope ~0.3: profiles/r06_q_context_placement.txt): four parts k1-like to one part filter-like
  auto score = [](const Cand& c) { return 4.f * c.k1 + c.filt; };
  size_t best = 0;
  for (size_t i = 1; i < cands.size(); i++)
    if (score(cands[i]) < score(cands[best])) best = i;
  ctx->placement_pick = (int)best;
  for (size_t i = 0; i < cands.size(); i++)
    if (i != best) drop(cands[i]);
  for (int i = 0; i < 3; i++) {
    ctx->planes[i].p = cands[best].planes[i];
    ctx->planes[i].n = plane_n;
    ctx->tmp[i].p = cands[best].tmp[i];
    ctx->tmp[i].n = tmp_n;
  }
  ctx->coeffs.p = cands[best].coeffs;
  ctx->coeffs.n = coeff_n;
  return JXLH_OK;
}

jxlh_status jxlh_frame_begin(jxlh_ctx* ctx, const jxlh_frame_params* p) {
  JXLH_ON_DEVICE(ctx);
  if (!ctx || !p || p->abi_version != JXLH_ABI_VERSION) return JXLH_ERR_INVALID_ARGUMENT;
  if (p->xsize == 0 || p->ysize == 0 || p->xsize > (1u << 20) || p->ysize > (1u << 20) || p->global_scale == 0 ||
      p->quant_lf == 0 || p->color_factor == 0 || p->epf_iters > 3)
    return JXLH_ERR_INVALID_ARGUMENT;
  // chroma subsampling (JPEG recompressions): jpeg_upsampling gives shifts of 0 or 1 per axis and channel,
  // relative to the most finely sampled channel (headers/frame_header.rs:252-253, :501-512)
  if (p->upsampling > 1 && p->upsampling != 2 && p->upsampling != 4 && p->upsampling != 8) return JXLH_ERR_INVALID_ARGUMENT;
  if (p->upsampling > 1) {  // FrameHeader::size() = ceil(size_upsampled / upsampling) (headers/frame_header.rs:555-561)
    const uint32_t n = p->upsampling;
    if (p->xsize_upsampled > p->xsize * n || p->ysize_upsampled > p->ysize * n ||
        (p->xsize_upsampled && (p->xsize_upsampled + n - 1) / n != p->xsize) ||
        (p->ysize_upsampled && (p->ysize_upsampled + n - 1) / n != p->ysize))
      return JXLH_ERR_INVALID_ARGUMENT;
  }
  uint32_t maxhs = 0, maxvs = 0;
  for (int c = 0; c < 3; c++) {
    if (p->hshift[c] > 1 || p->vshift[c] > 1) return JXLH_ERR_INVALID_ARGUMENT;
    maxhs |= p->hshift[c];
    maxvs |= p->vshift[c];
  }
  HIPCHK(ctx, hipSetDevice(ctx->device));
  ctx->params = *p;
  FrameDev& f = ctx->fd;
  std::memset(&f, 0, sizeof f);
  f.xsize = (int)p->xsize;
  f.ysize = (int)p->ysize;
  // FrameHeader::size_blocks (headers/frame_header.rs:564-569): whole blocks of the coarsest channel
  f.xblocks = (int)(((p->xsize + (8u << maxhs) - 1) / (8u << maxhs)) << maxhs);
  f.yblocks = (int)(((p->ysize + (8u << maxvs) - 1) / (8u << maxvs)) << maxvs);
  f.subsampled = (maxhs | maxvs) != 0;
  for (int c = 0; c < 3; c++) {
    f.hshift[c] = (int)p->hshift[c];
    f.vshift[c] = (int)p->vshift[c];
  }
  f.xgroups = (int)((p->xsize + kGroupDim - 1) / kGroupDim);
  f.ygroups = (int)((p->ysize + kGroupDim - 1) / kGroupDim);
  f.cmap_stride = (f.xblocks + 7) / 8;
  f.plane_stride = round_up((size_t)f.xblocks * 8, 64);
  const size_t plane_elems = f.plane_stride * (size_t)f.yblocks * 8;
  // a sharded frame is all-gathered in place with equal counts per rank: room for nranks whole bands
  const size_t gather_elems = (size_t)comm_nranks(ctx) * comm_rows_per_rank(ctx, f.ygroups) * kGroupDim * f.plane_stride;
  // K1 uses 32-bit pixel and coefficient offsets
  // planes are addressed with 32-bit BYTE offsets in the filter kernels, coefficients with 32-bit indices
  if (plane_elems >= (1ull << 30) || (size_t)f.xgroups * f.ygroups * 3 * kGroupArea >= (1ull << 31))
    return JXLH_ERR_UNSUPPORTED;
  ctx->ngroups = (size_t)f.xgroups * f.ygroups;
  {
    std::lock_guard<std::mutex> lock(ctx->sp_mutex);
    ctx->sp_pending.clear();
    ctx->sp_wide.clear();
    ctx->sp_used = 0;
    ctx->touched.assign(ctx->ngroups, 0);
    ctx->bucketed.assign(ctx->ngroups, 0);
    ctx->epoch_dirty = false;
    ctx->sp_sorted_valid = false;
    ctx->se_valid = false;
    ctx->route_live.clear();
    ctx->n_route = 0;
  }
  const size_t nblocks = (size_t)f.xblocks * f.yblocks;
  const size_t ncmap = (size_t)f.cmap_stride * ((f.yblocks + 7) / 8);
  jxlh_status st;
  // jxlh_ctx_tune_placement: the large buffers of a context that has none yet are picked from several candidate sets
  if (ctx->placement_trials > 1 && !ctx->planes[0].p && !ctx->tmp[0].p && !ctx->coeffs.p)
    if ((st = choose_placement(ctx, std::max(plane_elems, gather_elems), std::max(plane_elems + 8 * f.plane_stride, gather_elems),
                               ctx->ngroups * 3 * kGroupArea, plane_elems)) != JXLH_OK)
      return st;
  for (int c = 0; c < 3; c++) {
    if ((st = ensure(ctx, ctx->planes[c], std::max(plane_elems, gather_elems))) != JXLH_OK) return st;
    // + one block row: the scrap tile K1 stores the blocks a sub-sampled channel does not hold into
    if ((st = ensure(ctx, ctx->tmp[c], std::max(plane_elems + 8 * f.plane_stride, gather_elems))) != JXLH_OK) return st;
    if ((st = ensure(ctx, ctx->lf_raw[c], nblocks)) != JXLH_OK) return st;
    if ((st = ensure(ctx, ctx->lf_sm[c], nblocks)) != JXLH_OK) return st;
  }
  if ((st = ensure(ctx, ctx->sigma, nblocks)) != JXLH_OK) return st;
  if ((st = ensure(ctx, ctx->coeffs, ctx->ngroups * 3 * kGroupArea)) != JXLH_OK) return st;
  if ((st = ensure(ctx, ctx->raw_quant, nblocks)) != JXLH_OK) return st;
  if ((st = ensure(ctx, ctx->transform_map, nblocks)) != JXLH_OK) return st;
  if ((st = ensure(ctx, ctx->epf_map, nblocks)) != JXLH_OK) return st;
  if ((st = ensure(ctx, ctx->ytox, ncmap)) != JXLH_OK) return st;
  if ((st = ensure(ctx, ctx->ytob, ncmap)) != JXLH_OK) return st;
  if ((st = ensure(ctx, ctx->error_flag, 1)) != JXLH_OK) return st;
  if ((st = ensure(ctx, ctx->worklist, vardct_worklist_bytes(f))) != JXLH_OK) return st;
  vardct_worklist_reset(ctx->stream, ctx->worklist.p, &ctx->k1_launches);
  HIPCHK(ctx, hipMemsetAsync(ctx->error_flag.p, 0, sizeof(int), ctx->stream));
  // rects the caller never sets read as "not the first block of a varblock" (no work item, no stale map bytes of
  // an earlier frame reaching K1)
  HIPCHK(ctx, hipMemsetAsync(ctx->transform_map.p, 0, nblocks, ctx->stream));
  HIPCHK(ctx, hipMemsetAsync(ctx->raw_quant.p, 0, nblocks * sizeof(int32_t), ctx->stream));
  HIPCHK(ctx, hipMemsetAsync(ctx->epf_map.p, 0, nblocks, ctx->stream));
  for (int c = 0; c < 3; c++) {
    f.planes[c] = ctx->planes[c].p;
    f.tmp[c] = ctx->tmp[c].p;
    f.lf[c] = ctx->lf_raw[c].p;
  }
  f.scrap_off = (int)plane_elems;
  f.coeffs = ctx->coeffs.p;
  f.transform_map = ctx->transform_map.p;
  f.raw_quant = ctx->raw_quant.p;
  f.epf_map = ctx->epf_map.p;
  f.ytox = ctx->ytox.p;
  f.ytob = ctx->ytob.p;
  f.inv_sigma = ctx->sigma.p;
  // scalars, evaluated like the reference does on the host
  f.inv_global_scale = (float)(1 << 16) / (float)p->global_scale;        // quantizer.rs:79-81
  f.x_dm = powf(1.0f / 1.25f, (float)p->x_qm_scale - 2.0f);              // group.rs:395
  f.b_dm = powf(1.0f / 1.25f, (float)p->b_qm_scale - 2.0f);              // group.rs:396
  for (int i = 0; i < 4; i++) f.quant_biases[i] = p->quant_biases[i];
  f.color_factor = (float)p->color_factor;
  f.base_x = p->base_correlation_x;
  f.base_b = p->base_correlation_b;
  set_filter_params(f, *p);
  ctx->in_frame = true;
  // dequant tables persist across frames until replaced (library tables are per-decoder,
  // quant_weights.rs:356-374)
  ctx->tables_set = ctx->tables.p != nullptr && ctx->tables_set;
  if (ctx->tables_set) {
    f.tables = ctx->tables.p;
    f.tables_ok = ctx->tables_ok.p;
    for (int q = 0; q < JXLH_NUM_QUANT_TABLES; q++) f.table_offset[q] = ctx->table_offset[q];
  }
  {
    auto fin = [](float v) { return v == v && v > -3.0e38f && v < 3.0e38f; };
    auto bias_ok = [](float v) { return v >= 1e-6f && v <= 1e6f; };
    f.se_direct_ok = bias_ok(p->quant_biases[0]) && bias_ok(p->quant_biases[1]) && bias_ok(p->quant_biases[2]) &&
                     fin(p->quant_biases[3]) && fin(f.base_x) && fin(f.base_b) && !(p->flags & JXLH_FRAME_DENSE_DEQUANT);
    // ... and the table adjust_quant_bias is read from must hold no zero (AdjTable::nofast; same IEEE operations here)
    for (int i = 2; i < 128 && f.se_direct_ok; i++)
      if ((float)i - p->quant_biases[3] / (float)i == 0.0f) f.se_direct_ok = 0;
    static const bool off = [] { const char* e = getenv("JXLH_NO_DIRECT_ENTRIES"); return e && *e && *e != '0'; }();
    if (off) f.se_direct_ok = 0;  // A/B runs of whole suites
    ctx->params_direct_ok = f.se_direct_ok != 0;
    f.se_direct_ok = ctx->params_direct_ok && ctx->tables_set && ctx->tables_ok_host != 0;
  }
  ctx->lf_smoothed = false;
  ctx->rendered = false;
  ctx->has_special = ctx->has_large = false;
  ctx->strip_all_closed = true;
  ctx->strip_ran = false;
  for (auto& e : ctx->extra) e.set = e.done = false;
  for (auto& s : ctx->slots) s.used = false;
  for (int c = 0; c < 3; c++) ctx->result[c] = nullptr;
  ctx->chroma_lazy = false;
  return JXLH_OK;
}

jxlh_status jxlh_set_upsampling_weights(jxlh_ctx* ctx, const float* weights2, const float* weights4,
                                        const float* weights8) {
  JXLH_ON_DEVICE(ctx);
  if (!ctx) return JXLH_ERR_INVALID_ARGUMENT;
  const float* src[3] = {weights2, weights4, weights8};
  const size_t cnt[3] = {15, 55, 210};
  for (int i = 0; i < 3; i++) {
    if (src[i]) ctx->ups_weights[i].assign(src[i], src[i] + cnt[i]);
    else ctx->ups_weights[i].clear();
    ctx->ups_valid[i] = false;
  }
  return JXLH_OK;
}

jxlh_status jxlh_frame_set_dequant_tables(jxlh_ctx* ctx, const float* const tables[JXLH_NUM_QUANT_TABLES],
                                          const size_t n[JXLH_NUM_QUANT_TABLES]) {
  JXLH_ON_DEVICE(ctx);
  if (!ctx || !tables || !n) return JXLH_ERR_INVALID_ARGUMENT;
  if (!ctx->in_frame) return JXLH_ERR_BAD_STATE;
  size_t total = 0;
  for (int q = 0; q < JXLH_NUM_QUANT_TABLES; q++) {
    if (!tables[q] || n[q] != (size_t)quant_table_size(q)) return JXLH_ERR_INVALID_ARGUMENT;
    total += 3 * n[q];
  }
  jxlh_status st = ensure(ctx, ctx->tables, total);
  if (st != JXLH_OK) return st;
  size_t off = 0;
  for (int q = 0; q < JXLH_NUM_QUANT_TABLES; q++) {
    ctx->fd.table_offset[q] = (int)off;
    ctx->table_offset[q] = (int)off;
    HIPCHK(ctx, hipMemcpyAsync(ctx->tables.p + off, tables[q], 3 * n[q] * sizeof(float), hipMemcpyDefault,
                               ctx->stream));
    off += 3 * n[q];
  }
  ctx->fd.tables = ctx->tables.p;
  if (jxlh_status st2 = ensure(ctx, ctx->tables_ok, 1)) return st2;
  launch_check_tables(ctx->stream, ctx->tables.p, total, ctx->tables_ok.p);
  ctx->fd.tables_ok = ctx->tables_ok.p;
  ctx->tables_ok_host = 0;
  HIPCHK(ctx, hipMemcpyAsync(&ctx->tables_ok_host, ctx->tables_ok.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  ctx->tables_set = true;
  // like the other setters: the caller's buffers may be reused (or freed) as soon as the call returns
  JXLH_SYNC(ctx);
  ctx->fd.se_direct_ok = ctx->params_direct_ok && ctx->tables_ok_host != 0;
  return JXLH_OK;
}

static bool rect_ok(const jxlh_ctx* ctx, uint32_t x0, uint32_t y0, uint32_t w, uint32_t h) {
  return (uint64_t)x0 + w <= (uint64_t)ctx->fd.xblocks && (uint64_t)y0 + h <= (uint64_t)ctx->fd.yblocks;
}

jxlh_status jxlh_frame_set_lf_quantized(jxlh_ctx* ctx, uint32_t x0, uint32_t y0, uint32_t w, uint32_t h,
                                        const int32_t* qy, const int32_t* qx, const int32_t* qb, size_t stride,
                                        uint32_t extra_precision) {
  JXLH_ON_DEVICE(ctx);
  if (!ctx || !qy || !qx || !qb || stride < w || extra_precision > 3) return JXLH_ERR_INVALID_ARGUMENT;
  if (!ctx->in_frame) return JXLH_ERR_BAD_STATE;
  if (!rect_ok(ctx, x0, y0, w, h)) return JXLH_ERR_INVALID_ARGUMENT;
  if (w == 0 || h == 0) return JXLH_OK;
  const size_t n = (size_t)w * h;
  jxlh_status st = ensure(ctx, ctx->lfq, 3 * n);
  if (st != JXLH_OK) return st;
  // the scratch is reused by the next call: order uploads and kernel on the main stream
  const int32_t* src[3] = {qy, qx, qb};
  for (int c = 0; c < 3; c++) {
    st = copy2d(ctx, ctx->lfq.p + c * n, w * sizeof(int32_t), src[c], stride * sizeof(int32_t), w * sizeof(int32_t), h,
                ctx->stream);
    if (st != JXLH_OK) return st;
  }
  const jxlh_frame_params& p = ctx->params;
  // dequant_lf, modular/mod.rs:849-879
  const float inv_quant_lf = (float)(1 << 16) / ((float)p.global_scale * (float)p.quant_lf);
  const float mul = 1.0f / (float)(1u << extra_precision);
  const float fac_x = (p.lf_quant_factors[0] * inv_quant_lf) * mul;
  const float fac_y = (p.lf_quant_factors[1] * inv_quant_lf) * mul;
  const float fac_b = (p.lf_quant_factors[2] * inv_quant_lf) * mul;
  const float cfl_x = p.base_correlation_x + (float)p.ytox_lf / (float)p.color_factor;
  const float cfl_b = p.base_correlation_b + (float)p.ytob_lf / (float)p.color_factor;
  const size_t off = (size_t)y0 * ctx->fd.xblocks + x0;
  {
    ScopedKernelTimer t(ctx, "k0a_dequant_lf");
    if (ctx->fd.subsampled)  // !is444(): no chroma-from-luma; the samples beyond (size >> shift) are don't-cares
      launch_dequant_lf_plain(ctx->stream, ctx->lfq.p, ctx->lfq.p + n, ctx->lfq.p + 2 * n, w, ctx->lf_raw[0].p + off,
                              ctx->lf_raw[1].p + off, ctx->lf_raw[2].p + off, ctx->fd.xblocks, (int)w, (int)h, fac_x,
                              fac_y, fac_b);
    else
      launch_dequant_lf(ctx->stream, ctx->lfq.p, ctx->lfq.p + n, ctx->lfq.p + 2 * n, w, ctx->lf_raw[0].p + off,
                        ctx->lf_raw[1].p + off, ctx->lf_raw[2].p + off, ctx->fd.xblocks, (int)w, (int)h, fac_x, fac_y,
                        fac_b, cfl_x, cfl_b);
  }
  HIPCHK(ctx, hipGetLastError());
  // the host buffers may be reused by the caller as soon as we return
  JXLH_SYNC(ctx);
  ctx->lf_smoothed = false;
  return JXLH_OK;
}

jxlh_status jxlh_frame_set_lf(jxlh_ctx* ctx, uint32_t x0, uint32_t y0, uint32_t w, uint32_t h, const float* x,
                              const float* y, const float* b, size_t stride) {
  JXLH_ON_DEVICE(ctx);
  if (!ctx || !x || !y || !b || stride < w) return JXLH_ERR_INVALID_ARGUMENT;
  if (!ctx->in_frame) return JXLH_ERR_BAD_STATE;
  if (!rect_ok(ctx, x0, y0, w, h)) return JXLH_ERR_INVALID_ARGUMENT;
  const float* src[3] = {x, y, b};
  const size_t off = (size_t)y0 * ctx->fd.xblocks + x0;
  for (int c = 0; c < 3; c++) {
    jxlh_status st = copy2d(ctx, ctx->lf_raw[c].p + off, ctx->fd.xblocks * sizeof(float), src[c],
                            stride * sizeof(float), w * sizeof(float), h, ctx->stream);
    if (st != JXLH_OK) return st;
  }
  JXLH_SYNC(ctx);
  ctx->lf_smoothed = false;
  return JXLH_OK;
}

jxlh_status jxlh_frame_set_hf_meta(jxlh_ctx* ctx, uint32_t x0, uint32_t y0, uint32_t w, uint32_t h,
                                   const uint8_t* transform_map, const int32_t* raw_quant, const uint8_t* epf_map,
                                   size_t map_stride, const int8_t* ytox, const int8_t* ytob, size_t cmap_stride) {
  JXLH_ON_DEVICE(ctx);
  if (!ctx || !transform_map || !raw_quant || !epf_map || !ytox || !ytob || map_stride < w)
    return JXLH_ERR_INVALID_ARGUMENT;
  if (!ctx->in_frame) return JXLH_ERR_BAD_STATE;
  if (!rect_ok(ctx, x0, y0, w, h) || (x0 % 8) || (y0 % 8)) return JXLH_ERR_INVALID_ARGUMENT;
  const size_t cw = (w + 7) / 8, ch = (h + 7) / 8;
  if (cmap_stride < cw) return JXLH_ERR_INVALID_ARGUMENT;
  const size_t off = (size_t)y0 * ctx->fd.xblocks + x0;
  const size_t coff = (size_t)(y0 / 8) * ctx->fd.cmap_stride + x0 / 8;
  // which of the rarely used transform families the frame holds at all (their kernels are not even launched for a
  // frame without them: four empty launches cost ~20 us of a 0.45 ms K1).  A map that arrives in device memory is not
  // inspected: both families are then assumed present.
  if (is_device_ptr(transform_map)) {
    ctx->strip_all_closed = false;
  } else if (ctx->strip_all_closed) {
    // is every varblock a DCT with sides <= 32 inside one 32x32 quadrant of its 64x64 tile?  (rects start on tile
    // boundaries: x0, y0 % 8 == 0)
    bool closed = true;
    for (uint32_t y = 0; y < h && closed; y++) {
      const uint8_t* row = transform_map + (size_t)y * map_stride;
      for (uint32_t x = 0; x < w; x++) {
        if (row[x] < 128) continue;
        const int t = row[x] & 127;
        if (t >= JXLH_NUM_TRANSFORMS || t == 1 || t == 2 || t == 3 || t >= 12 ||
            (x & 3) + (uint32_t)covered_x(t) > 4 || (y & 3) + (uint32_t)covered_y(t) > 4) {
          closed = false;
          break;
        }
      }
    }
    ctx->strip_all_closed = closed;
  }
  if (is_device_ptr(transform_map)) {
    ctx->has_special = ctx->has_large = true;
  } else if (!(ctx->has_special && ctx->has_large)) {
    bool sp = false, lg = false;
    for (uint32_t y = 0; y < h; y++) {
      const uint8_t* row = transform_map + (size_t)y * map_stride;
      for (uint32_t x = 0; x < w; x++) {
        const uint8_t t = row[x] & 127;
        lg |= t >= 18;
        sp |= (t >= 1 && t <= 3) || (t >= 12 && t <= 17);
      }
    }
    ctx->has_special |= sp;
    ctx->has_large |= lg;
  }
  jxlh_status st;
  if ((st = copy2d(ctx, ctx->transform_map.p + off, ctx->fd.xblocks, transform_map, map_stride, w, h, ctx->stream)))
    return st;
  if ((st = copy2d(ctx, ctx->epf_map.p + off, ctx->fd.xblocks, epf_map, map_stride, w, h, ctx->stream))) return st;
  if ((st = copy2d(ctx, ctx->raw_quant.p + off, ctx->fd.xblocks * sizeof(int32_t), raw_quant,
                   map_stride * sizeof(int32_t), w * sizeof(int32_t), h, ctx->stream)))
    return st;
  if ((st = copy2d(ctx, ctx->ytox.p + coff, ctx->fd.cmap_stride, ytox, cmap_stride, cw, ch, ctx->stream))) return st;
  if ((st = copy2d(ctx, ctx->ytob.p + coff, ctx->fd.cmap_stride, ytob, cmap_stride, cw, ch, ctx->stream))) return st;
  JXLH_SYNC(ctx);
  return JXLH_OK;
}

}  // extern "C"

namespace jxlh_host {
namespace {
#include "upsampling_weights.inc"
// Upsample::new (render/stages/upsample.rs:31-66): the 15 / 55 / 210 weights are the upper triangle of the
// symmetric top-left quadrant of the (5n/2 x 2)^2 kernel image; expands to n*n kernels of 5x5 taps
void expand_upsampling_kernels(int n, const float* weights, float* flat) {
  const int half = n / 2, last = n - 1;
  for (int i = 0; i < 5 * half; i++) {
    for (int j = 0; j < 5 * half; j++) {
      const int y = std::min(i, j), x = std::max(i, j);
      const float wv = weights[5 * half * y - y * (y - 1) / 2 + x - y];
      const int oy = j / 5, ox = i / 5, ky = j % 5, kx = i % 5;
      flat[(oy * n + ox) * 25 + ky * 5 + kx] = wv;
      flat[((last - oy) * n + ox) * 25 + (4 - ky) * 5 + kx] = wv;
      flat[(oy * n + (last - ox)) * 25 + ky * 5 + (4 - kx)] = wv;
      flat[((last - oy) * n + (last - ox)) * 25 + (4 - ky) * 5 + (4 - kx)] = wv;
    }
  }
}
}  // namespace

jxlh_status ensure_jump_table(jxlh_ctx* ctx) {
  if (ctx->xs_jump.p) return JXLH_OK;
  static uint64_t table[16][128][2];
  static std::once_flag once;
  std::call_once(once, [] { xorshift_jump_table(table); });
  if (jxlh_status st = ensure(ctx, ctx->xs_jump, sizeof(table) / sizeof(uint64_t))) return st;
  HIPCHK(ctx, hipMemcpyAsync(ctx->xs_jump.p, table, sizeof(table), hipMemcpyHostToDevice, ctx->stream));
  JXLH_SYNC(ctx);
  return JXLH_OK;
}

bool noise_lut_is_zero(const float lut[8]) {
  for (int i = 0; i < 8; i++)
    if (lut[i] != 0.0f) return false;
  return true;
}

// chroma upsampling of the sub-sampled channels, tmp[c] -> planes[c], over the rows K1 produced for group rows
// [gr0, gr1) (the outermost rows of a halo group row read beyond the region, and nobody reads them)
void run_chroma_upsample(jxlh_ctx* ctx, int gr0, int gr1) {
  const FrameDev& f = ctx->fd;
  ScopedKernelTimer t(ctx, "k_chroma_upsample");
  const PixLayout lay = pix_layout(f);
  for (int c = 0; c < 3; c++) {
    const int hs = f.hshift[c], vs = f.vshift[c];
    if (!(hs | vs)) continue;
    const int cw = (f.xsize + (1 << hs) - 1) >> hs, ch = (f.ysize + (1 << vs) - 1) >> vs;
    const int sy0 = (gr0 * kGroupDim) >> vs, sy1 = min(ch, (gr1 * kGroupDim) >> vs);
    launch_chroma_upsample(ctx->stream, f.tmp[c], f.planes[c], lay, lay, hs, vs, cw, ch, sy0, sy1, f.xblocks * 8,
                           f.yblocks * 8);
  }
}

void materialise_chroma(jxlh_ctx* ctx) {
  if (!ctx->chroma_lazy) return;
  run_chroma_upsample(ctx, ctx->lazy_gr0, ctx->lazy_gr1);
  ctx->chroma_lazy = false;
}

jxlh_status upload_upsampling_kernels(jxlh_ctx* ctx, int n) {
  const int slot = n == 2 ? 0 : n == 4 ? 1 : 2;
  const float* dflt = n == 2 ? kDefaultUpsamplingWeights2 : n == 4 ? kDefaultUpsamplingWeights4 : kDefaultUpsamplingWeights8;
  if (!ctx->ups_valid[slot]) {
    const float* w = ctx->ups_weights[slot].empty() ? dflt : ctx->ups_weights[slot].data();
    std::vector<float> flat((size_t)n * n * 25);
    expand_upsampling_kernels(n, w, flat.data());
    if (jxlh_status st = ensure(ctx, ctx->ups_kernels_n[slot], flat.size())) return st;
    // pageable source: the copy is staged by the runtime before the call returns
    HIPCHK(ctx, hipMemcpyAsync(ctx->ups_kernels_n[slot].p, flat.data(), flat.size() * sizeof(float), hipMemcpyHostToDevice,
                               ctx->stream));
    JXLH_SYNC(ctx);
    ctx->ups_valid[slot] = true;
  }
  ctx->ups_kernels.p = ctx->ups_kernels_n[slot].p;  // one buffer per factor: an extra channel's factor may differ from the frame's
  return JXLH_OK;
}

// everything before K1: upload fences, sparse coefficient transport, K0b, K3 sigma
jxlh_status run_prologue(jxlh_ctx* ctx, RunPlan* plan) {
  FrameDev& f = ctx->fd;
  const jxlh_frame_params& p = ctx->params;
  // coefficient uploads issued on slot streams must land before K1
  for (auto& s : ctx->slots) {
    if (s.used) HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, s.done, 0));
  }
  // ---- sparse coefficient transport.  Preferred: K1 reads the pairs itself (bucketed by varblock slot
  // first) -- possible when every group arrived as pairs in this epoch and no value needed the wide
  // list.  Otherwise everything ends up in the dense slabs: groups whose content so far lived only in
  // the bucketed form are expanded from it, this epoch's pairs are zero-filled + scattered.
  bool sparse_k1 = false;
  {
    std::lock_guard<std::mutex> lock(ctx->sp_mutex);
    if (ctx->epoch_dirty) {
      ctx->sp_upload.swap(ctx->sp_pending);  // stays alive until the next run: the H2D copies read it
      ctx->sp_wide_upload.swap(ctx->sp_wide);
      ctx->sp_pending.clear();
      ctx->sp_wide.clear();
      const size_t ng = ctx->sp_upload.size(), nw = ctx->sp_wide_upload.size();
      if (jxlh_status st = ensure(ctx, ctx->sp_groups_dev, ng)) return st;
      if (jxlh_status st = ensure(ctx, ctx->sp_wide_dev, nw)) return st;
      if (jxlh_status st = ensure(ctx, ctx->group_dense, ctx->ngroups)) return st;
      bool all_pairs = nw == 0 && ng == ctx->ngroups && !(p.flags & JXLH_FRAME_EXPAND_SPARSE) && !plan->want_strip;
      for (size_t g = 0; all_pairs && g < ctx->ngroups; g++) all_pairs = ctx->touched[g] == 2;
      // pairs that ADD to a group's earlier passes need that group's dense slab
      std::vector<uint8_t> accum(ctx->ngroups, 0);
      for (const SparseGroup& sg : ctx->sp_upload) {
        if (sg.flags & 1u) {
          accum[sg.group] = 1;
          all_pairs = false;
        }
      }
      // every group arrived slot-bucketed (jxlh_submit_groups_slots): the pending set holds the frame the way the
      // transforms read it -- no sort, no unpacking, no copy
      bool any_bucketed = false, all_bucketed = all_pairs && ctx->bucketed.size() == ctx->ngroups;
      for (size_t g = 0; g < ctx->bucketed.size(); g++) {
        any_bucketed |= ctx->bucketed[g] != 0;
        all_bucketed = all_bucketed && ctx->bucketed[g] != 0;
      }
      const int pend = ctx->se_live ^ 1;
      // Per-group routing (round 6): every group of the frame arrived in this epoch and MOST of them slot-bucketed and
      // self-contained -- the others (a dense slab, plain pairs, a value outside the entries' 10 bits in `wide`, a pass
      // added to earlier content) are brought into their dense slabs and read from there (FrameDev::group_route), the
      // bucketed ones are still read in place.  Round 5 took the whole frame out of the in-place form for one such group.
      std::vector<uint8_t> route(ctx->ngroups, 1);
      size_t n_inplace = 0;
      bool mixed = false;
      if (any_bucketed && !all_bucketed && ctx->bucketed.size() == ctx->ngroups &&
          !(p.flags & JXLH_FRAME_EXPAND_SPARSE) && !plan->want_strip) {
        bool all_touched = true;
        for (size_t g = 0; g < ctx->ngroups; g++) all_touched = all_touched && ctx->touched[g] != 0;
        if (all_touched) {
          std::vector<uint8_t> wide_group(ctx->ngroups, 0);
          for (const uint2& w : ctx->sp_wide_upload) wide_group[w.x / (3u * kGroupArea)] = 1;  // (validated at submission)
          for (size_t g = 0; g < ctx->ngroups; g++) {
            route[g] = !(ctx->touched[g] == 2 && ctx->bucketed[g] && !accum[g] && !wide_group[g]);
            n_inplace += route[g] ? 0 : 1;
          }
          mixed = 2 * n_inplace >= ctx->ngroups;
        }
      }
      // (the group list is read by the sort and by the expansion: a frame that arrived slot-bucketed needs neither)
      if (ng && !all_bucketed && !mixed)  // (mixed: only the routed groups' descriptors, below)
        HIPCHK(ctx, hipMemcpyAsync(ctx->sp_groups_dev.p, ctx->sp_upload.data(), ng * sizeof(SparseGroup),
                                   hipMemcpyHostToDevice, ctx->stream));
      if (nw)
        HIPCHK(ctx, hipMemcpyAsync(ctx->sp_wide_dev.p, ctx->sp_wide_upload.data(), nw * sizeof(uint2),
                                   hipMemcpyHostToDevice, ctx->stream));
      if (ctx->sp_sorted_valid && !all_pairs) {
        // leaving the bucketed form: groups not resubmitted now (or only added to) need their dense slab -- unless
        // the slab already was where they lived (a dense-route group of a frame with per-group routing)
        ctx->flag_upload.assign(ctx->ngroups, 0);
        bool any = false;
        const bool had_routes = ctx->route_live.size() == ctx->ngroups;
        for (size_t g = 0; g < ctx->ngroups; g++)
          any |= (ctx->flag_upload[g] = ((ctx->touched[g] == 0 || accum[g]) && !(had_routes && ctx->route_live[g])) ? 1 : 0) != 0;
        if (any) {
          HIPCHK(ctx, hipMemcpyAsync(ctx->group_dense.p, ctx->flag_upload.data(), ctx->ngroups, hipMemcpyHostToDevice,
                                     ctx->stream));
          ScopedKernelTimer t(ctx, "k_expand_sparse");
          if (ctx->se_valid)
            launch_expand_entries(ctx->stream, ctx->coeffs.p, ctx->se_entries[ctx->se_live].p, ctx->se_counts[ctx->se_live].p,
                                  ctx->se_runs[ctx->se_live].p, ctx->group_dense.p, (int)ctx->ngroups);
          else
            launch_expand_sorted(ctx->stream, ctx->coeffs.p, ctx->sp_sorted.p, ctx->sp_slot_start.p, ctx->group_dense.p,
                                 (int)ctx->ngroups);
        }
      }
      if (any_bucketed && !all_bucketed) {
        // a mixed epoch (other groups as plain pairs or dense slabs, a wide entry, an added pass): the slot-bucketed
        // groups' entries become pair words at their reserved places of the pair buffer and take the general route
        if (jxlh_status st = ensure(ctx, ctx->bucketed_dev, ctx->ngroups)) return st;
        ctx->bucketed_upload = ctx->bucketed;  // stays alive until the next epoch: the copy reads it
        bool any_widened = !mixed;
        if (mixed)  // only the bucketed groups that leave the in-place form (often none: the routed group is a dense slab)
          for (size_t g = 0; g < ctx->ngroups; g++) any_widened |= (ctx->bucketed_upload[g] = ctx->bucketed[g] && route[g]) != 0;
        if (any_widened) {
          HIPCHK(ctx, hipMemcpyAsync(ctx->bucketed_dev.p, ctx->bucketed_upload.data(), ctx->ngroups, hipMemcpyHostToDevice,
                                     ctx->stream));
          ScopedKernelTimer t(ctx, "k_entries_to_pairs");
          launch_entries_to_pairs(ctx->stream, ctx->se_entries[pend].p, ctx->se_counts[pend].p, ctx->se_runs[pend].p,
                                  ctx->bucketed_dev.p, (int)ctx->ngroups, ctx->sp_pairs.p);
        }
      }
      if (all_bucketed || mixed) {
        // entries per coefficient of the groups that are read in place: from about three times d1's share (0.086 on the
        // synthetic frame) the 8x8 class is better off running its over-depth batches inline (FrameDev::se_dense_hint;
        // K1 at x1 / x2 / x4 density: 0.301 / 0.412 / 0.582 ms without, 0.320 / 0.424 / 0.557 with: profiles/r06_c_density.txt)
        uint64_t entries = 0, groups = 0;
        for (const SparseGroup& sg : ctx->sp_upload) {
          if (mixed && route[sg.group]) continue;
          entries += (uint64_t)sg.n[0] + sg.n[1] + sg.n[2];
          groups++;
        }
        // ... and from about 1.5 times d1's share most 16..32-point batches are beyond the direct path's depth: the
        // dense dequantisation pass for those classes outright instead of a direct launch that rejects them and a
        // fallback launch that picks them up (level 1; K1 at x2: 0.405 -> see profiles/r06_p_density.txt)
        const double share = groups ? (double)entries / (double)(groups * 3 * (uint64_t)kGroupArea) : 0.0;
        ctx->se_dense_hint = share > 0.25 ? 2 : share > 0.125 ? 1 : 0;
      }
      if (all_bucketed) {
        ctx->se_live = pend;  // the sets trade places: the next epoch's uploads go to the set read two frames ago
        ctx->se_valid = true;
        ctx->sp_sorted_valid = true;
        ctx->route_live.clear();
        ctx->n_route = 0;
      } else if (mixed) {
        // the routed groups' slabs: zero-fill + scatter of their pair words (their own, or the ones the widening above
        // made of their entries) + the wide values -- or, for an added pass, on top of what the slab holds
        if (jxlh_status st = ensure(ctx, ctx->route_dev, ctx->ngroups)) return st;
        ctx->route_upload = route;  // stays alive until the next epoch: the copy reads it
        HIPCHK(ctx, hipMemcpyAsync(ctx->route_dev.p, ctx->route_upload.data(), ctx->ngroups, hipMemcpyHostToDevice, ctx->stream));
        // (only the routed groups that arrived as pairs / entries have anything to expand: their descriptors are moved to
        // the front of the list -- a frame whose only routed group is a dense slab launches nothing here)
        size_t n_expand = 0;
        for (size_t i = 0; i < ng; i++)
          if (route[ctx->sp_upload[i].group]) std::swap(ctx->sp_upload[n_expand++], ctx->sp_upload[i]);
        if (n_expand)
          HIPCHK(ctx, hipMemcpyAsync(ctx->sp_groups_dev.p, ctx->sp_upload.data(), n_expand * sizeof(SparseGroup),
                                     hipMemcpyHostToDevice, ctx->stream));
        if (n_expand || nw) {
          ScopedKernelTimer t(ctx, "k_expand_sparse");
          launch_expand_sparse(ctx->stream, ctx->coeffs.p, ctx->sp_pairs.p, ctx->sp_groups_dev.p, (int)n_expand,
                               ctx->sp_wide_dev.p, (uint32_t)nw, nullptr);
        }
        ctx->se_live = pend;
        ctx->se_valid = true;
        ctx->sp_sorted_valid = true;
        ctx->route_live = route;
        ctx->n_route = (int)(ctx->ngroups - n_inplace);
      } else if (all_pairs) {
        const size_t capacity = ctx->ngroups * 3 * (size_t)kGroupArea;
        if (jxlh_status st = ensure(ctx, ctx->sp_sorted, capacity)) return st;
        if (jxlh_status st = ensure(ctx, ctx->sp_slot_start, ctx->ngroups * 3 * (size_t)kSlotTable)) return st;
        {
          ScopedKernelTimer t(ctx, "k_sort_sparse");
          launch_sort_sparse(ctx->stream, ctx->sp_pairs.p, ctx->sp_groups_dev.p, (int)ng, ctx->sp_sorted.p,
                             ctx->sp_slot_start.p);
        }
        ctx->se_valid = false;
        ctx->sp_sorted_valid = true;
        ctx->route_live.clear();
        ctx->n_route = 0;
      } else {
        ctx->route_live.clear();
        ctx->n_route = 0;
        if (ng || nw) {
          ScopedKernelTimer t(ctx, "k_expand_sparse");
          launch_expand_sparse(ctx->stream, ctx->coeffs.p, ctx->sp_pairs.p, ctx->sp_groups_dev.p, (int)ng,
                               ctx->sp_wide_dev.p, (uint32_t)nw, nullptr);
        }
        ctx->se_valid = false;
        ctx->sp_sorted_valid = false;
      }
      if (any_bucketed && !all_bucketed && !mixed) {  // the pending set has been read (it stays the pending one)
        if (!ctx->se_read[pend]) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->se_read[pend], hipEventDisableTiming));
        HIPCHK(ctx, hipEventRecord(ctx->se_read[pend], ctx->stream));
        ctx->se_read_valid[pend] = true;
      }
      ctx->sp_used = 0;
      ctx->bucketed.assign(ctx->ngroups, 0);
      ctx->touched.assign(ctx->ngroups, 0);
      ctx->epoch_dirty = false;
      if (ctx->sp_expanded) {  // the pair buffer has been consumed (bucketed or expanded)
        HIPCHK(ctx, hipEventRecord(ctx->sp_expanded, ctx->stream));
        ctx->sp_expanded_valid = true;
      }
    }
    sparse_k1 = ctx->sp_sorted_valid;
  }
  plan->sparse_k1 = sparse_k1;
  // ---- K0b: Frame::finalize_lf (frame/mod.rs:360-378)
  const bool smooth = p.do_lf_smoothing && f.xblocks > 2 && f.yblocks > 2;  // adaptive_lf_smoothing.rs:51-53
  if (smooth) {
    {  // out of place (raw -> smoothed), so re-running a frame repeats the full work
      const float inv_quant_lf = f.inv_global_scale / (float)p.quant_lf;  // quantizer.rs:82-84
      const float lf_factors[3] = {inv_quant_lf * p.lf_quant_factors[0], inv_quant_lf * p.lf_quant_factors[1],
                                   inv_quant_lf * p.lf_quant_factors[2]};
      const float* in[3] = {ctx->lf_raw[0].p, ctx->lf_raw[1].p, ctx->lf_raw[2].p};
      float* out[3] = {ctx->lf_sm[0].p, ctx->lf_sm[1].p, ctx->lf_sm[2].p};
      ScopedKernelTimer t(ctx, "k0b_lf_smooth");
      launch_lf_smooth(ctx->stream, in, out, f.xblocks, f.yblocks, lf_factors);
      ctx->lf_smoothed = true;
    }
    for (int c = 0; c < 3; c++) f.lf[c] = ctx->lf_sm[c].p;
  } else {
    for (int c = 0; c < 3; c++) f.lf[c] = ctx->lf_raw[c].p;
  }
  // ---- K3 sigma: SigmaSource::new (features/epf.rs:35-87)
  if (f.epf_iters > 0) {
    ScopedKernelTimer t(ctx, "k3_sigma_map");
    launch_sigma_map(ctx->stream, f, p.epf_quant_mul, p.epf_sharp_lut);
  }
  plan->halo_px = (f.gab ? 1 : 0) + (f.epf_iters >= 3 ? 3 : 0) + (f.epf_iters >= 1 ? 2 : 0) + (f.epf_iters >= 2 ? 1 : 0);
  // K1 writes the 8x8-tiled layout whenever the fused filter kernel is its only consumer
  plan->will_fuse = !(p.flags & JXLH_FRAME_UNFUSED_FILTERS) && (f.gab || f.epf_iters > 0);
  f.tiled = plan->will_fuse ? 1 : 0;
  plan->want_strip = plan->want_strip && !sparse_k1;
  return JXLH_OK;
}

// what the transforms read when the frame is resident in a bucketed sparse form: the slot-bucketed entries in place
// (se_*) or the sorted pair words (sp_sorted); all null = the dense slabs
static void set_sparse_view(jxlh_ctx* ctx, FrameDev& f, bool sparse_k1) {
  const bool ent = sparse_k1 && ctx->se_valid;
  f.sp_sorted = sparse_k1 && !ent ? ctx->sp_sorted.p : nullptr;
  f.sp_slot_start = sparse_k1 && !ent ? ctx->sp_slot_start.p : nullptr;
  f.se_entries = ent ? ctx->se_entries[ctx->se_live].p : nullptr;
  f.se_counts = ent ? ctx->se_counts[ctx->se_live].p : nullptr;
  f.se_runs = ent ? ctx->se_runs[ctx->se_live].p : nullptr;
  f.group_dense = sparse_k1 ? ctx->group_dense.p : nullptr;
  f.group_route = ent && ctx->n_route > 0 ? ctx->route_dev.p : nullptr;
  f.se_dense_hint = ent ? ctx->se_dense_hint : 0;
  f.k1_stats = ctx->timing ? 1 : 0;
}
static int dense_route_groups(const jxlh_ctx* ctx, bool sparse_k1) { return sparse_k1 && ctx->se_valid ? ctx->n_route : 0; }
// behind the transforms: the coefficient slabs are free again (dense resubmissions of the next frame wait for this,
// jxlh_submit_group), and so is the live set of the slot-bucketed form once it has become the pending one
static jxlh_status mark_coefficients_read(jxlh_ctx* ctx, bool sparse_k1) {
  if (!ctx->k1_done) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->k1_done, hipEventDisableTiming));
  HIPCHK(ctx, hipEventRecord(ctx->k1_done, ctx->stream));
  ctx->k1_done_valid = true;
  if (sparse_k1 && ctx->se_valid) {
    const int l = ctx->se_live;
    if (!ctx->se_read[l]) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->se_read[l], hipEventDisableTiming));
    HIPCHK(ctx, hipEventRecord(ctx->se_read[l], ctx->stream));
    ctx->se_read_valid[l] = true;
  }
  return JXLH_OK;
}

// K1 for group rows [gr0, gr1) (+ the chroma upsampling of a sub-sampled frame)
jxlh_status run_k1(jxlh_ctx* ctx, const RunPlan& plan, int gr0, int gr1) {
  FrameDev& f = ctx->fd;
  const jxlh_frame_params& p = ctx->params;
  const bool sparse_k1 = plan.sparse_k1;
  {
    ScopedKernelTimer t(ctx, "k1_vardct");
    set_sparse_view(ctx, f, sparse_k1);
    // (a whole-frame run rewrites every group's flag in k1_scan: no clearing launch then)
    if (sparse_k1 && !(gr0 == 0 && gr1 == f.ygroups))
      HIPCHK(ctx, hipMemsetAsync(ctx->group_dense.p, 0, ctx->ngroups, ctx->stream));
    // a sub-sampled channel is reconstructed at its own resolution into tmp[c] ...
    FrameDev fk = f;
    for (int c = 0; c < 3; c++)
      if (f.hshift[c] | f.vshift[c]) fk.planes[c] = f.tmp[c];
    launch_vardct_groups(ctx->stream, fk, gr0, gr1, ctx->worklist.p, &ctx->k1_launches, ctx->error_flag.p,
                         sparse_k1 ? ctx->coeffs.p : nullptr, nullptr, 0, ctx->has_special, ctx->has_large,
                         dense_route_groups(ctx, sparse_k1));
  }
  if (jxlh_status st = mark_coefficients_read(ctx, sparse_k1)) return st;
  ctx->chroma_lazy = false;
  if (f.subsampled) {
    // ... and brought to full resolution into planes[c] before any filter (frame/render.rs:569-576) -- or, when no
    // stage follows at all, only when the planes are asked for (materialise_chroma)
    const bool stages_follow = f.gab || f.epf_iters > 0 || p.upsampling > 1 || (p.noise && !noise_lut_is_zero(p.noise_lut));
    ctx->lazy_gr0 = gr0;
    ctx->lazy_gr1 = gr1;
    // A sharded frame gathers planes[c] band by band (jxlh_frame_allgather): the full-resolution chroma must exist
    // on every rank before the gather, and a deferred upsampling would cover only this rank's band afterwards.
    if (stages_follow || jxlh_host::comm_nranks(ctx) > 1) run_chroma_upsample(ctx, gr0, gr1);
    else ctx->chroma_lazy = true;
  }
  return JXLH_OK;
}

// Whole-frame runs of a 4:4:4 frame with Gaborish and / or EPF1 (+ EPF2) may go through the strip kernel (k_strip.hip):
// opt-in (JXLH_FRAME_STRIP), see the flag's comment in jxl_hip.h
bool strip_eligible(const jxlh_ctx* ctx) {
  const FrameDev& f = ctx->fd;
  const jxlh_frame_params& p = ctx->params;
  static const bool forced = [] {  // JXLH_STRIP=1: every eligible frame, whatever its flags say (A/B runs of whole suites)
    const char* e = getenv("JXLH_STRIP");
    return e && *e && *e != '0';
  }();
  return (forced || (p.flags & JXLH_FRAME_STRIP)) && !(p.flags & JXLH_FRAME_UNFUSED_FILTERS) && !f.subsampled &&
         f.epf_iters <= 2 && (f.gab || f.epf_iters >= 1) && comm_nranks(ctx) <= 1;
}

// transforms + stage list of the whole frame in the strip kernel; tiles it cannot take (k1_scan decides) go through
// K1's class kernels first
jxlh_status run_strip(jxlh_ctx* ctx, const RunPlan& plan) {
  FrameDev& f = ctx->fd;
  (void)plan;
  if (!ctx->cu_count) {
    hipDeviceProp_t prop;
    HIPCHK(ctx, hipGetDeviceProperties(&prop, ctx->device));
    ctx->cu_count = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  const int strips = strip_strips(f), tile_rows = strip_tile_rows(f);
  // The strips of a band spin on their neighbours' progress flags: every workgroup of the launch must be resident.
  // The occupancy query says how many are (two per CU on MI355X: the 77 KB window); a frame wider than that in strips
  // cannot take this path at all (the grid used to assume two per CU).
  // (per context: contexts may sit on devices of different sizes; a failed query is asked again -- ADVICE r05)
  if (ctx->strip_resident <= 0) ctx->strip_resident = strip_resident_workgroups(ctx->cu_count);
  const int resident = ctx->strip_resident;
  if (resident < strips) return JXLH_ERR_UNSUPPORTED;  // (the caller falls back to the two-kernel path)
  // a band is at least 4 tile rows (two extra transforms per band and strip)
  int bands = std::min((2 * ctx->cu_count + strips / 2) / strips, resident / strips);
  bands = std::max(1, std::min(bands, std::max(1, tile_rows / 4)));
  static const int forced_bands = [] {
    const char* e = getenv("JXLH_STRIP_BANDS");
    return e ? atoi(e) : 0;
  }();
  if (forced_bands > 0) bands = std::min(std::min(forced_bands, tile_rows), std::max(1, resident / strips));
  const size_t nblocks = (size_t)f.xblocks * f.yblocks;
  const bool fresh = ctx->strip_desc.n < nblocks;
  if (jxlh_status st = ensure(ctx, ctx->strip_desc, nblocks)) return st;
  if (jxlh_status st = ensure(ctx, ctx->strip_mode, (size_t)strips * tile_rows)) return st;
  if (jxlh_status st = ensure(ctx, ctx->strip_xchg, strip_xchg_floats(f))) return st;
  if (jxlh_status st = ensure(ctx, ctx->strip_flags, strip_flag_ints(f, bands))) return st;
  // blocks no varblock covers (rects never set, broken maps) keep an invalid descriptor
  if (fresh) HIPCHK(ctx, hipMemsetAsync(ctx->strip_desc.p, 0, nblocks * sizeof(uint2), ctx->stream));
  f.strip_desc = ctx->strip_desc.p;
  f.strip_mode = ctx->strip_mode.p;
  f.strip_flags = ctx->strip_flags.p;
  f.strip_nflags = (int)strip_flag_ints(f, bands);
  f.strips = strips;
  f.tile_rows = tile_rows;
  f.strip_all_closed = ctx->strip_all_closed ? 1 : 0;
  f.tiled = 1;
  set_sparse_view(ctx, f, false);
  {
    ScopedKernelTimer t(ctx, "k1_vardct");
    launch_vardct_groups(ctx->stream, f, 0, f.ygroups, ctx->worklist.p, &ctx->k1_launches, ctx->error_flag.p, nullptr,
                         nullptr, 0, ctx->has_special, ctx->has_large, 0);
  }
  {
    ScopedKernelTimer t(ctx, "k123_strip");
    static const float deadline_s = [] {
      const char* e = getenv("JXLH_STRIP_DEADLINE_S");
      return e ? (float)atof(e) : 4.0f;
    }();
    if (!launch_strip(ctx->stream, f, f.strip_desc, f.strip_mode, ctx->strip_xchg.p, ctx->strip_flags.p, bands,
                      ctx->error_flag.p, deadline_s)) {
      f.strip_desc = nullptr;
      f.strip_mode = nullptr;
      return JXLH_ERR_UNSUPPORTED;  // stage list not covered (strip_eligible should have said so)
    }
  }
  f.strip_desc = nullptr;  // band runs / re-renders of this frame take the two-kernel path
  f.strip_mode = nullptr;
  if (!ctx->k1_done) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->k1_done, hipEventDisableTiming));
  HIPCHK(ctx, hipEventRecord(ctx->k1_done, ctx->stream));
  ctx->k1_done_valid = true;
  ctx->chroma_lazy = false;
  ctx->rendered = true;
  ctx->strip_ran = true;
  // upsampling / noise behind the stage list: the strip kernel leaves the filtered planes in f.tmp
  return run_post_stages(ctx, f.tmp, 0, f.ysize, true);
}

// the stage list on group rows [group_row0, group_row1), then upsampling and noise
jxlh_status run_stages(jxlh_ctx* ctx, const RunPlan& plan, uint32_t group_row0, uint32_t group_row1) {
  const bool whole = group_row0 == 0 && group_row1 == (uint32_t)ctx->fd.ygroups;
  return run_stages_rows(ctx, plan, (int)group_row0 * kGroupDim, min((int)group_row1 * kGroupDim, ctx->fd.ysize), whole);
}

// ... on pixel rows [y_lo, y_hi)
jxlh_status run_stages_rows(jxlh_ctx* ctx, const RunPlan& plan, int y_lo, int y_hi, bool whole_frame) {
  FrameDev& f = ctx->fd;
  const jxlh_frame_params& p = ctx->params;
  (void)plan;
  // ---- stage list of frame/render.rs:569-622
  int stages[4], borders[4], ns = 0;
  if (f.gab) { stages[ns] = -1; borders[ns++] = 1; }
  if (f.epf_iters >= 3) { stages[ns] = 0; borders[ns++] = 3; }
  if (f.epf_iters >= 1) { stages[ns] = 1; borders[ns++] = 2; }
  if (f.epf_iters >= 2) { stages[ns] = 2; borders[ns++] = 1; }
  float* cur[3] = {f.planes[0], f.planes[1], f.planes[2]};
  float* oth[3] = {f.tmp[0], f.tmp[1], f.tmp[2]};
  if (!(p.flags & JXLH_FRAME_UNFUSED_FILTERS) && ns > 0) {
    // production path: the whole stage list in one pass over HBM (two for epf_iters == 3)
    ScopedKernelTimer t(ctx, "k23_fused_filters");
    const int where = launch_fused_filters(ctx->stream, f, y_lo, y_hi);
    if (where == 1) {
      for (int c = 0; c < 3; c++) {
        cur[c] = f.tmp[c];
        oth[c] = f.planes[c];
      }
      ns = 0;
    } else if (where == 2) {
      ns = 0;  // result back in f.planes
    }
  }
  for (int s = 0; s < ns; s++) {
    int later = 0;
    for (int k = s + 1; k < ns; k++) later += borders[k];
    const int y0 = max(0, y_lo - later), y1 = min(f.ysize, y_hi + later);
    if (stages[s] < 0) {
      ScopedKernelTimer t(ctx, "k2_gaborish");
      for (int c = 0; c < 3; c++)
        launch_gaborish(ctx->stream, cur[c], oth[c], f.xsize, f.ysize, f.plane_stride, f.gab_k[c][0], f.gab_k[c][1],
                        f.gab_k[c][2], y0, y1);
    } else {
      EpfArgs a;
      for (int c = 0; c < 3; c++) {
        a.in[c] = cur[c];
        a.out[c] = oth[c];
        a.scale[c] = f.epf_channel_scale[c];
      }
      a.inv_sigma = f.inv_sigma;
      a.stride = f.plane_stride;
      a.sigma_stride = (size_t)f.xblocks;
      a.w = f.xsize;
      a.h = f.ysize;
      a.sm = f.epf_sm[stages[s]];
      a.bsm = f.epf_bsm[stages[s]];
      static const char* names[3] = {"k3a_epf0", "k3b_epf1", "k3c_epf2"};
      ScopedKernelTimer t(ctx, names[stages[s]]);
      launch_epf(ctx->stream, stages[s], a, y0, y1);
    }
    for (int c = 0; c < 3; c++) {
      float* t = cur[c];
      cur[c] = oth[c];
      oth[c] = t;
    }
  }
  return run_post_stages(ctx, cur, y_lo, y_hi, whole_frame);
}

// what follows the filters: upsampling and noise on the finished planes `cur` (rows [y_lo, y_hi))
jxlh_status run_post_stages(jxlh_ctx* ctx, float* const cur[3], int y_lo, int y_hi, bool whole_frame) {
  FrameDev& f = ctx->fd;
  const jxlh_frame_params& p = ctx->params;
  for (int c = 0; c < 3; c++) ctx->result[c] = cur[c];
  ctx->res_w = f.xsize;
  ctx->res_h = f.ysize;
  ctx->res_stride = f.plane_stride;
  if (p.upsampling > 1) {
    // Upsample2x/4x/8x on the three colour channels (frame/render.rs:655-671).  The 5x5 window crosses band
    // edges, so an upsampled frame is run whole.
    if (!whole_frame) return JXLH_ERR_UNSUPPORTED;
    const int n = (int)p.upsampling;
    const int ow = p.xsize_upsampled ? (int)p.xsize_upsampled : f.xsize * n;
    const int oh = p.ysize_upsampled ? (int)p.ysize_upsampled : f.ysize * n;
    const size_t ostride = round_up((size_t)f.xsize * n, 64);
    if (jxlh_status st = upload_upsampling_kernels(ctx, n)) return st;
    for (int c = 0; c < 3; c++)
      if (jxlh_status st = ensure(ctx, ctx->ups[c], ostride * (size_t)f.ysize * n)) return st;
    ScopedKernelTimer t(ctx, "k_upsample");
    for (int c = 0; c < 3; c++) {
      launch_upsample(ctx->stream, n, cur[c], f.plane_stride, f.xsize, f.ysize, ctx->ups_kernels.p, ctx->ups[c].p,
                      ostride, ow, oh);
      ctx->result[c] = ctx->ups[c].p;
    }
    ctx->res_w = ow;
    ctx->res_h = oh;
    ctx->res_stride = ostride;
  }
  if (p.noise && !noise_lut_is_zero(p.noise_lut)) {  // AddNoiseStage returns early on an all-zero LUT (noise.rs:153-155)
    // render_noise_for_group + ConvolveNoise x3 + AddNoise (frame/decode.rs:578-668, frame/render.rs:673-683),
    // at the resolution of the result (after upsampling).  Random planes: the 256-row tile rows that cover
    // the band plus the convolution's 2-row border.
    const int W = ctx->res_w, H = ctx->res_h;
    const int ya = p.upsampling > 1 ? 0 : y_lo, yb = p.upsampling > 1 ? H : y_hi;
    if (jxlh_status st = ensure_jump_table(ctx)) return st;
    for (int c = 0; c < 3; c++)
      if (jxlh_status st = ensure(ctx, ctx->noise[c], ctx->res_stride * (size_t)H)) return st;
    float* nz[3] = {ctx->noise[0].p, ctx->noise[1].p, ctx->noise[2].p};
    const int ty0 = max(0, ya - 2) / 256, ty1 = (min(H, yb + 2) + 255) / 256;
    {
      ScopedKernelTimer t(ctx, "k_noise_generate");
      launch_noise_generate(ctx->stream, nz, ctx->res_stride, W, H, ty0, ty1, p.visible_frame_index,
                            p.nonvisible_frame_index, ctx->xs_jump.p);
    }
    const float ytox = p.base_correlation_x + (float)p.ytox_lf / (float)p.color_factor;  // y_to_x_lf
    const float ytob = p.base_correlation_b + (float)p.ytob_lf / (float)p.color_factor;
    ScopedKernelTimer t(ctx, "k_noise_apply");
    launch_noise_apply(ctx->stream, nz, ctx->res_stride, ctx->result, ctx->res_stride, W, H, ya, yb, p.noise_lut, ytox,
                       ytob);
  }
  // (a partial re-render still picks up a channel that was handed over after the last whole-frame run)
  bool pending_extra = false;
  for (int i = 0; i < JXLH_MAX_EXTRA_CHANNELS; i++) pending_extra |= ctx->extra[i].set && !ctx->extra[i].done;
  if (whole_frame || pending_extra)
    if (jxlh_status st = run_extra_channels(ctx)) return st;
  HIPCHK(ctx, hipGetLastError());
  return JXLH_OK;
}

// channels 3.. of the reference's pipeline: ConvertModularToF32Stage, then Upsample<N> by the channel's own factor
// (frame/render.rs:564-567, :624-637 / :655-671)
jxlh_status run_extra_channels(jxlh_ctx* ctx) {
  for (int i = 0; i < JXLH_MAX_EXTRA_CHANNELS; i++) {
    jxlh_ctx::ExtraChannel& e = ctx->extra[i];
    if (!e.set) continue;
    const size_t n = (size_t)e.w * e.h;
    e.done = false;
    if (jxlh_status st = ensure(ctx, e.f32, n)) return st;
    {
      ScopedKernelTimer t(ctx, "k_modular_to_f32");
      if (e.bits >> 8)  // floating-point samples (BitDepth::floating_point_sample, convert.rs:525-526)
        launch_float_samples_to_f32(ctx->stream, e.raw.p, n, e.bits & 0xffu, e.bits >> 8, e.f32.p);
      else
        launch_modular_to_f32(ctx->stream, e.raw.p, n, 1.0f / (float)((1ull << e.bits) - 1), e.f32.p);
    }
    // the frame's result size bounds the channel's (the padding of ceil(size / factor) * factor is cut off)
    const uint32_t full_w = (uint32_t)(ctx->res_w > 0 ? ctx->res_w : ctx->fd.xsize),
                   full_h = (uint32_t)(ctx->res_h > 0 ? ctx->res_h : ctx->fd.ysize);
    e.out_w = std::min(e.w * e.up, full_w);
    e.out_h = std::min(e.h * e.up, full_h);
    if (e.up > 1) {
      if (jxlh_status st = upload_upsampling_kernels(ctx, (int)e.up)) return st;
      e.out_stride = round_up((size_t)e.w * e.up, 64);
      if (jxlh_status st = ensure(ctx, e.out, e.out_stride * (size_t)e.h * e.up)) return st;
      ScopedKernelTimer t(ctx, "k_upsample");
      launch_upsample(ctx->stream, (int)e.up, e.f32.p, e.w, (int)e.w, (int)e.h, ctx->ups_kernels.p, e.out.p, e.out_stride,
                      (int)e.out_w, (int)e.out_h);
    } else {
      e.out_stride = e.w;
    }
    e.done = true;
  }
  return JXLH_OK;
}
}  // namespace jxlh_host

extern "C" {

jxlh_status jxlh_frame_run(jxlh_ctx* ctx, uint32_t group_row0, uint32_t group_row1) {
  JXLH_ON_DEVICE(ctx);
  if (!ctx) return JXLH_ERR_INVALID_ARGUMENT;
  if (!ctx->in_frame || !ctx->tables_set) return JXLH_ERR_BAD_STATE;
  FrameDev& f = ctx->fd;
  if (group_row1 > (uint32_t)f.ygroups) group_row1 = (uint32_t)f.ygroups;
  if (group_row0 >= group_row1) return JXLH_ERR_INVALID_ARGUMENT;
  RunPlan plan;
  const bool whole = group_row0 == 0 && group_row1 == (uint32_t)f.ygroups;
  plan.want_strip = whole && strip_eligible(ctx);
  if (jxlh_status st = run_prologue(ctx, &plan)) return st;
  ctx->strip_ran = false;
  if (plan.want_strip) {
    const jxlh_status st = run_strip(ctx, plan);
    if (st != JXLH_ERR_UNSUPPORTED) return st;
    ctx->fd.strip_desc = nullptr;  // (nothing was launched) the two-kernel path takes the frame
    ctx->fd.strip_mode = nullptr;
  }
  // ---- K1 on the band plus one halo group row on each side (filters read across it)
  // (vertical chroma upsampling reads one sub-sampled row beyond the band as well)
  const bool need_halo = plan.halo_px > 0 || f.subsampled;
  const int gr0 = need_halo && group_row0 > 0 ? (int)group_row0 - 1 : (int)group_row0;
  const int gr1 = need_halo && group_row1 < (uint32_t)f.ygroups ? (int)group_row1 + 1 : (int)group_row1;
  if (jxlh_status st = run_k1(ctx, plan, gr0, gr1)) return st;
  if (group_row0 == 0 && group_row1 == (uint32_t)f.ygroups) ctx->rendered = true;
  return run_stages(ctx, plan, group_row0, group_row1);
}

jxlh_status jxlh_frame_rerender_groups(jxlh_ctx* ctx, const uint32_t* group_ids, uint32_t count) {
  JXLH_ON_DEVICE(ctx);
  if (!ctx || (count && !group_ids)) return JXLH_ERR_INVALID_ARGUMENT;
  if (!ctx->in_frame || !ctx->tables_set) return JXLH_ERR_BAD_STATE;
  FrameDev& f = ctx->fd;
  for (uint32_t i = 0; i < count; i++)
    if (group_ids[i] >= ctx->ngroups) return JXLH_ERR_INVALID_ARGUMENT;
  if (count == 0) return JXLH_OK;
  const jxlh_frame_params& p = ctx->params;
  if (p.upsampling > 1) return JXLH_ERR_UNSUPPORTED;  // like a band run: the 5x5 upsampling window crosses groups
  // a rank of a sharded frame holds only its band: progressive re-renders run on unsharded contexts
  if (comm_nranks(ctx) > 1) return JXLH_ERR_UNSUPPORTED;
  const int ns = (f.gab ? 1 : 0) + (f.epf_iters >= 3 ? 1 : 0) + (f.epf_iters >= 1 ? 1 : 0) + (f.epf_iters >= 2 ? 1 : 0);
  // Re-rendering a group needs its neighbours' UNFILTERED pixels (the filters read across the group edge).  They
  // are still in `planes` when the stage list leaves its result in `tmp` (the fused path with up to two EPF passes,
  // or no filter at all); a stage list that ends in `planes` has overwritten them, a sub-sampled frame keeps them
  // in another form, and a frame that was never rendered has none: those render the frame again.
  const bool per_stage = (p.flags & JXLH_FRAME_UNFUSED_FILTERS) != 0;  // ping-pongs planes <-> tmp: kept only for one stage
  const bool unfiltered_kept = !ctx->strip_ran && (ns == 0 || (per_stage ? ns == 1 : result_in_tmp(ctx) != 0));
  // Noise is added IN PLACE to the result planes.  Without a filter stage the result lives in `planes`, the planes K1
  // writes: the groups that are not re-transformed would receive their noise a second time.
  const bool noise_in_place = ns == 0 && p.noise && !noise_lut_is_zero(p.noise_lut);
  if (!ctx->rendered || !unfiltered_kept || f.subsampled || noise_in_place) return jxlh_frame_run(ctx, 0, UINT32_MAX);
  RunPlan plan;
  if (jxlh_status st = run_prologue(ctx, &plan)) return st;
  // ---- transforms of exactly the listed groups
  ctx->rerender_upload.assign(group_ids, group_ids + count);
  std::sort(ctx->rerender_upload.begin(), ctx->rerender_upload.end());
  ctx->rerender_upload.erase(std::unique(ctx->rerender_upload.begin(), ctx->rerender_upload.end()),
                             ctx->rerender_upload.end());
  const int n = (int)ctx->rerender_upload.size();
  if (jxlh_status st = ensure(ctx, ctx->rerender_list, (size_t)n)) return st;
  HIPCHK(ctx, hipMemcpyAsync(ctx->rerender_list.p, ctx->rerender_upload.data(), n * sizeof(int), hipMemcpyHostToDevice,
                             ctx->stream));
  {
    ScopedKernelTimer t(ctx, "k1_vardct");
    set_sparse_view(ctx, f, plan.sparse_k1);
    if (plan.sparse_k1) HIPCHK(ctx, hipMemsetAsync(ctx->group_dense.p, 0, ctx->ngroups, ctx->stream));
    launch_vardct_groups(ctx->stream, f, 0, 0, ctx->worklist.p, &ctx->k1_launches, ctx->error_flag.p,
                         plan.sparse_k1 ? ctx->coeffs.p : nullptr, ctx->rerender_list.p, n, ctx->has_special,
                         ctx->has_large, dense_route_groups(ctx, plan.sparse_k1));
  }
  if (jxlh_status st = mark_coefficients_read(ctx, plan.sparse_k1)) return st;
  // ---- the filters on every pixel row the listed groups influence: their own rows widened by the stage list's
  // reach (mark_group_to_rerender's 3x3 neighbourhood, restricted to what can actually change), merged into bands
  int prev_lo = -1, prev_hi = -1;
  for (int i = 0; i <= n; i++) {
    int lo = -1, hi = -1;
    if (i < n) {
      const int gy = ctx->rerender_upload[i] / f.xgroups;
      lo = max(0, gy * kGroupDim - plan.halo_px);
      hi = min(f.ysize, (gy + 1) * kGroupDim + plan.halo_px);
    }
    if (i < n && prev_hi >= lo) {
      prev_hi = max(prev_hi, hi);
      continue;
    }
    if (prev_lo >= 0)
      if (jxlh_status st = run_stages_rows(ctx, plan, prev_lo, prev_hi, prev_lo == 0 && prev_hi == f.ysize)) return st;
    prev_lo = lo;
    prev_hi = hi;
  }
  return JXLH_OK;
}

jxlh_status jxlh_frame_set_extra_channel(jxlh_ctx* ctx, uint32_t ec, const int32_t* samples, size_t stride, uint32_t w,
                                         uint32_t h, uint32_t bits_per_sample, uint32_t ec_upsampling) {
  JXLH_ON_DEVICE(ctx);
  if (!ctx || !samples || ec >= JXLH_MAX_EXTRA_CHANNELS || w == 0 || h == 0 || stride < w ||
      !bit_depth_ok(bits_per_sample, 31) ||
      (ec_upsampling != 1 && ec_upsampling != 2 && ec_upsampling != 4 && ec_upsampling != 8))
    return JXLH_ERR_INVALID_ARGUMENT;
  if (!ctx->in_frame) return JXLH_ERR_BAD_STATE;
  if ((uint64_t)w * h >= (1ull << 31)) return JXLH_ERR_UNSUPPORTED;
  jxlh_ctx::ExtraChannel& e = ctx->extra[ec];
  if (jxlh_status st = ensure(ctx, e.raw, (size_t)w * h)) return st;
  if (jxlh_status st = copy2d(ctx, e.raw.p, (size_t)w * sizeof(int32_t), samples, stride * sizeof(int32_t),
                              (size_t)w * sizeof(int32_t), h, ctx->stream))
    return st;
  JXLH_SYNC(ctx);  // the caller's buffer may be reused as soon as the call returns
  e.w = w;
  e.h = h;
  e.bits = bits_per_sample;
  e.up = ec_upsampling;
  e.set = true;
  e.done = false;
  return JXLH_OK;
}

jxlh_status jxlh_frame_read_extra_channel(jxlh_ctx* ctx, uint32_t ec, const jxlh_plane* out) {
  JXLH_ON_DEVICE(ctx);
  if (!ctx || !out || ec >= JXLH_MAX_EXTRA_CHANNELS) return JXLH_ERR_INVALID_ARGUMENT;
  if (!ctx->in_frame) return JXLH_ERR_BAD_STATE;
  const jxlh_ctx::ExtraChannel& e = ctx->extra[ec];
  if (!e.set || !e.done) return JXLH_ERR_BAD_STATE;  // handed over but no jxlh_frame_run since
  if (!out->ptr || out->bytes_per_row < (size_t)e.out_w * sizeof(float) || out->num_rows < e.out_h ||
      out->bytes_between_rows < out->bytes_per_row)
    return JXLH_ERR_INVALID_ARGUMENT;
  const float* src = e.up > 1 ? e.out.p : e.f32.p;
  if (jxlh_status st = copy2d(ctx, out->ptr, out->bytes_between_rows, src, e.out_stride * sizeof(float),
                              (size_t)e.out_w * sizeof(float), e.out_h, ctx->stream))
    return st;
  return jxlh_ctx_sync(ctx);
}

// hand-over of caller-filled device buffers (jxl_hip.h "STREAM ORDERING OF DEVICE POINTERS"): every stream of the
// context -- the main one and the slot streams -- waits for the event
static jxlh_status all_streams_wait(jxlh_ctx* ctx, hipEvent_t e) {
  HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, e, 0));
  for (auto& s : ctx->slots) HIPCHK(ctx, hipStreamWaitEvent(s.stream, e, 0));
  return JXLH_OK;
}

jxlh_status jxlh_ctx_tune_placement(jxlh_ctx* ctx, int32_t trials, float* report, int32_t report_capacity, int32_t* n_report,
                                    int32_t* picked) {
  if (!ctx || trials < 0 || trials > 64 || (report_capacity > 0 && !report)) return JXLH_ERR_INVALID_ARGUMENT;
  if (trials > 0) ctx->placement_trials = trials;
  const int n = (int)ctx->placement_report.size();
  for (int i = 0; i < n && i < report_capacity; i++) report[i] = ctx->placement_report[(size_t)i];
  if (n_report) *n_report = n;
  if (picked) *picked = ctx->placement_pick;
  return JXLH_OK;
}

jxlh_status jxlh_ctx_wait_event(jxlh_ctx* ctx, void* hip_event) {
  JXLH_ON_DEVICE(ctx);
  if (!ctx || !hip_event) return JXLH_ERR_INVALID_ARGUMENT;
  return all_streams_wait(ctx, static_cast<hipEvent_t>(hip_event));
}

jxlh_status jxlh_ctx_wait_stream(jxlh_ctx* ctx, void* hip_stream) {
  JXLH_ON_DEVICE(ctx);
  if (!ctx) return JXLH_ERR_INVALID_ARGUMENT;
  if (!ctx->handover) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->handover, hipEventDisableTiming));
  // (an event may be re-recorded while earlier waits on it are still pending: a wait captures the record it follows)
  HIPCHK(ctx, hipEventRecord(ctx->handover, static_cast<hipStream_t>(hip_stream)));
  return all_streams_wait(ctx, ctx->handover);
}

jxlh_status jxlh_ctx_record_event(jxlh_ctx* ctx, void* hip_event) {
  JXLH_ON_DEVICE(ctx);
  if (!ctx || !hip_event) return JXLH_ERR_INVALID_ARGUMENT;
  HIPCHK(ctx, hipEventRecord(static_cast<hipEvent_t>(hip_event), ctx->stream));
  return JXLH_OK;
}

// the dataflow squeeze launch's error word (pinned host memory the kernel writes): non-zero = a wait between two levels
// outlasted its deadline; reported once
static jxlh_status flow_error_status(jxlh_ctx* ctx) {
  if (!ctx->host_flow_flag) return JXLH_OK;
  const int v = *reinterpret_cast<volatile int*>(ctx->host_flow_flag);
  if (v == 0) return JXLH_OK;
  *ctx->host_flow_flag = 0;
  ctx->last_error = "jxlh_unsqueeze_chain: a wait between two levels of the dataflow launch outlasted its deadline";
  return (jxlh_status)v;
}

jxlh_status jxlh_ctx_mark(jxlh_ctx* ctx, uint32_t* mark) {
  JXLH_ON_DEVICE(ctx);
  if (!ctx || !mark) return JXLH_ERR_INVALID_ARGUMENT;
  const uint32_t seq = ++ctx->mark_seq;
  hipEvent_t& e = ctx->marks[seq % JXLH_MAX_MARKS];
  if (!e) HIPCHK(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
  HIPCHK(ctx, hipEventRecord(e, ctx->stream));
  *mark = seq;
  return JXLH_OK;
}

jxlh_status jxlh_ctx_wait_mark(jxlh_ctx* ctx, uint32_t mark) {
  JXLH_ON_DEVICE(ctx);
  if (!ctx || mark == 0 || (int32_t)(ctx->mark_seq - mark) < 0) return JXLH_ERR_INVALID_ARGUMENT;  // never handed out
  // (a mark older than the ring: its slot holds a later point of the stream -- waiting for that one is sufficient)
  hipEvent_t e = ctx->marks[mark % JXLH_MAX_MARKS];
  if (!e) return JXLH_ERR_BAD_STATE;
  HIPCHK(ctx, hipEventSynchronize(e));
  return flow_error_status(ctx);  // (a streaming caller that only ever waits for marks sees a dataflow fault too)
}

jxlh_status jxlh_ctx_sync(jxlh_ctx* ctx) {
  JXLH_ON_DEVICE(ctx);
  if (!ctx) return JXLH_ERR_INVALID_ARGUMENT;
  if (ctx->flow_used && ctx->host_flow_flag) {
    // a dataflow squeeze launch ran since the last synchronisation: did one of its waits give up?
    ctx->flow_used = false;
    if (jxlh_status st = comm_wait_stream(ctx)) return st;
    if (jxlh_status st = flow_error_status(ctx)) return st;
  }
  if (ctx->in_frame && ctx->error_flag.p) {
    // read the flag on the context's own stream into pinned memory: a synchronous hipMemcpy would
    // go through the null stream and serialise against other contexts' work
    if (!ctx->host_flag) HIPCHK(ctx, hipHostMalloc(reinterpret_cast<void**>(&ctx->host_flag), sizeof(int), hipHostMallocDefault));
    HIPCHK(ctx, hipMemcpyAsync(ctx->host_flag, ctx->error_flag.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    if (jxlh_status st = comm_wait_stream(ctx)) return st;
    if (*ctx->host_flag != 0) return (jxlh_status)*ctx->host_flag;
    return JXLH_OK;
  }
  return comm_wait_stream(ctx);
}

}  // extern "C"
