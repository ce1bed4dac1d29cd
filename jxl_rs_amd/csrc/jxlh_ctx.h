// Host-side state behind the C ABI: the context structure and the small helpers abi.hip and comm.hip share.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "jxlh_internal.h"

using namespace jxlh;

// Every entry point that takes a context makes the context's device the calling thread's current device first: the
// ABI promises one submitting thread per slot plus whoever runs / reads the frame, and HIP's current device is
// per-thread state (a fresh thread sits on device 0).  hipSetDevice on the device already current costs ~70 ns.
#define JXLH_ON_DEVICE(ctx)                        \
  do {                                             \
    if ((ctx) != nullptr) (void)hipSetDevice((ctx)->device); \
  } while (0)

namespace jxlh_host {

struct Slot {
  hipStream_t stream = nullptr;
  hipEvent_t done = nullptr;
  // the last submission's host-to-device copies have landed -- recorded apart from `done` when device work follows the
  // copies on the slot's stream (the 12-bit entries' unpack kernel): jxlh_slot_wait / jxlh_slot_after are about the
  // copies (host buffers, the bus), the frame waits for `done`
  hipEvent_t copied = nullptr;
  bool copied_valid = false;
  bool used = false;
  uint8_t* stage8 = nullptr;  // device staging of the 3-byte sparse form (positions | values), grown on demand
  size_t stage8_cap = 0;
};

struct KernelTime {
  std::string name;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
  float total_ms = 0.f;
  int launches = 0;
};

template <class T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;  // elements
};

}  // namespace jxlh_host
using namespace jxlh_host;

namespace jxlh_host {
struct Comm;  // comm.hip
}

struct jxlh_ctx {
  jxlh_host::Comm* comm = nullptr;  // multi-GPU: rank / transport of this context (null = single GPU)
  int device = 0;
  hipStream_t stream = nullptr;
  std::vector<Slot> slots;
  hipEvent_t t0 = nullptr, t1 = nullptr;
  std::string last_error;
  // frame state
  bool in_frame = false;
  bool tables_set = false, lf_smoothed = false;
  jxlh_frame_params params;
  FrameDev fd;
  size_t ngroups = 0;
  DevBuf<float> planes[3], tmp[3], lf_raw[3], lf_sm[3], sigma, tables;
  int table_offset[JXLH_NUM_QUANT_TABLES] = {0};
  DevBuf<int32_t> coeffs, raw_quant, lfq;
  DevBuf<uint8_t> transform_map, epf_map;
  DevBuf<int8_t> ytox, ytob;
  DevBuf<int> error_flag;
  DevBuf<int> tables_ok;     // FrameDev::tables_ok
  int tables_ok_host = 0;    // ... read back when the tables are set
  bool params_direct_ok = false;  // the frame parameters' share of FrameDev::se_direct_ok
  DevBuf<uint8_t> rgb8;  // jxlh_frame_read_rgb8 staging for host destinations
  int* host_flag = nullptr;  // pinned
  DevBuf<uint8_t> worklist;
  DevBuf<int> rerender_list;          // group ids of jxlh_frame_rerender_groups on the device
  std::vector<int> rerender_upload;   // ... and their host copy (alive until the copy has run)
  bool rendered = false;              // a full jxlh_frame_run has happened in this frame
  bool has_special = false, has_large = false;  // transform families seen in the frame's maps (jxlh_frame_set_hf_meta)
  float* result[3] = {nullptr, nullptr, nullptr};
  // geometry of `result`: the frame itself, or its upsampled image (frame_header.upsampling > 1)
  int res_w = 0, res_h = 0;
  size_t res_stride = 0;
  // chroma-subsampled frame with nothing between the transforms and the output: the upsampling into planes[] is
  // deferred until somebody asks for the planes (the YCbCr output calls read the sub-sampled channels directly)
  bool chroma_lazy = false;
  int lazy_gr0 = 0, lazy_gr1 = 0;
  DevBuf<float> noise[3];      // random planes of the noise synthesis
  DevBuf<uint64_t> xs_jump;    // xorshift128+ jump matrices T^(2^j), uploaded on first use
  DevBuf<float> ups[3];        // upsampled planes
  // expanded 5x5 kernels per factor (2, 4, 8), uploaded on first use and whenever jxlh_set_upsampling_weights changes
  // the weights; ups_kernels = the set the last upload_upsampling_kernels call selected
  DevBuf<float> ups_kernels_n[3];
  bool ups_valid[3] = {false, false, false};
  struct { float* p = nullptr; } ups_kernels;
  std::vector<float> ups_weights[3];  // custom weights2 / weights4 / weights8 (empty = defaults)
  // stage hooks scratch
  DevBuf<float> hook_f[8];
  DevBuf<int32_t> hook_i[4];
  // jxlh_unsqueeze_chain's dataflow launches (k6_unsqueeze_flow): ticket + progress words, zeroed per launch; the error
  // word (zeroed when allocated and after an error was reported) is read back by the next jxlh_ctx_sync
  DevBuf<int> flow_words;
  bool flow_used = false;
  int* host_flow_flag = nullptr;  // pinned, device-visible: the dataflow launches' error word (0 = none)
  // jxlh_flow_profile (jxl_hip_dev.h): per-level timeline of the last dataflow launch
  DevBuf<unsigned long long> flow_prof;
  bool flow_prof_on = false;
  int flow_prof_levels = 0;
  // sparse coefficient transport (jxlh_submit_group(s)_sparse): pairs land in sp_pairs (bump
  // allocated, sized for a frame's worst case), are expanded by the next jxlh_frame_run
  std::mutex sp_mutex;
  DevBuf<uint32_t> sp_pairs;
  DevBuf<SparseGroup> sp_groups_dev;
  DevBuf<uint2> sp_wide_dev;
  std::vector<SparseGroup> sp_pending, sp_upload;
  std::vector<uint2> sp_wide, sp_wide_upload;
  size_t sp_used = 0;
  hipEvent_t sp_expanded = nullptr;
  bool sp_expanded_valid = false;
  // recorded behind the transforms of every jxlh_frame_run: dense resubmissions wait for it
  hipEvent_t k1_done = nullptr;
  bool k1_done_valid = false;
  uint32_t k1_launches = 0;  // parity selects the work-list counter set (vardct_worklist_reset / launch_vardct_groups)
  // K1 reading the pairs directly: the frame's pairs bucketed by varblock slot + slot tables.  Valid
  // while every group of the frame has been submitted sparse (once) and nothing was resubmitted.
  DevBuf<uint32_t> sp_sorted, sp_slot_start;
  DevBuf<uint8_t> group_dense;
  // Epochs: the submissions between two jxlh_frame_run calls.  touched[g]: 0 not resubmitted (keeps its
  // content), 1 dense slab, 2 pairs.  sp_sorted_valid: before this epoch every group's content lived in
  // the bucketed form (and only there).
  std::vector<uint8_t> touched, flag_upload;
  // groups submitted in the slot-bucketed form in this epoch (jxlh_submit_groups_slots): their entries, slot counts
  // and run descriptors sit in se_*[se_live ^ 1] exactly as uploaded.  If that is every group (and nothing is added to
  // earlier passes), jxlh_frame_run makes that set the live one and the transforms read it in place (round 5; round 4
  // unpacked it into pair words + slot tables at submission time, overwriting the tables the resident frame was read
  // through); otherwise the flagged groups' entries are widened into the pair buffer first.
  std::vector<uint8_t> bucketed, bucketed_upload;
  DevBuf<uint8_t> bucketed_dev;
  // The two sets trade places when a frame arrives entirely slot-bucketed, so the uploads of frame i + 1 never touch
  // what the transforms of frame i read; se_read[i]: recorded behind the last kernels that read set i.
  DevBuf<uint16_t> se_entries[2];
  DevBuf<uint8_t> se_counts[2];
  DevBuf<uint2> se_runs[2];
  hipEvent_t se_read[2] = {nullptr, nullptr};
  bool se_read_valid[2] = {false, false};
  int se_live = 0;
  bool se_valid = false;  // the resident bucketed form is se_*[se_live] (else, with sp_sorted_valid, the pair words)
  bool epoch_dirty = false;
  bool sp_sorted_valid = false;
  // per-group routing of a frame that is resident in the slot-bucketed form (round 6): route_live[g] != 0 = group g
  // lives in its dense slab, not in the live set (empty = no routed group); route_dev the device copy the scan reads
  std::vector<uint8_t> route_live, route_upload;
  DevBuf<uint8_t> route_dev;
  int n_route = 0;
  int se_dense_hint = 0;       // FrameDev::se_dense_hint of the live set
  // extra channels inside the frame path (jxlh_frame_set_extra_channel): as handed over, converted, upsampled
  struct ExtraChannel {
    bool set = false, done = false;
    uint32_t w = 0, h = 0, bits = 0, up = 1;
    uint32_t out_w = 0, out_h = 0;
    size_t out_stride = 0;
    DevBuf<int32_t> raw;
    DevBuf<float> f32, out;
  };
  ExtraChannel extra[JXLH_MAX_EXTRA_CHANNELS];
  // strip path (k_strip.hip): block descriptors / tile modes written by k1_scan, the strips' edge-column exchange
  // buffer, progress flags + ticket.  strip_all_closed: every rect of the transform map came from host memory and
  // every varblock in it is a small DCT inside its 64x64 tile (jxlh_frame_set_hf_meta); strip_ran: the last
  // jxlh_frame_run went through the strip kernel (`planes` then hold no unfiltered pixels).
  DevBuf<uint2> strip_desc;
  DevBuf<uint8_t> strip_mode;
  DevBuf<float> strip_xchg;
  DevBuf<int> strip_flags;
  bool strip_all_closed = true, strip_ran = false;
  int cu_count = 0;
  // jxlh_ctx_tune_placement: candidate sets the first allocation of the large buffers is picked from (<= 1: plain
  // allocation); what the last pick saw (k1-like ms, filter-like ms per candidate) and took
  int placement_trials = 1;
  std::vector<float> placement_report;
  int placement_pick = -1;
  int strip_resident = 0;  // strip_resident_workgroups(cu_count), 0 = not asked yet
  // jxlh_ctx_mark / jxlh_ctx_wait_mark: a ring of events on the main stream
  hipEvent_t handover = nullptr;  // jxlh_ctx_wait_stream: recorded on the caller's stream
  hipEvent_t marks[JXLH_MAX_MARKS] = {};
  uint32_t mark_seq = 0;
  // profiling
  bool timing = false;
  std::vector<KernelTime> ktimes;
};

namespace jxlh_host {

inline jxlh_status fail(jxlh_ctx* ctx, hipError_t e, const char* what) {
  if (ctx) {
    ctx->last_error = std::string(what) + ": " + hipGetErrorString(e);
  }
  (void)hipGetLastError();  // clear the sticky per-thread error so later checks start clean
  return e == hipErrorOutOfMemory ? JXLH_ERR_OUT_OF_MEMORY : JXLH_ERR_DEVICE;
}

#define HIPCHK(ctx, expr)                               \
  do {                                                  \
    hipError_t e_ = (expr);                             \
    if (e_ != hipSuccess) return fail(ctx, e_, #expr);  \
  } while (0)

// k_probe.hip: average ms of two byte movers with K1's and the filters' streams on a candidate set of buffers
jxlh_status probe_placement(jxlh_ctx* ctx, const int32_t* coeffs, size_t ngroups, float* const planes[3], float* const tmp[3],
                            size_t plane_elems, float* k1_like_ms, float* filter_like_ms);

template <class T>
jxlh_status ensure(jxlh_ctx* ctx, DevBuf<T>& b, size_t n) {
  if (b.n >= n && b.p) return JXLH_OK;
  if (b.p) {
    HIPCHK(ctx, hipFree(b.p));
    b.p = nullptr;
    b.n = 0;
  }
  if (n == 0) return JXLH_OK;
  HIPCHK(ctx, hipMalloc(reinterpret_cast<void**>(&b.p), n * sizeof(T)));
  b.n = n;
  return JXLH_OK;
}

template <class T>
void release(DevBuf<T>& b) {
  if (b.p) (void)hipFree(b.p);
  b.p = nullptr;
  b.n = 0;
}

struct ScopedKernelTimer {
  jxlh_ctx* ctx;
  hipEvent_t a = nullptr, b = nullptr;
  KernelTime* kt = nullptr;
  ScopedKernelTimer(jxlh_ctx* c, const char* name) : ctx(c) {
    if (!ctx->timing) return;
    for (auto& k : ctx->ktimes)
      if (k.name == name) kt = &k;
    if (!kt) {
      ctx->ktimes.push_back(KernelTime{name, {}, 0.f, 0});
      kt = &ctx->ktimes.back();
    }
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    (void)hipEventRecord(a, ctx->stream);
  }
  ~ScopedKernelTimer() {
    if (!kt) return;
    (void)hipEventRecord(b, ctx->stream);
    kt->pending.emplace_back(a, b);
  }
};

inline void drain_timers(jxlh_ctx* ctx) {
  for (auto& k : ctx->ktimes) {
    for (auto& pr : k.pending) {
      (void)hipEventSynchronize(pr.second);
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) {
        k.total_ms += ms;
        k.launches += 1;
      }
      (void)hipEventDestroy(pr.first);
      (void)hipEventDestroy(pr.second);
    }
    k.pending.clear();
  }
}

inline size_t round_up(size_t v, size_t m) { return (v + m - 1) / m * m; }

// plane -> device 2-D copy helper (pointers may be host or device)
inline bool is_device_ptr(const void* p) {
  hipPointerAttribute_t attr;
  if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return attr.type == hipMemoryTypeDevice;
}

inline jxlh_status copy2d(jxlh_ctx* ctx, void* dst, size_t dpitch, const void* src, size_t spitch, size_t width_bytes,
                   size_t height, hipStream_t s) {
  if (width_bytes == 0 || height == 0) return JXLH_OK;
  if (dpitch == width_bytes && spitch == width_bytes) {  // contiguous on both sides: one linear copy
    HIPCHK(ctx, hipMemcpyAsync(dst, src, width_bytes * height, hipMemcpyDefault, s));
    return JXLH_OK;
  }
  HIPCHK(ctx, hipMemcpy2DAsync(dst, dpitch, src, spitch, width_bytes, height, hipMemcpyDefault, s));
  return JXLH_OK;
}

// jxlh_frame_run in three pieces (abi.hip), so that a sharded run (comm.hip) can put the halo exchange between the
// transforms and the filters
struct RunPlan {
  bool sparse_k1 = false;
  int halo_px = 0;       // rows the filters read beyond a band
  bool will_fuse = false;
  bool want_strip = false;  // in: the caller would run the strip kernel (dense slabs needed); out: it will
};
jxlh_status run_prologue(jxlh_ctx* ctx, RunPlan* plan);
jxlh_status run_k1(jxlh_ctx* ctx, const RunPlan& plan, int gr0, int gr1);
jxlh_status run_stages(jxlh_ctx* ctx, const RunPlan& plan, uint32_t group_row0, uint32_t group_row1);
jxlh_status run_stages_rows(jxlh_ctx* ctx, const RunPlan& plan, int y_lo, int y_hi, bool whole_frame);
jxlh_status run_post_stages(jxlh_ctx* ctx, float* const cur[3], int y_lo, int y_hi, bool whole_frame);
jxlh_status run_extra_channels(jxlh_ctx* ctx);  // ConvertModularToF32 + Upsample of the channels handed over
bool strip_eligible(const jxlh_ctx* ctx);
jxlh_status run_strip(jxlh_ctx* ctx, const RunPlan& plan);
// Where run_stages leaves the finished planes (1 = f.tmp, 0 = f.planes): a property of the frame's stage list, so a
// rank that filtered nothing (empty band) still knows where the gathered frame lives.
inline int result_in_tmp(const jxlh_ctx* ctx) {
  const FrameDev& f = ctx->fd;
  const int ns = (f.gab ? 1 : 0) + (f.epf_iters >= 3 ? 1 : 0) + (f.epf_iters >= 1 ? 1 : 0) + (f.epf_iters >= 2 ? 1 : 0);
  if (ns == 0) return 0;
  if (!(ctx->params.flags & JXLH_FRAME_UNFUSED_FILTERS)) return f.epf_iters >= 3 ? 0 : 1;
  return ns & 1;
}
void set_filter_params(FrameDev& f, const jxlh_frame_params& p);
// abi_frame.hip: pieces of the frame pipeline the read-out and stage-hook entry points share
bool noise_lut_is_zero(const float lut[8]);
void materialise_chroma(jxlh_ctx* ctx);          // deferred chroma upsampling of a sub-sampled frame, if still pending
jxlh_status ensure_jump_table(jxlh_ctx* ctx);    // xorshift128+ jump matrices of the noise generator
jxlh_status upload_upsampling_kernels(jxlh_ctx* ctx, int n);

// host or device source -> context-owned device scratch / back, on the main stream (stage hooks, Modular entry points)
template <class T>
jxlh_status stage_in(jxlh_ctx* ctx, DevBuf<T>& b, const T* src, size_t n) {
  jxlh_status st = ensure(ctx, b, n);
  if (st != JXLH_OK) return st;
  HIPCHK(ctx, hipMemcpyAsync(b.p, src, n * sizeof(T), hipMemcpyDefault, ctx->stream));
  return JXLH_OK;
}
// abi_output.hip
jxlh_status convert_band_to_output(jxlh_ctx* ctx, const jxlh_output_desc* d, uint32_t y0, uint32_t y1, void* out,
                                   size_t bytes_per_row);
// comm.hip
void comm_release(jxlh_ctx* ctx);
int comm_nranks(const jxlh_ctx* ctx);
// hipStreamSynchronize with a deadline while collectives may be queued.  EVERY blocking wait on the context's stream goes
// through it (JXLH_SYNC): on a sharded context a stuck collective then surfaces as JXLH_ERR_DEVICE from whichever call
// waits first -- a read-out as well as jxlh_ctx_sync -- instead of parking the process in the driver.
jxlh_status comm_wait_stream(jxlh_ctx* ctx);
#define JXLH_SYNC(ctx)                                        \
  do {                                                        \
    if (jxlh_status st_ = comm_wait_stream(ctx)) return st_;  \
  } while (0)
template <class T>
jxlh_status stage_out(jxlh_ctx* ctx, T* dst, const T* src, size_t n) {
  HIPCHK(ctx, hipMemcpyAsync(dst, src, n * sizeof(T), hipMemcpyDefault, ctx->stream));
  JXLH_SYNC(ctx);
  return JXLH_OK;
}
int comm_rows_per_rank(const jxlh_ctx* ctx, int ygroups);

}  // namespace jxlh_host
using namespace jxlh_host;
