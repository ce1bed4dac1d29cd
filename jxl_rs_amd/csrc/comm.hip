// Multi-GPU entry points of the C ABI (include/jxl_hip.h, "multi-GPU" section).
//
// Partitioning (reference join point frame/render.rs:461-479): a frame is cut into contiguous bands
// of group rows, one band per rank; a rank runs the transforms (K1) on exactly its band, receives the block row its
// filters read across each band edge from the neighbour rank (halo EXCHANGE: one 8-pixel block row x 3 channels per
// edge, 0.8 MB at 8K -- not a recomputed group row), filters its band, and the finished bands are all-gathered so
// that every rank holds the whole frame.
//
// Two transports behind the same band logic:
//   RCCL   one process per GPU (torch.distributed.run / MPI style launch): the library owns an RCCL communicator
//          (ncclCommInitRank from a 128-byte id the caller distributes out of band); halo = grouped ncclSend/ncclRecv,
//          gather = in-place ncclAllGather per plane, all enqueued on the context's stream.  librccl is opened with
//          dlopen on first use, so a single-GPU deployment has no dependency on it.
//   local  the peers are contexts of ONE process (several GPUs driven by one decoder process, or -- for tests --
//          several contexts on one GPU): direct device-to-device copies ordered by events.
#include <dlfcn.h>

#include <chrono>
#include <cstdlib>
#include <thread>
#include <rccl/rccl.h>  // types and enums only: every function is resolved at run time

#include "jxlh_ctx.h"

namespace jxlh_host {

struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t*) = nullptr;  // optional
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;                          // optional
};

static RcclApi* rccl_api(std::string* err) {
  static RcclApi api;
  static std::mutex m;
  std::lock_guard<std::mutex> lock(m);
  if (api.handle) return &api;
  // a copy another component of the process already loaded (PyTorch ships one) is reused through its soname
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  for (const char* n : names) {
    h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (h) break;
  }
  if (!h) {
    if (err) *err = std::string("dlopen librccl: ") + dlerror();
    return nullptr;
  }
  bool ok = true;
  auto sym = [&](const char* name) {
    void* p = dlsym(h, name);
    if (!p) {
      ok = false;
      if (err) *err = std::string("librccl lacks ") + name;
    }
    return p;
  };
  api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
  api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
  api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
  api.AllGather = reinterpret_cast<decltype(api.AllGather)>(sym("ncclAllGather"));
  api.Send = reinterpret_cast<decltype(api.Send)>(sym("ncclSend"));
  api.Recv = reinterpret_cast<decltype(api.Recv)>(sym("ncclRecv"));
  api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(sym("ncclGroupStart"));
  api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(sym("ncclGroupEnd"));
  api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
  if (!ok) {
    dlclose(h);
    return nullptr;
  }
  api.CommGetAsyncError = reinterpret_cast<decltype(api.CommGetAsyncError)>(dlsym(h, "ncclCommGetAsyncError"));
  api.CommAbort = reinterpret_cast<decltype(api.CommAbort)>(dlsym(h, "ncclCommAbort"));
  api.handle = h;
  return &api;
}

struct Comm {
  int rank = 0, nranks = 1;
  ncclComm_t nccl = nullptr;          // RCCL transport
  std::vector<jxlh_ctx*> peers;       // local transport (peers[rank] == the owning context)
  hipEvent_t k1_ev = nullptr;         // local: this rank's transforms are done (the neighbours copy its edge rows)
  hipEvent_t done_ev = nullptr;       // local: this rank's band is finished (the peers copy it in the gather)
  hipEvent_t pulled_ev = nullptr;     // local: this rank has copied its neighbours' edge rows (they may overwrite them)
  hipEvent_t gathered_ev = nullptr;   // local: this rank has copied the other bands (their owners may start the next frame)
  bool gathered_valid = false;
  // what was enqueued last on the stream through this communicator: named when a wait times out (comm_wait_stream)
  std::string last_op;
  double timeout_s = 120.0;           // JXLH_COMM_TIMEOUT_S; <= 0 waits forever
  // a wait on this communicator's stream expired or RCCL reported an asynchronous error: the stream still holds the
  // stuck collective, so the communicator is ABORTED on release (ncclCommDestroy would wait for it) and every later
  // wait fails at once instead of sitting out another deadline
  bool failed = false;
};

static jxlh_status nccl_fail(jxlh_ctx* ctx, const RcclApi* api, ncclResult_t r, const char* what) {
  ctx->last_error = std::string(what) + ": " + (api && api->GetErrorString ? api->GetErrorString(r) : "rccl error");
  return JXLH_ERR_DEVICE;
}
#define NCCLCHK(ctx, api, expr)                                     \
  do {                                                              \
    ncclResult_t r_ = (expr);                                       \
    if (r_ != ncclSuccess) return nccl_fail(ctx, api, r_, #expr);   \
  } while (0)

// inside ncclGroupStart ... ncclGroupEnd: an error must not leave the group open (every later RCCL call of the thread
// would be queued into it)
#define NCCLCHK_IN_GROUP(ctx, api, expr)                            \
  do {                                                              \
    ncclResult_t r_ = (expr);                                       \
    if (r_ != ncclSuccess) {                                        \
      (void)(api)->GroupEnd();                                      \
      return nccl_fail(ctx, api, r_, #expr);                        \
    }                                                               \
  } while (0)

void comm_release(jxlh_ctx* ctx) {
  Comm* c = ctx->comm;
  if (!c) return;
  (void)hipSetDevice(ctx->device);
  if (c->nccl) {
    if (RcclApi* api = rccl_api(nullptr)) {
      // a communicator with a collective that will never complete is torn down without waiting for it
      if (c->failed && api->CommAbort) (void)api->CommAbort(c->nccl);
      else (void)api->CommDestroy(c->nccl);
    }
  }
  if (c->k1_ev) (void)hipEventDestroy(c->k1_ev);
  if (c->done_ev) (void)hipEventDestroy(c->done_ev);
  if (c->pulled_ev) (void)hipEventDestroy(c->pulled_ev);
  if (c->gathered_ev) (void)hipEventDestroy(c->gathered_ev);
  delete c;
  ctx->comm = nullptr;
}

// Waits for the context's stream like hipStreamSynchronize, but with a deadline when the stream may hold collectives
// of an RCCL communicator with other ranks: a peer that never arrives (crashed rank, mismatched call order, a link
// that went away) would otherwise park this process in the driver for ever.  On expiry -- or when RCCL reports an
// asynchronous error -- the call returns JXLH_ERR_DEVICE and jxlh_last_error names rank, world size and the last
// collective that was enqueued; the stream is left as it is (the caller tears the job down).
jxlh_status comm_wait_stream(jxlh_ctx* ctx) {
  Comm* c = ctx->comm;
  if (!c || !c->nccl || c->nranks <= 1 || c->timeout_s <= 0.0) {
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return JXLH_OK;
  }
  if (c->failed) {
    ctx->last_error = "rank " + std::to_string(c->rank) + " of " + std::to_string(c->nranks) +
                      ": the communicator failed earlier (" + c->last_op + "); destroy it (jxlh_comm_destroy / jxlh_ctx_destroy)";
    return JXLH_ERR_DEVICE;
  }
  RcclApi* api = rccl_api(nullptr);
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned spins = 0;; spins++) {
    const hipError_t q = hipStreamQuery(ctx->stream);
    if (q == hipSuccess) return JXLH_OK;
    if (q != hipErrorNotReady) return fail(ctx, q, "hipStreamQuery");
    (void)hipGetLastError();
    if (api && api->CommGetAsyncError) {
      ncclResult_t ar = ncclSuccess;
      if (api->CommGetAsyncError(c->nccl, &ar) == ncclSuccess && ar != ncclSuccess && ar != ncclInProgress) {
        ctx->last_error = "rank " + std::to_string(c->rank) + " of " + std::to_string(c->nranks) + ": RCCL asynchronous error (" +
                          (api->GetErrorString ? api->GetErrorString(ar) : "?") + ") after: " + c->last_op;
        c->failed = true;
        return JXLH_ERR_DEVICE;
      }
    }
    const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (el > c->timeout_s) {
      ctx->last_error = "rank " + std::to_string(c->rank) + " of " + std::to_string(c->nranks) + ": the stream did not finish within " +
                        std::to_string((int)c->timeout_s) + " s (JXLH_COMM_TIMEOUT_S); last collective enqueued: " +
                        (c->last_op.empty() ? std::string("none") : c->last_op);
      c->failed = true;
      return JXLH_ERR_DEVICE;
    }
    if (spins > 2000) std::this_thread::sleep_for(std::chrono::microseconds(spins > 20000 ? 1000 : 50));
  }
}

int comm_nranks(const jxlh_ctx* ctx) { return ctx->comm ? ctx->comm->nranks : 1; }

// contiguous bands: ceil(ygroups / nranks) group rows per rank; trailing ranks may own nothing
static void band_of(int ygroups, int nranks, int rank, int* r0, int* r1) {
  const int per = (ygroups + nranks - 1) / nranks;
  *r0 = std::min(rank * per, ygroups);
  *r1 = std::min((rank + 1) * per, ygroups);
}
int comm_rows_per_rank(const jxlh_ctx* ctx, int ygroups) {
  const int n = comm_nranks(ctx);
  return (ygroups + n - 1) / n;
}

// one block row (8 pixel rows) of plane c in the layout K1 just wrote: offset and length in floats
static void block_row_span(const FrameDev& f, int by, size_t* off, size_t* count) {
  if (f.tiled) {
    *off = (size_t)by * f.xblocks * 64;
    *count = (size_t)f.xblocks * 64;
  } else {
    *off = (size_t)by * 8 * f.plane_stride;
    *count = 8 * f.plane_stride;
  }
}

static bool exchange_applies(const jxlh_ctx* ctx, const RunPlan& plan) {
  // chroma-subsampled frames keep the recomputed halo group row (their vertical upsampling reads across the band
  // edge in the sub-sampled domain); frames without filters need no halo at all
  return plan.halo_px > 0 && !ctx->fd.subsampled;
}

// transforms of the own band (plus recomputed halo group rows when the exchange does not apply)
static jxlh_status shard_k1(jxlh_ctx* ctx, RunPlan* plan, int* r0, int* r1) {
  if (!ctx->in_frame || !ctx->tables_set) return JXLH_ERR_BAD_STATE;
  if (ctx->params.upsampling > 1) return JXLH_ERR_UNSUPPORTED;  // the 5x5 upsampling window crosses bands: run whole
  const Comm* c = ctx->comm;
  band_of(ctx->fd.ygroups, c->nranks, c->rank, r0, r1);
  if (jxlh_status st = run_prologue(ctx, plan)) return st;
  if (*r0 >= *r1) return JXLH_OK;
  int g0 = *r0, g1 = *r1;
  if (!exchange_applies(ctx, *plan) && (plan->halo_px > 0 || ctx->fd.subsampled)) {
    g0 = std::max(0, g0 - 1);
    g1 = std::min(ctx->fd.ygroups, g1 + 1);
  }
  return run_k1(ctx, *plan, g0, g1);
}

}  // namespace jxlh_host

extern "C" {

jxlh_status jxlh_comm_unique_id(uint8_t id[JXLH_COMM_ID_BYTES]) {
  if (!id) return JXLH_ERR_INVALID_ARGUMENT;
  static_assert(JXLH_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
  RcclApi* api = rccl_api(nullptr);
  if (!api) return JXLH_ERR_DEVICE;
  ncclUniqueId u;
  if (api->GetUniqueId(&u) != ncclSuccess) return JXLH_ERR_DEVICE;
  std::memcpy(id, u.internal, JXLH_COMM_ID_BYTES);
  return JXLH_OK;
}

jxlh_status jxlh_comm_init(jxlh_ctx* ctx, const uint8_t id[JXLH_COMM_ID_BYTES], int32_t rank, int32_t nranks) {
  JXLH_ON_DEVICE(ctx);
  if (!ctx || !id || nranks < 1 || rank < 0 || rank >= nranks) return JXLH_ERR_INVALID_ARGUMENT;
  if (ctx->comm || ctx->in_frame) return JXLH_ERR_BAD_STATE;  // before the first jxlh_frame_begin: it sizes the planes
  std::string err;
  RcclApi* api = rccl_api(&err);
  if (!api) {
    ctx->last_error = err;
    return JXLH_ERR_DEVICE;
  }
  HIPCHK(ctx, hipSetDevice(ctx->device));
  Comm* c = new (std::nothrow) Comm();
  if (!c) return JXLH_ERR_OUT_OF_MEMORY;
  c->rank = rank;
  c->nranks = nranks;
  if (const char* e = getenv("JXLH_COMM_TIMEOUT_S")) c->timeout_s = atof(e);
  ncclUniqueId u;
  std::memcpy(u.internal, id, JXLH_COMM_ID_BYTES);
  const ncclResult_t r = api->CommInitRank(&c->nccl, nranks, u, rank);
  if (r != ncclSuccess) {
    delete c;
    return nccl_fail(ctx, api, r, "ncclCommInitRank");
  }
  ctx->comm = c;
  return JXLH_OK;
}

jxlh_status jxlh_comm_init_local(jxlh_ctx* const peers[], int32_t nranks) {
  if (!peers || nranks < 1) return JXLH_ERR_INVALID_ARGUMENT;
  for (int i = 0; i < nranks; i++) {
    if (!peers[i]) return JXLH_ERR_INVALID_ARGUMENT;
    if (peers[i]->comm || peers[i]->in_frame) return JXLH_ERR_BAD_STATE;
    for (int k = 0; k < i; k++)
      if (peers[k] == peers[i]) return JXLH_ERR_INVALID_ARGUMENT;
  }
  for (int i = 0; i < nranks; i++) {
    jxlh_ctx* ctx = peers[i];
    HIPCHK(ctx, hipSetDevice(ctx->device));
    Comm* c = new (std::nothrow) Comm();
    if (!c) return JXLH_ERR_OUT_OF_MEMORY;
    c->rank = i;
    c->nranks = nranks;
    c->peers.assign(peers, peers + nranks);
    if (hipEventCreateWithFlags(&c->k1_ev, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->done_ev, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->pulled_ev, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->gathered_ev, hipEventDisableTiming) != hipSuccess) {
      delete c;
      return JXLH_ERR_DEVICE;
    }
    // peers on other GPUs: direct access for the copies (already-enabled is fine)
    for (int k = 0; k < nranks; k++) {
      if (peers[k]->device != ctx->device) {
        (void)hipDeviceEnablePeerAccess(peers[k]->device, 0);
        (void)hipGetLastError();
      }
    }
    ctx->comm = c;
  }
  return JXLH_OK;
}

jxlh_status jxlh_comm_destroy(jxlh_ctx* ctx) {
  JXLH_ON_DEVICE(ctx);
  if (!ctx) return JXLH_ERR_INVALID_ARGUMENT;
  if (!ctx->comm) return JXLH_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  comm_release(ctx);
  return JXLH_OK;
}

jxlh_status jxlh_comm_band(jxlh_ctx* ctx, int32_t* rank, int32_t* nranks, uint32_t* group_row0, uint32_t* group_row1) {
  JXLH_ON_DEVICE(ctx);
  if (!ctx) return JXLH_ERR_INVALID_ARGUMENT;
  const int n = ctx->comm ? ctx->comm->nranks : 1, r = ctx->comm ? ctx->comm->rank : 0;
  if (rank) *rank = r;
  if (nranks) *nranks = n;
  if (group_row0 || group_row1) {
    if (!ctx->in_frame) return JXLH_ERR_BAD_STATE;
    int r0, r1;
    band_of(ctx->fd.ygroups, n, r, &r0, &r1);
    if (group_row0) *group_row0 = (uint32_t)r0;
    if (group_row1) *group_row1 = (uint32_t)r1;
  }
  return JXLH_OK;
}

// ---- RCCL transport ------------------------------------------------------------------------------------------
jxlh_status jxlh_frame_run_sharded(jxlh_ctx* ctx) {
  JXLH_ON_DEVICE(ctx);
  if (!ctx) return JXLH_ERR_INVALID_ARGUMENT;
  if (!ctx->comm || !ctx->comm->nccl) return JXLH_ERR_BAD_STATE;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  Comm* c = ctx->comm;
  RcclApi* api = rccl_api(nullptr);
  RunPlan plan;
  int r0, r1;
  if (jxlh_status st = shard_k1(ctx, &plan, &r0, &r1)) return st;
  const FrameDev& f = ctx->fd;
  if (exchange_applies(ctx, plan) && c->nranks > 1) {
    // neighbours = the adjacent NON-EMPTY bands (empty bands only trail)
    int up0, up1, dn0, dn1;
    const bool mine = r0 < r1;
    bool has_up = false, has_dn = false;
    if (mine && c->rank > 0) {
      band_of(f.ygroups, c->nranks, c->rank - 1, &up0, &up1);
      has_up = up0 < up1;
    }
    if (mine && c->rank + 1 < c->nranks) {
      band_of(f.ygroups, c->nranks, c->rank + 1, &dn0, &dn1);
      has_dn = dn0 < dn1;
    }
    if (has_up || has_dn) {
      c->last_op = std::string("halo exchange of the band edges (ncclSend/ncclRecv with rank") + (has_up ? " " + std::to_string(c->rank - 1) : "") +
                   (has_dn ? " " + std::to_string(c->rank + 1) : "") + ", group rows " + std::to_string(r0) + ".." + std::to_string(r1) + ")";
      NCCLCHK(ctx, api, api->GroupStart());
      for (int ch = 0; ch < 3; ch++) {
        size_t off, cnt;
        if (has_up) {  // my first block row goes up, the block row above my band comes down
          block_row_span(f, r0 * kGroupBlocks, &off, &cnt);
          NCCLCHK_IN_GROUP(ctx, api, api->Send(f.planes[ch] + off, cnt, ncclFloat32, c->rank - 1, c->nccl, ctx->stream));
          block_row_span(f, r0 * kGroupBlocks - 1, &off, &cnt);
          NCCLCHK_IN_GROUP(ctx, api, api->Recv(f.planes[ch] + off, cnt, ncclFloat32, c->rank - 1, c->nccl, ctx->stream));
        }
        if (has_dn) {
          block_row_span(f, r1 * kGroupBlocks - 1, &off, &cnt);
          NCCLCHK_IN_GROUP(ctx, api, api->Send(f.planes[ch] + off, cnt, ncclFloat32, c->rank + 1, c->nccl, ctx->stream));
          block_row_span(f, r1 * kGroupBlocks, &off, &cnt);
          NCCLCHK_IN_GROUP(ctx, api, api->Recv(f.planes[ch] + off, cnt, ncclFloat32, c->rank + 1, c->nccl, ctx->stream));
        }
      }
      NCCLCHK(ctx, api, api->GroupEnd());
    }
  }
  if (r0 >= r1) return JXLH_OK;
  return run_stages(ctx, plan, (uint32_t)r0, (uint32_t)r1);
}

jxlh_status jxlh_frame_allgather(jxlh_ctx* ctx) {
  JXLH_ON_DEVICE(ctx);
  if (!ctx) return JXLH_ERR_INVALID_ARGUMENT;
  if (!ctx->comm || !ctx->comm->nccl || !ctx->in_frame) return JXLH_ERR_BAD_STATE;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  Comm* c = ctx->comm;
  RcclApi* api = rccl_api(nullptr);
  const FrameDev& f = ctx->fd;
  // a rank whose band is empty ran no stage: the frame's stage list says where the result lives
  float* res[3];
  for (int ch = 0; ch < 3; ch++) res[ch] = ctx->result[ch] = result_in_tmp(ctx) ? f.tmp[ch] : f.planes[ch];
  ctx->res_w = f.xsize;
  ctx->res_h = f.ysize;
  ctx->res_stride = f.plane_stride;
  const size_t count = (size_t)comm_rows_per_rank(ctx, f.ygroups) * kGroupDim * f.plane_stride;
  c->last_op = "ncclAllGather of the finished planes (" + std::to_string(count * 4) + " bytes per rank and plane)";
  NCCLCHK(ctx, api, api->GroupStart());
  for (int ch = 0; ch < 3; ch++)
    NCCLCHK_IN_GROUP(ctx, api, api->AllGather(res[ch] + (size_t)c->rank * count, res[ch], count, ncclFloat32, c->nccl, ctx->stream));
  NCCLCHK(ctx, api, api->GroupEnd());
  return JXLH_OK;
}

// where a band run left the finished planes + the geometry the read-out stages use
static void set_band_result(jxlh_ctx* ctx) {
  const FrameDev& f = ctx->fd;
  for (int ch = 0; ch < 3; ch++) ctx->result[ch] = result_in_tmp(ctx) ? f.tmp[ch] : f.planes[ch];
  ctx->res_w = f.xsize;
  ctx->res_h = f.ysize;
  ctx->res_stride = f.plane_stride;
}

jxlh_status jxlh_frame_allgather_output(jxlh_ctx* ctx, const jxlh_output_desc* d, void* out, size_t bytes_per_row) {
  JXLH_ON_DEVICE(ctx);
  if (!ctx || !d || !out) return JXLH_ERR_INVALID_ARGUMENT;
  if (!ctx->comm || !ctx->comm->nccl || !ctx->in_frame) return JXLH_ERR_BAD_STATE;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  Comm* c = ctx->comm;
  RcclApi* api = rccl_api(nullptr);
  const FrameDev& f = ctx->fd;
  if (bytes_per_row < (size_t)f.xsize * d->channels * (d->bits / 8)) return JXLH_ERR_INVALID_ARGUMENT;
  set_band_result(ctx);
  int r0, r1;
  band_of(f.ygroups, c->nranks, c->rank, &r0, &r1);
  if (jxlh_status st = convert_band_to_output(ctx, d, (uint32_t)std::min(r0 * kGroupDim, f.ysize),
                                              (uint32_t)std::min(r1 * kGroupDim, f.ysize), out, bytes_per_row))
    return st;
  const size_t count = (size_t)comm_rows_per_rank(ctx, f.ygroups) * kGroupDim * bytes_per_row;
  c->last_op = "ncclAllGather of the converted image (" + std::to_string(count) + " bytes per rank)";
  NCCLCHK(ctx, api, api->AllGather(static_cast<char*>(out) + (size_t)c->rank * count, out, count, ncclUint8, c->nccl, ctx->stream));
  return JXLH_OK;
}

jxlh_status jxlh_frames_allgather_output_local(jxlh_ctx* const peers[], int32_t n, const jxlh_output_desc* d,
                                               void* const outs[], size_t bytes_per_row) {
  if (!peers || !outs || !d || n < 1) return JXLH_ERR_INVALID_ARGUMENT;
  for (int i = 0; i < n; i++)
    if (!peers[i] || !outs[i] || !peers[i]->comm || peers[i]->comm->nccl || peers[i]->comm->nranks != n || !peers[i]->in_frame)
      return JXLH_ERR_BAD_STATE;
  // every rank converts its band into its own image buffer ...
  for (int i = 0; i < n; i++) {
    jxlh_ctx* c = peers[i];
    HIPCHK(c, hipSetDevice(c->device));
    const FrameDev& f = c->fd;
    if (bytes_per_row < (size_t)f.xsize * d->channels * (d->bits / 8)) return JXLH_ERR_INVALID_ARGUMENT;
    set_band_result(c);
    int r0, r1;
    band_of(f.ygroups, n, i, &r0, &r1);
    if (jxlh_status st = convert_band_to_output(c, d, (uint32_t)std::min(r0 * kGroupDim, f.ysize),
                                                (uint32_t)std::min(r1 * kGroupDim, f.ysize), outs[i], bytes_per_row))
      return st;
    HIPCHK(c, hipEventRecord(c->comm->done_ev, c->stream));
  }
  // ... and pulls the other ranks' bands
  for (int i = 0; i < n; i++) {
    jxlh_ctx* dst = peers[i];
    HIPCHK(dst, hipSetDevice(dst->device));
    const FrameDev& f = dst->fd;
    for (int k = 0; k < n; k++) {
      if (k == i) continue;
      int r0, r1;
      band_of(f.ygroups, n, k, &r0, &r1);
      const size_t y0 = (size_t)std::min(r0 * kGroupDim, f.ysize), y1 = (size_t)std::min(r1 * kGroupDim, f.ysize);
      if (y0 >= y1) continue;
      HIPCHK(dst, hipStreamWaitEvent(dst->stream, peers[k]->comm->done_ev, 0));
      HIPCHK(dst, hipMemcpyAsync(static_cast<char*>(outs[i]) + y0 * bytes_per_row,
                                 static_cast<const char*>(outs[k]) + y0 * bytes_per_row, (y1 - y0) * bytes_per_row,
                                 hipMemcpyDefault, dst->stream));
    }
  }
  return JXLH_OK;
}

jxlh_status jxlh_comm_allgather(jxlh_ctx* ctx, void* buf, size_t bytes_per_rank) {
  JXLH_ON_DEVICE(ctx);
  if (!ctx || !buf) return JXLH_ERR_INVALID_ARGUMENT;
  if (!ctx->comm || !ctx->comm->nccl) return JXLH_ERR_BAD_STATE;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  Comm* c = ctx->comm;
  RcclApi* api = rccl_api(nullptr);
  c->last_op = "jxlh_comm_allgather (" + std::to_string(bytes_per_rank) + " bytes per rank)";
  NCCLCHK(ctx, api, api->AllGather(static_cast<char*>(buf) + (size_t)c->rank * bytes_per_rank, buf, bytes_per_rank,
                                   ncclUint8, c->nccl, ctx->stream));
  return JXLH_OK;
}

jxlh_status jxlh_comm_allgather_local(jxlh_ctx* const peers[], int32_t n, void* const bufs[], size_t bytes_per_rank) {
  if (!peers || !bufs || n < 1) return JXLH_ERR_INVALID_ARGUMENT;
  for (int i = 0; i < n; i++)
    if (!peers[i] || !bufs[i] || !peers[i]->comm || peers[i]->comm->nccl || peers[i]->comm->nranks != n ||
        peers[i]->comm->rank != i)
      return JXLH_ERR_BAD_STATE;
  // every rank's part is final once its stream reaches this point ...
  for (int i = 0; i < n; i++) {
    HIPCHK(peers[i], hipSetDevice(peers[i]->device));
    HIPCHK(peers[i], hipEventRecord(peers[i]->comm->done_ev, peers[i]->stream));
  }
  // ... and every rank pulls the other ranks' parts
  for (int i = 0; i < n; i++) {
    jxlh_ctx* dst = peers[i];
    HIPCHK(dst, hipSetDevice(dst->device));
    for (int k = 0; k < n; k++) {
      if (k == i) continue;
      HIPCHK(dst, hipStreamWaitEvent(dst->stream, peers[k]->comm->done_ev, 0));
      HIPCHK(dst, hipMemcpyAsync(static_cast<char*>(bufs[i]) + (size_t)k * bytes_per_rank,
                                 static_cast<const char*>(bufs[k]) + (size_t)k * bytes_per_rank, bytes_per_rank,
                                 hipMemcpyDefault, dst->stream));
    }
  }
  return JXLH_OK;
}

// ---- local transport -----------------------------------------------------------------------------------------
jxlh_status jxlh_frames_run_sharded_local(jxlh_ctx* const peers[], int32_t n) {
  if (!peers || n < 1) return JXLH_ERR_INVALID_ARGUMENT;
  for (int i = 0; i < n; i++)
    if (!peers[i] || !peers[i]->comm || peers[i]->comm->nccl || peers[i]->comm->nranks != n || peers[i]->comm->rank != i)
      return JXLH_ERR_BAD_STATE;
  std::vector<RunPlan> plan(n);
  std::vector<int> r0(n), r1(n);
  // phase 1: every rank's transforms
  for (int i = 0; i < n; i++) {
    jxlh_ctx* ctx = peers[i];
    HIPCHK(ctx, hipSetDevice(ctx->device));
    for (int k = 0; k < n; k++)  // the previous frame's gather may still be reading this rank's band
      if (k != i && peers[k]->comm->gathered_valid) HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, peers[k]->comm->gathered_ev, 0));
    if (jxlh_status st = shard_k1(ctx, &plan[i], &r0[i], &r1[i])) return st;
    HIPCHK(ctx, hipEventRecord(ctx->comm->k1_ev, ctx->stream));
  }
  // phase 2: every rank pulls the edge block rows it reads across its band edges from the neighbours ...
  for (int i = 0; i < n; i++) {
    jxlh_ctx* ctx = peers[i];
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const FrameDev& f = ctx->fd;
    if (r0[i] < r1[i] && exchange_applies(ctx, plan[i])) {
      for (int side = 0; side < 2; side++) {
        const int nb = side == 0 ? i - 1 : i + 1;
        if (nb < 0 || nb >= n || r0[nb] >= r1[nb]) continue;
        jxlh_ctx* src = peers[nb];
        if (src->fd.tiled != f.tiled || src->fd.xblocks != f.xblocks || src->fd.plane_stride != f.plane_stride)
          return JXLH_ERR_BAD_STATE;  // the peers decode the same frame
        HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, src->comm->k1_ev, 0));
        const int by = side == 0 ? r0[i] * kGroupBlocks - 1 : r1[i] * kGroupBlocks;
        size_t off, cnt;
        block_row_span(f, by, &off, &cnt);
        for (int ch = 0; ch < 3; ch++)
          HIPCHK(ctx, hipMemcpyAsync(f.planes[ch] + off, src->fd.planes[ch] + off, cnt * sizeof(float), hipMemcpyDefault,
                                     ctx->stream));
      }
    }
    HIPCHK(ctx, hipEventRecord(ctx->comm->pulled_ev, ctx->stream));
  }
  // ... phase 3: and filters its band once the neighbours have taken their copies (a stage list that ends in
  // `planes` overwrites the rows they read)
  for (int i = 0; i < n; i++) {
    jxlh_ctx* ctx = peers[i];
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (r0[i] >= r1[i]) continue;
    if (i > 0) HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, peers[i - 1]->comm->pulled_ev, 0));
    if (i + 1 < n) HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, peers[i + 1]->comm->pulled_ev, 0));
    if (jxlh_status st = run_stages(ctx, plan[i], (uint32_t)r0[i], (uint32_t)r1[i])) return st;
  }
  for (int i = 0; i < n; i++) {
    jxlh_ctx* ctx = peers[i];
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipEventRecord(ctx->comm->done_ev, ctx->stream));
  }
  return JXLH_OK;
}

jxlh_status jxlh_frames_allgather_local(jxlh_ctx* const peers[], int32_t n) {
  if (!peers || n < 1) return JXLH_ERR_INVALID_ARGUMENT;
  for (int i = 0; i < n; i++)
    if (!peers[i] || !peers[i]->comm || peers[i]->comm->nccl || peers[i]->comm->nranks != n || !peers[i]->in_frame)
      return JXLH_ERR_BAD_STATE;
  const int which = result_in_tmp(peers[0]);  // a property of the stage list, identical on every peer
  for (int i = 0; i < n; i++) {
    jxlh_ctx* dst = peers[i];
    HIPCHK(dst, hipSetDevice(dst->device));
    const FrameDev& f = dst->fd;
    for (int c = 0; c < 3; c++) dst->result[c] = which ? f.tmp[c] : f.planes[c];
    dst->res_w = f.xsize;
    dst->res_h = f.ysize;
    dst->res_stride = f.plane_stride;
    for (int k = 0; k < n; k++) {
      if (k == i) continue;
      jxlh_ctx* src = peers[k];
      int r0, r1;
      band_of(f.ygroups, n, k, &r0, &r1);
      if (r0 >= r1) continue;
      HIPCHK(dst, hipStreamWaitEvent(dst->stream, src->comm->done_ev, 0));
      const size_t off = (size_t)r0 * kGroupDim * f.plane_stride;
      const size_t rows = (size_t)std::min(r1 * kGroupDim, f.ysize) - (size_t)r0 * kGroupDim;
      for (int c = 0; c < 3; c++) {
        const float* s = (which ? src->fd.tmp[c] : src->fd.planes[c]) + off;
        HIPCHK(dst, hipMemcpyAsync(dst->result[c] + off, s, rows * f.plane_stride * sizeof(float), hipMemcpyDefault,
                                   dst->stream));
      }
    }
    HIPCHK(dst, hipEventRecord(dst->comm->gathered_ev, dst->stream));
    dst->comm->gathered_valid = true;
  }
  return JXLH_OK;
}

}  // extern "C"
