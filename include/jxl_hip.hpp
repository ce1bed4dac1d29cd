// C++ host side over the C ABI of include/jxl_hip.h: the calls a jxl-rs maintainer's shim makes, under the names of
// the reference functions they replace, with RAII and exceptions instead of status codes.  Header-only, no HIP or
// torch types: links against libjxl_hip.so only.  (The reference is Rust; INTEGRATION.md shows the same surface as an
// `extern "C"` block and a `RenderPipeline` implementation.  This header is what the compiled-language parity test
// tests/cpp/frame_parity.cc drives.)
//
//   reference (jxl/src/...)                                   here
//   Frame::from_header_and_toc + prepare_render_pipeline      VarDctFrame::begin            frame/decode.rs:172-204, frame/render.rs:907
//   decode_hf_global (dequant matrices)                       VarDctFrame::decode_hf_global frame/quant_weights.rs:347-351
//   decode_lf_group -> dequant_lf                             VarDctFrame::decode_lf_group  frame/modular/mod.rs:837-929
//   decode_hf_metadata                                        VarDctFrame::decode_hf_metadata  frame/modular/mod.rs:984-1081
//   decode_vardct_group (entropy loop stays on the host)      VarDctFrame::decode_vardct_group[_sparse]  frame/group.rs:509-613
//   finalize_lf, SigmaSource::new, the render pipeline        VarDctFrame::finalize_and_render  frame/mod.rs:360-378, frame/render.rs:569-683
//   pipeline output (save stages)                             VarDctFrame::read_planes / read_rgb8 / read_output
#pragma once
#include <array>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "jxl_hip.h"

namespace jxlh {

class Error : public std::runtime_error {  // -> Error::Gpu(code) on the Rust side
 public:
  Error(jxlh_status st, const char* where, const std::string& detail)
      : std::runtime_error(std::string(where) + ": " + jxlh_status_string(st) + (detail.empty() ? "" : " / " + detail)),
        status(st) {}
  jxlh_status status;
};

class Context {
 public:
  explicit Context(int device = 0, int n_slots = 1) : n_slots_(n_slots) {
    const jxlh_status st = jxlh_ctx_create(device, n_slots, &c_);
    if (st != JXLH_OK) throw Error(st, "jxlh_ctx_create", "");
  }
  ~Context() { jxlh_ctx_destroy(c_); }
  Context(const Context&) = delete;
  Context& operator=(const Context&) = delete;
  jxlh_ctx* raw() const { return c_; }
  int n_slots() const { return n_slots_; }
  void check(jxlh_status st, const char* where) const {
    if (st != JXLH_OK) throw Error(st, where, jxlh_last_error(c_));
  }
  void sync() { check(jxlh_ctx_sync(c_), "jxlh_ctx_sync"); }
  void* alloc_pinned(size_t bytes) {
    void* p = nullptr;
    check(jxlh_alloc_pinned(c_, bytes, &p), "jxlh_alloc_pinned");
    return p;
  }
  void free_pinned(void* p) { check(jxlh_free_pinned(c_, p), "jxlh_free_pinned"); }

 private:
  jxlh_ctx* c_ = nullptr;
  int n_slots_ = 1;
};

// One VarDCT frame on the device: the order of calls is the order of Frame's sections in the codestream.
class VarDctFrame {
 public:
  static jxlh_frame_params default_params(uint32_t xsize, uint32_t ysize) {
    jxlh_frame_params p;
    const jxlh_status st = jxlh_default_frame_params(&p, xsize, ysize);
    if (st != JXLH_OK) throw Error(st, "jxlh_default_frame_params", "");
    return p;
  }
  VarDctFrame(Context& ctx, const jxlh_frame_params& p) : ctx_(ctx), p_(p) {
    ctx_.check(jxlh_frame_begin(ctx_.raw(), &p_), "jxlh_frame_begin");
  }
  const jxlh_frame_params& params() const { return p_; }

  void decode_hf_global(const std::array<std::vector<float>, JXLH_NUM_QUANT_TABLES>& tables) {
    const float* ptr[JXLH_NUM_QUANT_TABLES];
    size_t n[JXLH_NUM_QUANT_TABLES];
    for (int t = 0; t < JXLH_NUM_QUANT_TABLES; t++) {
      ptr[t] = tables[t].data();
      n[t] = tables[t].size() / 3;
    }
    ctx_.check(jxlh_frame_set_dequant_tables(ctx_.raw(), ptr, n), "jxlh_frame_set_dequant_tables");
  }
  // rect in blocks; the three modular channels in coded order Y, X, B
  void decode_lf_group(uint32_t x0, uint32_t y0, uint32_t w, uint32_t h, const int32_t* qy, const int32_t* qx,
                       const int32_t* qb, size_t stride, uint32_t extra_precision = 0) {
    ctx_.check(jxlh_frame_set_lf_quantized(ctx_.raw(), x0, y0, w, h, qy, qx, qb, stride, extra_precision),
               "jxlh_frame_set_lf_quantized");
  }
  void decode_hf_metadata(uint32_t x0, uint32_t y0, uint32_t w, uint32_t h, const uint8_t* transform_map,
                          const int32_t* raw_quant, const uint8_t* epf_map, size_t map_stride, const int8_t* ytox,
                          const int8_t* ytob, size_t cmap_stride) {
    ctx_.check(jxlh_frame_set_hf_meta(ctx_.raw(), x0, y0, w, h, transform_map, raw_quant, epf_map, map_stride, ytox, ytob,
                                      cmap_stride),
               "jxlh_frame_set_hf_meta");
  }
  // the group's dense coefficient slab (3 x 65536 i32) as decode_vardct_group fills it; asynchronous per slot
  void decode_vardct_group(uint32_t group, const int32_t* coeffs, int slot = 0) {
    ctx_.check(jxlh_submit_group(ctx_.raw(), slot, group, coeffs, JXLH_GROUP_COMPLETE), "jxlh_submit_group");
  }
  // ... or the (position, value) updates its entropy loop produces (frame/group.rs:557-572)
  void decode_vardct_group_sparse(uint32_t group, const jxlh_coeff16* pairs, const uint32_t n[3],
                                  const jxlh_coeff32* wide = nullptr, uint32_t n_wide = 0, int slot = 0) {
    ctx_.check(jxlh_submit_group_sparse(ctx_.raw(), slot, group, pairs, n, wide, n_wide, JXLH_GROUP_COMPLETE),
               "jxlh_submit_group_sparse");
  }
  void slot_wait(int slot = 0) { ctx_.check(jxlh_slot_wait(ctx_.raw(), slot), "jxlh_slot_wait"); }
  // finalize_lf + SigmaSource::new + transforms + the frame's stage list, for group rows [row0, row1)
  void finalize_and_render(uint32_t group_row0 = 0, uint32_t group_row1 = 0xFFFFFFFFu) {
    ctx_.check(jxlh_frame_run(ctx_.raw(), group_row0, group_row1), "jxlh_frame_run");
  }
  // tight f32 planes X, Y, B of out_width() x out_height()
  void read_planes(float* x, float* y, float* b) {
    const size_t w = out_width(), h = out_height();
    const jxlh_plane pl[3] = {{x, w * sizeof(float), h, w * sizeof(float)},
                              {y, w * sizeof(float), h, w * sizeof(float)},
                              {b, w * sizeof(float), h, w * sizeof(float)}};
    ctx_.check(jxlh_frame_read_planes(ctx_.raw(), pl), "jxlh_frame_read_planes");
  }
  // one 256 x 256 group of the result per channel, the unit RenderPipeline::set_buffer_for_group moves
  // (render/mod.rs:124-137); buffers of `pitch` floats per row, at least the group's size rounded up to 16 pixels
  void read_group_planes(uint32_t group, float* x, float* y, float* b, size_t pitch, size_t rows) {
    const uint32_t xg = (out_width() + JXLH_GROUP_DIM - 1) / JXLH_GROUP_DIM;
    const jxlh_plane pl[3] = {{x, pitch * sizeof(float), rows, pitch * sizeof(float)},
                              {y, pitch * sizeof(float), rows, pitch * sizeof(float)},
                              {b, pitch * sizeof(float), rows, pitch * sizeof(float)}};
    ctx_.check(jxlh_frame_read_planes_rect(ctx_.raw(), (group % xg) * JXLH_GROUP_DIM, (group / xg) * JXLH_GROUP_DIM,
                                           JXLH_GROUP_DIM, JXLH_GROUP_DIM, pl),
               "jxlh_frame_read_planes_rect");
  }
  void read_rgb8(const jxlh_xyb_params& xyb, uint32_t channels, uint8_t* out) {
    ctx_.check(jxlh_frame_read_rgb8(ctx_.raw(), &xyb, channels, 0, out_height(), out, (size_t)out_width() * channels),
               "jxlh_frame_read_rgb8");
  }
  void read_output(const jxlh_output_desc& d, void* out) {
    ctx_.check(jxlh_frame_read_output(ctx_.raw(), &d, 0, out_height(), out,
                                      (size_t)out_width() * d.channels * (d.bits / 8)),
               "jxlh_frame_read_output");
  }
  uint32_t out_width() const {
    const uint32_t n = p_.upsampling > 1 ? p_.upsampling : 1;
    return p_.xsize_upsampled ? p_.xsize_upsampled : p_.xsize * n;
  }
  uint32_t out_height() const {
    const uint32_t n = p_.upsampling > 1 ? p_.upsampling : 1;
    return p_.ysize_upsampled ? p_.ysize_upsampled : p_.ysize * n;
  }

 private:
  Context& ctx_;
  jxlh_frame_params p_;
};

}  // namespace jxlh
