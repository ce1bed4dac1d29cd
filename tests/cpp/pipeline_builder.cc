// The C++ mirror of the reference's stage traits + RenderPipelineBuilder (include/jxl_hip_pipeline.hpp) from compiled
// code.  Built and run by tests/test_cpp_host.py.
//   pipeline_builder host           the lowering of stage lists onto jxlh_frame_params / jxlh_output_desc and the
//                                   rejection of lists outside the device path -- pure host logic, no GPU
//   pipeline_builder modular W H    Modular stage lists: the builder's I32 -> U8 special case, the f32 conversion, f32 + filters
//   pipeline_builder gpu W H ITERS  a frame assembled exactly as Frame::build_render_pipeline assembles it
//                                   (frame/render.rs:526-790), decoded through GpuRenderPipeline in two passes
//                                   (incomplete groups, then complete + mark_group_to_rerender), against the oracle
#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <string>

#include <hip/hip_runtime_api.h>

#include "jxl_hip_pipeline.hpp"
#include "synth_frame.hpp"

using namespace jxlh;

namespace {
int g_failed = 0;
void expect(bool ok, const char* what) {
  if (!ok) {
    g_failed++;
    fprintf(stderr, "FAILED: %s\n", what);
  }
}
// the status build() / lower() fails with, or JXLH_OK
jxlh_status status_of(const std::function<void()>& f, std::string* msg = nullptr) {
  try {
    f();
  } catch (const Error& e) {
    if (msg) *msg = e.what();
    return e.status;
  }
  return JXLH_OK;
}

// RestorationFilter defaults as jxlh_default_frame_params holds them
struct Rf {
  float gab_w1[3], gab_w2[3], pass0, pass2, border_sad_mul;
  std::array<float, 3> channel_scale;
};
Rf rf_of(const jxlh_frame_params& p) {
  Rf r;
  for (int c = 0; c < 3; c++) {
    r.gab_w1[c] = p.gab_w1[c];
    r.gab_w2[c] = p.gab_w2[c];
    r.channel_scale[c] = p.epf_channel_scale[c];
  }
  r.pass0 = p.epf_pass0_sigma_scale;
  r.pass2 = p.epf_pass2_sigma_scale;
  r.border_sad_mul = p.epf_border_sad_mul;
  return r;
}

// the filter part of Frame::build_render_pipeline (frame/render.rs:578-622)
RenderPipelineBuilder add_filters(RenderPipelineBuilder b, const Rf& rf, bool gab, int epf_iters) {
  if (gab) {
    b = std::move(b)
            .add_inout_stage(GaborishStage{0, rf.gab_w1[0], rf.gab_w2[0]})
            .add_inout_stage(GaborishStage{1, rf.gab_w1[1], rf.gab_w2[1]})
            .add_inout_stage(GaborishStage{2, rf.gab_w1[2], rf.gab_w2[2]});
  }
  if (epf_iters >= 3) b = std::move(b).add_inout_stage(Epf0Stage{rf.pass0, rf.border_sad_mul, rf.channel_scale});
  if (epf_iters >= 1) b = std::move(b).add_inout_stage(Epf1Stage{1.0f, rf.border_sad_mul, rf.channel_scale});
  if (epf_iters >= 2) b = std::move(b).add_inout_stage(Epf2Stage{rf.pass2, rf.border_sad_mul, rf.channel_scale});
  return b;
}

jxlh_xyb_params some_xyb() {
  jxlh_xyb_params x{};
  for (int i = 0; i < 9; i++) x.opsin_inverse_matrix[i] = (i % 4 == 0) ? 1.0f : 0.01f * (float)i;
  for (int i = 0; i < 3; i++) {
    x.bias_cbrt[i] = -0.15f;
    x.scaled_bias[i] = -0.0038f;
  }
  x.intensity_scale = 1.0f;
  return x;
}

int host_checks() {
  const jxlh_frame_params base = VarDctFrame::default_params(1000, 700);
  const Rf rf = rf_of(base);
  // ---- 1. the common VarDCT list: Gaborish x3, EPF1, EPF2, XYB, sRGB, U8 x3, save RGBA
  {
    auto b = add_filters(RenderPipelineBuilder(3, {1000, 700}, 0, 8, base), rf, true, 2);
    b = std::move(b)
            .add_inplace_stage(XybStage{0, some_xyb()})
            .add_inplace_stage(FromLinearStage{0, JXLH_TF_SRGB, 0.0f, {0.f, 0.f, 0.f}})
            .add_inout_stage(ConvertF32ToU8Stage{0, 8})
            .add_inout_stage(ConvertF32ToU8Stage{1, 8})
            .add_inout_stage(ConvertF32ToU8Stage{2, 8})
            .add_save_stage({0, 1, 2}, 0, 4, 8);
    const LoweredPipeline lp = b.lower();
    expect(lp.frame.gab == 1 && lp.frame.epf_iters == 2 && lp.frame.upsampling == 1 && lp.frame.noise == 0, "list 1: stage-derived fields");
    expect(lp.frame.gab_w1[1] == rf.gab_w1[1] && lp.frame.epf_pass2_sigma_scale == rf.pass2, "list 1: weights carried over");
    // jxl/src/render/mod.rs:28-36: Gaborish 1 + EPF1 2 + EPF2 1
    expect(lp.input_border.x == 4 && lp.input_border.y == 4, "list 1: accumulated border is 4");
    expect(lp.has_output && lp.output.color == JXLH_COLOR_XYB && lp.output.transfer == JXLH_TF_SRGB && lp.output.bits == 8 &&
               lp.output.channels == 4,
           "list 1: output descriptor");
    expect(lp.stages.size() == 11 && lp.stages[0] == "Gaborish filter for channel 0", "list 1: Display strings");
  }
  // ---- 2. epf_iters = 3, planar f32 save: border 1 + 3 + 2 + 1
  {
    auto b = add_filters(RenderPipelineBuilder(3, {1000, 700}, 0, 8, base), rf, true, 3);
    const LoweredPipeline lp = std::move(b).add_save_stage({0, 1, 2}, 0, 3, 32).lower();
    expect(lp.frame.epf_iters == 3 && lp.input_border.x == 7 && !lp.has_output, "list 2: three EPF passes");
  }
  // ---- 3. a JPEG recompression: 4:2:0 chroma, no filters, YCbCr, U8, RGB
  {
    auto b = RenderPipelineBuilder(3, {1000, 700}, 0, 8, base)
                 .add_inout_stage(HorizontalChromaUpsample{0})
                 .add_inout_stage(VerticalChromaUpsample{0})
                 .add_inout_stage(HorizontalChromaUpsample{2})
                 .add_inout_stage(VerticalChromaUpsample{2})
                 .add_inplace_stage(YcbcrToRgbStage{0})
                 .add_inout_stage(ConvertF32ToU8Stage{0, 8})
                 .add_inout_stage(ConvertF32ToU8Stage{1, 8})
                 .add_inout_stage(ConvertF32ToU8Stage{2, 8})
                 .add_save_stage({0, 1, 2}, 0, 3, 8);
    const LoweredPipeline lp = b.lower();
    expect(lp.frame.hshift[0] == 1 && lp.frame.vshift[0] == 1 && lp.frame.hshift[1] == 0 && lp.frame.hshift[2] == 1 &&
               lp.frame.gab == 0 && lp.frame.epf_iters == 0,
           "list 3: chroma shifts");
    expect(lp.output.color == JXLH_COLOR_YCBCR && lp.output.channels == 3, "list 3: YCbCr output");
  }
  // ---- 4. 2x frame upsampling + noise: size is size_upsampled, three noise temporaries behind the image channels
  {
    jxlh_frame_params small = VarDctFrame::default_params(500, 350);
    auto b = add_filters(RenderPipelineBuilder(6, {1000, 700}, 1, 8, small), rf_of(small), true, 1);
    std::array<float, 8> lut{0.1f, 0.2f, 0.3f, 0.4f, 0.5f, 0.6f, 0.7f, 0.8f};
    b = std::move(b)
            .add_inout_stage(Upsample2x{nullptr, 0})
            .add_inout_stage(Upsample2x{nullptr, 1})
            .add_inout_stage(Upsample2x{nullptr, 2})
            .add_inout_stage(ConvolveNoiseStage{3})
            .add_inout_stage(ConvolveNoiseStage{4})
            .add_inout_stage(ConvolveNoiseStage{5})
            .add_inplace_stage(AddNoiseStage{lut, 3, -2, 3})
            .add_save_stage({0, 1, 2}, 0, 3, 32);
    const LoweredPipeline lp = b.lower();
    expect(lp.frame.upsampling == 2 && lp.frame.xsize_upsampled == 1000 && lp.frame.ysize_upsampled == 700, "list 4: upsampling");
    expect(lp.frame.noise == 1 && lp.frame.noise_lut[7] == 0.8f && lp.frame.ytox_lf == 3 && lp.frame.ytob_lf == -2, "list 4: noise");
    expect(lp.input_border.x == 3, "list 4: border counts the stages before the upsampling");
  }
  // ---- 5. lists outside the device path
  std::string msg;
  expect(status_of([&] { (void)add_filters(RenderPipelineBuilder(3, {1000, 700}, 0, 8, base), rf, true, 2)
                                   .add_inplace_stage(CpuOnlyStage{"patches"}).add_save_stage({0, 1, 2}, 0, 3, 32).lower(); },
                   &msg) == JXLH_ERR_UNSUPPORTED && msg.find("patches") != std::string::npos,
         "patches stage is rejected by name");
  expect(status_of([&] { (void)RenderPipelineBuilder(3, {1000, 700}, 0, 8, base)
                                   .add_inout_stage(Epf1Stage{1.0f, rf.border_sad_mul, rf.channel_scale})
                                   .add_inout_stage(GaborishStage{0, 0.1f, 0.05f}).add_inout_stage(GaborishStage{1, 0.1f, 0.05f})
                                   .add_inout_stage(GaborishStage{2, 0.1f, 0.05f}).add_save_stage({0, 1, 2}, 0, 3, 32).lower(); }) ==
             JXLH_ERR_UNSUPPORTED,
         "EPF before Gaborish is not the reference's order");
  expect(status_of([&] { (void)RenderPipelineBuilder(3, {1000, 700}, 0, 8, base)
                                   .add_inout_stage(GaborishStage{0, 0.1f, 0.05f}).add_inout_stage(GaborishStage{1, 0.1f, 0.05f})
                                   .add_save_stage({0, 1, 2}, 0, 3, 32).lower(); }) == JXLH_ERR_INVALID_ARGUMENT,
         "Gaborish on two channels");
  expect(status_of([&] { (void)RenderPipelineBuilder(3, {1000, 700}, 0, 8, base)
                                   .add_inout_stage(Epf2Stage{rf.pass2, rf.border_sad_mul, rf.channel_scale})
                                   .add_save_stage({0, 1, 2}, 0, 3, 32).lower(); }) == JXLH_ERR_UNSUPPORTED,
         "EPF2 without EPF1");
  expect(status_of([&] { (void)RenderPipelineBuilder(3, {1000, 700}, 0, 8, base)
                                   .add_inout_stage(Epf1Stage{0.5f, rf.border_sad_mul, rf.channel_scale})
                                   .add_save_stage({0, 1, 2}, 0, 3, 32).lower(); }) == JXLH_ERR_UNSUPPORTED,
         "EPF1 sigma scale is 1 in the reference's list");
  expect(status_of([&] { (void)RenderPipelineBuilder(3, {1000, 700}, 1, 8, base)
                                   .add_inout_stage(Upsample4x{nullptr, 0}).add_inout_stage(Upsample4x{nullptr, 1})
                                   .add_inout_stage(Upsample4x{nullptr, 2}).add_save_stage({0, 1, 2}, 0, 3, 32).lower(); }) ==
             JXLH_ERR_INVALID_ARGUMENT,
         "4x stages with a downsampling shift of 1");
  expect(status_of([&] { (void)RenderPipelineBuilder(4, {1000, 700}, 0, 8, base)
                                   .add_inout_stage(Upsample2x{nullptr, 3}).add_save_stage({0, 1, 2}, 0, 3, 32).lower(); }) ==
             JXLH_ERR_INVALID_ARGUMENT,
         "extra-channel upsampling without the channel's Modular -> f32 conversion");
  // ---- 5b. extra channels (frame/render.rs:564-567, :624-637, :655-668)
  {
    // early: each channel's own factor, behind the filters and before the frame's upsampling
    auto b = RenderPipelineBuilder(5, {1000, 700}, 0, 8, base)
                 .add_inout_stage(ConvertModularToF32Stage{3, 8})
                 .add_inout_stage(ConvertModularToF32Stage{4, 16});
    const LoweredPipeline lp = add_filters(std::move(b), rf, true, 2)
                                   .add_inout_stage(Upsample4x{nullptr, 4})
                                   .add_save_stage({0, 1, 2}, 0, 3, 32)
                                   .lower();
    expect(lp.modular == LoweredPipeline::Modular::kNone && lp.extra[0].bits == 8 && lp.extra[0].upsampling == 1 &&
               lp.extra[1].bits == 16 && lp.extra[1].upsampling == 4 && lp.extra[2].bits == 0 && lp.frame.upsampling == 1,
           "extra channels: bit depths and the channel's own factor");
    expect(lp.input_border.x == 4, "extra channels: their upsampling does not touch the colour channels' border");
    // late: every ec_upsampling equals the frame's, so the extra channels follow channel 2 (frame/render.rs:655-668)
    static const float w2[15] = {0};
    jxlh_frame_params half = VarDctFrame::default_params(500, 350);
    const LoweredPipeline late = RenderPipelineBuilder(4, {1000, 700}, 1, 8, half)
                                     .add_inout_stage(ConvertModularToF32Stage{3, 10})
                                     .add_inout_stage(Upsample2x{w2, 0}).add_inout_stage(Upsample2x{w2, 1})
                                     .add_inout_stage(Upsample2x{w2, 2}).add_inout_stage(Upsample2x{w2, 3})
                                     .add_save_stage({0, 1, 2}, 0, 3, 32).lower();
    expect(late.frame.upsampling == 2 && late.extra[0].upsampling == 2 && late.extra[0].bits == 10 && late.weights_by_factor[0] == w2,
           "extra channels upsampled together with the colour channels");
    expect(status_of([&] { (void)RenderPipelineBuilder(4, {1000, 700}, 1, 8, half)
                                     .add_inout_stage(ConvertModularToF32Stage{3, 10})
                                     .add_inout_stage(Upsample2x{w2, 0}).add_inout_stage(Upsample2x{w2, 3})
                                     .add_save_stage({0, 1, 2}, 0, 3, 32).lower(); }) == JXLH_ERR_INVALID_ARGUMENT,
           "an extra channel between the colour channels' upsampling stages");
    expect(status_of([&] { (void)RenderPipelineBuilder(5, {1000, 700}, 0, 8, base)
                                     .add_inout_stage(ConvertModularToF32Stage{4, 8})
                                     .add_save_stage({0, 1, 2}, 0, 3, 32).lower(); }) == JXLH_ERR_INVALID_ARGUMENT,
           "extra channels out of order");
    expect(status_of([&] { (void)RenderPipelineBuilder(4, {1000, 700}, 0, 8, base)
                                     .add_inout_stage(ConvertModularToF32Stage{3, 8}).add_inout_stage(Upsample2x{nullptr, 3})
                                     .add_inout_stage(Upsample2x{nullptr, 3}).add_save_stage({0, 1, 2}, 0, 3, 32).lower(); }) ==
               JXLH_ERR_INVALID_ARGUMENT,
           "an extra channel upsampled twice");
    expect(status_of([&] { (void)RenderPipelineBuilder(3 + JXLH_MAX_EXTRA_CHANNELS + 1, {1000, 700}, 0, 8, base)
                                     .add_inout_stage(ConvertModularToF32Stage{3 + JXLH_MAX_EXTRA_CHANNELS, 8})
                                     .add_save_stage({0, 1, 2}, 0, 3, 32).lower(); }) == JXLH_ERR_UNSUPPORTED,
           "more extra channels than the device path holds");
  }
  expect(status_of([&] { (void)add_filters(RenderPipelineBuilder(3, {1000, 700}, 0, 8, base), rf, true, 2).lower(); }) ==
             JXLH_ERR_INVALID_ARGUMENT,
         "a pipeline without save stage");
  expect(status_of([&] { (void)RenderPipelineBuilder(3, {1000, 700}, 0, 8, base)
                                   .add_inplace_stage(FromLinearStage{0, JXLH_TF_PQ, 10000.f, {0.f, 0.f, 0.f}})
                                   .add_save_stage({0, 1, 2}, 0, 3, 32).lower(); }) == JXLH_ERR_UNSUPPORTED,
         "transfer function without XybStage");
  expect(status_of([&] { (void)RenderPipelineBuilder(3, {1000, 700}, 0, 8, base)
                                   .add_inplace_stage(XybStage{0, some_xyb()})
                                   .add_inout_stage(ConvertF32ToU16Stage{0, 16}).add_inout_stage(ConvertF32ToU16Stage{1, 16})
                                   .add_inout_stage(ConvertF32ToU16Stage{2, 16}).add_save_stage({0, 1, 2}, 0, 3, 8).lower(); }) ==
             JXLH_ERR_INVALID_ARGUMENT,
         "save format and conversion disagree");
  expect(status_of([&] { (void)RenderPipelineBuilder(3, {999, 700}, 0, 8, base).add_save_stage({0, 1, 2}, 0, 3, 32).lower(); }) ==
             JXLH_ERR_INVALID_ARGUMENT,
         "pipeline size and frame size disagree");
  // ---- 6. Modular frames: the builder's special case (builder.rs:152-170) and the f32 route
  {
    auto list = [&](uint8_t depth, uint8_t out_bits) {
      auto b = RenderPipelineBuilder(3, {1000, 700}, 0, 8, base)
                   .add_inout_stage(ConvertModularToF32Stage{0, depth})
                   .add_inout_stage(ConvertModularToF32Stage{1, depth})
                   .add_inout_stage(ConvertModularToF32Stage{2, depth});
      if (out_bits == 8)
        return std::move(b).add_inout_stage(ConvertF32ToU8Stage{0, 8}).add_inout_stage(ConvertF32ToU8Stage{1, 8})
            .add_inout_stage(ConvertF32ToU8Stage{2, 8}).add_save_stage({0, 1, 2}, 0, 3, 8).lower();
      return std::move(b).add_save_stage({0, 1, 2}, 0, 3, 32).lower();
    };
    const LoweredPipeline a = list(8, 8), b4 = list(4, 8), f = list(12, 32);
    expect(a.modular == LoweredPipeline::Modular::kI32ToU8 && a.i32_to_u8_multiplier == 1 && a.i32_to_u8_max == 255, "8-bit Modular -> U8: multiplier 1");
    expect(b4.modular == LoweredPipeline::Modular::kI32ToU8 && b4.i32_to_u8_multiplier == 17, "4-bit Modular -> U8: multiplier 255 / 15");
    expect(f.modular == LoweredPipeline::Modular::kToF32 && f.modular_bits == 12 && !f.has_output, "12-bit Modular -> f32 planes");
    expect(status_of([&] { (void)list(5, 8); }) == JXLH_ERR_UNSUPPORTED, "8 % 5 != 0: no special case, integer output stays on the CPU pipeline");
    jxlh_frame_params pm = base;
    pm.epf_sigma_for_modular = 1.5f;
    auto withf = add_filters(RenderPipelineBuilder(3, {1000, 700}, 0, 8, pm)
                                 .add_inout_stage(ConvertModularXYBToF32Stage{0, {0.01f, 0.02f, 0.03f}}),
                             rf, true, 1);
    const LoweredPipeline x = std::move(withf).add_save_stage({0, 1, 2}, 0, 3, 32).lower();
    expect(x.modular == LoweredPipeline::Modular::kXybToF32 && x.frame.gab == 1 && x.frame.epf_iters == 1 &&
               x.modular_quant_factors[2] == 0.03f && x.input_border.x == 3,
           "lossy Modular (XYB) with Gaborish + EPF1");
    expect(status_of([&] { (void)RenderPipelineBuilder(3, {1000, 700}, 0, 8, base)
                                     .add_inout_stage(ConvertModularToF32Stage{0, 8}).add_inout_stage(ConvertModularToF32Stage{1, 8})
                                     .add_save_stage({0, 1, 2}, 0, 3, 32).lower(); }) == JXLH_ERR_INVALID_ARGUMENT,
           "Modular conversion on two channels");
    expect(status_of([&] { (void)RenderPipelineBuilder(4, {1000, 700}, 0, 8, base)
                                     .add_inout_stage(ConvertModularToF32Stage{3, 8}).add_save_stage({0, 1, 2}, 0, 3, 32).lower(); }) ==
               JXLH_OK,
           "a VarDCT frame with one extra channel (its conversion is the only Modular stage of the list)");
    expect(status_of([&] { (void)RenderPipelineBuilder(3, {1000, 700}, 0, 8, base)
                                     .add_inout_stage(ConvertModularToF32Stage{0, 8}).add_inout_stage(ConvertModularToF32Stage{1, 8})
                                     .add_inout_stage(ConvertModularToF32Stage{2, 8}).add_inplace_stage(XybStage{0, some_xyb()})
                                     .add_save_stage({0, 1, 2}, 0, 3, 32).lower(); }) == JXLH_ERR_UNSUPPORTED,
           "colour stages behind a Modular conversion");
  }
  // BORDER / SHIFT constants are the reference's
  static_assert(GaborishStage::BORDER.x == 1 && Epf0Stage::BORDER.x == 3 && Epf1Stage::BORDER.y == 2 && Epf2Stage::BORDER.x == 1);
  static_assert(Upsample8x::SHIFT.x == 3 && Upsample2x::BORDER.x == 2 && HorizontalChromaUpsample::SHIFT.x == 1 &&
                HorizontalChromaUpsample::SHIFT.y == 0 && VerticalChromaUpsample::BORDER.y == 1 && ConvolveNoiseStage::BORDER.x == 2);
  printf("host checks: %s\n", g_failed ? "FAILED" : "ok");
  return g_failed ? 1 : 0;
}

int gpu_frame(int w, int h, int epf_iters) {
  synth::Frame F;
  if (!synth::make(w, h, epf_iters, &F)) return 2;
  try {
    Context ctx(0, 2);
    jxlh_frame_params base = VarDctFrame::default_params((uint32_t)w, (uint32_t)h);
    // two extra channels as frame/render.rs:564-567, :624-637 adds them: 8-bit alpha coded at half resolution
    // (ec_upsampling 2) and a 16-bit channel at full resolution
    auto b0 = RenderPipelineBuilder(5, {(size_t)w, (size_t)h}, 0, 8, base)
                  .add_inout_stage(ConvertModularToF32Stage{3, 8})
                  .add_inout_stage(ConvertModularToF32Stage{4, 16});
    auto b = add_filters(std::move(b0), rf_of(base), true, epf_iters).add_inout_stage(Upsample2x{nullptr, 3});
    auto pipe = std::move(b).add_save_stage({0, 1, 2}, 0, 3, 32).build(ctx);
    expect((int)pipe->lowered().frame.epf_iters == epf_iters, "built pipeline carries the stage list");
    const int aw = (w + 1) / 2, ah = (h + 1) / 2;
    std::vector<int32_t> alpha((size_t)aw * ah), depth((size_t)w * h);
    uint32_t lcg = 12345u;
    for (auto& v : alpha) v = (int32_t)((lcg = lcg * 1664525u + 1013904223u) >> 24);
    for (auto& v : depth) v = (int32_t)((lcg = lcg * 1664525u + 1013904223u) >> 16);
    pipe->set_extra_channel_buffer(0, alpha.data(), (size_t)aw, (uint32_t)aw, (uint32_t)ah);
    pipe->set_extra_channel_buffer(1, depth.data(), (size_t)w, (uint32_t)w, (uint32_t)h);
    VarDctFrame& frame = pipe->frame();
    frame.decode_hf_global(F.tables);
    frame.decode_lf_group(0, 0, (uint32_t)F.xb, (uint32_t)F.yb, F.qy.data(), F.qx.data(), F.qb.data(), (size_t)F.xb);
    frame.decode_hf_metadata(0, 0, (uint32_t)F.xb, (uint32_t)F.yb, F.tmap.data(), F.rq.data(), F.epf.data(), (size_t)F.xb,
                             F.ytox.data(), F.ytob.data(), (size_t)F.cw);
    // pass 1: every group handed over, the odd ones with zeroed coefficients and `complete = false`
    std::vector<int32_t> zeros((size_t)3 * 65536, 0);
    for (int g = 0; g < F.ngroups; g++)
      pipe->set_buffer_for_group((uint32_t)g, g % 2 == 0, g % 2 ? zeros.data() : &F.coeffs[(size_t)g * 3 * 65536], g % 2);
    pipe->do_render();
    // pass 2: the odd groups arrive complete; every other one is ALSO marked for re-rendering (render/mod.rs:146), the
    // rest rely on set_buffer_for_group alone, which the reference renders just the same (render/mod.rs:128-136)
    for (int g = 1; g < F.ngroups; g += 2) {
      pipe->set_buffer_for_group((uint32_t)g, true, &F.coeffs[(size_t)g * 3 * 65536], g % 2);
      if (g % 4 == 1) pipe->mark_group_to_rerender((uint32_t)g);
    }
    pipe->do_render();
    ctx.sync();
    pipe->check_buffer_sizes((size_t)w * sizeof(float), (size_t)h);
    std::vector<float> out[3];
    for (auto& o : out) o.resize((size_t)w * h);
    pipe->save_planes(out[0].data(), out[1].data(), out[2].data());
    size_t bad = 0;
    for (int c = 0; c < 3; c++)
      for (int y = 0; y < h; y++)
        if (memcmp(&out[c][(size_t)y * w], &F.pl[c][(size_t)y * F.stride], sizeof(float) * w) != 0) bad++;
    // the extra channels against the oracle's ConvertModularToF32 / Upsample2x
    size_t bad_ec = 0;
    {
      std::vector<float> af((size_t)aw * ah), up((size_t)aw * 2 * ah * 2), df((size_t)w * h), got((size_t)w * h);
      jxlo_modular_to_f32(alpha.data(), alpha.size(), 8, af.data());
      jxlo_upsample(2, nullptr, af.data(), aw, ah, (size_t)aw, up.data(), (size_t)aw * 2);
      pipe->save_extra_channel(0, got.data(), (size_t)w);
      for (int y = 0; y < h; y++)
        if (memcmp(&got[(size_t)y * w], &up[(size_t)y * aw * 2], sizeof(float) * w) != 0) bad_ec++;
      jxlo_modular_to_f32(depth.data(), depth.size(), 16, df.data());
      pipe->save_extra_channel(1, got.data(), (size_t)w);
      if (memcmp(got.data(), df.data(), sizeof(float) * df.size()) != 0) bad_ec++;
    }
    bool threw = false;
    try {
      pipe->check_buffer_sizes((size_t)w * sizeof(float) - 1, (size_t)h);
    } catch (const Error& e) {
      threw = e.status == JXLH_ERR_INVALID_ARGUMENT;
    }
    // the same frame with the colour tail of frame/render.rs:755-790 in the stage list: XybStage, FromLinearStage (sRGB),
    // three ConvertF32ToU8Stage, save as RGBA -- against the oracle's output stage on the oracle's planes
    size_t bad8 = 0;
    {
      JxloXybParams ox;
      const float mat[9] = {11.031566901960783f, -9.866943921568629f, -0.16462299647058826f, -3.254147380392157f,
                            4.418770392156863f, -0.16462299647058826f, -3.6588512862745097f, 2.7129230470588235f,
                            1.9459282392156863f};
      const float bias[3] = {-0.0037930732552754493f, -0.0037930732552754493f, -0.0037930732552754493f};
      jxlo_xyb_params(mat, bias, 255.0f, &ox);
      jxlh_xyb_params gx;
      memcpy(gx.opsin_inverse_matrix, ox.mat, sizeof(ox.mat));
      memcpy(gx.bias_cbrt, ox.bias_cbrt, sizeof(ox.bias_cbrt));
      memcpy(gx.scaled_bias, ox.scaled_bias, sizeof(ox.scaled_bias));
      gx.intensity_scale = ox.intensity_scale;
      Context ctx2(0, 1);
      auto b2 = add_filters(RenderPipelineBuilder(3, {(size_t)w, (size_t)h}, 0, 8, base), rf_of(base), true, epf_iters);
      auto pipe2 = std::move(b2)
                       .add_inplace_stage(XybStage{0, gx})
                       .add_inplace_stage(FromLinearStage{0, JXLH_TF_SRGB, 0.0f, {0.f, 0.f, 0.f}})
                       .add_inout_stage(ConvertF32ToU8Stage{0, 8})
                       .add_inout_stage(ConvertF32ToU8Stage{1, 8})
                       .add_inout_stage(ConvertF32ToU8Stage{2, 8})
                       .add_save_stage({0, 1, 2}, 0, 4, 8)
                       .build(ctx2);
      VarDctFrame& f2 = pipe2->frame();
      f2.decode_hf_global(F.tables);
      f2.decode_lf_group(0, 0, (uint32_t)F.xb, (uint32_t)F.yb, F.qy.data(), F.qx.data(), F.qb.data(), (size_t)F.xb);
      f2.decode_hf_metadata(0, 0, (uint32_t)F.xb, (uint32_t)F.yb, F.tmap.data(), F.rq.data(), F.epf.data(), (size_t)F.xb,
                            F.ytox.data(), F.ytob.data(), (size_t)F.cw);
      for (int g = 0; g < F.ngroups; g++) pipe2->set_buffer_for_group((uint32_t)g, true, &F.coeffs[(size_t)g * 3 * 65536]);
      pipe2->do_render();
      std::vector<uint8_t> got((size_t)w * h * 4), want((size_t)w * h * 4);
      pipe2->check_buffer_sizes((size_t)w * 4, (size_t)h);
      pipe2->save(got.data());
      jxlo_xyb_to_rgb8(&ox, F.pl[0].data(), F.pl[1].data(), F.pl[2].data(), (size_t)w, (size_t)h, F.stride, want.data(),
                       (size_t)w * 4, 4);
      for (int y = 0; y < h; y++)
        if (memcmp(&got[(size_t)y * w * 4], &want[(size_t)y * w * 4], (size_t)w * 4) != 0) bad8++;
    }
    printf("%dx%d epf_iters=%d groups=%d through RenderPipelineBuilder, two passes: %zu differing rows, extra channels: %zu differing "
           "rows, RGBA8 tail: %zu differing rows, error path %s\n",
           w, h, epf_iters, F.ngroups, bad, bad_ec, bad8, threw ? "ok" : "MISSING");
    return (bad == 0 && bad_ec == 0 && bad8 == 0 && threw && !g_failed) ? 0 : 1;
  } catch (const std::exception& e) {
    fprintf(stderr, "device path failed: %s\n", e.what());
    return 3;
  }
}
// Modular lists on the device: the I32 -> U8 special case against jxlo_i32_to_u8, the f32 conversion against
// jxlo_modular_to_f32, and the f32 route with Gaborish + EPF1 (constant sigma) run end to end on device planes
int gpu_modular(int w, int h) {
  try {
    Context ctx(0, 1);
    const jxlh_frame_params base = VarDctFrame::default_params((uint32_t)w, (uint32_t)h);
    synth::Rng rng{77u + (uint64_t)w * 131 + (uint64_t)h};
    const size_t n = (size_t)w * h;
    std::vector<int32_t> pl[3];
    for (auto& p : pl) {
      p.resize(n);
      for (auto& v : p) v = rng.range(-3, 40);  // 5-bit samples with a few out-of-range ones (the stage clamps)
    }
    const int32_t* in[3] = {pl[0].data(), pl[1].data(), pl[2].data()};
    // (a) 4-bit... use 5 bits -> no special case; 4-bit: multiplier 17
    auto pa = RenderPipelineBuilder(3, {(size_t)w, (size_t)h}, 0, 8, base)
                  .add_inout_stage(ConvertModularToF32Stage{0, 4}).add_inout_stage(ConvertModularToF32Stage{1, 4})
                  .add_inout_stage(ConvertModularToF32Stage{2, 4}).add_inout_stage(ConvertF32ToU8Stage{0, 8})
                  .add_inout_stage(ConvertF32ToU8Stage{1, 8}).add_inout_stage(ConvertF32ToU8Stage{2, 8})
                  .add_save_stage({0, 1, 2}, 0, 3, 8).build_modular(ctx);
    std::vector<uint8_t> got(n * 3), want(n * 3), ch(n);
    pa->render_u8(in, (size_t)w, got.data(), (size_t)w * 3);
    for (int c = 0; c < 3; c++) {
      jxlo_i32_to_u8(pl[c].data(), n, 17, 255, ch.data());
      for (size_t i = 0; i < n; i++) want[i * 3 + c] = ch[i];
    }
    const bool u8_ok = got == want;
    // (b) 12-bit samples to f32 planes
    auto pb = RenderPipelineBuilder(3, {(size_t)w, (size_t)h}, 0, 8, base)
                  .add_inout_stage(ConvertModularToF32Stage{0, 12}).add_inout_stage(ConvertModularToF32Stage{1, 12})
                  .add_inout_stage(ConvertModularToF32Stage{2, 12}).add_save_stage({0, 1, 2}, 0, 3, 32).build_modular(ctx);
    std::vector<float> f[3], fw(n);
    float* fo[3];
    for (int c = 0; c < 3; c++) {
      f[c].resize(n);
      fo[c] = f[c].data();
    }
    pb->render_f32(in, fo, fo);
    ctx.sync();
    bool f32_ok = true;
    for (int c = 0; c < 3; c++) {
      jxlo_modular_to_f32(pl[c].data(), n, 12, fw.data());
      f32_ok = f32_ok && memcmp(fw.data(), f[c].data(), n * sizeof(float)) == 0;
    }
    // (c) the f32 route with Gaborish + EPF1 on device planes: equal to the two ABI calls made by hand
    bool filt_ok = true;
    if (w % 4 == 0) {
      jxlh_frame_params pm = base;
      pm.epf_sigma_for_modular = 1.25f;
      auto pc = add_filters(RenderPipelineBuilder(3, {(size_t)w, (size_t)h}, 0, 8, pm)
                                .add_inout_stage(ConvertModularToF32Stage{0, 8}).add_inout_stage(ConvertModularToF32Stage{1, 8})
                                .add_inout_stage(ConvertModularToF32Stage{2, 8}),
                            rf_of(pm), true, 1);
      auto pipe = std::move(pc).add_save_stage({0, 1, 2}, 0, 3, 32).build_modular(ctx);
      float *dt[3], *dout[3], *dref_in[3], *dref_out[3];
      int32_t* din[3];
      for (int c = 0; c < 3; c++) {
        if (hipMalloc((void**)&dt[c], n * 4) || hipMalloc((void**)&dout[c], n * 4) || hipMalloc((void**)&dref_in[c], n * 4) ||
            hipMalloc((void**)&dref_out[c], n * 4) || hipMalloc((void**)&din[c], n * 4))
          return fprintf(stderr, "hipMalloc failed\n"), 3;
        if (hipMemcpy(din[c], pl[c].data(), n * 4, hipMemcpyHostToDevice)) return 3;
      }
      const int32_t* cin[3] = {din[0], din[1], din[2]};
      pipe->render_f32(cin, dt, dout);
      for (int c = 0; c < 3; c++) ctx.check(jxlh_modular_to_f32(ctx.raw(), din[c], n, 8, dref_in[c]), "to_f32");
      jxlh_frame_params pl2 = pipe->lowered().frame;
      ctx.check(jxlh_modular_frame_filters(ctx.raw(), &pl2, dref_in, dref_out, (uint32_t)w, (uint32_t)h, (size_t)w), "filters");
      ctx.sync();
      std::vector<float> a(n), b(n);
      size_t changed = 0;
      for (int c = 0; c < 3; c++) {
        if (hipMemcpy(a.data(), dout[c], n * 4, hipMemcpyDeviceToHost) || hipMemcpy(b.data(), dref_out[c], n * 4, hipMemcpyDeviceToHost))
          return 3;
        filt_ok = filt_ok && memcmp(a.data(), b.data(), n * 4) == 0;
        jxlo_modular_to_f32(pl[c].data(), n, 8, fw.data());
        for (size_t i = 0; i < n; i++) changed += a[i] != fw[i];
        (void)hipFree(dt[c]); (void)hipFree(dout[c]); (void)hipFree(dref_in[c]); (void)hipFree(dref_out[c]); (void)hipFree(din[c]);
      }
      filt_ok = filt_ok && changed > n / 2;  // the filters really ran
    }
    printf("%dx%d Modular lists: I32->U8 %s, to f32 %s, f32 + filters %s\n", w, h, u8_ok ? "ok" : "DIFFERS", f32_ok ? "ok" : "DIFFERS",
           filt_ok ? "ok" : "DIFFERS");
    return (u8_ok && f32_ok && filt_ok) ? 0 : 1;
  } catch (const std::exception& e) {
    fprintf(stderr, "device path failed: %s\n", e.what());
    return 3;
  }
}
}  // namespace

int main(int argc, char** argv) {
  if (argc < 2 || !strcmp(argv[1], "host")) return host_checks();
  if (!strcmp(argv[1], "gpu") && argc >= 5) return gpu_frame(atoi(argv[2]), atoi(argv[3]), atoi(argv[4]));
  if (!strcmp(argv[1], "modular") && argc >= 4) return gpu_modular(atoi(argv[2]), atoi(argv[3]));
  fprintf(stderr, "usage: pipeline_builder host | gpu W H EPF_ITERS\n");
  return 2;
}
