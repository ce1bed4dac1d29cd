// C++ mirror of the reference's stage traits and pipeline builder for the path this library replaces
// (jxl/src/render/mod.rs:52-157 RenderPipelineInOutStage / RenderPipelineInPlaceStage / RenderPipeline,
// jxl/src/render/builder.rs:19-120 RenderPipelineBuilder, and the stage list of
// Frame::build_render_pipeline, jxl/src/frame/render.rs:526-790).
//
// The reference assembles a frame's post-processing as a list of stage objects and lets a RenderPipeline
// implementation run them row chunk by row chunk.  On the device the same list is ONE launch sequence behind
// jxlh_frame_run (K1, the fused Gaborish / EPF kernel, chroma / frame upsampling, noise) plus one output pass
// (jxlh_frame_read_output): so the builder here takes the stages under the reference's names, with the reference's
// constructor arguments, in the reference's order, checks that the list is one the device path implements (anything
// else is JXLH_ERR_UNSUPPORTED naming the stage: that stage list stays on the CPU pipeline), and LOWERS it onto
// jxlh_frame_params + jxlh_output_desc.  GpuRenderPipeline then carries the trait's methods for this path:
// set_buffer_for_group (with the `complete` flag of render/mod.rs:128-136), mark_group_to_rerender, do_render,
// check_buffer_sizes.  BORDER / SHIFT of every stage are the reference's constants, and the pipeline reports the
// accumulated input border the way RenderPipelineShared does (render/internal.rs) -- 4 pixels for Gaborish + EPF1 + EPF2,
// jxl/src/render/mod.rs:28-36 -- which is the halo the fused kernel stages and the sharded path exchanges.
//
// Header-only, C++17, no HIP types; tests/cpp/pipeline_builder.cc drives it (host-only checks without a GPU, a whole
// frame against the oracle with one).
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <memory>
#include <string>
#include <utility>
#include <variant>
#include <vector>

#include "jxl_hip.hpp"

namespace jxlh {

// ---------------------------------------------------------------------------------------------------------------
// Stage descriptors.  Names, constructor argument order and the BORDER / SHIFT constants follow
// jxl/src/render/stages/*.rs; `Display` strings are the reference's (they end up in error messages).
struct Border {
  uint8_t x, y;
};

struct HorizontalChromaUpsample {  // chroma_upsample.rs:15, :67-68
  int channel;
  static constexpr Border BORDER{1, 0}, SHIFT{1, 0};
  std::string display() const { return "chroma upsample of channel " + std::to_string(channel) + ", horizontally"; }
  bool uses_channel(int c) const { return c == channel; }
};
struct VerticalChromaUpsample {  // chroma_upsample.rs:93, :154-155
  int channel;
  static constexpr Border BORDER{0, 1}, SHIFT{0, 1};
  std::string display() const { return "chroma upsample of channel " + std::to_string(channel) + ", vertically"; }
  bool uses_channel(int c) const { return c == channel; }
};
struct GaborishStage {  // gaborish.rs:20, :94-95
  int channel;
  float weight1, weight2;
  static constexpr Border BORDER{1, 1}, SHIFT{0, 0};
  std::string display() const { return "Gaborish filter for channel " + std::to_string(channel); }
  bool uses_channel(int c) const { return c == channel; }
};
template <int PASS>
struct EpfStage {  // epf/epf{0,1,2}.rs:35-48; the SigmaSource argument is the frame's own (jxlh_frame_run builds it)
  float sigma_scale, border_sad_mul;
  std::array<float, 3> channel_scale;
  static constexpr Border BORDER{PASS == 0 ? 3 : PASS == 1 ? 2 : 1, PASS == 0 ? 3 : PASS == 1 ? 2 : 1}, SHIFT{0, 0};
  std::string display() const {
    return "EPF stage " + std::to_string(PASS) + " with sigma scale: " + std::to_string(sigma_scale) +
           ", border_sad_mul: " + std::to_string(border_sad_mul);
  }
  bool uses_channel(int c) const { return c < 3; }
};
using Epf0Stage = EpfStage<0>;
using Epf1Stage = EpfStage<1>;
using Epf2Stage = EpfStage<2>;
template <int N>
struct Upsample {  // upsample.rs:22, :396-397.  weights: CustomTransformData::weights{2,4,8} or nullptr = defaults
  const float* weights;
  int channel;
  static constexpr Border BORDER{2, 2}, SHIFT{N == 2 ? 1 : N == 4 ? 2 : 3, N == 2 ? 1 : N == 4 ? 2 : 3};
  std::string display() const {
    return std::to_string(N) + "x" + std::to_string(N) + " upsampling of channel " + std::to_string(channel);
  }
  bool uses_channel(int c) const { return c == channel; }
};
using Upsample2x = Upsample<2>;
using Upsample4x = Upsample<4>;
using Upsample8x = Upsample<8>;
struct ConvolveNoiseStage {  // noise.rs:22, :87-88
  int channel;
  static constexpr Border BORDER{2, 2}, SHIFT{0, 0};
  std::string display() const { return "convolve noise for channel " + std::to_string(channel); }
  bool uses_channel(int c) const { return c == channel; }
};
struct AddNoiseStage {  // noise.rs:115-128 (in place).  lut = Noise::lut, ytox_lf / ytob_lf = ColorCorrelationParams
  std::array<float, 8> lut;
  int32_t ytox_lf, ytob_lf;
  int first_channel;
  std::string display() const {
    return "add noise for channels [" + std::to_string(first_channel) + "," + std::to_string(first_channel + 1) + "," +
           std::to_string(first_channel + 2) + "]";
  }
  bool uses_channel(int c) const { return c < 3 || (c >= first_channel && c < first_channel + 3); }
};
struct XybStage {  // xyb.rs:173 (in place); params = XybParams::new(opsin, intensity_target), xyb.rs:147-163
  int first_channel;
  jxlh_xyb_params params;
  std::string display() const { return "XYB to linear for channel [0,1,2]"; }
  bool uses_channel(int c) const { return c >= first_channel && c < first_channel + 3; }
};
struct YcbcrToRgbStage {  // ycbcr.rs:16 (in place)
  int first_channel;
  std::string display() const { return "YCbCr to RGB for channel [0,1,2]"; }
  bool uses_channel(int c) const { return c >= first_channel && c < first_channel + 3; }
};
struct FromLinearStage {  // from_linear.rs:20 (in place); transfer = JXLH_TF_*, param as jxlh_output_desc::tf_param
  int first_channel;
  uint32_t transfer;
  float param;
  std::array<float, 3> hlg_luminance_rgb;
  std::string display() const { return "Apply transfer function " + std::to_string(transfer) + " to channel [0,1,2]"; }
  bool uses_channel(int c) const { return c >= first_channel && c < first_channel + 3; }
};
struct ConvertF32ToU8Stage {  // convert.rs:555, :612-613
  int channel;
  uint8_t bit_depth;
  static constexpr Border BORDER{0, 0}, SHIFT{0, 0};
  std::string display() const {
    return "convert F32 to U8 in channel " + std::to_string(channel) + " with bit depth " + std::to_string(bit_depth);
  }
  bool uses_channel(int c) const { return c == channel; }
};
struct ConvertF32ToU16Stage {  // convert.rs:724, :767-768
  int channel;
  uint8_t bit_depth;
  static constexpr Border BORDER{0, 0}, SHIFT{0, 0};
  std::string display() const {
    return "convert F32 to U16 in channel " + std::to_string(channel) + " with bit depth " + std::to_string(bit_depth);
  }
  bool uses_channel(int c) const { return c == channel; }
};
struct ConvertModularToF32Stage {  // convert.rs:351, :512-513 (bit_depth: integer samples of that many bits)
  int channel;
  uint8_t bit_depth;
  static constexpr Border BORDER{0, 0}, SHIFT{0, 0};
  std::string display() const {
    return "convert modular data to F32 in channel " + std::to_string(channel) + " with bit depth " + std::to_string(bit_depth);
  }
  bool uses_channel(int c) const { return c == channel; }
};
struct ConvertModularXYBToF32Stage {  // convert.rs:284, :309-310; quant_factors = LfQuantFactors::quant_factors (X, Y, B)
  int first_channel;
  std::array<float, 3> quant_factors;
  static constexpr Border BORDER{0, 0}, SHIFT{0, 0};
  std::string display() const { return "convert modular xyb data to F32 in channels 0..3"; }
  bool uses_channel(int c) const { return c >= first_channel && c < first_channel + 3; }
};
// A stage of the reference this path does not run on the device (patches, splines, blending, extend, spot colour,
// premultiplied alpha, extra-channel conversions ...): adding one makes build() fail with JXLH_ERR_UNSUPPORTED.
struct CpuOnlyStage {
  std::string name;
  std::string display() const { return name; }
  bool uses_channel(int) const { return true; }
};
// add_save_stage (builder.rs:87-105): channels, output buffer index, interleaved colour type and sample format
struct SaveStage {
  std::vector<int> channels;
  int output_buffer_index;
  uint32_t color_channels;  // 3 = RGB, 4 = RGBA (fill_opaque_alpha)
  uint32_t bits;            // 8, 16, or 32 = the f32 planes themselves (JxlDataFormat::F32)
  std::string display() const { return "save stage for buffer " + std::to_string(output_buffer_index); }
  bool uses_channel(int c) const {
    for (int ch : channels)
      if (ch == c) return true;
    return false;
  }
};

using Stage = std::variant<ConvertModularToF32Stage, ConvertModularXYBToF32Stage, HorizontalChromaUpsample, VerticalChromaUpsample, GaborishStage, Epf0Stage, Epf1Stage, Epf2Stage,
                           Upsample2x, Upsample4x, Upsample8x, ConvolveNoiseStage, AddNoiseStage, XybStage, YcbcrToRgbStage,
                           FromLinearStage, ConvertF32ToU8Stage, ConvertF32ToU16Stage, CpuOnlyStage, SaveStage>;

inline std::string stage_display(const Stage& s) {
  return std::visit([](const auto& st) { return st.display(); }, s);
}

// ---------------------------------------------------------------------------------------------------------------
// What a stage list lowers to.
struct LoweredPipeline {
  jxlh_frame_params frame;     // the caller's base parameters with the stage-derived fields overwritten
  bool has_output = false;     // a colour / conversion tail was given: read through jxlh_frame_read_output
  jxlh_output_desc output{};
  const float* upsampling_weights = nullptr;
  const float* weights_by_factor[3] = {nullptr, nullptr, nullptr};  // 2x, 4x, 8x: the frame's and the extra channels'
  // Modular frames (Encoding::Modular: the list opens with the Modular -> f32 conversions, frame/render.rs:553-567)
  enum class Modular { kNone, kToF32, kXybToF32, kI32ToU8 } modular = Modular::kNone;
  uint32_t modular_bits = 0;                       // kToF32: bits per integer sample
  std::array<float, 3> modular_quant_factors{};    // kXybToF32
  int32_t i32_to_u8_multiplier = 0, i32_to_u8_max = 0;  // kI32ToU8: ConvertI32ToU8Stage::new(c, mult, max), builder.rs:152-170
  Border input_border{0, 0};   // accumulated BORDER of the in-out stages before any upsampling, in input pixels
  // Extra channels (pipeline channels 3..): ConvertModularToF32Stage::new(3 + ec, ec_bit_depth) (frame/render.rs:564-567)
  // and the channel's own Upsample{2,4,8}x::new(transform_data, 3 + ec) (frame/render.rs:624-637, or with the colour
  // channels when every ec_upsampling equals the frame's, :655-668).  bits == 0: the list does not name the channel.
  struct Extra {
    uint32_t bits = 0, upsampling = 1;
  };
  std::array<Extra, JXLH_MAX_EXTRA_CHANNELS> extra{};
  std::vector<std::string> stages;  // Display strings, in order (diagnostics; what `info!("adding stage")` logs)
};

class GpuRenderPipeline;

// RenderPipelineBuilder (builder.rs:19-120).  `base` carries what is not a stage: frame size, quantiser and colour
// correlation fields, the EPF sharpness LUT / quant_mul of the sigma map, flags.
class RenderPipelineBuilder {
 public:
  // builder.rs:70-85: num_channels (3 colour + extra + noise temporaries), size = FrameHeader::size_upsampled(),
  // downsampling_shift = log2(upsampling), log_group_size = FrameHeader::log_group_dim() (8 on this path)
  RenderPipelineBuilder(size_t num_channels, std::pair<size_t, size_t> size, size_t downsampling_shift, size_t log_group_size,
                        const jxlh_frame_params& base)
      : num_channels_(num_channels), size_(size), downsampling_shift_(downsampling_shift), log_group_size_(log_group_size),
        base_(base) {}

  template <class S>
  RenderPipelineBuilder add_inout_stage(S stage) && {
    stages_.emplace_back(std::move(stage));
    return std::move(*this);
  }
  template <class S>
  RenderPipelineBuilder add_inplace_stage(S stage) && {
    stages_.emplace_back(std::move(stage));
    return std::move(*this);
  }
  RenderPipelineBuilder add_save_stage(std::vector<int> channels, int output_buffer_index, uint32_t color_channels,
                                       uint32_t bits) && {
    stages_.emplace_back(SaveStage{std::move(channels), output_buffer_index, color_channels, bits});
    return std::move(*this);
  }
  RenderPipelineBuilder add_extend_stage() && {
    stages_.emplace_back(CpuOnlyStage{"extend to image dimensions"});
    return std::move(*this);
  }

  // The host-side half of build(): validation + lowering, no device needed.  Throws Error(JXLH_ERR_UNSUPPORTED)
  // for a stage list outside this path and Error(JXLH_ERR_INVALID_ARGUMENT) for an inconsistent one.
  LoweredPipeline lower() const;
  // builder.rs:120: returns the pipeline (here: begins the frame on the context with the lowered parameters)
  std::unique_ptr<GpuRenderPipeline> build(Context& ctx) &&;
  // the same for a Modular frame's list (it opens with ConvertModularToF32Stage x3 or ConvertModularXYBToF32Stage)
  std::unique_ptr<class GpuModularPipeline> build_modular(Context& ctx) &&;

 private:
  [[noreturn]] static void fail(jxlh_status st, const std::string& what) { throw Error(st, "RenderPipelineBuilder::build", what); }
  size_t num_channels_;
  std::pair<size_t, size_t> size_;
  size_t downsampling_shift_, log_group_size_;
  jxlh_frame_params base_;
  std::vector<Stage> stages_;
};

inline LoweredPipeline RenderPipelineBuilder::lower() const {
  LoweredPipeline lp;
  lp.frame = base_;
  jxlh_frame_params& p = lp.frame;
  if (log_group_size_ != 8 + downsampling_shift_ && log_group_size_ != 8)
    fail(JXLH_ERR_UNSUPPORTED, "group dimension other than 256 (FrameHeader::log_group_dim)");
  if (num_channels_ < 3) fail(JXLH_ERR_INVALID_ARGUMENT, "fewer than three channels");
  // stage-derived fields start from "no stage"
  p.gab = 0;
  p.epf_iters = 0;
  p.upsampling = 1;
  p.noise = 0;
  for (int c = 0; c < 3; c++) p.hshift[c] = p.vshift[c] = 0;
  // The reference's order (frame/render.rs:568-790) as phases; a stage may only appear in a phase >= the current one.
  enum Phase { kModular, kChroma, kGab, kEpf0, kEpf1, kEpf2, kUpsample, kNoiseConvolve, kNoiseAdd, kColour, kTransfer, kConvert, kSave, kDone };
  int phase = kModular;
  auto enter = [&](int ph, const Stage& s) {
    if (ph < phase) fail(JXLH_ERR_UNSUPPORTED, "stage '" + stage_display(s) + "' out of the order of Frame::build_render_pipeline");
    phase = ph;
  };
  bool gab_seen[3] = {false, false, false};
  int ups_seen = 0, ups_factor = 0, conv_seen = 0, convert_seen = 0, modular_seen = 0;
  const float*(&ec_weights)[3] = lp.weights_by_factor;  // index n >> 2: factor 2, 4, 8
  uint32_t convert_bits = 0;
  bool have_colour = false, have_tf = false, have_save = false, pre_upsample = true, epf1_seen = false, epf2_seen = false;
  Border border{0, 0};
  auto add_border = [&](Border b) {
    if (pre_upsample) {
      border.x = (uint8_t)(border.x + b.x);
      border.y = (uint8_t)(border.y + b.y);
    }
  };
  lp.output.color = JXLH_COLOR_NONE;
  lp.output.transfer = JXLH_TF_LINEAR;
  lp.output.bits = 0;
  lp.output.channels = 3;
  for (const Stage& s : stages_) {
    lp.stages.push_back(stage_display(s));
    if (const auto* st = std::get_if<CpuOnlyStage>(&s)) {
      fail(JXLH_ERR_UNSUPPORTED, "stage '" + st->name + "' is not part of the device path");
    } else if (const auto* m = std::get_if<ConvertModularToF32Stage>(&s)) {
      enter(kModular, s);
      if (m->bit_depth < 1 || m->bit_depth > 31) fail(JXLH_ERR_INVALID_ARGUMENT, "bit depth");
      if (m->channel > 2) {  // an extra channel: any frame encoding, after the colour conversions
        const int ec = m->channel - 3;
        if (ec >= JXLH_MAX_EXTRA_CHANNELS) fail(JXLH_ERR_UNSUPPORTED, "more extra channels than JXLH_MAX_EXTRA_CHANNELS");
        if (m->channel >= (int)num_channels_ || lp.extra[ec].bits || (ec > 0 && !lp.extra[ec - 1].bits))
          fail(JXLH_ERR_INVALID_ARGUMENT, "Modular -> f32 conversion of extra channels: channels 3.. in order, once each");
        if (modular_seen != 0 && modular_seen != 3) fail(JXLH_ERR_INVALID_ARGUMENT, "extra channel conversion between the colour conversions");
        lp.extra[ec].bits = m->bit_depth;
        continue;
      }
      if (m->channel != modular_seen || lp.modular == LoweredPipeline::Modular::kXybToF32 || lp.extra[0].bits)
        fail(JXLH_ERR_INVALID_ARGUMENT, "Modular -> f32 conversion: channels 0, 1, 2 in order, before the extra channels");
      if (modular_seen && m->bit_depth != lp.modular_bits) fail(JXLH_ERR_UNSUPPORTED, "colour channels of different bit depths");
      lp.modular = LoweredPipeline::Modular::kToF32;
      lp.modular_bits = m->bit_depth;
      modular_seen++;
    } else if (const auto* mx = std::get_if<ConvertModularXYBToF32Stage>(&s)) {
      enter(kModular, s);
      if (mx->first_channel != 0 || lp.modular != LoweredPipeline::Modular::kNone) fail(JXLH_ERR_INVALID_ARGUMENT, "one XYB conversion on channels 0..2");
      lp.modular = LoweredPipeline::Modular::kXybToF32;
      lp.modular_quant_factors = mx->quant_factors;
      modular_seen = 3;
    } else if (const auto* h = std::get_if<HorizontalChromaUpsample>(&s)) {
      enter(kChroma, s);
      if (h->channel < 0 || h->channel > 2) fail(JXLH_ERR_INVALID_ARGUMENT, "chroma upsampling of a non-colour channel");
      p.hshift[h->channel] = 1;
    } else if (const auto* v = std::get_if<VerticalChromaUpsample>(&s)) {
      enter(kChroma, s);
      if (v->channel < 0 || v->channel > 2) fail(JXLH_ERR_INVALID_ARGUMENT, "chroma upsampling of a non-colour channel");
      p.vshift[v->channel] = 1;
    } else if (const auto* g = std::get_if<GaborishStage>(&s)) {
      enter(kGab, s);
      if (g->channel < 0 || g->channel > 2 || gab_seen[g->channel]) fail(JXLH_ERR_INVALID_ARGUMENT, "Gaborish: one stage per colour channel");
      gab_seen[g->channel] = true;
      p.gab = 1;
      p.gab_w1[g->channel] = g->weight1;
      p.gab_w2[g->channel] = g->weight2;
      if (g->channel == 0) add_border(GaborishStage::BORDER);
    } else if (const auto* e0 = std::get_if<Epf0Stage>(&s)) {
      enter(kEpf0, s);
      p.epf_pass0_sigma_scale = e0->sigma_scale;
      p.epf_border_sad_mul = e0->border_sad_mul;
      for (int c = 0; c < 3; c++) p.epf_channel_scale[c] = e0->channel_scale[c];
      p.epf_iters = 3;  // confirmed below: EPF0 only ever runs together with EPF1 and EPF2 (epf_iters >= 3)
      add_border(Epf0Stage::BORDER);
      phase = kEpf1;
    } else if (const auto* e1 = std::get_if<Epf1Stage>(&s)) {
      enter(kEpf1, s);
      if (e1->sigma_scale != 1.0f) fail(JXLH_ERR_UNSUPPORTED, "EPF1 with a sigma scale other than 1 (frame/render.rs:608-609)");
      p.epf_border_sad_mul = e1->border_sad_mul;
      for (int c = 0; c < 3; c++) p.epf_channel_scale[c] = e1->channel_scale[c];
      if (p.epf_iters == 0) p.epf_iters = 1;
      add_border(Epf1Stage::BORDER);
      phase = kEpf2;
      epf1_seen = true;
    } else if (const auto* e2 = std::get_if<Epf2Stage>(&s)) {
      enter(kEpf2, s);
      if (!epf1_seen) fail(JXLH_ERR_UNSUPPORTED, "EPF2 without EPF1 (epf_iters >= 2 implies the first pass)");
      p.epf_pass2_sigma_scale = e2->sigma_scale;
      if (p.epf_iters < 2) p.epf_iters = 2;
      add_border(Epf2Stage::BORDER);
      phase = kUpsample;
      epf2_seen = true;
    } else if (std::holds_alternative<Upsample2x>(s) || std::holds_alternative<Upsample4x>(s) || std::holds_alternative<Upsample8x>(s)) {
      enter(kUpsample, s);
      int n = 0, ch = 0;
      const float* w = nullptr;
      if (const auto* u = std::get_if<Upsample2x>(&s)) n = 2, ch = u->channel, w = u->weights;
      if (const auto* u = std::get_if<Upsample4x>(&s)) n = 4, ch = u->channel, w = u->weights;
      if (const auto* u = std::get_if<Upsample8x>(&s)) n = 8, ch = u->channel, w = u->weights;
      if (ch > 2) {  // an extra channel's own factor, or the frame's when it follows channel 2 (late_ec_upsample)
        const int ec = ch - 3;
        if (ec >= JXLH_MAX_EXTRA_CHANNELS || !lp.extra[ec].bits) fail(JXLH_ERR_INVALID_ARGUMENT, "upsampling of an extra channel the list never converted to f32");
        if (lp.extra[ec].upsampling != 1) fail(JXLH_ERR_INVALID_ARGUMENT, "an extra channel is upsampled once");
        if (ups_seen != 0 && (ups_seen != 3 || n != ups_factor)) fail(JXLH_ERR_INVALID_ARGUMENT, "extra channels upsampled with the colour channels use the frame's factor, after channel 2");
        if (ec_weights[n >> 2] && ec_weights[n >> 2] != w) fail(JXLH_ERR_INVALID_ARGUMENT, "one weight table per upsampling factor (CustomTransformData)");
        ec_weights[n >> 2] = w;
        lp.extra[ec].upsampling = (uint32_t)n;
        continue;
      }
      if (ec_weights[n >> 2] && ec_weights[n >> 2] != w) fail(JXLH_ERR_INVALID_ARGUMENT, "one weight table per upsampling factor (CustomTransformData)");
      if ((ups_factor && ups_factor != n) || ch != ups_seen) fail(JXLH_ERR_INVALID_ARGUMENT, "frame upsampling: channels 0, 1, 2 with one factor");
      if (ups_seen && w != lp.upsampling_weights) fail(JXLH_ERR_INVALID_ARGUMENT, "frame upsampling: one weight table for the three channels");
      ups_factor = n;
      ups_seen++;
      lp.upsampling_weights = w;
      ec_weights[n >> 2] = w;
      p.upsampling = (uint32_t)n;
      pre_upsample = false;
    } else if (const auto* cn = std::get_if<ConvolveNoiseStage>(&s)) {
      enter(kNoiseConvolve, s);
      if (cn->channel != (int)num_channels_ - 3 + conv_seen) fail(JXLH_ERR_INVALID_ARGUMENT, "noise convolution: the three temporaries behind the image channels");
      conv_seen++;
    } else if (const auto* an = std::get_if<AddNoiseStage>(&s)) {
      enter(kNoiseAdd, s);
      if (conv_seen != 3 || an->first_channel != (int)num_channels_ - 3) fail(JXLH_ERR_INVALID_ARGUMENT, "AddNoise needs the three convolved noise channels");
      p.noise = 1;
      for (int i = 0; i < 8; i++) p.noise_lut[i] = an->lut[i];
      p.ytox_lf = an->ytox_lf;
      p.ytob_lf = an->ytob_lf;
      phase = kColour;
    } else if (const auto* x = std::get_if<XybStage>(&s)) {
      enter(kColour, s);
      if (x->first_channel != 0 || have_colour) fail(JXLH_ERR_INVALID_ARGUMENT, "one colour stage on channels 0..2");
      have_colour = true;
      lp.output.color = JXLH_COLOR_XYB;
      lp.output.xyb = x->params;
      phase = kTransfer;
    } else if (const auto* yc = std::get_if<YcbcrToRgbStage>(&s)) {
      enter(kColour, s);
      if (yc->first_channel != 0 || have_colour) fail(JXLH_ERR_INVALID_ARGUMENT, "one colour stage on channels 0..2");
      have_colour = true;
      lp.output.color = JXLH_COLOR_YCBCR;
      phase = kConvert;  // no transfer-function stage behind YCbCr (frame/render.rs:755-763)
    } else if (const auto* tf = std::get_if<FromLinearStage>(&s)) {
      enter(kTransfer, s);
      if (lp.output.color != JXLH_COLOR_XYB || have_tf || tf->first_channel != 0) fail(JXLH_ERR_UNSUPPORTED, "FromLinearStage without a preceding XybStage");
      if (tf->transfer > JXLH_TF_GAMMA) fail(JXLH_ERR_INVALID_ARGUMENT, "unknown transfer function");
      have_tf = true;
      lp.output.transfer = tf->transfer;
      lp.output.tf_param = tf->param;
      for (int i = 0; i < 3; i++) lp.output.hlg_luminance_rgb[i] = tf->hlg_luminance_rgb[i];
      phase = kConvert;
    } else if (std::holds_alternative<ConvertF32ToU8Stage>(s) || std::holds_alternative<ConvertF32ToU16Stage>(s)) {
      enter(kConvert, s);
      int ch;
      uint32_t bits, depth;
      if (const auto* c8 = std::get_if<ConvertF32ToU8Stage>(&s)) ch = c8->channel, bits = 8, depth = c8->bit_depth;
      else ch = std::get<ConvertF32ToU16Stage>(s).channel, bits = 16, depth = std::get<ConvertF32ToU16Stage>(s).bit_depth;
      if (ch != convert_seen || ch > 2 || (convert_bits && convert_bits != bits)) fail(JXLH_ERR_INVALID_ARGUMENT, "integer conversion: channels 0, 1, 2 with one format");
      if (depth != bits) fail(JXLH_ERR_UNSUPPORTED, "integer output with a bit depth below the sample size");
      convert_bits = bits;
      convert_seen++;
    } else if (const auto* sv = std::get_if<SaveStage>(&s)) {
      enter(kSave, s);
      if (have_save) fail(JXLH_ERR_UNSUPPORTED, "more than one save stage (extra-channel outputs stay on the CPU pipeline)");
      if (sv->channels != std::vector<int>{0, 1, 2} || sv->output_buffer_index != 0) fail(JXLH_ERR_UNSUPPORTED, "save stage other than the colour channels into buffer 0");
      if (sv->bits == 32) {
        if (convert_seen) fail(JXLH_ERR_INVALID_ARGUMENT, "f32 save stage behind an integer conversion");
        if (sv->color_channels != 3) fail(JXLH_ERR_UNSUPPORTED, "planar f32 output has three channels");
      } else {
        if (convert_seen != 3 || convert_bits != sv->bits) fail(JXLH_ERR_INVALID_ARGUMENT, "save format and conversion stages disagree");
        if (sv->color_channels != 3 && sv->color_channels != 4) fail(JXLH_ERR_INVALID_ARGUMENT, "RGB or RGBA");
        lp.output.bits = sv->bits;
        lp.output.channels = sv->color_channels;
        lp.has_output = true;
      }
      have_save = true;
      phase = kDone;
    }
  }
  // consistency of the whole list
  if (lp.modular == LoweredPipeline::Modular::kToF32 && modular_seen != 3) fail(JXLH_ERR_INVALID_ARGUMENT, "Modular -> f32 conversion on some channels only");
  if (lp.modular != LoweredPipeline::Modular::kNone) {
    // what this path runs for a Modular frame: the conversion, Gaborish / EPF with the constant sigma of
    // features/epf.rs:81-84 (jxlh_modular_frame_filters), then either the planes themselves or -- the builder's special
    // case -- straight to bytes
    if (p.upsampling != 1 || p.noise || p.hshift[0] | p.hshift[1] | p.hshift[2] | p.vshift[0] | p.vshift[1] | p.vshift[2])
      fail(JXLH_ERR_UNSUPPORTED, "upsampling / noise / chroma subsampling on a Modular frame");
    const bool untouched = !p.gab && p.epf_iters == 0 && !have_colour && !have_tf;
    if (convert_seen == 3) {
      // builder.rs:152-170: ConvertModularToF32(c, d) whose next use is ConvertF32ToU8(c, b) with b % d == 0 becomes
      // ConvertI32ToU8Stage(c, ((1 << b) - 1) / ((1 << d) - 1), (1 << b) - 1) and the second stage disappears
      if (lp.modular == LoweredPipeline::Modular::kToF32 && untouched && convert_bits == 8 && 8 % lp.modular_bits == 0) {
        lp.modular = LoweredPipeline::Modular::kI32ToU8;
        lp.i32_to_u8_multiplier = 255 / ((1 << lp.modular_bits) - 1);
        lp.i32_to_u8_max = 255;
      } else {
        fail(JXLH_ERR_UNSUPPORTED, "integer output of a Modular frame other than the I32 -> U8 special case");
      }
    } else if (have_colour || have_tf) {
      fail(JXLH_ERR_UNSUPPORTED, "colour stages on a Modular frame");
    }
  }
  if (p.gab && !(gab_seen[0] && gab_seen[1] && gab_seen[2])) fail(JXLH_ERR_INVALID_ARGUMENT, "Gaborish on some channels only");
  if (p.epf_iters == 3 && !(epf1_seen && epf2_seen)) fail(JXLH_ERR_UNSUPPORTED, "EPF0 without EPF1 and EPF2 (epf_iters >= 3 runs all three)");
  if (ups_seen != 0 && ups_seen != 3) fail(JXLH_ERR_INVALID_ARGUMENT, "frame upsampling on some channels only");
  if ((size_t)ups_factor != (ups_factor ? (size_t)1 << downsampling_shift_ : 0) && ups_factor != 0) fail(JXLH_ERR_INVALID_ARGUMENT, "upsampling factor and downsampling_shift disagree");
  if (!ups_factor && downsampling_shift_ != 0) fail(JXLH_ERR_INVALID_ARGUMENT, "downsampling_shift without upsampling stages");
  if (conv_seen != 0 && !p.noise) fail(JXLH_ERR_INVALID_ARGUMENT, "noise convolution without AddNoise");
  if (convert_seen != 0 && convert_seen != 3) fail(JXLH_ERR_INVALID_ARGUMENT, "integer conversion on some channels only");
  if (!have_save) fail(JXLH_ERR_INVALID_ARGUMENT, "no save stage");
  if (lp.has_output && lp.output.color == JXLH_COLOR_NONE && have_tf) fail(JXLH_ERR_INVALID_ARGUMENT, "transfer function without colour stage");
  // size = FrameHeader::size_upsampled() (builder.rs:70-85): the frame itself is size >> downsampling_shift
  const uint32_t n = p.upsampling;
  if (size_.first == 0 || size_.second == 0) fail(JXLH_ERR_INVALID_ARGUMENT, "empty frame");
  p.xsize_upsampled = n > 1 ? (uint32_t)size_.first : 0;
  p.ysize_upsampled = n > 1 ? (uint32_t)size_.second : 0;
  if (n > 1) {
    if ((size_.first + n - 1) / n != p.xsize || (size_.second + n - 1) / n != p.ysize)
      fail(JXLH_ERR_INVALID_ARGUMENT, "size_upsampled does not belong to the base frame size");
  } else if (size_.first != p.xsize || size_.second != p.ysize) {
    fail(JXLH_ERR_INVALID_ARGUMENT, "pipeline size and frame size disagree");
  }
  lp.input_border = border;
  return lp;
}

// RenderPipeline (render/mod.rs:116-157) for this path: inputs are a group's coefficient slabs instead of its pixel
// buffers (the transforms run on the device too), everything else keeps the trait's meaning.
class GpuRenderPipeline {
 public:
  GpuRenderPipeline(Context& ctx, LoweredPipeline lp) : ctx_(ctx), lp_(std::move(lp)), frame_(ctx, lp_.frame) {
    const float* const* w = lp_.weights_by_factor;
    if (w[0] || w[1] || w[2])  // CustomTransformData::weights{2,4,8} of the factors in use; the others keep their state
      ctx_.check(jxlh_set_upsampling_weights(ctx_.raw(), w[0], w[1], w[2]), "jxlh_set_upsampling_weights");
  }
  VarDctFrame& frame() { return frame_; }  // decode_hf_global / decode_lf_group / decode_hf_metadata go here
  const LoweredPipeline& lowered() const { return lp_; }
  // render/mod.rs:128-136.  complete = false: a progressive pass that leaves the group open
  void set_buffer_for_group(uint32_t group_id, bool complete, const int32_t* coeffs, int slot = 0) {
    ctx_.check(jxlh_submit_group(ctx_.raw(), slot, group_id, coeffs, complete ? JXLH_GROUP_COMPLETE : 0u), "jxlh_submit_group");
    // the reference renders every group it is handed (render/mod.rs:128-136): once the frame has been rendered, a group
    // that receives new coefficients is re-rendered whether or not the caller also marks it (duplicates are merged by
    // jxlh_frame_rerender_groups)
    if (!dirty_first_) rerender_.push_back(group_id);
  }
  // render/mod.rs:146
  void mark_group_to_rerender(uint32_t g) { rerender_.push_back(g); }
  // render/mod.rs:141-144: nothing of the frame lies outside what the groups cover on this path
  void render_outside_frame() {}
  // render/mod.rs:138: the caller's buffer must hold out_height rows of out_width * channels samples
  void check_buffer_sizes(size_t bytes_per_row, size_t rows) const {
    const size_t need = lp_.has_output ? (size_t)frame_.out_width() * lp_.output.channels * (lp_.output.bits / 8)
                                       : (size_t)frame_.out_width() * sizeof(float);
    if (bytes_per_row < need || rows < frame_.out_height())
      throw Error(JXLH_ERR_INVALID_ARGUMENT, "GpuRenderPipeline::check_buffer_sizes", "output buffer too small");
  }
  // what the reference does when the last group of a pass has been handed over (frame/decode.rs:547-558, :703-711)
  void do_render() {
    for (int s = 0; s < ctx_.n_slots(); s++) ctx_.check(jxlh_slot_wait(ctx_.raw(), s), "jxlh_slot_wait");
    if (!rerender_.empty() && !dirty_first_) {
      ctx_.check(jxlh_frame_rerender_groups(ctx_.raw(), rerender_.data(), (uint32_t)rerender_.size()), "jxlh_frame_rerender_groups");
    } else {
      frame_.finalize_and_render();
    }
    dirty_first_ = false;
    rerender_.clear();
  }
  // the save stage: interleaved integers through the colour tail, or the three f32 planes
  void save(void* out) {
    if (lp_.has_output) frame_.read_output(lp_.output, out);
    else throw Error(JXLH_ERR_INVALID_ARGUMENT, "GpuRenderPipeline::save", "planar f32 pipeline: use save_planes");
  }
  void save_planes(float* c0, float* c1, float* c2) { frame_.read_planes(c0, c1, c2); }
  // An extra channel's integer samples as the Modular decoder leaves them (w x h at the channel's own resolution, host
  // or device memory).  The stage list decides what happens to them: ConvertModularToF32Stage with the channel's bit
  // depth, then its Upsample stage if it has one; both run inside do_render() behind the colour channels' stages.
  void set_extra_channel_buffer(uint32_t ec, const int32_t* samples, size_t stride, uint32_t w, uint32_t h) {
    if (ec >= JXLH_MAX_EXTRA_CHANNELS || !lp_.extra[ec].bits)
      throw Error(JXLH_ERR_INVALID_ARGUMENT, "GpuRenderPipeline::set_extra_channel_buffer", "the stage list does not name this extra channel");
    ctx_.check(jxlh_frame_set_extra_channel(ctx_.raw(), ec, samples, stride, w, h, lp_.extra[ec].bits, lp_.extra[ec].upsampling),
               "jxlh_frame_set_extra_channel");
  }
  // the save stage of an extra channel: one f32 plane of the frame's output size (out_width() x out_height())
  void save_extra_channel(uint32_t ec, float* out, size_t stride) {
    jxlh_plane pl{out, (size_t)frame_.out_width() * sizeof(float), frame_.out_height(), stride * sizeof(float)};
    ctx_.check(jxlh_frame_read_extra_channel(ctx_.raw(), ec, &pl), "jxlh_frame_read_extra_channel");
  }

 private:
  Context& ctx_;
  LoweredPipeline lp_;
  VarDctFrame frame_;
  std::vector<uint32_t> rerender_;
  bool dirty_first_ = true;
};

// A Modular frame's stage list on the device: the samples come out of the Modular inverse transforms (jxlh_rct,
// jxlh_palette*, jxlh_unsqueeze_chain) as three i32 planes, host or device memory.
class GpuModularPipeline {
 public:
  GpuModularPipeline(Context& ctx, LoweredPipeline lp) : ctx_(ctx), lp_(std::move(lp)) {
    if (lp_.modular == LoweredPipeline::Modular::kNone)
      throw Error(JXLH_ERR_INVALID_ARGUMENT, "GpuModularPipeline", "the stage list holds no Modular conversion");
  }
  const LoweredPipeline& lowered() const { return lp_; }
  // ConvertI32ToU8Stage: interleaved bytes (3 or 4 per pixel) in one pass
  void render_u8(const int32_t* const planes[3], size_t stride, void* out, size_t bytes_per_row) {
    if (lp_.modular != LoweredPipeline::Modular::kI32ToU8)
      throw Error(JXLH_ERR_INVALID_ARGUMENT, "GpuModularPipeline::render_u8", "not the I32 -> U8 special case");
    ctx_.check(jxlh_modular_to_rgb8(ctx_.raw(), planes, stride, lp_.frame.xsize, lp_.frame.ysize, lp_.i32_to_u8_multiplier,
                                    lp_.i32_to_u8_max, lp_.output.channels, out, bytes_per_row),
               "jxlh_modular_to_rgb8");
  }
  // conversion to f32 and, if the list holds them, Gaborish / EPF with the frame's constant sigma
  // (jxlh_modular_frame_filters: device planes, 16-byte aligned, row stride a multiple of 4 floats, tmp != out).
  // in[] in coded order for an XYB frame (Y, X, B); the result (X, Y, B) lands in out[]; tmp[] is only used when the
  // list holds filters.  Planes are tight: stride = xsize, which must then be a multiple of 4.
  void render_f32(const int32_t* const in[3], float* const tmp[3], float* const out[3]) {
    const size_t w = lp_.frame.xsize, h = lp_.frame.ysize, n = w * h;
    const bool filters = lp_.frame.gab || lp_.frame.epf_iters;
    float* const* dst = filters ? tmp : out;
    if (lp_.modular == LoweredPipeline::Modular::kXybToF32) {
      ctx_.check(jxlh_modular_xyb_to_f32(ctx_.raw(), in[0], in[1], in[2], n, lp_.modular_quant_factors.data(), dst[0], dst[1], dst[2]),
                 "jxlh_modular_xyb_to_f32");
    } else if (lp_.modular == LoweredPipeline::Modular::kToF32) {
      for (int c = 0; c < 3; c++)
        ctx_.check(jxlh_modular_to_f32(ctx_.raw(), in[c], n, lp_.modular_bits, dst[c]), "jxlh_modular_to_f32");
    } else {
      throw Error(JXLH_ERR_INVALID_ARGUMENT, "GpuModularPipeline::render_f32", "the list lowers to the I32 -> U8 special case");
    }
    if (filters) {
      if (w % 4) throw Error(JXLH_ERR_UNSUPPORTED, "GpuModularPipeline::render_f32", "filters need a row length that is a multiple of 4 samples");
      ctx_.check(jxlh_modular_frame_filters(ctx_.raw(), &lp_.frame, tmp, out, (uint32_t)w, (uint32_t)h, w), "jxlh_modular_frame_filters");
    }
  }

 private:
  Context& ctx_;
  LoweredPipeline lp_;
};

inline std::unique_ptr<GpuRenderPipeline> RenderPipelineBuilder::build(Context& ctx) && {
  return std::make_unique<GpuRenderPipeline>(ctx, lower());
}

inline std::unique_ptr<GpuModularPipeline> RenderPipelineBuilder::build_modular(Context& ctx) && {
  return std::make_unique<GpuModularPipeline>(ctx, lower());
}

}  // namespace jxlh
