"""ctypes binding of the C ABI in include/jxl_hip.h (libjxl_hip.so, built in-tree by
jxl_rs_amd/csrc/Makefile).  This is the same surface a Rust `extern "C"` block binds
(INTEGRATION.md); Python is only the test / bench harness language here.

There is no fallback: if the shared library is missing, importing this module raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# JXLH_LIBRARY: developer override used to A/B kernel build variants (tools/build_variant.sh)
LIB_PATH = os.environ.get("JXLH_LIBRARY") or os.path.join(_HERE, "libjxl_hip.so")

NUM_TRANSFORMS = 27
NUM_QUANT_TABLES = 17
GROUP_DIM = 256

OK = 0
ERR_INVALID_ARGUMENT = -1
ERR_OUT_OF_MEMORY = -2
ERR_DEVICE = -3
ERR_BAD_STATE = -4
ERR_INVALID_TRANSFORM = -5
ERR_UNSUPPORTED = -6
ERR_INVALID_BLOCK_SIZE = -7
ERR_BLOCK_OUT_OF_BOUNDS = -8
FRAME_UNFUSED_FILTERS = 1
FRAME_EXPAND_SPARSE = 2
FRAME_STRIP = 4
FRAME_DENSE_DEQUANT = 8
GROUP_COMPLETE = 1
GROUP_ACCUMULATE = 2
GROUP_ENTRIES12 = 4

# every symbol include/jxl_hip.h declares (checked by tests/test_abi_symbols.py)
ABI_SYMBOLS = [
    "jxlh_default_frame_params", "jxlh_ctx_create", "jxlh_ctx_destroy", "jxlh_status_string", "jxlh_last_error",
    "jxlh_alloc_pinned", "jxlh_free_pinned", "jxlh_frame_begin", "jxlh_frame_set_dequant_tables",
    "jxlh_frame_set_lf_quantized", "jxlh_frame_set_lf", "jxlh_frame_set_hf_meta", "jxlh_submit_group",
    "jxlh_submit_group_sparse", "jxlh_submit_groups_sparse", "jxlh_submit_groups_sparse8", "jxlh_submit_groups_sparse4",
    "jxlh_submit_groups_slots", "jxlh_slot_wait", "jxlh_slot_after",
    "jxlh_frame_coeff_buffer", "jxlh_frame_run", "jxlh_ctx_sync", "jxlh_ctx_mark", "jxlh_ctx_wait_mark", "jxlh_frame_read_planes",
    "jxlh_frame_read_planes_rect", "jxlh_frame_read_planes_rect_async",
    "jxlh_frame_device_planes", "jxlh_frame_set_extra_channel", "jxlh_frame_read_extra_channel", "jxlh_frame_read_lf", "jxlh_frame_read_rgb8", "jxlh_frame_read_rgb8_async",
    "jxlh_frame_read_rgb16", "jxlh_frame_read_ycbcr_rgb8", "jxlh_frame_read_ycbcr_rgb16", "jxlh_frame_read_output",
    "jxlh_frame_read_output_async", "jxlh_stage_chroma_upsample", "jxlh_stage_upsample",
    "jxlh_set_upsampling_weights", "jxlh_stage_noise_generate", "jxlh_stage_noise_convolve", "jxlh_stage_noise_add",
    "jxlh_stage_gaborish", "jxlh_stage_epf", "jxlh_stage_lf_smooth", "jxlh_stage_transform_to_pixels", "jxlh_rct",
    "jxlh_palette", "jxlh_palette_delta", "jxlh_modular_to_rgb8", "jxlh_modular_to_f32", "jxlh_modular_xyb_to_f32",
    "jxlh_unsqueeze", "jxlh_unsqueeze_planes", "jxlh_smooth_unsqueeze", "jxlh_unsqueeze_rct", "jxlh_palette_delta_wp",
    "jxlh_unsqueeze_levels", "jxlh_unsqueeze_chain", "jxlh_abi_version", "jxlh_covered_blocks_x", "jxlh_covered_blocks_y",
    "jxlh_quant_table_for_type", "jxlh_quant_table_size", "jxlh_comm_unique_id", "jxlh_comm_init",
    "jxlh_comm_init_local", "jxlh_comm_destroy", "jxlh_comm_band", "jxlh_frame_run_sharded", "jxlh_frame_allgather",
    "jxlh_frames_run_sharded_local", "jxlh_frames_allgather_local", "jxlh_comm_allgather",
    "jxlh_frame_rerender_groups", "jxlh_comm_allgather_local", "jxlh_palette_strided", "jxlh_modular_frame_filters",
    "jxlh_ctx_wait_stream", "jxlh_ctx_wait_event", "jxlh_ctx_tune_placement", "jxlh_ctx_record_event",
    "jxlh_frame_allgather_output", "jxlh_frames_allgather_output_local",
    "jxlh_host_pack_slots", "jxlh_host_pack_slots_many", "jxlh_slot_writer_create", "jxlh_slot_writer_destroy", "jxlh_slot_writer_begin_group",
    "jxlh_slot_writer_begin_varblock", "jxlh_slot_writer_add", "jxlh_slot_writer_add_many", "jxlh_slot_writer_end_group",
]
# developer / bench instruments: include/jxl_hip_dev.h (same library, not part of the drop-in boundary)
DEV_SYMBOLS = [
    "jxlh_timer_start", "jxlh_timer_stop", "jxlh_kernel_timing_enable", "jxlh_kernel_timing_get",
    "jxlh_kernel_timing_reset", "jxlh_selftest_recip", "jxlh_probe_copy_bandwidth", "jxlh_frame_path",
    "jxlh_flow_profile", "jxlh_frame_k1_counters", "jxlh_probe_placement",
]


class FrameParams(C.Structure):
    """jxlh_frame_params."""
    _fields_ = [
        ("abi_version", C.c_uint32),
        ("xsize", C.c_uint32), ("ysize", C.c_uint32),
        ("global_scale", C.c_uint32), ("quant_lf", C.c_uint32),
        ("lf_quant_factors", C.c_float * 3),
        ("quant_biases", C.c_float * 4),
        ("x_qm_scale", C.c_uint32), ("b_qm_scale", C.c_uint32),
        ("color_factor", C.c_uint32),
        ("base_correlation_x", C.c_float), ("base_correlation_b", C.c_float),
        ("ytox_lf", C.c_int32), ("ytob_lf", C.c_int32),
        ("gab", C.c_uint32),
        ("gab_w1", C.c_float * 3), ("gab_w2", C.c_float * 3),
        ("epf_iters", C.c_uint32),
        ("epf_sharp_lut", C.c_float * 8),
        ("epf_channel_scale", C.c_float * 3),
        ("epf_quant_mul", C.c_float), ("epf_pass0_sigma_scale", C.c_float),
        ("epf_pass2_sigma_scale", C.c_float), ("epf_border_sad_mul", C.c_float),
        ("do_lf_smoothing", C.c_uint32),
        ("flags", C.c_uint32),
        ("hshift", C.c_uint32 * 3), ("vshift", C.c_uint32 * 3),
        ("epf_sigma_for_modular", C.c_float),
        ("upsampling", C.c_uint32),
        ("xsize_upsampled", C.c_uint32), ("ysize_upsampled", C.c_uint32),
        ("noise", C.c_uint32), ("noise_lut", C.c_float * 8),
        ("visible_frame_index", C.c_uint32), ("nonvisible_frame_index", C.c_uint32),
    ]


class OutputDesc(C.Structure):
    """jxlh_output_desc."""
    _fields_ = [("color", C.c_uint32), ("transfer", C.c_uint32), ("xyb", C.c_float * 16), ("tf_param", C.c_float),
                ("hlg_luminance_rgb", C.c_float * 3), ("bits", C.c_uint32), ("channels", C.c_uint32)]


COLOR_XYB, COLOR_YCBCR, COLOR_NONE = 0, 1, 2
TF = {"linear": 0, "srgb": 1, "bt709": 2, "pq": 3, "hlg": 4, "gamma": 5}


class Plane(C.Structure):
    """jxlh_plane == RawImageBuffer."""
    _fields_ = [("ptr", C.c_void_p), ("bytes_per_row", C.c_size_t), ("num_rows", C.c_size_t),
                ("bytes_between_rows", C.c_size_t)]


class JxlHipError(RuntimeError):
    def __init__(self, status, where, detail=""):
        self.status = status
        super().__init__(f"{where}: status {status} {detail}")


_LOADED = None


def _lib():
    global _LOADED
    if _LOADED is None:
        _LOADED = load()
    return _LOADED


def load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make -C jxl_rs_amd/csrc` (or __graft_entry__.build()); "
            "there is no CPU fallback for the device path")
    L = C.CDLL(LIB_PATH)
    vp, i32, u32, sz, fp = C.c_void_p, C.c_int32, C.c_uint32, C.c_size_t, C.POINTER(C.c_float)
    L.jxlh_default_frame_params.argtypes = [C.POINTER(FrameParams), u32, u32]
    L.jxlh_ctx_create.argtypes = [i32, i32, C.POINTER(vp)]
    L.jxlh_ctx_destroy.argtypes = [vp]
    L.jxlh_ctx_destroy.restype = None
    L.jxlh_status_string.argtypes = [i32]
    L.jxlh_status_string.restype = C.c_char_p
    L.jxlh_last_error.argtypes = [vp]
    L.jxlh_last_error.restype = C.c_char_p
    L.jxlh_alloc_pinned.argtypes = [vp, sz, C.POINTER(vp)]
    L.jxlh_free_pinned.argtypes = [vp, vp]
    L.jxlh_frame_begin.argtypes = [vp, C.POINTER(FrameParams)]
    L.jxlh_frame_read_rgb8.argtypes = [vp, vp, u32, u32, u32, vp, sz]
    L.jxlh_frame_read_rgb16.argtypes = [vp, vp, u32, u32, u32, vp, sz]
    L.jxlh_frame_read_rgb8_async.argtypes = [vp, vp, u32, u32, u32, vp, sz]
    L.jxlh_frame_read_output.argtypes = [vp, C.POINTER(OutputDesc), u32, u32, vp, sz]
    L.jxlh_frame_read_output_async.argtypes = [vp, C.POINTER(OutputDesc), u32, u32, vp, sz]
    L.jxlh_frame_read_ycbcr_rgb8.argtypes = [vp, u32, u32, u32, vp, sz]
    L.jxlh_frame_read_ycbcr_rgb16.argtypes = [vp, u32, u32, u32, vp, sz]
    L.jxlh_stage_chroma_upsample.argtypes = [vp, vp, vp, u32, u32, i32]
    L.jxlh_stage_upsample.argtypes = [vp, i32, vp, vp, u32, u32]
    L.jxlh_stage_noise_generate.argtypes = [vp, u32, u32, u32, u32, C.POINTER(vp)]
    L.jxlh_stage_noise_convolve.argtypes = [vp, vp, vp, u32, u32]
    L.jxlh_stage_noise_add.argtypes = [vp, C.POINTER(FrameParams), C.POINTER(vp), C.POINTER(vp), sz]
    L.jxlh_set_upsampling_weights.argtypes = [vp, vp, vp, vp]
    L.jxlh_selftest_recip.argtypes = [vp, u32, u32, C.POINTER(C.c_uint64)]
    if hasattr(L, "jxlh_flow_profile"):  # absent from older builds used in A/B runs (JXLH_LIBRARY)
        L.jxlh_flow_profile.argtypes = [vp, i32, C.POINTER(i32), C.POINTER(C.c_uint64), i32]
    if hasattr(L, "jxlh_frame_k1_counters"):
        L.jxlh_frame_k1_counters.argtypes = [vp, vp, i32]
    if hasattr(L, "jxlh_ctx_tune_placement"):
        L.jxlh_ctx_tune_placement.argtypes = [vp, i32, vp, i32, C.POINTER(i32), C.POINTER(i32)]
    if hasattr(L, "jxlh_probe_placement"):
        L.jxlh_probe_placement.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.jxlh_frame_set_dequant_tables.argtypes = [vp, C.POINTER(vp), C.POINTER(sz)]
    L.jxlh_frame_set_lf_quantized.argtypes = [vp, u32, u32, u32, u32, vp, vp, vp, sz, u32]
    L.jxlh_frame_set_lf.argtypes = [vp, u32, u32, u32, u32, vp, vp, vp, sz]
    L.jxlh_frame_set_hf_meta.argtypes = [vp, u32, u32, u32, u32, vp, vp, vp, sz, vp, vp, sz]
    L.jxlh_submit_group.argtypes = [vp, i32, u32, vp, u32]
    L.jxlh_slot_wait.argtypes = [vp, i32]
    L.jxlh_submit_group_sparse.argtypes = [vp, i32, u32, vp, vp, vp, u32, u32]
    L.jxlh_submit_groups_sparse.argtypes = [vp, i32, u32, vp, vp, vp, vp, u32, u32]
    if hasattr(L, "jxlh_submit_groups_slots"):  # absent from older builds used in A/B runs (JXLH_LIBRARY)
        L.jxlh_submit_groups_sparse4.argtypes = [vp, i32, u32, vp, vp, vp, vp, vp, vp, vp, u32, u32]
        L.jxlh_submit_groups_slots.argtypes = [vp, i32, u32, vp, vp, vp, vp, vp, u32, u32]
    L.jxlh_submit_groups_sparse8.argtypes = [vp, i32, u32, vp, vp, vp, vp, vp, u32, u32]
    L.jxlh_frame_coeff_buffer.argtypes = [vp, C.POINTER(vp), C.POINTER(sz)]
    L.jxlh_frame_run.argtypes = [vp, u32, u32]
    L.jxlh_ctx_sync.argtypes = [vp]
    if hasattr(L, "jxlh_slot_after"):
        L.jxlh_slot_after.argtypes = [vp, i32, vp, i32]
    if hasattr(L, "jxlh_ctx_mark"):  # absent from older builds used in A/B runs (JXLH_LIBRARY)
        L.jxlh_ctx_mark.argtypes = [vp, C.POINTER(u32)]
        L.jxlh_ctx_wait_mark.argtypes = [vp, u32]
    L.jxlh_frame_read_planes.argtypes = [vp, C.POINTER(Plane)]
    L.jxlh_frame_read_planes_rect.argtypes = [vp, u32, u32, u32, u32, C.POINTER(Plane)]
    L.jxlh_frame_read_planes_rect_async.argtypes = [vp, u32, u32, u32, u32, C.POINTER(Plane)]
    L.jxlh_frame_device_planes.argtypes = [vp, C.POINTER(vp), C.POINTER(sz)]
    L.jxlh_frame_read_lf.argtypes = [vp, vp, vp, vp, sz]
    L.jxlh_timer_start.argtypes = [vp]
    L.jxlh_timer_stop.argtypes = [vp, fp]
    L.jxlh_kernel_timing_enable.argtypes = [vp, i32]
    L.jxlh_kernel_timing_get.argtypes = [vp, i32, C.POINTER(C.c_char_p), fp, C.POINTER(i32)]
    L.jxlh_kernel_timing_reset.argtypes = [vp]
    L.jxlh_stage_gaborish.argtypes = [vp, vp, vp, u32, u32, sz, C.c_float, C.c_float]
    L.jxlh_stage_epf.argtypes = [vp, i32, C.POINTER(FrameParams), C.POINTER(vp), C.POINTER(vp), u32, u32, sz, vp, sz]
    L.jxlh_stage_lf_smooth.argtypes = [vp, C.POINTER(FrameParams), C.POINTER(vp), C.POINTER(vp), u32, u32]
    L.jxlh_stage_transform_to_pixels.argtypes = [vp, i32, u32, vp, vp, vp]
    L.jxlh_rct.argtypes = [vp, vp, vp, vp, sz, i32, i32]
    L.jxlh_palette.argtypes = [vp, vp, sz, vp, i32, sz, i32, i32, vp]
    L.jxlh_palette_delta.argtypes = [vp, vp, u32, u32, vp, i32, i32, sz, i32, i32, i32, vp]
    L.jxlh_unsqueeze_levels.argtypes = [vp, i32, i32, vp, C.POINTER(vp), sz, u32, u32, C.POINTER(vp), sz]
    L.jxlh_unsqueeze_chain.argtypes = [vp, i32, i32, vp, C.POINTER(vp), sz, u32, u32, C.POINTER(vp), sz, i32, i32]
    L.jxlh_palette_delta_wp.argtypes = [vp, vp, u32, u32, vp, i32, i32, sz, i32, i32, vp, vp]
    L.jxlh_modular_to_rgb8.argtypes = [vp, C.POINTER(vp), sz, u32, u32, i32, i32, u32, vp, sz]
    L.jxlh_modular_to_f32.argtypes = [vp, vp, sz, u32, vp]
    L.jxlh_modular_xyb_to_f32.argtypes = [vp, vp, vp, vp, sz, vp, vp, vp, vp]
    L.jxlh_unsqueeze.argtypes = [vp, i32, vp, sz, vp, sz, u32, u32, vp, sz]
    L.jxlh_unsqueeze_rct.argtypes = [vp, i32, C.POINTER(vp), sz, C.POINTER(vp), sz, u32, u32, C.POINTER(vp), sz, i32, i32]
    L.jxlh_smooth_unsqueeze.argtypes = [vp, i32, vp, sz, u32, u32, u32, u32, vp, sz, u32, u32]
    L.jxlh_unsqueeze_planes.argtypes = [vp, i32, i32, C.POINTER(vp), sz, C.POINTER(vp), sz, u32, u32, C.POINTER(vp), sz]
    L.jxlh_abi_version.restype = u32
    L.jxlh_comm_unique_id.argtypes = [vp]
    L.jxlh_comm_init.argtypes = [vp, vp, i32, i32]
    L.jxlh_comm_init_local.argtypes = [C.POINTER(vp), i32]
    L.jxlh_comm_destroy.argtypes = [vp]
    L.jxlh_comm_band.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(u32), C.POINTER(u32)]
    L.jxlh_frame_run_sharded.argtypes = [vp]
    L.jxlh_frame_allgather.argtypes = [vp]
    L.jxlh_frames_run_sharded_local.argtypes = [C.POINTER(vp), i32]
    L.jxlh_frames_allgather_local.argtypes = [C.POINTER(vp), i32]
    L.jxlh_comm_allgather.argtypes = [vp, vp, sz]
    L.jxlh_probe_copy_bandwidth.argtypes = [vp, sz, i32, fp]
    if hasattr(L, "jxlh_frame_path"):  # absent from older builds used in A/B runs (JXLH_LIBRARY)
        L.jxlh_frame_path.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
    if hasattr(L, "jxlh_frame_set_extra_channel"):
        L.jxlh_frame_set_extra_channel.argtypes = [vp, u32, vp, sz, u32, u32, u32, u32]
        L.jxlh_frame_read_extra_channel.argtypes = [vp, u32, C.POINTER(Plane)]
    L.jxlh_frame_rerender_groups.argtypes = [vp, vp, u32]
    L.jxlh_comm_allgather_local.argtypes = [C.POINTER(vp), i32, C.POINTER(vp), sz]
    L.jxlh_palette_strided.argtypes = [vp, vp, sz, vp, i32, sz, i32, i32, vp, sz]
    L.jxlh_modular_frame_filters.argtypes = [vp, C.POINTER(FrameParams), C.POINTER(vp), C.POINTER(vp), u32, u32, sz]
    # host side of the slot-bucketed form (csrc/host_pack.hip): plain CPU code, no context
    if hasattr(L, "jxlh_frame_allgather_output"):
        L.jxlh_frame_allgather_output.argtypes = [vp, C.POINTER(OutputDesc), vp, sz]
        L.jxlh_frames_allgather_output_local.argtypes = [C.POINTER(vp), i32, C.POINTER(OutputDesc), C.POINTER(vp), sz]
    if hasattr(L, "jxlh_ctx_wait_stream"):
        L.jxlh_ctx_wait_stream.argtypes = [vp, vp]
        L.jxlh_ctx_wait_event.argtypes = [vp, vp]
        L.jxlh_ctx_record_event.argtypes = [vp, vp]
    if hasattr(L, "jxlh_host_pack_slots"):  # absent from older builds used in A/B runs (JXLH_LIBRARY)
        L.jxlh_host_pack_slots.argtypes = [vp, u32, u32, vp, sz, vp, vp, vp, u32, C.POINTER(u32)]
    if hasattr(L, "jxlh_host_pack_slots_many"):
        L.jxlh_host_pack_slots_many.argtypes = [vp, vp, u32, u32, vp, sz, vp, vp, vp, u32, C.POINTER(u32), C.POINTER(sz)]
        L.jxlh_slot_writer_create.argtypes = [C.POINTER(vp)]
        L.jxlh_slot_writer_destroy.argtypes = [vp]
        L.jxlh_slot_writer_destroy.restype = None
        L.jxlh_slot_writer_begin_group.argtypes = [vp, u32, u32, vp, sz, vp, vp, u32]
        L.jxlh_slot_writer_begin_varblock.argtypes = [vp, u32, u32]
        L.jxlh_slot_writer_add.argtypes = [vp, u32, u32, i32]
        L.jxlh_slot_writer_add_many.argtypes = [vp, u32, vp, vp, sz]
        L.jxlh_slot_writer_end_group.argtypes = [vp, vp, C.POINTER(u32)]
    for name in ("jxlh_covered_blocks_x", "jxlh_covered_blocks_y", "jxlh_quant_table_for_type",
                 "jxlh_quant_table_size"):
        getattr(L, name).argtypes = [i32]
        getattr(L, name).restype = i32
    return L


def host_pack_slots(group_coeffs, group_id=0, bits12=False, entries=None, slot_counts=None, wide_capacity=4096):
    """jxlh_host_pack_slots: one group's dense slab (3 x 65536 i32) -> (entries, slot_counts [3, 1024] u8, n [3] u32,
    wide [k, 2] u32) -- the arguments of Context.submit_groups_slots.  Values outside the entries' range are split into
    repeated in-range entries (they add up on the device); `wide` only holds what no slot has room for.  entries /
    slot_counts: optional preallocated outputs (numpy arrays, e.g. views of pinned memory)."""
    L = _lib()
    g = np.ascontiguousarray(group_coeffs, dtype=np.int32).reshape(-1)
    assert g.size == 3 * 65536
    cap = 3 * 65536 + 64 * 1024  # room for split values
    if entries is None:
        entries = np.empty(cap * 3 // 2 if bits12 else cap, np.uint8 if bits12 else np.uint16)
    else:
        cap = entries.size * 2 // 3 if bits12 else entries.size
    if slot_counts is None:
        slot_counts = np.empty((3, 1024), np.uint8)
    n = np.zeros(3, np.uint32)
    wide = np.zeros((max(wide_capacity, 1), 2), np.uint32)
    nw = C.c_uint32(0)
    st = L.jxlh_host_pack_slots(_addr(g), group_id, GROUP_ENTRIES12 if bits12 else 0, _addr(entries), cap, _addr(slot_counts),
                                _addr(n), _addr(wide), wide_capacity, C.byref(nw))
    if st != 0:
        raise JxlHipError(st, "host_pack_slots")
    total = int(n.sum())
    return entries[: total * 3 // 2 if bits12 else total], slot_counts, n, wide[: nw.value]


def host_pack_slots_many(coeffs, group_ids, bits12=False, entries=None, slot_counts=None, n=None, wide_capacity=4096):
    """jxlh_host_pack_slots_many: the dense slabs of a batch of groups (coeffs [k, 3 x 65536] i32, C-contiguous rows) ->
    (entries, slot_counts [k, 3, 1024] u8, n [k, 3] u32, wide [m, 2] u32): the arrays of ONE Context.submit_groups_slots
    call for group_ids.  One C call for the whole batch (no per-group Python): what a decoder thread does for its share
    of a frame.  entries / slot_counts / n: optional preallocated outputs."""
    L = _lib()
    coeffs = np.asarray(coeffs)
    k = int(coeffs.shape[0])
    assert coeffs.dtype == np.int32 and coeffs[0].size == 3 * 65536 and all(coeffs[i].flags.c_contiguous for i in (0, k - 1))
    ids = np.ascontiguousarray(group_ids, dtype=np.uint32)
    assert ids.size == k
    ptrs = np.array([coeffs[i].ctypes.data for i in range(k)], dtype=np.uint64)
    cap = k * (3 * 65536 + 64 * 1024)
    if entries is None:
        entries = np.empty(cap * 3 // 2 if bits12 else cap, np.uint8 if bits12 else np.uint16)
    else:
        cap = entries.size * 2 // 3 if bits12 else entries.size
    if slot_counts is None:
        slot_counts = np.empty((k, 3, 1024), np.uint8)
    if n is None:
        n = np.zeros((k, 3), np.uint32)
    wide = np.zeros((max(wide_capacity, 1), 2), np.uint32)
    nw, used = C.c_uint32(0), C.c_size_t(0)
    st = L.jxlh_host_pack_slots_many(_addr(ptrs), _addr(ids), k, GROUP_ENTRIES12 if bits12 else 0, _addr(entries), cap,
                                     _addr(slot_counts), _addr(n), _addr(wide), wide_capacity, C.byref(nw), C.byref(used))
    if st != 0:
        raise JxlHipError(st, "host_pack_slots_many")
    total = int(used.value)
    return entries[: total * 3 // 2 if bits12 else total], slot_counts, n, wide[: nw.value]


class SlotWriter:
    """jxlh_slot_writer_*: the slot-bucketed form written the way the entropy loop produces coefficients
    (frame/group.rs:557-575): varblock by varblock, `coeffs[c][offset + pos] += v` one update at a time."""

    def __init__(self):
        self.L = _lib()
        p = C.c_void_p()
        st = self.L.jxlh_slot_writer_create(C.byref(p))
        if st != 0:
            raise JxlHipError(st, "slot_writer_create")
        self._w = p

    def close(self):
        if self._w:
            self.L.jxlh_slot_writer_destroy(self._w)
            self._w = None

    __del__ = close

    def _chk(self, st, where):
        if st != 0:
            raise JxlHipError(st, where)

    def begin_group(self, group_id=0, bits12=False, capacity=3 * 65536 + 65536, wide_capacity=4096):
        self._bits12 = bits12
        self._entries = np.empty(capacity * 3 // 2 if bits12 else capacity, np.uint8 if bits12 else np.uint16)
        self._counts = np.empty((3, 1024), np.uint8)
        self._wide = np.zeros((max(wide_capacity, 1), 2), np.uint32)
        self._chk(self.L.jxlh_slot_writer_begin_group(self._w, group_id, GROUP_ENTRIES12 if bits12 else 0, _addr(self._entries),
                                                      capacity, _addr(self._counts), _addr(self._wide), wide_capacity),
                  "slot_writer_begin_group")

    def begin_varblock(self, first_slot, num_slots):
        self._chk(self.L.jxlh_slot_writer_begin_varblock(self._w, first_slot, num_slots), "slot_writer_begin_varblock")

    def add(self, channel, pos, value):
        self._chk(self.L.jxlh_slot_writer_add(self._w, channel, pos, value), "slot_writer_add")

    def add_many(self, channel, pos, value):
        pos = np.ascontiguousarray(pos, dtype=np.uint32)
        value = np.ascontiguousarray(value, dtype=np.int32)
        self._chk(self.L.jxlh_slot_writer_add_many(self._w, channel, _addr(pos), _addr(value), len(pos)), "slot_writer_add_many")

    def end_group(self):
        n = np.zeros(3, np.uint32)
        nw = C.c_uint32(0)
        self._chk(self.L.jxlh_slot_writer_end_group(self._w, _addr(n), C.byref(nw)), "slot_writer_end_group")
        total = int(n.sum())
        return self._entries[: total * 3 // 2 if self._bits12 else total], self._counts, n, self._wide[: nw.value]


def comm_unique_id():
    """128-byte RCCL id (rank 0 creates it, the launcher broadcasts it to every rank)"""
    buf = (C.c_uint8 * 128)()
    st = load().jxlh_comm_unique_id(buf)
    if st != OK:
        raise JxlHipError(st, "jxlh_comm_unique_id")
    return bytes(buf)


def _ctx_array(ctxs):
    arr = (C.c_void_p * len(ctxs))(*[c._ctx for c in ctxs])
    return arr


def comm_init_local(ctxs):
    """contexts of this process become ranks 0..n-1 of an in-process group (direct device copies)"""
    st = ctxs[0].L.jxlh_comm_init_local(_ctx_array(ctxs), len(ctxs))
    if st != OK:
        raise JxlHipError(st, "jxlh_comm_init_local")


def frames_run_sharded_local(ctxs):
    st = ctxs[0].L.jxlh_frames_run_sharded_local(_ctx_array(ctxs), len(ctxs))
    if st != OK:
        raise JxlHipError(st, "jxlh_frames_run_sharded_local", ctxs[0].L.jxlh_last_error(ctxs[0]._ctx).decode())


def frames_allgather_output_local(ctxs, desc, out_ptrs, bytes_per_row):
    arr = _ctx_array(ctxs)
    outs = (C.c_void_p * len(ctxs))(*out_ptrs)
    st = load().jxlh_frames_allgather_output_local(arr, len(ctxs), C.byref(desc), outs, bytes_per_row)
    if st != 0:
        raise JxlHipError(st, "frames_allgather_output_local")


def frames_allgather_local(ctxs):
    st = ctxs[0].L.jxlh_frames_allgather_local(_ctx_array(ctxs), len(ctxs))
    if st != OK:
        raise JxlHipError(st, "jxlh_frames_allgather_local", ctxs[0].L.jxlh_last_error(ctxs[0]._ctx).decode())


def comm_allgather_local(ctxs, bufs, bytes_per_rank):
    """in-place all-gather of one buffer per in-process rank (bufs[i]: device pointer / tensor of rank i)"""
    arr = (C.c_void_p * len(bufs))(*[_addr(b).value for b in bufs])
    st = ctxs[0].L.jxlh_comm_allgather_local(_ctx_array(ctxs), len(ctxs), arr, bytes_per_rank)
    if st != OK:
        raise JxlHipError(st, "jxlh_comm_allgather_local")


def _addr(a):
    """Address of a numpy array or a raw integer device pointer."""
    if isinstance(a, (int, np.integer)):
        return C.c_void_p(int(a))
    if hasattr(a, "data_ptr"):  # torch tensor (device memory plumbing only)
        return C.c_void_p(a.data_ptr())
    if isinstance(a, DeviceArray):
        return C.c_void_p(a.ptr)
    return C.c_void_p(a.ctypes.data)


class DeviceArray:
    """A device buffer for harness code (tests, bench) that hands DEVICE pointers to the C ABI.  Allocated through the
    HIP runtime the library itself is linked against (ctypes on the already-loaded libamdhip64), not through torch: a second HIP
    runtime in the process (torch bundles its own) does not see the GPU once the first one holds it."""
    _hip = None

    @classmethod
    def hip(cls):
        if cls._hip is None:
            import ctypes as C
            cls._hip = C.CDLL("libamdhip64.so")
            cls._hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
            cls._hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
            cls._hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
            cls._hip.hipFree.argtypes = [C.c_void_p]
            cls._hip.hipSetDevice.argtypes = [C.c_int]
            cls._hip.hipStreamSynchronize.argtypes = [C.c_void_p]
        return cls._hip

    @classmethod
    def _settle(cls):
        """hipMemset returns before the fill has run, and a hipMemcpy from pageable memory may return once the data sits in
        the runtime's staging buffer: both are ordered on the NULL stream only, and the library's streams are non-blocking
        ones -- a kernel enqueued right behind such a call could run before it (seen as a 0.2 % flake of
        tests/soak_squeeze.py: an output plane zeroed after the kernel had written it).  Wait for the NULL stream."""
        cls._chk(cls.hip().hipStreamSynchronize(None), "hipStreamSynchronize(NULL)")

    @classmethod
    def _chk(cls, rc, what):
        # explicit checks, not asserts: under `python -O` an assert (and the HIP call inside it) would vanish
        if rc != 0:
            raise JxlHipError(ERR_DEVICE, what, f"hip error {rc}")

    def __init__(self, array=None, nbytes=None, device=None):
        """device: HIP device ordinal the buffer lives on (default: the calling thread's current device -- device 0
        unless something set another one; pass the Context's ordinal on multi-GPU hosts)"""
        import ctypes as C
        self.nbytes = array.nbytes if array is not None else nbytes
        self.ptr = 0
        if device is not None:
            self._chk(self.hip().hipSetDevice(int(device)), "hipSetDevice")
        p = C.c_void_p()
        self._chk(self.hip().hipMalloc(C.byref(p), max(self.nbytes, 16)), "hipMalloc")
        self.ptr = p.value
        if array is not None:
            a = np.ascontiguousarray(array)
            self._chk(self.hip().hipMemcpy(self.ptr, a.ctypes.data, a.nbytes, 1), "hipMemcpy H2D")
        else:
            self._chk(self.hip().hipMemset(self.ptr, 0, self.nbytes), "hipMemset")
        self._settle()

    def upload(self, array, byte_offset=0):
        a = np.ascontiguousarray(array)
        self._chk(self.hip().hipMemcpy(self.ptr + byte_offset, a.ctypes.data, a.nbytes, 1), "hipMemcpy H2D")
        self._settle()

    def download(self, dtype, count, byte_offset=0):
        out = np.empty(count, dtype=dtype)
        self._chk(self.hip().hipMemcpy(out.ctypes.data, self.ptr + byte_offset, out.nbytes, 2), "hipMemcpy D2H")
        return out

    def free(self):
        if self.ptr:
            self.hip().hipFree(self.ptr)
            self.ptr = 0


class SqueezeLevel(C.Structure):  # jxlh_squeeze_level
    _fields_ = [("horizontal", C.c_int32), ("out_w", C.c_uint32), ("out_h", C.c_uint32), ("res", C.c_void_p * 3),
                ("res_stride", C.c_size_t)]


class Context:
    """jxlh_ctx wrapper; raises JxlHipError on any non-zero status."""

    def __init__(self, device=0, n_slots=1):
        self.L = load()
        self.device = int(device)
        self._ctx = C.c_void_p()
        st = self.L.jxlh_ctx_create(device, n_slots, C.byref(self._ctx))
        if st != OK:
            raise JxlHipError(st, "jxlh_ctx_create", self.L.jxlh_status_string(st).decode())
        self._keep = {}  # slot -> host arrays an asynchronous H2D copy of that slot may still read

    def close(self):
        if self._ctx:
            for addr in getattr(self, "_pinned", []):
                self.L.jxlh_free_pinned(self._ctx, C.c_void_p(addr))
            self._pinned = []
            self.L.jxlh_ctx_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, st, where):
        if st != OK:
            raise JxlHipError(st, where, self.L.jxlh_status_string(st).decode() + " / " +
                              self.L.jxlh_last_error(self._ctx).decode())

    def default_params(self, xsize, ysize):
        p = FrameParams()
        self._chk(self.L.jxlh_default_frame_params(C.byref(p), xsize, ysize), "default_frame_params")
        return p

    # ---- frame ----
    def frame_begin(self, params):
        self.params = params
        # FrameHeader::size_blocks: whole blocks of the coarsest (sub-sampled) channel
        mh, mv = max(params.hshift), max(params.vshift)
        self.xblocks = -(-params.xsize // (8 << mh)) << mh
        self.yblocks = -(-params.ysize // (8 << mv)) << mv
        self._chk(self.L.jxlh_frame_begin(self._ctx, C.byref(params)), "frame_begin")

    def set_dequant_tables(self, tables):
        tabs = [np.ascontiguousarray(t, dtype=np.float32) for t in tables]
        ptrs = (C.c_void_p * NUM_QUANT_TABLES)(*[t.ctypes.data for t in tabs])
        sizes = (C.c_size_t * NUM_QUANT_TABLES)(*[t.size // 3 for t in tabs])
        self._chk(self.L.jxlh_frame_set_dequant_tables(self._ctx, ptrs, sizes), "set_dequant_tables")

    def set_lf_quantized(self, qy, qx, qb, x0=0, y0=0, extra_precision=0):
        qy, qx, qb = [np.ascontiguousarray(a, dtype=np.int32) for a in (qy, qx, qb)]
        h, w = qy.shape
        self._chk(self.L.jxlh_frame_set_lf_quantized(self._ctx, x0, y0, w, h, _addr(qy), _addr(qx), _addr(qb), w,
                                                     extra_precision), "set_lf_quantized")

    def set_lf(self, x, y, b, x0=0, y0=0):
        x, y, b = [np.ascontiguousarray(a, dtype=np.float32) for a in (x, y, b)]
        h, w = x.shape
        self._chk(self.L.jxlh_frame_set_lf(self._ctx, x0, y0, w, h, _addr(x), _addr(y), _addr(b), w), "set_lf")

    def set_hf_meta(self, transform_map, raw_quant, epf_map, ytox, ytob, x0=0, y0=0):
        tm = np.ascontiguousarray(transform_map, dtype=np.uint8)
        rq = np.ascontiguousarray(raw_quant, dtype=np.int32)
        em = np.ascontiguousarray(epf_map, dtype=np.uint8)
        yx = np.ascontiguousarray(ytox, dtype=np.int8)
        yb = np.ascontiguousarray(ytob, dtype=np.int8)
        h, w = tm.shape
        self._chk(self.L.jxlh_frame_set_hf_meta(self._ctx, x0, y0, w, h, _addr(tm), _addr(rq), _addr(em), w,
                                                _addr(yx), _addr(yb), yx.shape[1]), "set_hf_meta")

    def submit_group(self, group_id, coeffs, slot=0, flags=GROUP_COMPLETE):
        if isinstance(coeffs, np.ndarray):
            coeffs = np.ascontiguousarray(coeffs, dtype=np.int32)
            assert coeffs.size == 3 * 65536
            self._keep.setdefault(slot, []).append(coeffs)  # async H2D: keep alive until the slot is waited on
        self._chk(self.L.jxlh_submit_group(self._ctx, slot, group_id, _addr(coeffs), flags), "submit_group")

    def submit_group_sparse(self, group_id, pairs, n, wide=None, slot=0, flags=GROUP_COMPLETE):
        """pairs: uint32 array of little-endian {u16 pos; i16 val} (X run, Y run, B run); n: 3 counts;
        wide: uint32 array [k, 2] of (channel*65536 + pos, value) for values outside i16."""
        pairs = np.ascontiguousarray(pairs, dtype=np.uint32)
        n = np.ascontiguousarray(n, dtype=np.uint32)
        nw = 0 if wide is None else len(wide)
        wide = None if nw == 0 else np.ascontiguousarray(wide, dtype=np.uint32)
        self._keep.setdefault(slot, []).append((pairs, n, wide))
        self._chk(self.L.jxlh_submit_group_sparse(self._ctx, slot, group_id, _addr(pairs) if pairs.size else None,
                                                  _addr(n), None if wide is None else _addr(wide), nw, flags),
                  "submit_group_sparse")

    def submit_groups_sparse(self, group_ids, pairs, n, wide=None, slot=0, flags=GROUP_COMPLETE):
        """Batched form; pairs may be a numpy array or a raw host address (pinned memory)."""
        group_ids = np.ascontiguousarray(group_ids, dtype=np.uint32)
        n = np.ascontiguousarray(n, dtype=np.uint32)
        nw = 0 if wide is None else len(wide)
        wide = None if nw == 0 else np.ascontiguousarray(wide, dtype=np.uint32)
        addr = pairs if isinstance(pairs, int) else _addr(np.ascontiguousarray(pairs, dtype=np.uint32))
        self._keep.setdefault(slot, []).append((group_ids, pairs, n, wide))
        self._chk(self.L.jxlh_submit_groups_sparse(self._ctx, slot, len(group_ids), _addr(group_ids), addr, _addr(n),
                                                   None if wide is None else _addr(wide), nw, flags),
                  "submit_groups_sparse")

    def submit_groups_sparse8(self, group_ids, pos, val, n, wide=None, slot=0, flags=GROUP_COMPLETE):
        """3-byte form: pos (uint16) and val (int8) arrays, or raw host addresses (pinned memory)."""
        group_ids = np.ascontiguousarray(group_ids, dtype=np.uint32)
        n = np.ascontiguousarray(n, dtype=np.uint32)
        nw = 0 if wide is None else len(wide)
        wide = None if nw == 0 else np.ascontiguousarray(wide, dtype=np.uint32)
        if not isinstance(pos, int):
            pos = np.ascontiguousarray(pos, dtype=np.uint16)
            val = np.ascontiguousarray(val, dtype=np.int8)
        pa = pos if isinstance(pos, int) else _addr(pos)
        va = val if isinstance(val, int) else _addr(val)
        self._keep.setdefault(slot, []).append((group_ids, pos, val, n, wide))
        self._chk(self.L.jxlh_submit_groups_sparse8(self._ctx, slot, len(group_ids), _addr(group_ids), pa, va, _addr(n),
                                                    None if wide is None else _addr(wide), nw, flags),
                  "submit_groups_sparse8")

    def submit_groups_sparse4(self, group_ids, entries, seg_counts, pos8=None, val8=None, n8=None, wide=None, slot=0,
                              flags=GROUP_COMPLETE):
        """2-byte form (synth.to_sparse4): arrays, or raw host addresses (pinned memory) for entries / seg_counts /
        pos8 / val8."""
        group_ids = np.ascontiguousarray(group_ids, dtype=np.uint32)
        nw = 0 if wide is None else len(wide)
        wide = None if nw == 0 else np.ascontiguousarray(wide, dtype=np.uint32)

        def prep(a, dt):
            if a is None or isinstance(a, int):
                return a, (a if a else None)
            a = np.ascontiguousarray(a, dtype=dt)
            return a, (_addr(a) if a.size else None)
        entries, ea = prep(entries, np.uint16)
        seg_counts, ca = prep(seg_counts, np.uint16)
        pos8, pa = prep(pos8, np.uint16)
        val8, va = prep(val8, np.int8)
        n8 = None if n8 is None else np.ascontiguousarray(n8, dtype=np.uint32)
        self._keep.setdefault(slot, []).append((group_ids, entries, seg_counts, pos8, val8, n8, wide))
        self._chk(self.L.jxlh_submit_groups_sparse4(self._ctx, slot, len(group_ids), _addr(group_ids), ea, ca, pa, va,
                                                    None if n8 is None else _addr(n8),
                                                    None if wide is None else _addr(wide), nw, flags),
                  "submit_groups_sparse4")

    def submit_groups_slots(self, group_ids, entries, slot_counts, n, wide=None, slot=0, flags=GROUP_COMPLETE):
        """slot-bucketed 2-byte form (synth.to_slots): arrays, or raw host addresses (pinned memory) for entries /
        slot_counts"""
        group_ids = np.ascontiguousarray(group_ids, dtype=np.uint32)
        n = np.ascontiguousarray(n, dtype=np.uint32)
        nw = 0 if wide is None else len(wide)
        wide = None if nw == 0 else np.ascontiguousarray(wide, dtype=np.uint32)
        if not isinstance(entries, int):
            entries = np.ascontiguousarray(entries, dtype=np.uint8 if (flags & GROUP_ENTRIES12) else np.uint16)
        if not isinstance(slot_counts, int):
            slot_counts = np.ascontiguousarray(slot_counts, dtype=np.uint8)
        ea = entries if isinstance(entries, int) else (_addr(entries) if entries.size else None)
        ca = slot_counts if isinstance(slot_counts, int) else _addr(slot_counts)
        self._keep.setdefault(slot, []).append((group_ids, entries, slot_counts, n, wide))
        self._chk(self.L.jxlh_submit_groups_slots(self._ctx, slot, len(group_ids), _addr(group_ids), ea, ca, _addr(n),
                                                  None if wide is None else _addr(wide), nw, flags),
                  "submit_groups_slots")

    def alloc_pinned(self, nbytes):
        """Pinned host buffer (jxlh_alloc_pinned) as a uint8 numpy view; freed with the context."""
        p = C.c_void_p()
        self._chk(self.L.jxlh_alloc_pinned(self._ctx, nbytes, C.byref(p)), "alloc_pinned")
        buf = (C.c_uint8 * nbytes).from_address(p.value)
        self._pinned = getattr(self, "_pinned", []) + [p.value]
        return np.frombuffer(buf, dtype=np.uint8), p.value

    def slot_wait(self, slot=0):
        self._chk(self.L.jxlh_slot_wait(self._ctx, slot), "slot_wait")
        self._keep.pop(slot, None)

    def slot_after(self, slot, after_ctx, after_slot):
        """submissions on (self, slot) from now on start when the uploads enqueued so far on (after_ctx, after_slot) have
        landed (jxlh_slot_after): device-side ordering across contexts, the host does not block"""
        self._chk(self.L.jxlh_slot_after(self._ctx, slot, after_ctx._ctx, after_slot), "slot_after")

    def coeff_buffer(self):
        p, n = C.c_void_p(), C.c_size_t()
        self._chk(self.L.jxlh_frame_coeff_buffer(self._ctx, C.byref(p), C.byref(n)), "frame_coeff_buffer")
        return p.value, n.value

    def frame_run(self, group_row0=0, group_row1=0xFFFFFFFF):
        self._chk(self.L.jxlh_frame_run(self._ctx, group_row0, group_row1), "frame_run")

    def rerender_groups(self, group_ids):
        ids = np.ascontiguousarray(group_ids, dtype=np.uint32)
        self._chk(self.L.jxlh_frame_rerender_groups(self._ctx, _addr(ids), len(ids)), "frame_rerender_groups")

    def sync(self):
        self._chk(self.L.jxlh_ctx_sync(self._ctx), "ctx_sync")
        self._keep.clear()

    def wait_stream(self, hip_stream=None):
        """jxlh_ctx_wait_stream: whatever is enqueued so far on the caller's stream (None = the NULL stream) happens before
        what this context enqueues from now on (device-side; the hand-over of caller-filled device buffers)"""
        self._chk(self.L.jxlh_ctx_wait_stream(self._ctx, hip_stream), "ctx_wait_stream")

    def wait_event(self, hip_event):
        self._chk(self.L.jxlh_ctx_wait_event(self._ctx, hip_event), "ctx_wait_event")

    def record_event(self, hip_event):
        self._chk(self.L.jxlh_ctx_record_event(self._ctx, hip_event), "ctx_record_event")

    def mark(self):
        """a point in the main stream (jxlh_ctx_mark): everything enqueued so far"""
        m = C.c_uint32(0)
        self._chk(self.L.jxlh_ctx_mark(self._ctx, C.byref(m)), "ctx_mark")
        return m.value

    def wait_mark(self, mark):
        """blocks until that point is reached -- not for work enqueued after it (jxlh_ctx_wait_mark)"""
        self._chk(self.L.jxlh_ctx_wait_mark(self._ctx, mark), "ctx_wait_mark")

    def probe_copy_bandwidth(self, nbytes, reps=10):
        """GB/s (read + written) of a float4 device-to-device copy of nbytes"""
        v = C.c_float()
        self._chk(self.L.jxlh_probe_copy_bandwidth(self._ctx, nbytes, reps, C.byref(v)), "probe_copy_bandwidth")
        return v.value

    def set_extra_channel(self, ec, samples, bits_per_sample, ec_upsampling=1, exp_bits=0):
        """hands Modular channel 3 + ec over as decoded (i32 [h, w]); processed by the next frame_run"""
        a = np.ascontiguousarray(samples, dtype=np.int32)
        h, w = a.shape
        self._chk(self.L.jxlh_frame_set_extra_channel(self._ctx, ec, _addr(a), w, w, h, bits_per_sample | exp_bits << 8,
                                                      ec_upsampling),
                  "frame_set_extra_channel")

    def read_extra_channel(self, ec, out_w, out_h):
        out = np.zeros((out_h, out_w), dtype=np.float32)
        pl = Plane(out.ctypes.data, out_w * 4, out_h, out_w * 4)
        self._chk(self.L.jxlh_frame_read_extra_channel(self._ctx, ec, C.byref(pl)), "frame_read_extra_channel")
        return out

    def tune_placement(self, trials=0):
        """jxlh_ctx_tune_placement: trials > 0 sets the number of candidate sets the next first allocation of the large
        buffers is picked from; returns (ratings [(k1-like ms, filter-like ms), ...] of the last pick, index taken)"""
        rep = np.zeros(128, np.float32)
        n, pick = C.c_int32(0), C.c_int32(-1)
        self._chk(self.L.jxlh_ctx_tune_placement(self._ctx, int(trials), _addr(rep), rep.size, C.byref(n), C.byref(pick)),
                  "ctx_tune_placement")
        return [(float(rep[2 * i]), float(rep[2 * i + 1])) for i in range(min(n.value, rep.size) // 2)], pick.value

    def probe_placement(self):
        """jxlh_probe_placement: (k1-like ms, filter-like ms) of two byte movers on the context's own buffers"""
        a, b = C.c_float(), C.c_float()
        self._chk(self.L.jxlh_probe_placement(self._ctx, C.byref(a), C.byref(b)), "probe_placement")
        return a.value, b.value

    def k1_counters(self):
        """jxlh_frame_k1_counters: dict of the last transform launch's work-list counters"""
        out = np.zeros(29, np.int32)
        self._chk(self.L.jxlh_frame_k1_counters(self._ctx, _addr(out), 29), "frame_k1_counters")
        names = ["dct8", "dct16x8", "dct8x16", "dct16x16", "dct32x8", "dct8x32", "dct32x16", "dct16x32", "dct32x32"]
        nb = [8, 8, 8, 4, 4, 8, 4, 4, 2]   # varblocks per batch (Shape::NB)
        return {"varblocks": {k: int(v) for k, v in zip(names + ["special", "large"], out[:11])},
                "fallback_batches": {k: int(v) for k, v in zip(names, out[11:20])},
                "batches": {k: int(-(-int(v) // b)) for k, v, b in zip(names, out[:9], nb)},
                "dense_route_varblocks": {k: int(v) for k, v in zip(names, out[20:29])}}

    def frame_path(self):
        """(strip kernel ran, 64x64 tiles of the frame, tiles left to the transform class kernels) of the last frame_run"""
        a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
        self._chk(self.L.jxlh_frame_path(self._ctx, C.byref(a), C.byref(b), C.byref(c)), "frame_path")
        return bool(a.value), int(b.value), int(c.value)

    # ---- multi-GPU (one process per GPU: RCCL communicator owned by the library)
    def comm_init(self, unique_id, rank, nranks):
        """unique_id: the 128 bytes rank 0 obtained from comm_unique_id() and the launcher distributed"""
        buf = (C.c_uint8 * 128).from_buffer_copy(bytes(unique_id))
        self._chk(self.L.jxlh_comm_init(self._ctx, buf, rank, nranks), "comm_init")

    def comm_destroy(self):
        self._chk(self.L.jxlh_comm_destroy(self._ctx), "comm_destroy")

    def comm_band(self):
        """(rank, nranks, group_row0, group_row1) of this context inside the current frame"""
        r, n, a, b = C.c_int32(), C.c_int32(), C.c_uint32(), C.c_uint32()
        self._chk(self.L.jxlh_comm_band(self._ctx, C.byref(r), C.byref(n), C.byref(a), C.byref(b)), "comm_band")
        return r.value, n.value, a.value, b.value

    def frame_run_sharded(self):
        self._chk(self.L.jxlh_frame_run_sharded(self._ctx), "frame_run_sharded")

    def frame_allgather(self):
        self._chk(self.L.jxlh_frame_allgather(self._ctx), "frame_allgather")

    def comm_allgather(self, dev_ptr, bytes_per_rank):
        self._chk(self.L.jxlh_comm_allgather(self._ctx, _addr(dev_ptr), bytes_per_rank), "comm_allgather")

    @property
    def out_size(self):
        """(width, height) of what the jxlh_frame_read_* calls return: the frame, or its upsampled image"""
        p = self.params
        n = max(1, p.upsampling)
        return (p.xsize_upsampled or p.xsize * n, p.ysize_upsampled or p.ysize * n)

    def read_planes(self):
        w, h = self.out_size
        out = [np.zeros((h, w), dtype=np.float32) for _ in range(3)]
        planes = (Plane * 3)(*[Plane(o.ctypes.data, w * 4, h, w * 4) for o in out])
        self._chk(self.L.jxlh_frame_read_planes(self._ctx, planes), "frame_read_planes")
        return out

    def read_planes_rect(self, x0, y0, w, h, out=None):
        """jxlh_frame_read_planes_rect: the rect [x0, x0 + w) x [y0, y0 + h) of the three finished planes (cut at the
        result's edge) into `out` (three float32 arrays of at least the cut size; allocated h x w, zeroed, when None)"""
        if out is None:
            out = [np.zeros((h, w), dtype=np.float32) for _ in range(3)]
        planes = (Plane * 3)(*[Plane(o.ctypes.data, o.shape[1] * 4, o.shape[0], o.strides[0]) for o in out])
        self._chk(self.L.jxlh_frame_read_planes_rect(self._ctx, x0, y0, w, h, planes), "frame_read_planes_rect")
        return out

    def read_group_planes(self, group_id):
        """the reference's unit of hand-over (RenderPipeline::set_buffer_for_group, render/mod.rs:128-137): group
        `group_id` of the result on the 256 x 256 grid, in buffers rounded up to 16 pixels
        (group_size_for_channel, render/internal.rs:144-167); returns (planes, (w, h) of the group's part of the frame)"""
        W, H = self.out_size
        xg = (W + 255) // 256
        x0, y0 = (group_id % xg) * 256, (group_id // xg) * 256
        gw, gh = min(256, W - x0), min(256, H - y0)
        bw, bh = (min(256, W) + 15) // 16 * 16, (min(256, H) + 15) // 16 * 16
        out = [np.zeros((bh, bw), dtype=np.float32) for _ in range(3)]
        self.read_planes_rect(x0, y0, bw, bh, out)
        return out, (gw, gh)

    def read_rgb8(self, xyb_params, channels=3, y0=0, y1=None, out=None):
        """8-bit interleaved sRGB of rows [y0, y1) (jxlh_frame_read_rgb8).  xyb_params: 16 floats in
        jxlh_xyb_params order.  out: optional device pointer (int) with tight rows; else a numpy array is returned."""
        pr = np.ascontiguousarray(xyb_params, dtype=np.float32)
        assert pr.size == 16
        y1 = self.out_size[1] if y1 is None else y1
        if out is not None:
            self._chk(self.L.jxlh_frame_read_rgb8(self._ctx, _addr(pr), channels, y0, y1, C.c_void_p(int(out)),
                                                  self.out_size[0] * channels), "frame_read_rgb8")
            return None
        arr = np.zeros((y1 - y0, self.out_size[0], channels), dtype=np.uint8)
        self._chk(self.L.jxlh_frame_read_rgb8(self._ctx, _addr(pr), channels, y0, y1, _addr(arr),
                                              self.out_size[0] * channels), "frame_read_rgb8")
        return arr

    def read_rgb16(self, xyb_params, channels=3, y0=0, y1=None):
        """16-bit interleaved sRGB of rows [y0, y1) (jxlh_frame_read_rgb16) as a uint16 array."""
        pr = np.ascontiguousarray(xyb_params, dtype=np.float32)
        assert pr.size == 16
        y1 = self.out_size[1] if y1 is None else y1
        arr = np.zeros((y1 - y0, self.out_size[0], channels), dtype=np.uint16)
        self._chk(self.L.jxlh_frame_read_rgb16(self._ctx, _addr(pr), channels, y0, y1, _addr(arr),
                                               self.out_size[0] * channels * 2), "frame_read_rgb16")
        return arr

    @staticmethod
    def output_desc(color=COLOR_XYB, transfer="srgb", xyb_params=None, tf_param=0.0, lum=(0.2627, 0.678, 0.0593), bits=8,
                    channels=3):
        d = OutputDesc()
        d.color, d.transfer, d.bits, d.channels, d.tf_param = color, TF[transfer], bits, channels, tf_param
        if xyb_params is not None:
            for i, v in enumerate(np.asarray(xyb_params, dtype=np.float32).ravel()):
                d.xyb[i] = float(v)
        for i in range(3):
            d.hlg_luminance_rgb[i] = lum[i]
        return d

    def frame_allgather_output(self, desc, out_ptr, bytes_per_row):
        """jxlh_frame_allgather_output: this rank's band converted into the device image at out_ptr, bands all-gathered"""
        self._chk(self.L.jxlh_frame_allgather_output(self._ctx, C.byref(desc), out_ptr, bytes_per_row), "frame_allgather_output")

    def read_output(self, color=COLOR_XYB, transfer="srgb", xyb_params=None, tf_param=0.0, lum=(0.2627, 0.678, 0.0593),
                    bits=8, channels=3, y0=0, y1=None):
        """jxlh_frame_read_output: interleaved 8- or 16-bit samples after the frame's colour stage"""
        d = self.output_desc(color, transfer, xyb_params, tf_param, lum, bits, channels)
        w, h = self.out_size
        y1 = h if y1 is None else y1
        arr = np.zeros((y1 - y0, w, channels), dtype=np.uint8 if bits == 8 else np.uint16)
        self._chk(self.L.jxlh_frame_read_output(self._ctx, C.byref(d), y0, y1, _addr(arr), w * channels * (bits // 8)),
                  "frame_read_output")
        return arr

    def read_ycbcr_rgb8(self, channels=3, y0=0, y1=None):
        """8-bit interleaved RGB of a YCbCr frame (planes Cb, Y, Cr; jxlh_frame_read_ycbcr_rgb8)."""
        y1 = self.out_size[1] if y1 is None else y1
        arr = np.zeros((y1 - y0, self.out_size[0], channels), dtype=np.uint8)
        self._chk(self.L.jxlh_frame_read_ycbcr_rgb8(self._ctx, channels, y0, y1, _addr(arr),
                                                    self.out_size[0] * channels), "frame_read_ycbcr_rgb8")
        return arr

    def read_ycbcr_rgb16(self, channels=3, y0=0, y1=None):
        y1 = self.out_size[1] if y1 is None else y1
        arr = np.zeros((y1 - y0, self.out_size[0], channels), dtype=np.uint16)
        self._chk(self.L.jxlh_frame_read_ycbcr_rgb16(self._ctx, channels, y0, y1, _addr(arr),
                                                     self.out_size[0] * channels * 2), "frame_read_ycbcr_rgb16")
        return arr

    def device_planes(self):
        ptrs = (C.c_void_p * 3)()
        stride = C.c_size_t()
        self._chk(self.L.jxlh_frame_device_planes(self._ctx, ptrs, C.byref(stride)), "frame_device_planes")
        return [ptrs[i] for i in range(3)], stride.value

    def read_lf(self):
        out = [np.zeros((self.yblocks, self.xblocks), dtype=np.float32) for _ in range(3)]
        self._chk(self.L.jxlh_frame_read_lf(self._ctx, _addr(out[0]), _addr(out[1]), _addr(out[2]), self.xblocks),
                  "frame_read_lf")
        return out

    # ---- timing ----
    def timer_start(self):
        self._chk(self.L.jxlh_timer_start(self._ctx), "timer_start")

    def timer_stop(self):
        ms = C.c_float()
        self._chk(self.L.jxlh_timer_stop(self._ctx, C.byref(ms)), "timer_stop")
        return ms.value

    def kernel_timing(self, enable):
        self._chk(self.L.jxlh_kernel_timing_enable(self._ctx, 1 if enable else 0), "kernel_timing_enable")

    def kernel_timing_reset(self):
        self._chk(self.L.jxlh_kernel_timing_reset(self._ctx), "kernel_timing_reset")

    def kernel_times(self):
        out = {}
        i = 0
        while True:
            name, ms, n = C.c_char_p(), C.c_float(), C.c_int32()
            st = self.L.jxlh_kernel_timing_get(self._ctx, i, C.byref(name), C.byref(ms), C.byref(n))
            if st != OK:
                break
            out[name.value.decode()] = (ms.value, n.value)
            i += 1
        return out

    # ---- stage hooks ----
    def selftest_recip(self, lo=1.0, hi=16.0):
        """Number of floats in [lo, hi) whose fast EPF reciprocal differs from IEEE 1/w."""
        lo_b = int(np.float32(lo).view(np.uint32))
        hi_b = int(np.float32(hi).view(np.uint32))
        bad = C.c_uint64(0)
        self._chk(self.L.jxlh_selftest_recip(self._ctx, C.c_uint32(lo_b), C.c_uint32(hi_b), C.byref(bad)),
                  "selftest_recip")
        return int(bad.value)

    def flow_profile(self, enable=True):
        """timeline of the last dataflow squeeze launch (if profiling was on): a list of per-level dicts, times in us
        from the first level's start; then switches the profile on / off for the launches that follow"""
        n = C.c_int32(0)
        rows = (C.c_uint64 * (11 * 16))()
        self._chk(self.L.jxlh_flow_profile(self._ctx, 1 if enable else 0, C.byref(n), rows, 16), "flow_profile")
        if n.value == 0:
            return []
        t0 = min(rows[11 * i] for i in range(n.value))
        return [{"start_us": (rows[11 * i] - t0) / 100.0, "end_us": (rows[11 * i + 1] - t0) / 100.0,
                 "poll_wait_us_sum": rows[11 * i + 2] / 100.0, "polls": int(rows[11 * i + 3]),
                 "lifetime_us_sum": rows[11 * i + 4] / 100.0,
                 "mover_phases_us_sum": [rows[11 * i + 5 + k] / 100.0 for k in range(6)]} for i in range(n.value)]

    def stage_gaborish(self, plane, w1, w2, w=None, h=None):
        plane = np.ascontiguousarray(plane, dtype=np.float32)
        H, S = plane.shape
        w = S if w is None else w
        h = H if h is None else h
        out = np.zeros_like(plane)
        self._chk(self.L.jxlh_stage_gaborish(self._ctx, _addr(plane), _addr(out), w, h, S, w1, w2), "stage_gaborish")
        return out

    def stage_epf(self, stage, params, planes, inv_sigma, w=None, h=None):
        planes = [np.ascontiguousarray(a, dtype=np.float32) for a in planes]
        H, S = planes[0].shape
        w = S if w is None else w
        h = H if h is None else h
        sig = np.ascontiguousarray(inv_sigma, dtype=np.float32)
        out = [np.zeros_like(planes[0]) for _ in range(3)]
        pin = (C.c_void_p * 3)(*[a.ctypes.data for a in planes])
        pout = (C.c_void_p * 3)(*[a.ctypes.data for a in out])
        self._chk(self.L.jxlh_stage_epf(self._ctx, stage, C.byref(params), pin, pout, w, h, S, _addr(sig),
                                        sig.shape[1]), "stage_epf")
        return out

    def stage_lf_smooth(self, params, lf):
        lf = [np.ascontiguousarray(a, dtype=np.float32) for a in lf]
        h, w = lf[0].shape
        out = [np.zeros_like(lf[0]) for _ in range(3)]
        pin = (C.c_void_p * 3)(*[a.ctypes.data for a in lf])
        pout = (C.c_void_p * 3)(*[a.ctypes.data for a in out])
        self._chk(self.L.jxlh_stage_lf_smooth(self._ctx, C.byref(params), pin, pout, w, h), "stage_lf_smooth")
        return out

    def set_upsampling_weights(self, w2=None, w4=None, w8=None):
        arrs = [None if w is None else np.ascontiguousarray(w, dtype=np.float32) for w in (w2, w4, w8)]
        for a, n in zip(arrs, (15, 55, 210)):
            assert a is None or a.size == n
        self._chk(self.L.jxlh_set_upsampling_weights(self._ctx, *[None if a is None else _addr(a) for a in arrs]),
                  "set_upsampling_weights")

    def stage_upsample(self, n, plane):
        plane = np.ascontiguousarray(plane, dtype=np.float32)
        h, w = plane.shape
        out = np.zeros((h * n, w * n), dtype=np.float32)
        self._chk(self.L.jxlh_stage_upsample(self._ctx, n, _addr(plane), _addr(out), w, h), "stage_upsample")
        return out

    def stage_noise_generate(self, visible, nonvisible, w, h):
        out = [np.zeros((h, w), dtype=np.float32) for _ in range(3)]
        ptrs = (C.c_void_p * 3)(*[a.ctypes.data for a in out])
        self._chk(self.L.jxlh_stage_noise_generate(self._ctx, visible, nonvisible, w, h, ptrs), "stage_noise_generate")
        return out

    def stage_noise_convolve(self, plane):
        plane = np.ascontiguousarray(plane, dtype=np.float32)
        h, w = plane.shape
        out = np.zeros_like(plane)
        self._chk(self.L.jxlh_stage_noise_convolve(self._ctx, _addr(plane), _addr(out), w, h), "stage_noise_convolve")
        return out

    def stage_noise_add(self, params, planes, rnd):
        pl = [np.ascontiguousarray(a, dtype=np.float32).copy() for a in planes]
        rn = [np.ascontiguousarray(a, dtype=np.float32) for a in rnd]
        pp = (C.c_void_p * 3)(*[a.ctypes.data for a in pl])
        pr = (C.c_void_p * 3)(*[a.ctypes.data for a in rn])
        self._chk(self.L.jxlh_stage_noise_add(self._ctx, C.byref(params), pp, pr, pl[0].size), "stage_noise_add")
        return pl

    def stage_chroma_upsample(self, plane, horizontal):
        plane = np.ascontiguousarray(plane, dtype=np.float32)
        h, w = plane.shape
        out = np.zeros((h, 2 * w) if horizontal else (2 * h, w), dtype=np.float32)
        self._chk(self.L.jxlh_stage_chroma_upsample(self._ctx, _addr(plane), _addr(out), w, h, 1 if horizontal else 0),
                  "stage_chroma_upsample")
        return out

    def stage_transform_to_pixels(self, ttype, coeffs, lf):
        """coeffs [n, cx*cy*64], lf [n, cx*cy] -> pixels [n, cy*8, cx*8]."""
        cx, cy = self.L.jxlh_covered_blocks_x(ttype), self.L.jxlh_covered_blocks_y(ttype)
        coeffs = np.ascontiguousarray(coeffs, dtype=np.float32).reshape(-1, cx * cy * 64)
        lf = np.ascontiguousarray(lf, dtype=np.float32).reshape(-1, cx * cy)
        n = coeffs.shape[0]
        assert lf.shape[0] == n
        out = np.zeros((n, cy * 8, cx * 8), dtype=np.float32)
        self._chk(self.L.jxlh_stage_transform_to_pixels(self._ctx, ttype, n, _addr(coeffs), _addr(lf), _addr(out)),
                  "stage_transform_to_pixels")
        return out

    # ---- modular ----
    def rct(self, planes, op, perm):
        ps = [np.ascontiguousarray(a, dtype=np.int32).copy() for a in planes]
        self._chk(self.L.jxlh_rct(self._ctx, _addr(ps[0]), _addr(ps[1]), _addr(ps[2]), ps[0].size, op, perm), "rct")
        return ps

    def palette(self, index, palette, num_colors, nb_channels, bit_depth):
        idx = np.ascontiguousarray(index, dtype=np.int32)
        pal = np.ascontiguousarray(palette, dtype=np.int32)
        out = np.zeros((nb_channels,) + idx.shape, dtype=np.int32)
        self._chk(self.L.jxlh_palette(self._ctx, _addr(idx), idx.size, _addr(pal), num_colors, pal.shape[1],
                                      nb_channels, bit_depth, _addr(out)), "palette")
        return out

    def modular_to_rgb8(self, planes, multiplier, maxv, channels=3):
        pl = [np.ascontiguousarray(a, dtype=np.int32) for a in planes]
        h, w = pl[0].shape
        out = np.zeros((h, w, channels), dtype=np.uint8)
        ptrs = (C.c_void_p * 3)(*[a.ctypes.data for a in pl])
        self._chk(self.L.jxlh_modular_to_rgb8(self._ctx, ptrs, w, w, h, multiplier, maxv, channels, _addr(out), w * channels),
                  "modular_to_rgb8")
        return out

    def modular_to_f32(self, plane, bits, exp_bits=0):
        a = np.ascontiguousarray(plane, dtype=np.int32)
        out = np.zeros(a.shape, dtype=np.float32)
        self._chk(self.L.jxlh_modular_to_f32(self._ctx, _addr(a), a.size, bits | exp_bits << 8, _addr(out)), "modular_to_f32")
        return out

    def modular_xyb_to_f32(self, y, x, b, quant_factors):
        y, x, b = [np.ascontiguousarray(v, dtype=np.int32) for v in (y, x, b)]
        q = np.ascontiguousarray(quant_factors, dtype=np.float32)
        out = [np.zeros(y.shape, dtype=np.float32) for _ in range(3)]
        self._chk(self.L.jxlh_modular_xyb_to_f32(self._ctx, _addr(y), _addr(x), _addr(b), y.size, _addr(q),
                                                 *[_addr(o) for o in out]), "modular_xyb_to_f32")
        return out

    def palette_delta(self, index, palette, num_colors, num_deltas, bit_depth, predictor):
        """index [h, w]; palette [nb_channels, >= num_colors + num_deltas] -> [nb_channels, h, w] (jxlh_palette_delta)"""
        idx = np.ascontiguousarray(index, dtype=np.int32)
        pal = np.ascontiguousarray(palette, dtype=np.int32)
        h, w = idx.shape
        out = np.zeros((pal.shape[0], h, w), dtype=np.int32)
        self._chk(self.L.jxlh_palette_delta(self._ctx, _addr(idx), w, h, _addr(pal), num_colors, num_deltas, pal.shape[1],
                                            pal.shape[0], bit_depth, predictor, _addr(out)), "palette_delta")
        return out

    def palette_delta_wp(self, index, palette, num_colors, num_deltas, bit_depth, wp_header):
        """The Weighted-predictor branch of the palette step; wp_header = (p1c, p2c, p3ca..p3ce, w0..w3)."""
        idx = np.ascontiguousarray(index, dtype=np.int32)
        pal = np.ascontiguousarray(palette, dtype=np.int32)
        hdr = np.ascontiguousarray(wp_header, dtype=np.uint32)
        assert hdr.size == 11
        h, w = idx.shape
        out = np.zeros((pal.shape[0], h, w), dtype=np.int32)
        self._chk(self.L.jxlh_palette_delta_wp(self._ctx, _addr(idx), w, h, _addr(pal), num_colors, num_deltas,
                                               pal.shape[1], pal.shape[0], bit_depth, _addr(hdr), _addr(out)),
                  "palette_delta_wp")
        return out

    def unsqueeze_planes(self, horizontal, avg, res, out, out_w, out_h, avg_stride, res_stride, out_stride):
        """Device-resident batched step: avg/res/out are lists (<= 3) of device pointers / tensors."""
        n = len(avg)
        av = (C.c_void_p * n)(*[_addr(a).value for a in avg])
        rv = (C.c_void_p * n)(*[_addr(a).value for a in res])
        ov = (C.c_void_p * n)(*[_addr(a).value for a in out])
        self._chk(self.L.jxlh_unsqueeze_planes(self._ctx, 1 if horizontal else 0, n, av, avg_stride, rv, res_stride,
                                               out_w, out_h, ov, out_stride), "unsqueeze_planes")

    def unsqueeze_levels(self, levels, base, base_stride, base_w, base_h, out, out_stride):
        """Several squeeze steps in one call (device-resident): levels = [(horizontal, out_w, out_h, [res planes],
        res_stride)], base / out = lists of n_planes device pointers."""
        n_planes = len(base)
        arr = (SqueezeLevel * len(levels))()
        for i, (hz, ow, oh, res, rstride) in enumerate(levels):
            arr[i].horizontal = 1 if hz else 0
            arr[i].out_w, arr[i].out_h = ow, oh
            for p in range(3):
                arr[i].res[p] = _addr(res[p]).value if p < n_planes and res[p] is not None else None
            arr[i].res_stride = rstride
        bv = (C.c_void_p * n_planes)(*[_addr(a).value for a in base])
        ov = (C.c_void_p * n_planes)(*[_addr(a).value for a in out])
        self._chk(self.L.jxlh_unsqueeze_levels(self._ctx, n_planes, len(levels), C.cast(arr, C.c_void_p), bv, base_stride,
                                               base_w, base_h, ov, out_stride), "unsqueeze_levels")

    def unsqueeze_chain(self, levels, base, base_stride, base_w, base_h, out, out_stride, rct=None):
        """The inverse of a whole squeeze transform (+ the RCT after it, rct = (op, perm)) in one call; arguments as
        unsqueeze_levels."""
        n_planes = len(base)
        arr = (SqueezeLevel * len(levels))()
        for i, (hz, ow, oh, res, rstride) in enumerate(levels):
            arr[i].horizontal = 1 if hz else 0
            arr[i].out_w, arr[i].out_h = ow, oh
            for p in range(3):
                arr[i].res[p] = _addr(res[p]).value if p < n_planes and res[p] is not None else None
            arr[i].res_stride = rstride
        bv = (C.c_void_p * n_planes)(*[_addr(a).value for a in base])
        ov = (C.c_void_p * n_planes)(*[_addr(a).value for a in out])
        op, perm = rct if rct is not None else (-1, 0)
        self._chk(self.L.jxlh_unsqueeze_chain(self._ctx, n_planes, len(levels), C.cast(arr, C.c_void_p), bv, base_stride,
                                              base_w, base_h, ov, out_stride, op, perm), "unsqueeze_chain")

    def unsqueeze_rct(self, horizontal, avg, res, out, out_w, out_h, avg_stride, res_stride, out_stride, op, perm):
        """Unsqueeze of three device-resident planes fused with the inverse RCT on them."""
        av = (C.c_void_p * 3)(*[_addr(a).value for a in avg])
        rv = (C.c_void_p * 3)(*[_addr(a).value for a in res])
        ov = (C.c_void_p * 3)(*[_addr(a).value for a in out])
        self._chk(self.L.jxlh_unsqueeze_rct(self._ctx, 1 if horizontal else 0, av, avg_stride, rv, res_stride, out_w,
                                            out_h, ov, out_stride, op, perm), "unsqueeze_rct")

    SMOOTH_H, SMOOTH_V, SMOOTH_2D = 0, 1, 2
    SMOOTH_CVT_NEAREST_EVEN = 0x100  # the x86 back-ends' as_i32 (cvtps2dq) instead of truncation

    def smooth_unsqueeze(self, kind, avg, out_w, out_h, x0=0, y0=0, out=None, cvt_rne=False):
        """smooth_{h,v,2d}_unsqueeze (squeeze.rs:908-1225): `avg` the whole average channel, host array."""
        kind |= self.SMOOTH_CVT_NEAREST_EVEN if cvt_rne else 0
        avg = np.ascontiguousarray(avg, dtype=np.int32)
        if out is None:
            out = np.zeros((out_h, out_w), dtype=np.int32)
        self._chk(self.L.jxlh_smooth_unsqueeze(self._ctx, kind, _addr(avg), avg.shape[1], avg.shape[1], avg.shape[0],
                                               x0, y0, _addr(out), out.shape[1], out_w, out_h), "smooth_unsqueeze")
        return out

    def smooth_unsqueeze_dev(self, kind, avg, avg_stride, avg_w, avg_h, out, out_stride, out_w, out_h, x0=0, y0=0):
        """Device-resident form (pointers / DeviceArray / tensors)."""
        self._chk(self.L.jxlh_smooth_unsqueeze(self._ctx, kind, _addr(avg), avg_stride, avg_w, avg_h, x0, y0,
                                               _addr(out), out_stride, out_w, out_h), "smooth_unsqueeze")

    def unsqueeze(self, horizontal, avg, res, out_w, out_h):
        avg = np.ascontiguousarray(avg, dtype=np.int32)
        res = np.ascontiguousarray(res, dtype=np.int32)
        out = np.zeros((out_h, out_w), dtype=np.int32)
        self._chk(self.L.jxlh_unsqueeze(self._ctx, 1 if horizontal else 0, _addr(avg), avg.shape[1],
                                        _addr(res) if res.size else None, max(res.shape[1], 1) if res.ndim == 2 else 1,
                                        out_w, out_h, _addr(out), out_w), "unsqueeze")
        return out
