//! Safe host side of the MI355X JPEG XL reconstruction path for jxl-rs.
//!
//! Mirrors `include/jxl_hip.hpp` (the C++ rendering of the same surface, compiled and tested in this repository by
//! `tests/cpp/frame_parity.cc`): every method carries the name of the jxl-rs function whose work it takes over.
//! This crate depends on nothing but `jxl_hip_sys`; the three seams where the `jxl` crate calls it are listed in
//! INTEGRATION.md section 2.  NOT compiled in the build image of this repository (no Rust toolchain there).
use jxl_hip_sys as sys;
use std::ffi::CStr;
use std::os::raw::c_void;

/// `Error::Gpu(..)` payload for jxl-rs' `#[non_exhaustive] enum Error` (jxl/src/error.rs:15-17).
#[derive(Debug, Clone, PartialEq, Eq)]
pub enum HipError {
    InvalidArgument,
    OutOfMemory,
    Device(String),
    BadState,
    /// -> `Error::InvalidVarDCTTransform`
    InvalidTransform,
    /// a valid stream feature outside the device path: the caller keeps the CPU pipeline for this frame
    Unsupported,
    /// -> `Error::InvalidBlockSizeForChromaSubsampling`
    InvalidBlockSize,
    /// -> `Error::HFBlockOutOfBounds`
    BlockOutOfBounds,
    Unknown(i32),
}

pub type Result<T> = std::result::Result<T, HipError>;

fn check(ctx: *const sys::jxlh_ctx, st: sys::jxlh_status) -> Result<()> {
    match st {
        sys::JXLH_OK => Ok(()),
        sys::JXLH_ERR_INVALID_ARGUMENT => Err(HipError::InvalidArgument),
        sys::JXLH_ERR_OUT_OF_MEMORY => Err(HipError::OutOfMemory),
        sys::JXLH_ERR_DEVICE => {
            let msg = if ctx.is_null() {
                String::new()
            } else {
                // SAFETY: jxlh_last_error returns a NUL-terminated string owned by the context
                unsafe { CStr::from_ptr(sys::jxlh_last_error(ctx)) }.to_string_lossy().into_owned()
            };
            Err(HipError::Device(msg))
        }
        sys::JXLH_ERR_BAD_STATE => Err(HipError::BadState),
        sys::JXLH_ERR_INVALID_TRANSFORM => Err(HipError::InvalidTransform),
        sys::JXLH_ERR_UNSUPPORTED => Err(HipError::Unsupported),
        sys::JXLH_ERR_INVALID_BLOCK_SIZE => Err(HipError::InvalidBlockSize),
        sys::JXLH_ERR_BLOCK_OUT_OF_BOUNDS => Err(HipError::BlockOutOfBounds),
        other => Err(HipError::Unknown(other)),
    }
}

/// One device context (one HIP stream for the kernels + one upload stream per decoding thread).
pub struct Context {
    raw: *mut sys::jxlh_ctx,
}
// SAFETY: frame-level calls are made by one thread; jxlh_submit_group* is re-entrant per `slot`
// (include/jxl_hip.h "Conventions"), which is how `submit_*` below are used from the runner's threads.
unsafe impl Send for Context {}
unsafe impl Sync for Context {}

impl Context {
    /// `n_slots` = the JxlParallelRunner's thread count (jxl/src/api/mod.rs:77-81).
    pub fn new(device: i32, n_slots: i32) -> Result<Self> {
        let mut raw = std::ptr::null_mut();
        // SAFETY: out pointer is valid
        check(std::ptr::null(), unsafe { sys::jxlh_ctx_create(device, n_slots, &mut raw) })?;
        Ok(Self { raw })
    }
    pub fn raw(&self) -> *mut sys::jxlh_ctx {
        self.raw
    }
    fn ok(&self, st: sys::jxlh_status) -> Result<()> {
        check(self.raw, st)
    }
    pub fn default_params(xsize: u32, ysize: u32) -> sys::jxlh_frame_params {
        // SAFETY: plain-old-data struct, fully written by the call
        let mut p: sys::jxlh_frame_params = unsafe { std::mem::zeroed() };
        unsafe { sys::jxlh_default_frame_params(&mut p, xsize, ysize) };
        p
    }
    /// blocks until the kernels queued so far are done; reports device-side stream errors
    pub fn sync(&self) -> Result<()> {
        self.ok(unsafe { sys::jxlh_ctx_sync(self.raw) })
    }
    /// a point in the context's stream: everything enqueued so far (`jxlh_ctx_mark`)
    pub fn mark(&self) -> Result<u32> {
        let mut m = 0u32;
        self.ok(unsafe { sys::jxlh_ctx_mark(self.raw, &mut m) })?;
        Ok(m)
    }
    /// blocks until `mark` has been reached -- not for work enqueued after it: a decoder streaming frames through one
    /// context waits for frame i's mark after it has submitted and enqueued frame i + 1
    pub fn wait_mark(&self, mark: u32) -> Result<()> {
        self.ok(unsafe { sys::jxlh_ctx_wait_mark(self.raw, mark) })
    }
    /// Hand-over of device buffers the CALLER filled (jxl_hip.h "STREAM ORDERING OF DEVICE POINTERS"): the context's
    /// streams are non-blocking, so work queued on the NULL stream -- `hipMemset` / `hipMemcpy` of a plane that is then
    /// passed as a device pointer -- is ordered in front of the context's next calls with this.  Call it after the
    /// last fill and before the first entry point that takes the pointer; the host does not block.
    pub fn wait_default_stream(&self) -> Result<()> {
        self.ok(unsafe { sys::jxlh_ctx_wait_stream(self.raw, std::ptr::null_mut()) })
    }
    /// The same for a `hipStream_t` of the caller.
    /// # Safety
    /// `hip_stream` must be a live stream of the context's device.
    pub unsafe fn wait_stream(&self, hip_stream: *mut c_void) -> Result<()> {
        self.ok(sys::jxlh_ctx_wait_stream(self.raw, hip_stream))
    }
    /// ... and for a `hipEvent_t` the caller has recorded behind its fill.
    /// # Safety
    /// `hip_event` must be a live, recorded event.
    pub unsafe fn wait_event(&self, hip_event: *mut c_void) -> Result<()> {
        self.ok(sys::jxlh_ctx_wait_event(self.raw, hip_event))
    }
    /// The other direction: records the caller's event behind everything the context has enqueued on its main stream,
    /// so a caller stream can wait for the results on the device.
    /// # Safety
    /// `hip_event` must be a live event of the context's device.
    pub unsafe fn record_event(&self, hip_event: *mut c_void) -> Result<()> {
        self.ok(sys::jxlh_ctx_record_event(self.raw, hip_event))
    }
    /// pinned host memory for coefficient slabs / pair lists (replaces `VarDctBuffers::coeffs_storage`)
    /// `jxlh_ctx_tune_placement`: the context's next first allocation of its large buffers becomes a pick among `trials`
    /// candidate placements rated on the device (the driver's placement moves the transforms by up to 10 %; setup cost
    /// only, results unchanged).  `trials = 0` only queries.  Returns the last pick's ratings (ms of the two probe
    /// kernels per candidate) and the candidate taken (-1: none yet).
    pub fn tune_placement(&self, trials: i32) -> Result<(Vec<(f32, f32)>, i32)> {
        let mut rep = [0f32; 128];
        let (mut n, mut pick) = (0i32, -1i32);
        check(self.raw, unsafe { sys::jxlh_ctx_tune_placement(self.raw, trials, rep.as_mut_ptr(), rep.len() as i32, &mut n, &mut pick) })?;
        let k = (n.max(0) as usize).min(rep.len()) / 2;
        Ok(((0..k).map(|i| (rep[2 * i], rep[2 * i + 1])).collect(), pick))
    }
    pub fn alloc_pinned(&self, bytes: usize) -> Result<PinnedBuf<'_>> {
        let mut p: *mut c_void = std::ptr::null_mut();
        self.ok(unsafe { sys::jxlh_alloc_pinned(self.raw, bytes, &mut p) })?;
        Ok(PinnedBuf { ctx: self, ptr: p, len: bytes })
    }
}

impl Drop for Context {
    fn drop(&mut self) {
        // SAFETY: created by jxlh_ctx_create, destroyed once
        unsafe { sys::jxlh_ctx_destroy(self.raw) }
    }
}

pub struct PinnedBuf<'a> {
    ctx: &'a Context,
    ptr: *mut c_void,
    len: usize,
}
impl PinnedBuf<'_> {
    pub fn as_mut_slice<T: Copy>(&mut self) -> &mut [T] {
        // SAFETY: hipHostMalloc memory is suitably aligned for any scalar; len is the allocation size
        unsafe { std::slice::from_raw_parts_mut(self.ptr as *mut T, self.len / std::mem::size_of::<T>()) }
    }
    pub fn as_ptr(&self) -> *const c_void {
        self.ptr
    }
}
impl Drop for PinnedBuf<'_> {
    fn drop(&mut self) {
        unsafe { sys::jxlh_free_pinned(self.ctx.raw, self.ptr) };
    }
}

/// A VarDCT frame on the device.  Call order = the reference's frame flow.
pub struct VarDctFrame<'a> {
    ctx: &'a Context,
    pub xsize: u32,
    pub ysize: u32,
}

impl<'a> VarDctFrame<'a> {
    /// `Frame::from_header_and_toc` + `prepare_render_pipeline` (frame/decode.rs:172-204, frame/render.rs:907)
    pub fn begin(ctx: &'a Context, p: &sys::jxlh_frame_params) -> Result<Self> {
        ctx.ok(unsafe { sys::jxlh_frame_begin(ctx.raw, p) })?;
        Ok(Self { ctx, xsize: p.xsize, ysize: p.ysize })
    }
    /// `decode_hf_global`: `DequantMatrices::matrix(table, c)` for the 17 tables (frame/quant_weights.rs:1081-1086)
    pub fn decode_hf_global(&self, tables: &[&[f32]; 17]) -> Result<()> {
        let ptrs: Vec<*const f32> = tables.iter().map(|t| t.as_ptr()).collect();
        let n: Vec<usize> = tables.iter().map(|t| t.len() / 3).collect();
        self.ctx.ok(unsafe { sys::jxlh_frame_set_dequant_tables(self.ctx.raw, ptrs.as_ptr(), n.as_ptr()) })
    }
    /// `decode_lf_group` -> `dequant_lf` (frame/modular/mod.rs:837-929): rect in blocks, channels in coded order Y, X, B
    #[allow(clippy::too_many_arguments)]
    pub fn decode_lf_group(&self, x0: u32, y0: u32, w: u32, h: u32, qy: &[i32], qx: &[i32], qb: &[i32], stride: usize,
                           extra_precision: u32) -> Result<()> {
        if w == 0 || h == 0 || stride < w as usize {
            return Err(HipError::InvalidArgument);
        }
        let need = (h as usize - 1) * stride + w as usize;
        if qy.len() < need || qx.len() < need || qb.len() < need {
            return Err(HipError::InvalidArgument);
        }
        self.ctx.ok(unsafe {
            sys::jxlh_frame_set_lf_quantized(self.ctx.raw, x0, y0, w, h, qy.as_ptr(), qx.as_ptr(), qb.as_ptr(), stride,
                                             extra_precision)
        })
    }
    /// `decode_hf_metadata` (frame/modular/mod.rs:984-1081): HfMetadata maps of a rect in blocks
    #[allow(clippy::too_many_arguments)]
    pub fn decode_hf_metadata(&self, x0: u32, y0: u32, w: u32, h: u32, transform_map: &[u8], raw_quant: &[i32],
                              epf_map: &[u8], map_stride: usize, ytox: &[i8], ytob: &[i8], cmap_stride: usize) -> Result<()> {
        // the C side reads h rows of w entries at map_stride, and one colour-tile entry per 8 x 8 blocks
        if w == 0 || h == 0 || map_stride < w as usize || cmap_stride < (w as usize).div_ceil(8) {
            return Err(HipError::InvalidArgument);
        }
        let need = (h as usize - 1) * map_stride + w as usize;
        let cneed = ((h as usize).div_ceil(8) - 1) * cmap_stride + (w as usize).div_ceil(8);
        if transform_map.len() < need || raw_quant.len() < need || epf_map.len() < need || ytox.len() < cneed
            || ytob.len() < cneed {
            return Err(HipError::InvalidArgument);
        }
        self.ctx.ok(unsafe {
            sys::jxlh_frame_set_hf_meta(self.ctx.raw, x0, y0, w, h, transform_map.as_ptr(), raw_quant.as_ptr(),
                                        epf_map.as_ptr(), map_stride, ytox.as_ptr(), ytob.as_ptr(), cmap_stride)
        })
    }
    /// the `if let Some(pixels)` branch of `decode_vardct_group` (frame/group.rs:579-611): the group's dense slab,
    /// 3 x 65536 i32 in pinned memory; asynchronous, reuse the slab after `slot_wait`.  `complete = false` is a
    /// progressive pass whose coefficients will be added to by later passes (`set_buffer_for_group(.., complete, ..)`).
    ///
    /// # Safety
    /// The copy is asynchronous: `coeffs` must stay alive and unmodified until `slot_wait(slot)` has returned (the
    /// borrow this call holds ends when it returns, the device still reads the slab).
    pub unsafe fn decode_vardct_group(&self, slot: i32, group: u32, coeffs: &[i32], complete: bool) -> Result<()> {
        if coeffs.len() != 3 * 65536 {
            return Err(HipError::InvalidArgument);
        }
        let flags = if complete { sys::JXLH_GROUP_COMPLETE } else { 0 };
        self.ctx.ok(sys::jxlh_submit_group(self.ctx.raw, slot, group, coeffs.as_ptr(), flags))
    }
    /// the same from the entropy loop's updates (frame/group.rs:557-572 emits `(position, value)` instead of
    /// `coeffs[position] += value`): `pairs` = X run, Y run, B run
    ///
    /// # Safety
    /// Asynchronous like `decode_vardct_group`: `pairs` and `wide` must stay alive and unmodified until
    /// `slot_wait(slot)` has returned.
    pub unsafe fn decode_vardct_group_sparse(&self, slot: i32, group: u32, pairs: &[sys::jxlh_coeff16], n: [u32; 3],
                                      wide: &[sys::jxlh_coeff32], complete: bool) -> Result<()> {
        if pairs.len() != (n[0] + n[1] + n[2]) as usize {
            return Err(HipError::InvalidArgument);
        }
        let flags = if complete { sys::JXLH_GROUP_COMPLETE } else { 0 };
        self.ctx.ok(sys::jxlh_submit_group_sparse(self.ctx.raw, slot, group, pairs.as_ptr(), n.as_ptr(), wide.as_ptr(),
                                                  wide.len() as u32, flags))
    }
    pub fn slot_wait(&self, slot: i32) -> Result<()> {
        self.ctx.ok(unsafe { sys::jxlh_slot_wait(self.ctx.raw, slot) })
    }
    /// `Frame::finalize_lf` + `SigmaSource::new` + the reconstruction of every submitted group + the Gaborish / EPF
    /// stages of frame/render.rs:569-622 (and chroma upsampling / upsampling / noise when the parameters ask for them)
    pub fn finalize_and_render(&self) -> Result<()> {
        self.ctx.ok(unsafe { sys::jxlh_frame_run(self.ctx.raw, 0, u32::MAX) })
    }
    /// `mark_group_to_rerender` + re-render (render/mod.rs:143-146): after more passes arrived for `groups`
    pub fn rerender_groups(&self, groups: &[u32]) -> Result<()> {
        self.ctx.ok(unsafe { sys::jxlh_frame_rerender_groups(self.ctx.raw, groups.as_ptr(), groups.len() as u32) })
    }
    /// the save stage for planar f32 XYB output; `out[c]` = `RawImageBuffer` of channel c
    ///
    /// # Safety
    /// Every `out[c]` must describe writable memory (host or device) of `num_rows` rows of `bytes_per_row` bytes at
    /// `bytes_between_rows`: the descriptors carry raw pointers the compiler cannot check.
    pub unsafe fn read_planes(&self, out: &[sys::jxlh_plane; 3]) -> Result<()> {
        self.ctx.ok(sys::jxlh_frame_read_planes(self.ctx.raw, out.as_ptr()))
    }
    /// One 256 x 256 group of the finished planes, the unit `RenderPipeline::set_buffer_for_group` moves
    /// (render/mod.rs:124-137): `out[c]` = the `RawImageBuffer` of the `Image<f32>` `pipeline.get_buffer::<f32>(c)`
    /// returned (the group's size rounded up to 16 pixels, render/internal.rs:144-167).  `xgroups` =
    /// `frame_header.size_groups().0`.  Pixels beyond the frame's edge are left untouched.
    ///
    /// # Safety
    /// See `read_planes`.
    pub unsafe fn read_group_planes(&self, group: u32, xgroups: u32, out: &[sys::jxlh_plane; 3]) -> Result<()> {
        if xgroups == 0 {
            return Err(HipError::InvalidArgument);
        }
        let (x0, y0) = ((group % xgroups) * sys::JXLH_GROUP_DIM, (group / xgroups) * sys::JXLH_GROUP_DIM);
        self.ctx.ok(sys::jxlh_frame_read_planes_rect(self.ctx.raw, x0, y0, sys::JXLH_GROUP_DIM, sys::JXLH_GROUP_DIM,
                                                     out.as_ptr()))
    }
    /// An extra channel as the Modular decoder leaves it (channel 3 + `ec` of the reference's pipeline): `samples` = `h`
    /// rows of `w` i32 at `stride`.  The next `finalize_and_render` applies `ConvertModularToF32Stage::new(3 + ec,
    /// bits_per_sample)` and, for `ec_upsampling` 2 / 4 / 8, the channel's own `Upsample{2,4,8}x` (frame/render.rs:564-567,
    /// :624-637).  The samples are copied before the call returns.
    pub fn set_extra_channel(&self, ec: u32, samples: &[i32], stride: usize, w: u32, h: u32, bits_per_sample: u32,
                             ec_upsampling: u32) -> Result<()> {
        if w == 0 || h == 0 || stride < w as usize || samples.len() < (h as usize - 1) * stride + w as usize {
            return Err(HipError::InvalidArgument);
        }
        self.ctx.ok(unsafe {
            sys::jxlh_frame_set_extra_channel(self.ctx.raw, ec, samples.as_ptr(), stride, w, h, bits_per_sample, ec_upsampling)
        })
    }
    /// the save stage of an extra channel: one f32 plane of the frame's output size
    ///
    /// # Safety
    /// `out` must describe writable memory (host or device), see `read_planes`.
    pub unsafe fn read_extra_channel(&self, ec: u32, out: &sys::jxlh_plane) -> Result<()> {
        self.ctx.ok(sys::jxlh_frame_read_extra_channel(self.ctx.raw, ec, out))
    }
    /// XybStage + FromLinearStage(sRGB) + ConvertF32ToU8Stage, interleaved, rows [y0, y1)
    pub fn read_rgb8(&self, p: &sys::jxlh_xyb_params, channels: u32, y0: u32, y1: u32, out: &mut [u8],
                     bytes_per_row: usize) -> Result<()> {
        let rows = y1.checked_sub(y0).ok_or(HipError::InvalidArgument)? as usize;
        if out.len() < rows * bytes_per_row {
            return Err(HipError::InvalidArgument);
        }
        self.ctx.ok(unsafe {
            sys::jxlh_frame_read_rgb8(self.ctx.raw, p, channels, y0, y1, out.as_mut_ptr() as *mut c_void, bytes_per_row)
        })
    }
}

/// The slot-bucketed coefficient form written by the entropy loop itself (`jxlh_slot_writer_*`, include/jxl_hip.h):
/// `current_coeffs[coeff_index] += coeff` of `decode_vardct_group` (jxl/src/frame/group.rs:557-575) becomes `add`, a
/// varblock's start (`coeffs_offset`, `cx * cy`: group.rs:612) `begin_varblock`.  One writer per runner thread.  Values
/// beyond the entry's range are split into repeated in-range entries (they add up on the device); only what no slot can
/// hold lands in `wide`.  Plain CPU code: no context, no device.
pub struct SlotWriter {
    raw: *mut sys::jxlh_slot_writer,
}
// SAFETY: a writer is used by one thread at a time (it holds no thread-affine state)
unsafe impl Send for SlotWriter {}

impl SlotWriter {
    pub fn new() -> Result<Self> {
        let mut raw = std::ptr::null_mut();
        check(std::ptr::null(), unsafe { sys::jxlh_slot_writer_create(&mut raw) })?;
        Ok(Self { raw })
    }
    /// `entries` (u16 each; room for the group's updates incl. split values), `slot_counts` (3 x 1024, zeroed by the
    /// call) and `wide` are the buffers `jxlh_submit_groups_slots` will read -- typically slices of pinned memory.
    pub fn begin_group(&mut self, group_id: u32, entries: &mut [u16], slot_counts: &mut [u8; 3 * 1024],
                       wide: &mut [sys::jxlh_coeff32]) -> Result<()> {
        check(std::ptr::null(), unsafe {
            sys::jxlh_slot_writer_begin_group(self.raw, group_id, 0, entries.as_mut_ptr() as *mut c_void, entries.len(),
                                              slot_counts.as_mut_ptr(), wide.as_mut_ptr(), wide.len() as u32)
        })
    }
    /// `first_slot` = `coeffs_offset / 64`, `num_slots` = `cx * cy`; varblocks in decode order
    pub fn begin_varblock(&mut self, first_slot: u32, num_slots: u32) -> Result<()> {
        check(std::ptr::null(), unsafe { sys::jxlh_slot_writer_begin_varblock(self.raw, first_slot, num_slots) })
    }
    /// `coeffs[channel][coeffs_offset + pos] += value`
    #[inline]
    pub fn add(&mut self, channel: u32, pos: u32, value: i32) -> Result<()> {
        check(std::ptr::null(), unsafe { sys::jxlh_slot_writer_add(self.raw, channel, pos, value) })
    }
    /// closes the group: entries per channel (the `n` of `jxlh_submit_groups_slots`) and the number of `wide` values
    pub fn end_group(&mut self) -> Result<([u32; 3], u32)> {
        let mut n = [0u32; 3];
        let mut nw = 0u32;
        check(std::ptr::null(), unsafe { sys::jxlh_slot_writer_end_group(self.raw, n.as_mut_ptr(), &mut nw) })?;
        Ok((n, nw))
    }
}

impl Drop for SlotWriter {
    fn drop(&mut self) {
        // SAFETY: created by jxlh_slot_writer_create, destroyed once
        unsafe { sys::jxlh_slot_writer_destroy(self.raw) }
    }
}

/// Dense group slabs -> the arrays of ONE `jxlh_submit_groups_slots` call (`jxlh_host_pack_slots_many`): for a decoder
/// that keeps the reference's per-group `Vec<i32>` (`Frame::hf_coefficients`, jxl/src/frame/mod.rs; 3 x 65536 each).
/// `entries` (u16 each), `slot_counts` (groups x 3 x 1024) and `n` (groups x 3) are written group after group.  Returns
/// (entries written, `wide` values).  Plain CPU code, any thread: each runner thread packs its share of the frame.
pub fn pack_group_slabs(group_coeffs: &[&[i32]], group_ids: &[u32], entries: &mut [u16], slot_counts: &mut [u8],
                        n: &mut [u32], wide: &mut [sys::jxlh_coeff32]) -> Result<(usize, u32)> {
    let k = group_coeffs.len();
    if group_ids.len() != k || slot_counts.len() < k * 3 * 1024 || n.len() < k * 3 || group_coeffs.iter().any(|g| g.len() != 3 * 65536) {
        return Err(HipError::InvalidArgument);
    }
    let ptrs: Vec<*const i32> = group_coeffs.iter().map(|g| g.as_ptr()).collect();
    let (mut nw, mut used) = (0u32, 0usize);
    check(std::ptr::null(), unsafe {
        sys::jxlh_host_pack_slots_many(ptrs.as_ptr(), group_ids.as_ptr(), k as u32, 0, entries.as_mut_ptr() as *mut c_void,
                                       entries.len(), slot_counts.as_mut_ptr(), n.as_mut_ptr(), wide.as_mut_ptr(),
                                       wide.len() as u32, &mut nw, &mut used)
    })?;
    Ok((used, nw))
}
