//! headroom-hip -- thin binding of `libheadroom_hip.so` (`include/headroom_hip.h`) for M-Igashi/headroom.
//!
//! UNCOMPILED HERE: the build image has no Rust toolchain.  `tests/test_rust_crate.py` parses this file and the C
//! header and checks that the extern block lists every exported symbol with the header's argument count and that
//! every `#[repr(C)]` struct has the header's fields in the header's order.
//!
//! What it replaces in the reference: the body of `analyze_file_with_target` up to the gain decision
//! (`src/analyzer.rs:253-290`: spawn `ffmpeg -af loudnorm=print_format=json`, parse `input_i` / `input_tp`, reject
//! non-finite values) for a whole batch of decoded tracks, i.e. the work of `analyze_files`' `par_iter`
//! (`src/cli.rs:403-435`).  `analyze_batch` below returns the fields of `AudioAnalysis` (`src/analyzer.rs:52-65`) that
//! come from the measurement; path and file name stay with the caller.

use std::ffi::CStr;
use std::os::raw::{c_char, c_int, c_void};
use std::ptr;

/// Raw C ABI: a one-to-one mirror of `include/headroom_hip.h`.
#[allow(non_camel_case_types, dead_code)]
pub mod sys {
    use std::os::raw::{c_char, c_int, c_void};

    pub const HR_MAX_CH: usize = 8;
    pub const HR_MAX_SAMPLE_RATE: u32 = 768_000;
    pub const HR_NOALLOC_MAX_HOP_CHANNELS: u32 = 65_536;
    pub const HR_SHARD_AGAIN: c_int = 1;

    pub const HR_OK: c_int = 0;
    pub const HR_ERR_INVALID: c_int = -1;
    pub const HR_ERR_NO_DEVICE: c_int = -2;
    pub const HR_ERR_HIP: c_int = -3;
    pub const HR_ERR_OOM: c_int = -4;
    pub const HR_ERR_RATE: c_int = -5;
    pub const HR_ERR_FORMAT: c_int = -6;
    pub const HR_ERR_TOO_LARGE: c_int = -7;
    pub const HR_ERR_NOT_READY: c_int = -8;

    pub const HR_FMT_F32: u16 = 0;
    pub const HR_FMT_S16: u16 = 1;
    pub const HR_FMT_S24: u16 = 2;
    pub const HR_FMT_S32: u16 = 3;
    pub const HR_FMT_F64: u16 = 4;

    pub const HR_GATE_EBUR128: i32 = 0;
    pub const HR_GATE_EXACT: i32 = 1;
    pub const HR_GATE_LOUDNORM: i32 = 2;

    pub const HR_TP_SWR192K: i32 = 0;
    pub const HR_TP_BS1770: i32 = 1;
    pub const HR_TP_SAMPLE: i32 = 2;

    pub const HR_TRACK_DEVICE_PCM: u32 = 1;

    pub const HR_GAIN_FFMPEG_LOSSLESS: i32 = 0;
    pub const HR_GAIN_MP3_LOSSLESS: i32 = 1;
    pub const HR_GAIN_MP3_REENCODE: i32 = 2;
    pub const HR_GAIN_AAC_LOSSLESS: i32 = 3;
    pub const HR_GAIN_AAC_REENCODE: i32 = 4;
    pub const HR_GAIN_NONE: i32 = 5;

    pub const HR_KIND_LOSSLESS: c_int = 0;
    pub const HR_KIND_MP3: c_int = 1;
    pub const HR_KIND_AAC: c_int = 2;

    /// Opaque per-GPU context.
    #[repr(C)]
    pub struct hr_ctx {
        _private: [u8; 0],
    }

    #[repr(C)]
    #[derive(Clone, Copy)]
    pub struct hr_config {
        pub device: i32,
        pub gate_mode: i32,
        pub tp_mode: i32,
        pub engine: i32,
        pub stream: *mut c_void,
        pub stage_bytes: u64,
    }

    #[repr(C)]
    #[derive(Clone, Copy)]
    pub struct hr_track {
        pub pcm: *const c_void,
        pub frames: u64,
        pub sample_rate: u32,
        pub channels: u16,
        pub fmt: u16,
        pub ch_weight: *const f32,
        pub flags: u32,
        pub reserved: u32,
    }

    #[repr(C)]
    #[derive(Debug, Clone, Copy)]
    pub struct hr_result {
        pub integrated_lufs: f64,
        pub true_peak_dbtp: f64,
        pub rel_threshold_lufs: f64,
        pub n_blocks: u64,
        pub n_gated: u64,
        pub peak_per_ch: [f64; HR_MAX_CH],
        pub status: i32,
        pub reserved: i32,
        pub loudness_range_lu: f64,
        pub lra_low_lufs: f64,
        pub lra_high_lufs: f64,
        pub lra_threshold_lufs: f64,
        pub max_momentary_lufs: f64,
        pub max_shortterm_lufs: f64,
    }

    #[repr(C)]
    #[derive(Clone, Copy)]
    pub struct hr_shard_edge {
        pub rec: [[[f64; 9]; HR_MAX_CH]; 6],
        pub tail: [[f64; HR_MAX_CH]; 16],
        pub n_hops: u64,
        pub hop_tail: [[f64; HR_MAX_CH]; 29],
        pub range_ok: i32,
        pub reserved: i32,
    }

    #[repr(C)]
    #[derive(Clone, Copy)]
    pub struct hr_shard_partial {
        pub hist: [u64; 8001],
        pub sum_kept: f64,
        pub n_kept: u64,
        pub n_blocks: u64,
        pub peak_per_ch: [f64; HR_MAX_CH],
        pub st_hist: [u64; 8001],
        pub st_sum_kept: f64,
        pub st_n_kept: u64,
        pub n_shortterm: u64,
        pub max_m_power: f64,
        pub max_s_power: f64,
        pub range_ok: i32,
        pub reserved: i32,
        pub sum_gated: f64,
        pub n_gated: u64,
    }

    #[repr(C)]
    #[derive(Clone, Copy, Default)]
    pub struct hr_pcm_info {
        pub frames: u64,
        pub data_offset: u64,
        pub sample_rate: u32,
        pub channel_mask: u32,
        pub channels: u16,
        pub fmt: u16,
        pub big_endian: i32,
    }

    #[repr(C)]
    #[derive(Clone, Copy, Default)]
    pub struct hr_tp_target {
        pub split: i32,
        pub high: f64,
        pub low: f64,
    }

    #[repr(C)]
    #[derive(Clone, Copy, Default)]
    pub struct hr_analysis {
        pub input_i: f64,
        pub input_tp: f64,
        pub has_bitrate: i32,
        pub bitrate_kbps: u32,
        pub target_tp: f64,
        pub headroom: f64,
        pub gain_method: i32,
        pub effective_gain: f64,
        pub lossless_gain_steps: i32,
        pub status: i32,
    }

    extern "C" {
        pub fn hr_device_count() -> c_int;
        pub fn hr_init(cfg: *const hr_config, out: *mut *mut hr_ctx) -> c_int;
        pub fn hr_destroy(ctx: *mut hr_ctx);
        pub fn hr_alloc_pinned(bytes: usize) -> *mut c_void;
        pub fn hr_free_pinned(p: *mut c_void);
        pub fn hr_analyze_batch(ctx: *mut hr_ctx, tracks: *const hr_track, n: usize, out: *mut hr_result) -> c_int;
        pub fn hr_analyze_device_noalloc(ctx: *mut hr_ctx, track: *const hr_track, out: *mut hr_result) -> c_int;
        pub fn hr_block_energies(ctx: *mut hr_ctx, track: *const hr_track, hop_sums: *mut f64, cap_hops: u64,
                                 block_powers: *mut f64, cap_blocks: u64, n_hops: *mut u64, n_blocks: *mut u64) -> c_int;
        pub fn hr_shard_head(ctx: *mut hr_ctx, shard: *const hr_track, head: *mut f64) -> c_int;
        pub fn hr_shard_pass1(ctx: *mut hr_ctx, shard: *const hr_track, left_tail: *const f64, right_head: *const f64,
                              is_first: c_int, is_last: c_int, first_hop: u64, right_edge_out: *mut hr_shard_edge) -> c_int;
        pub fn hr_shard_pass2(ctx: *mut hr_ctx, left_edge: *const hr_shard_edge, partial_out: *mut hr_shard_partial) -> c_int;
        pub fn hr_shard_pass3(ctx: *mut hr_ctx, reduced: *const hr_shard_partial, partial_out: *mut hr_shard_partial) -> c_int;
        pub fn hr_shard_finish(ctx: *mut hr_ctx, reduced: *const hr_shard_partial, channels: c_int, out: *mut hr_result) -> c_int;
        // the same exchange with the records resident on the device (images of hr_shard_edge / hr_shard_partial in
        // device memory of the caller: what ncclSend / ncclRecv / ncclAllGather move); HR_SHARD_AGAIN from the finish asks
        // every rank for hr_shard_redo_device and one more gather
        pub fn hr_shard_pass1_device(ctx: *mut hr_ctx, shard: *const hr_track, left_tail: *const f64, right_head: *const f64,
                                     is_first: c_int, is_last: c_int, first_hop: u64, edge_dev_out: *mut c_void) -> c_int;
        pub fn hr_shard_pass2_device(ctx: *mut hr_ctx, left_edge_dev: *const c_void, partial_dev_out: *mut c_void) -> c_int;
        pub fn hr_shard_redo_device(ctx: *mut hr_ctx, partial_dev_inout: *mut c_void) -> c_int;
        pub fn hr_shard_finish_gathered(ctx: *mut hr_ctx, partials_dev: *const c_void, n: c_int, channels: c_int, out: *mut hr_result) -> c_int;
        pub fn hr_analyze_stream_sharded(ctxs: *const *mut hr_ctx, n_ctx: c_int, track: *const hr_track, out: *mut hr_result) -> c_int;
        pub fn hr_synth_fill(ctx: *mut hr_ctx, device_pcm: *mut c_void, seed: u64, sample_rate: u32, channels: u32, frames: u64) -> c_int;
        pub fn hr_last_timing(ctx: *mut hr_ctx, total_ms: *mut f32, main_kernel_ms: *mut f32) -> c_int;
        pub fn hr_strerror(code: c_int) -> *const c_char;
        pub fn hr_pcm_probe(file_bytes: *const c_void, n: usize, info: *mut hr_pcm_info) -> c_int;
        pub fn hr_pcm_load_file(path: *const c_char, pcm_out: *mut *mut c_void, track: *mut hr_track, weights: *mut f32) -> c_int;
        pub fn hr_gain_decision(input_i: f64, input_tp: f64, kind: c_int, has_bitrate: c_int, bitrate_kbps: u32,
                                mode: *const hr_tp_target, out: *mut hr_analysis) -> c_int;
        pub fn hr_loudnorm_text_values(r: *const hr_result, input_i: *mut f64, input_tp: *mut f64) -> c_int;
        pub fn hr_gain_decision_for_result(r: *const hr_result, kind: c_int, has_bitrate: c_int, bitrate_kbps: u32,
                                           mode: *const hr_tp_target, out: *mut hr_analysis) -> c_int;
        pub fn hr_loudnorm_first_pass(ctx: *mut hr_ctx, tracks: *const hr_track, n: usize, out: *mut hr_result) -> c_int;
        pub fn hr_apply_gain(ctx: *mut hr_ctx, track: *const hr_track, gain_db: f64, out_big_endian: c_int, out: *mut c_void,
                             clipped: *mut u64) -> c_int;
        pub fn hr_pcm_gain_file(ctx: *mut hr_ctx, path: *const c_char, gain_db: f64, clipped: *mut u64) -> c_int;
        pub fn hr_report_csv_header(buf: *mut c_char, cap: usize) -> c_int;
        pub fn hr_report_csv_row(filename: *const c_char, a: *const hr_analysis, buf: *mut c_char, cap: usize) -> c_int;
    }
}

/// An error code of the library with its text (`hr_strerror`).
#[derive(Debug, Clone, PartialEq, Eq)]
pub struct HrError {
    pub code: i32,
    pub message: String,
}

impl HrError {
    fn from_code(code: c_int) -> Self {
        // SAFETY: hr_strerror returns a pointer to a static NUL-terminated string for every code.
        let message = unsafe { CStr::from_ptr(sys::hr_strerror(code)) }.to_string_lossy().into_owned();
        HrError { code, message }
    }
}

impl std::fmt::Display for HrError {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        write!(f, "headroom_hip: {} ({})", self.message, self.code)
    }
}

impl std::error::Error for HrError {}

fn check(code: c_int) -> Result<(), HrError> {
    if code == sys::HR_OK {
        Ok(())
    } else {
        Err(HrError::from_code(code))
    }
}

/// `hr_config`, written out by hand (a raw pointer field has no `Default`).
#[derive(Debug, Clone, Copy)]
pub struct HrConfig {
    pub device: i32,
    pub gate_mode: i32,
    pub tp_mode: i32,
    pub engine: i32,
    pub stage_bytes: u64,
}

impl Default for HrConfig {
    fn default() -> Self {
        HrConfig { device: 0, gate_mode: sys::HR_GATE_EBUR128, tp_mode: sys::HR_TP_SWR192K, engine: 0, stage_bytes: 0 }
    }
}

impl HrConfig {
    fn to_sys(self) -> sys::hr_config {
        sys::hr_config {
            device: self.device,
            gate_mode: self.gate_mode,
            tp_mode: self.tp_mode,
            engine: self.engine,
            stream: ptr::null_mut(),
            stage_bytes: self.stage_bytes,
        }
    }
}

/// Interleaved PCM of one decoded audio stream (what ffmpeg's decoder hands the filter for `-map 0:a:0`).
#[derive(Debug, Clone, Copy)]
pub enum Pcm<'a> {
    F32(&'a [f32]),
    S16(&'a [i16]),
    /// packed 3-byte little endian
    S24(&'a [u8]),
    S32(&'a [i32]),
    F64(&'a [f64]),
}

#[derive(Debug, Clone, Copy)]
pub struct Track<'a> {
    pub pcm: Pcm<'a>,
    pub sample_rate: u32,
    pub channels: u16,
    /// `None`: FFmpeg's weights on the default layout of the channel count (LFE 0, back / side speakers 1.41)
    pub ch_weight: Option<&'a [f32]>,
}

impl<'a> Track<'a> {
    fn to_sys(&self) -> Result<sys::hr_track, HrError> {
        let ch = self.channels as usize;
        if ch == 0 || ch > sys::HR_MAX_CH {
            return Err(HrError::from_code(sys::HR_ERR_FORMAT));
        }
        let (ptr, samples, fmt) = match self.pcm {
            Pcm::F32(s) => (s.as_ptr() as *const c_void, s.len(), sys::HR_FMT_F32),
            Pcm::S16(s) => (s.as_ptr() as *const c_void, s.len(), sys::HR_FMT_S16),
            Pcm::S24(s) => (s.as_ptr() as *const c_void, s.len() / 3, sys::HR_FMT_S24),
            Pcm::S32(s) => (s.as_ptr() as *const c_void, s.len(), sys::HR_FMT_S32),
            Pcm::F64(s) => (s.as_ptr() as *const c_void, s.len(), sys::HR_FMT_F64),
        };
        if let Some(w) = self.ch_weight {
            if w.len() < ch {
                return Err(HrError::from_code(sys::HR_ERR_INVALID));
            }
        }
        Ok(sys::hr_track {
            pcm: ptr,
            frames: (samples / ch) as u64,
            sample_rate: self.sample_rate,
            channels: self.channels,
            fmt,
            ch_weight: self.ch_weight.map_or(ptr::null(), |w| w.as_ptr()),
            flags: 0,
            reserved: 0,
        })
    }
}

/// `GainMethod`, `src/analyzer.rs:36-50`.
#[derive(Debug, Clone, Copy, PartialEq, Eq)]
pub enum GainMethod {
    FfmpegLossless,
    Mp3Lossless,
    Mp3Reencode,
    AacLossless,
    AacReencode,
    None,
}

impl GainMethod {
    fn from_sys(v: i32) -> Self {
        match v {
            sys::HR_GAIN_FFMPEG_LOSSLESS => GainMethod::FfmpegLossless,
            sys::HR_GAIN_MP3_LOSSLESS => GainMethod::Mp3Lossless,
            sys::HR_GAIN_MP3_REENCODE => GainMethod::Mp3Reencode,
            sys::HR_GAIN_AAC_LOSSLESS => GainMethod::AacLossless,
            sys::HR_GAIN_AAC_REENCODE => GainMethod::AacReencode,
            _ => GainMethod::None,
        }
    }
}

/// `scanner::is_mp3` / `is_aac` (`src/analyzer.rs:292-294`).
#[derive(Debug, Clone, Copy, PartialEq, Eq)]
pub enum FileKind {
    Lossless,
    Mp3,
    Aac,
}

/// `TpTargetMode`, `src/analyzer.rs:157-164`.
#[derive(Debug, Clone, Copy, PartialEq)]
pub enum TpTargetMode {
    Uniform(f64),
    SplitBitrate { high: f64, low: f64 },
}

impl TpTargetMode {
    fn to_sys(self) -> sys::hr_tp_target {
        match self {
            TpTargetMode::Uniform(t) => sys::hr_tp_target { split: 0, high: t, low: t },
            TpTargetMode::SplitBitrate { high, low } => sys::hr_tp_target { split: 1, high, low },
        }
    }
}

/// The fields of `AudioAnalysis` (`src/analyzer.rs:52-65`) that the measurement and the decision produce; the caller
/// adds `filename` and `path`.  `input_i` / `input_tp` are the two-decimal values the reference parses out of
/// loudnorm's text (`src/analyzer.rs:273-280`); `exact` keeps the full-precision meter reading.
#[derive(Debug, Clone)]
pub struct AudioAnalysis {
    pub input_i: f64,
    pub input_tp: f64,
    pub bitrate_kbps: Option<u32>,
    pub target_tp: f64,
    pub headroom: f64,
    pub gain_method: GainMethod,
    pub effective_gain: f64,
    pub lossless_gain_steps: i32,
    pub exact: sys::hr_result,
}

/// What the caller knows about a file besides its samples (`src/analyzer.rs:292-302`).
#[derive(Debug, Clone, Copy)]
pub struct FileInfo {
    pub kind: FileKind,
    pub bitrate_kbps: Option<u32>,
}

/// A per-GPU context (`hr_init` / `hr_destroy`).  Calls on one context are serialised inside the library, so a
/// `&Meter` may be shared between rayon workers; batching the files of a `par_iter` into one call is what makes the
/// GPU worth having.
pub struct Meter {
    ctx: *mut sys::hr_ctx,
}

// SAFETY: hr_ctx is internally synchronised (a mutex per context); see include/headroom_hip.h.
unsafe impl Send for Meter {}
unsafe impl Sync for Meter {}

impl Meter {
    /// Replaces `check_ffmpeg()` (`src/analyzer.rs:348-354`): fails with `HR_ERR_NO_DEVICE` when there is no GPU.
    pub fn new(cfg: HrConfig) -> Result<Self, HrError> {
        let raw = cfg.to_sys();
        let mut ctx: *mut sys::hr_ctx = ptr::null_mut();
        // SAFETY: both pointers are valid for the call; the library fills `ctx` on success only.
        check(unsafe { sys::hr_init(&raw, &mut ctx) })?;
        Ok(Meter { ctx })
    }

    pub fn device_count() -> i32 {
        // SAFETY: no arguments.
        unsafe { sys::hr_device_count() }
    }

    /// Raw measurements of a batch, in input order; a bad track carries its own status and does not fail the batch
    /// (`src/cli.rs:419-430`).
    pub fn measure_batch(&self, tracks: &[Track<'_>]) -> Result<Vec<Result<sys::hr_result, HrError>>, HrError> {
        let mut raw = Vec::with_capacity(tracks.len());
        let mut bad: Vec<Option<HrError>> = Vec::with_capacity(tracks.len());
        for t in tracks {
            match t.to_sys() {
                Ok(r) => {
                    raw.push(r);
                    bad.push(None);
                }
                Err(e) => {
                    // keeps the indices aligned: an empty track that the library reports as silence, overridden below
                    raw.push(sys::hr_track {
                        pcm: ptr::null(),
                        frames: 0,
                        sample_rate: 48_000,
                        channels: 1,
                        fmt: sys::HR_FMT_F32,
                        ch_weight: ptr::null(),
                        flags: 0,
                        reserved: 0,
                    });
                    bad.push(Some(e));
                }
            }
        }
        // SAFETY: hr_result is plain data; the library overwrites every element.
        let mut out: Vec<sys::hr_result> = vec![unsafe { std::mem::zeroed() }; tracks.len()];
        // SAFETY: `raw` and `out` hold `tracks.len()` elements each and outlive the call; the PCM slices are borrowed
        // for the whole call through `tracks`.
        check(unsafe { sys::hr_analyze_batch(self.ctx, raw.as_ptr(), raw.len(), out.as_mut_ptr()) })?;
        Ok(out
            .into_iter()
            .zip(bad)
            .map(|(r, b)| match b {
                Some(e) => Err(e),
                None if r.status != sys::HR_OK => Err(HrError::from_code(r.status)),
                None => Ok(r),
            })
            .collect())
    }

    /// `analyze_file_with_target` for a batch of decoded tracks: measurement, loudnorm's two-decimal text rounding,
    /// the finite check and the gain decision (`src/analyzer.rs:253-346`).  `Err` per track corresponds to the `Err`
    /// arm of the reference (unreadable track, silence: "Non-finite loudness measurement").
    pub fn analyze_batch(&self, tracks: &[Track<'_>], info: &[FileInfo], tp_mode: TpTargetMode)
                         -> Result<Vec<Result<AudioAnalysis, HrError>>, HrError> {
        assert_eq!(tracks.len(), info.len());
        let mode = tp_mode.to_sys();
        let measured = self.measure_batch(tracks)?;
        Ok(measured
            .into_iter()
            .zip(info)
            .map(|(m, fi)| {
                let r = m?;
                let kind = match fi.kind {
                    FileKind::Lossless => sys::HR_KIND_LOSSLESS,
                    FileKind::Mp3 => sys::HR_KIND_MP3,
                    FileKind::Aac => sys::HR_KIND_AAC,
                };
                let mut a = sys::hr_analysis::default();
                // SAFETY: all pointers are valid for the call.
                check(unsafe {
                    sys::hr_gain_decision_for_result(&r, kind, fi.bitrate_kbps.is_some() as c_int, fi.bitrate_kbps.unwrap_or(0),
                                                     &mode, &mut a)
                })?;
                Ok(AudioAnalysis {
                    input_i: a.input_i,
                    input_tp: a.input_tp,
                    bitrate_kbps: if a.has_bitrate != 0 { Some(a.bitrate_kbps) } else { None },
                    target_tp: a.target_tp,
                    headroom: a.headroom,
                    gain_method: GainMethod::from_sys(a.gain_method),
                    effective_gain: a.effective_gain,
                    lossless_gain_steps: a.lossless_gain_steps,
                    exact: r,
                })
            })
            .collect())
    }

    /// In-place lossless gain of a WAV / AIFF file, as `apply_gain_ffmpeg` does (`src/processor.rs:53-87`).
    pub fn gain_file(&self, path: &std::path::Path, gain_db: f64) -> Result<u64, HrError> {
        let c = std::ffi::CString::new(path.to_string_lossy().as_bytes()).map_err(|_| HrError::from_code(sys::HR_ERR_INVALID))?;
        let mut clipped = 0u64;
        // SAFETY: `c` is NUL-terminated and outlives the call.
        check(unsafe { sys::hr_pcm_gain_file(self.ctx, c.as_ptr() as *const c_char, gain_db, &mut clipped) })?;
        Ok(clipped)
    }
}

impl Drop for Meter {
    fn drop(&mut self) {
        // SAFETY: ctx came from hr_init and is destroyed exactly once.
        unsafe { sys::hr_destroy(self.ctx) }
    }
}
