"""headroom_amd -- Python harness over the C ABI of the MI355X loudness / true-peak meter.

The product is `libheadroom_hip.so` (HIP kernels + C ABI, include/headroom_hip.h).  This module only
binds it with ctypes for the tests, the bench and `torch.distributed` multi-GPU drivers; it contains
no arithmetic of its own and no fallback: importing it without the built library raises.
"""
import ctypes as C
import math
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# HEADROOM_HIP_LIB: another build of the same library (experiments, packaging)
LIB_PATH = os.environ.get("HEADROOM_HIP_LIB") or os.path.join(_HERE, "libheadroom_hip.so")

MAX_CH = 8
FMT_F32, FMT_S16, FMT_S24, FMT_S32, FMT_F64 = range(5)
GATE_EBUR128, GATE_EXACT, GATE_LOUDNORM = range(3)
TP_SWR192K, TP_BS1770, TP_SAMPLE = range(3)
TRACK_DEVICE_PCM = 1
OK, ERR_INVALID, ERR_NO_DEVICE, ERR_HIP, ERR_OOM, ERR_RATE, ERR_FORMAT, ERR_TOO_LARGE, ERR_NOT_READY = 0, -1, -2, -3, -4, -5, -6, -7, -8
NOALLOC_MAX_HOP_CHANNELS = 65536
SHARD_AGAIN = 1
GAIN_FFMPEG_LOSSLESS, GAIN_MP3_LOSSLESS, GAIN_MP3_REENCODE, GAIN_AAC_LOSSLESS, GAIN_AAC_REENCODE, GAIN_NONE = range(6)
KIND_LOSSLESS, KIND_MP3, KIND_AAC = range(3)
_FMT_BYTES = {FMT_F32: 4, FMT_S16: 2, FMT_S24: 3, FMT_S32: 4, FMT_F64: 8}
_NP_FMT = {np.dtype("float32"): FMT_F32, np.dtype("int16"): FMT_S16, np.dtype("int32"): FMT_S32,
           np.dtype("float64"): FMT_F64}


class HrConfig(C.Structure):
    _fields_ = [("device", C.c_int32), ("gate_mode", C.c_int32), ("tp_mode", C.c_int32), ("engine", C.c_int32),
                ("stream", C.c_void_p), ("stage_bytes", C.c_uint64)]


class HrTrack(C.Structure):
    _fields_ = [("pcm", C.c_void_p), ("frames", C.c_uint64), ("sample_rate", C.c_uint32), ("channels", C.c_uint16),
                ("fmt", C.c_uint16), ("ch_weight", C.POINTER(C.c_float)), ("flags", C.c_uint32), ("reserved", C.c_uint32)]


class HrResult(C.Structure):
    _fields_ = [("integrated_lufs", C.c_double), ("true_peak_dbtp", C.c_double), ("rel_threshold_lufs", C.c_double),
                ("n_blocks", C.c_uint64), ("n_gated", C.c_uint64), ("peak_per_ch", C.c_double * MAX_CH),
                ("status", C.c_int32), ("reserved", C.c_int32),
                ("loudness_range_lu", C.c_double), ("lra_low_lufs", C.c_double), ("lra_high_lufs", C.c_double),
                ("lra_threshold_lufs", C.c_double), ("max_momentary_lufs", C.c_double), ("max_shortterm_lufs", C.c_double)]


class HrShardEdge(C.Structure):
    _fields_ = [("rec", C.c_double * 9 * MAX_CH * 6), ("tail", C.c_double * MAX_CH * 16), ("n_hops", C.c_uint64),
                ("hop_tail", C.c_double * MAX_CH * 29), ("range_ok", C.c_int32), ("reserved", C.c_int32)]


class HrShardPartial(C.Structure):
    _fields_ = [("hist", C.c_uint64 * 8001), ("sum_kept", C.c_double), ("n_kept", C.c_uint64), ("n_blocks", C.c_uint64),
                ("peak_per_ch", C.c_double * MAX_CH),
                ("st_hist", C.c_uint64 * 8001), ("st_sum_kept", C.c_double), ("st_n_kept", C.c_uint64),
                ("n_shortterm", C.c_uint64), ("max_m_power", C.c_double), ("max_s_power", C.c_double),
                ("range_ok", C.c_int32), ("reserved", C.c_int32), ("sum_gated", C.c_double), ("n_gated", C.c_uint64)]


class HrTpTarget(C.Structure):
    _fields_ = [("split", C.c_int32), ("high", C.c_double), ("low", C.c_double)]


class HrAnalysis(C.Structure):
    _fields_ = [("input_i", C.c_double), ("input_tp", C.c_double), ("has_bitrate", C.c_int32), ("bitrate_kbps", C.c_uint32),
                ("target_tp", C.c_double), ("headroom", C.c_double), ("gain_method", C.c_int32),
                ("effective_gain", C.c_double), ("lossless_gain_steps", C.c_int32), ("status", C.c_int32)]


class HrPcmInfo(C.Structure):
    _fields_ = [("frames", C.c_uint64), ("data_offset", C.c_uint64), ("sample_rate", C.c_uint32), ("channel_mask", C.c_uint32),
                ("channels", C.c_uint16), ("fmt", C.c_uint16), ("big_endian", C.c_int32)]


EXPORTS = ["hr_device_count", "hr_init", "hr_destroy", "hr_alloc_pinned", "hr_free_pinned", "hr_analyze_batch",
           "hr_block_energies", "hr_analyze_stream_sharded", "hr_shard_head", "hr_shard_pass1", "hr_shard_pass2", "hr_shard_pass3", "hr_shard_finish",
           "hr_synth_fill", "hr_last_timing", "hr_strerror", "hr_gain_decision", "hr_pcm_probe", "hr_pcm_load_file",
           "hr_report_csv_header", "hr_report_csv_row", "hr_apply_gain", "hr_loudnorm_first_pass", "hr_pcm_gain_file",
           "hr_analyze_device_noalloc", "hr_loudnorm_text_values", "hr_gain_decision_for_result",
           "hr_shard_pass1_device", "hr_shard_pass2_device", "hr_shard_redo_device", "hr_shard_finish_gathered"]

_lib = None


def lib():
    """Loads libheadroom_hip.so; raises if it has not been built (there is no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(make -C headroom_amd/csrc). The meter has no CPU fallback.")
    # PyTorch's ROCm wheel bundles its own HIP runtime.  One process must use one runtime, and torch
    # cannot start on a runtime somebody else loaded first ("No HIP GPUs are available"), so when torch
    # is installed let it load its runtime before this library binds to libamdhip64 by SONAME.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    L.hr_device_count.restype = C.c_int
    L.hr_init.argtypes = [C.POINTER(HrConfig), C.POINTER(C.c_void_p)]
    L.hr_destroy.argtypes = [C.c_void_p]
    L.hr_destroy.restype = None
    L.hr_alloc_pinned.argtypes = [C.c_size_t]
    L.hr_alloc_pinned.restype = C.c_void_p
    L.hr_free_pinned.argtypes = [C.c_void_p]
    L.hr_free_pinned.restype = None
    L.hr_analyze_batch.argtypes = [C.c_void_p, C.POINTER(HrTrack), C.c_size_t, C.POINTER(HrResult)]
    L.hr_analyze_device_noalloc.argtypes = [C.c_void_p, C.POINTER(HrTrack), C.POINTER(HrResult)]
    L.hr_block_energies.argtypes = [C.c_void_p, C.POINTER(HrTrack), C.POINTER(C.c_double), C.c_uint64,
                                    C.POINTER(C.c_double), C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.hr_shard_head.argtypes = [C.c_void_p, C.POINTER(HrTrack), C.POINTER(C.c_double)]
    L.hr_shard_pass1.argtypes = [C.c_void_p, C.POINTER(HrTrack), C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int,
                                 C.c_int, C.c_uint64, C.POINTER(HrShardEdge)]
    L.hr_shard_pass2.argtypes = [C.c_void_p, C.POINTER(HrShardEdge), C.POINTER(HrShardPartial)]
    L.hr_shard_pass3.argtypes = [C.c_void_p, C.POINTER(HrShardPartial), C.POINTER(HrShardPartial)]
    L.hr_shard_finish.argtypes = [C.c_void_p, C.POINTER(HrShardPartial), C.c_int, C.POINTER(HrResult)]
    L.hr_analyze_stream_sharded.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(HrTrack), C.POINTER(HrResult)]
    L.hr_shard_pass1_device.argtypes = [C.c_void_p, C.POINTER(HrTrack), C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int,
                                        C.c_int, C.c_uint64, C.c_void_p]
    L.hr_shard_pass2_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.hr_shard_redo_device.argtypes = [C.c_void_p, C.c_void_p]
    L.hr_shard_finish_gathered.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(HrResult)]
    L.hr_pcm_probe.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(HrPcmInfo)]
    L.hr_pcm_load_file.argtypes = [C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(HrTrack), C.POINTER(C.c_float)]
    L.hr_report_csv_header.argtypes = [C.c_char_p, C.c_size_t]
    L.hr_report_csv_row.argtypes = [C.c_char_p, C.POINTER(HrAnalysis), C.c_char_p, C.c_size_t]
    L.hr_loudnorm_first_pass.argtypes = [C.c_void_p, C.POINTER(HrTrack), C.c_size_t, C.POINTER(HrResult)]
    L.hr_pcm_gain_file.argtypes = [C.c_void_p, C.c_char_p, C.c_double, C.POINTER(C.c_uint64)]
    L.hr_apply_gain.argtypes = [C.c_void_p, C.POINTER(HrTrack), C.c_double, C.c_int, C.c_void_p, C.POINTER(C.c_uint64)]
    L.hr_synth_fill.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint64]
    L.hr_last_timing.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.hr_strerror.argtypes = [C.c_int]
    L.hr_strerror.restype = C.c_char_p
    L.hr_gain_decision.argtypes = [C.c_double, C.c_double, C.c_int, C.c_int, C.c_uint32, C.POINTER(HrTpTarget),
                                   C.POINTER(HrAnalysis)]
    L.hr_loudnorm_text_values.argtypes = [C.POINTER(HrResult), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.hr_gain_decision_for_result.argtypes = [C.POINTER(HrResult), C.c_int, C.c_int, C.c_uint32, C.POINTER(HrTpTarget),
                                              C.POINTER(HrAnalysis)]
    _lib = L
    return L


class HeadroomError(RuntimeError):
    def __init__(self, code, what=""):
        self.code = code
        super().__init__(f"{what}: {lib().hr_strerror(code).decode()} ({code})")


def _check(code, what):
    if code != OK:
        raise HeadroomError(code, what)


def device_count():
    return lib().hr_device_count()


def _result_dict(r, channels):
    return {"integrated_lufs": r.integrated_lufs, "true_peak_dbtp": r.true_peak_dbtp,
            "rel_threshold_lufs": r.rel_threshold_lufs, "n_blocks": int(r.n_blocks), "n_gated": int(r.n_gated),
            "peak_per_ch": [r.peak_per_ch[i] for i in range(channels)], "status": int(r.status),
            "loudness_range_lu": r.loudness_range_lu, "lra_low_lufs": r.lra_low_lufs, "lra_high_lufs": r.lra_high_lufs,
            "lra_threshold_lufs": r.lra_threshold_lufs, "max_momentary_lufs": r.max_momentary_lufs,
            "max_shortterm_lufs": r.max_shortterm_lufs}


class TrackSpec:
    """One PCM stream for the C ABI: numpy array (host), or a raw device pointer / torch CUDA tensor."""

    def __init__(self, pcm, sample_rate, channels=None, fmt=None, weights=None, frames=None):
        self.keep = []
        flags = 0
        if isinstance(pcm, np.ndarray):
            if fmt == FMT_S24:
                buf = np.ascontiguousarray(pcm, dtype=np.uint8)
                frames = buf.size // (3 * channels)
            else:
                buf = np.ascontiguousarray(pcm)
                if buf.ndim == 1:
                    buf = buf[:, None]
                frames, ch = buf.shape
                channels = channels or ch
                fmt = _NP_FMT[buf.dtype] if fmt is None else fmt
            ptr = buf.ctypes.data
            self.keep.append(buf)
        elif hasattr(pcm, "data_ptr"):  # torch tensor
            t = pcm
            if t.is_cuda:
                flags |= TRACK_DEVICE_PCM
            assert t.is_contiguous()
            if fmt is None:
                import torch
                fmt = {torch.float32: FMT_F32, torch.int16: FMT_S16, torch.int32: FMT_S32, torch.float64: FMT_F64}[t.dtype]
            if fmt == FMT_S24:
                frames = t.numel() // (3 * channels)
            else:
                if t.dim() == 1:
                    channels = channels or 1
                    frames = t.numel() // channels
                else:
                    frames, ch = t.shape
                    channels = channels or ch
            ptr = t.data_ptr()
            self.keep.append(t)
        else:  # (device pointer int, explicit geometry)
            ptr = int(pcm)
            flags |= TRACK_DEVICE_PCM
            assert frames is not None and channels and fmt is not None
        wptr = C.POINTER(C.c_float)()
        if weights is not None:
            w = np.ascontiguousarray(weights, dtype=np.float32)
            self.keep.append(w)
            wptr = w.ctypes.data_as(C.POINTER(C.c_float))
        self.channels = channels
        self.c = HrTrack(ptr, frames, sample_rate, channels, fmt, wptr, flags, 0)


class TrackBatch:
    """An hr_track array and its hr_result array, marshalled once (Meter.analyze_prepared)."""

    def __init__(self, specs):
        self.specs = list(specs)  # keeps the PCM buffers alive
        self.n = len(self.specs)
        self.tracks = (HrTrack * max(self.n, 1))(*[s.c for s in self.specs])
        self.out = (HrResult * max(self.n, 1))()

    def statuses(self):
        return [int(self.out[i].status) for i in range(self.n)]

    def results(self):
        return [_result_dict(self.out[i], self.specs[i].channels or 0) for i in range(self.n)]


class Meter:
    """A per-GPU context of the meter (hr_init / hr_destroy)."""

    def __init__(self, device=0, gate_mode=GATE_EBUR128, tp_mode=TP_SWR192K, engine=0, stage_bytes=0, stream=None):
        self._h = C.c_void_p()
        cfg = HrConfig(device, gate_mode, tp_mode, engine, stream, stage_bytes)
        _check(lib().hr_init(C.byref(cfg), C.byref(self._h)), "hr_init")
        self.device = device
        self.gate_mode = gate_mode

    def close(self):
        if self._h:
            lib().hr_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def analyze_batch(self, specs):
        n = len(specs)
        arr = (HrTrack * max(n, 1))(*[s.c for s in specs])
        out = (HrResult * max(n, 1))()
        _check(lib().hr_analyze_batch(self._h, arr, n, out), "hr_analyze_batch")
        return [_result_dict(out[i], specs[i].channels or 0) for i in range(n)]

    def analyze_prepared(self, batch):
        """hr_analyze_batch on a TrackBatch: the descriptor and result arrays are the caller's and live across calls, as
        they would in a C or Rust caller; nothing is converted.  batch.results() turns the last results into dicts."""
        _check(lib().hr_analyze_batch(self._h, batch.tracks, batch.n, batch.out), "hr_analyze_batch")

    def analyze_device_noalloc(self, spec):
        """hr_analyze_device_noalloc: one device-resident track, workspace from the library's static device array.
        Returns (rc, result dict): the caller looks at rc (size / readiness errors are part of the contract)."""
        r = HrResult()
        rc = lib().hr_analyze_device_noalloc(self._h, C.byref(spec.c), C.byref(r))
        return rc, _result_dict(r, spec.channels or 0)

    def loudnorm_first_pass(self, specs):
        """hr_loudnorm_first_pass: what `ffmpeg -af loudnorm=print_format=json` measures (src/analyzer.rs:253-267)."""
        n = len(specs)
        arr = (HrTrack * max(n, 1))(*[s.c for s in specs])
        res = (HrResult * max(n, 1))()
        _check(lib().hr_loudnorm_first_pass(self._h, arr, n, res), "hr_loudnorm_first_pass")
        return [_result_dict(res[i], specs[i].channels) for i in range(n)]

    def analyze(self, pcm, sample_rate, **kw):
        return self.analyze_batch([TrackSpec(pcm, sample_rate, **kw)])[0]

    def block_energies(self, spec):
        hop = spec.c.sample_rate // 10
        nh = spec.c.frames // hop if hop else 0
        H = np.zeros((max(nh, 1), spec.channels), dtype=np.float64)
        P = np.zeros(max(nh, 1), dtype=np.float64)
        n_h, n_b = C.c_uint64(), C.c_uint64()
        _check(lib().hr_block_energies(self._h, C.byref(spec.c), H.ctypes.data_as(C.POINTER(C.c_double)), nh,
                                       P.ctypes.data_as(C.POINTER(C.c_double)), nh, C.byref(n_h), C.byref(n_b)),
               "hr_block_energies")
        return H[:n_h.value], P[:n_b.value]

    def apply_gain(self, spec, gain_db, big_endian=False, out_device_ptr=None):
        """hr_apply_gain: packed 24-bit PCM of the track times 10^(gain_db/20).  Host track: returns (uint8 array,
        clipped); device track: writes frames*channels*3 bytes at out_device_ptr and returns (None, clipped)."""
        import numpy as np
        clipped = C.c_uint64(0)
        if out_device_ptr is not None:
            _check(lib().hr_apply_gain(self._h, C.byref(spec.c), gain_db, int(big_endian), C.c_void_p(int(out_device_ptr)),
                                       C.byref(clipped)), "hr_apply_gain")
            return None, int(clipped.value)
        nbytes = int(spec.c.frames) * spec.channels * 3
        out = np.zeros(max(nbytes, 1), dtype=np.uint8)
        _check(lib().hr_apply_gain(self._h, C.byref(spec.c), gain_db, int(big_endian), out.ctypes.data_as(C.c_void_p),
                                   C.byref(clipped)), "hr_apply_gain")
        return out[:nbytes], int(clipped.value)

    def gain_file(self, path, gain_db):
        """hr_pcm_gain_file: rewrites a WAV / AIFF file in place with the gain applied (24-bit PCM); returns clipped."""
        clipped = C.c_uint64(0)
        _check(lib().hr_pcm_gain_file(self._h, str(path).encode(), gain_db, C.byref(clipped)), "hr_pcm_gain_file")
        return int(clipped.value)

    def synth_fill(self, device_ptr, seed, sample_rate, channels, frames):
        _check(lib().hr_synth_fill(self._h, C.c_void_p(int(device_ptr)), seed, sample_rate, channels, frames), "hr_synth_fill")

    def last_timing(self):
        a, b = C.c_float(), C.c_float()
        lib().hr_last_timing(self._h, C.byref(a), C.byref(b))
        return a.value, b.value

    # ---- time shards (one process per GPU; the exchange is the caller's: torch.distributed) ----
    def shard_head(self, spec):
        head = np.zeros((16, MAX_CH), dtype=np.float64)
        _check(lib().hr_shard_head(self._h, C.byref(spec.c), head.ctypes.data_as(C.POINTER(C.c_double))), "hr_shard_head")
        return head

    def shard_tail(self, spec):
        """The last 16 frames of a shard as doubles (hr_shard_head on an offset view): what the right
        neighbour needs as left_tail before anybody has run pass 1."""
        t = spec.c
        assert t.frames >= 16
        off = (t.frames - 16) * t.channels * _FMT_BYTES[t.fmt]
        view = TrackSpec.__new__(TrackSpec)
        view.keep, view.channels = spec.keep, spec.channels
        view.c = HrTrack((t.pcm or 0) + off, 16, t.sample_rate, t.channels, t.fmt, t.ch_weight, t.flags, 0)
        return self.shard_head(view)

    def shard_pass1(self, spec, left_tail, right_head, first_hop=0):
        edge = HrShardEdge()
        lp = left_tail.ctypes.data_as(C.POINTER(C.c_double)) if left_tail is not None else None
        rp = right_head.ctypes.data_as(C.POINTER(C.c_double)) if right_head is not None else None
        _check(lib().hr_shard_pass1(self._h, C.byref(spec.c), lp, rp, int(left_tail is None), int(right_head is None),
                                    int(first_hop), C.byref(edge)), "hr_shard_pass1")
        return edge

    def shard_pass2(self, left_edge):
        part = HrShardPartial()
        _check(lib().hr_shard_pass2(self._h, C.byref(left_edge) if left_edge is not None else None, C.byref(part)),
               "hr_shard_pass2")
        return part

    def shard_pass3(self, reduced):
        """HR_GATE_EXACT: this shard's blocks above both gates, given the all-reduced sums of pass 2."""
        part = HrShardPartial()
        _check(lib().hr_shard_pass3(self._h, C.byref(reduced), C.byref(part)), "hr_shard_pass3")
        return part

    # ---- the same with the records resident on the device: `edge_ptr` / `partial_ptr` / `gathered_ptr` are device addresses
    # of the caller's buffers (sizeof(HrShardEdge) / sizeof(HrShardPartial) bytes, n partial images back to back) ----
    def shard_pass1_device(self, spec, left_tail, right_head, first_hop, edge_ptr):
        lp = left_tail.ctypes.data_as(C.POINTER(C.c_double)) if left_tail is not None else None
        rp = right_head.ctypes.data_as(C.POINTER(C.c_double)) if right_head is not None else None
        _check(lib().hr_shard_pass1_device(self._h, C.byref(spec.c), lp, rp, int(left_tail is None), int(right_head is None),
                                           int(first_hop), C.c_void_p(int(edge_ptr))), "hr_shard_pass1_device")

    def shard_pass2_device(self, left_edge_ptr, partial_ptr):
        _check(lib().hr_shard_pass2_device(self._h, C.c_void_p(int(left_edge_ptr)) if left_edge_ptr else None,
                                           C.c_void_p(int(partial_ptr))), "hr_shard_pass2_device")

    def shard_redo_device(self, partial_ptr):
        _check(lib().hr_shard_redo_device(self._h, C.c_void_p(int(partial_ptr))), "hr_shard_redo_device")

    def shard_finish_gathered(self, gathered_ptr, n, channels):
        """(rc, result): rc is OK or SHARD_AGAIN (every rank then runs shard_redo_device, gathers and finishes again)."""
        r = HrResult()
        rc = lib().hr_shard_finish_gathered(self._h, C.c_void_p(int(gathered_ptr)), n, channels, C.byref(r))
        if rc not in (OK, SHARD_AGAIN):
            raise HeadroomError(rc, "hr_shard_finish_gathered")
        return rc, _result_dict(r, channels)

    def shard_finish(self, partial, channels):
        r = HrResult()
        _check(lib().hr_shard_finish(self._h, C.byref(partial), channels, C.byref(r)), "hr_shard_finish")
        return _result_dict(r, channels)


def report_csv(rows):
    """rows: (filename, analysis dict as returned by gain_decision) -> the CSV text of src/report.rs generate_csv."""
    buf = C.create_string_buffer(4096)
    n = lib().hr_report_csv_header(buf, len(buf))
    assert n > 0
    out = buf.raw[:n].decode()
    for name, a in rows:
        rec = HrAnalysis(**{f: a[f] for f, _ in HrAnalysis._fields_})
        n = lib().hr_report_csv_row(name.encode(), C.byref(rec), buf, len(buf))
        _check(n if n < 0 else OK, "hr_report_csv_row")
        out += buf.raw[:n].decode()
    return out


def pcm_probe(data):
    """hr_pcm_probe on a bytes object: (rc, dict of the header fields)."""
    info = HrPcmInfo()
    rc = lib().hr_pcm_probe(data, len(data), C.byref(info))
    return rc, {f: getattr(info, f) for f, _ in HrPcmInfo._fields_}


class LoadedFile:
    """hr_pcm_load_file: a WAV / AIFF file in pinned host memory, usable as a TrackSpec-like object."""

    def __init__(self, path):
        self._pcm = C.c_void_p()
        self.c = HrTrack()
        self.weights = (C.c_float * MAX_CH)()
        _check(lib().hr_pcm_load_file(os.fsencode(path), C.byref(self._pcm), C.byref(self.c), self.weights), "hr_pcm_load_file")
        self.channels = self.c.channels

    def close(self):
        if self._pcm:
            lib().hr_free_pinned(self._pcm)
            self._pcm = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def analyze_stream_sharded(meters, spec):
    """hr_analyze_stream_sharded: one stream, time-sharded over the given Meters (one per GPU) in this process."""
    arr = (C.c_void_p * len(meters))(*[m._h for m in meters])
    r = HrResult()
    _check(lib().hr_analyze_stream_sharded(arr, len(meters), C.byref(spec.c), C.byref(r)), "hr_analyze_stream_sharded")
    return _result_dict(r, spec.channels)


def gain_decision(input_i, input_tp, kind=KIND_LOSSLESS, bitrate_kbps=None, split=None):
    """src/analyzer.rs:284-326 through the C ABI.  split = (high, low) selects SplitBitrate."""
    mode = HrTpTarget(1, split[0], split[1]) if split else None
    out = HrAnalysis()
    rc = lib().hr_gain_decision(input_i, input_tp, kind, int(bitrate_kbps is not None), int(bitrate_kbps or 0),
                                C.byref(mode) if mode else None, C.byref(out))
    return rc, {f: getattr(out, f) for f, _ in HrAnalysis._fields_}


def _as_result(lufs, dbtp, status=OK):
    r = HrResult()
    r.integrated_lufs, r.true_peak_dbtp, r.status = lufs, dbtp, status
    return r


def loudnorm_text_values(lufs, dbtp):
    """hr_loudnorm_text_values: the two numbers after loudnorm's "%.2f" printing and the analyzer's parse."""
    a, b = C.c_double(), C.c_double()
    r = _as_result(lufs, dbtp)
    _check(lib().hr_loudnorm_text_values(C.byref(r), C.byref(a), C.byref(b)), "hr_loudnorm_text_values")
    return a.value, b.value


def gain_decision_for_result(lufs, dbtp, kind=KIND_LOSSLESS, bitrate_kbps=None, split=None, status=OK):
    """hr_gain_decision_for_result on a measurement (src/analyzer.rs:273-326: text rounding, then the decision)."""
    mode = HrTpTarget(1, split[0], split[1]) if split else None
    out = HrAnalysis()
    r = _as_result(lufs, dbtp, status)
    rc = lib().hr_gain_decision_for_result(C.byref(r), kind, int(bitrate_kbps is not None), int(bitrate_kbps or 0),
                                           C.byref(mode) if mode else None, C.byref(out))
    return rc, {f: getattr(out, f) for f, _ in HrAnalysis._fields_}
