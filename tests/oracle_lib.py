"""ctypes binding of the CPU oracle (oracle/libhr_oracle.so).

Test infrastructure only: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  The product package (headroom_amd) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "libhr_oracle.so")

FMT_F32, FMT_S16, FMT_S24, FMT_S32, FMT_F64 = range(5)
GATE_EBUR128, GATE_EXACT, GATE_LOUDNORM = range(3)
TP_SWR192K, TP_BS1770, TP_SAMPLE = range(3)
MAX_CH = 8

_NP_FMT = {np.dtype("float32"): FMT_F32, np.dtype("int16"): FMT_S16,
           np.dtype("int32"): FMT_S32, np.dtype("float64"): FMT_F64}


class Track(C.Structure):
    _fields_ = [("pcm", C.c_void_p), ("frames", C.c_uint64), ("sample_rate", C.c_uint32),
                ("channels", C.c_uint16), ("fmt", C.c_uint16), ("ch_weight", C.POINTER(C.c_float))]


class Result(C.Structure):
    _fields_ = [("integrated_lufs", C.c_double), ("true_peak_dbtp", C.c_double),
                ("rel_threshold_lufs", C.c_double), ("n_blocks", C.c_uint64),
                ("n_gated", C.c_uint64), ("peak_per_ch", C.c_double * MAX_CH), ("status", C.c_int)]


class Range(C.Structure):
    _fields_ = [("lra", C.c_double), ("lra_low", C.c_double), ("lra_high", C.c_double), ("lra_threshold", C.c_double),
                ("max_momentary", C.c_double), ("max_shortterm", C.c_double), ("n_shortterm", C.c_uint64)]


class KCoef(C.Structure):
    _fields_ = [("pre_b", C.c_double * 3), ("pre_a", C.c_double * 3),
                ("rlb_b", C.c_double * 3), ("rlb_a", C.c_double * 3)]


def build():
    """(Re)build the oracle library if it is missing or older than its sources."""
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("hr_oracle.c", "hr_oracle.h", "Makefile")]
    if (not os.path.exists(LIB_PATH)
            or os.path.getmtime(LIB_PATH) < max(os.path.getmtime(s) for s in srcs)):
        subprocess.check_call(["make", "-C", ORACLE_DIR], stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        L.hro_kweight_coeffs.argtypes = [C.c_double, C.POINTER(KCoef)]
        L.hro_default_weights.argtypes = [C.c_int, C.POINTER(C.c_float)]
        L.hro_hop_sums.argtypes = [C.POINTER(Track), C.POINTER(C.c_double), C.c_uint64]
        L.hro_hop_sums.restype = C.c_uint64
        L.hro_block_powers.argtypes = [C.POINTER(C.c_double), C.c_uint64, C.c_int,
                                       C.POINTER(C.c_float), C.c_uint32, C.POINTER(C.c_double)]
        L.hro_block_powers.restype = C.c_uint64
        L.hro_gate.argtypes = [C.POINTER(C.c_double), C.c_uint64, C.c_int, C.POINTER(Result),
                               C.POINTER(C.c_uint64)]
        L.hro_range_stats.argtypes = [C.POINTER(C.c_double), C.c_uint64, C.c_int, C.POINTER(C.c_float), C.c_uint32,
                                      C.c_int, C.POINTER(Range)]
        L.hro_apply_gain.argtypes = [C.POINTER(Track), C.c_double, C.c_int, C.POINTER(C.c_uint8)]
        L.hro_apply_gain.restype = C.c_uint64
        L.hro_swr_bank.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(C.c_int), C.POINTER(C.c_double)]
        L.hro_swr_bank.restype = C.c_int
        L.hro_bs1770_taps.restype = C.POINTER(C.c_double)
        L.hro_true_peak.argtypes = [C.POINTER(Track), C.c_int, C.c_int, C.POINTER(C.c_double)]
        L.hro_measure.argtypes = [C.POINTER(Track), C.c_int, C.c_int, C.POINTER(Result)]
        L.hro_ebur128_literal.argtypes = [C.POINTER(Track), C.POINTER(Result), C.POINTER(C.c_double),
                                          C.c_uint64]
        L.hro_loudnorm_measure.argtypes = [C.POINTER(Track), C.POINTER(Result)]
        L.hro_synth_track.argtypes = [C.c_uint64, C.c_uint32, C.c_int, C.c_uint64,
                                      C.POINTER(C.c_float)]
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def make_track(pcm, fs, channels=None, fmt=None, weights=None):
    """pcm: numpy array [frames, channels] (or 1-D for mono); for s24 pass uint8 [frames*ch*3] and fmt.

    Returns (Track, keepalive)."""
    if fmt == FMT_S24:
        buf = np.ascontiguousarray(pcm, dtype=np.uint8)
        assert channels
        frames = buf.size // (3 * channels)
    else:
        buf = np.ascontiguousarray(pcm)
        if buf.ndim == 1:
            buf = buf[:, None]
        frames, ch = buf.shape
        channels = channels or ch
        fmt = _NP_FMT[buf.dtype] if fmt is None else fmt
    w = None
    wptr = C.POINTER(C.c_float)()
    if weights is not None:
        w = np.ascontiguousarray(weights, dtype=np.float32)
        wptr = w.ctypes.data_as(C.POINTER(C.c_float))
    t = Track(buf.ctypes.data, frames, fs, channels, fmt, wptr)
    return t, (buf, w)


def kcoef(fs):
    k = KCoef()
    lib().hro_kweight_coeffs(float(fs), C.byref(k))
    return {n: np.array(getattr(k, n)) for n in ("pre_b", "pre_a", "rlb_b", "rlb_a")}


def default_weights(channels):
    w = np.zeros(channels, dtype=np.float32)
    lib().hro_default_weights(channels, w.ctypes.data_as(C.POINTER(C.c_float)))
    return w


def hop_sums(pcm, fs, **kw):
    t, keep = make_track(pcm, fs, **kw)
    nh = t.frames // (fs // 10)
    H = np.zeros((max(nh, 1), t.channels), dtype=np.float64)
    n = lib().hro_hop_sums(C.byref(t), _dp(H), nh)
    return H[:n]


def block_powers(H, fs, weights=None):
    H = np.ascontiguousarray(H, dtype=np.float64)
    nh, ch = H.shape
    w = default_weights(ch) if weights is None else np.ascontiguousarray(weights, dtype=np.float32)
    P = np.zeros(max(nh, 1), dtype=np.float64)
    nb = lib().hro_block_powers(_dp(H), nh, ch, w.ctypes.data_as(C.POINTER(C.c_float)), fs, _dp(P))
    return P[:nb]


def result_dict(r, channels):
    return {"integrated_lufs": r.integrated_lufs, "true_peak_dbtp": r.true_peak_dbtp,
            "rel_threshold_lufs": r.rel_threshold_lufs, "n_blocks": r.n_blocks, "n_gated": r.n_gated,
            "peak_per_ch": [r.peak_per_ch[i] for i in range(channels)], "status": r.status}


def gate(P, mode=GATE_EBUR128, want_hist=False):
    P = np.ascontiguousarray(P, dtype=np.float64)
    r = Result()
    nbins = 8001 if mode == GATE_EBUR128 else 1000
    hist = np.zeros(nbins, dtype=np.uint64)
    lib().hro_gate(_dp(P), P.size, mode, C.byref(r), hist.ctypes.data_as(C.POINTER(C.c_uint64)))
    d = result_dict(r, 0)
    if want_hist:
        d["hist"] = hist
    return d


def range_stats(pcm, fs, gate_mode=GATE_EBUR128, weights=None, **kw):
    """Loudness range and maximum momentary / short-term loudness of a track (hro_range_stats on its hop sums)."""
    H = np.ascontiguousarray(hop_sums(pcm, fs, **kw))
    nh, ch = H.shape
    w = default_weights(ch) if weights is None else np.ascontiguousarray(weights, dtype=np.float32)
    o = Range()
    lib().hro_range_stats(_dp(H), nh, ch, w.ctypes.data_as(C.POINTER(C.c_float)), fs, gate_mode, C.byref(o))
    return {f: getattr(o, f) for f, _ in Range._fields_}


def apply_gain(pcm, fs, gain_db, big_endian=False, **kw):
    """hro_apply_gain: (packed s24 bytes as uint8 array, number of saturated samples)."""
    t, keep = make_track(pcm, fs, **kw)
    out = np.zeros(t.frames * t.channels * 3, dtype=np.uint8)
    clipped = lib().hro_apply_gain(C.byref(t), gain_db, int(big_endian), out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out, int(clipped)


def swr_bank(in_rate, out_rate=192000):
    step = C.c_int(0)
    P = lib().hro_swr_bank(in_rate, out_rate, C.byref(step), None)
    if P == 0:
        return 0, 0, None
    taps = np.zeros((P + 1, 32), dtype=np.float64)
    lib().hro_swr_bank(in_rate, out_rate, C.byref(step), _dp(taps))
    return P, step.value, taps


def bs1770_taps():
    p = lib().hro_bs1770_taps()
    return np.ctypeslib.as_array(p, shape=(4, 12)).copy()


def true_peak(pcm, fs, tp_mode=TP_SWR192K, flush_end=0, **kw):
    t, keep = make_track(pcm, fs, **kw)
    pk = np.zeros(MAX_CH, dtype=np.float64)
    lib().hro_true_peak(C.byref(t), tp_mode, flush_end, _dp(pk))
    return pk[:t.channels]


def measure(pcm, fs, gate_mode=GATE_EBUR128, tp_mode=TP_SWR192K, **kw):
    t, keep = make_track(pcm, fs, **kw)
    r = Result()
    rc = lib().hro_measure(C.byref(t), gate_mode, tp_mode, C.byref(r))
    assert rc == 0, rc
    return result_dict(r, t.channels)


def ebur128_literal(pcm, fs, **kw):
    t, keep = make_track(pcm, fs, **kw)
    r = Result()
    cap = t.frames // (fs // 10) + 1
    P = np.zeros(cap, dtype=np.float64)
    lib().hro_ebur128_literal(C.byref(t), C.byref(r), _dp(P), cap)
    d = result_dict(r, t.channels)
    d["block_powers"] = P[:r.n_blocks]
    return d


def loudnorm_measure(pcm, fs, **kw):
    t, keep = make_track(pcm, fs, **kw)
    r = Result()
    lib().hro_loudnorm_measure(C.byref(t), C.byref(r))
    return result_dict(r, t.channels)


class SwrPlan(C.Structure):
    _fields_ = [("phases", C.c_int), ("linear", C.c_int), ("d0", C.c_uint32), ("s0", C.c_uint32), ("frac_scale", C.c_uint32),
                ("inv_src_incr", C.c_double)]


def swr_plan(in_rate, out_rate=192000):
    """hro_swr_plan: libswresample's phase count and dst_incr / src_incr bookkeeping for this rate pair."""
    pl = SwrPlan()
    L = lib()
    L.hro_swr_plan.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(SwrPlan)]
    L.hro_swr_plan.restype = C.c_int
    L.hro_swr_plan(in_rate, out_rate, C.byref(pl))
    return {f: getattr(pl, f) for f, _ in SwrPlan._fields_}


def synth_track(seed, fs, channels, frames):
    out = np.zeros((frames, channels), dtype=np.float32)
    lib().hro_synth_track(seed, fs, channels, frames, out.ctypes.data_as(C.POINTER(C.c_float)))
    return out
