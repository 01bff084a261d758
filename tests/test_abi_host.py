"""CPU-side tests (no GPU): the C-ABI library loads and exports every symbol include/headroom_hip.h
declares, struct layouts match the header, the host-side mirror of the analyzer's scalar logic
(src/analyzer.rs:157-187, :284-326) behaves like the reference, and the product's own constant tables
agree with the oracle's."""
import ctypes as C
import math
import os
import re
import subprocess

import numpy as np
import pytest

import headroom_amd as hr

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "headroom_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(hr_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(hr.EXPORTS), declared ^ set(hr.EXPORTS)
    L = hr.lib()
    for name in declared:
        assert getattr(L, name) is not None


def test_struct_sizes_match_header():
    # compile a tiny C program against the header and compare sizeof()
    src = r'''
#include <stdio.h>
#include "headroom_hip.h"
int main(void){ printf("%zu %zu %zu %zu %zu %zu %zu\n", sizeof(hr_config), sizeof(hr_track), sizeof(hr_result),
  sizeof(hr_shard_edge), sizeof(hr_shard_partial), sizeof(hr_tp_target), sizeof(hr_analysis)); return 0; }'''
    exe = "/tmp/hr_sizes"
    subprocess.run(["gcc", "-x", "c", "-", "-I", os.path.join(ROOT, "include"), "-o", exe], input=src.encode(), check=True)
    sizes = list(map(int, subprocess.check_output([exe]).split()))
    mine = [C.sizeof(t) for t in (hr.HrConfig, hr.HrTrack, hr.HrResult, hr.HrShardEdge, hr.HrShardPartial,
                                  hr.HrTpTarget, hr.HrAnalysis)]
    assert sizes == mine


def test_no_device_means_error_not_fallback():
    if hr.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(hr.HeadroomError) as e:
        hr.Meter()
    assert e.value.code == hr.ERR_NO_DEVICE


def test_shipped_library_has_no_experiment_switches():
    """Timing experiments (HR_ABLATE and friends make the results wrong on purpose) compile only into
    libheadroom_hip_exp.so (-DHR_EXPERIMENTS); the shipped library must not even contain their names."""
    data = open(hr.LIB_PATH, "rb").read()
    for name in (b"HR_ABLATE", b"HR_FUSED_SPLIT", b"HR_SERIAL_TAIL", b"HR_WAVE_MAP", b"HR_SPLIT_GEOM", b"HR_EXP_", b"k_fused_split",
                 b"HR_ALIGN_MODE", b"HR_ALIGN_SUPER", b"HR_ALIGN_STRIDE", b"HR_HOST_TIMING"):
        assert name not in data, name


def test_strerror():
    assert hr.lib().hr_strerror(0) == b"ok"
    assert b"no HIP device" in hr.lib().hr_strerror(hr.ERR_NO_DEVICE)


# ---- gain decision: src/analyzer.rs:304-326 ---------------------------------------------------

def test_gain_none_below_min_effective_gain():
    rc, a = hr.gain_decision(-9.0, -0.52)  # headroom 0.02 < 0.05
    assert rc == 0 and a["gain_method"] == hr.GAIN_NONE and a["effective_gain"] == 0.0
    assert abs(a["headroom"] - 0.02) < 1e-12 and a["target_tp"] == -0.5


def test_gain_lossless_uses_full_headroom():
    rc, a = hr.gain_decision(-14.0, -3.2)
    assert a["gain_method"] == hr.GAIN_FFMPEG_LOSSLESS
    assert abs(a["effective_gain"] - 2.7) < 1e-12 and a["lossless_gain_steps"] == 0
    assert a["has_bitrate"] == 0


def test_gain_mp3_steps_of_1_5_db():
    rc, a = hr.gain_decision(-12.0, -5.0, kind=hr.KIND_MP3, bitrate_kbps=320)  # headroom 4.5 -> 3 steps
    assert a["gain_method"] == hr.GAIN_MP3_LOSSLESS and a["lossless_gain_steps"] == 3
    assert abs(a["effective_gain"] - 4.5) < 1e-12
    rc, a = hr.gain_decision(-12.0, -1.9, kind=hr.KIND_MP3, bitrate_kbps=320)  # headroom 1.4 -> re-encode
    assert a["gain_method"] == hr.GAIN_MP3_REENCODE and a["lossless_gain_steps"] == 0
    assert abs(a["effective_gain"] - 1.4) < 1e-12
    rc, a = hr.gain_decision(-12.0, -3.6, kind=hr.KIND_AAC, bitrate_kbps=256)  # headroom 3.1 -> 2 steps
    assert a["gain_method"] == hr.GAIN_AAC_LOSSLESS and a["lossless_gain_steps"] == 2
    rc, a = hr.gain_decision(-12.0, -1.0, kind=hr.KIND_AAC)
    assert a["gain_method"] == hr.GAIN_AAC_REENCODE


def test_tp_target_split_bitrate():
    # TpTargetMode::SplitBitrate(high, low): lossless and >= 256 kbps -> high, else (or unknown) -> low
    for kind, kbps, want in ((hr.KIND_LOSSLESS, None, -0.5), (hr.KIND_MP3, 320, -0.5), (hr.KIND_MP3, 256, -0.5),
                             (hr.KIND_MP3, 192, -1.0), (hr.KIND_AAC, None, -1.0)):
        rc, a = hr.gain_decision(-10.0, -6.0, kind=kind, bitrate_kbps=kbps, split=(-0.5, -1.0))
        assert a["target_tp"] == want, (kind, kbps)


def test_non_finite_measurement_is_rejected():
    # src/analyzer.rs:284-290
    for i, tp in ((-math.inf, -3.0), (-14.0, -math.inf), (math.nan, 0.0)):
        rc, a = hr.gain_decision(i, tp)
        assert rc == hr.ERR_INVALID and a["status"] == hr.ERR_INVALID


# ---- the measurement as the reference sees it: loudnorm's "%.2f" text, parsed (src/analyzer.rs:102-106, 273-280) ----

def ref_decision(input_i_text, input_tp_text, kind=None, kbps=None, target=-0.5):
    """src/analyzer.rs:273-326 restated on the printed strings (Uniform target): what the reference would decide."""
    i, tp = float(input_i_text), float(input_tp_text)
    if not (math.isfinite(i) and math.isfinite(tp)):
        return None
    headroom = target - tp
    lossy = kind in (hr.KIND_MP3, hr.KIND_AAC)
    if headroom < 0.05:
        return (hr.GAIN_NONE, 0.0, 0)
    if not lossy:
        return (hr.GAIN_FFMPEG_LOSSLESS, headroom, 0)
    steps = math.floor(headroom / 1.5)
    if steps >= 1:
        return (hr.GAIN_AAC_LOSSLESS if kind == hr.KIND_AAC else hr.GAIN_MP3_LOSSLESS, steps * 1.5, steps)
    return (hr.GAIN_AAC_REENCODE if kind == hr.KIND_AAC else hr.GAIN_MP3_REENCODE, headroom, 0)


def test_text_values_round_like_printf_and_parse():
    for v in (-0.5451, -0.5449, -0.545, -0.555, -2.004, -1.996, -14.004999, -23.0, 0.004999, 0.005, 2.675, -70.0, 1e-9, -1e-9):
        i, tp = hr.loudnorm_text_values(v, v)
        assert i == tp == float("%.2f" % v), v
    i, tp = hr.loudnorm_text_values(-math.inf, -math.inf)
    assert i == -math.inf and tp == -math.inf
    i, tp = hr.loudnorm_text_values(math.nan, math.inf)
    assert math.isnan(i) and tp == math.inf


def test_decision_on_thresholds_follows_the_rounded_values():
    # values straddling MIN_EFFECTIVE_GAIN (0.05 dB, src/analyzer.rs:307) and multiples of GAIN_STEP (1.5 dB, :313):
    # the reference decides on the two-decimal text, the unrounded value would decide differently
    cases = [(-0.5451, hr.KIND_LOSSLESS, None), (-0.5449, hr.KIND_LOSSLESS, None), (-0.546, hr.KIND_LOSSLESS, None),
             (-0.554, hr.KIND_LOSSLESS, None), (-0.5549, hr.KIND_MP3, 320), (-2.004, hr.KIND_MP3, 320),
             (-1.996, hr.KIND_MP3, 320), (-1.9951, hr.KIND_MP3, 320), (-2.0049, hr.KIND_AAC, 256), (-3.504, hr.KIND_AAC, 256),
             (-3.496, hr.KIND_AAC, 256), (-3.4949, hr.KIND_MP3, 128), (-5.0, hr.KIND_MP3, 320), (-0.5, hr.KIND_LOSSLESS, None)]
    flips = 0
    for tp, kind, kbps in cases:
        want = ref_decision("%.2f" % -14.237, "%.2f" % tp, kind, kbps)
        rc, a = hr.gain_decision_for_result(-14.237, tp, kind=kind, bitrate_kbps=kbps)
        assert rc == 0
        assert a["input_tp"] == float("%.2f" % tp) and a["input_i"] == -14.24
        assert (a["gain_method"], a["lossless_gain_steps"]) == (want[0], want[2]), (tp, a, want)
        assert abs(a["effective_gain"] - want[1]) < 1e-12 and abs(a["headroom"] - (-0.5 - float("%.2f" % tp))) < 1e-12
        rc, raw = hr.gain_decision(-14.237, tp, kind=kind, bitrate_kbps=kbps)
        flips += (raw["gain_method"], raw["lossless_gain_steps"]) != (a["gain_method"], a["lossless_gain_steps"])
    assert flips >= 4  # the cases really sit on the thresholds: the unrounded decision differs on several of them
    # the one from the review: -0.5451 dBTP prints as "-0.55" -> headroom 0.05 -> processed; unrounded 0.0451 -> skipped
    rc, a = hr.gain_decision_for_result(-10.0, -0.5451)
    assert a["gain_method"] == hr.GAIN_FFMPEG_LOSSLESS and abs(a["effective_gain"] - 0.05) < 1e-12
    rc, a = hr.gain_decision(-10.0, -0.5451)
    assert a["gain_method"] == hr.GAIN_NONE


def test_decision_for_result_rejects_silence_and_failed_tracks():
    rc, a = hr.gain_decision_for_result(-math.inf, -math.inf)  # loudnorm prints "-inf": src/analyzer.rs:282-290
    assert rc == hr.ERR_INVALID and a["status"] == hr.ERR_INVALID and a["gain_method"] == hr.GAIN_NONE
    rc, a = hr.gain_decision_for_result(-14.0, -3.0, status=hr.ERR_RATE)
    assert rc == hr.ERR_RATE and a["status"] == hr.ERR_RATE


def test_csv_row_double_rounds_like_the_reference():
    # report.rs:53-58 prints {:.1} of the already two-decimal value: -0.549 -> "-0.55" -> "-0.6" there (ties: the binary
    # value of -0.55 is just below the tie), whereas one rounding of -0.549 gives "-0.5"
    rc, a = hr.gain_decision_for_result(-14.0, -0.549)
    text = hr.report_csv([("a.wav", a)]).splitlines()[1].split(",")
    assert text[4] == "%.1f" % float("%.2f" % -0.549)
    rc, raw = hr.gain_decision(-14.0, -0.549)
    assert hr.report_csv([("a.wav", raw)]).splitlines()[1].split(",")[4] == "-0.5"


# ---- the product's tables vs the oracle's (two independent builds of the same published formulas) --

def _tables_exe():
    exe = "/tmp/hr_tables_dump"
    src = os.path.join(ROOT, "tests", "tables_dump.cpp")
    subprocess.check_call(["g++", "-O2", "-std=c++17", src, os.path.join(ROOT, "headroom_amd", "csrc", "hr_tables.cpp"),
                           "-o", exe])
    return exe


@pytest.mark.parametrize("fs", [44100, 48000, 96000])
def test_product_tables_match_oracle(oracle, fs):
    out = subprocess.check_output([_tables_exe(), str(fs)]).decode().split()
    vals = list(map(float, out))
    k = oracle.kcoef(fs)
    mine = vals[:7]
    ref = [k["pre_b"][0], k["pre_b"][1], k["pre_b"][2], k["pre_a"][1], k["pre_a"][2], k["rlb_a"][1], k["rlb_a"][2]]
    assert mine == ref  # bit-identical: same formulas, same libm
    P, step = int(vals[7]), int(vals[8])
    oP, ostep, otaps = oracle.swr_bank(fs)
    assert (P, step) == (oP, ostep)
    taps = np.array(vals[10:10 + (P + 1) * 32]).reshape(P + 1, 32)
    assert np.array_equal(taps, otaps)
    assert vals[9] < 1e-25  # |M^3|: the carry series is converged after three hops


def test_ebur128_bin_edges_agree_with_the_libm_formula():
    """The kernels decide f_ebur128's histogram bins (and the first bins of its -10 / -20 LU gates) by comparing powers
    with host-built edges instead of taking a device log10.  Every edge must be exactly where the formula the oracle
    evaluates (host libm, same expression order) changes its answer: a power one ulp below an edge -- 1e-16 relative --
    lands in the lower bin, the edge itself in the upper one."""
    vals = np.array(list(map(float, subprocess.check_output([_tables_exe(), "edges"]).decode().split())))
    assert vals.size == 3 * 8001

    def bin_of(p, offset, gate):
        loud = (-0.691 + 10.0 * math.log10(p)) + offset
        if gate and not loud >= -70.0:
            return -1
        return min(max(int((loud - (-70.0)) * 100), 0), 8000)

    for k, off in enumerate((0.0, -10.0, -20.0)):
        e = vals[k * 8001:(k + 1) * 8001]
        assert np.all(np.diff(e) > 0)
        for b in range(8001):
            if k and b == 0:
                assert e[0] == 0.0
                continue
            assert bin_of(e[b], off, k == 0) >= b, (k, b)
            assert bin_of(np.nextafter(e[b], 0.0), off, k == 0) < b, (k, b)
        # and in between the edges the kernel's search (largest b with edge[b] <= p) gives the formula's bin
        rng = np.random.default_rng(5 + k)
        ps = 10.0 ** rng.uniform(-8.5, 1.2, 4000)
        for p in ps:
            want = bin_of(p, off, k == 0)
            got = int(np.searchsorted(e, p, side="right")) - 1 if p >= e[0] else -1
            assert got == want, (k, p, got, want)


def test_cpu_carry_algebra(oracle):
    """The zero-state + carry-correction algebra of k_hop_zero_state / k_hop_combine, on the host."""
    exe = "/tmp/hr_carry_check"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-mfma", "-ffp-contract=off", "-c",
                           os.path.join(ROOT, "tests", "cpu_carry_check.cpp"), "-o", "/tmp/hr_cc_a.o"])
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-c", os.path.join(ROOT, "headroom_amd", "csrc", "hr_tables.cpp"),
                           "-o", "/tmp/hr_cc_b.o"])
    subprocess.check_call(["g++", "/tmp/hr_cc_a.o", "/tmp/hr_cc_b.o", oracle.LIB_PATH, "-o", exe, "-lm",
                           "-Wl,-rpath," + os.path.dirname(oracle.LIB_PATH)])
    for args in (("48000", "2", "1", "0"), ("44100", "2", "2", "1"), ("96000", "1", "3", "2")):
        subprocess.check_call([exe, *args], stdout=subprocess.DEVNULL)


@pytest.mark.parametrize("fs", [44100, 48000, 96000])
def test_tp_screen_tiles_and_error_bound(fs):
    """f16 coefficient tiles of the MFMA true-peak screen: every output of every tile is reproduced within
    the screen's error bound, and the host f32->f16 conversion rounds to nearest even."""
    exe = "/tmp/hr_tp_tiles_check"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-mfma", os.path.join(ROOT, "tests", "cpu_tp_tiles_check.cpp"),
                           os.path.join(ROOT, "headroom_amd", "csrc", "hr_tables.cpp"), "-o", exe])
    subprocess.check_call([exe, str(fs)], stdout=subprocess.DEVNULL)
    subprocess.check_call([exe, str(fs), "bs"], stdout=subprocess.DEVNULL)   # the BS.1770 Annex 2 bank


# ---- CSV report rows: src/report.rs:29-59 ----------------------------------------------------------------------

def test_report_csv_rows_follow_report_rs():
    rows = []
    rc, a = hr.gain_decision(-14.04, -3.26)                                        # lossless, headroom 2.76
    rows.append(("track one.flac", a))
    rc, a = hr.gain_decision(-9.96, -5.0, kind=hr.KIND_MP3, bitrate_kbps=320)      # 3 native steps
    rows.append(("b,with comma.mp3", a))
    rc, a = hr.gain_decision(-12.0, -1.9, kind=hr.KIND_AAC, bitrate_kbps=192, split=(-0.5, -1.0))  # re-encode, low target
    rows.append(('say "hi".m4a', a))
    rc, a = hr.gain_decision(-8.0, -0.52)                                          # nothing to gain
    rows.append(("loud.wav", a))
    text = hr.report_csv(rows)
    assert text.splitlines() == [
        "Filename,Format,Bitrate (kbps),LUFS,True Peak (dBTP),Target (dBTP),Headroom (dB),Method,Effective Gain (dB)",
        "track one.flac,Lossless,-,-14.0,-3.3,-0.5,+2.8,ffmpeg,+2.8",
        '"b,with comma.mp3",MP3,320,-10.0,-5.0,-0.5,+4.5,native,+4.5',
        '"say ""hi"".m4a",AAC,192,-12.0,-1.9,-1.0,+0.9,re-encode,+0.9',
        "loud.wav,-,-,-8.0,-0.5,-0.5,+0.0,none,+0.0",
    ]
    assert text.endswith("\n")


def test_report_csv_row_buffer_too_small():
    rc, a = hr.gain_decision(-14.0, -3.0)
    rec = hr.HrAnalysis(**{f: a[f] for f, _ in hr.HrAnalysis._fields_})
    buf = C.create_string_buffer(8)
    assert hr.lib().hr_report_csv_row(b"x.wav", C.byref(rec), buf, len(buf)) == hr.ERR_TOO_LARGE
