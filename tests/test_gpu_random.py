"""Seeded random differential test: random rates, channel counts, sample formats, lengths, levels, gate and true-peak
modes, all in ONE hr_analyze_batch per mode pair (mixed batches exercise the item sorting, the plane splitting and the
generic fallbacks together), each track against the oracle."""
import math

import numpy as np
import pytest

import headroom_amd as hr
import signals

pytestmark = pytest.mark.gpu

RATES = [8000, 11025, 16000, 22050, 32000, 44100, 48000, 88200, 96000, 176400, 192000]  # (11.025 / 22.05 kHz: phase-interpolated)
GATES = {"ebur128": 0, "exact": 1, "loudnorm": 2}
TPS = {"swr192k": 0, "bs1770": 1, "sample": 2}


def make_case(rng):
    fs = int(rng.choice(RATES))
    ch = int(rng.choice([1, 2, 2, 2, 3, 4, 5, 6, 8]))
    secs = float(rng.choice([0.05, 0.35, 0.45, 1.0, 2.3, 3.6, 5.2]))
    frames = max(1, int(fs * secs) + int(rng.integers(0, 50)))
    level = float(rng.uniform(-40.0, -1.0))
    x = signals.seeded_noise(int(rng.integers(1, 1 << 30)), frames, ch, level).astype(np.float64)
    kind = rng.choice(["plain", "quiet_start", "tone", "silent_channel", "dc"])
    if kind == "quiet_start":
        x[: frames // 2] *= 10.0 ** (-rng.uniform(20, 70) / 20.0)
    elif kind == "tone":
        n = np.arange(frames)
        x[:, 0] += 0.3 * np.sin(2 * np.pi * rng.uniform(20.0, 0.45 * fs) * n / fs)
    elif kind == "silent_channel":
        x[:, -1] = 0.0
    elif kind == "dc":
        x += rng.uniform(-0.2, 0.2)
    x = np.clip(x, -0.999, 0.999)
    fmt = rng.choice(["f32", "s16", "s24", "s32", "f64"])
    kw = {}
    if fmt == "f32":
        pcm = x.astype(np.float32)
    elif fmt == "s16":
        pcm = signals.to_s16(x)
    elif fmt == "s24":
        pcm = signals.to_s24_bytes(x)
        kw = {"channels": ch, "fmt": hr.FMT_S24}
    elif fmt == "s32":
        pcm = (x * 2147483647.0).astype(np.int32)
    else:
        pcm = x
    if rng.random() < 0.2:
        kw["weights"] = rng.choice([0.0, 0.5, 1.0, 1.41], size=ch).astype(np.float32)
    return pcm, fs, kw, (fs, ch, fmt, kind, frames)


@pytest.mark.parametrize("gate,tp,seed", [("ebur128", "swr192k", 1), ("loudnorm", "swr192k", 2), ("exact", "bs1770", 3),
                                           ("ebur128", "bs1770", 4), ("ebur128", "sample", 5), ("ebur128", "swr192k", 6)])
def test_random_mixed_batches_match_the_oracle(oracle, gate, tp, seed):
    rng = np.random.default_rng(seed)
    cases = [make_case(rng) for _ in range(28)]
    m = hr.Meter(device=0, gate_mode=GATES[gate], tp_mode=TPS[tp])
    m2 = hr.Meter(device=0, gate_mode=GATES[gate], tp_mode=TPS[tp], engine=2 + seed % 2)  # rows of a whole hop only / of a fraction
    try:
        got = m.analyze_batch([hr.TrackSpec(p, fs, **kw) for p, fs, kw, _ in cases])
        got2 = m2.analyze_batch([hr.TrackSpec(p, fs, **kw) for p, fs, kw, _ in cases])
    finally:
        m.close()
        m2.close()
    for a, b in zip(got, got2):  # the row length changes nothing but rounding in the 13th digit of an energy
        assert a["peak_per_ch"] == b["peak_per_ch"] and a["n_gated"] == b["n_gated"]
        assert a["integrated_lufs"] == b["integrated_lufs"] or abs(a["integrated_lufs"] - b["integrated_lufs"]) < 1e-9
    for (pcm, fs, kw, tag), g in zip(cases, got):
        ref = oracle.measure(pcm, fs, gate_mode=GATES[gate], tp_mode=TPS[tp], **kw)
        assert g["status"] == 0, tag
        assert g["n_blocks"] == ref["n_blocks"] and g["n_gated"] == ref["n_gated"], (tag, g, ref)
        if math.isfinite(ref["integrated_lufs"]):
            assert abs(g["integrated_lufs"] - ref["integrated_lufs"]) < 1e-6, (tag, g, ref)
        else:
            assert g["integrated_lufs"] == ref["integrated_lufs"], (tag, g, ref)
        assert g["peak_per_ch"] == ref["peak_per_ch"], (tag, g, ref)
        w = kw.get("weights")
        rr = oracle.range_stats(pcm, fs, GATES[gate], **kw)
        assert abs(g["loudness_range_lu"] - rr["lra"]) < 1e-9, (tag, g, rr)
        if math.isfinite(rr["max_momentary"]):
            assert abs(g["max_momentary_lufs"] - rr["max_momentary"]) < 1e-6, (tag, g, rr)
