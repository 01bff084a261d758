"""hr_loudnorm_first_pass with k_first_pass_rows (whole 192 kHz hops measured straight from the input PCM, only the end of
the track resampled into memory) against the old path (engine = 1: the whole stream resampled into memory and measured
by the generic kernels) and the oracle's hro_loudnorm_measure: peaks bit-identical (same FMA chain), loudness within
1e-6 LU of the oracle and within 1e-9 LU of the old path (same operations per hop, in the same order).
Oracle: oracle/hr_oracle.c, parity unpinned against FFmpeg itself (see its header)."""
import math

import numpy as np
import pytest

import headroom_amd as hr
import signals

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def both():
    new, old = hr.Meter(device=0), hr.Meter(device=0, engine=1)
    yield new, old
    new.close()
    old.close()


def check(oracle, both, cases):
    new, old = both
    specs = [hr.TrackSpec(p, fs, **kw) for p, fs, kw in cases]
    got = new.loudnorm_first_pass(specs)
    was = old.loudnorm_first_pass(specs)
    for (p, fs, kw), g, w in zip(cases, got, was):
        ref = oracle.loudnorm_measure(p, fs, **kw)
        assert g["status"] == w["status"] == 0
        assert g["n_blocks"] == w["n_blocks"] == ref["n_blocks"] and g["n_gated"] == w["n_gated"], (fs, g, w)
        assert g["peak_per_ch"] == w["peak_per_ch"] == ref["peak_per_ch"], (fs, g, ref)
        for key in ("integrated_lufs", "loudness_range_lu", "max_momentary_lufs", "max_shortterm_lufs"):
            if math.isfinite(w[key]):
                assert abs(g[key] - w[key]) < 1e-9, (fs, key, g, w)
            else:
                assert g[key] == w[key] or (math.isnan(g[key]) and math.isnan(w[key]))
        if math.isfinite(ref["integrated_lufs"]):
            assert abs(g["integrated_lufs"] - ref["integrated_lufs"]) < 1e-6, (fs, g, ref)


@pytest.mark.parametrize("fs", [44100, 48000, 88200, 96000, 176400, 128000, 32000, 8000])  # (176.4 / 128 kHz: fewer than two outputs per input position)
def test_rates(oracle, both, fs):
    secs = 7.7 if fs <= 48000 else 3.3   # more than one wave of 64 hops at the common rates
    x = signals.seeded_noise(500 + fs, int(fs * secs) + 29, 2, -14.0)
    x[: fs] *= 0.02
    x[int(fs * 1.5) + 3, 0] = 0.97   # a peak in the first hops, one near the end
    x[-40, 1] = -0.99
    check(oracle, both, [(x, fs, {})])


def test_formats_channel_counts_and_lengths(oracle, both):
    fs = 44100
    cases = [(signals.to_s16(signals.seeded_noise(1, fs * 3 + 1, 1, -9.0)), fs, {}),
             (signals.to_s24_bytes(signals.seeded_noise(2, fs * 2 + 77, 2, -11.0)), fs, {"channels": 2, "fmt": hr.FMT_S24}),
             (signals.seeded_noise(3, 48000 * 3 + 5, 6, -12.0), 48000, {}),
             (signals.seeded_noise(4, fs * 2, 2, -12.0).astype(np.float64), fs, {}),
             (signals.seeded_noise(5, 2 * 4410 + 64, 2, -12.0), fs, {}),        # the shortest track that takes the new kernel
             (signals.seeded_noise(6, 2 * 4410 + 63, 2, -12.0), fs, {}),        # one frame less: old path throughout
             (signals.seeded_noise(7, 4410 * 5 + 16, 2, -12.0), fs, {}),        # the last hop's window ends with the track
             (signals.seeded_noise(8, 4410 * 5 + 15, 2, -12.0), fs, {}),
             (signals.seeded_noise(9, 20, 2, -12.0), 48000, {}),
             (np.zeros((48000, 2), np.float32), 48000, {})]
    check(oracle, both, cases)


def test_device_pcm_and_low_frequency_content(oracle, both):
    import torch
    fs = 48000
    n = fs * 9 + 1234
    t = np.arange(n) / fs
    x = signals.seeded_noise(10, n, 2, -30.0)
    x[:, 0] += (0.4 + 0.3 * np.sin(2 * np.pi * 25.0 * t)).astype(np.float32)   # DC + 25 Hz: the carry across hops matters
    dev = torch.from_numpy(x).cuda()
    new, old = both
    ref = oracle.loudnorm_measure(x, fs)
    for m in (new, old):
        g = m.loudnorm_first_pass([hr.TrackSpec(dev, fs)])[0]
        assert g["peak_per_ch"] == ref["peak_per_ch"]
        assert abs(g["integrated_lufs"] - ref["integrated_lufs"]) < 1e-6


def test_committed_first_pass_vectors(both):
    """tests/golden/first_pass_vectors.json (generated from the oracle by make_golden_first_pass.py, no reference on the
    GPU box): same generator on the device (hr_synth_fill), both paths of hr_loudnorm_first_pass."""
    import json
    import os
    import torch
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "first_pass_vectors.json")) as fh:
        cases = json.load(fh)["cases"]
    for m in both:
        bufs, specs = [], []
        for c in cases:
            t = torch.empty((c["frames"], c["channels"]), dtype=torch.float32, device="cuda:0")
            m.synth_fill(t.data_ptr(), c["seed"], c["fs"], c["channels"], c["frames"])
            bufs.append(t)
            specs.append(hr.TrackSpec(t, c["fs"]))
        torch.cuda.synchronize()
        for c, g in zip(cases, m.loudnorm_first_pass(specs)):
            assert g["status"] == 0 and g["n_blocks"] == c["n_blocks"] and g["n_gated"] == c["n_gated"], (c["seed"], g)
            assert [float.hex(float(x)) for x in g["peak_per_ch"][: c["channels"]]] == c["peak_per_ch"], (c["seed"], g)
            want = float.fromhex(c["integrated_lufs"])
            assert g["integrated_lufs"] == want or abs(g["integrated_lufs"] - want) < 1e-6, (c["seed"], g, want)
