"""5.1 / 7.1 tracks read DIRECTLY by the fused kernel (hr_kernels_fused_mc.hip, round 3) against the oracle and against the
stereo-plane path of rounds 1-2 (hr_config.engine = 5), through the C ABI.

The direct read does the arithmetic of the plane path in the same order on the same values (the planes hold the samples
exactly), so gating counts and peaks must agree BIT FOR BIT with engine 5 and the loudness figures to 1e-9 LU (identical
where both paths pick the same row length), and everything with the oracle to the bars of tests/test_gpu_parity.py (counts equal, LUFS < 1e-6, peaks bit-identical).  BASELINE
configs[3] is the 96 kHz s24 5.1 case.
"""
import math

import numpy as np
import pytest

import headroom_amd as hr
import signals

pytestmark = pytest.mark.gpu

W51 = [1, 1, 1, 0, 1.41, 1.41]
W71 = [1, 1, 1, 0, 1.41, 1.41, 1.41, 1.41]
FLOAT_KEYS = ("integrated_lufs", "true_peak_dbtp", "rel_threshold_lufs", "loudness_range_lu", "lra_low_lufs", "lra_high_lufs",
              "lra_threshold_lufs", "max_momentary_lufs", "max_shortterm_lufs")


@pytest.fixture(scope="module")
def meters():
    cache = {}

    def get(engine=0, tp=0, gate=0):
        key = (engine, tp, gate)
        if key not in cache:
            cache[key] = hr.Meter(device=0, engine=engine, tp_mode=tp, gate_mode=gate)
        return cache[key]

    yield get
    for m in cache.values():
        m.close()


def identical(a, b):
    assert a["status"] == b["status"] == 0
    assert a["n_blocks"] == b["n_blocks"] and a["n_gated"] == b["n_gated"]
    assert a["peak_per_ch"] == b["peak_per_ch"], (a, b)
    # (the host picks the rows' length per launch from the waves it counts, and a direct item is channels / 2 waves: where the
    # two paths end up with different row lengths the hop energies differ in the 13th digit -- other summation trees)
    for k in FLOAT_KEYS:
        assert a[k] == b[k] or abs(a[k] - b[k]) < 1e-9 or (math.isnan(a[k]) and math.isnan(b[k])), (k, a, b)


def against_oracle(oracle, got, pcm, fs, tp=0, gate=0, **kw):
    ref = oracle.measure(pcm, fs, gate_mode=gate, tp_mode=tp, **kw)
    assert got["n_blocks"] == ref["n_blocks"] and got["n_gated"] == ref["n_gated"], (got, ref)
    if math.isfinite(ref["integrated_lufs"]):
        assert abs(got["integrated_lufs"] - ref["integrated_lufs"]) < 1e-6, (got, ref)
    assert got["peak_per_ch"] == ref["peak_per_ch"], (got, ref)


def as_format(pcm, kind):
    if kind == "s24":
        return signals.to_s24_bytes(pcm), {"channels": pcm.shape[1], "fmt": hr.FMT_S24}
    if kind == "s16":
        return signals.to_s16(pcm), {}
    return pcm, {}


def noise_track(seed, fs, ch, seconds, ragged=0, level=-14.0):
    n = int(fs * seconds) + ragged
    pcm = signals.seeded_noise(seed, n, ch, level)
    pcm[: fs] *= 0.02                                       # a quiet start: the relative gate has something to do
    t = np.arange(n) / fs
    for c in range(ch):                                     # every channel its own level and a tone of its own
        pcm[:, c] = pcm[:, c] * (0.4 + 0.1 * c) + 0.2 * np.sin(2 * np.pi * (200.0 + 333.0 * c) * t + c)
    return pcm.astype(np.float32)


@pytest.mark.parametrize("tp", [0, 1])           # libswresample's bank to 192 kHz, BS.1770 Annex 2
@pytest.mark.parametrize("kind", ["s24", "f32", "s16"])
@pytest.mark.parametrize("fs,ch", [(96000, 6), (48000, 6), (44100, 6), (48000, 8), (96000, 8), (44100, 8)])
def test_direct_read_equals_planes_and_oracle(oracle, meters, fs, ch, kind, tp):
    w = W51 if ch == 6 else W71
    pcm = noise_track(fs + ch, fs, ch, 4.3, ragged=777)
    data, kw = as_format(pcm, kind)
    spec = hr.TrackSpec(data, fs, weights=w, **kw)
    direct = meters(0, tp).analyze_batch([spec])[0]
    planes = meters(5, tp).analyze_batch([spec])[0]
    identical(direct, planes)
    against_oracle(oracle, direct, data, fs, tp=tp, weights=w, **kw)
    if ch == 6:  # engine 0 took the four-wave workgroup (the LFE has no weight), engine 6 keeps the three-wave one: same rows
        three = meters(6, tp).analyze_batch([spec])[0]
        assert {k: direct[k] for k in FLOAT_KEYS + ("n_blocks", "n_gated", "peak_per_ch") if direct[k] == direct[k]} == \
               {k: three[k] for k in FLOAT_KEYS + ("n_blocks", "n_gated", "peak_per_ch") if three[k] == three[k]}


@pytest.mark.parametrize("kind", ["s24", "f32", "s16"])
@pytest.mark.parametrize("weights", [[1, 0, 1, 1, 1.41, 1.41], [1, 1, 1, 0.5, 1.41, 0], [1, 0, 1, 0, 1, 1.41], [0, 0, 0, 0, 0, 1],
                                     [1, 1, 1, 0.25, 1.41, 1.41], [1, 0, 0, 0, 1, 0]])
def test_four_wave_5_1_picks_the_weighted_second_channels(oracle, meters, kind, weights):
    """The four-wave 5.1 workgroup K-weights, of channels 1 / 3 / 5, the two that carry loudness weight, whichever they are; a
    layout that weights all three stays on the three-wave workgroup.  96 kHz rows and 44.1 kHz rows (odd hop bytes)."""
    for fs in (96000, 44100):
        pcm = noise_track(fs + int(sum(weights) * 100), fs, 6, 7.1, ragged=501)
        data, kw = as_format(pcm, kind)
        spec = hr.TrackSpec(data, fs, weights=weights, **kw)
        got = meters(0).analyze_batch([spec])[0]
        identical(got, meters(6).analyze_batch([spec])[0])
        identical(got, meters(5).analyze_batch([spec])[0])
        against_oracle(oracle, got, data, fs, weights=weights, **kw)


def test_config4_shape_with_every_row_length(oracle, meters):
    """BASELINE configs[3]: 96 kHz 5.1 s24.  13.3 s = 133 hops: two full waves of rows and a bit, so the default engine mixes
    whole-hop rows and rows of a sixth of a hop; engines 2 / 3 / 4 pin the choice."""
    fs = 96000
    pcm = noise_track(424242, fs, 6, 13.3, ragged=4001)
    data, kw = as_format(pcm, "s24")
    spec = hr.TrackSpec(data, fs, weights=W51, **kw)
    planes = meters(5).analyze_batch([spec])[0]
    against_oracle(oracle, planes, data, fs, weights=W51, **kw)
    for engine in (0, 2, 3, 4, 6):
        got = meters(engine).analyze_batch([spec])[0]
        assert got["peak_per_ch"] == planes["peak_per_ch"] and got["n_gated"] == planes["n_gated"]
        assert abs(got["integrated_lufs"] - planes["integrated_lufs"]) < 1e-9   # (other row lengths: other summation trees)
        assert abs(got["loudness_range_lu"] - planes["loudness_range_lu"]) < 1e-9


@pytest.mark.parametrize("kind", ["s24", "f32"])
@pytest.mark.parametrize("ragged", [0, 8, 22, 65, 333, 1001])
def test_largest_sample_near_the_end_of_a_track_read_directly(oracle, meters, kind, ragged):
    """The end of a track is where the fused kernel's final rows, the zeroed frames and the generic tail kernel meet; with
    18-byte frames the 16-byte chunk grid repeats every 8 frames."""
    fs, ch = 48000, 6
    frames = 3 * 4800 + ragged
    base = signals.seeded_noise(999 + ragged, frames, ch, -40.0)
    for dist, chan in [(1, 0), (16, ch - 1), (17, 3), (40, ch - 1), (ragged + 20, 2)]:
        pcm = base.copy()
        pcm[frames - dist, chan] = 0.9
        data, kw = as_format(pcm, kind)
        spec = hr.TrackSpec(data, fs, **kw)
        got = meters(0).analyze_batch([spec])[0]
        identical(got, meters(5).analyze_batch([spec])[0])
        against_oracle(oracle, got, data, fs, **kw)


def test_tones_overflow_the_candidate_lists_of_every_pair(oracle, meters):
    """A sustained full-scale tone lists every crest: the waves' lists overflow and k_true_peak_rows redoes their positions
    exactly, channel pair by channel pair (the LFE pair carries a tone on one channel and noise on the other)."""
    fs, ch = 48000, 6
    n = fs * 8 + 123
    t = np.arange(n) / fs
    pcm = np.zeros((n, ch), np.float32)
    pcm[:, 0] = 0.999 * np.sin(2 * np.pi * 997.0 * t)
    pcm[:, 1] = 0.5 * np.sin(2 * np.pi * 3001.0 * t + 1.0)
    pcm[:, 2] = 0.999 * np.sin(2 * np.pi * 440.0 * t + 0.3)
    pcm[:, 3] = signals.seeded_noise(5, n, 1, -20.0)[:, 0]
    pcm[:, 4] = 0.25 * np.sin(2 * np.pi * 11025.0 * t + 0.7)
    pcm[:, 5] = 0.8 * np.sin(2 * np.pi * 60.0 * t)
    for kind in ("f32", "s24"):
        data, kw = as_format(pcm, kind)
        spec = hr.TrackSpec(data, fs, weights=W51, **kw)
        got = meters(0).analyze_batch([spec])[0]
        identical(got, meters(5).analyze_batch([spec])[0])
        against_oracle(oracle, got, data, fs, weights=W51, **kw)


def test_mixed_batch_and_device_pcm(oracle, meters):
    """5.1 s24, 7.1 f32, stereo and mono tracks of three rates in one call, host and device buffers: every kernel class of
    the launch gets its own candidate lists; results in input order."""
    import torch
    specs, refs, keep = [], [], []
    for i, (fs, ch, kind) in enumerate([(96000, 6, "s24"), (48000, 2, "f32"), (48000, 8, "f32"), (44100, 6, "s16"), (44100, 1, "f32"),
                                        (96000, 6, "s24"), (48000, 6, "f32")]):
        pcm = noise_track(100 + i, fs, ch, 2.7 + 0.4 * i, ragged=13 * i)
        data, kw = as_format(pcm, kind)
        refs.append((data, fs, kw))
        if i % 2:
            dev = torch.from_numpy(np.ascontiguousarray(data).reshape(-1)).cuda()
            keep.append(dev)
            fmt = {"s24": hr.FMT_S24, "s16": hr.FMT_S16, "f32": hr.FMT_F32}[kind]
            specs.append(hr.TrackSpec(dev.data_ptr(), fs, channels=ch, fmt=fmt, frames=len(pcm)))
        else:
            specs.append(hr.TrackSpec(data, fs, **kw))
    got = meters(0).analyze_batch(specs)
    planes = meters(5).analyze_batch(specs)
    for g, p, (data, fs, kw) in zip(got, planes, refs):
        identical(g, p)
        against_oracle(oracle, g, data, fs, **kw)


def test_time_shards_of_a_5_1_stream(oracle, meters):
    """hr_analyze_stream_sharded on a 5.1 stream: the shards' rows read the halos of their neighbours in the track's own
    channel count, a shard's last hop is a fused row too."""
    fs, ch = 48000, 6
    pcm = noise_track(31337, fs, ch, 40.0, ragged=321)
    pcm[4800 * 13 - 3, 4] = 0.97          # a peak right in front of a cut
    ref = oracle.measure(pcm, fs, weights=W51)
    ms = [hr.Meter(device=0) for _ in range(3)]
    try:
        got = hr.analyze_stream_sharded(ms, hr.TrackSpec(pcm, fs, weights=W51))
    finally:
        for m in ms:
            m.close()
    whole = meters(0).analyze_batch([hr.TrackSpec(pcm, fs, weights=W51)])[0]
    assert got["peak_per_ch"] == whole["peak_per_ch"] == ref["peak_per_ch"]
    assert got["n_blocks"] == ref["n_blocks"] and got["n_gated"] == ref["n_gated"]
    assert abs(got["integrated_lufs"] - ref["integrated_lufs"]) < 1e-6 and abs(got["integrated_lufs"] - whole["integrated_lufs"]) < 1e-9


def test_noalloc_entry_reads_5_1_directly(oracle, meters):
    """hr_analyze_device_noalloc needs no plane buffer for a 5.1 track any more."""
    import torch
    fs, ch = 96000, 6
    pcm = noise_track(8080, fs, ch, 3.2, ragged=55)
    data, kw = as_format(pcm, "s24")
    dev = torch.from_numpy(data.reshape(-1)).cuda()
    spec = hr.TrackSpec(dev.data_ptr(), fs, channels=ch, fmt=hr.FMT_S24, frames=len(pcm), weights=W51)
    m = meters(0)
    ref = m.analyze_batch([spec])[0]
    rc, got = m.analyze_device_noalloc(spec)
    assert rc == hr.OK
    identical(got, ref)
    against_oracle(oracle, got, data, fs, weights=W51, **kw)
