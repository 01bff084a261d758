"""Rows whose piece grid is shifted onto cache lines (hr_fused_core.hpp, FusedItem::phi; round 3): stereo f32 at 44.1 kHz and its
multiples, tracks of more than 512 hops in a 128-byte aligned buffer.  Hops 512 b + j + 8 r (r = 0 .. 63) form one item, every
row's 16-frame pieces start phi(j) frames before the row, the K-weighting starts phi frames into piece 0, the tile patterns start
inside their sequence.  hr_config.engine = 7 keeps every row as it was: counts and peaks must agree bit for bit, loudness to
1e-9 (the hand-over of q[0..1] to q[2..3] happens a few frames later: 15th digit), and everything with the oracle to the bars
of tests/test_gpu_parity.py."""
import math

import numpy as np
import pytest

import headroom_amd as hr
import signals

pytestmark = pytest.mark.gpu

GATES = {"ebur128": 0, "exact": 1, "loudnorm": 2}
TPS = {"swr192k": 0, "bs1770": 1, "sample": 2}
FLOAT_KEYS = ("integrated_lufs", "true_peak_dbtp", "rel_threshold_lufs", "loudness_range_lu", "lra_low_lufs", "lra_high_lufs",
              "lra_threshold_lufs", "max_momentary_lufs", "max_shortterm_lufs")


@pytest.fixture(scope="module")
def meters():
    cache = {}

    def get(engine=0, tp="swr192k", gate="ebur128"):
        key = (engine, tp, gate)
        if key not in cache:
            cache[key] = hr.Meter(device=0, engine=engine, tp_mode=TPS[tp], gate_mode=GATES[gate])
        return cache[key]

    yield get
    for m in cache.values():
        m.close()


def music_like(seed, fs, seconds, ragged=0):
    """noise whose level moves, a tone per channel, a quiet start and a few isolated samples near the top"""
    n = int(fs * seconds) + ragged
    x = signals.seeded_noise(seed, n, 2, -16.0)
    t = np.arange(n) / fs
    env = 0.35 + 0.3 * np.sin(2 * np.pi * 0.07 * t + seed) ** 2
    x *= env[:, None].astype(np.float32)
    x[:, 0] += (0.2 * np.sin(2 * np.pi * 997.0 * t)).astype(np.float32)
    x[:, 1] += (0.15 * np.sin(2 * np.pi * 3001.0 * t + 1.0)).astype(np.float32)
    x[: fs] *= 0.02
    rng = np.random.default_rng(seed)
    for p in rng.integers(fs, n - 100, size=12):
        x[p, rng.integers(0, 2)] = 0.93 + 0.005 * rng.random()
    return x.astype(np.float32)


def same(a, b):
    assert a["status"] == b["status"] == 0
    assert a["n_blocks"] == b["n_blocks"] and a["n_gated"] == b["n_gated"]
    assert a["peak_per_ch"] == b["peak_per_ch"], (a, b)
    for k in FLOAT_KEYS:
        assert a[k] == b[k] or abs(a[k] - b[k]) < 1e-9 or (math.isnan(a[k]) and math.isnan(b[k])), (k, a[k], b[k])


def against_oracle(oracle, got, pcm, fs, tp="swr192k", gate="ebur128"):
    ref = oracle.measure(pcm, fs, gate_mode=GATES[gate], tp_mode=TPS[tp])
    assert got["n_blocks"] == ref["n_blocks"] and got["n_gated"] == ref["n_gated"], (got, ref)
    assert abs(got["integrated_lufs"] - ref["integrated_lufs"]) < 1e-6, (got, ref)
    assert got["peak_per_ch"] == ref["peak_per_ch"], (got, ref)


@pytest.mark.parametrize("fs,seconds", [(44100, 60.3), (44100, 110.0), (88200, 55.0), (44100, 51.3), (44100, 51.2)])
def test_hop_energies_with_shifted_grids(oracle, meters, fs, seconds):
    """K-weighting only (the kernel without the screen): hop sums and block powers against the oracle, hop by hop; 51.2 s is
    512 hops exactly (no block fits in front of the final row), 51.3 s one block and one hop."""
    pcm = music_like(fs + int(seconds), fs, seconds, ragged=321)
    H, P = meters(0).block_energies(hr.TrackSpec(pcm, fs))
    H7, P7 = meters(7).block_energies(hr.TrackSpec(pcm, fs))
    Href = oracle.hop_sums(pcm, fs)
    Pref = oracle.block_powers(Href, fs)
    assert H.shape == Href.shape and P.shape == Pref.shape
    assert np.max(np.abs(H / Href - 1)) < 1e-9
    assert np.max(np.abs(P / Pref - 1)) < 1e-9
    assert np.max(np.abs(H / H7 - 1)) < 1e-12


@pytest.mark.parametrize("fs,seconds", [(176400, 53.0), (12345, 70.0), (22050, 60.0), (14700, 60.0)])
def test_other_rates_whose_hops_are_off_the_line_grid(oracle, meters, fs, seconds):
    """176.4 kHz (offsets 0 / 8 frames), rates libswresample interpolates between phases (12 345 Hz: K-weighting on the fused
    kernel, peaks on the generic one), 22.05 kHz (odd offsets: rows stay as they were), 14.7 kHz (hop 1470)."""
    pcm = music_like(fs, fs, seconds, ragged=19)
    got = meters(0).analyze(pcm, fs)
    same(got, meters(7).analyze(pcm, fs))
    against_oracle(oracle, got, pcm, fs)


@pytest.mark.parametrize("tp", ["swr192k", "bs1770", "sample"])
@pytest.mark.parametrize("fs,seconds", [(44100, 60.3), (44100, 154.7), (88200, 55.0)])
def test_shifted_grids_equal_plain_rows_and_oracle(oracle, meters, fs, seconds, tp):
    pcm = music_like(7 * fs + int(seconds), fs, seconds, ragged=77)
    got = meters(0, tp).analyze(pcm, fs)
    same(got, meters(7, tp).analyze(pcm, fs))
    against_oracle(oracle, got, pcm, fs, tp=tp)


def test_peaks_at_every_offset_of_a_shifted_grid(oracle, meters):
    """One isolated sample per hop of the first block, at positions that walk through the piece grid: the first and last frames
    of hops (where a shifted row's first tile and the row in front of it overlap), and everything in between."""
    fs = 44100
    hop = fs // 10
    base = signals.seeded_noise(4242, 70 * fs + 11, 2, -50.0)
    for trial, stride in enumerate((1, 7, 16)):
        pcm = base.copy()
        want = 0.5
        for h in range(1, 512, 5):
            pos = h * hop + ((h * stride + trial) % hop if stride > 1 else (h % 3) - 1)
            pcm[pos, h & 1] = want
            want += 0.0007
        got = meters(0).analyze(pcm, fs)
        same(got, meters(7).analyze(pcm, fs))
        against_oracle(oracle, got, pcm, fs)


def test_tones_overflow_the_lists_of_strided_items(oracle, meters):
    """A full-scale tone lists every crest: the waves' lists overflow, and the exact redo covers the 64 separate hops of a
    strided item."""
    fs = 44100
    n = 62 * fs + 5
    t = np.arange(n) / fs
    pcm = np.stack([0.9 * np.sin(2 * np.pi * 997.0 * t), 0.7 * np.sin(2 * np.pi * 5512.5 * t + 0.4)], axis=1).astype(np.float32)
    # the largest crests sit in single hops late in the block (a stationary tone peaks the same everywhere, and would hide a
    # redo that covers the wrong hops): hop 389 on the left, hop 470 on the right
    hop = fs // 10
    pcm[389 * hop: 390 * hop, 0] *= np.float32(1.09)
    pcm[470 * hop: 471 * hop, 1] *= np.float32(1.2)
    got = meters(0).analyze(pcm, fs)
    same(got, meters(7).analyze(pcm, fs))
    against_oracle(oracle, got, pcm, fs)


def test_device_buffers_on_and_off_the_line_grid_and_batches(oracle, meters):
    """Device-resident tracks: one on a 128-byte boundary (shifted grids), one 16 and one 64 bytes off it (rows as they were), a
    short one, in one launch with a 48 kHz track; results in input order, equal to engine 7 and the oracle."""
    import torch
    fs = 44100
    keep, specs, refs = [], [], []
    for i, (off_bytes, seconds, rate) in enumerate([(0, 70.0, fs), (16, 64.0, fs), (64, 58.0, fs), (0, 9.0, fs), (0, 61.0, 48000), (0, 120.5, fs)]):
        pcm = music_like(900 + i, rate, seconds, ragged=13 * i)
        flat = torch.zeros(pcm.size + 64, dtype=torch.float32, device="cuda")
        assert flat.data_ptr() % 256 == 0
        view = flat[off_bytes // 4: off_bytes // 4 + pcm.size]
        view.copy_(torch.from_numpy(pcm.reshape(-1)))
        keep.append(flat)
        specs.append(hr.TrackSpec(view.data_ptr(), rate, channels=2, fmt=hr.FMT_F32, frames=len(pcm)))
        refs.append((pcm, rate))
    got = meters(0).analyze_batch(specs)
    plain = meters(7).analyze_batch(specs)
    for g, p, (pcm, rate) in zip(got, plain, refs):
        same(g, p)
        against_oracle(oracle, g, pcm, rate)


@pytest.mark.parametrize("gate", ["ebur128", "loudnorm", "exact"])
def test_time_shards_of_a_44k_stream(oracle, meters, gate):
    """Shards cut at multiples of 8 hops start on the line grid and take shifted grids; the others do not."""
    fs = 44100
    pcm = music_like(31337, fs, 200.0, ragged=99)
    ref = oracle.measure(pcm, fs, gate_mode=GATES[gate])
    ms = [hr.Meter(device=0, gate_mode=GATES[gate]) for _ in range(3)]
    try:
        got = hr.analyze_stream_sharded(ms, hr.TrackSpec(pcm, fs))
    finally:
        for m in ms:
            m.close()
    whole = meters(0, gate=gate).analyze(pcm, fs)
    assert got["peak_per_ch"] == whole["peak_per_ch"] == ref["peak_per_ch"]
    assert got["n_blocks"] == ref["n_blocks"] and got["n_gated"] == ref["n_gated"] == whole["n_gated"]
    assert abs(got["integrated_lufs"] - ref["integrated_lufs"]) < 1e-6 and abs(got["integrated_lufs"] - whole["integrated_lufs"]) < 1e-9


def test_planes_of_other_layouts_and_the_no_alloc_entry(oracle, meters):
    """Mono and 3-channel 44.1 kHz tracks are split into stereo f32 planes, which take shifted grids like any stereo track; the
    entry point that allocates nothing measures a long stereo track the same way as hr_analyze_batch."""
    import torch
    fs = 44100
    base = music_like(5150, fs, 58.0, ragged=41)
    mono = np.ascontiguousarray(base[:, :1])
    three = np.ascontiguousarray(np.concatenate([base, base[::-1, :1] * np.float32(0.5)], axis=1))
    for pcm in (mono, three):
        got = meters(0).analyze(pcm, fs)
        same(got, meters(7).analyze(pcm, fs))
        ref = oracle.measure(pcm, fs)
        assert got["n_blocks"] == ref["n_blocks"] and got["n_gated"] == ref["n_gated"]
        assert abs(got["integrated_lufs"] - ref["integrated_lufs"]) < 1e-6 and got["peak_per_ch"] == ref["peak_per_ch"]
    dev = torch.from_numpy(base.reshape(-1)).cuda()
    spec = hr.TrackSpec(dev.data_ptr(), fs, channels=2, fmt=hr.FMT_F32, frames=len(base))
    m = meters(0)
    want = m.analyze_batch([spec])[0]
    rc, got = m.analyze_device_noalloc(spec)
    assert rc == hr.OK
    same(got, want)  # (the batch call may give the track's last hops rows of a third of a hop, this entry never does: 13th digit)
    against_oracle(oracle, got, base, fs)
