"""Sustained tones make every crest a candidate of the fused kernel's f16 screen: a wave's candidate list runs out of
slots, the wave is flagged, and k_true_peak_rows redoes exactly the positions that wave screened (the rest of the track
keeps its screened result).  Results stay bit-identical to the oracle -- whole-track tones, a tone that fills only some
waves, a tone that runs into the end of the track, time shards (halos), integer formats, both interpolation banks, every
row length of the fused kernel.  Oracle: oracle/hr_oracle.c (parity unpinned against FFmpeg itself, see its header).
"""
import numpy as np
import pytest

import headroom_amd as hr
import signals
from test_gpu_parity import GATES, TPS, check_against_oracle, meters  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu


def tone(fs, n, f0=997.0, f1=3001.0, amp=0.999):
    t = np.arange(n) / fs
    return (np.stack([np.sin(2 * np.pi * f0 * t), np.sin(2 * np.pi * f1 * t + 1.0)], axis=1) * amp).astype(np.float32)


def noise_with_tone(fs, seconds, t0, t1, seed, ragged=333):
    n = int(fs * seconds) + ragged
    x = signals.seeded_noise(seed, n, 2, -26.0)
    a, b = int(t0 * fs), min(int(t1 * fs), n)
    x[a:b] = tone(fs, b - a)
    return x


@pytest.mark.parametrize("engine", [0, 2, 3])
@pytest.mark.parametrize("tp", ["swr192k", "bs1770"])
@pytest.mark.parametrize("fs", [44100, 48000, 88200, 96000, 32000])
def test_a_tone_that_fills_some_waves(oracle, meters, fs, tp, engine):
    # 23 s: the tone holds seconds 7.3 .. 16.9, i.e. at least one whole 64-hop wave and parts of its neighbours
    pcm = noise_with_tone(fs, 23, 7.3, 16.9, 11 + fs)
    check_against_oracle(oracle, meters("ebur128", tp, engine), pcm, fs, tp=tp)
    # the largest value sits in the screened part: a spike in the noise above the tone's crests
    pcm[int(2.5 * fs) + 7, 0] = 1.0
    pcm[int(20.1 * fs) + 3, 1] = -1.0
    check_against_oracle(oracle, meters("ebur128", tp, engine), pcm, fs, tp=tp)


@pytest.mark.parametrize("engine", [0, 3])
@pytest.mark.parametrize("ragged", [0, 5, 40, 700, 3001])
@pytest.mark.parametrize("fs", [44100, 48000])
def test_a_tone_that_runs_into_the_end_of_the_track(oracle, meters, fs, ragged, engine):
    pcm = noise_with_tone(fs, 9, 1.2, 99.0, 5 + ragged, ragged=ragged)
    check_against_oracle(oracle, meters("ebur128", "swr192k", engine), pcm, fs)
    check_against_oracle(oracle, meters("ebur128", "swr192k", engine), tone(fs, 3 * fs // 10 + ragged), fs)  # three hops of tone


@pytest.mark.parametrize("kind", ["s16", "s24"])
@pytest.mark.parametrize("fs", [44100, 48000])
def test_integer_tracks_with_a_tone(oracle, meters, fs, kind):
    pcm = noise_with_tone(fs, 15, 0.0, 8.1, 77)
    if kind == "s16":
        check_against_oracle(oracle, meters(), signals.to_s16(pcm), fs)
    else:
        check_against_oracle(oracle, meters(), signals.to_s24_bytes(pcm), fs, channels=2, fmt=hr.FMT_S24)


def test_a_batch_where_only_some_tracks_overflow(oracle, meters):
    fs = 44100
    pcms = [signals.seeded_noise(1, fs * 8 + 10, 2, -12.0), tone(fs, fs * 8 + 99), noise_with_tone(fs, 14, 3.0, 11.0, 3),
            signals.seeded_noise(2, fs * 3, 2, -20.0), tone(fs, fs * 7, 440.0, 441.0, 0.5)]
    got = meters().analyze_batch([hr.TrackSpec(p, fs) for p in pcms])
    for g, p in zip(got, pcms):
        ref = oracle.measure(p, fs, gate_mode=0, tp_mode=0)
        assert g["status"] == 0 and g["peak_per_ch"] == ref["peak_per_ch"]
        assert abs(g["integrated_lufs"] - ref["integrated_lufs"]) < 1e-6


@pytest.mark.parametrize("fs", [44100, 48000])
def test_time_shards_of_a_tone(oracle, meters, fs):
    """Shards bring halos (the neighbour's 16 frames on either side): the redo reads them as the screen does."""
    hop = fs // 10
    pcm = tone(fs, hop * 95 + 1234)
    ms = [hr.Meter() for _ in range(3)]
    try:
        ref = oracle.measure(pcm, fs, gate_mode=0, tp_mode=0)
        got = hr.analyze_stream_sharded(ms, hr.TrackSpec(pcm, fs))
        assert got["peak_per_ch"] == ref["peak_per_ch"]
        assert abs(got["integrated_lufs"] - ref["integrated_lufs"]) < 1e-6
    finally:
        for m in ms:
            m.close()


def test_tones_through_stereo_planes(oracle, meters):
    """Mono and 5.1 tracks are measured as stereo planes: a plane's overflowed waves are redone on the plane."""
    fs = 48000
    n = fs * 8 + 321
    t = np.arange(n) / fs
    mono = (0.98 * np.sin(2 * np.pi * 1234.5 * t)).astype(np.float32)[:, None]
    check_against_oracle(oracle, meters(), mono, fs)
    six = signals.seeded_noise(9, n, 6, -30.0)
    six[:, 2] = 0.97 * np.sin(2 * np.pi * 777.0 * t)
    six[:, 5] = 0.9 * np.sin(2 * np.pi * 2500.0 * t + 0.3)
    check_against_oracle(oracle, meters(), six, fs)


def test_a_tone_through_the_entry_point_that_allocates_nothing(oracle):
    """hr_analyze_device_noalloc redoes overflowed waves too (its work items come out of the static device array)."""
    import torch
    fs = 44100
    pcm = noise_with_tone(fs, 12, 2.0, 99.0, 31)
    ref = oracle.measure(pcm, fs, gate_mode=0, tp_mode=0)
    dev = torch.from_numpy(pcm).cuda()
    with hr.Meter(device=0) as m:
        assert m.analyze_batch([hr.TrackSpec(dev, fs)])[0]["peak_per_ch"] == ref["peak_per_ch"]  # (makes the rate's tables resident)
        for _ in range(2):
            rc, got = m.analyze_device_noalloc(hr.TrackSpec(dev, fs))
            assert rc == hr.OK
            assert got["peak_per_ch"] == ref["peak_per_ch"]
            assert abs(got["integrated_lufs"] - ref["integrated_lufs"]) < 1e-6
