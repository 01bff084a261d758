"""hr_apply_gain (lossless gain for PCM files, src/processor.rs:53-87) against the oracle's restatement of the FFmpeg
chain `-af volume={g}dB -c:a pcm_s24le|pcm_s24be`: byte-exact.  "Parity unpinned" with respect to FFmpeg itself."""
import numpy as np
import pytest

import headroom_amd as hr
import signals

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def meter():
    m = hr.Meter(device=0)
    yield m
    m.close()


def inputs():
    fs = 44100
    x = signals.seeded_noise(71, fs + 3, 2, -1.0)          # odd sample count: exercises the ragged tail
    x[100] = [1.0, -1.0]
    x[101] = [0.99999994, -0.99999994]
    x6 = signals.seeded_noise(72, 9001, 6, -6.0)
    yield "f32", x, {}
    yield "s16", signals.to_s16(x), {}
    yield "s24", signals.to_s24_bytes(x.astype(np.float64)), {"channels": 2, "fmt": hr.FMT_S24}
    yield "s32", (x.astype(np.float64) * 2147483647.0).astype(np.int32), {}
    yield "f64", x.astype(np.float64) * 1.0000001, {}
    yield "f32x6", x6, {}
    yield "mono1", x[:1, :1].copy(), {}


@pytest.mark.parametrize("gain_db", [0.0, 2.76, 6.0206, -3.0, 14.5])
def test_apply_gain_is_byte_exact(oracle, meter, gain_db):
    for name, pcm, kw in inputs():
        for be in (False, True):
            ref, ref_clip = oracle.apply_gain(pcm, 44100, gain_db, be, **kw)
            got, got_clip = meter.apply_gain(hr.TrackSpec(pcm, 44100, **kw), gain_db, be)
            assert got_clip == ref_clip, (name, be)
            assert np.array_equal(got, ref), (name, be, np.flatnonzero(got != ref)[:8])


def test_unity_gain_is_identity_on_24_bit_and_clip_counts(oracle, meter):
    x = signals.seeded_noise(74, 4097, 2, -3.0).astype(np.float64)
    raw = signals.to_s24_bytes(x)
    got, clipped = meter.apply_gain(hr.TrackSpec(raw, 48000, channels=2, fmt=hr.FMT_S24), 0.0)
    assert clipped == 0 and np.array_equal(got, raw)
    full = np.full((64, 2), 0.75, dtype=np.float32)
    got, clipped = meter.apply_gain(hr.TrackSpec(full, 48000), 6.0)     # 0.75 * 1.995 > 1: every sample saturates
    assert clipped == 128
    assert np.array_equal(got.reshape(-1, 3), np.tile(np.array([0xFF, 0xFF, 0x7F], np.uint8), (128, 1)))


def test_apply_gain_on_device_pcm(oracle, meter):
    import torch
    x = signals.seeded_noise(75, 48000 * 2 + 1, 2, -9.0)
    dev = torch.from_numpy(x).cuda()
    out = torch.zeros(x.size * 3 + 16, dtype=torch.uint8, device="cuda")
    _, clipped = meter.apply_gain(hr.TrackSpec(dev, 48000), 3.5, out_device_ptr=out.data_ptr())
    ref, ref_clip = oracle.apply_gain(x, 48000, 3.5)
    torch.cuda.synchronize()
    assert clipped == ref_clip and np.array_equal(out[: x.size * 3].cpu().numpy(), ref)
    assert int(out[x.size * 3:].sum()) == 0                                 # nothing written past the end


def test_apply_gain_rejects_bad_arguments(meter):
    x = np.zeros((16, 2), np.float32)
    with pytest.raises(hr.HeadroomError):
        meter.apply_gain(hr.TrackSpec(x, 48000), float("nan"))
