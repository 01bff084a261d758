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


# ---- file level: apply_gain_ffmpeg (src/processor.rs:53-87) ---------------------------------------------------------

def _chunks(raw, big):
    import struct
    pos, out = 12, []
    while pos + 8 <= len(raw):
        size = struct.unpack(">I" if big else "<I", raw[pos + 4:pos + 8])[0]
        out.append((raw[pos:pos + 4], raw[pos + 8:pos + 8 + size]))
        pos += 8 + size + (size & 1)
    return out


def test_gain_file_wav_in_place(oracle, meter, tmp_path):
    import containers
    x = signals.seeded_noise(81, 30001, 2, -6.0).astype(np.float64)
    bext = (b"bext", b"B" * 603)                       # odd size: padded
    lst = (b"LIST", b"INFOINAM\x06\x00\x00\x00title\x00")
    raw, seen = containers.wav_bytes(x, 44100, "s16", extra_chunks=[bext, lst])
    raw += b"id3 " + (5).to_bytes(4, "little") + b"ID3v2\x00"   # a chunk AFTER the samples
    raw = raw[:4] + (len(raw) - 8).to_bytes(4, "little") + raw[8:]
    path = tmp_path / "song.wav"
    path.write_bytes(raw)
    clipped = meter.gain_file(path, 3.25)
    ref, ref_clip = oracle.apply_gain(seen, 44100, 3.25)
    out = path.read_bytes()
    assert not (tmp_path / "song.tmp.wav").exists() and clipped == ref_clip
    assert out[:4] == b"RIFF" and int.from_bytes(out[4:8], "little") == len(out) - 8
    ck = _chunks(out, False)
    assert [t for t, _ in ck] == [b"fmt ", b"bext", b"LIST", b"data", b"id3 "]
    assert ck[1][1] == bext[1] and ck[2][1] == lst[1] and ck[4][1] == b"ID3v2"
    assert ck[3][1] == ref.tobytes()
    rc, info = hr.pcm_probe(out)
    assert rc == 0 and info["fmt"] == hr.FMT_S24 and info["frames"] == 30001 and info["channels"] == 2 and info["sample_rate"] == 44100
    # the rewritten file measures like the gained signal
    lf = hr.LoadedFile(str(path))
    got = meter.analyze_batch([lf])[0]
    lf.close()
    want = oracle.measure(ref, 44100, channels=2, fmt=oracle.FMT_S24)
    assert abs(got["integrated_lufs"] - want["integrated_lufs"]) < 1e-6 and got["peak_per_ch"] == want["peak_per_ch"]


def test_gain_file_wav_5_1_keeps_the_channel_mask(oracle, meter, tmp_path):
    import containers
    x = signals.seeded_noise(82, 4800, 6, -12.0).astype(np.float64)
    raw, seen = containers.wav_bytes(x, 48000, "f32", extensible_mask=0x60F)
    path = tmp_path / "surround.wav"
    path.write_bytes(raw)
    meter.gain_file(path, -2.0)
    out = path.read_bytes()
    rc, info = hr.pcm_probe(out)
    assert rc == 0 and info["fmt"] == hr.FMT_S24 and info["channel_mask"] == 0x60F and info["channels"] == 6
    ref, _ = oracle.apply_gain(seen, 48000, -2.0)
    assert dict(_chunks(out, False))[b"data"] == ref.tobytes()


@pytest.mark.parametrize("aifc", [None, b"sowt", b"fl32"])
def test_gain_file_aiff_in_place(oracle, meter, tmp_path, aifc):
    import containers
    x = signals.seeded_noise(83, 12345, 2, -9.0).astype(np.float64)
    kind = "f32" if aifc == b"fl32" else "s16"
    raw, seen = containers.aiff_bytes(x, 48000, kind, aifc=aifc)
    raw += b"ID3 " + (6).to_bytes(4, "big") + b"ID3v2."
    raw = raw[:4] + (len(raw) - 8).to_bytes(4, "big") + raw[8:]
    path = tmp_path / "take.aif"
    path.write_bytes(raw)
    clipped = meter.gain_file(path, 1.5)
    out = path.read_bytes()
    assert not (tmp_path / "take.tmp.aif").exists()
    assert out[:4] == b"FORM" and out[8:12] == b"AIFF" and int.from_bytes(out[4:8], "big") == len(out) - 8
    ck = _chunks(out, True)
    assert [t for t, _ in ck] == [b"COMM", b"SSND", b"ID3 "] and ck[2][1] == b"ID3v2."
    ref, ref_clip = oracle.apply_gain(seen, 48000, 1.5, True)
    assert clipped == ref_clip and ck[1][1][8:] == ref.tobytes()
    rc, info = hr.pcm_probe(out)
    assert rc == 0 and info["fmt"] == hr.FMT_S24 and info["frames"] == 12345 and info["big_endian"] == 1 and info["sample_rate"] == 48000


def test_gain_file_rejects_what_it_cannot_rewrite(meter, tmp_path):
    p = tmp_path / "not_audio.wav"
    p.write_bytes(b"RIFF\x04\x00\x00\x00WAVE")
    with pytest.raises(hr.HeadroomError):
        meter.gain_file(p, 1.0)
    with pytest.raises(hr.HeadroomError):
        meter.gain_file(tmp_path / "missing.wav", 1.0)
    assert p.read_bytes() == b"RIFF\x04\x00\x00\x00WAVE"
