"""File in -> LUFS / dBTP out without ffmpeg for the PCM containers (hr_pcm_load_file + hr_analyze_batch), against
the oracle on the samples the file holds."""
import numpy as np
import pytest

import containers
import headroom_amd as hr
import signals

pytestmark = pytest.mark.gpu


def _check(oracle, m, path, seen, fs, ch, fmt, weights=None):
    f = hr.LoadedFile(str(path))
    try:
        assert (f.c.sample_rate, f.c.channels, f.c.fmt) == (fs, ch, fmt)
        got = m.analyze_batch([f])[0]
    finally:
        f.close()
    kw = {"channels": ch, "fmt": fmt} if fmt == hr.FMT_S24 else {}
    ref = oracle.measure(seen, fs, weights=weights, **kw)
    assert got["status"] == 0 and got["n_blocks"] == ref["n_blocks"] and got["n_gated"] == ref["n_gated"]
    assert abs(got["integrated_lufs"] - ref["integrated_lufs"]) < 1e-6
    assert got["peak_per_ch"] == ref["peak_per_ch"]


@pytest.mark.parametrize("kind,fmt", [("s16", hr.FMT_S16), ("s24", hr.FMT_S24), ("f32", hr.FMT_F32)])
def test_wav_files(oracle, tmp_path, kind, fmt):
    fs = 44100
    pcm = signals.seeded_noise(41, fs * 3 + 99, 2, -15.0).astype(np.float64)
    data, seen = containers.wav_bytes(pcm, fs, kind)
    p = tmp_path / f"t_{kind}.wav"
    p.write_bytes(data)
    with hr.Meter() as m:
        _check(oracle, m, p, seen, fs, 2, fmt)


def test_wav_5_1_extensible_uses_the_channel_mask(oracle, tmp_path):
    fs = 48000
    pcm = signals.seeded_noise(42, fs * 2 + 10, 6, -18.0).astype(np.float64)
    data, seen = containers.wav_bytes(pcm, fs, "s24", extensible_mask=0x60F)  # FL FR FC LFE SL SR
    p = tmp_path / "t51.wav"
    p.write_bytes(data)
    with hr.Meter() as m:
        _check(oracle, m, p, seen, fs, 6, hr.FMT_S24, weights=np.array([1, 1, 1, 0, 1.41, 1.41], dtype=np.float32))


def test_channel_mask_weights_cover_ffmpegs_back_mask(tmp_path):
    """Every speaker of f_ebur128's BACK_MASK gets 1.41 (BL BR BC SL SR TBL TBC TBR), LFE 0, the rest 1."""
    fs = 48000
    pcm = signals.seeded_noise(45, 4800, 8, -20.0).astype(np.float64)
    # FL FR FC LFE BC(0x100) SL SR TBC(0x10000); then FL FR BL BR TFL(0x1000) TFR(0x4000) TBL(0x8000) TBR(0x20000)
    for mask, want in ((0x7 | 0x8 | 0x100 | 0x600 | 0x10000, [1, 1, 1, 0, 1.41, 1.41, 1.41, 1.41]),
                       (0x3 | 0x30 | 0x1000 | 0x4000 | 0x8000 | 0x20000, [1, 1, 1.41, 1.41, 1, 1, 1.41, 1.41])):
        data, _ = containers.wav_bytes(pcm, fs, "s16", extensible_mask=mask)
        p = tmp_path / f"m{mask:x}.wav"
        p.write_bytes(data)
        f = hr.LoadedFile(str(p))
        assert [round(float(f.weights[c]), 2) for c in range(8)] == want, hex(mask)
        f.close()


@pytest.mark.parametrize("kind,aifc,fmt", [("s16", None, hr.FMT_S16), ("s24", None, hr.FMT_S24), ("s16", b"sowt", hr.FMT_S16),
                                            ("f32", b"fl32", hr.FMT_F32)])
def test_aiff_files(oracle, tmp_path, kind, aifc, fmt):
    fs = 44100
    pcm = signals.seeded_noise(43, fs * 2 + 7, 2, -13.0).astype(np.float64)
    data, seen = containers.aiff_bytes(pcm, fs, kind, aifc=aifc)
    p = tmp_path / "t.aif"
    p.write_bytes(data)
    with hr.Meter() as m:
        _check(oracle, m, p, seen, fs, 2, fmt)


def test_header_larger_than_the_probe_window_and_missing_file(oracle, tmp_path):
    fs = 48000
    pcm = signals.seeded_noise(44, fs + 5, 2, -12.0).astype(np.float64)
    data, seen = containers.wav_bytes(pcm, fs, "s16", extra_chunks=((b"JUNK", b"\0" * (3 << 20)),))
    p = tmp_path / "big_header.wav"
    p.write_bytes(data)
    with hr.Meter() as m:
        _check(oracle, m, p, seen, fs, 2, hr.FMT_S16)
    with pytest.raises(hr.HeadroomError):
        hr.LoadedFile(str(tmp_path / "nope.wav"))
    q = tmp_path / "not_audio.wav"
    q.write_bytes(b"hello world, this is not a wave file")
    with pytest.raises(hr.HeadroomError) as e:
        hr.LoadedFile(str(q))
    assert e.value.code == hr.ERR_FORMAT
