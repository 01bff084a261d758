"""examples/headroom_scan (a C++ host on the C ABI: load -> hr_analyze_batch -> gain decision -> CSV, optionally the
lossless gain) on BASELINE configs[0]: a 3-minute 44.1 kHz stereo synthetic WAV, beside a 5.1 AIFF, a silent file and
a file that is not audio.  The expected CSV comes from the oracle's measurement and the reference's report format."""
import os
import subprocess

import numpy as np
import pytest

import containers
import headroom_amd as hr
import signals

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "examples", "headroom_scan")


def expected_row(oracle, name, seen, fs, **kw):
    ref = oracle.measure(seen, fs, **kw)
    rc, a = hr.gain_decision_for_result(ref["integrated_lufs"], ref["true_peak_dbtp"])
    assert rc == 0
    return hr.report_csv([(name, a)]).splitlines()[1], a


def test_headroom_scan_reports_and_applies(oracle, tmp_path):
    if not os.path.exists(EXE):  # normally built by __graft_entry__.build() and shipped with the tree
        subprocess.call(["make", "-C", os.path.dirname(EXE)], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    if not os.path.exists(EXE):
        pytest.skip("examples/headroom_scan is not built")
    fs = 44100
    x = oracle.synth_track(20261111, fs, 2, fs * 180).astype(np.float64)          # configs[0]
    raw, seen = containers.wav_bytes(x, fs, "s16")
    (tmp_path / "three minutes.wav").write_bytes(raw)
    x6 = signals.seeded_noise(7, 48000 * 4, 6, -20.0).astype(np.float64)
    raw6, seen6 = containers.aiff_bytes(x6, 48000, "s24")
    (tmp_path / "surround.aif").write_bytes(raw6)
    rawz, _ = containers.wav_bytes(np.zeros((fs, 2)), fs, "s16")
    (tmp_path / "silence.wav").write_bytes(rawz)
    (tmp_path / "notes.wav").write_bytes(b"this is not a RIFF file")
    files = [str(tmp_path / n) for n in ("three minutes.wav", "surround.aif", "silence.wav", "notes.wav")]

    p = subprocess.run([EXE] + files, capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    row1, a1 = expected_row(oracle, "three minutes.wav", seen, fs)
    row2, a2 = expected_row(oracle, "surround.aif", seen6, 48000, channels=6, fmt=oracle.FMT_S24)
    lines = p.stdout.splitlines()
    assert lines[0].startswith("Filename,Format,Bitrate (kbps),LUFS")
    assert lines[1:] == [row1, row2], (lines, row1, row2)
    assert "silence.wav: Non-finite loudness measurement" in p.stderr          # src/analyzer.rs:284-290
    assert "notes.wav" in p.stderr                                              # skipped with a warning, not fatal

    # --apply: the lossless gain brings the true peak to the target, and a second scan finds nothing left to do
    assert a1["gain_method"] == hr.GAIN_FFMPEG_LOSSLESS
    p = subprocess.run([EXE, "--apply", files[0]], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and "applied" in p.stderr, p.stderr
    p = subprocess.run([EXE, files[0]], capture_output=True, text=True, timeout=120)
    cols = p.stdout.splitlines()[1].split(",")
    assert cols[1] == "-" and cols[7] == "none", cols                           # headroom below MIN_EFFECTIVE_GAIN now
    assert abs(float(cols[4]) - float(cols[5])) <= 0.1                          # true peak sits at the target
