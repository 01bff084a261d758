"""Host code under AddressSanitizer + UndefinedBehaviorSanitizer.  No GPU.

* `make -C headroom_amd/csrc asan` builds the host-only product sources -- container probe / loader (hr_ingest.cpp), the
  file half of the lossless gain (hr_gainfile.cpp), the analyzer's scalar logic (hr_analyzer.cpp), the rate tables
  (hr_tables.cpp) -- with tests/host/asan_harness.cpp; its `fuzz` mode drives hr_pcm_probe / hr_pcm_load_file /
  hr_pcm_gain_file on mutated WAV / AIFF files and checks the never-half-written contract of src/processor.rs:58,80-86
  (a rejected file is byte-identical afterwards, no temp file stays behind).
* `make -C oracle asan` runs the oracle's entry points on exact-size buffers of awkward geometry.
"""
import os
import subprocess

import numpy as np

import containers

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# (the harness of this image preloads a library of its own: the sanitizer runtime must simply not insist on being first)
ENV = dict(os.environ, ASAN_OPTIONS="verify_asan_link_order=0:detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")


def run(cmd, **kw):
    p = subprocess.run(cmd, env=ENV, capture_output=True, text=True, timeout=600, **kw)
    assert p.returncode == 0, (cmd, p.stdout[-2000:], p.stderr[-6000:])
    assert "ERROR: AddressSanitizer" not in p.stderr and "runtime error" not in p.stderr, p.stderr[-6000:]
    return p.stdout


def test_host_sources_under_asan_fuzz_and_logic(tmp_path):
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "headroom_amd", "csrc"), "asan"], stdout=subprocess.DEVNULL)
    harness = os.path.join(ROOT, "tests", "host", "asan_harness")
    rng = np.random.default_rng(3)
    pcm = rng.uniform(-0.5, 0.5, size=(37, 2))
    pcm6 = rng.uniform(-0.5, 0.5, size=(19, 6))
    n = 0
    for kind in ("s16", "s24", "s32", "f32", "f64"):
        for data in (containers.wav_bytes(pcm, 44100, kind)[0],
                     containers.wav_bytes(pcm6, 96000, kind, extensible_mask=0x3F, extra_chunks=[(b"LIST", b"abcde"), (b"bext", b"x" * 33)])[0]):
            (tmp_path / f"seed{n}.wav").write_bytes(data)
            n += 1
    for kind, aifc in (("s16", None), ("s24", None), ("s16", b"sowt"), ("f32", b"fl32"), ("f64", b"fl64"), ("s24", b"NONE")):
        for data in (containers.aiff_bytes(pcm, 48000, kind, aifc=aifc)[0],
                     containers.aiff_bytes(pcm, 48000, kind, aifc=aifc, ssnd_offset=6, ssnd_first=True)[0]):
            (tmp_path / f"seed{n}.aiff").write_bytes(data)
            n += 1
    out = run([harness, "fuzz", str(tmp_path), "6000", "20261111"])
    assert "every rejected file intact" in out, out
    # (the unmutated seeds are among the cases: some files must have made it through all three entry points)
    ok = int(out.split("gain ok ")[1].split(";")[0])
    assert ok > 100, out
    out = run([harness, "logic"])
    assert "no sanitizer report" in out


def test_oracle_under_asan():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], stdout=subprocess.DEVNULL)
    out = run([os.path.join(ROOT, "oracle", "_asan", "oracle_asan")])
    assert "no report" in out
