"""Soak run for the fused kernel: long stereo f32 tracks (several waves per track) of awkward signals -- clipped and
limited material (thousands of samples at the ceiling: candidate lists overflow), full-scale sines and squares, sweeps,
impulses, DC -- each compared with the oracle.  python tools/soak_long.py FIRST_SEED N_SEEDS"""
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

import headroom_amd as hr  # noqa: E402
import oracle_lib  # noqa: E402

KINDS = ["clipped", "limited", "sine_fs", "square", "sweep", "impulses", "dc_small", "noise"]


def make(rng, kind):
    fs = int(rng.choice([44100, 44100, 48000, 96000, 88200]))
    secs = float(rng.uniform(13.0, 45.0))
    n = int(fs * secs) + int(rng.integers(0, 999))
    t = np.arange(n) / fs
    x = rng.uniform(-1.0, 1.0, size=(n, 2))
    if kind == "clipped":
        x = np.clip(x * rng.uniform(1.2, 4.0), -1.0, 1.0) * rng.uniform(0.5, 0.999)
    elif kind == "limited":
        env = 0.6 + 0.4 * np.sin(2 * np.pi * 0.3 * t)[:, None]
        x = np.tanh(3.0 * x * env) * 0.98
    elif kind == "sine_fs":
        f = rng.uniform(30.0, 0.45 * fs)
        x = np.stack([np.sin(2 * np.pi * f * t), np.sin(2 * np.pi * f * t + 0.7)], axis=1) * 0.999
    elif kind == "square":
        f = rng.uniform(50.0, 5000.0)
        x = np.sign(np.sin(2 * np.pi * f * t))[:, None] * np.array([0.9, 0.5])[None, :]
    elif kind == "sweep":
        f1 = rng.uniform(2000.0, 0.49 * fs)
        ph = 2 * np.pi * (20.0 * t + (f1 - 20.0) * t * t / (2 * secs))
        x = np.stack([np.sin(ph), np.cos(ph)], axis=1) * 0.8
    elif kind == "impulses":
        x = np.zeros((n, 2))
        idx = rng.integers(0, n, size=200)
        x[idx, rng.integers(0, 2, size=200)] = rng.uniform(-1, 1, size=200)
    elif kind == "dc_small":
        x = 0.5 + x * 10.0 ** (-rng.uniform(40, 80) / 20.0)
    else:
        x = x * 10.0 ** (-rng.uniform(3, 40) / 20.0)
    return x.astype(np.float32), fs, (kind, fs, n)


def main():
    first, count = int(sys.argv[1]), int(sys.argv[2])
    oracle_lib.build()
    t0 = time.time()
    for seed in range(first, first + count):
        rng = np.random.default_rng(seed)
        cases = [make(rng, k) for k in KINDS]
        gate = seed % 3
        for tp in (0, 1, 2):
            m = hr.Meter(device=0, gate_mode=gate, tp_mode=tp, engine=(0, 2, 3, 4)[(seed + gate + tp) % 4])  # row lengths of the fused kernel
            try:
                got = m.analyze_batch([hr.TrackSpec(p, fs) for p, fs, _ in cases])
                first = m.loudnorm_first_pass([hr.TrackSpec(p[: fs * 6], fs) for p, fs, _ in cases]) if tp == 0 else None
            finally:
                m.close()
            for (pcm, fs, tag), g in zip(cases, got):
                ref = oracle_lib.measure(pcm, fs, gate_mode=gate, tp_mode=tp)
                check(g, ref, (seed, gate, tp, tag))
            if first:
                for (pcm, fs, tag), g in zip(cases, first):
                    ref = oracle_lib.loudnorm_measure(pcm[: fs * 6], fs)
                    # libebur128's 0.1 LU histogram makes the result a step function of the block energies, and the
                    # oracle's 4th-order direct form II differs from the GPU's two biquads by ~1e-8 at 192 kHz: a block
                    # that sits on a bin boundary moves the mean by 0.1 LU / blocks (seed 133: 1.7e-3 LU, 57 blocks)
                    check(g, ref, (seed, "loudnorm first pass", tag), tol=0.11 / max(1, ref["n_blocks"]) + 1e-6)
            print(f"seed {seed} gate_mode {gate} tp_mode {tp}: {len(cases)} tracks ok, {time.time() - t0:.0f} s", flush=True)


def check(g, ref, tag, tol=1e-6):
    ok = g["status"] == 0 and g["n_blocks"] == ref["n_blocks"]
    if "n_gated" in ref:
        ok = ok and g["n_gated"] == ref["n_gated"]
    if math.isfinite(ref["integrated_lufs"]):
        ok = ok and abs(g["integrated_lufs"] - ref["integrated_lufs"]) < tol
    else:
        ok = ok and g["integrated_lufs"] == ref["integrated_lufs"]
    ok = ok and g["peak_per_ch"] == ref["peak_per_ch"]
    if not ok:
        print("MISMATCH", tag, g, ref, flush=True)
        sys.exit(1)


if __name__ == "__main__":
    main()
