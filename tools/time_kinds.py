"""Fused-pass time per signal kind (device-resident stereo f32, 44.1 kHz, 32 tracks x 60 s each): how much slower awkward
material (clipped / limited: many outputs near the ceiling, candidate lists fill up) is than the bench's noise."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import headroom_amd as hr  # noqa: E402

fs, secs, ntr = 44100, 60, 32
n = fs * secs
rng = np.random.default_rng(7)
t = np.arange(n) / fs


def gen(kind):
    x = rng.uniform(-1.0, 1.0, size=(n, 2))
    if kind == "noise":
        x *= 0.5
    elif kind == "clip_x1.5":
        x = np.clip(x * 1.5, -1, 1) * 0.98
    elif kind == "clip_x4":
        x = np.clip(x * 4.0, -1, 1) * 0.98
    elif kind == "tanh":
        x = np.tanh(3.0 * x) * 0.98
    elif kind == "sine_fs":
        x = np.stack([np.sin(2 * np.pi * 997.0 * t), np.sin(2 * np.pi * 1499.0 * t)], axis=1) * 0.999
    elif kind == "square":
        x = np.sign(np.sin(2 * np.pi * 440.0 * t))[:, None] * np.array([0.9, 0.9])[None, :]
    return x.astype(np.float32)


for kind in ["noise", "clip_x1.5", "clip_x4", "tanh", "sine_fs", "square"]:
    base = torch.from_numpy(gen(kind)).cuda()
    bufs = [base.clone() for _ in range(ntr)]
    torch.cuda.synchronize()
    m = hr.Meter(device=0)
    specs = [hr.TrackSpec(b.data_ptr(), fs, channels=2, fmt=hr.FMT_F32, frames=n) for b in bufs]
    for _ in range(3):
        res = m.analyze_batch(specs)
    a, b = m.last_timing()
    print(f"{kind:10s} total {a:.3f} ms fused {b:.3f} ms  ({ntr * secs / 3600 / (a / 1e3):.0f} audio-hours/s)  tp {res[0]['true_peak_dbtp']:.3f}", flush=True)
    m.close()
