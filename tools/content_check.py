"""Mastered-music-like content through hr_analyze_batch: how many candidate groups does the true-peak screen list, do
lists overflow (whole-track fallback to the generic f64 kernel), what does a step cost?  HR_DEBUG=1 prints the list
statistics on stderr.  Usage: PYTHONPATH=. python tools/content_check.py [tracks] [seconds]"""
import os, sys, time
os.environ.setdefault("HR_DEBUG", "1")
import torch
import headroom_amd as hr

n_tracks = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seconds = int(sys.argv[2]) if len(sys.argv) > 2 else 180
fs = 44100
frames = fs * seconds
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(7)


def coloured(n, sigma):
    x = torch.randn(2, n + 7, device=dev, generator=g)
    y = torch.nn.functional.avg_pool1d(x[None], 8, 1)[0] * (8 ** 0.5)  # crude low-pass: music is not white
    return (0.5 * y[:, :n] + 0.5 * x[:, :n]) * sigma


def make(kind, n):
    if kind == "normalized":      # peak-normalised, unprocessed: one maximum
        x = coloured(n, 0.2)
        x = x / x.abs().max()
    elif kind == "clipped_light":  # ~0.1 % of samples flat-topped
        x = coloured(n, 0.3).clamp_(-0.98, 0.98)
    elif kind == "clipped_heavy":  # loudness-war master: ~2 % flat-topped
        x = coloured(n, 0.4).clamp_(-0.95, 0.95)
    elif kind == "soft_limited":   # tanh saturation: many peaks near the ceiling, none exactly on it
        x = 0.97 * torch.tanh(coloured(n, 0.6))
    elif kind == "edm":            # big bass tone + noise, clipped
        t = torch.arange(n, device=dev, dtype=torch.float32) / fs
        x = (0.7 * torch.sin(2 * torch.pi * 55.0 * t)[None] + coloured(n, 0.15)).clamp_(-0.9, 0.9)
    elif kind == "sine":           # pathological: full-scale tone
        t = torch.arange(n, device=dev, dtype=torch.float64) / fs
        x = (0.99 * torch.sin(2 * torch.pi * 997.0 * t)).float()[None].repeat(2, 1)
    else:
        raise ValueError(kind)
    return x.t().contiguous()  # interleaved [frames, 2]


kinds = sys.argv[3].split(",") if len(sys.argv) > 3 else ["normalized", "clipped_light", "clipped_heavy", "soft_limited", "edm", "sine"]
with hr.Meter(device=0) as m:
    for kind in kinds:
        base = [make(kind, frames) for _ in range(min(n_tracks, 8))]
        pcm = [base[i % len(base)] if i < len(base) else base[i % len(base)].clone() for i in range(n_tracks)]
        batch = hr.TrackBatch([hr.TrackSpec(p, fs) for p in pcm])
        sys.stderr.write(f"== {kind}\n"); sys.stderr.flush()
        m.analyze_prepared(batch)
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            m.analyze_prepared(batch)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        r = batch.results()[0]
        print(f"{kind:14s} {n_tracks} x {seconds}s: {min(ts):8.2f} ms per step, kernel {m.last_timing()[0]:.2f} ms, "
              f"{n_tracks * seconds / 3600 / (min(ts) * 1e-3):7.0f} audio-hours/s  lufs {r['integrated_lufs']:.2f} tp {r['true_peak_dbtp']:.3f}", flush=True)
        del pcm, base, batch
        torch.cuda.empty_cache()
