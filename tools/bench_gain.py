#!/usr/bin/env python3
"""Times hr_apply_gain on device-resident PCM (run under rocprofv3 --kernel-trace --stats for the kernel's own duration).
usage: python3 tools/bench_gain.py [f32|s16|s24]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import headroom_amd as hr

fmt = sys.argv[1] if len(sys.argv) > 1 else "f32"
frames, ch = 48000 * 3600, 2            # one hour of 48 kHz stereo
m = hr.Meter(device=0)
x = torch.empty((frames, ch), dtype=torch.float32, device="cuda")
m.synth_fill(x.data_ptr(), 5, 48000, ch, frames)
if fmt == "s16":
    src = torch.clamp(torch.round(x * 32768.0), -32768, 32767).to(torch.int16)
    spec, bps = hr.TrackSpec(src, 48000), 2
elif fmt == "s24":
    q = torch.clamp(torch.round(x * 8388608.0), -8388608, 8388607).to(torch.int32).reshape(-1)
    src = torch.stack((q & 0xff, (q >> 8) & 0xff, (q >> 16) & 0xff), dim=1).to(torch.uint8).reshape(-1)
    spec, bps = hr.TrackSpec(src.data_ptr(), 48000, channels=ch, fmt=hr.FMT_S24, frames=frames), 3
else:
    src = x
    spec, bps = hr.TrackSpec(src, 48000), 4
out = torch.empty(frames * ch * 3, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
m.apply_gain(spec, 2.5, out_device_ptr=out.data_ptr())
t0 = time.perf_counter()
reps = 10
for _ in range(reps):
    _, clipped = m.apply_gain(spec, 2.5, out_device_ptr=out.data_ptr())
dt = (time.perf_counter() - t0) / reps
n = frames * ch
print(f"hr_apply_gain {fmt}: {dt * 1e3:.3f} ms per call (host-timed, sync included), {n * (bps + 3) / dt / 1e9:.0f} GB/s read+written, "
      f"{3600 / dt / 3600:.0f} audio-hours/s, clipped {clipped}")
m.close()
