"""How fast is the literal loudnorm first pass (resample to 192 kHz f64, measure there)?
python tools/first_pass_rate.py TRACKS SECONDS [RATE [ENGINE]]   (engine 1: the all-in-memory path)"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import headroom_amd as hr
n, secs = int(sys.argv[1]), float(sys.argv[2])
fs = int(sys.argv[3]) if len(sys.argv) > 3 else 44100
engine = int(sys.argv[4]) if len(sys.argv) > 4 else 0
frames = int(fs * secs)
dev = torch.device("cuda:0")
pcm = torch.empty((n, frames, 2), dtype=torch.float32, device=dev)
m = hr.Meter(device=0, engine=engine)
for i in range(n):
    m.synth_fill(pcm[i].data_ptr(), 1000 + i, fs, 2, frames)
torch.cuda.synchronize()
specs = [hr.TrackSpec(pcm[i].data_ptr(), fs, channels=2, fmt=hr.FMT_F32, frames=frames) for i in range(n)]
m.loudnorm_first_pass(specs[:1])
m.loudnorm_first_pass(specs)
dts = []
for _ in range(3):
    t0 = time.perf_counter()
    r = m.loudnorm_first_pass(specs)
    dts.append(time.perf_counter() - t0)
dt = min(dts)
print("first pass: %d x %.0f s tracks in %.1f ms = %.1f audio-hours/s" % (n, secs, dt * 1e3, n * secs / 3600 / dt), r[0]["integrated_lufs"], r[0]["true_peak_dbtp"])
t0 = time.perf_counter()
r2 = m.analyze_batch(specs)
dt = time.perf_counter() - t0
print("ebur128 at the input rate: %.1f ms = %.1f audio-hours/s" % (dt * 1e3, n * secs / 3600 / dt), r2[0]["integrated_lufs"], r2[0]["true_peak_dbtp"])
