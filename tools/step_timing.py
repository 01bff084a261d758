import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import headroom_amd as hr
FS, CH, SECONDS = 44100, 2, 180
frames = FS * SECONDS
n = 1000
m = hr.Meter(device=0)
pcm = torch.empty((n, frames, CH), dtype=torch.float32, device="cuda")
for i in range(n):
    m.synth_fill(pcm[i].data_ptr(), 1000 + i, FS, CH, frames)
specs = [hr.TrackSpec(pcm[i].data_ptr(), FS, channels=CH, fmt=hr.FMT_F32, frames=frames) for i in range(n)]
m.analyze_batch(specs)
for _ in range(5):
    t0 = time.perf_counter()
    m.analyze_batch(specs)
    dt = (time.perf_counter() - t0) * 1e3
    tot, main = m.last_timing()
    print(f"wall {dt:.2f} ms  ev0->ev1 {tot:.2f} ms  fused {main:.2f} ms  outside events {dt - tot:.2f} ms")
import ctypes as C
L = hr.lib()
arr = (hr.HrTrack * n)(*[s.c for s in specs])
res = (hr.HrResult * n)()
for _ in range(4):
    t0 = time.perf_counter()
    arr2 = (hr.HrTrack * n)(*[s.c for s in specs])
    t1 = time.perf_counter()
    L.hr_analyze_batch(m._h, arr, n, res)
    t2 = time.perf_counter()
    d = [hr._result_dict(res[i], 2) for i in range(n)]
    t3 = time.perf_counter()
    tot, main = m.last_timing()
    print(f"marshal {1e3*(t1-t0):.2f}  C call {1e3*(t2-t1):.2f} (events {tot:.2f})  dicts {1e3*(t3-t2):.2f} ms")
