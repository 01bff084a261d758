#!/bin/bash
# experiments library: order of the shifted-grid workgroups (HR_ALIGN_SUPER: chunks of four (track, block) pairs per group whose
# eight offsets are emitted together: 1 = the eight workgroups of a chunk adjacent, 8 = eight apart (one XCD), 1000000 = a run per offset)
mkdir -p gpurun_out/r3
export HEADROOM_HIP_LIB=$PWD/headroom_amd/libheadroom_hip_exp.so
for rep in 1 2; do
for sup in 1 8 2 1000000; do
  HR_ALIGN_SUPER=$sup timeout -k 10 200 python bench.py --tracks 1000 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r3/u_super$sup.json 2> gpurun_out/r3/u_super$sup.err || { tail -5 gpurun_out/r3/u_super$sup.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/r3/u_super$sup.json").read().strip().splitlines()[-1])
print("super", $sup, "audio-h/s", round(d["value"],1), "ms/step", round(d["ms_per_step"],3), "kernel_ms", round(d["roofline"]["kernel_ms"],3))
PY
done
done
