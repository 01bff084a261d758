#!/bin/bash
# experiments library: what rows that are `stride` hops apart cost WITHOUT the grid shift (HR_ALIGN_MODE=1 HR_ALIGN_STRIDE=s),
# 500 tracks of the headline workload; engine 7 = rows as they were
mkdir -p gpurun_out/r3
export HEADROOM_HIP_LIB=$PWD/headroom_amd/libheadroom_hip_exp.so
for rep in 1 2; do
for cfg in "7 1" "0 1" "0 2" "0 8" "0 0"; do
  set -- $cfg
  HR_ALIGN_MODE=$([ $2 = 0 ] && echo 0 || echo 1) HR_ALIGN_STRIDE=$2 timeout -k 10 200 python bench.py --tracks 500 --steps 5 --warmup 2 --no-cpu-baseline --engine $1 > gpurun_out/r3/x_stride$2.json 2> gpurun_out/r3/x_stride$2.err || { tail -5 gpurun_out/r3/x_stride$2.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/r3/x_stride$2.json").read().strip().splitlines()[-1])
print("engine", $1, "stride", $2, "kernel_ms", round(d["roofline"]["kernel_ms"],3))
PY
done
done
