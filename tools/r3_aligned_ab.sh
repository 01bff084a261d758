#!/bin/bash
# A/B of the line-aligned rows (engine 0) against rows as they were (engine 7), 500 tracks of the headline workload;
# HR_ALIGN_MODE=1 (experiment): strided items without the grid shift
mkdir -p gpurun_out/r3
for cfg in "7 0" "0 0" "7 0" "0 0" "7 0" "0 0" "7 0" "0 0" "7 0" "0 0"; do
  set -- $cfg
  HR_ALIGN_MODE=$2 timeout -k 10 200 python bench.py --tracks 1000 --steps 5 --warmup 2 --no-cpu-baseline --engine $1 > gpurun_out/r3/p_engine$1_$2.json 2> gpurun_out/r3/p_engine$1_$2.err || { tail -5 gpurun_out/r3/p_engine$1_$2.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/r3/p_engine$1_$2.json").read().strip().splitlines()[-1])
print("engine", $1, "align_mode", $2, "audio-h/s", round(d["value"],1), "ms/step", round(d["ms_per_step"],3), "kernel_ms", round(d["roofline"]["kernel_ms"],3))
PY
done
