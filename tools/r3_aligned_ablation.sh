#!/bin/bash
# VERDICT r2 item 5(a), bounded: what would line-aligned PCM pieces buy the headline kernel?  The experiments library rounds
# every row's base down to a 128-byte line (HR_ABLATE bit 2048: results wrong, same instruction stream, same bytes per step).
mkdir -p gpurun_out/r3
export HEADROOM_HIP_LIB=$PWD/headroom_amd/libheadroom_hip_exp.so
for ab in 0 2048 0 2048; do
  HR_ABLATE=$ab timeout -k 10 200 python bench.py --tracks 500 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r3/g_aligned_$ab.json 2> gpurun_out/r3/g_aligned_$ab.err || { tail -5 gpurun_out/r3/g_aligned_$ab.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/r3/g_aligned_$ab.json").read().strip().splitlines()[-1])
print("HR_ABLATE", $ab, "audio-h/s", round(d["value"],1), "kernel_ms", d["roofline"]["kernel_ms"])
PY
done
