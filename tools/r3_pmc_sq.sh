#!/bin/bash
# usage: tools/r3_pmc_sq.sh  -- instruction counters of the fused kernel for rows as they were (engine 7), shifted grids (engine 0)
# and, on the experiments library, strided rows without the shift (HR_ALIGN_MODE=1); 200 tracks
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
run() {
  tag=$1; eng=$2; shift 2
  out=$R/gpurun_out/r3/sq_$tag
  mkdir -p $out
  ( export "$@"; timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVES SQ_WAIT_INST_ANY SQ_INSTS_VMEM --output-format csv -d $out -o pmc -- python3 $R/bench.py --tracks 200 --steps 1 --warmup 0 --no-cpu-baseline --engine $eng > $out/bench.json 2> $out/bench.err ) || { echo "pass $tag failed"; tail -3 $out/bench.err; return 1; }
  python3 - "$out" "$tag" <<'PY'
import csv, collections, glob, sys
f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)[0]
acc = collections.OrderedDict(); ms = 0
for r in csv.DictReader(open(f)):
    if 'k_fused_f32s' in r['Kernel_Name']:
        acc[r['Counter_Name']] = acc.get(r['Counter_Name'], 0) + float(r['Counter_Value'])
        ms = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
print(sys.argv[2], 'ms=%.3f' % ms, ' '.join('%s=%.5g' % kv for kv in acc.items()))
PY
}
run rows_as_they_were 7 HR_DUMMY=0 && run shifted_grids 0 HR_DUMMY=0 && run strided_no_shift 0 HEADROOM_HIP_LIB=$R/headroom_amd/libheadroom_hip_exp.so HR_ALIGN_MODE=1
