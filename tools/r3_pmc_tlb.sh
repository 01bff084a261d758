#!/bin/bash
# usage: tools/r3_pmc_tlb.sh  -- address-translation counters of the fused kernel (TCP_UTCL1_*), 200 tracks, for rows as they
# were (engine 7), shifted grids (engine 0) and, on the experiments library, strided rows without the shift
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
run() {  # tag, engine, extra env...
  tag=$1; eng=$2; shift 2
  out=$R/gpurun_out/r3/tlb_$tag
  mkdir -p $out
  env "$@" true
  ( export "$@" 2>/dev/null; timeout -s KILL 120 rocprofv3 --pmc TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum --output-format csv -d $out -o pmc -- python3 $R/bench.py --tracks 200 --steps 1 --warmup 0 --no-cpu-baseline --engine $eng > $out/bench.json 2> $out/bench.err ) || { echo "pass $tag failed"; tail -3 $out/bench.err; return 1; }
  python3 - "$out" "$tag" <<'PY'
import csv, collections, glob, sys
f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)[0]
acc = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if 'k_fused_f32s' in r['Kernel_Name']:
        acc[r['Counter_Name']] = acc.get(r['Counter_Name'], 0) + float(r['Counter_Value'])
print(sys.argv[2], ' '.join('%s=%.4g' % kv for kv in acc.items()))
PY
}
run rows_as_they_were 7 HR_DUMMY=0 && run shifted_grids 0 HR_DUMMY=0 && run strided_no_shift 0 HEADROOM_HIP_LIB=$R/headroom_amd/libheadroom_hip_exp.so HR_ALIGN_MODE=1
