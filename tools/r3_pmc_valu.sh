#!/bin/bash
# usage: tools/r3_pmc_valu.sh <tag> <tracks>  -- one rocprofv3 --pmc pass (SQ_INSTS_VALU, GRBM_GUI_ACTIVE) of the headline bench;
# prints VALU issue slots per sample and the clock the part held -> what profiles/valu_profile.json records
set -e
tag=$1; tracks=$2
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r3/$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $out -o pmc -- python3 $R/bench.py --tracks $tracks --steps 2 --warmup 1 --no-cpu-baseline > $out/bench.json 2> $out/bench.err
python3 - "$out" "$tracks" <<'PY'
import csv, collections, glob, json, sys
f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)[0]
tracks = int(sys.argv[2])
agg = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if 'k_fused_f32s' in r['Kernel_Name']:
        a = agg.setdefault(r['Dispatch_Id'], collections.OrderedDict())
        a[r['Counter_Name']] = a.get(r['Counter_Name'], 0) + float(r['Counter_Value'])
        a['_ms'] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
last = list(agg.values())[-1]
samples = tracks * 7938000 * 2
res = {"tracks": tracks, "kernel_ms_under_pmc": last['_ms'], "SQ_INSTS_VALU": last.get('SQ_INSTS_VALU'), "GRBM_GUI_ACTIVE": last.get('GRBM_GUI_ACTIVE'),
       "valu_issue_slots_per_sample": last['SQ_INSTS_VALU'] / (samples / 64.0), "clock_observed_ghz": last['GRBM_GUI_ACTIVE'] / 8 / last['_ms'] / 1e6}
json.dump(res, open(sys.argv[1] + '/valu.json', 'w'), indent=1)
print(json.dumps(res))
PY
