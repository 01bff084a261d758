#!/bin/bash
# usage: tools/r3_pmc_traffic.sh <tag> <kernel substring> <algorithmic bytes per launch> <bench args...>
# one rocprofv3 --pmc pass (memory-side request counters only) of one bench invocation -> gpurun_out/r3/<tag>/traffic.json
set -e
tag=$1; match=$2; algo=$3; shift 3
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r3/$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $out -o pmc -- python3 $R/bench.py "$@" > $out/bench.json 2> $out/bench.err
csv=$(find $out -name '*counter_collection.csv' | head -1)
python3 $R/tools/pmc_traffic.py "$csv" "$match" "$algo" $out/traffic.json "python3 bench.py $*"
