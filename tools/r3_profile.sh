#!/bin/bash
# usage: tools/r3_profile.sh <tag> <bench args...>   -- kernel-trace stats of one bench invocation into gpurun_out/r3/<tag>
set -e
tag=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r3/$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o prof -- python3 $R/bench.py "$@" > $out/bench.json 2> $out/bench.err
find $out -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
