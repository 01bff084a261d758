#!/bin/bash
# Samples shader clock and power while the bench's hot path runs (is the kernel power-limited?).
# usage: tools/clock_watch.sh [bench args...]   (run on the GPU box)
python bench.py --steps 4000 --warmup 1 --no-cpu-baseline "$@" > /tmp/cw_bench.json 2>/tmp/cw_bench.err &
BP=$!
sleep 9; rocm-smi --showclocks --showpower 2>&1 | head -30
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
  if ! kill -0 $BP 2>/dev/null; then break; fi
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power \(W\)|Socket" | tr '\n' ' '; echo
  sleep 1
done
wait $BP
echo "bench rc=$?"; grep -o "kernel_ms[^,]*" /tmp/cw_bench.json
echo "idle:"; sleep 2; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)|Socket" | tr '\n' ' '; echo
