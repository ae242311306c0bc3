#!/bin/bash
# Kernel traces of the 128-graph step and of the 1024-graph (vocab 500) step, with one steady-state step of each
# printed as a timeline.  Usage: tools/trace_steps.sh <tag>; output under gpurun_out/.
set -e -o pipefail
V=${1:-t}
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d $O/tr_${V}_128 -o a --output-format csv -- python3 bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-bulk --no-configs --config3-steps 0 --profile-steps 1 > $O/tr_${V}_128.log 2>&1
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d $O/tr_${V}_1024 -o a --output-format csv -- python3 bench.py --scaling strong --steps 12 --warmup 4 --no-cpu-baseline --no-bulk --no-configs --config3-steps 0 --profile-steps 1 > $O/tr_${V}_1024.log 2>&1
for n in 128 1024; do
  T=$(find $O/tr_${V}_$n -name '*kernel_trace.csv' | head -1)
  S=$(find $O/tr_${V}_$n -name '*kernel_stats.csv' | head -1)
  python3 tools/timeline.py $T > $O/timeline_${V}_$n.txt
  cp $S $O/stats_${V}_$n.csv
  rm -rf $O/tr_${V}_$n
done
tail -n 2 $O/timeline_${V}_128.txt; tail -n 2 $O/timeline_${V}_1024.txt
