#!/bin/bash
set -o pipefail
out=$GRAFT_REPO_ROOT/gpurun_out/r4_ft; mkdir -p $out
export TMPDIR=/tmp
cd /tmp; rm -rf /tmp/kt
N=6 timeout -k 10 400 rocprofv3 --kernel-trace --stats -d /tmp/kt -o b --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/ft_step_time.py > $out/ft_prof_final.log 2>&1 || { tail -20 $out/ft_prof_final.log; exit 1; }
cp /tmp/kt/b_kernel_stats.csv $out/finetune_kernel_stats_final.csv
python3 $GRAFT_REPO_ROOT/tools/ft_lanes.py /tmp/kt/b_kernel_trace.csv > $out/ft_lanes_final.txt; head -8 $out/ft_lanes_final.txt
head -12 $out/finetune_kernel_stats_final.csv | cut -c1-150
