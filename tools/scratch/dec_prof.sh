#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/decprof; mkdir -p $out
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 python3 $GRAFT_REPO_ROOT/tools/dec_step_prof.py > $out/run.log 2>&1; tail -1 $out/run.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/dp -o dp --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/dec_step_prof.py > $out/prof.log 2>&1
cp /tmp/dp/*kernel_stats.csv $out/kernel_stats.csv; ls $out
