#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/ftprof; mkdir -p $out
export TMPDIR=/tmp
cd /tmp
export COMIC_TUNE_CACHE=$out/tiles.json
timeout -k 10 300 python3 $GRAFT_REPO_ROOT/tools/ft_step_prof.py > $out/run.log 2>&1; tail -1 $out/run.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/fp -o fp --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/ft_step_prof.py > $out/prof.log 2>&1
cp /tmp/fp/*kernel_stats.csv $out/kernel_stats.csv; ls $out
