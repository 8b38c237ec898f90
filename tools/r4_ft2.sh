#!/bin/bash
set -o pipefail
out=$GRAFT_REPO_ROOT/gpurun_out/r4_ft; mkdir -p $out
export TMPDIR=/tmp COMIC_TUNE_CACHE=$out/tiles.json
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -q -m gpu -x -k "finetune or backward or cnn_train or grad" > $out/tests_ft.log 2>&1 || { tail -30 $out/tests_ft.log; exit 1; }
tail -2 $out/tests_ft.log
timeout -k 10 300 python3 tools/ft_step_time.py 2>&1 | tail -1
timeout -k 10 300 python3 tools/ft_step_time.py 2>&1 | tail -1
cd /tmp; rm -rf /tmp/kt
N=4 timeout -k 10 400 rocprofv3 --kernel-trace --stats -d /tmp/kt -o b --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/ft_step_time.py > $out/ft_prof.log 2>&1 || { tail -20 $out/ft_prof.log; exit 1; }
python3 $GRAFT_REPO_ROOT/tools/ft_lanes.py /tmp/kt/b_kernel_trace.csv | tee $out/ft_lanes2.txt
gzip -c /tmp/kt/b_kernel_trace.csv > $out/ft_kernel_trace2.csv.gz
