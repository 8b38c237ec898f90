#!/bin/bash
set -o pipefail
out=$GRAFT_REPO_ROOT/gpurun_out/r4_ft; mkdir -p $out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -q -m gpu -x -k "finetune or backward or cnn_train or grad" > $out/tests_ft3.log 2>&1 || { tail -40 $out/tests_ft3.log; exit 1; }
tail -2 $out/tests_ft3.log
PHASES=1 timeout -k 10 300 python3 tools/ft_step_time.py 2>&1 | tail -8
