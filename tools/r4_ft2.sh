#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r4_ft2; mkdir -p $out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_path.py tests/test_gpu_dp.py -q -m gpu -k "backward or finetune or dp" > $out/tests.log 2>&1; tail -3 $out/tests.log | cut -c1-300
export COMIC_TUNE_CACHE=$out/tiles.json
N=20 timeout -k 10 300 python3 tools/ft_step_time.py 2>&1 | tail -1
N=20 timeout -k 10 300 python3 tools/ft_step_time.py 2>&1 | tail -1
