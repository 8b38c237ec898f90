#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r4_ftl; mkdir -p $out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -m pytest tests/test_gpu_path.py tests/test_gpu_cli.py tests/test_gpu_dp.py -q -m gpu -x -k "backward or finetune or scst_cli or dp" > $out/tests.log 2>&1; tail -2 $out/tests.log | cut -c1-300; grep -E "^E  " $out/tests.log | head -5
for r in 1 2; do
COMIC_TUNE_CACHE=$out/t0.json N=20 timeout -k 10 300 python3 tools/ft_step_time.py 2>&1 | tail -1
done
