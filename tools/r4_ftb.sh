#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r4_ftb; mkdir -p $out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_path.py -q -m gpu -x -k "backward or finetune" > $out/tests.log 2>&1; tail -2 $out/tests.log | cut -c1-300
echo -n "heuristic bwd tiles: "; COMIC_AUTOTUNE_BWD=0 COMIC_TUNE_CACHE=$out/t0.json N=20 timeout -k 10 300 python3 tools/ft_step_time.py 2>&1 | tail -1
echo -n "tuned bwd tiles    : "; COMIC_TUNE_CACHE=$out/t1.json N=20 timeout -k 10 400 python3 tools/ft_step_time.py 2>&1 | tail -1
echo -n "heuristic bwd tiles: "; COMIC_AUTOTUNE_BWD=0 COMIC_TUNE_CACHE=$out/t0.json N=20 timeout -k 10 300 python3 tools/ft_step_time.py 2>&1 | tail -1
echo -n "tuned bwd tiles    : "; COMIC_TUNE_CACHE=$out/t1.json N=20 timeout -k 10 400 python3 tools/ft_step_time.py 2>&1 | tail -1
python3 -c "
import json; d=json.load(open('$out/t1.json')); print({k:v for k,v in d.items() if k.endswith(':bwd')})"
