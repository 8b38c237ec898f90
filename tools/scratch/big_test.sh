#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/bigb; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_path.py -x -q -k "persistent_time_loop_equals or train_step_matches_oracle" > $out/tests.log 2>&1; rc=$?; tail -12 $out/tests.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python3 tools/dec_step_prof.py > $out/run.log 2>&1; tail -1 $out/run.log
COMIC_PERSIST_STAMPS=1 N=3 timeout -k 10 300 python3 tools/dec_step_prof.py 2>&1 | grep "persist stamps bwd" | tail -1
M=64 C=2048 CG=2048 timeout -k 10 300 python3 tools/dec_step_prof.py 2>&1 | tail -1
M=25 C=2048 CG=2048 timeout -k 10 300 python3 tools/dec_step_prof.py 2>&1 | tail -1
