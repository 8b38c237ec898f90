#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/v1prof; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_path.py tests/test_gpu_cli.py -x -q -k "inception_v1 or default_backbone" > $out/tests.log 2>&1; rc=$?; tail -5 $out/tests.log; [ $rc -ne 0 ] && exit $rc
COMIC_POOL_REWRITE=0 NET=inception_v1 B=640 timeout -k 10 300 python3 tools/run_cnn.py > $out/run3.log 2>&1; tail -2 $out/run3.log | cut -c1-200
