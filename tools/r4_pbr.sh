#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r4_pbr; mkdir -p $out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_path.py tests/test_gpu_ops.py -q -m gpu -x -k "pool or fused_pools or weight_stationary or inception_v3_forward or stem_stream or walk" > $out/tests.log 2>&1; tail -3 $out/tests.log | cut -c1-300; grep -E "^E  " $out/tests.log | head
for r in 1 2; do
COMIC_TUNE_CACHE=$out/tiles.json B=1280 GRAPH=1 timeout -k 10 300 python3 tools/run_cnn.py 2>&1 | grep -E "cnn forward"
done
COMIC_TUNE_CACHE=$out/tiles.json B=1280 timeout -k 10 300 python3 tools/op_times.py 2>&1 | grep -E "pbr|x4  25x25|sum of"
