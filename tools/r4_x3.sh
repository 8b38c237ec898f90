#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r4_x3; mkdir -p $out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_path.py tests/test_gpu_ops.py -q -m gpu -s -k "bf16x3 or gradient_clipping or sampled_decode or inception_v3_forward_224 or fused_pools or image_resident" > $out/tests.log 2>&1; tail -3 $out/tests.log | cut -c1-300; grep -E "bf16x3 worst|^FAILED|^ERROR|^E  " $out/tests.log | head -20
export COMIC_TUNE_CACHE=$out/tiles.json
for nb in 64 1280; do
X3=1 COMIC_POOL_REWRITE=0 B=$nb GRAPH=1 timeout -k 10 300 python3 tools/run_cnn.py 2>&1 | grep "cnn forward"
done
B=1280 GRAPH=1 timeout -k 10 300 python3 tools/run_cnn.py 2>&1 | grep "cnn forward"
