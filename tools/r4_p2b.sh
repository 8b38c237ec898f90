#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r4_p2; mkdir -p $out
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do
echo -n "separate 1a: "; COMIC_FUSE_1A=0 COMIC_TUNE_CACHE=$out/tiles_sep.json B=1280 GRAPH=1 timeout -k 10 300 python3 tools/run_cnn.py 2>&1 | grep "cnn forward"
echo -n "fused 1a   : "; COMIC_TUNE_CACHE=$out/tiles_fus.json B=1280 GRAPH=1 timeout -k 10 300 python3 tools/run_cnn.py 2>&1 | grep "cnn forward"
done
