#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r4_p2; mkdir -p $out
cd $GRAFT_REPO_ROOT
COMIC_TUNE_CACHE=$out/tiles_fus.json B=1280 timeout -k 10 300 python3 tools/op_times.py 2>&1 | grep -v amdgpu > $out/op_times_1280.txt; cat $out/op_times_1280.txt | cut -c1-170
