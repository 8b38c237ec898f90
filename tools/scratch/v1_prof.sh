#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/v1prof; mkdir -p $out
export TMPDIR=/tmp
cd /tmp
COMIC_POOL_REWRITE=0 NET=inception_v1 B=640 timeout -k 10 300 python3 $GRAFT_REPO_ROOT/tools/run_cnn.py > $out/run.log 2>&1 && tail -2 $out/run.log &&
COMIC_POOL_REWRITE=0 NET=inception_v1 B=640 COMIC_TUNE_CACHE=$out/tiles.json timeout -k 10 300 python3 $GRAFT_REPO_ROOT/tools/run_cnn.py > $out/run2.log 2>&1 &&
COMIC_POOL_REWRITE=0 NET=inception_v1 B=640 COMIC_TUNE_CACHE=$out/tiles.json timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/v1p -o v1 --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/run_cnn.py > $out/prof.log 2>&1
cp /tmp/v1p/*kernel_stats.csv $out/kernel_stats.csv; cp /tmp/v1p/*kernel_trace.csv $out/kernel_trace.csv 2>/dev/null; ls $out
