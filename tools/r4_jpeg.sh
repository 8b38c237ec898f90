#!/bin/bash
# split JPEG decoder on the GPU box: parity tests, loader throughput on both file sets, kernel stats
set -o pipefail
out=gpurun_out/r4_jpeg; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_jpeg_split.py -x -q -m gpu > $out/tests.log 2>&1 || { tail -30 $out/tests.log; exit 1; }
tail -3 $out/tests.log
THREADS=16 NPROCS=16 SPLIT_THREADS=8,16 timeout -k 10 400 python tools/loader_bench.py > $out/loader_smooth.log 2>&1 || { tail -30 $out/loader_smooth.log; exit 1; }
cat $out/loader_smooth.log
FILES=photo THREADS=16 NPROCS=16 SPLIT_THREADS=8,16 timeout -k 10 400 python tools/loader_bench.py > $out/loader_photo.log 2>&1 || { tail -30 $out/loader_photo.log; exit 1; }
cat $out/loader_photo.log
cd /tmp && export TMPDIR=/tmp
export FILES=photo THREADS= NPROCS= SPLIT_THREADS=16
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$out/prof -o loader --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/loader_bench.py > $GRAFT_REPO_ROOT/$out/prof.log 2>&1 || { tail -20 $GRAFT_REPO_ROOT/$out/prof.log; exit 1; }
cd $GRAFT_REPO_ROOT
f=$(find $out/prof -name '*kernel_stats.csv' | head -1); cp $f $out/loader_kernel_stats.csv; cat $out/loader_kernel_stats.csv
tail -4 $out/prof.log
rm -rf $out/prof
