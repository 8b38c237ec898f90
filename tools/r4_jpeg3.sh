#!/bin/bash
set -o pipefail
out=gpurun_out/r4_jpeg; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_jpeg_split.py tests/test_gpu_ops.py -x -q -m gpu -k "jpeg or split or decode or preprocess or loader" > $out/tests5.log 2>&1 || { tail -30 $out/tests3.log; exit 1; }
tail -2 $out/tests5.log
FILES=photo THREADS= NPROCS= SPLIT_THREADS=8,16 timeout -k 10 300 python tools/loader_bench.py > $out/loader_photo5.log 2>&1 || { tail -30 $out/loader_photo5.log; exit 1; }
cat $out/loader_photo5.log
THREADS= NPROCS= SPLIT_THREADS=16 timeout -k 10 300 python tools/loader_bench.py > $out/loader_smooth5.log 2>&1 || { tail -30 $out/loader_smooth5.log; exit 1; }
cat $out/loader_smooth5.log
MODES=split timeout -k 10 300 python tools/train_files_bench.py > $out/train_files5.log 2>&1 || { tail -30 $out/train_files5.log; exit 1; }
grep "^loader" $out/train_files5.log
cd /tmp && export TMPDIR=/tmp
export FILES=photo THREADS= NPROCS= SPLIT_THREADS=16
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d /tmp/kt -o loader --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/loader_bench.py > $GRAFT_REPO_ROOT/$out/prof5.log 2>&1 || { tail -20 $GRAFT_REPO_ROOT/$out/prof5.log; exit 1; }
cd $GRAFT_REPO_ROOT
f=$(find /tmp/kt -name '*kernel_stats.csv' | head -1); cp "$f" $out/loader_kernel_stats5.csv; cut -c1-200 $out/loader_kernel_stats5.csv
