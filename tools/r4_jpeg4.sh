#!/bin/bash
set -o pipefail
out=gpurun_out/r4_jpeg; mkdir -p $out
MODES=split,processes timeout -k 10 400 python tools/train_files_bench.py > $out/train_files4.log 2>&1 || { tail -30 $out/train_files4.log; exit 1; }
grep "^loader" $out/train_files4.log
timeout -k 10 900 python -m pytest tests/test_gpu_cli.py -x -q -m gpu > $out/tests4.log 2>&1 || { tail -30 $out/tests4.log; exit 1; }
tail -2 $out/tests4.log
cd /tmp && export TMPDIR=/tmp
export FILES=photo THREADS= NPROCS= SPLIT_THREADS=16
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d /tmp/kt -o loader --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/loader_bench.py > $GRAFT_REPO_ROOT/$out/prof.log 2>&1 || { tail -20 $GRAFT_REPO_ROOT/$out/prof.log; exit 1; }
cd $GRAFT_REPO_ROOT
f=$(find /tmp/kt -name '*kernel_stats.csv' | head -1); cp "$f" $out/loader_kernel_stats.csv; cat $out/loader_kernel_stats.csv
