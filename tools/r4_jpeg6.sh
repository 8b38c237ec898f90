#!/bin/bash
set -o pipefail
out=gpurun_out/r4_jpeg; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export FILES=photo THREADS= NPROCS= SPLIT_THREADS=16
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d /tmp/kt -o loader --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/loader_bench.py > $GRAFT_REPO_ROOT/$out/prof6.log 2>&1 || { tail -20 $GRAFT_REPO_ROOT/$out/prof6.log; exit 1; }
cd $GRAFT_REPO_ROOT
f=$(find /tmp/kt -name '*kernel_stats.csv' | head -1); cp "$f" $out/loader_kernel_stats6.csv; cut -c1-170 $out/loader_kernel_stats6.csv
grep -E "split|files" $out/prof6.log
