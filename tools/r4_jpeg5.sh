#!/bin/bash
set -o pipefail
out=gpurun_out/r4_jpeg; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_jpeg_split.py tests/test_gpu_ops.py -x -q -m gpu -k "jpeg or split or decode or preprocess or loader" > $out/tests6.log 2>&1 || { tail -30 $out/tests6.log; exit 1; }
tail -2 $out/tests6.log
FILES=photo THREADS= NPROCS= SPLIT_THREADS=8,16 timeout -k 10 300 python tools/loader_bench.py 2>&1 | grep -E "split|files"
PROFILE=1 MODES=split,cache STEPS=400 WARM=100 timeout -k 10 500 python tools/train_files_bench.py 2>&1 | grep -E "^loader|host:"
