#!/bin/bash
set -o pipefail
out=gpurun_out/r4_jpeg; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_jpeg_split.py tests/test_gpu_ops.py -x -q -m gpu -k "jpeg or split or decode or preprocess or loader" > $out/tests3.log 2>&1 || { tail -30 $out/tests3.log; exit 1; }
tail -2 $out/tests3.log
FILES=photo THREADS= NPROCS= SPLIT_THREADS=8,16 timeout -k 10 300 python tools/loader_bench.py > $out/loader_photo3.log 2>&1 || { tail -30 $out/loader_photo3.log; exit 1; }
cat $out/loader_photo3.log
THREADS= NPROCS= SPLIT_THREADS=16 timeout -k 10 300 python tools/loader_bench.py > $out/loader_smooth3.log 2>&1 || { tail -30 $out/loader_smooth3.log; exit 1; }
cat $out/loader_smooth3.log
MODES=split timeout -k 10 300 python tools/train_files_bench.py > $out/train_files3.log 2>&1 || { tail -30 $out/train_files3.log; exit 1; }
grep "^loader" $out/train_files3.log
