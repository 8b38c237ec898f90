#!/bin/bash
# training from files with the three loaders, loader stream on / off; manager-level parity test; loader kernel stats
set -o pipefail
out=gpurun_out/r4_jpeg; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_jpeg_split.py tests/test_gpu_ops.py -x -q -m gpu -k "jpeg or split or decode or preprocess or loader" > $out/tests2.log 2>&1 || { tail -30 $out/tests2.log; exit 1; }
tail -3 $out/tests2.log
timeout -k 10 400 python tools/train_files_bench.py > $out/train_files.log 2>&1 || { tail -30 $out/train_files.log; exit 1; }
grep "^loader" $out/train_files.log
COMIC_LOADER_STREAM=0 MODES=split,processes timeout -k 10 300 python tools/train_files_bench.py > $out/train_files_nostream.log 2>&1 || { tail -30 $out/train_files_nostream.log; exit 1; }
grep "^loader" $out/train_files_nostream.log
