#!/bin/bash
# experiment: the post-loop gradient products on two lanes (COMIC_GRAD_LANES) vs one stream
set -e
out=gpurun_out/lanes; mkdir -p $out
export TMPDIR=/tmp
export COMIC_TUNE_CACHE=$out/tiles.json
timeout -k 10 600 python -m pytest tests/test_gpu_path.py -x -q -m gpu -k "test_decoder_train_step_matches_oracle or test_persistent or test_train_step_full_batch or test_cnn_finetune" > $out/t.log 2>&1 || { tail -40 $out/t.log; exit 1; }
tail -1 $out/t.log
for m in 1 0 1 0; do
  COMIC_GRAD_LANES=$m timeout -k 10 400 python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras > $out/bench_$m.log 2>&1
  echo "lanes $m: $(tail -1 $out/bench_$m.log | cut -c100-200)"
done
