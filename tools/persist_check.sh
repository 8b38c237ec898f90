#!/bin/bash
# persistent decoder loop: one small parity case under a short timeout, then the rest, then the phase clock
set -e
out=gpurun_out/persist; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 240 python -m pytest tests/test_gpu_path.py -x -q -m gpu -k "test_decoder_train_step_matches_oracle and kw1 and False-False" > $out/t1.log 2>&1 || { tail -30 $out/t1.log; exit 1; }
tail -3 $out/t1.log
timeout -k 10 600 python -m pytest tests/test_gpu_path.py -x -q -m gpu -k "test_decoder_train_step_matches_oracle or test_persistent or test_train_step_full_batch" > $out/t2.log 2>&1 || { tail -40 $out/t2.log; exit 1; }
tail -3 $out/t2.log
export COMIC_TUNE_CACHE=$out/tiles.json
COMIC_PERSIST_STAMPS=1 COMIC_GRAPH_DEC=0 COMIC_OVERLAP=0 timeout -k 10 400 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $out/bench_stamps.log 2>&1 || { tail -20 $out/bench_stamps.log; exit 1; }
grep "persist stamps" $out/bench_stamps.log | tail -4
timeout -k 10 400 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $out/bench_p1.log 2>&1 || { tail -20 $out/bench_p1.log; exit 1; }
tail -1 $out/bench_p1.log | cut -c1-250
