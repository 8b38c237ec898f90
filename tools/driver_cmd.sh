#!/bin/bash
# the driver's round-end bench command (BENCH_rNN.json: --gpus 1 --steps 20 --warmup 5), then its eager kernel trace
set -e
out=gpurun_out/driver_cmd; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 1100 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench.log 2> $out/bench.err || { tail -30 $out/bench.err; exit 1; }
tail -1 $out/bench.log | cut -c1-300
export COMIC_TUNE_CACHE=$out/tiles.json
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $out/tune.log 2>&1
COMIC_GRAPH_CNN=0 COMIC_GRAPH_DEC=0 COMIC_OVERLAP=0 timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $out/kt --output-format csv -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $out/bench_eager.log 2>&1
tail -1 $out/bench_eager.log | cut -c1-200
