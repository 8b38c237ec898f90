#!/bin/bash
# full GPU test suite, then the eager kernel trace of the bench (per-kernel times), then the default bench line
set -e
out=gpurun_out/r2_full; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $out/tests.log 2>&1 || { tail -40 $out/tests.log; exit 1; }
tail -3 $out/tests.log
export COMIC_TUNE_CACHE=$out/tiles.json
timeout -k 10 400 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $out/bench.log 2>&1 || { tail -20 $out/bench.log; exit 1; }
tail -1 $out/bench.log | cut -c1-250
COMIC_GRAPH_CNN=0 COMIC_GRAPH_DEC=0 COMIC_OVERLAP=0 timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $out/kt --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $out/bench_eager.log 2>&1 || { tail -20 $out/bench_eager.log; exit 1; }
tail -1 $out/bench_eager.log | cut -c1-250
