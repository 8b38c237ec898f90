#!/bin/bash
# persistent decoder loop: bench with and without it, then an eager kernel trace with it
set -e
out=gpurun_out/r2_persist; mkdir -p $out
export TMPDIR=/tmp
export COMIC_TUNE_CACHE=$out/tiles.json
timeout -k 10 400 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $out/bench_p1.log 2>&1 || { tail -20 $out/bench_p1.log; exit 1; }
tail -1 $out/bench_p1.log | cut -c1-400
COMIC_PERSIST=0 timeout -k 10 400 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $out/bench_p0.log 2>&1 || { tail -20 $out/bench_p0.log; exit 1; }
tail -1 $out/bench_p0.log | cut -c1-400
COMIC_OVERLAP=0 timeout -k 10 400 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $out/bench_p1_noov.log 2>&1 || { tail -20 $out/bench_p1_noov.log; exit 1; }
tail -1 $out/bench_p1_noov.log | cut -c1-400
COMIC_PERSIST=0 COMIC_OVERLAP=0 timeout -k 10 400 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $out/bench_p0_noov.log 2>&1 || { tail -20 $out/bench_p0_noov.log; exit 1; }
tail -1 $out/bench_p0_noov.log | cut -c1-400
COMIC_GRAPH_CNN=0 COMIC_OVERLAP=0 timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $out/kt --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $out/bench_eager.log 2>&1 || { tail -20 $out/bench_eager.log; exit 1; }
