#!/bin/bash
# phase clock of the persistent decoder loop (diagnostic; every launch synchronises)
set -e
out=gpurun_out/r2_persist; mkdir -p $out
export TMPDIR=/tmp
export COMIC_TUNE_CACHE=$out/tiles.json
COMIC_PERSIST_STAMPS=1 COMIC_OVERLAP=0 timeout -k 10 400 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $out/bench_stamps.log 2>&1 || { tail -20 $out/bench_stamps.log; exit 1; }
grep "persist stamps" $out/bench_stamps.log | tail -4
