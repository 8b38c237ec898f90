#!/bin/bash
set -o pipefail
out=gpurun_out/r4_last; mkdir -p $out
timeout -k 10 1000 python -m pytest tests -q -m gpu > $out/tests.log 2>&1; rc=$?; tail -2 $out/tests.log | cut -c1-200
grep -E "^FAILED|^ERROR" $out/tests.log | head
[ $rc -eq 0 ] || exit 1
export COMIC_TUNE_CACHE=$out/tiles.json
timeout -k 10 900 python3 bench.py > $out/bench_default.log 2> $out/bench_default.err || { tail -20 $out/bench_default.err; exit 1; }
tail -1 $out/bench_default.log | cut -c1-330
