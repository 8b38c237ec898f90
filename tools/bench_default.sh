#!/bin/bash
# the driver's default bench invocation (extras + CPU baseline), as at round end
set -e
out=gpurun_out/r2_default; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 1100 python3 bench.py > $out/bench.log 2> $out/bench.err || { tail -30 $out/bench.err; exit 1; }
tail -1 $out/bench.log
