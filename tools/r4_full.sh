#!/bin/bash
# full GPU suite + default bench line
out=$GRAFT_REPO_ROOT/gpurun_out/r4_full; mkdir -p $out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests -q -m gpu > $out/tests.log 2>&1; tail -5 $out/tests.log | cut -c1-300
grep -E "^FAILED|^ERROR" $out/tests.log | head -20
export COMIC_TUNE_CACHE=$out/tiles.json
timeout -k 10 500 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $out/bench.log 2> $out/bench.err || { tail -20 $out/bench.err; exit 1; }
tail -1 $out/bench.log | cut -c1-400
