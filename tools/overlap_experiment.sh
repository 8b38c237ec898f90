#!/bin/bash
# overlap experiment: LDS the overlapped encoder kernels reserve (1 workgroup per CU above 80 KB), encoder group
set -e
out=gpurun_out/r2_prio; mkdir -p $out
export TMPDIR=/tmp
export COMIC_TUNE_CACHE=$out/tiles.json
for kb in 84 0 48 120; do
  COMIC_POLITE_LDS_KB=$kb timeout -k 10 400 python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras > $out/bench_lds$kb.log 2>&1 || { tail -20 $out/bench_lds$kb.log; exit 1; }
  echo "polite lds $kb: $(tail -1 $out/bench_lds$kb.log | cut -c100-230)"
done
