#!/bin/bash
# round-2 profile set: tools/prof_bench.sh (tune, eager kernel trace, PMC passes over the CNN forward) + phase clocks of the persistent loops
set -e
out=gpurun_out/prof_round
bash tools/prof_bench.sh $out
B=1920 python3 tools/pmc_traffic.py $out $out/cnn_hbm_traffic.json > $out/pmc_traffic.log 2>&1
export TMPDIR=/tmp
export COMIC_TUNE_CACHE=$out/tiles.json
COMIC_PERSIST_STAMPS=1 COMIC_GRAPH_DEC=0 COMIC_OVERLAP=0 timeout -k 10 400 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $out/bench_stamps.log 2>&1
grep "persist stamps" $out/bench_stamps.log | tail -4 > $out/persist_phases.txt
cat $out/persist_phases.txt
tail -1 $out/bench_eager.log | cut -c1-200
