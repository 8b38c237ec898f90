#!/bin/bash
# round-4 baseline: full GPU suite, driver bench command, CU-mask census + overlap premise
out=$GRAFT_REPO_ROOT/gpurun_out/r4_base; mkdir -p $out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -q -m gpu > $out/tests.log 2>&1; tail -3 $out/tests.log | cut -c1-300
grep -E "^FAILED|^ERROR" $out/tests.log | head -20
export COMIC_TUNE_CACHE=$out/tiles.json
timeout -k 10 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench.log 2> $out/bench.err || { tail -20 $out/bench.err; exit 1; }
tail -1 $out/bench.log | cut -c1-600
hipcc --offload-arch=gfx950 -O2 tools/micro/cu_mask_census.hip -o /tmp/cu_mask_census && timeout -k 10 60 /tmp/cu_mask_census > $out/census.txt 2>&1; cat $out/census.txt
for g in 1 0; do GRAPH=$g timeout -k 10 300 python3 tools/ovl_premise.py > $out/premise_g$g.txt 2>$out/premise_g$g.err || { tail -5 $out/premise_g$g.err; }; cat $out/premise_g$g.txt; done
