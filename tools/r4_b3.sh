#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r4_b3; mkdir -p $out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export COMIC_TUNE_CACHE=$out/tiles.json
for b in 32 64; do B=$b GRAPH=1 timeout -k 10 300 python3 tools/ovl_premise.py > $out/premise_b$b.txt 2>$out/premise_b$b.err || { tail -5 $out/premise_b$b.err; }; cat $out/premise_b$b.txt; done
