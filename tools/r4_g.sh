#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/r4_g; mkdir -p $out
export COMIC_TUNE_CACHE=$out/tiles.json
run() { echo -n "$*: "; env "$@" timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>$out/err.log | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], 'cnn iso', d['roofline']['cnn_forward_ms'], 'dec', d['decoder_roofline']['ms_per_step'])"; }
run COMIC_X=0
run COMIC_GRAPH_DEC=0
run COMIC_X=0
run COMIC_GRAPH_DEC=0
