#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/r4_ovl; mkdir -p $out
export COMIC_TUNE_CACHE=$out/tiles.json
run() { echo -n "$1: "; env $1 timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>$out/err.log | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], 'cnn iso', d['roofline']['cnn_forward_ms'], 'in-region', d['roofline'].get('in_timed_region',{}).get('cnn_forward_ms'), 'dec', d['decoder_roofline']['ms_per_step'])"; }
run COMIC_X=0
run COMIC_OVERLAP=0
run COMIC_POLITE_LDS_KB=0
run COMIC_POLITE_LDS_KB=64
run COMIC_ENC_GROUP=10
run COMIC_ENC_GROUP=5
