#!/bin/bash
# overlap experiment: the XE step with the encoder serial / overlapped at several LDS reservations of its conv workgroups
out=gpurun_out/ovl; mkdir -p $out
export COMIC_TUNE_CACHE=$out/tiles.json
B="python3 bench.py --gpus 1 --steps 40 --warmup 5 --no-cpu-baseline --no-extras"
run() { name=$1; shift; env "$@" timeout -k 10 300 $B > $out/$name.log 2>&1; echo "$name $(tail -1 $out/$name.log | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["in_timed_region"]["cnn_forward_ms"], d["roofline"]["cnn_forward_ms"])')"; }
run base A=1 &&
run base2 A=1 &&
run serial COMIC_OVERLAP=0 &&
run lds0 COMIC_POLITE_LDS_KB=0 &&
run lds120 COMIC_POLITE_LDS_KB=120 &&
run lds160 COMIC_POLITE_LDS_KB=160 &&
run grp10 COMIC_ENC_GROUP=10 &&
run grp40 COMIC_ENC_GROUP=40
