#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r4_bench; mkdir -p $out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export COMIC_TUNE_CACHE=$out/tiles.json
timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench.log 2> $out/bench.err || { tail -20 $out/bench.err; exit 1; }
python3 - <<'P'
import json,os
d=json.loads(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r4_bench/bench.log').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['cnn_forward_ms'], d['decoder_roofline']['ms_per_step'])
ex=d['extras']
print(json.dumps(ex.get('xe_x3'))[:1500])
print(json.dumps(ex.get('xe_f32'))[:900])
for k in ('beam3_captions_per_sec','scst_images_per_sec','cnn_finetune_images_per_sec'): print(k, ex.get(k))
P
