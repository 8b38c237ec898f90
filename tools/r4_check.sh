#!/bin/bash
# full GPU suite + smoke + the driver's bench command
set -o pipefail
out=gpurun_out/r4_check; mkdir -p $out
timeout -k 10 1000 python -m pytest tests -q -m gpu -x > $out/tests.log 2>&1; rc=$?; tail -3 $out/tests.log | cut -c1-300
grep -E "^FAILED|^ERROR" $out/tests.log | head -20
[ $rc -eq 0 ] || exit 1
timeout -k 10 120 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
export COMIC_TUNE_CACHE=$out/tiles.json
timeout -k 10 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench.log 2> $out/bench.err || { tail -20 $out/bench.err; exit 1; }
tail -1 $out/bench.log | cut -c1-400
python3 - <<'P'
import json
l=[x for x in open('gpurun_out/r4_check/bench.log') if x.startswith('{')][-1]
j=json.loads(l); print(json.dumps(j['extras'].get('input_pipeline'), indent=1)); print({k: j['extras'][k] for k in ('beam3_captions_per_sec','scst_images_per_sec','cnn_finetune_images_per_sec')})
P
