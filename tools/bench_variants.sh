#!/bin/bash
# The driver's bench line under COMIC_OVERLAP=0 / COMIC_POLITE_LDS_KB=0 against the default, twice each, on ONE box (value,
# ms per step, roofline.frac, decoder ms, the forward inside the timed region).   bash tools/bench_variants.sh
mkdir -p gpurun_out
for v in "X=1" "COMIC_OVERLAP=0" "COMIC_POLITE_LDS_KB=0" "X=1" "COMIC_OVERLAP=0" "COMIC_POLITE_LDS_KB=0"; do
  env $v python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/bv.json 2> gpurun_out/bv.err
  python -c "
import json
d=json.loads(open('gpurun_out/bv.json').read().strip().splitlines()[-1])
print('$v', d['value'], d['ms_per_step'], d['roofline']['frac'], d['decoder_roofline']['ms_per_step'], d['roofline'].get('in_timed_region'))
"
done
