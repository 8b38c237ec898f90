#!/bin/bash
out=gpurun_out/final; mkdir -p $out
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1; tail -2 $out/smoke.log
COMIC_DIST_BACKEND=gloo timeout -k 10 500 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 6 --warmup 2 --no-cpu-baseline --no-extras > $out/gloo2.log 2>&1; grep '^{"metric"' $out/gloo2.log | cut -c1-260
