#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r4_b2; mkdir -p $out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_path.py -q -m gpu -k "backward or finetune" > $out/tests.log 2>&1; tail -3 $out/tests.log | cut -c1-300
export COMIC_TUNE_CACHE=$out/tiles.json
for b in 32 64; do B=$b GRAPH=1 timeout -k 10 300 python3 tools/ovl_premise.py > $out/premise_b$b.txt 2>$out/premise_b$b.err || { tail -5 $out/premise_b$b.err; }; cat $out/premise_b$b.txt; done
export M=25 C=2048 CG=2048 B=64 N=30
timeout -k 10 300 python tools/dec_step_time.py 2>&1 | tail -1
cd /tmp; rm -rf /tmp/kt
N=6 timeout -k 10 500 rocprofv3 --kernel-trace --stats -d /tmp/kt -o b --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/dec_step_time.py > $out/prof.log 2>&1 || { tail -20 $out/prof.log; exit 1; }
python3 $GRAFT_REPO_ROOT/tools/step_timeline.py /tmp/kt/b_kernel_trace.csv > $out/step_timeline.txt
cat $out/step_timeline.txt | cut -c1-130
