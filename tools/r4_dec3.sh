#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r4_dec3; mkdir -p $out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests/test_gpu_path.py -x -q -m gpu -k "decoder_train_step or persistent or split_train or pipelined or finetune_step_end or large_memory or run_ahead or full_batch or scst_step" > $out/t_path.log 2>&1 || { tail -30 $out/t_path.log; exit 1; }
tail -1 $out/t_path.log
export M=25 C=2048 CG=2048 B=64 N=30
timeout -k 10 300 python tools/dec_step_time.py 2>&1 | tail -1
cd /tmp; rm -rf /tmp/kt
N=6 timeout -k 10 500 rocprofv3 --kernel-trace --stats -d /tmp/kt -o b --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/dec_step_time.py > $out/prof.log 2>&1 || { tail -20 $out/prof.log; exit 1; }
python3 $GRAFT_REPO_ROOT/tools/step_timeline.py /tmp/kt/b_kernel_trace.csv > $out/step_timeline.txt
cat $out/step_timeline.txt | cut -c1-110
