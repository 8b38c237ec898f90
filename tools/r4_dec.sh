#!/bin/bash
# round 4: grouped GEMM tests, decoder-step A/B (COMIC_GROUP_GEMM=0 vs default) and the step's kernel timeline
out=$GRAFT_REPO_ROOT/gpurun_out/r4_dec; mkdir -p $out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "gemm" > $out/t_gemm.log 2>&1 || { tail -40 $out/t_gemm.log; exit 1; }
tail -2 $out/t_gemm.log
timeout -k 10 1200 python -m pytest tests/test_gpu_path.py -q -m gpu -k "decoder_train_step or persistent or split_train or pipelined or finetune_step_end or large_memory or run_ahead" > $out/t_path.log 2>&1 || tail -40 $out/t_path.log | grep -E "^FAILED|passed|failed"
tail -2 $out/t_path.log
timeout -k 10 200 python tools/gg_diag.py > $out/diag.log 2>&1; tail -45 $out/diag.log
export M=25 C=2048 CG=2048 B=64 N=30
COMIC_GROUP_GEMM=0 timeout -k 10 300 python tools/dec_step_time.py > $out/time_old.log 2>&1 || { tail -20 $out/time_old.log; exit 1; }
tail -1 $out/time_old.log
timeout -k 10 300 python tools/dec_step_time.py > $out/time_new.log 2>&1 || { tail -20 $out/time_new.log; exit 1; }
tail -1 $out/time_new.log
cd /tmp
N=6 timeout -k 10 500 rocprofv3 --kernel-trace --stats -d /tmp/kt -o b --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/dec_step_time.py > $out/prof.log 2>&1 || { tail -20 $out/prof.log; exit 1; }
python3 $GRAFT_REPO_ROOT/tools/step_timeline.py /tmp/kt/b_kernel_trace.csv > $out/step_timeline.txt
cp /tmp/kt/b_kernel_stats.csv $out/
tail -3 $out/step_timeline.txt
