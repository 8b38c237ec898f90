#!/bin/bash
# round 4: decoder-step timing of two library builds + timeline
out=$GRAFT_REPO_ROOT/gpurun_out/r4_dec2; mkdir -p $out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "gemm" > $out/t_gemm.log 2>&1 || { tail -40 $out/t_gemm.log; exit 1; }
tail -1 $out/t_gemm.log
export M=25 C=2048 CG=2048 B=64 N=30

timeout -k 10 300 python tools/dec_step_time.py 2>&1 | tail -1
PKG=$GRAFT_REPO_ROOT/comic-compact-image-captioning-with-attention_amd


cd /tmp
for v in A; do
  if [ $v = B ]; then export COMIC_HIP_LIB=$PKG/lib/libcomic_hip_B.so; fi
  rm -rf /tmp/kt
  N=6 timeout -k 10 500 rocprofv3 --kernel-trace --stats -d /tmp/kt -o b --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/dec_step_time.py > $out/prof_$v.log 2>&1 || { tail -20 $out/prof_$v.log; exit 1; }
  python3 $GRAFT_REPO_ROOT/tools/step_timeline.py /tmp/kt/b_kernel_trace.csv > $out/step_timeline_$v.txt
  grep "gemm_group\|^step" $out/step_timeline_$v.txt
done
