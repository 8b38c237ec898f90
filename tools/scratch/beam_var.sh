#!/bin/bash
# kernel trace of the beam decode under library variants (COMIC_HIP_LIB)
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/r3_beamvar; mkdir -p $out
cd /tmp
for v in "$@"; do
  lib=$GRAFT_REPO_ROOT/comic-compact-image-captioning-with-attention_amd/lib/libcomic_hip$v.so
  COMIC_HIP_LIB=$lib GRAPH=0 timeout -k 10 200 rocprofv3 --kernel-trace --stats -d /tmp/prof$v -o beam --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/beam_time.py > $out/run$v.log 2>&1 || exit 1
  echo "== variant '$v'"; grep "beam_logits_kernel\|beam_merge2\|lstm_step\|beam_pack_y" /tmp/prof$v/*kernel_stats.csv | cut -d'"' -f2,3 | cut -c1-60,100-200 | awk -F, '{print $1, $(NF-6), $(NF-4)}' 
  cp /tmp/prof$v/*kernel_stats.csv $out/stats$v.csv
done
