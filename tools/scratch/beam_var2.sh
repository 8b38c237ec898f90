#!/bin/bash
# correctness screen of library variants on the small-row configurations
for v in "$@"; do
  export COMIC_HIP_LIB=$GRAFT_REPO_ROOT/comic-compact-image-captioning-with-attention_amd/lib/libcomic_hip$v.so
  for cfg in 20,3,25599,512 7,5,25599,512 20,3,25599,512 50,3,25599,512; do
    n=$(CFG=$cfg timeout -k 10 200 python tools/scratch/beam_dbg2.py 2>&1 | grep -c "ok True")
    echo "variant '$v' cfg $cfg: $n of 18 ok"
  done
  timeout -k 10 100 python tools/beam_time.py | tail -1
done
