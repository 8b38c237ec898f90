#!/bin/bash
# phase clocks of the persistent decoder loops (COMIC_PERSIST_STAMPS=1) at the three attention-memory sizes: M = 25 (COMIC-256 on
# InceptionV3 @224), M = 64 (@299), M = 196 (Inception-V1 Mixed_4f, the reference CLI's default)
out=gpurun_out/phases; mkdir -p $out
: > $out/phases.txt
for geo in "25 2048 2048" "64 2048 2048" "196 832 1024"; do
  set -- $geo
  echo "== M = $1 (C = $2, Cg = $3), batch 64, T' = 29, COMIC-256 decoder" >> $out/phases.txt
  M=$1 C=$2 CG=$3 COMIC_PERSIST_STAMPS=1 N=4 timeout -k 10 300 python3 tools/dec_step_prof.py 2>&1 | grep "persist stamps\|decoder step" | tail -3 >> $out/phases.txt
done
cat $out/phases.txt
