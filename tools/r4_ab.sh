#!/bin/bash
# A/B of two builds of the library on one box: whole InceptionV3 forward (autotuned per build) + single layers
out=$GRAFT_REPO_ROOT/gpurun_out/r4_ab; mkdir -p $out
cd $GRAFT_REPO_ROOT
LIBA=$PWD/comic-compact-image-captioning-with-attention_amd/lib/libcomic_hip_A.so
for r in 1 2; do
  echo "=== A"; COMIC_HIP_LIB=$LIBA COMIC_TUNE_CACHE=$out/tiles_A.json B=${NB:-1280} GRAPH=1 timeout -k 10 300 python3 tools/run_cnn.py 2>&1 | grep "cnn forward"
  echo "=== B"; COMIC_TUNE_CACHE=$out/tiles_B.json B=${NB:-1280} GRAPH=1 timeout -k 10 300 python3 tools/run_cnn.py 2>&1 | grep "cnn forward"
done
export CASES='1280,12,12,768,192,1,1,1,SAME;1280,12,12,192,192,7,1,1,SAME;1280,25,25,288,384,3,3,2,VALID;1280,5,5,2048,384,1,1,1,SAME' TILES=44,38,28,1 REPS=10
echo "=== A"; COMIC_HIP_LIB=$LIBA timeout -k 10 300 python3 tools/conv_variants.py 2>&1 | grep -v amdgpu
echo "=== B"; timeout -k 10 300 python3 tools/conv_variants.py 2>&1 | grep -v amdgpu
