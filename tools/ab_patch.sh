CASES='1280,54,54,80,192,3,3,1,VALID;1280,25,25,64,96,3,3,1,SAME;1280,25,25,48,64,5,5,1,SAME;1280,25,25,96,96,3,3,1,SAME;64,54,54,80,192,3,3,1,VALID;64,25,25,64,96,3,3,1,SAME'
export CASES TILES=13,14,15,18,19,20,22,23,48,49,50,52
for r in 1 2; do
echo "=== A (old pitch)"; COMIC_HIP_LIB=$PWD/comic-compact-image-captioning-with-attention_amd/lib/libcomic_hip_A.so python tools/conv_variants.py 2>&1 | grep -v amdgpu
echo "=== B (bank-aligned pitch)"; python tools/conv_variants.py 2>&1 | grep -v amdgpu
done
