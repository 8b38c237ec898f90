#!/bin/bash
# profile set, part A (CNN forward): per-launch MFMA / LDS / L2 counters at 1280 and 64 images, HBM traffic at 1280.
# Every step writes under $OUT, default gpurun_out/prof (progress for the runner's silence check); the summaries are copied to profiles/.
out=${OUT:-gpurun_out/prof}; mkdir -p $out
export TMPDIR=/tmp
echo "== counters, 1280 images"; timeout -k 10 420 bash tools/pmc_mfma.sh $out 1280 || exit 1
echo "== counters, 64 images";   timeout -k 10 300 bash tools/pmc_mfma.sh $out 64 || exit 1
echo "== traffic, 1280 images";  timeout -k 10 300 bash tools/pmc_cnn.sh $out 1280 || exit 1
ls $out | head -40
