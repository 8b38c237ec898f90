#!/bin/bash
# round-3 counter passes over the InceptionV3 forward alone: HBM traffic at the forward sizes the bench may use, MFMA / LDS
# counters at 1280 and 64 images
out=gpurun_out/r3_prof; mkdir -p $out
for B in 1920 1280 640; do
  bash tools/pmc_cnn.sh $out $B || { echo "traffic pass at $B failed"; exit 1; }
  echo "traffic $B done"
done
for B in 1280 64; do
  bash tools/pmc_mfma.sh $out $B || { echo "mfma pass at $B failed"; exit 1; }
  echo "mfma $B done"
done
ls $out | head -50
