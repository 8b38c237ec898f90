#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes over the CNN forward alone at B images per forward -> <out>/cnn_hbm_traffic_<B>.json
set -e
out=$1; B=${2:-1280}; mkdir -p $out
export TMPDIR=/tmp
export COMIC_TUNE_CACHE=$out/tiles_$B.json
export B
python3 tools/run_cnn.py > $out/run_cnn_tune_$B.log 2>&1
rm -rf $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/pmc_FETCH_SIZE --output-format csv -- python3 tools/run_cnn.py > $out/pmc_fetch_$B.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/pmc_WRITE_SIZE --output-format csv -- python3 tools/run_cnn.py > $out/pmc_write_$B.log 2>&1
python3 tools/pmc_traffic.py $out $out/cnn_hbm_traffic_$B.json
