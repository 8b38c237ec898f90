#!/bin/bash
# per-round profile set: (1) tune once, cache the tiles; (2) kernel-trace stats of the eager bench with the cached tiles
# (graph replays are invisible to the kernel trace); (3) FETCH_SIZE / WRITE_SIZE passes over the CNN forward alone.
out=$1; mkdir -p $out
export TMPDIR=/tmp
export COMIC_TUNE_CACHE=$out/tiles.json
export B=${B:-1920}    # images per encoder forward = 64 x bench.py DEFAULT_ENC_GROUP
python3 bench.py --steps 30 --warmup 2 --no-cpu-baseline --no-extras > $out/bench_tune.log 2>&1 || exit 1
python3 tools/run_cnn.py > $out/run_cnn_tune.log 2>&1 || exit 1
COMIC_GRAPH_CNN=0 COMIC_GRAPH_DEC=0 COMIC_OVERLAP=0 rocprofv3 --kernel-trace --stats -d $out/kt --output-format csv -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $out/bench_eager.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/pmc_FETCH_SIZE --output-format csv -- python3 tools/run_cnn.py > $out/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/pmc_WRITE_SIZE --output-format csv -- python3 tools/run_cnn.py > $out/pmc_write.log 2>&1 || exit 1
