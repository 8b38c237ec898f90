#!/bin/bash
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/r3_beamradix; mkdir -p $out
cd /tmp
SPEC=radix B=32 W=7 timeout -k 10 100 python3 $GRAFT_REPO_ROOT/tools/beam_time.py | tail -1
SPEC=radix B=32 W=7 GRAPH=0 timeout -k 10 200 rocprofv3 --kernel-trace --stats -d /tmp/profr -o beam --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/beam_time.py > $out/run.log 2>&1
cp /tmp/profr/*kernel_stats.csv $out/stats.csv
