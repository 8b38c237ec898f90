#!/bin/bash
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/r3_scsttrain; mkdir -p $out
cd /tmp
timeout -k 10 200 python3 $GRAFT_REPO_ROOT/tools/scst_train_time.py | tail -2
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/profs -o st --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/scst_train_time.py > $out/run.log 2>&1
cp /tmp/profs/*kernel_stats.csv $out/stats.csv
