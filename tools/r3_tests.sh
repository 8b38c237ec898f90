#!/bin/bash
# round 3: full GPU suite, then the eager kernel trace of the beam-3 decode (tools/beam_time.py)
out=gpurun_out/r3_tests; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -q -m gpu > $out/tests.log 2>&1; rc=$?
tail -15 $out/tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 120 python3 tools/beam_time.py > $out/beam_time.log 2>&1; cat $out/beam_time.log | tail -1
cd /tmp && GRAPH=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_beam -o beam --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/beam_time.py > $GRAFT_REPO_ROOT/$out/beam_eager.log 2>&1
cp /tmp/prof_beam/*kernel_stats.csv $GRAFT_REPO_ROOT/$out/beam3_kernel_stats.csv
