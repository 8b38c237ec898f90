#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r4_t1; mkdir -p $out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -q -m gpu > $out/tests.log 2>&1; tail -3 $out/tests.log | cut -c1-300
grep -E "^FAILED|^ERROR" $out/tests.log | head -20
