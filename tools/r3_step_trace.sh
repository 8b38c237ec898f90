#!/bin/bash
# eager kernel trace of the bench's decoder step (no graphs, no encoder overlap), timeline of one step
out=$GRAFT_REPO_ROOT/gpurun_out/r3_steptrace; mkdir -p $out
export TMPDIR=/tmp
cd /tmp
COMIC_GRAPH_CNN=0 COMIC_GRAPH_DEC=0 COMIC_OVERLAP=0 timeout -k 10 500 rocprofv3 --kernel-trace --stats -d /tmp/kt -o b --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $out/bench_eager.log 2>&1 || { tail -20 $out/bench_eager.log; exit 1; }
tail -1 $out/bench_eager.log | cut -c1-200
python3 $GRAFT_REPO_ROOT/tools/step_timeline.py /tmp/kt/b_kernel_trace.csv > $out/step_timeline.txt
cp /tmp/kt/b_kernel_stats.csv $out/
tail -3 $out/step_timeline.txt
