#!/bin/bash
# round-4 profile set -> gpurun_out/r4_prof: finetune + SCST + bench kernel stats, counters of the forward, error table
out=$GRAFT_REPO_ROOT/gpurun_out/r4_prof; mkdir -p $out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export COMIC_TUNE_CACHE=$out/tiles.json
timeout -k 10 200 python3 tools/plan_error_table.py > $out/bf16_error_by_block.txt 2>$out/err_table.err || tail -3 $out/err_table.err
cat $out/bf16_error_by_block.txt
# cnn_finetune step
cd /tmp; rm -rf /tmp/kt
N=6 timeout -k 10 400 rocprofv3 --kernel-trace --stats -d /tmp/kt -o b --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/ft_step_time.py > $out/ft_prof.log 2>&1 || { tail -20 $out/ft_prof.log; exit 1; }
tail -1 $out/ft_prof.log; cp /tmp/kt/b_kernel_stats.csv $out/finetune_kernel_stats.csv; cp /tmp/kt/b_kernel_trace.csv $out/finetune_kernel_trace.csv
# SCST step (bench's extra geometry)
rm -rf /tmp/kt
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d /tmp/kt -o b --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/scst_time.py > $out/scst_prof.log 2>&1 || { tail -20 $out/scst_prof.log; exit 1; }
tail -3 $out/scst_prof.log; cp /tmp/kt/b_kernel_stats.csv $out/scst_kernel_stats.csv
# bench, eager
rm -rf /tmp/kt
COMIC_GRAPH_CNN=0 COMIC_GRAPH_DEC=0 COMIC_OVERLAP=0 timeout -k 10 500 rocprofv3 --kernel-trace --stats -d /tmp/kt -o b --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $out/bench_eager.log 2>&1 || { tail -20 $out/bench_eager.log; exit 1; }
tail -1 $out/bench_eager.log | cut -c1-200; cp /tmp/kt/b_kernel_stats.csv $out/bench_steps20_eager_kernel_stats.csv
cd $GRAFT_REPO_ROOT
bash tools/pmc_mfma.sh $out 1280 || echo "mfma 1280 failed"
