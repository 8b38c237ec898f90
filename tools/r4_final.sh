#!/bin/bash
# round-4 final validation + profile set -> gpurun_out/r4_final
out=$GRAFT_REPO_ROOT/gpurun_out/r4_final; mkdir -p $out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -q -m gpu > $out/tests.log 2>&1; tail -3 $out/tests.log | cut -c1-300
grep -E "^FAILED|^ERROR" $out/tests.log | head -20
timeout -k 10 120 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
export COMIC_TUNE_CACHE=$out/tiles.json
timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench.log 2> $out/bench.err || { tail -20 $out/bench.err; exit 1; }
tail -1 $out/bench.log | cut -c1-300
# cnn_finetune step: tune un-profiled, then kernel stats
timeout -k 10 300 python3 tools/ft_step_time.py 2>&1 | tail -1
cd /tmp; rm -rf /tmp/kt
N=6 timeout -k 10 400 rocprofv3 --kernel-trace --stats -d /tmp/kt -o b --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/ft_step_time.py > $out/ft_prof.log 2>&1 || { tail -20 $out/ft_prof.log; exit 1; }
cp /tmp/kt/b_kernel_stats.csv $out/finetune_kernel_stats.csv
# SCST step as the bench runs it (rollouts of 29 steps, hypotheses cut at caption lengths)
rm -rf /tmp/kt
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d /tmp/kt -o b --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/scst_step_timeline.py > $out/scst_prof.log 2>&1 || { tail -20 $out/scst_prof.log; exit 1; }
grep "step " $out/scst_prof.log | tail -1 | cut -c1-300; cp /tmp/kt/b_kernel_stats.csv $out/scst_kernel_stats.csv
# bench, eager kernel stats + decoder step timeline
rm -rf /tmp/kt
COMIC_GRAPH_CNN=0 COMIC_GRAPH_DEC=0 COMIC_OVERLAP=0 timeout -k 10 500 rocprofv3 --kernel-trace --stats -d /tmp/kt -o b --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $out/bench_eager.log 2>&1 || { tail -20 $out/bench_eager.log; exit 1; }
tail -1 $out/bench_eager.log | cut -c1-200; cp /tmp/kt/b_kernel_stats.csv $out/bench_steps20_eager_kernel_stats.csv
python3 $GRAFT_REPO_ROOT/tools/step_timeline.py /tmp/kt/b_kernel_trace.csv > $out/decoder_step_timeline.txt; tail -1 $out/decoder_step_timeline.txt
cd $GRAFT_REPO_ROOT
bash tools/pmc_mfma.sh $out 1280 || echo "mfma 1280 failed"
bash tools/pmc_mfma.sh $out 64 || echo "mfma 64 failed"
for nb in 1920 1280 640; do
  rm -rf $out/tr_$nb; mkdir -p $out/tr_$nb
  bash tools/pmc_cnn.sh $out/tr_$nb $nb || echo "traffic $nb failed"
done
ls $out | head -40
