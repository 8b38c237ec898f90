#!/bin/bash
# profile set, part B: eager kernel stats of the bench command, decoder step timeline, beam-3 / SCST / cnn_finetune stats.
out=${OUT:-gpurun_out/prof}; mkdir -p $out
export TMPDIR=/tmp
export COMIC_TUNE_CACHE=$out/tiles_bench.json
prof() {   # name, then the program
  name=$1; shift
  rm -rf $out/kt_$name
  timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $out/kt_$name --output-format csv -- "$@" > $out/$name.log 2>&1 || { echo "$name failed"; tail -5 $out/$name.log; return 1; }
  f=$(ls $out/kt_$name/*/*kernel_stats.csv | head -1); cp $f $out/${name}_kernel_stats.csv; echo "$name: $(wc -l < $f) kernels"
}
echo "== bench (tune)"; timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $out/bench_tune.log 2>&1 || exit 1
echo "== bench eager trace"; COMIC_GRAPH_CNN=0 COMIC_GRAPH_DEC=0 COMIC_OVERLAP=0 prof bench_steps20_eager python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras || exit 1
t=$(ls $out/kt_bench_steps20_eager/*/*kernel_trace.csv | head -1); python3 tools/step_timeline.py $t > $out/decoder_step_timeline.txt; tail -1 $out/decoder_step_timeline.txt
echo "== beam-3"; GRAPH=0 prof beam3 python3 tools/beam_time.py
echo "== cnn_finetune"; prof finetune python3 tools/ft_step_prof.py && { t=$(ls $out/kt_finetune/*/*kernel_trace.csv | head -1); python3 tools/ft_lanes.py $t > $out/finetune_lanes.txt; head -3 $out/finetune_lanes.txt; }
echo "== cnn_finetune, bf16x3 plan"; X3=1 N=4 prof finetune_x3 python3 tools/ft_step_time.py
echo "== scst"; GRAPH=0 prof scst python3 tools/scst_step_timeline.py
python3 tools/scst_step_timeline.py 2>&1 | tail -1 > $out/scst_step_timeline.txt; cat $out/scst_step_timeline.txt
