#!/bin/bash
# Decoder training step alone (tools/dec_step_time.py) under rocprofv3 --kernel-trace: the step's launch timeline.
#   B=64 OUT=gpurun_out/dec bash tools/dec_prof.sh
out=${OUT:-gpurun_out/dec}; mkdir -p $out
export TMPDIR=/tmp
B=${B:-64}
M=25 C=2048 CG=2048 B=$B N=20 python3 tools/dec_step_time.py | tail -1
M=25 C=2048 CG=2048 B=$B N=20 GRAPH=1 python3 tools/dec_step_time.py | tail -1
rm -rf $out/kt_$B
M=25 C=2048 CG=2048 B=$B N=8 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/kt_$B --output-format csv -- python3 tools/dec_step_time.py > $out/prof_$B.log 2>&1 || { tail -5 $out/prof_$B.log; exit 1; }
t=$(ls $out/kt_$B/*/*kernel_trace.csv | head -1)
python3 tools/step_timeline.py $t > $out/decoder_step_timeline_B$B.txt
cat $out/decoder_step_timeline_B$B.txt
