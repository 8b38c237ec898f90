#!/bin/bash
# MFMA / LDS / wait counters of the InceptionV3 forward alone, per launch (rocprofv3 --pmc passes with --kernel-trace only),
# plus the per-launch HIP-event table of tools/op_times.py.   pmc_mfma.sh <out> <B>
# -> <out>/op_times_<B>.log, <out>/pmc_<set>_<B>/..., reduced by tools/pmc_mfma.py into <out>/cnn_mfma_counters_<B>.json
out=$1; B=${2:-1280}; mkdir -p $out
export TMPDIR=/tmp
export COMIC_TUNE_CACHE=$out/tiles_$B.json
export B
[ -f $out/counters.txt ] || rocprofv3 -L > $out/counters.txt 2>&1
python3 tools/run_cnn.py > $out/run_cnn_tune_$B.log 2>&1 || { tail -5 $out/run_cnn_tune_$B.log; exit 1; }
tail -2 $out/run_cnn_tune_$B.log | cut -c1-200
python3 tools/op_times.py > $out/op_times_$B.log 2>&1 || { tail -5 $out/op_times_$B.log; exit 1; }
tail -1 $out/op_times_$B.log
pass() {   # name counters...
  name=$1; shift
  rm -rf $out/pmc_${name}_$B
  rocprofv3 --pmc "$@" --kernel-trace -d $out/pmc_${name}_$B --output-format csv -- python3 tools/run_cnn.py > $out/pmc_${name}_$B.log 2>&1 \
    || { echo "pass $name failed"; tail -3 $out/pmc_${name}_$B.log; }
}
pass mfma SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE
pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM
pass l2 TCC_HIT_sum TCC_MISS_sum
python3 tools/pmc_mfma.py $out $B
