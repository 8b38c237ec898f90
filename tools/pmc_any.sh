#!/bin/bash
# three rocprofv3 --pmc passes (each with --kernel-trace only) over a python script; pmc_any.sh <out> <script> [filter]
out=$1; script=$2; flt=$3; mkdir -p $out
export TMPDIR=/tmp
pass() {
  name=$1; shift
  rm -rf $out/pmc_$name
  timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace -d $out/pmc_$name --output-format csv -- python3 $script > $out/pmc_$name.log 2>&1 \
    || { echo "pass $name failed"; tail -3 $out/pmc_$name.log; }
}
pass mfma SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE
pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU
pass l2 TCC_HIT_sum TCC_MISS_sum
python3 tools/pmc_by_kernel.py $out $flt | cut -c1-420
