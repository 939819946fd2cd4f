#!/bin/bash
# The GPU suite once per user-selectable non-default value of every environment switch the product still reads (round 6: ten of them;
# README "Environment switches").  -> gpurun_out/<tag>_nondefault_suite.txt     usage: bash tools/nondefault_suite.sh r6
TAG=${1:-r6}; OUT=gpurun_out/${TAG}_nondefault_suite.txt; : > $OUT
for env in "SED_DDP_COMM_DTYPE=bf16" "SED_GEMM_DYN=0" "SED_LN_FOLD=0" "SED_ENC_W2=f16" "SED_ENC_W2=f8:qkv,fc2" "SED_DW_STREAM=0" "SED_OVERLAP_TEACHER=0" \
           "SED_GEMM_CUS=224" "SED_GEMM_RB=7" "SED_HOST_THREADS=4"; do
  echo "== $env" >> $OUT
  # (the RCCL banner of the DDP tests follows pytest's own summary on stdout: pick the summary lines, not the last two)
  env $env python -m pytest tests -m gpu -q 2>&1 | grep -E "^(FAILED|ERROR) |[0-9]+ (passed|failed)" | cut -c1-400 >> $OUT
done
cat $OUT
