# The GPU suite under the user-selectable non-default switches: each line = one full `pytest -m gpu` run.
#   gpurun --timeout 1800 -- 'bash tools/nondefault_suite.sh r5'   -> gpurun_out/<tag>/nondefault_suite.txt
TAG=${1:-r5}; O=gpurun_out/$TAG; mkdir -p $O; OUT=$O/nondefault_suite.txt; : > $OUT
for env in "SED_DDP_COMM_DTYPE=bf16" "SED_GEMM_DYN=0" "SED_LN_FOLD=0" "SED_ENC_W2=f16" "SED_LN_DUAL=0"; do
  echo "== $env" >> $OUT
  env $env python -m pytest tests -m gpu -q -x > $O/nondefault_$$.log 2>&1; grep -E "passed|failed|error|Error|assert" $O/nondefault_$$.log | tail -6 >> $OUT; rm -f $O/nondefault_$$.log
done
cat $OUT
