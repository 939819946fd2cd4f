mkdir -p gpurun_out/r2; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in s_ilp s_agpr s_iter s_mem; do
  export SED_HIP_LIB=$GRAFT_REPO_ROOT/tools/ablate/variants/$v.so
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2/rpx_$v -o p -- python tools/relpos_bench.py > /dev/null 2>&1
  rm -f gpurun_out/r2/rpx_$v/p_kernel_trace.csv
done
