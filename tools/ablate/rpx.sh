mkdir -p gpurun_out/r2; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "relpos or rel_pos or decoder or xl" 2>&1 | grep -E "passed|failed|Error|error" | tail -5
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2/rpx_new -o p -- python tools/relpos_bench.py > /dev/null 2>&1
rm -f gpurun_out/r2/rpx_new/p_kernel_trace.csv
