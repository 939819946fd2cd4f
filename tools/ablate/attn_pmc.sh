mkdir -p gpurun_out/r2; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python tools/attn_bench.py > gpurun_out/r2/attn_bench.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2/at_stats -o p -- python tools/attn_bench.py > /dev/null 2>&1
rm -f gpurun_out/r2/at_stats/p_kernel_trace.csv
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/r2/at_pmc1 -o p -- python tools/attn_bench.py > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA --kernel-trace --output-format csv -d gpurun_out/r2/at_pmc2 -o p -- python tools/attn_bench.py > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM_WR --kernel-trace --output-format csv -d gpurun_out/r2/at_pmc3 -o p -- python tools/attn_bench.py > /dev/null 2>&1
rm -f gpurun_out/r2/at_pmc*/p_kernel_trace.csv
