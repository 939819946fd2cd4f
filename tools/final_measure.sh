# Round-end measurement set (run through gpurun): GPU tests, smoke, the full bench line, the other modes, rocprofv3 kernel stats and the
# PMC passes (separate runs, counters only).  The kernel-stats runs execute 2 warm-up + 4 timed steps = 6 steps (no instrumented extra steps).  Everything lands in gpurun_out/r3h; the summaries to keep are copied to profiles/ by hand.
mkdir -p gpurun_out/r3h; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r3h
python -m pytest tests -m gpu -q > $O/gpu_tests.txt 2>&1; grep -E "passed|failed|error" $O/gpu_tests.txt | tail -2
python __graft_entry__.py smoke > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
python bench.py > $O/bench_line.json 2> $O/bench_err.txt; cut -c1-200 $O/bench_line.json
for m in pretrain finetune1 pmam val; do python bench.py --mode $m --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_$m.json; cut -c1-160 $O/bench_$m.json; done
SED_OVERLAP_TEACHER=0 SED_DW_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timer > $O/prof_bench_line.json 2>/dev/null
rm -f $O/prof/p_kernel_trace.csv
SED_DW_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_pmam -o p -- python bench.py --mode pmam --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timer > /dev/null 2>&1
rm -f $O/prof_pmam/p_kernel_trace.csv
SED_DW_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_pretrain -o p -- python bench.py --mode pretrain --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timer > /dev/null 2>&1
rm -f $O/prof_pretrain/p_kernel_trace.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -o f -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timer > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -o w -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timer > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_m -o m -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timer > /dev/null 2>&1
SED_OVERLAP_TEACHER=0 SED_DW_STREAM=0 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_a -o a -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timer > /dev/null 2>&1
python tools/gemm_traffic.py $O/pmc_f/f_counter_collection.csv $O/pmc_w/w_counter_collection.csv $O/r3_gemm_traffic_finetune2.json > /dev/null
python tools/mfma_util.py $O/pmc_m/m_counter_collection.csv $O/r3_gemm_mfma_busy_finetune2.json > /dev/null
python tools/pmc_summary.py $O/pmc_a/a_counter_collection.csv $O/r3_attn_pmc.json mhsa relpos logmel absmax layernorm gemm_tn > /dev/null
rm -rf $O/pmc_f $O/pmc_w $O/pmc_m $O/pmc_a
python tools/frontend_bench.py > $O/frontend_bench.txt 2>&1
ls $O
