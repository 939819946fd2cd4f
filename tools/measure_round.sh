# Round-end measurement set, ONE gpurun call:   gpurun --timeout 3000 -- 'bash tools/measure_round.sh r4'
# Everything lands in gpurun_out/<tag>/ ; the summaries to keep are copied to profiles/<tag>_* by hand (profiles/README.md lists them).
#   1. GPU tests + smoke          2. the bench line (finetune2) and the other modes         3. rocprofv3 kernel stats, one stream, 3 modes
#   4. PMC passes (counters only, separate runs): FETCH_SIZE, WRITE_SIZE, MFMA busy, attention / rel-pos / frontend / LayerNorm counters
#   5. per-shape GEMM table, micro-benchmarks (attention, rel-pos, frontend)
TAG=${1:-r6}; O=gpurun_out/$TAG; mkdir -p $O; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
HEAD=$(cat .gpurun_head 2>/dev/null || echo unknown)
python -m pytest tests -m gpu -q > $O/gpu_tests.txt 2>&1; grep -E "passed|failed|error" $O/gpu_tests.txt | tail -2
python __graft_entry__.py smoke > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
python bench.py > $O/bench_line.json 2> $O/bench_err.txt; cut -c1-200 $O/bench_line.json
for m in pretrain finetune1 pmam val dasm dasm_train; do python bench.py --mode $m --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_$m.json; cut -c1-160 $O/bench_$m.json; done
python bench.py --mode dasm --dasm-queries 407 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_dasm407.json; cut -c1-160 $O/bench_dasm407.json
for st in finetune2 pretrain; do taskset -c 0-1 python bench.py --mode pipe --pipe-step $st --steps 12 --warmup 3 --no-cpu-baseline --no-kernel-timer 2>/dev/null | tail -1 > $O/bench_pipe_${st}_2cpu.json; cut -c1-160 $O/bench_pipe_${st}_2cpu.json; done
for m in finetune2 pretrain pmam dasm_train; do
  SED_OVERLAP_TEACHER=0 SED_DW_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$m -o p -- python bench.py --mode $m --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timer > /dev/null 2>&1
  rm -f $O/prof_$m/p_kernel_trace.csv
done
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timer"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -o f -- $B > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -o w -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_m -o m -- $B > /dev/null 2>&1
SED_OVERLAP_TEACHER=0 SED_DW_STREAM=0 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_a -o a -- $B > /dev/null 2>&1
python tools/gemm_traffic.py $O/pmc_f/f_counter_collection.csv $O/pmc_w/w_counter_collection.csv $O/gemm_traffic_finetune2.json $HEAD > /dev/null
python tools/mfma_util.py $O/pmc_m/m_counter_collection.csv $O/gemm_mfma_busy_finetune2.json $HEAD > /dev/null
python tools/pmc_summary.py $O/pmc_a/a_counter_collection.csv $O/attn_pmc.json mhsa relpos logmel absmax layernorm gemm_tn > /dev/null
rm -rf $O/pmc_f $O/pmc_w $O/pmc_m $O/pmc_a
python tools/gemm_shapes.py > $O/gemm_shapes.txt 2>/dev/null
REPS=10 python tools/attn_bench.py > $O/attn_bench.txt 2>&1
python tools/frontend_bench.py > $O/frontend_bench.txt 2>&1
python tools/dasm_bench.py 24 16 64 407 > $O/dasm_bench.txt 2>&1
SED_DW_STREAM=0 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_d -o a -- python bench.py --mode dasm_train --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timer > /dev/null 2>&1
python tools/pmc_summary.py $(find $O/pmc_d -name "*counter_collection.csv") $O/dasm_pmc.json xattn gemm_f32 logmel > /dev/null; rm -rf $O/pmc_d
bash tools/gemm_traffic_shapes.sh $TAG > /dev/null 2>&1
bash tools/nondefault_suite.sh $TAG > /dev/null 2>&1
ls $O
