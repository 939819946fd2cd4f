# Round-3 micro-benchmark evidence quoted in DESIGN.md (one gpurun call): outputs under gpurun_out/r3e, copied to profiles/ by hand
O=gpurun_out/r3e; mkdir -p $O
python tools/gemm_shapes.py > $O/gemm_shapes.txt 2>/dev/null
python tools/fold_bench.py 2>/dev/null | grep -v amdgpu > $O/ln_fold_bench.txt
python tools/attn_bench.py 2>/dev/null | grep -v amdgpu > $O/attn_bench.txt
bash tools/relpos_prof.sh 2>/dev/null | grep -v passed > $O/relpos_kernels.txt
python tools/lora_bench.py 2>/dev/null | grep lora > $O/lora_grad_bench.txt
python tools/frontend_bench.py 2>/dev/null | tail -1 > $O/frontend_bench.txt
timeout 300 tools/ablate/pp_lab.bin 5 2>&1 | grep -E "QUAD|S1 with mfma 16x16x32 prio|DUAL 2x" > $O/gemm_lab_quad.txt
bash tools/ab_env.sh SED_LN_FOLD=0 SED_RELPOS_DKDV=recompute SED_LN_BWD16=0 SED_GEMM_RB=0 SED_GEMM_STAGGER=800 SED_RELPOS_FWD_NW=8 SED_GEMM_PERSIST=0 > $O/instep_ab.txt 2>&1
ls -la $O
