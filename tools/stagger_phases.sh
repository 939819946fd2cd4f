mkdir -p gpurun_out/s2e
for cfg in "0 2" "2600 8" "2600 4" "1300 8" "2600 16" "2000 2"; do set -- $cfg; echo "== STAGGER=$1 PHASES=$2"; SED_GEMM_STAGGER=$1 SED_GEMM_STAGGER_PHASES=$2 python tools/gemm_shapes.py 2>&1 | grep "^finetune\|211904\|epi3 \|qkv3 " | cut -c1-105; done > gpurun_out/s2e/stagger.txt 2>&1
cat gpurun_out/s2e/stagger.txt
