#!/bin/bash
# producer_ablate.sh : tools/producer_bench.py over the product library and every tools/ablate/variants/abl_*.so (built beforehand with
# tools/ablate/build_variant.sh abl_X -DABL_X) -> gpurun_out/producer_ablate.txt
mkdir -p gpurun_out
{ python tools/producer_bench.py; for v in tools/ablate/variants/abl_*.so; do SED_HIP_LIB=$PWD/$v python tools/producer_bench.py; done; } 2>&1 | grep "|" | tee gpurun_out/producer_ablate.txt
