#!/bin/bash
# attn_variants.sh : builds attention.hip with different -D / compiler flags, links each against the other objects of the last library build,
# and (on the GPU box) runs tools/attn_bench.py on every variant.  Usage:  build: tools/attn_variants.sh build ;  run: tools/attn_variants.sh run
cd "$(dirname "$0")/.."
V=tools/ablate/variants
mkdir -p $V
FF="-ffast-math -fno-finite-math-only -mllvm -amdgpu-mfma-vgpr-form=1"
names=(main noxcd)
flags=("" "-DATT_NO_XCD")
if [ "$1" = build ]; then
  for i in "${!names[@]}"; do
    ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result $FF ${flags[$i]} -c transformer4sed_amd/csrc/attention.hip -o /tmp/attn_${names[$i]}.o &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/attn_${names[$i]}.o transformer4sed_amd/build/gemm.o transformer4sed_amd/build/relpos_attention.o \
        transformer4sed_amd/build/norm_elem.o transformer4sed_amd/build/frontend.o transformer4sed_amd/build/pmam.o -o $V/attn_${names[$i]}.so && echo built ${names[$i]} ) &
  done
  wait
else
  for n in "${names[@]}"; do echo "== $n"; SED_HIP_LIB=$PWD/$V/attn_$n.so REPS=10 python tools/attn_bench.py 2>&1 | grep mhsa; done
fi
