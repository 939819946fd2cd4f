#!/bin/bash
# build_variant.sh NAME [extra hipcc flags...] : full library build with extra -D flags -> tools/ablate/variants/NAME.so
set -e
cd "$(dirname "$0")/../.."
name=$1; shift
out=tools/ablate/variants/$name.so
mkdir -p tools/ablate/variants /tmp/variant_$name
for f in gemm attention relpos_attention norm_elem frontend pmam dasm; do
  ff=""; case $f in attention|relpos_attention) ff="-ffast-math -fno-finite-math-only -mllvm -amdgpu-mfma-vgpr-form=1";; gemm|pmam) ff="-ffast-math -fno-finite-math-only";; esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result \
     $ff "$@" -c transformer4sed_amd/csrc/$f.hip -o /tmp/variant_$name/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/variant_$name/*.o -o $out
echo built $out
