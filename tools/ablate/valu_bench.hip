// Developer tool: issue rate of the VALU instructions the attention softmax is made of (v_exp_f32, v_pk_fma_f32, v_max3_f32, v_cvt_pk_f16_f32)
// against v_mfma_f32_32x32x16_f16, one wave per SIMD and three waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/ablate/valu_bench.hip -o tools/ablate/valu_bench.bin && tools/ablate/valu_bench.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s\n", hipGetErrorString(e_)); return 1; } } while (0)
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int KIND>
__global__ void k(float* out, int iters) {
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 0.001f + i;
    f32x2 p[4];
    for (int i = 0; i < 4; ++i) p[i] = f32x2{a[2 * i], a[2 * i + 1]};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (KIND == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(a[u]));
            if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(p[u & 3]));
            if (KIND == 2) asm volatile("v_max3_f32 %0, %0, %0, %0" : "+v"(a[u]));
            if (KIND == 3) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a[u]));
            if (KIND == 4) asm volatile("v_cvt_pk_f16_f32 %0, %0, %0" : "+v"(a[u]));
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += a[i];
    for (int i = 0; i < 4; ++i) s += p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    float* out; CHECK(hipMalloc(&out, 256 * 4 * 1024 * 4));
    const char* names[5] = {"v_exp_f32", "v_pk_fma_f32", "v_max3_f32", "v_fma_f32", "v_cvt_pk_f16_f32"};
    void (*ks[5])(float*, int) = {k<0>, k<1>, k<2>, k<3>, k<4>};
    const int iters = 20000;
    for (int waves = 1; waves <= 3; waves += 2)
        for (int kind = 0; kind < 5; ++kind) {
            hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
            ks[kind]<<<256, 256 * waves>>>(out, 100);
            CHECK(hipEventRecord(e0));
            ks[kind]<<<256, 256 * waves>>>(out, iters);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            // instructions per SIMD = waves * iters * 8; report ns per wave-instruction per SIMD (at 2.4 GHz: 1.67 ns = 4 cycles)
            printf("%d wave(s)/SIMD %-18s %.3f ms  %.2f ns per wave-instruction per SIMD\n", waves, names[kind], ms, ms * 1e6 / ((double)waves * iters * 8));
        }
    return 0;
}
