// Developer tool: do MFMA and VALU work overlap on one SIMD (a) across waves -- each wave alternating an MFMA phase and a VALU phase, the
// structure of the attention kernels -- and (b) inside one wave, the same instructions interleaved 1 MFMA : k VALU?
// Per iteration and wave: 16 v_mfma_f32_32x32x16_f16 (two dependent chains of 8 -> 512 matrix-pipe cycles) and NV vector instructions
// (a mix of v_exp_f32 / v_pk_fma_f32 / v_max3_f32 like the softmax: 30 % transcendental).
//   hipcc --offload-arch=gfx950 -O3 tools/ablate/overlap_bench.hip -o tools/ablate/overlap_bench.bin && tools/ablate/overlap_bench.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s\n", hipGetErrorString(e_)); return 1; } } while (0)
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// one "vector instruction" of the softmax mix; idx selects the kind and the register it works on
#ifndef VKIND
#define VKIND 0
#endif
#define VOP(i)                                                                             \
    {                                                                                      \
        if (VKIND == 0) {                                                                  \
            if ((i) % 10 < 3) asm volatile("v_exp_f32 %0, %0" : "+v"(a[(i) & 7]));         \
            else if ((i) % 10 < 7) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(p[(i) & 3])); \
            else asm volatile("v_max3_f32 %0, %0, %0, %0" : "+v"(a[(i) & 7]));              \
        }                                                                                  \
        if (VKIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(a[(i) & 7]));               \
        if (VKIND == 2) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(p[(i) & 3]));    \
        if (VKIND == 3) asm volatile("v_max3_f32 %0, %0, %0, %0" : "+v"(a[(i) & 7]));      \
        if (VKIND == 4) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a[(i) & 7]));       \
        if (VKIND == 5) asm volatile("v_add_f32 %0, %0, %0" : "+v"(a[(i) & 7]));           \
        if (VKIND == 6) asm volatile("v_mul_f32 %0, %0, %0" : "+v"(a[(i) & 7]));           \
        if (VKIND == 7) asm volatile("v_max_f32 %0, %0, %0" : "+v"(a[(i) & 7]));           \
        if (VKIND == 8) asm volatile("v_cvt_pk_f16_f32 %0, %0, %0" : "+v"(a[(i) & 7]));    \
        if (VKIND == 9) asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(p[(i) & 3]));        \
        if (VKIND == 10) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(p[(i) & 3]));       \
        if (VKIND == 11) asm volatile("v_mov_b32 %0, %0" : "+v"(a[(i) & 7]));              \
        if (VKIND == 12) asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(a[(i) & 7]) : "v"(a[((i) + 1) & 7]));  \
        /* round 5: 16-bit packed arithmetic and the half-precision transcendental (could p = 2^(s - m) run on packed halves?) */ \
        if (VKIND == 13) asm volatile("v_pk_fma_f16 %0, %0, %0, %0" : "+v"(a[(i) & 7]));   \
        if (VKIND == 14) asm volatile("v_pk_mul_f16 %0, %0, %0" : "+v"(a[(i) & 7]));       \
        if (VKIND == 15) asm volatile("v_pk_add_f16 %0, %0, %0" : "+v"(a[(i) & 7]));       \
        if (VKIND == 16) asm volatile("v_exp_f16 %0, %0" : "+v"(a[(i) & 7]));              \
        if (VKIND == 17) asm volatile("v_cvt_f32_f16 %0, %0" : "+v"(a[(i) & 7]));          \
        if (VKIND == 18) asm volatile("v_pk_add_u16 %0, %0, %0" : "+v"(a[(i) & 7]));       \
        if (VKIND == 19) asm volatile("v_dot2c_f32_f16 %0, %1, %1" : "+v"(a[(i) & 7]) : "v"(a[((i) + 1) & 7]));  \
        if (VKIND == 20) asm volatile("v_fract_f32 %0, %0" : "+v"(a[(i) & 7]));            \
    }
#define MOP(j) { if ((j) & 1) c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, c1, 0, 0, 0); else c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, c0, 0, 0, 0); }

// MODE 0: phases (16 MFMAs, then NV vector ops)   1: interleaved (1 MFMA : NV / 16 vector ops)   2: MFMAs only   3: vector ops only
template <int MODE, int NV>
__global__ void k(float* out, int iters) {
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 1e-6f + i * 0.01f;
    f32x2 p[4];
    for (int i = 0; i < 4; ++i) p[i] = f32x2{a[2 * i], a[2 * i + 1]};
    f16x8 fa, fb;
    for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)(0.001f * (threadIdx.x & 7) + 0.01f * i); fb[i] = (_Float16)(0.002f * i); }
    f32x16 c0, c1;
    for (int i = 0; i < 16; ++i) { c0[i] = 0.f; c1[i] = 0.f; }
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 16; ++j) MOP(j)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < NV; ++i) VOP(i)
            __builtin_amdgcn_sched_barrier(0);
        } else if (MODE == 1) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                MOP(j)
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < NV / 16; ++i) VOP(j * (NV / 16) + i)
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if (MODE == 2) {
#pragma unroll
            for (int j = 0; j < 16; ++j) MOP(j)
        } else {
#pragma unroll
            for (int i = 0; i < NV; ++i) VOP(i)
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += a[i];
    for (int i = 0; i < 4; ++i) s += p[i].x + p[i].y;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE, int NV>
int run(float* out, int waves, const char* what) {
    const int iters = 4000;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    k<MODE, NV><<<256, 256 * waves>>>(out, 50);
    CHECK(hipEventRecord(e0));
    k<MODE, NV><<<256, 256 * waves>>>(out, iters);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    // per SIMD: waves * iters iterations of (16 MFMAs + NV vector ops)
    const double ns_iter = ms * 1e6 / ((double)waves * iters);
    const double tf = (MODE == 3) ? 0.0 : 256.0 * 4 * waves * iters * 16 * (2.0 * 32 * 32 * 16) / (ms * 1e-3) / 1e12;
    printf("%d wave(s)/SIMD  NV=%3d  %-34s %8.3f ms  %7.1f ns per (16 MFMA + NV) per SIMD   %7.1f TFLOP/s\n", waves, NV, what, ms, ns_iter, tf);
    return 0;
}
int main() {
    float* out; CHECK(hipMalloc(&out, 256 * 4 * 1024 * 4));
    printf("VKIND %d\n", VKIND);
    for (int waves = 1; waves <= 3; waves += 2) {
        run<2, 112>(out, waves, "MFMA only");
        run<3, 112>(out, waves, "vector ops only");
        run<0, 112>(out, waves, "phases: 16 MFMA then 112 vector");
        run<1, 112>(out, waves, "interleaved 1 MFMA : 7 vector");
    }
    return 0;
}
