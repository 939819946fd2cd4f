// Developer tool: matrix-pipe rate of the block-scaled fp8 MFMA (v_mfma_scale_f32_16x16x128_f8f6f4, unit e8m0 scales) against the f16
// MFMA the GEMMs use (v_mfma_f32_16x16x32_f16), plus a numerics probe (e4m3 products accumulate exactly in fp32).  Background: the
// evaluation-mode encoder multiplies every activation with W_hi AND W_lo = W - f16(W) in f16 (two K passes); the lo product only needs
// ~4 significant bits of either operand (it is 2^-11 of the result), i.e. it could run on the fp8 path (validation).
//   hipcc --offload-arch=gfx950 -O3 tools/ablate/fp8_mfma_bench.hip -o tools/ablate/fp8_mfma_bench.bin && tools/ablate/fp8_mfma_bench.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s\n", hipGetErrorString(e_)); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

template <int KIND>      // 0: f16 16x16x32   1: fp8 (e4m3) 16x16x128, scales 2^0
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned abits, unsigned bbits) {
    f32x4 c[4];
    for (int i = 0; i < 4; ++i) c[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    f16x8 fa, fb;
    for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)1.0f; fb[i] = (_Float16)0.5f; }
    i32x8 qa, qb;
    for (int i = 0; i < 8; ++i) { qa[i] = (int)abits; qb[i] = (int)bbits; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (KIND == 0) c[j & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, c[j & 3], 0, 0, 0);
            else c[j & 3] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(qa, qb, c[j & 3], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    float* out;
    const int blocks = 256 * 4, iters = 4000;
    CHECK(hipMalloc(&out, blocks * 256 * sizeof(float)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float h[4];
    for (int kind = 0; kind < 2; ++kind) {
        for (int rep = 0; rep < 2; ++rep) {
            CHECK(hipEventRecord(e0));
            if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, iters, 0u, 0u);
            else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 0x38383838u, 0x30303030u);   // e4m3: 1.0 = 0x38, 0.5 = 0x30
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double kdim = kind == 0 ? 32 : 128;
            const double flops = 2.0 * 16 * 16 * kdim * 16.0 * iters * (blocks * 4.0);
            if (rep == 1) {
                CHECK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
                // every accumulator element: iters * 4 MFMAs per chain * K * (1.0 * 0.5); a lane sums 4 chains x 4 elements
                const double want = 16.0 * (double)iters * 4 * kdim * 0.5;
                printf("%s  %8.2f ms  %8.1f TFLOP/s   lane sum %.6g (exact: %.6g)\n", kind == 0 ? "f16 16x16x32      " : "fp8 16x16x128 (MX)", ms,
                       flops / ms * 1e-9, (double)h[0], want);
            }
        }
    }
    // numerics of one fp8 product: e4m3 1.75 (0x3e) x e4m3 -0.40625 (0xad = -(1 + 5/8) 2^-2) over K = 128
    hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, out, 1, 0x3e3e3e3eu, 0xadadadadu);
    CHECK(hipMemcpy(h, out, sizeof(float), hipMemcpyDeviceToHost));
    printf("fp8 probe: 4 chains x 4 MFMAs of K = 128 with a = 1.75, b = -0.40625, 4 elements per lane and chain: lane sum %.6f (exact %.6f)\n", h[0],
           16.0 * 4 * 128 * 1.75 * -0.40625);
    return 0;
}
