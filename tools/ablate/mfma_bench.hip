// Developer tool: sustained MFMA rate / clock under the power limit for the two f16 MFMA shapes, on random operands held in
// registers (no LDS / memory traffic in the loop).  8 waves per CU, 2 per SIMD, alternating like the GEMM's wave rows.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "../../transformer4sed_amd/csrc/common.h"
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(4))) float f32x4v;
__device__ unsigned long long g_clk[2];

template <int SHAPE, int NW>
__global__ __launch_bounds__(NW * 64) void mfma_kernel(const bf16_t* __restrict__ src, float* __restrict__ out, int iters, int zero) {
    const int tid = threadIdx.x;
    f16x8_t a[8], b[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = *reinterpret_cast<const f16x8_t*>(src + ((size_t)(blockIdx.x * 512 + tid) * 12 + i) * 8 % (1 << 20));
#pragma unroll
    for (int i = 0; i < 4; ++i) b[i] = *reinterpret_cast<const f16x8_t*>(src + ((size_t)(blockIdx.x * 512 + tid) * 12 + 8 + i) * 8 % (1 << 20));
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    float res = 0.f;
    if (SHAPE == 32) {
        f32x16_t acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(i + ks) & 7], b[(i + ks) & 3], acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) res += acc[i][0] + acc[i][7];
    } else {
        f32x4v acc[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) acc[i] = f32x4v{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 32; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(i + ks) & 7], b[(i + ks) & 3], acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 32; ++i) res += acc[i][0] + acc[i][3];
    }
    if (blockIdx.x == 17 && tid == 0) { g_clk[0] = __builtin_readcyclecounter() - c0; g_clk[1] = wall_clock64() - r0; }
    if (zero) out[blockIdx.x * NW * 64 + tid] = res;
}
__global__ void fill_kernel(bf16_t* p, size_t n, unsigned seed, float scale, int mode) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)(i * 2654435761u) ^ seed;
        h ^= h >> 15; h *= 0x2c1b3c6du; h ^= h >> 12; h *= 0x297a2d39u; h ^= h >> 15;
        float v = ((h & 0xffff) / 32768.0f - 1.0f) * scale;
        if (mode == 1) v = 0.f;
        if (mode == 2) v = fabsf(v);
        p[i] = f2h(v);
    }
}
template <int SHAPE>
static void run(const char* name, bf16_t* src, float* out, int iters) {
    mfma_kernel<SHAPE, 8><<<256, 512>>>(src, out, 100, 1);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0));
    mfma_kernel<SHAPE, 8><<<256, 512>>>(src, out, iters, 1);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long hc[2];
    CHECK(hipMemcpyFromSymbol(hc, HIP_SYMBOL(g_clk), sizeof(hc)));
    const double flops = 256.0 * 8 * iters * 32.0 * 32768.0;   // per iteration per wave: 32 MFMA-32x32x16 equivalents
    printf("%-28s %8.3f ms %8.1f TF/s  clk %.2f GHz  util %.1f%%\n", name, ms, flops / ms / 1e9, (double)hc[0] / (hc[1] * 10.0),
           100.0 * (2.0 * iters * 32 * 32) / (double)hc[0]);
}
int main() {
    bf16_t* src; float* out;
    CHECK(hipMalloc(&src, 2 << 20)); CHECK(hipMalloc(&out, 256 * 512 * 4));
    for (int mode = 0; mode < 3; ++mode) {
        fill_kernel<<<256, 256>>>(src, 1 << 20, 0x777u, mode == 0 ? 1.0f : 1.0f, mode);
        printf("operands: %s\n", mode == 0 ? "uniform [-1,1)" : mode == 1 ? "zeros" : "uniform [0,1) (sign bit constant)");
        for (int rep = 0; rep < 2; ++rep) {
            run<32>("v_mfma_f32_32x32x16_f16", src, out, 20000);
            run<16>("v_mfma_f32_16x16x32_f16", src, out, 20000);
        }
    }
    return 0;
}
