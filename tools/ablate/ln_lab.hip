// Developer tool: LayerNorm-forward variants (fp32 in -> f16 out + mean / rstd), cold input (a 1 GiB buffer is rewritten between
// timed launches), M = 38080 and 211904 rows of 768.
//   RW rows per wave, NT_LD / NT_ST non-temporal loads / stores, PERSIST grid-stride over row groups
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "../../transformer4sed_amd/csrc/common.h"
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
#define DM 768
typedef float f4v __attribute__((ext_vector_type(4)));
typedef unsigned u2v __attribute__((ext_vector_type(2)));

template <int RW, bool NT_LD, bool NT_ST, bool PERSIST, int NWAVE>
__global__ __launch_bounds__(NWAVE * 64) void ln_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        bf16_t* __restrict__ y, float* __restrict__ mean, float* __restrict__ rstd, int M) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f4v g[3], b[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) { g[i] = reinterpret_cast<const f4v*>(gamma)[lane + 64 * i]; b[i] = reinterpret_cast<const f4v*>(beta)[lane + 64 * i]; }
    const int ngroups = (M + RW - 1) / RW;
    for (int grp = blockIdx.x * NWAVE + wave; grp < ngroups; grp += PERSIST ? gridDim.x * NWAVE : ngroups) {
        const int row0 = grp * RW;
        f4v r[RW][3];
#pragma unroll
        for (int k = 0; k < RW; ++k) {
            const int row = row0 + k < M ? row0 + k : M - 1;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const f4v* p = reinterpret_cast<const f4v*>(x + (size_t)row * DM) + lane + 64 * i;
                r[k][i] = NT_LD ? __builtin_nontemporal_load(p) : *p;
            }
        }
        float mu[RW], rs[RW];
#pragma unroll
        for (int k = 0; k < RW; ++k) {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 3; ++i) s += r[k][i][0] + r[k][i][1] + r[k][i][2] + r[k][i][3];
            mu[k] = wave_sum(s) * (1.0f / DM);
        }
#pragma unroll
        for (int k = 0; k < RW; ++k) {
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int c = 0; c < 4; ++c) { const float d = r[k][i][c] - mu[k]; q += d * d; }
            rs[k] = rsqrtf(wave_sum(q) * (1.0f / DM) + 1e-6f);
        }
#pragma unroll
        for (int k = 0; k < RW; ++k) {
            const int row = row0 + k;
            if (row >= M) break;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                float o[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) o[c] = (r[k][i][c] - mu[k]) * rs[k] * g[i][c] + b[i][c];
                u2v pk = {pack2<true>(o[0], o[1]), pack2<true>(o[2], o[3])};
                u2v* q = reinterpret_cast<u2v*>(y + (size_t)row * DM) + lane + 64 * i;
                if (NT_ST) __builtin_nontemporal_store(pk, q); else *q = pk;
            }
            if (lane == 0) { mean[row] = mu[k]; rstd[row] = rs[k]; }
        }
    }
}
__global__ void fillk(float* p, size_t n, float v) { for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v + (float)(i & 1023) * 1e-3f; }

template <int RW, bool NT_LD, bool NT_ST, bool PERSIST, int NWAVE>
void run(const char* name, const float* x, const float* g, const float* b, bf16_t* y, float* mu, float* rs, int M, float* flush) {
    const int ngroups = (M + RW - 1) / RW;
    const int grid = PERSIST ? 256 * (32 / NWAVE) : (ngroups + NWAVE - 1) / NWAVE;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float tot = 0;
    for (int it = 0; it < 5; ++it) {
        fillk<<<2048, 256>>>(flush, (size_t)256 << 20, (float)it);
        CHECK(hipEventRecord(e0));
        ln_kernel<RW, NT_LD, NT_ST, PERSIST, NWAVE><<<grid, NWAVE * 64>>>(x, g, b, y, mu, rs, M);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (it) tot += ms;
    }
    printf("M=%6d %-44s %7.1f us  %5.2f TB/s\n", M, name, tot / 4 * 1e3, (double)M * DM * 6 / (tot / 4) / 1e9);
}
int main() {
    float* flush; CHECK(hipMalloc(&flush, (size_t)1 << 30));
    for (int M : {38080, 211904}) {
        float *x, *g, *b, *mu, *rs; bf16_t* y;
        CHECK(hipMalloc(&x, (size_t)M * DM * 4)); CHECK(hipMalloc(&y, (size_t)M * DM * 2)); CHECK(hipMalloc(&g, DM * 4)); CHECK(hipMalloc(&b, DM * 4));
        CHECK(hipMalloc(&mu, M * 4)); CHECK(hipMalloc(&rs, M * 4));
        fillk<<<2048, 256>>>(x, (size_t)M * DM, 0.5f); fillk<<<4, 256>>>(g, DM, 1.f); fillk<<<4, 256>>>(b, DM, 0.f);
        run<2, false, false, false, 4>("RW2 4 waves (shipped)", x, g, b, y, mu, rs, M, flush);
        run<1, false, false, false, 4>("RW1 4 waves", x, g, b, y, mu, rs, M, flush);
        run<4, false, false, false, 4>("RW4 4 waves", x, g, b, y, mu, rs, M, flush);
        run<2, true, false, false, 4>("RW2 nt loads", x, g, b, y, mu, rs, M, flush);
        run<2, false, true, false, 4>("RW2 nt stores", x, g, b, y, mu, rs, M, flush);
        run<2, true, true, false, 4>("RW2 nt loads + stores", x, g, b, y, mu, rs, M, flush);
        run<2, false, false, false, 8>("RW2 8 waves", x, g, b, y, mu, rs, M, flush);
        run<2, false, false, false, 1>("RW2 1 wave", x, g, b, y, mu, rs, M, flush);
        run<2, false, false, true, 4>("RW2 persistent 2048 WGs", x, g, b, y, mu, rs, M, flush);
        run<4, true, true, true, 4>("RW4 persistent nt", x, g, b, y, mu, rs, M, flush);
        run<1, true, true, true, 8>("RW1 persistent nt 8 waves", x, g, b, y, mu, rs, M, flush);
        CHECK(hipFree(x)); CHECK(hipFree(y)); CHECK(hipFree(g)); CHECK(hipFree(b)); CHECK(hipFree(mu)); CHECK(hipFree(rs));
    }
    return 0;
}
