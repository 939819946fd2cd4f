// Developer tool (not part of the product library): ablation variants of the 128x128x64 GEMM main loop.
//   ABL bit0: skip the per-K-tile DMA (only the first tile is loaded)   bit1: skip ds_reads (fragments loaded once)
//   ABL bit2: skip barriers
#include "../../transformer4sed_amd/csrc/common.h"
#define TILE 128
#define BK 64
template <int ABL, int WAVES_N>
__global__ __launch_bounds__(256) void abl_kernel(const bf16_t* A, const bf16_t* B, bf16_t* C, int M, int N, int K) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2][2][TILE * BK * 2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int ntn = N / TILE, ntm = (M + TILE - 1) / TILE, nwg = ntm * ntn;
    const int t = xcd_remap(blockIdx.x, nwg);
    const int group_size = 8 * ntn, gid = t / group_size, first_m = gid * 8;
    const int gm = (ntm - first_m) < 8 ? (ntm - first_m) : 8;
    const int tin = t - gid * group_size;
    const int m0 = (first_m + tin % gm) * TILE, n0 = (tin / gm) * TILE;
    const int nk = K / BK;
    const int prow = lane >> 3, pch = lane & 7;
    const bf16_t* asrc[4];
    const bf16_t* bsrc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (wave * 4 + i) * 8 + prow;
        const int cl = pch ^ ((row >> 1) & 7);
        int am = m0 + row; am = am < M ? am : M - 1;
        asrc[i] = A + (size_t)am * K + cl * 8;
        bsrc[i] = B + (size_t)(n0 + row) * K + cl * 8;
    }
#define DMA(kt, buf)                                                                                                    \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                     \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(asrc[i] + (size_t)(kt) * BK),  \
                                         (__attribute__((address_space(3))) void*)(&lds[buf][0][(wave * 4 + i) * 1024]),\
                                         16, 0, 0);                                                                     \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bsrc[i] + (size_t)(kt) * BK),  \
                                         (__attribute__((address_space(3))) void*)(&lds[buf][1][(wave * 4 + i) * 1024]),\
                                         16, 0, 0);                                                                     \
    }
    f32x16_t acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int lr = lane & 31, lg = lane >> 5;
    int arow[2], brow[2];
    for (int i = 0; i < 2; ++i) { arow[i] = wm * 64 + i * 32 + lr; brow[i] = wn * 64 + i * 32 + lr; }
    DMA(0, 0);
    __syncthreads();
    s16x8_t caf[2], cbf[2];
    for (int i = 0; i < 2; ++i) {
        caf[i] = *reinterpret_cast<const s16x8_t*>(lds[0][0] + arow[i] * 128 + ((lg ^ ((arow[i] >> 1) & 7)) << 4));
        cbf[i] = *reinterpret_cast<const s16x8_t*>(lds[0][1] + brow[i] * 128 + ((lg ^ ((brow[i] >> 1) & 7)) << 4));
    }
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (!(ABL & 1)) { if (kt + 1 < nk) DMA(kt + 1, buf ^ 1); }
        const unsigned char* la = lds[(ABL & 1) ? 0 : buf][0];
        const unsigned char* lb = lds[(ABL & 1) ? 0 : buf][1];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int ch = 2 * s + lg;
            s16x8_t af[2], bfr[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (ABL & 2) { af[i] = caf[i]; bfr[i] = cbf[i]; }
                else {
                    af[i] = *reinterpret_cast<const s16x8_t*>(la + arow[i] * 128 + ((ch ^ ((arow[i] >> 1) & 7)) << 4));
                    bfr[i] = *reinterpret_cast<const s16x8_t*>(lb + brow[i] * 128 + ((ch ^ ((brow[i] >> 1) & 7)) << 4));
                }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) { if (ABL & 8) { asm volatile("" ::"v"(af[i]), "v"(bfr[j])); } else acc[i][j] = mfma32t<true>(bfr[j], af[i], acc[i][j]); }
        }
        if (!(ABL & 4)) __syncthreads();
    }
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + wm * 64 + i * 32 + lr;
        if (m >= M) continue;
        for (int j = 0; j < 2; ++j) for (int q = 0; q < 4; ++q) {
            const int n = n0 + wn * 64 + j * 32 + 8 * q + 4 * lg;
            uint2 pk;
            pk.x = pack2<true>(acc[i][j][4 * q], acc[i][j][4 * q + 1]);
            pk.y = pack2<true>(acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
            *reinterpret_cast<uint2*>(C + (size_t)m * N + n) = pk;
        }
    }
}
extern "C" int abl_launch(int abl, const void* A, const void* B, void* C, int M, int N, int K, hipStream_t s) {
    dim3 grid(((M + 127) / 128) * (N / 128));
#define L(X) case X: hipLaunchKernelGGL((abl_kernel<X, 2>), grid, dim3(256), 0, s, (const bf16_t*)A, (const bf16_t*)B, (bf16_t*)C, M, N, K); break;
    switch (abl) { L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7) L(8) L(10) L(14) }
    return (int)hipGetLastError();
}

// ---- v2-style: 128 x 256 tile, 8 waves, NST LDS stages, DMA (NST-1) tiles ahead, counted vmcnt + raw barrier
template <int ABL, int NST>
__global__ __launch_bounds__(512) void abl2_kernel(const bf16_t* A, const bf16_t* B, bf16_t* C, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds2[];
    const int STAGE = 48 * 1024;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int ntn = N / 256, ntm = (M + 127) / 128, nwg = ntm * ntn;
    const int t = xcd_remap(blockIdx.x, nwg);
    const int group_size = 8 * ntn, gid = t / group_size, first_m = gid * 8;
    const int gm = (ntm - first_m) < 8 ? (ntm - first_m) : 8;
    const int tin = t - gid * group_size;
    const int m0 = (first_m + tin % gm) * 128, n0 = (tin / gm) * 256;
    const int nk = K / BK;
    const int prow = lane >> 3, pch = lane & 7;
    const bf16_t* src[6];
    int dst[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int p = wave * 6 + i;
        const bool isB = p >= 16;
        const int row = (isB ? p - 16 : p) * 8 + prow;
        const int cl = pch ^ ((row >> 1) & 7);
        int am = m0 + row; am = am < M ? am : M - 1;
        src[i] = isB ? B + (size_t)(n0 + row) * K + cl * 8 : A + (size_t)am * K + cl * 8;
        dst[i] = (isB ? 16384 : 0) + (isB ? p - 16 : p) * 1024;
    }
#define DMA2(kt, stage)                                                                                                  \
    _Pragma("unroll") for (int i = 0; i < 6; ++i)                                                                        \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + (size_t)(kt) * BK),    \
                                         (__attribute__((address_space(3))) void*)(lds2 + (stage) * STAGE + dst[i]), 16, 0, 0);
    f32x16_t acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int lr = lane & 31, lg = lane >> 5;
    int arow[2], brow[2];
    for (int i = 0; i < 2; ++i) { arow[i] = wm * 64 + i * 32 + lr; brow[i] = wn * 64 + i * 32 + lr; }
#pragma unroll
    for (int p = 0; p < NST - 1; ++p) if (p < nk) { DMA2(p, p); }
    int stage = 0;
    for (int it = 0; it < nk; ++it) {
        // outstanding newer tiles after this wait: min(NST-2, nk-1-it)
        const int newer = (nk - 1 - it) < (NST - 2) ? (nk - 1 - it) : (NST - 2);
        if (newer >= 2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if (newer == 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (it + NST - 1 < nk) {
            int st2 = stage + NST - 1; st2 = st2 >= NST ? st2 - NST : st2;
            if (!(ABL & 1)) { DMA2(it + NST - 1, st2); }
        }
        const unsigned char* la = lds2 + ((ABL & 1) ? 0 : stage) * STAGE;
        const unsigned char* lb = la + 16384;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int ch = 2 * s + lg;
            s16x8_t af[2], bfr[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                af[i] = *reinterpret_cast<const s16x8_t*>(la + arow[i] * 128 + ((ch ^ ((arow[i] >> 1) & 7)) << 4));
                bfr[i] = *reinterpret_cast<const s16x8_t*>(lb + brow[i] * 128 + ((ch ^ ((brow[i] >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) { if (ABL & 8) { asm volatile("" ::"v"(af[i]), "v"(bfr[j])); } else acc[i][j] = mfma32t<true>(bfr[j], af[i], acc[i][j]); }
        }
        stage = stage + 1 == NST ? 0 : stage + 1;
    }
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + wm * 64 + i * 32 + lr;
        if (m >= M) continue;
        for (int j = 0; j < 2; ++j) for (int q = 0; q < 4; ++q) {
            const int n = n0 + wn * 64 + j * 32 + 8 * q + 4 * lg;
            uint2 pk;
            pk.x = pack2<true>(acc[i][j][4 * q], acc[i][j][4 * q + 1]);
            pk.y = pack2<true>(acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
            *reinterpret_cast<uint2*>(C + (size_t)m * N + n) = pk;
        }
    }
}
extern "C" int abl2_launch(int abl, int nst, const void* A, const void* B, void* C, int M, int N, int K, hipStream_t s) {
    dim3 grid(((M + 127) / 128) * (N / 256));
#define L2(X, S) if (abl == X && nst == S) { hipFuncSetAttribute((const void*)abl2_kernel<X, S>, hipFuncAttributeMaxDynamicSharedMemorySize, S * 48 * 1024); \
    hipLaunchKernelGGL((abl2_kernel<X, S>), grid, dim3(512), S * 48 * 1024, s, (const bf16_t*)A, (const bf16_t*)B, (bf16_t*)C, M, N, K); }
    L2(0, 2) L2(0, 3) L2(1, 3) L2(8, 3) L2(8, 2)
    return (int)hipGetLastError();
}
