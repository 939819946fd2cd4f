// Developer tool (not part of the product library): main-loop laboratory for the 256 x 256 x 64 "ping-pong" GEMM.
// Stand-alone program (no torch): builds variants of the K loop as template instances, checks each against a naive reference on
// sampled outputs and times it on random data.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ablate/pp_lab.hip -o gpurun_out/pp_lab && gpurun_out/pp_lab
// Variant parameters:
//   SCHED 0: four quadrant phases per K tile, reads 12/4/8/0   1: quadrant phases, B0 of the next tile prefetched in P4 (8/4/8/4)
//         2: two half-tile phases per K tile (16 MFMAs per section, reads 16/8)
//         3: one phase per K tile (32 MFMAs per section, all 24 fragment reads up front)
//   ABL bit0: no DMA after the prologue   bit1: fragment reads only in the first K tile   bit2: no barriers   (timing only)
//   PRIO: s_setprio 1 around the MFMA sections      PREWAIT: lgkmcnt(0) before the first barrier instead of after it
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../../transformer4sed_amd/csrc/common.h"

#define BK 64
#define T256 256
#define LDS_BYTES (128 * 1024)
typedef __attribute__((address_space(3))) void* lds_ptr_t;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__device__ unsigned long long* g_trace = nullptr;
__device__ unsigned long long g_clk[4];
template <int SCHED, int ABL, bool PRIO, bool PREWAIT, int TRACE = 0>
__global__ __launch_bounds__(512) void pp_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, bf16_t* __restrict__ C,
                                                 int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds3[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int ntn = N / T256, ntm = (M + T256 - 1) / T256, nwg = ntm * ntn;
    const int t = xcd_remap(blockIdx.x, nwg);
    const int group_size = 4 * ntn, gid = t / group_size, first_m = gid * 4;
    const int gm = (ntm - first_m) < 4 ? (ntm - first_m) : 4;
    const int tin = t - gid * group_size;
    const int m0 = (first_m + tin % gm) * T256, n0 = (tin / gm) * T256;
    const int nk = K / BK;
    const unsigned long long clk0 = __builtin_readcyclecounter(), rt0 = wall_clock64();
    const int rows_a = (M - m0) < T256 ? (M - m0) : T256;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)(A + (size_t)m0 * K), 0, rows_a * K * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)(B + (size_t)n0 * K), 0, T256 * K * 2, 0x00020000);
    // DMA slots (16 pieces of 8 rows x 128 B each, 2 per wave): 0 = A rows of half 0, 1 = B block-0 rows (SCHED 1: rows with
    // bit 5 clear; otherwise rows 0-127), 2 = the other B rows, 3 = A rows of half 1
    const int prow = lane >> 3, pch = lane & 7;
    int vo[4][2], ld_[4][2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int p = 2 * wave + e;
        const int ra0 = (p >> 3) * 128 + (p & 7) * 8, ra1 = ra0 + 64;
        int rb0, rb1;
        if (SCHED == 1) { rb0 = (p >> 2) * 64 + (p & 3) * 8; rb1 = rb0 + 32; }
        else { rb0 = p * 8; rb1 = 128 + p * 8; }
        const int rows[4] = {ra0, rb0, rb1, ra1};
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) {
            const int row = rows[sl] + prow;
            const int cl = pch ^ ((row >> 1) & 7);
            const bool is_b = (sl == 1 || sl == 2);
            vo[sl][e] = row * K * 2 + cl * 16;
            ld_[sl][e] = (is_b ? 65536 : 0) + rows[sl] * 128;
        }
    }
#define PP_DMA(SL, KT)                                                                                                    \
    if (!(ABL & 1)) {                                                                                                     \
        const int so_ = (KT) * (BK * 2), st_ = ((KT) & 1) << 15;                                                          \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(((SL) == 1 || (SL) == 2) ? rb : ra, (lds_ptr_t)(lds3 + st_ + ld_[SL][0]), 16, vo[SL][0], so_, 0, 0); \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(((SL) == 1 || (SL) == 2) ? rb : ra, (lds_ptr_t)(lds3 + st_ + ld_[SL][1]), 16, vo[SL][1], so_, 0, 0); \
    }
#define PP_DMA_ALWAYS(SL, KT)                                                                                             \
    {                                                                                                                     \
        const int so_ = (KT) * (BK * 2), st_ = ((KT) & 1) << 15;                                                          \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(((SL) == 1 || (SL) == 2) ? rb : ra, (lds_ptr_t)(lds3 + st_ + ld_[SL][0]), 16, vo[SL][0], so_, 0, 0); \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(((SL) == 1 || (SL) == 2) ? rb : ra, (lds_ptr_t)(lds3 + st_ + ld_[SL][1]), 16, vo[SL][1], so_, 0, 0); \
    }
    f32x16_t acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int lr = lane & 31, lg = lane >> 5, sw = (lr >> 1) & 7;
    int aaddr[4], baddr[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const int c = ((2 * ks + lg) ^ sw) << 4;
        aaddr[ks] = wm * 16384 + lr * 128 + c;
        baddr[ks] = 65536 + wn * 8192 + lr * 128 + c;
    }
    s16x8_t fa[SCHED == 3 ? 4 : 2][4], fb[2][4];
    unsigned long long stamp[17];
#pragma unroll
    for (int i = 0; i < 17; ++i) stamp[i] = 0;
    const bool rd_on = !(ABL & 2);
    bool trace_on = false;
#define PP_RD_A(IH) PP_RD_A_T(IH, it)
#define PP_RD_A_T(IH, T_)                                                                                                 \
    if (rd_on || (T_) == 0) {                                                                                               \
        _Pragma("unroll") for (int ii = 0; ii < 2; ++ii)                                                                  \
            _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                                              \
                fa[(SCHED == 3 ? 2 * (IH) : 0) + ii][ks] = *reinterpret_cast<const s16x8_t*>(lds3 + aaddr[ks] + (2 * (IH) + ii) * 4096); \
    }
#define PP_RD_B(J, XOR) PP_RD_B_T(J, J, XOR, it)
#define PP_RD_B_T(BR, JR, XOR, T_)                                                                                        \
    if (rd_on || (T_) == 0) {                                                                                             \
        _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                                                  \
            fb[BR][ks] = *reinterpret_cast<const s16x8_t*>(lds3 + (baddr[ks] ^ (XOR)) + (JR) * 4096);                     \
    }
#define PP_BAR() if (!(ABL & 4)) __builtin_amdgcn_s_barrier();
#define PP_STAMP(I_) if (TRACE && trace_on) { stamp[I_] = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); }
#define PP_SYNC_IN() PP_SYNC_IN_P(0)
#define PP_SYNC_IN_P(PH_)                                                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                                                    \
    if (TRACE & 2) PP_STAMP(4 * (PH_) + 1)                                                                             \
    if (PREWAIT) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                       \
    PP_BAR()                                                                                                              \
    if (!PREWAIT) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                                    \
    PP_STAMP(4 * (PH_) + 2)                                                                                            \
    if (PRIO) __builtin_amdgcn_s_setprio(1);
#define PP_SYNC_OUT() PP_SYNC_OUT_P(0)
#define PP_SYNC_OUT_P(PH_)                                                                                                \
    if (PRIO) __builtin_amdgcn_s_setprio(0);                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                                                    \
    PP_STAMP(4 * (PH_) + 3)                                                                                            \
    PP_BAR()                                                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                                                    \
    PP_STAMP(4 * (PH_) + 4)
#define PP_MFMA_Q(IH, J) PP_MFMA_QB(IH, J, J)
#define PP_MFMA_QB(IH, J, BR) PP_MFMA_QBP(IH, J, BR, 0)
#define PP_MFMA_QBP(IH, J, BR, PH_)                                                                                       \
    PP_SYNC_IN_P(PH_)                                                                                                          \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                                                      \
        _Pragma("unroll") for (int ii = 0; ii < 2; ++ii)                                                                  \
            acc[2 * (IH) + ii][J] = mfma32t<true>(fb[BR][ks], fa[(SCHED == 3 ? 2 * (IH) : 0) + ii][ks], acc[2 * (IH) + ii][J]); \
    PP_SYNC_OUT_P(PH_)
#define PP_MFMA_H(IH)                                                                                                     \
    PP_SYNC_IN()                                                                                                          \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                                                      \
        _Pragma("unroll") for (int ii = 0; ii < 2; ++ii)                                                                  \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                 \
                acc[2 * (IH) + ii][j] = mfma32t<true>(fb[j][ks], fa[(SCHED == 3 ? 2 * (IH) : 0) + ii][ks], acc[2 * (IH) + ii][j]); \
    PP_SYNC_OUT()
#define PP_FLIP() _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) { aaddr[ks] ^= 0x8000; baddr[ks] ^= 0x8000; }

    if (SCHED == 0) {
        PP_DMA_ALWAYS(0, 0) PP_DMA_ALWAYS(1, 0) PP_DMA_ALWAYS(2, 0) PP_DMA_ALWAYS(3, 0)
        if (nk > 1 && !(ABL & 1)) { PP_DMA(0, 1) PP_DMA(1, 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
        else { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        __builtin_amdgcn_s_barrier();
        if (wm == 1) PP_BAR()
        for (int it = 0; it < nk; ++it) {
            if (it + 1 < nk) PP_DMA(2, it + 1)
            PP_RD_A(0) PP_RD_B(0, 0)
            PP_MFMA_Q(0, 0)
            if (it + 1 < nk) PP_DMA(3, it + 1)
            PP_RD_B(1, 0)
            PP_MFMA_Q(0, 1)
            if (it + 2 < nk) PP_DMA(0, it + 2)
            PP_RD_A(1)
            PP_MFMA_Q(1, 1)
            if (it + 2 < nk && !(ABL & 1)) { PP_DMA(1, it + 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
            else { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
            PP_MFMA_Q(1, 0)
            if (!(ABL & 1)) PP_FLIP()
        }
        if (wm == 0) PP_BAR()
    } else if (SCHED == 1) {
        // balanced reads 8/4/8/4: the next tile's B0 fragments are read in P4 into the registers B1 vacated after P3, so the
        // two B register sets swap roles every tile (two tiles per loop trip; nk even).  Slots of tile t+1: B0 at P2(t-1), A0 at
        // P3(t-1), B1 at P4(t-1), A1 at P1(t); retired by vmcnt(4) at P3(t); first read (B0) at P4(t).
#define PP_TILE1(T_, X, Y)                                                                                                \
        if ((T_) + 1 < nk) PP_DMA(3, (T_) + 1)                                                                            \
        PP_RD_A_T(0, T_)                                                                                                  \
        PP_MFMA_QBP(0, 0, X, 0)                                                                                               \
        if ((T_) + 2 < nk) PP_DMA(1, (T_) + 2)                                                                            \
        PP_RD_B_T(Y, 1, 0, T_)                                                                                            \
        PP_MFMA_QBP(0, 1, Y, 1)                                                                                               \
        if ((T_) + 2 < nk && !(ABL & 1)) { PP_DMA(0, (T_) + 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }         \
        else { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }                                                         \
        PP_RD_A_T(1, T_)                                                                                                  \
        PP_MFMA_QBP(1, 1, Y, 2)                                                                                               \
        if ((T_) + 2 < nk) PP_DMA(2, (T_) + 2)                                                                            \
        if ((T_) + 1 < nk) { PP_RD_B_T(Y, 0, (ABL & 1) ? 0 : 0x8000, (rd_on ? 0 : 1)) }                                   \
        PP_MFMA_QBP(1, 0, X, 3)                                                                                               \
        if (!(ABL & 1)) PP_FLIP()
        PP_DMA_ALWAYS(1, 0) PP_DMA_ALWAYS(0, 0) PP_DMA_ALWAYS(2, 0) PP_DMA_ALWAYS(3, 0)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (nk > 1) { PP_DMA(1, 1) PP_DMA(0, 1) PP_DMA(2, 1) }
        if (wm == 1) PP_BAR()
        PP_RD_B_T(0, 0, 0, 0)
        for (int it = 0; it < nk; it += 2) {
            trace_on = TRACE && (it == 4);
            PP_STAMP(0)
            PP_TILE1(it, 0, 1)
            trace_on = false;
            PP_TILE1(it + 1, 1, 0)
        }
        if (wm == 0) PP_BAR()
    } else if (SCHED == 2) {
        // tile t+1: B slots (1, 2) issued at H2(t-1), A slots (0, 3) at H1(t); retired at H2(t) (4 younger pieces = B of t+2)
        PP_DMA_ALWAYS(0, 0) PP_DMA_ALWAYS(1, 0) PP_DMA_ALWAYS(2, 0) PP_DMA_ALWAYS(3, 0)
        if (nk > 1 && !(ABL & 1)) { PP_DMA(1, 1) PP_DMA(2, 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
        else { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        __builtin_amdgcn_s_barrier();
        if (wm == 1) PP_BAR()
        for (int it = 0; it < nk; ++it) {
            if (it + 1 < nk) { PP_DMA(0, it + 1) PP_DMA(3, it + 1) }
            PP_RD_A(0) PP_RD_B(0, 0) PP_RD_B(1, 0)
            PP_MFMA_H(0)
            if (it + 2 < nk && !(ABL & 1)) { PP_DMA(1, it + 2) PP_DMA(2, it + 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
            else { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
            PP_RD_A(1)
            PP_MFMA_H(1)
            if (!(ABL & 1)) PP_FLIP()
        }
        if (wm == 0) PP_BAR()
    } else {
        // SCHED 3: all fragments of a K tile are read in one section (24 reads, 32 MFMAs per section).  Tile it+1 is issued at the
        // top of tile it's load section (its stage was last read one section earlier by both wave rows: PREWAIT) and retired at
        // the end of the MFMA section, ahead of the barrier that precedes its first read.
        PP_DMA_ALWAYS(0, 0) PP_DMA_ALWAYS(1, 0) PP_DMA_ALWAYS(2, 0) PP_DMA_ALWAYS(3, 0)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (wm == 1) PP_BAR()
        for (int it = 0; it < nk; ++it) {
            if (it + 1 < nk) { PP_DMA(0, it + 1) PP_DMA(1, it + 1) PP_DMA(2, it + 1) PP_DMA(3, it + 1) }
            PP_RD_A(0) PP_RD_A(1) PP_RD_B(0, 0) PP_RD_B(1, 0)
            PP_SYNC_IN()
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = mfma32t<true>(fb[j][ks], fa[i][ks], acc[i][j]);
            if (PRIO) __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            PP_BAR()
            __builtin_amdgcn_sched_barrier(0);
            if (!(ABL & 1)) PP_FLIP()
        }
        if (wm == 0) PP_BAR()
    }
    if (blockIdx.x == 17 && tid == 0) { g_clk[0] = __builtin_readcyclecounter() - clk0; g_clk[1] = wall_clock64() - rt0; }
    if (TRACE && g_trace != nullptr && lane == 0 && blockIdx.x < 512) {
#pragma unroll
        for (int i = 0; i < 17; ++i) g_trace[((size_t)blockIdx.x * 8 + wave) * 17 + i] = stamp[i];
    }
    // plain epilogue: accumulator block (i, j) holds C^T -- lane = row, register quad q = columns 8 q + 4 lg .. + 3
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * 128 + i * 32 + lr;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = n0 + wn * 64 + j * 32 + 8 * q + 4 * lg;
                uint2 pk;
                pk.x = pack2<true>(acc[i][j][4 * q], acc[i][j][4 * q + 1]);
                pk.y = pack2<true>(acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
                *reinterpret_cast<uint2*>(C + (size_t)m * N + n) = pk;
            }
    }
}


// ---- the same ping-pong loop (SCHED 1: balanced quadrant phases) on v_mfma_f32_16x16x32_f16 -------------------------------------
// wave tile 128 x 64 = 8 x 4 blocks of 16 x 16 (f32x4 accumulators, C^T: lane & 15 = row, register r = column 4 (lane >> 4) + r);
// a K tile is two 32-deep k-steps; fragment (16 rows, 32 k): lane reads row (lane & 15), 16-byte chunk 4 ks + (lane >> 4).
typedef __attribute__((ext_vector_type(4))) float f32x4v;
template <int ABL, bool PRIO>
__global__ __launch_bounds__(512) void pp16_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, bf16_t* __restrict__ C,
                                                   int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds3[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int ntn = N / T256, ntm = (M + T256 - 1) / T256, nwg = ntm * ntn;
    const int t = xcd_remap(blockIdx.x, nwg);
    const int group_size = 4 * ntn, gid = t / group_size, first_m = gid * 4;
    const int gm = (ntm - first_m) < 4 ? (ntm - first_m) : 4;
    const int tin = t - gid * group_size;
    const int m0 = (first_m + tin % gm) * T256, n0 = (tin / gm) * T256;
    const int nk = K / BK;
    const unsigned long long clk0 = __builtin_readcyclecounter(), rt0 = wall_clock64();
    const int rows_a = (M - m0) < T256 ? (M - m0) : T256;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)(A + (size_t)m0 * K), 0, rows_a * K * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)(B + (size_t)n0 * K), 0, T256 * K * 2, 0x00020000);
    const int prow = lane >> 3, pch = lane & 7;
    int vo[4][2], ld_[4][2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int p = 2 * wave + e;
        const int ra0 = (p >> 3) * 128 + (p & 7) * 8, ra1 = ra0 + 64;
        const int rb0 = (p >> 2) * 64 + (p & 3) * 8, rb1 = rb0 + 32;
        const int rows[4] = {ra0, rb0, rb1, ra1};
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) {
            const int row = rows[sl] + prow;
            const int cl = pch ^ ((row >> 1) & 7);
            const bool is_b = (sl == 1 || sl == 2);
            vo[sl][e] = row * K * 2 + cl * 16;
            ld_[sl][e] = (is_b ? 65536 : 0) + rows[sl] * 128;
        }
    }
#define Q_DMA(SL, KT)                                                                                                     \
    if (!(ABL & 1)) {                                                                                                     \
        const int so_ = (KT) * (BK * 2), st_ = ((KT) & 1) << 15;                                                          \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(((SL) == 1 || (SL) == 2) ? rb : ra, (lds_ptr_t)(lds3 + st_ + ld_[SL][0]), 16, vo[SL][0], so_, 0, 0); \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(((SL) == 1 || (SL) == 2) ? rb : ra, (lds_ptr_t)(lds3 + st_ + ld_[SL][1]), 16, vo[SL][1], so_, 0, 0); \
    }
#define Q_DMA_ALWAYS(SL, KT)                                                                                              \
    {                                                                                                                     \
        const int so_ = (KT) * (BK * 2), st_ = ((KT) & 1) << 15;                                                          \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(((SL) == 1 || (SL) == 2) ? rb : ra, (lds_ptr_t)(lds3 + st_ + ld_[SL][0]), 16, vo[SL][0], so_, 0, 0); \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(((SL) == 1 || (SL) == 2) ? rb : ra, (lds_ptr_t)(lds3 + st_ + ld_[SL][1]), 16, vo[SL][1], so_, 0, 0); \
    }
    f32x4v acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};
    const int l15 = lane & 15, lq = lane >> 4, sw = (l15 >> 1) & 7;
    int aaddr[2], baddr[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int c = ((4 * ks + lq) ^ sw) << 4;
        aaddr[ks] = wm * 16384 + l15 * 128 + c;
        baddr[ks] = 65536 + wn * 8192 + l15 * 128 + c;
    }
    // rows of fragment block i: 16 i + l15 -> swizzle ((16 i + l15) >> 1) & 7 = sw ^ ... no: 16 i adds 8 to (row >> 1): & 7 unchanged
    f16x8_t fa[4][2], fb[2][2][2];   // fa[ii][ks]: 4 row blocks of the current half; fb[set][jj][ks]: 2 column blocks
    const bool rd_on = !(ABL & 2);
#define Q_RD_A(IH, T_)                                                                                                    \
    if (rd_on || (T_) == 0) {                                                                                             \
        _Pragma("unroll") for (int ii = 0; ii < 4; ++ii)                                                                  \
            _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                              \
                fa[ii][ks] = *reinterpret_cast<const f16x8_t*>(lds3 + aaddr[ks] + (4 * (IH) + ii) * 2048);                \
    }
#define Q_RD_B(SET, JH, XOR, T_)                                                                                          \
    if (rd_on || (T_) == 0) {                                                                                             \
        _Pragma("unroll") for (int jj = 0; jj < 2; ++jj)                                                                  \
            _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                              \
                fb[SET][jj][ks] = *reinterpret_cast<const f16x8_t*>(lds3 + (baddr[ks] ^ (XOR)) + (2 * (JH) + jj) * 2048); \
    }
#define Q_BAR() if (!(ABL & 4)) __builtin_amdgcn_s_barrier();
#define Q_MFMA(IH, JH, SET)                                                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                                                    \
    Q_BAR()                                                                                                               \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                                    \
    if (PRIO) __builtin_amdgcn_s_setprio(1);                                                                              \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                                      \
        _Pragma("unroll") for (int ii = 0; ii < 4; ++ii)                                                                  \
            _Pragma("unroll") for (int jj = 0; jj < 2; ++jj)                                                              \
                acc[4 * (IH) + ii][2 * (JH) + jj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[SET][jj][ks], fa[ii][ks], acc[4 * (IH) + ii][2 * (JH) + jj], 0, 0, 0); \
    if (PRIO) __builtin_amdgcn_s_setprio(0);                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                                                    \
    Q_BAR()                                                                                                               \
    __builtin_amdgcn_sched_barrier(0);
#define Q_FLIP() _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) { aaddr[ks] ^= 0x8000; baddr[ks] ^= 0x8000; }
#define Q_TILE(T_, X, Y)                                                                                                  \
    if ((T_) + 1 < nk) Q_DMA(3, (T_) + 1)                                                                                 \
    Q_RD_A(0, T_)                                                                                                         \
    Q_MFMA(0, 0, X)                                                                                                       \
    if ((T_) + 2 < nk) Q_DMA(1, (T_) + 2)                                                                                 \
    Q_RD_B(Y, 1, 0, T_)                                                                                                   \
    Q_MFMA(0, 1, Y)                                                                                                       \
    if ((T_) + 2 < nk && !(ABL & 1)) { Q_DMA(0, (T_) + 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }              \
    else { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }                                                             \
    Q_RD_A(1, T_)                                                                                                         \
    Q_MFMA(1, 1, Y)                                                                                                       \
    if ((T_) + 2 < nk) Q_DMA(2, (T_) + 2)                                                                                 \
    if ((T_) + 1 < nk) { Q_RD_B(Y, 0, (ABL & 1) ? 0 : 0x8000, (rd_on ? 0 : 1)) }                                          \
    Q_MFMA(1, 0, X)                                                                                                       \
    if (!(ABL & 1)) Q_FLIP()
    Q_DMA_ALWAYS(1, 0) Q_DMA_ALWAYS(0, 0) Q_DMA_ALWAYS(2, 0) Q_DMA_ALWAYS(3, 0)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (nk > 1) { Q_DMA(1, 1) Q_DMA(0, 1) Q_DMA(2, 1) }
    if (wm == 1) Q_BAR()
    Q_RD_B(0, 0, 0, 0)
    for (int it = 0; it < nk; it += 2) {
        Q_TILE(it, 0, 1)
        Q_TILE(it + 1, 1, 0)
    }
    if (wm == 0) Q_BAR()
    if (blockIdx.x == 17 && tid == 0) { g_clk[0] = __builtin_readcyclecounter() - clk0; g_clk[1] = wall_clock64() - rt0; }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int m = m0 + wm * 128 + i * 16 + l15;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 64 + j * 16 + 4 * lq;
            uint2 pk;
            pk.x = pack2<true>(acc[i][j][0], acc[i][j][1]);
            pk.y = pack2<true>(acc[i][j][2], acc[i][j][3]);
            *reinterpret_cast<uint2*>(C + (size_t)m * N + n) = pk;
        }
    }
}


// ---- "dual": two independent 4-wave workgroups per CU, 128 x 256 tile each (wave = 128 x 64), BK = 32, three 24-KiB stages ------
// Purpose: let one workgroup's epilogue overlap the other's K loop (the 8-wave kernel leaves the matrix pipe idle for the whole
// epilogue: 9-30 us per tile at K = 768).  Single-stream software pipeline per wave: DMA of step s+2, fragment reads of step s+1,
// MFMAs of step s; one barrier per 32-deep step.  64-byte LDS rows, chunk c of row r stored at c ^ ((3 * (r >> 2)) & 3).
#define D_STAGE 24576
#define D_LDS (3 * D_STAGE)
template <int EPI_SPIN>
__global__ __launch_bounds__(256, 2) void dual_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, bf16_t* __restrict__ C,
                                                      int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds3[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntn = N / 256, ntm = (M + 127) / 128, nwg = ntm * ntn;
    const int t = xcd_remap(blockIdx.x, nwg);
    const int group_size = 8 * ntn, gid = t / group_size, first_m = gid * 8;
    const int gm = (ntm - first_m) < 8 ? (ntm - first_m) : 8;
    const int tin = t - gid * group_size;
    const int m0 = (first_m + tin % gm) * 128, n0 = (tin / gm) * 256;
    const int ns = K / 32;
    const unsigned long long clk0 = __builtin_readcyclecounter(), rt0 = wall_clock64();
    const int rows_a = (M - m0) < 128 ? (M - m0) : 128;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)(A + (size_t)m0 * K), 0, rows_a * K * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)(B + (size_t)n0 * K), 0, 256 * K * 2, 0x00020000);
    // DMA pieces (16 rows x 64 B): wave w fetches A pieces 2w, 2w+1 and B pieces 4w .. 4w+3 (its own 64 B rows)
    const int prow = lane >> 2, pc = lane & 3;
    const int lchunk = pc ^ ((3 * (prow >> 2)) & 3);
    int voa[2], vob[4];
#pragma unroll
    for (int e = 0; e < 2; ++e) voa[e] = ((2 * wn + e) * 16 + prow) * K * 2 + lchunk * 16;
#pragma unroll
    for (int e = 0; e < 4; ++e) vob[e] = ((4 * wn + e) * 16 + prow) * K * 2 + lchunk * 16;
#define D_DMA(S_, STG_)                                                                                                   \
    {                                                                                                                     \
        const int so_ = (S_) * 64, sb_ = (STG_) * D_STAGE;                                                                \
        _Pragma("unroll") for (int e = 0; e < 2; ++e)                                                                     \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr_t)(lds3 + sb_ + (2 * wn + e) * 1024), 16, voa[e], so_, 0, 0); \
        _Pragma("unroll") for (int e = 0; e < 4; ++e)                                                                     \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_ptr_t)(lds3 + sb_ + 8192 + (4 * wn + e) * 1024), 16, vob[e], so_, 0, 0); \
    }
    f32x4v acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};
    const int l15 = lane & 15, lq = lane >> 4;
    const int cswz = (lq ^ ((3 * (l15 >> 2)) & 3)) << 4;
    const unsigned lbase = (unsigned)(size_t)lds3;
    const unsigned abase = lbase + l15 * 64 + cswz, bbase = lbase + 8192 + (wn * 64 + l15) * 64 + cswz;
    f16x8_t fa[2][4], fb[2][4];   // fa[h]: the 4 row blocks of A half h of the current step; fb[set]: the 4 column blocks of a step
#define D_RD1(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF) : "memory")
#define D_READ_A(H, STG_)                                                                                                 \
    {                                                                                                                     \
        const unsigned aa_ = abase + (STG_) * D_STAGE;                                                                    \
        D_RD1(fa[H][0], aa_, (H) * 4096); D_RD1(fa[H][1], aa_, (H) * 4096 + 1024); D_RD1(fa[H][2], aa_, (H) * 4096 + 2048); D_RD1(fa[H][3], aa_, (H) * 4096 + 3072); \
    }
#define D_READ_B(SET, STG_)                                                                                               \
    {                                                                                                                     \
        const unsigned bb_ = bbase + (STG_) * D_STAGE;                                                                    \
        D_RD1(fb[SET][0], bb_, 0); D_RD1(fb[SET][1], bb_, 1024); D_RD1(fb[SET][2], bb_, 2048); D_RD1(fb[SET][3], bb_, 3072); \
    }
    // counted LDS wait; ties the fragment registers it covers to the wait so that no consumer is scheduled above it
#define D_WAIT_AB(CNT, H, SET)                                                                                            \
    asm volatile("s_waitcnt lgkmcnt(" #CNT ")"                                                                            \
                 : "+v"(fa[H][0]), "+v"(fa[H][1]), "+v"(fa[H][2]), "+v"(fa[H][3]), "+v"(fb[SET][0]), "+v"(fb[SET][1]), "+v"(fb[SET][2]), "+v"(fb[SET][3]) \
                 :: "memory");
#define D_WAIT_A(CNT, H)                                                                                                  \
    asm volatile("s_waitcnt lgkmcnt(" #CNT ")" : "+v"(fa[H][0]), "+v"(fa[H][1]), "+v"(fa[H][2]), "+v"(fa[H][3]) :: "memory");
#define D_MFMA(H, SET)                                                                                                    \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                         \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                                     \
            acc[4 * (H) + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[SET][j], fa[H][i], acc[4 * (H) + i][j], 0, 0, 0);
    // one 32-deep step S_ (B fragments in set X; the next step's go to set Y); stage indices st0 (this step), st1, st2 in SGPRs
#define D_STEP(S_, X, Y)                                                                                                  \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                                      \
    __builtin_amdgcn_s_barrier();                                                                                         \
    if ((S_) + 2 < ns) D_DMA((S_) + 2, st2)                                                                               \
    D_READ_A(1, st0)                                                                                                      \
    D_WAIT_AB(4, 0, X)                                                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                                    \
    D_MFMA(0, X)                                                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                                                    \
    if ((S_) + 1 < ns) { D_READ_B(Y, st1) D_READ_A(0, st1) D_WAIT_A(8, 1) } else { D_WAIT_A(0, 1) }                       \
    __builtin_amdgcn_sched_barrier(0);                                                                                    \
    D_MFMA(1, X)                                                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                                                    \
    { const int t_ = st0; st0 = st1; st1 = st2; st2 = t_; }
    int st0 = 0, st1 = 1, st2 = 2;
    D_DMA(0, 0)
    if (ns > 1) { D_DMA(1, 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); } else { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    __builtin_amdgcn_s_barrier();
    D_READ_B(0, 0) D_READ_A(0, 0)
    for (int s_ = 0; s_ < ns; s_ += 2) {
        D_STEP(s_, 0, 1)
        D_STEP(s_ + 1, 1, 0)
    }
    if (blockIdx.x == 17 && tid == 0) { g_clk[0] = __builtin_readcyclecounter() - clk0; g_clk[1] = wall_clock64() - rt0; }
    if (EPI_SPIN > 0) {   // stand-in for a long epilogue: EPI_SPIN x 1000 cycles of VALU-free waiting
        const unsigned long long e0 = __builtin_readcyclecounter();
        while (__builtin_readcyclecounter() - e0 < (unsigned long long)EPI_SPIN * 1000) __builtin_amdgcn_s_sleep(8);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int m = m0 + i * 16 + l15;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 64 + j * 16 + 4 * lq;
            uint2 pk;
            pk.x = pack2<true>(acc[i][j][0], acc[i][j][1]);
            pk.y = pack2<true>(acc[i][j][2], acc[i][j][3]);
            *reinterpret_cast<uint2*>(C + (size_t)m * N + n) = pk;
        }
    }
}

__global__ void fill_kernel(bf16_t* p, size_t n, unsigned seed, float scale) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)(i * 2654435761u) ^ seed;
        h ^= h >> 15; h *= 0x2c1b3c6du; h ^= h >> 12; h *= 0x297a2d39u; h ^= h >> 15;
        p[i] = f2h(((h & 0xffff) / 32768.0f - 1.0f) * scale);
    }
}
__global__ void ref_kernel(const bf16_t* A, const bf16_t* B, const int* mm, const int* nn, float* out, int K, int ns) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= ns) return;
    const bf16_t* a = A + (size_t)mm[s] * K;
    const bf16_t* b = B + (size_t)nn[s] * K;
    float acc = 0.f;
    for (int k = 0; k < K; ++k) acc += h2f(a[k]) * h2f(b[k]);
    out[s] = acc;
}
__global__ void gather_kernel(const bf16_t* C, const int* mm, const int* nn, float* out, int N, int ns) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < ns) out[s] = h2f(C[(size_t)mm[s] * N + nn[s]]);
}


// ---- "quad": ONE 4-wave workgroup per CU, wave tile 128 x 128 (8 x 8 blocks of 16 x 16: 256 accumulator registers), one wave per SIMD
// with the 512-register budget.  Per 64-deep K tile a wave reads 32 fragments (16 per 32-deep k-step) for 128 MFMAs -- two thirds of
// the LDS fragment traffic of the 8-wave kernel per FLOP.  Single-stream software pipeline: fragments of step s + 1 are read while the
// MFMAs of step s issue; tile t + 1 is DMA'd while tile t is consumed; two workgroup barriers per K tile.
template <int ABL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void quad_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B,
                                                                                          bf16_t* __restrict__ C, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds3[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int ntn = N / T256, ntm = (M + T256 - 1) / T256, nwg = ntm * ntn;
    const int t = xcd_remap(blockIdx.x, nwg);
    const int group_size = 4 * ntn, gid = t / group_size, first_m = gid * 4;
    const int gm = (ntm - first_m) < 4 ? (ntm - first_m) : 4;
    const int tin = t - gid * group_size;
    const int m0 = (first_m + tin % gm) * T256, n0 = (tin / gm) * T256;
    const int nk = K / BK;
    const unsigned long long clk0 = __builtin_readcyclecounter(), rt0 = wall_clock64();
    const int rows_a = (M - m0) < T256 ? (M - m0) : T256;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)(A + (size_t)m0 * K), 0, rows_a * K * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)(B + (size_t)n0 * K), 0, T256 * K * 2, 0x00020000);
    // DMA: 32 A pieces + 32 B pieces of 8 rows per K tile; wave w fetches A pieces 8 w .. 8 w + 7 and B pieces 8 w .. 8 w + 7
    const int prow = lane >> 3, pch = lane & 7;
    int voa[8], vob[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int row = (8 * wave + e) * 8 + prow;
        const int cl = pch ^ ((row >> 1) & 7);
        voa[e] = row * K * 2 + cl * 16;
        vob[e] = row * K * 2 + cl * 16;
    }
#define QD_DMA(KT)                                                                                                        \
    {                                                                                                                     \
        const int so_ = (KT) * (BK * 2), st_ = ((KT) & 1) << 15;                                                          \
        _Pragma("unroll") for (int e = 0; e < 8; ++e) {                                                                   \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr_t)(lds3 + st_ + (8 * wave + e) * 1024), 16, voa[e], so_, 0, 0); \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_ptr_t)(lds3 + 65536 + st_ + (8 * wave + e) * 1024), 16, vob[e], so_, 0, 0); \
        }                                                                                                                 \
    }
    f32x4v acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};
    const int l15 = lane & 15, lq = lane >> 4, sw = (l15 >> 1) & 7;
    // fragment (block b of 16 rows, k-step ks): row 16 b + l15, 16-byte chunk (4 ks + lq) ^ sw
    const unsigned lbase = (unsigned)(size_t)lds3;
    unsigned aaddr[2], baddr[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int c = ((4 * ks + lq) ^ sw) << 4;
        aaddr[ks] = lbase + wm * 16384 + l15 * 128 + c;
        baddr[ks] = lbase + 65536 + wn * 16384 + l15 * 128 + c;
    }
    f16x8_t fa[2][8], fb[2][8];      // [register set][block]
    // LDS reads as asm (the compiler's own wait insertion would serialise the sets at the loop back edge); explicit waits + ties
#define QD_RD1(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF) : "memory")
#define QD_READ(SET, KS, STG)                                                                                             \
    {                                                                                                                     \
        QD_RD1(fa[SET][0], aaddr[KS], ((STG) << 15) + 0 * 2048); QD_RD1(fb[SET][0], baddr[KS], ((STG) << 15) + 0 * 2048); \
        QD_RD1(fa[SET][1], aaddr[KS], ((STG) << 15) + 1 * 2048); QD_RD1(fb[SET][1], baddr[KS], ((STG) << 15) + 1 * 2048); \
        QD_RD1(fa[SET][2], aaddr[KS], ((STG) << 15) + 2 * 2048); QD_RD1(fb[SET][2], baddr[KS], ((STG) << 15) + 2 * 2048); \
        QD_RD1(fa[SET][3], aaddr[KS], ((STG) << 15) + 3 * 2048); QD_RD1(fb[SET][3], baddr[KS], ((STG) << 15) + 3 * 2048); \
        QD_RD1(fa[SET][4], aaddr[KS], ((STG) << 15) + 4 * 2048); QD_RD1(fb[SET][4], baddr[KS], ((STG) << 15) + 4 * 2048); \
        QD_RD1(fa[SET][5], aaddr[KS], ((STG) << 15) + 5 * 2048); QD_RD1(fb[SET][5], baddr[KS], ((STG) << 15) + 5 * 2048); \
        QD_RD1(fa[SET][6], aaddr[KS], ((STG) << 15) + 6 * 2048); QD_RD1(fb[SET][6], baddr[KS], ((STG) << 15) + 6 * 2048); \
        QD_RD1(fa[SET][7], aaddr[KS], ((STG) << 15) + 7 * 2048); QD_RD1(fb[SET][7], baddr[KS], ((STG) << 15) + 7 * 2048); \
    }
#define QD_TIE(SET)                                                                                                       \
    asm volatile("" : "+v"(fa[SET][0]), "+v"(fa[SET][1]), "+v"(fa[SET][2]), "+v"(fa[SET][3]), "+v"(fa[SET][4]), "+v"(fa[SET][5]), \
                      "+v"(fa[SET][6]), "+v"(fa[SET][7]), "+v"(fb[SET][0]), "+v"(fb[SET][1]), "+v"(fb[SET][2]), "+v"(fb[SET][3]), \
                      "+v"(fb[SET][4]), "+v"(fb[SET][5]), "+v"(fb[SET][6]), "+v"(fb[SET][7]) :: "memory");
#define QD_MFMA(SET)                                                                                                      \
    _Pragma("unroll") for (int i = 0; i < 8; ++i)                                                                         \
        _Pragma("unroll") for (int j = 0; j < 8; ++j)                                                                     \
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[SET][j], fa[SET][i], acc[i][j], 0, 0, 0);
    // one K tile whose operands sit in stage STG (compile-time): set 0 holds its k-step 0 on entry.  Memory instructions are sprinkled
    // between the rows of MFMAs (8 per row, 128 matrix-pipe cycles): a lone wave per SIMD issues in order, so a block of 16 reads /
    // 16 DMA instructions in front of a batch would idle the pipe for their whole issue time.  Reads end two rows before the batch
    // that consumes them.  Tiles past the end are fetched out of the buffer's range (zeros) and never consumed: no branches.
#define QD_TILE(T_, STG) \
    { \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); QD_TIE(0) __builtin_amdgcn_sched_barrier(0); \
    QD_RD1(fb[1][0], baddr[1], ((STG) << 15) + 0 * 2048); QD_RD1(fb[1][1], baddr[1], ((STG) << 15) + 1 * 2048); QD_RD1(fb[1][2], baddr[1], ((STG) << 15) + 2 * 2048); __builtin_amdgcn_sched_barrier(0); \
    _Pragma("unroll") for (int j = 0; j < 8; ++j) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[0][j]) : "v"(fb[0][j]), "v"(fa[0][0])); __builtin_amdgcn_sched_barrier(0); \
    QD_RD1(fb[1][3], baddr[1], ((STG) << 15) + 3 * 2048); QD_RD1(fb[1][4], baddr[1], ((STG) << 15) + 4 * 2048); QD_RD1(fb[1][5], baddr[1], ((STG) << 15) + 5 * 2048); __builtin_amdgcn_sched_barrier(0); \
    _Pragma("unroll") for (int j = 0; j < 8; ++j) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[1][j]) : "v"(fb[0][j]), "v"(fa[0][1])); __builtin_amdgcn_sched_barrier(0); \
    QD_RD1(fb[1][6], baddr[1], ((STG) << 15) + 6 * 2048); QD_RD1(fb[1][7], baddr[1], ((STG) << 15) + 7 * 2048); QD_RD1(fa[1][0], aaddr[1], ((STG) << 15) + 0 * 2048); __builtin_amdgcn_sched_barrier(0); \
    _Pragma("unroll") for (int j = 0; j < 8; ++j) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[2][j]) : "v"(fb[0][j]), "v"(fa[0][2])); __builtin_amdgcn_sched_barrier(0); \
    QD_RD1(fa[1][1], aaddr[1], ((STG) << 15) + 1 * 2048); QD_RD1(fa[1][2], aaddr[1], ((STG) << 15) + 2 * 2048); QD_RD1(fa[1][3], aaddr[1], ((STG) << 15) + 3 * 2048); __builtin_amdgcn_sched_barrier(0); \
    _Pragma("unroll") for (int j = 0; j < 8; ++j) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[3][j]) : "v"(fb[0][j]), "v"(fa[0][3])); __builtin_amdgcn_sched_barrier(0); \
    QD_RD1(fa[1][4], aaddr[1], ((STG) << 15) + 4 * 2048); QD_RD1(fa[1][5], aaddr[1], ((STG) << 15) + 5 * 2048); __builtin_amdgcn_sched_barrier(0); \
    _Pragma("unroll") for (int j = 0; j < 8; ++j) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[4][j]) : "v"(fb[0][j]), "v"(fa[0][4])); __builtin_amdgcn_sched_barrier(0); \
    QD_RD1(fa[1][6], aaddr[1], ((STG) << 15) + 6 * 2048); QD_RD1(fa[1][7], aaddr[1], ((STG) << 15) + 7 * 2048); __builtin_amdgcn_sched_barrier(0); \
    _Pragma("unroll") for (int j = 0; j < 8; ++j) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[5][j]) : "v"(fb[0][j]), "v"(fa[0][5])); __builtin_amdgcn_sched_barrier(0); \
    _Pragma("unroll") for (int j = 0; j < 8; ++j) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[6][j]) : "v"(fb[0][j]), "v"(fa[0][6])); __builtin_amdgcn_sched_barrier(0); \
    _Pragma("unroll") for (int j = 0; j < 8; ++j) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[7][j]) : "v"(fb[0][j]), "v"(fa[0][7])); __builtin_amdgcn_sched_barrier(0); \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); QD_TIE(1) __builtin_amdgcn_sched_barrier(0); \
    const int so2_ = ((T_) + 2) * (BK * 2), st2_ = (STG) << 15; \
    _Pragma("unroll") for (int j = 0; j < 8; ++j) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[0][j]) : "v"(fb[1][j]), "v"(fa[1][0])); __builtin_amdgcn_sched_barrier(0); \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); \
    QD_RD1(fb[0][0], baddr[0], (((STG) ^ 1) << 15) + 0 * 2048); QD_RD1(fb[0][1], baddr[0], (((STG) ^ 1) << 15) + 1 * 2048); QD_RD1(fb[0][2], baddr[0], (((STG) ^ 1) << 15) + 2 * 2048); __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr_t)(lds3 + st2_ + (8 * wave + 0) * 1024), 16, voa[0], so2_, 0, 0); __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr_t)(lds3 + st2_ + (8 * wave + 1) * 1024), 16, voa[1], so2_, 0, 0); __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr_t)(lds3 + st2_ + (8 * wave + 2) * 1024), 16, voa[2], so2_, 0, 0); __builtin_amdgcn_sched_barrier(0); \
    _Pragma("unroll") for (int j = 0; j < 8; ++j) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[1][j]) : "v"(fb[1][j]), "v"(fa[1][1])); __builtin_amdgcn_sched_barrier(0); \
    QD_RD1(fb[0][3], baddr[0], (((STG) ^ 1) << 15) + 3 * 2048); QD_RD1(fb[0][4], baddr[0], (((STG) ^ 1) << 15) + 4 * 2048); QD_RD1(fb[0][5], baddr[0], (((STG) ^ 1) << 15) + 5 * 2048); __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr_t)(lds3 + st2_ + (8 * wave + 3) * 1024), 16, voa[3], so2_, 0, 0); __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr_t)(lds3 + st2_ + (8 * wave + 4) * 1024), 16, voa[4], so2_, 0, 0); __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr_t)(lds3 + st2_ + (8 * wave + 5) * 1024), 16, voa[5], so2_, 0, 0); __builtin_amdgcn_sched_barrier(0); \
    _Pragma("unroll") for (int j = 0; j < 8; ++j) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[2][j]) : "v"(fb[1][j]), "v"(fa[1][2])); __builtin_amdgcn_sched_barrier(0); \
    QD_RD1(fb[0][6], baddr[0], (((STG) ^ 1) << 15) + 6 * 2048); QD_RD1(fb[0][7], baddr[0], (((STG) ^ 1) << 15) + 7 * 2048); QD_RD1(fa[0][0], aaddr[0], (((STG) ^ 1) << 15) + 0 * 2048); __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr_t)(lds3 + st2_ + (8 * wave + 6) * 1024), 16, voa[6], so2_, 0, 0); __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr_t)(lds3 + st2_ + (8 * wave + 7) * 1024), 16, voa[7], so2_, 0, 0); __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_ptr_t)(lds3 + 65536 + st2_ + (8 * wave + 0) * 1024), 16, vob[0], so2_, 0, 0); __builtin_amdgcn_sched_barrier(0); \
    _Pragma("unroll") for (int j = 0; j < 8; ++j) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[3][j]) : "v"(fb[1][j]), "v"(fa[1][3])); __builtin_amdgcn_sched_barrier(0); \
    QD_RD1(fa[0][1], aaddr[0], (((STG) ^ 1) << 15) + 1 * 2048); QD_RD1(fa[0][2], aaddr[0], (((STG) ^ 1) << 15) + 2 * 2048); QD_RD1(fa[0][3], aaddr[0], (((STG) ^ 1) << 15) + 3 * 2048); __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_ptr_t)(lds3 + 65536 + st2_ + (8 * wave + 1) * 1024), 16, vob[1], so2_, 0, 0); __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_ptr_t)(lds3 + 65536 + st2_ + (8 * wave + 2) * 1024), 16, vob[2], so2_, 0, 0); __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_ptr_t)(lds3 + 65536 + st2_ + (8 * wave + 3) * 1024), 16, vob[3], so2_, 0, 0); __builtin_amdgcn_sched_barrier(0); \
    _Pragma("unroll") for (int j = 0; j < 8; ++j) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[4][j]) : "v"(fb[1][j]), "v"(fa[1][4])); __builtin_amdgcn_sched_barrier(0); \
    QD_RD1(fa[0][4], aaddr[0], (((STG) ^ 1) << 15) + 4 * 2048); QD_RD1(fa[0][5], aaddr[0], (((STG) ^ 1) << 15) + 5 * 2048); __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_ptr_t)(lds3 + 65536 + st2_ + (8 * wave + 4) * 1024), 16, vob[4], so2_, 0, 0); __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_ptr_t)(lds3 + 65536 + st2_ + (8 * wave + 5) * 1024), 16, vob[5], so2_, 0, 0); __builtin_amdgcn_sched_barrier(0); \
    _Pragma("unroll") for (int j = 0; j < 8; ++j) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[5][j]) : "v"(fb[1][j]), "v"(fa[1][5])); __builtin_amdgcn_sched_barrier(0); \
    QD_RD1(fa[0][6], aaddr[0], (((STG) ^ 1) << 15) + 6 * 2048); QD_RD1(fa[0][7], aaddr[0], (((STG) ^ 1) << 15) + 7 * 2048); __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_ptr_t)(lds3 + 65536 + st2_ + (8 * wave + 6) * 1024), 16, vob[6], so2_, 0, 0); __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_ptr_t)(lds3 + 65536 + st2_ + (8 * wave + 7) * 1024), 16, vob[7], so2_, 0, 0); __builtin_amdgcn_sched_barrier(0); \
    _Pragma("unroll") for (int j = 0; j < 8; ++j) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[6][j]) : "v"(fb[1][j]), "v"(fa[1][6])); __builtin_amdgcn_sched_barrier(0); \
    _Pragma("unroll") for (int j = 0; j < 8; ++j) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[7][j]) : "v"(fb[1][j]), "v"(fa[1][7])); __builtin_amdgcn_sched_barrier(0); \
    }
    QD_DMA(0)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (nk > 1) QD_DMA(1)
    QD_READ(0, 0, 0)
    for (int it = 0; it < nk; it += 2) {       // (nk even)
        QD_TILE(it, 0)
        QD_TILE(it + 1, 1)
    }
    if (blockIdx.x == 17 && tid == 0) { g_clk[0] = __builtin_readcyclecounter() - clk0; g_clk[1] = wall_clock64() - rt0; }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int m = m0 + wm * 128 + i * 16 + l15;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int n = n0 + wn * 128 + j * 16 + 4 * lq;
            uint2 pk;
            pk.x = pack2<true>(acc[i][j][0], acc[i][j][1]);
            pk.y = pack2<true>(acc[i][j][2], acc[i][j][3]);
            *reinterpret_cast<uint2*>(C + (size_t)m * N + n) = pk;
        }
    }
}

typedef void (*kern_t)(const bf16_t*, const bf16_t*, bf16_t*, int, int, int);
struct Variant { const char* name; kern_t k; bool check; int threads = 512; int lds = LDS_BYTES; int tm = 256; };

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 10;
    std::vector<Variant> vs = {
        {"S0 quad 12/4/8/0 prio postwait", pp_kernel<0, 0, true, false>, true},
        {"S0 quad prio PREWAIT          ", pp_kernel<0, 0, true, true>, true},
        {"S0 quad noprio                ", pp_kernel<0, 0, false, false>, true},
        {"S1 quad 8/4/8/4 prio postwait ", pp_kernel<1, 0, true, false>, true},
        {"S1 quad 8/4/8/4 prio prewait  ", pp_kernel<1, 0, true, true>, true},
        {"S1 with mfma 16x16x32 prio    ", pp16_kernel<0, true>, true},
        {"S1 with mfma 16x16x32 noprio  ", pp16_kernel<0, false>, true},
        {"S1/16x16 ABL noDMA            ", pp16_kernel<1, true>, false},
        {"S1/16x16 ABL noDMA noRD       ", pp16_kernel<3, true>, false},
        {"QUAD 4 waves x 128x128, 512 reg", quad_kernel<0>, true, 256, LDS_BYTES, 256},
        {"DUAL 2x(4 waves,128x256,BK32) ", dual_kernel<0>, true, 256, D_LDS, 128},
        {"DUAL + 10k-cycle fake epilogue", dual_kernel<10>, true, 256, D_LDS, 128},
        {"S2 half 16/8 prio prewait     ", pp_kernel<2, 0, true, true>, true},
        {"S2 half 16/8 prio postwait    ", pp_kernel<2, 0, true, false>, true},
        {"S2 half noprio prewait        ", pp_kernel<2, 0, false, true>, true},
        {"S3 tile 24 prio prewait       ", pp_kernel<3, 0, true, true>, true},
        {"S3 tile noprio prewait        ", pp_kernel<3, 0, false, true>, true},
        {"S0 ABL noDMA                  ", pp_kernel<0, 1, true, false>, false},
        {"S0 ABL noDMA noRD             ", pp_kernel<0, 3, true, false>, false},
        {"S0 ABL noDMA noRD noBAR       ", pp_kernel<0, 7, true, false>, false},
        {"S0 ABL noRD                   ", pp_kernel<0, 2, true, false>, false},
        {"S2 ABL noDMA                  ", pp_kernel<2, 1, true, true>, false},
        {"S2 ABL noDMA noRD             ", pp_kernel<2, 3, true, true>, false},
        {"S2 ABL noDMA noRD noBAR       ", pp_kernel<2, 7, true, true>, false},
        {"S3 ABL noDMA                  ", pp_kernel<3, 1, true, true>, false},
        {"S3 ABL noDMA noRD             ", pp_kernel<3, 3, true, true>, false},
    };
    {
        unsigned long long* dtr;
        CHECK(hipMalloc(&dtr, 512 * 8 * 17 * 8));
        CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_trace), &dtr, sizeof(dtr)));
        const int M = 8192, N = 8192, K = 8192;
        bf16_t *A, *B, *C;
        CHECK(hipMalloc(&A, (size_t)M * K * 2)); CHECK(hipMalloc(&B, (size_t)N * K * 2)); CHECK(hipMalloc(&C, (size_t)M * N * 2));
        fill_kernel<<<2048, 256>>>(A, (size_t)M * K, 0x1234u, 1.0f);
        fill_kernel<<<2048, 256>>>(B, (size_t)N * K, 0x9876u, 0.05f);
        for (int mode = 0; mode < 2; ++mode) {
            kern_t k = mode == 0 ? (kern_t)pp_kernel<1, 0, true, false, 1> : (kern_t)pp_kernel<1, 0, true, false, 3>;
            CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
            CHECK(hipMemset(dtr, 0, 512 * 8 * 17 * 8));
            k<<<dim3(32 * 32), 512, LDS_BYTES>>>(A, B, C, M, N, K);
            CHECK(hipDeviceSynchronize());
            std::vector<unsigned long long> h(512 * 8 * 17);
            CHECK(hipMemcpy(h.data(), dtr, h.size() * 8, hipMemcpyDeviceToHost));
            printf("trace mode %d (S1, 8192^3, K tile 4): per phase [start->load issued->mfma start->mfma end->next start], cycles\n", mode);
            for (int blk : {0, 100, 300}) for (int w : {0, 4}) {
                const unsigned long long* st = &h[((size_t)blk * 8 + w) * 17];
                printf("  blk %3d wave %d:", blk, w);
                for (int ph = 0; ph < 4; ++ph) {
                    const long long s0 = st[4 * ph], s1 = st[4 * ph + 1], s2 = st[4 * ph + 2], s3 = st[4 * ph + 3];
                    const long long nx = (long long)st[4 * ph + 4];
                    printf("  P%d: ld %lld bar %lld mfma %lld out %lld |", ph + 1, s1 ? s1 - s0 : -1, s1 ? s2 - s1 : s2 - s0, s3 - s2, nx ? nx - s3 : -1);
                }
                printf("\n");
            }
        }
        CHECK(hipFree(A)); CHECK(hipFree(B)); CHECK(hipFree(C));
        dtr = nullptr;
        CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_trace), &dtr, sizeof(dtr)));
    }
    const int shapes[][3] = {{8192, 8192, 8192}, {38080, 3072, 768}, {38080, 768, 3072}, {211904, 768, 768}};
    for (auto& v : vs) CHECK(hipFuncSetAttribute((const void*)v.k, hipFuncAttributeMaxDynamicSharedMemorySize, v.lds));
    for (const auto& sh : shapes) {
        const int M = sh[0], N = sh[1], K = sh[2];
        bf16_t *A, *B, *C;
        CHECK(hipMalloc(&A, (size_t)M * K * 2)); CHECK(hipMalloc(&B, (size_t)N * K * 2)); CHECK(hipMalloc(&C, (size_t)M * N * 2));
        fill_kernel<<<2048, 256>>>(A, (size_t)M * K, 0x1234u, 1.0f);
        fill_kernel<<<2048, 256>>>(B, (size_t)N * K, 0x9876u, 0.05f);
        const int ns = 8192;
        std::vector<int> hm(ns), hn(ns);
        srand(7);
        for (int s = 0; s < ns; ++s) { hm[s] = (s < 512) ? (M - 1 - (s & 255)) : rand() % M; hn[s] = rand() % N; }
        int *dm, *dn; float *dref, *dout;
        CHECK(hipMalloc(&dm, ns * 4)); CHECK(hipMalloc(&dn, ns * 4)); CHECK(hipMalloc(&dref, ns * 4)); CHECK(hipMalloc(&dout, ns * 4));
        CHECK(hipMemcpy(dm, hm.data(), ns * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dn, hn.data(), ns * 4, hipMemcpyHostToDevice));
        ref_kernel<<<(ns + 255) / 256, 256>>>(A, B, dm, dn, dref, K, ns);
        std::vector<float> href(ns), hout(ns);
        CHECK(hipMemcpy(href.data(), dref, ns * 4, hipMemcpyDeviceToHost));
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        for (auto& v : vs) {
            CHECK(hipMemset(C, 0, (size_t)M * N * 2));
            const dim3 grid(((M + v.tm - 1) / v.tm) * (N / 256));
            v.k<<<grid, v.threads, v.lds>>>(A, B, C, M, N, K);
            CHECK(hipDeviceSynchronize());
            double maxerr = -1.0;
            if (v.check) {
                gather_kernel<<<(ns + 255) / 256, 256>>>(C, dm, dn, dout, N, ns);
                CHECK(hipMemcpy(hout.data(), dout, ns * 4, hipMemcpyDeviceToHost));
                maxerr = 0.0;
                for (int s = 0; s < ns; ++s) {
                    const double e = fabs((double)hout[s] - href[s]) / (1.0 + fabs((double)href[s]));
                    if (e > maxerr) maxerr = e;
                }
            }
            v.k<<<grid, v.threads, v.lds>>>(A, B, C, M, N, K);
            CHECK(hipEventRecord(e0));
            for (int r = 0; r < reps; ++r) v.k<<<grid, v.threads, v.lds>>>(A, B, C, M, N, K);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            ms /= reps;
            unsigned long long hc[4];
            CHECK(hipMemcpyFromSymbol(hc, HIP_SYMBOL(g_clk), sizeof(hc)));
            printf("M%-6d N%-5d K%-5d %s %8.3f ms %8.1f TF/s  err %s%.2e  clk %.2f GHz (blk 17: %llu cyc)\n", M, N, K, v.name, ms, 2.0 * M * N * K / ms / 1e9,
                   (v.check && maxerr > 2e-2) ? "FAIL " : "", maxerr, hc[1] ? (double)hc[0] / ((double)hc[1] * 10.0) : 0.0, hc[0]);
            fflush(stdout);
        }
        CHECK(hipFree(A)); CHECK(hipFree(B)); CHECK(hipFree(C)); CHECK(hipFree(dm)); CHECK(hipFree(dn)); CHECK(hipFree(dref)); CHECK(hipFree(dout));
    }
    return 0;
}
