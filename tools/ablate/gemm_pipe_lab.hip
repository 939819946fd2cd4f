// Laboratory (developer tool, not product): NT GEMM whose EPILOGUE OF TILE i RUNS INSIDE THE K LOOP OF TILE i + 1.
//
// The product's 256 x 256 kernel (csrc/gemm.hip gemm_nt_pp_kernel) holds 128 accumulator registers per wave and can start nothing
// while it converts and stores them: 45 % of a window-shape GEMM's time is epilogue that no MFMA overlaps.  Here the
// workgroup tile is 256 x 128 (wave tile 64 x 64 = 64 accumulator registers) and every wave owns TWO accumulator sets: while the K loop
// of tile i + 1 fills one, the other (tile i) is biased / activated / packed / stored in 16 small pieces, one per K tile, placed in the
// "load" segment of the wave -- the interval in which its SIMD partner (the other wave group) runs its MFMA cluster.
//   * 8 waves = 2 groups x (2 x 2) waves; group g owns rows 128 g .. 128 g + 127; the groups alternate LOAD and MFMA segments, one
//     s_barrier per segment (2 per K tile), exactly one group on the matrix pipe at any time.
//   * K tiles of 32 (one 16x16x32 MFMA step): stage = A 256 x 64 B + B 128 x 64 B = 24 KB, ring of 4 stages; operands by LDS-DMA, three
//     1-KB pieces per wave and K tile, issued three K tiles ahead, continuous across output tiles (the next tile's first stages are
//     requested during the last K tiles of the current one).
//   * 64-byte LDS rows, 16-byte chunk c of row r stored at slot c ^ ((r >> 2) & 3): the 16 lanes of a ds_read_b128 group hit 16 different
//     bank quads.
//   * B rows permuted on the DMA source side like the product (pp_brow_src) so that a lane's accumulators of column blocks (0, 1) and
//     (2, 3) are 8 consecutive output columns: 16-byte stores straight from the registers.
// Build / run (links the product library for the baseline):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffast-math -fno-finite-math-only tools/ablate/gemm_pipe_lab.hip -Ltransformer4sed_amd -lsed_hip \
//         -Wl,-rpath,$PWD/transformer4sed_amd -o tools/ablate/gemm_pipe_lab.bin && tools/ablate/gemm_pipe_lab.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../../transformer4sed_amd/csrc/common.h"
#include "../../include/sed_hip.h"
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef __attribute__((address_space(3))) void* lds_ptr_t;
#define LBK 32
#define LNS 4
#define LSTAGE (24 * 1024)
#define LB_OFF (16 * 1024)
#define LBIAS_OFF (LNS * LSTAGE)      // per-wave 64-float strips of column constants behind the operand ring

__device__ __forceinline__ int lab_brow_src(int row) { return (row & ~0x1C) | ((row & 0x10) >> 2) | ((row & 0x0C) << 1); }
__device__ __forceinline__ f32x4_t lab_mfma(s16x8_t a, s16x8_t b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}

struct LabArgs {
    const bf16_t* A; const bf16_t* B; const float* bias; bf16_t* C;
    int M, N, K, lda, ldb, ldc, group_m, gelu;
};

// EPI_MODE 0: epilogue pieces in the LOAD segments of the next tile's K loop (the point of this file)
//          1: the whole epilogue after the K loop (same kernel, nothing overlapped: the control)
// PLACE 0: operand DMA and epilogue piece in the LOAD segment; 1: both behind the MFMA cluster (the wave then waits for its partner's LOAD
//       anyway); 2: DMA behind the MFMA cluster, epilogue piece in the LOAD segment
template <int EPI_MODE, bool GELU, int PLACE>
__global__ __launch_bounds__(512) void gemm_pipe_kernel(const LabArgs g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wq = wave & 3, wm2 = wq >> 1, wn2 = wq & 1;
    const int ntn = g.N / 128, ntm = (g.M + 255) / 256, nwg = ntm * ntn, nk = g.K / LBK;
    const int l15 = lane & 15, lq = lane >> 4;
    auto tile_mn = [&](int t_, int& m0_, int& n0_) {
        const int t = xcd_remap(t_, nwg);
        const int GM = g.group_m, gs = GM * ntn, gid = t / gs, first_m = gid * GM;
        const int gm = (ntm - first_m) < GM ? (ntm - first_m) : GM;
        const int tin = t - gid * gs;
        m0_ = (first_m + tin % gm) * 256;
        n0_ = (tin / gm) * 128;
    };
    // ---- DMA geometry: wave's pieces {wave, wave + 8} = A rows 16 p .., {wave + 16} = B rows 16 (p - 16) ..; lane = (row r = lane >> 2, slot = lane & 3)
    const int pr = lane >> 2, psl = lane & 3, pch = psl ^ ((pr >> 2) & 3);
    int vo[3], ldo[3];
#pragma unroll
    for (int e = 0; e < 3; ++e) {
        const int p = wave + 8 * e;
        if (e < 2) { vo[e] = (16 * p + pr) * g.lda * 2 + pch * 16; ldo[e] = p * 1024; }
        else { vo[e] = lab_brow_src(16 * (p - 16) + pr) * g.ldb * 2 + pch * 16; ldo[e] = LB_OFF + (p - 16) * 1024; }
    }
    // ---- fragment addresses (within a stage)
    const int foff = l15 * 64 + ((lq ^ ((l15 >> 2) & 3)) << 4);
    const int fa_base = (grp * 128 + wm2 * 64) * 64 + foff, fb_base = LB_OFF + (wn2 * 64) * 64 + foff;

    __amdgpu_buffer_rsrc_t ra, rb, ran, rbn;          // current / next tile operand panels
    auto make_rsrc = [&](int m0_, int n0_, __amdgpu_buffer_rsrc_t& a_, __amdgpu_buffer_rsrc_t& b_) {
        const int rows = (g.M - m0_) < 256 ? (g.M - m0_) : 256;
        a_ = __builtin_amdgcn_make_buffer_rsrc((void*)(g.A + (size_t)m0_ * g.lda), 0, rows * g.lda * 2, 0x00020000);
        b_ = __builtin_amdgcn_make_buffer_rsrc((void*)(g.B + (size_t)n0_ * g.ldb), 0, 128 * g.ldb * 2, 0x00020000);
    };
#define LAB_DMA(RA_, RB_, KT_, STG_)                                                                                             \
    {                                                                                                                            \
        const int so_ = (KT_) * (LBK * 2), sb_ = (STG_) * LSTAGE;                                                                \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(RA_, (lds_ptr_t)(lds + sb_ + ldo[0]), 16, vo[0], so_, 0, 0);                    \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(RA_, (lds_ptr_t)(lds + sb_ + ldo[1]), 16, vo[1], so_, 0, 0);                    \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(RB_, (lds_ptr_t)(lds + sb_ + ldo[2]), 16, vo[2], so_, 0, 0);                    \
    }

    f32x4_t accA[4][4], accB[4][4];
    s16x8_t fa[4], fb[4];
    const unsigned bias_lds = (unsigned)(size_t)(lds_ptr_t)(lds + LBIAS_OFF + wave * 256 + 8 * lq * 4);
    const __amdgpu_buffer_rsrc_t rbias = __builtin_amdgcn_make_buffer_rsrc((void*)g.bias, 0, g.N * 4, 0x00020000);
    bf16_t* crow = nullptr;                           // previous tile: C + (row of this lane in block 0) * ldc + first column of this lane
    int mrem = 0;                                     // previous tile: rows left from that row (store guard)
    uint2 half_ = make_uint2(0u, 0u);                 // 4 packed outputs waiting for their partner block
    int tl = blockIdx.x;
    if (tl >= nwg) return;
    int m0, n0, m0n = 0, n0n = 0;
    tile_mn(tl, m0, n0);
    make_rsrc(m0, n0, ra, rb);
    unsigned gk = 0;                                  // running K-tile counter of this workgroup: stage = gk & 3
    // prologue: the first three K tiles of the first tile
    LAB_DMA(ra, rb, 0, 0) LAB_DMA(ra, rb, 1, 1) LAB_DMA(ra, rb, 2, 2)

    // one epilogue piece: accumulator block (I_, J_) of the PREVIOUS tile -> 4 halves per lane; every second piece stores 16 bytes
#define LAB_EPI_PIECE(ACCP, I_, J_)                                                                                              \
    {                                                                                                                            \
        f32x4_t bv_;      /* (inline asm: through a C++ read the compiler puts `s_waitcnt vmcnt(0)` in front -- the strip was written by DMA) */ \
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(bv_) : "v"(bias_lds + (32 * ((J_) >> 1) + 4 * ((J_) & 1)) * 4) : "memory"); \
        f32x2v x0_ = {ACCP[I_][J_][0] + bv_[0], ACCP[I_][J_][1] + bv_[1]}, x1_ = {ACCP[I_][J_][2] + bv_[2], ACCP[I_][J_][3] + bv_[3]}; \
        if (GELU) { x0_ = gelu_fast2(x0_); x1_ = gelu_fast2(x1_); }                                                              \
        const uint2 h_ = make_uint2(pack2h(x0_.x, x0_.y), pack2h(x1_.x, x1_.y));                                                 \
        if (((J_) & 1) == 0) half_ = h_;                                                                                         \
        else if (16 * (I_) < mrem) *reinterpret_cast<uint4*>(crow + (size_t)(16 * (I_)) * g.ldc + 32 * ((J_) >> 1)) = make_uint4(half_.x, half_.y, h_.x, h_.y); \
    }
    // K tile KT_ of the current tile into ACC; EP_ in 0 .. 15: epilogue piece of ACCP in the LOAD segment (EPI_ON).
    // VM_: vmcnt bound that guarantees this wave's pieces of the NEXT K tile have landed = operations issued after them: 2 K tiles x 3
    // pieces + the epilogue stores of K tiles KT_ - 2 .. KT_ (pieces with odd index store).  Group 0 waits before the barrier that ends
    // its MFMA segment, group 1 before the barrier that ends its LOAD segment: both are the barrier in front of group 0's next LOAD.
#define LAB_VMWAIT(N_) asm volatile("s_waitcnt vmcnt(" #N_ ")" ::: "memory")
#define LAB_KT(ACC, ACCP, KT_, EP_, EPI_ON, VMA_, VM0_, VM1B_, VM1C_)                                                            \
    {                                                                                                                            \
        const int st_ = (gk & 3) * LSTAGE;                                                                                       \
        const bool tail_ = !have_next && (KT_) + 3 >= nk;                                                                        \
        if (PLACE == 0) {                                                                                                        \
            if ((KT_) == nk - 1) LAB_BIAS_DMA(n0)                                                                                \
            const int kn_ = (KT_) + 3;                                                                                           \
            if (kn_ < nk) LAB_DMA(ra, rb, kn_, (gk + 3) & 3)                                                                     \
            else if (have_next) LAB_DMA(ran, rbn, kn_ - nk, (gk + 3) & 3)                                                        \
        }                                                                                                                        \
        _Pragma("unroll") for (int ii = 0; ii < 4; ++ii) fa[ii] = *reinterpret_cast<const s16x8_t*>(lds + st_ + fa_base + ii * 1024); \
        _Pragma("unroll") for (int jj = 0; jj < 4; ++jj) fb[jj] = *reinterpret_cast<const s16x8_t*>(lds + st_ + fb_base + jj * 1024); \
        if (EPI_MODE == 0 && (EPI_ON) && (EP_) >= 0 && PLACE != 1) LAB_EPI_PIECE(ACCP, ((EP_) >> 2), ((EP_) & 3))                \
        __builtin_amdgcn_sched_barrier(0);                                                                                       \
        if (grp == 1) {                                                                                                          \
            if (tail_) LAB_VMWAIT(0);                                                                                            \
            else if (PLACE == 0) LAB_VMWAIT(VMA_);                                                                               \
            else if (PLACE == 1) LAB_VMWAIT(VM1B_);                                                                              \
            else LAB_VMWAIT(VM1C_);                                                                                              \
        }                                                                                                                        \
        __builtin_amdgcn_s_barrier();                                                                                            \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                                       \
        _Pragma("unroll") for (int ii = 0; ii < 4; ++ii)                                                                         \
            _Pragma("unroll") for (int jj = 0; jj < 4; ++jj) ACC[ii][jj] = lab_mfma(fb[jj], fa[ii], ACC[ii][jj]);                \
        __builtin_amdgcn_sched_barrier(0);                                                                                       \
        if (EPI_MODE == 0 && (EPI_ON) && (EP_) >= 0 && PLACE == 1) LAB_EPI_PIECE(ACCP, ((EP_) >> 2), ((EP_) & 3))                \
        if (PLACE != 0) {                                                                                                        \
            if ((KT_) == nk - 1) LAB_BIAS_DMA(n0)                                                                                \
            const int kn_ = (KT_) + 3;                                                                                           \
            if (kn_ < nk) LAB_DMA(ra, rb, kn_, (gk + 3) & 3)                                                                     \
            else if (have_next) LAB_DMA(ran, rbn, kn_ - nk, (gk + 3) & 3)                                                        \
        }                                                                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                                       \
        if (grp == 0) {                                                                                                          \
            if (tail_) LAB_VMWAIT(0);                                                                                            \
            else if (PLACE == 0) LAB_VMWAIT(VMA_);                                                                               \
            else LAB_VMWAIT(VM0_);                                                                                               \
        }                                                                                                                        \
        __builtin_amdgcn_s_barrier();                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                                       \
        ++gk;                                                                                                                    \
    }
#define LAB_ZERO(ACC) _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 4; ++j) ACC[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    // previous-tile bookkeeping of the epilogue: row pointer of this lane (the column constants travel by LDS-DMA, LAB_BIAS_DMA)
#define LAB_PREV(M0_, N0_)                                                                                                       \
    {                                                                                                                            \
        const int nb_ = (N0_) + wn2 * 64, mb_ = (M0_) + grp * 128 + wm2 * 64 + l15;                                              \
        crow = g.C + (size_t)mb_ * g.ldc + nb_ + 8 * lq;                                                                         \
        mrem = g.M - mb_;                                                                                                        \
    }
    // The wave's 64 column constants of the CURRENT tile -> its LDS strip, by DMA (a load into registers would make the compiler put
    // `s_waitcnt vmcnt(0)` into the K loops that reuse those registers -- it cannot see the hand-placed counted waits).  Issued in the
    // last K tile's LOAD segment, in front of that segment's operand pieces: the next counted wait covers it.
#define LAB_BIAS_DMA(N0_)                                                                                                        \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rbias, (lds_ptr_t)(lds + LBIAS_OFF + wave * 256), 4, lane * 4, ((N0_) + wn2 * 64) * 4, 0, 0);
    // one output tile: K loop into ACC with the previous tile's epilogue (ACCP) spread over its first 16 K tiles (EPI_ON)
#define LAB_TILE(ACC, ACCP, EPI_ON)                                                                                              \
    {                                                                                                                            \
        const int tl_next = tl + (int)gridDim.x;                                                                                 \
        const bool have_next = tl_next < nwg;                                                                                    \
        if (have_next) { tile_mn(tl_next, m0n, n0n); make_rsrc(m0n, n0n, ran, rbn); }                                            \
        LAB_ZERO(ACC)                                                                                                            \
        if (EPI_MODE == 0 && (EPI_ON)) {                                                                                         \
            LAB_KT(ACC, ACCP, 0, 0, true, 6, 7, 4, 4) LAB_KT(ACC, ACCP, 1, 1, true, 7, 7, 3, 4) LAB_KT(ACC, ACCP, 2, 2, true, 7, 7, 4, 4) LAB_KT(ACC, ACCP, 3, 3, true, 8, 7, 3, 4)  \
            LAB_KT(ACC, ACCP, 4, 4, true, 7, 7, 4, 4) LAB_KT(ACC, ACCP, 5, 5, true, 8, 7, 3, 4) LAB_KT(ACC, ACCP, 6, 6, true, 7, 7, 4, 4) LAB_KT(ACC, ACCP, 7, 7, true, 8, 7, 3, 4)  \
            LAB_KT(ACC, ACCP, 8, 8, true, 7, 7, 4, 4) LAB_KT(ACC, ACCP, 9, 9, true, 8, 7, 3, 4) LAB_KT(ACC, ACCP, 10, 10, true, 7, 7, 4, 4) LAB_KT(ACC, ACCP, 11, 11, true, 8, 7, 3, 4) \
            LAB_KT(ACC, ACCP, 12, 12, true, 7, 7, 4, 4) LAB_KT(ACC, ACCP, 13, 13, true, 8, 7, 3, 4) LAB_KT(ACC, ACCP, 14, 14, true, 7, 7, 4, 4) LAB_KT(ACC, ACCP, 15, 15, true, 8, 7, 3, 4) \
            LAB_KT(ACC, ACCP, 16, -1, false, 7, 7, 4, 4) LAB_KT(ACC, ACCP, 17, -1, false, 7, 6, 3, 3)                            \
            for (int kt = 18; kt < nk; ++kt) LAB_KT(ACC, ACCP, kt, -1, false, 6, 6, 3, 3)                                        \
        } else {                                                                                                                 \
            LAB_KT(ACC, ACCP, 0, -1, false, 6, 7, 4, 4)                                                                          \
            for (int kt = 1; kt < nk; ++kt) LAB_KT(ACC, ACCP, kt, -1, false, 6, 6, 3, 3)                                         \
        }                                                                                                                        \
        LAB_PREV(m0, n0)                                                                                                         \
        if (EPI_MODE == 1) {                                                                                                     \
            _Pragma("unroll") for (int e = 0; e < 16; ++e) LAB_EPI_PIECE(ACC, (e >> 2), (e & 3))                                 \
        }                                                                                                                        \
        tl = tl_next; m0 = m0n; n0 = n0n; ra = ran; rb = rbn;                                                                    \
    }
    // the first K tile's pieces must have landed before anybody reads: own pieces, then everybody's
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();       // group 1 runs one segment behind group 0
    LAB_TILE(accA, accB, false)                       // first tile: nothing to finish yet
    while (true) {
        if (tl >= nwg) {                              // drain: the last tile's epilogue, nothing left to overlap it with
            if (EPI_MODE == 0) { _Pragma("unroll") for (int e = 0; e < 16; ++e) LAB_EPI_PIECE(accA, (e >> 2), (e & 3)) }
            break;
        }
        LAB_TILE(accB, accA, true)
        if (tl >= nwg) {
            if (EPI_MODE == 0) { _Pragma("unroll") for (int e = 0; e < 16; ++e) LAB_EPI_PIECE(accB, (e >> 2), (e & 3)) }
            break;
        }
        LAB_TILE(accA, accB, true)
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();       // pairs with group 1's last barrier
}

// ---------------------------------------------------------------------------------------------------------------------
static float h2f_host(unsigned short h) {
    const unsigned s = (h >> 15) & 1, e = (h >> 10) & 31, m = h & 1023;
    float v;
    if (e == 0) v = ldexpf((float)m, -24);
    else if (e == 31) v = m ? NAN : INFINITY;
    else v = ldexpf((float)(m | 1024), (int)e - 25);
    return s ? -v : v;
}
__global__ void fill_f16(bf16_t* p, size_t n, unsigned seed, float scale) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + seed;
        x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = f2h(((int)(x & 0xFFFF) - 32768) * (scale / 32768.0f));
    }
}
__global__ void fill_f32(float* p, size_t n, unsigned seed, float scale) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + seed;
        x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = ((int)(x & 0xFFFF) - 32768) * (scale / 32768.0f);
    }
}

int main(int argc, char** argv) {
    struct Shape { int M, N, K; const char* what; };
    std::vector<Shape> shapes = {{211904, 3072, 768, "fc1 + GELU, teacher windows"}, {38080, 3072, 768, "fc1 + GELU, student / global"},
                                 {211904, 768, 3072, "fc2 shape (plain bias epilogue)"}, {211904, 768, 768, "proj shape (plain bias epilogue)"},
                                 {211904, 2304, 768, "qkv shape (plain bias epilogue)"}, {8192, 8192, 8192, "square 8192"}};
    int ncu = 256;
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0)); ncu = prop.multiProcessorCount;
    for (const Shape& s : shapes) {
        const size_t na = (size_t)s.M * s.K, nb = (size_t)s.N * s.K, nc = (size_t)s.M * s.N;
        bf16_t *A, *B, *C0, *C1; float* bias;
        CHECK(hipMalloc(&A, na * 2)); CHECK(hipMalloc(&B, nb * 2)); CHECK(hipMalloc(&C0, nc * 2 + 65536)); CHECK(hipMalloc(&C1, nc * 2 + 65536));
        CHECK(hipMalloc(&bias, s.N * 4));
        fill_f16<<<1024, 256>>>(A, na, 1u, 1.0f); fill_f16<<<1024, 256>>>(B, nb, 7u, 1.7f / sqrtf((float)s.K)); fill_f32<<<16, 256>>>(bias, s.N, 3u, 0.2f);
        const int gelu = s.N == 3072 ? 1 : 0;
        LabArgs g{A, B, bias, C1, s.M, s.N, s.K, s.K, s.K, s.N, 8, gelu};
        if (s.K / LBK < 18) { printf("K too small for the lab kernel\n"); continue; }
        const int nwg = ((s.M + 255) / 256) * (s.N / 128);
        const int grid = nwg < ncu ? nwg : ncu;
        const int ldsb = LNS * LSTAGE + 8 * 256;
#define LAB_INST(E_, G_, P_) CHECK(hipFuncSetAttribute((const void*)gemm_pipe_kernel<E_, G_, P_>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb));
        LAB_INST(0, true, 0) LAB_INST(0, true, 1) LAB_INST(0, true, 2) LAB_INST(1, true, 1) LAB_INST(0, false, 0) LAB_INST(0, false, 1) LAB_INST(0, false, 2) LAB_INST(1, false, 1)
#define LAB_GO(E_, G_, P_) hipLaunchKernelGGL((gemm_pipe_kernel<E_, G_, P_>), dim3(grid), dim3(512), ldsb, 0, g)
        auto launch = [&](int mode, int place) {
            if (gelu) { if (mode == 1) LAB_GO(1, true, 1); else if (place == 0) LAB_GO(0, true, 0); else if (place == 1) LAB_GO(0, true, 1); else LAB_GO(0, true, 2); }
            else { if (mode == 1) LAB_GO(1, false, 1); else if (place == 0) LAB_GO(0, false, 0); else if (place == 1) LAB_GO(0, false, 1); else LAB_GO(0, false, 2); }
        };
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        auto time_it = [&](auto fn) {
            for (int i = 0; i < 3; ++i) fn();
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            const int reps = 10;
            for (int i = 0; i < reps; ++i) fn();
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            return ms / reps;
        };
        const float t_prod = time_it([&]() {
            // product: epi 2 = f16(acc + bias); epi 3 with outH = NULL = gelu only
            const int rc = gelu ? sed_gemm_nt(A, B, s.M, s.N, s.K, s.K, s.K, 3, bias, nullptr, nullptr, nullptr, C0, nullptr, s.N, 1.0f, 1, 1, 0)
                                : sed_gemm_nt(A, B, s.M, s.N, s.K, s.K, s.K, 2, bias, nullptr, nullptr, C0, nullptr, nullptr, s.N, 1.0f, 1, 1, 0);
            if (rc != 0) { printf("sed_gemm_nt rc %d\n", rc); exit(1); }
        });
        const float t_pipe0 = time_it([&]() { launch(0, 0); });
        const float t_pipe = time_it([&]() { launch(0, 1); });
        const float t_pipe2 = time_it([&]() { launch(0, 2); });
        const float t_ctrl = time_it([&]() { launch(1, 1); });
        CHECK(hipGetLastError());
        // compare (pipe result is in C1 from the control run, bit-identical arithmetic; re-run mode 0 for the check)
        CHECK(hipMemset(C1, 0, nc * 2));
        launch(0, 1);
        CHECK(hipDeviceSynchronize());
        const size_t ncheck = 1 << 20;
        std::vector<unsigned short> h0(ncheck), h1(ncheck);
        double worst = 0.0; size_t bad = 0;
        for (int part = 0; part < 3; ++part) {
            const size_t off = part == 0 ? 0 : (part == 1 ? (nc / 2 / s.N) * s.N : nc - ncheck);
            CHECK(hipMemcpy(h0.data(), C0 + off, ncheck * 2, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(h1.data(), C1 + off, ncheck * 2, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < ncheck; ++i) {
                const double d = fabs((double)h2f_host(h0[i]) - (double)h2f_host(h1[i]));
                if (d > worst) worst = d;
                if (h0[i] != h1[i]) ++bad;
            }
        }
        const double fl = 2.0 * s.M * s.N * s.K;
        printf("M=%6d N=%5d K=%5d  %-32s product %7.3f ms %6.1f TF | pipelined: all in LOAD %7.3f (x%.3f)  behind MFMAs %7.3f ms %6.1f TF (x%.3f)  DMA behind MFMAs %7.3f (x%.3f) | "
               "epilogue after the loop %7.3f ms %6.1f TF | max |diff| %.1e (%zu of %zu differ)\n",
               s.M, s.N, s.K, s.what, t_prod, fl / t_prod / 1e9, t_pipe0, t_prod / t_pipe0, t_pipe, fl / t_pipe / 1e9, t_prod / t_pipe, t_pipe2, t_prod / t_pipe2,
               t_ctrl, fl / t_ctrl / 1e9, worst, bad, 3 * ncheck);
        CHECK(hipFree(A)); CHECK(hipFree(B)); CHECK(hipFree(C0)); CHECK(hipFree(C1)); CHECK(hipFree(bias));
    }
    return 0;
}
