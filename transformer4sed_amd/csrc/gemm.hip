// bf16 MFMA GEMM (NT form) with fused epilogues, for gfx950.
//
//   C[M,N] = A[M,K] . B[N,K]^T        A, B bf16 row-major with K contiguous (nn.Linear weight layout for B)
//
// 128x128x64 workgroup tile, 4 waves (2x2), each wave a 64x64 sub-tile = 2x2 v_mfma_f32_32x32x16_bf16 blocks,
// fp32 accumulation.  Operand tiles are register-staged (global_load_dwordx4 issued before the MFMA phase,
// ds_write_b128 after it) into double-buffered, XOR-swizzled LDS so that the fragment ds_read_b128s are
// bank-conflict free (128-byte rows: 16-B chunk index ^= (row >> 1) & 7).  One barrier per K tile.
// Workgroup ids are remapped XCD-aware so that neighbouring tiles (which share A rows / B columns) hit the
// same per-XCD L2.
//
// Replaces (reference, all relative to /root/reference): F.linear / nn.Linear calls at
// src/models/passt/passt.py:271,274,332,342; src/models/transformer/transformerXL.py:382,493,584;
// timm Mlp fc1/fc2 in the context blocks; the conv2d of passt.py:307 (as im2col GEMM);
// src/models/passt/passt_sed.py:196 (mlm_mlp) and their autograd backward GEMMs.
#include <stdlib.h>

#include "common.h"
#include "../../include/sed_hip.h"

enum {
    EPI_F32 = 0,          // outF = acc*alpha + bias
    EPI_F32_RESID = 1,    // outF = resF + acc + bias                (residual stream; resF may alias outF)
    EPI_BF16 = 2,         // outH = bf16(acc + bias)
    EPI_GELU = 3,         // outH = bf16(h = acc + bias), outH2 = bf16(gelu(h))
    EPI_DGELU = 4,        // outH = bf16(acc * gelu'(auxH))
    EPI_ATOMIC = 5,       // atomicAdd(outF, acc*alpha)               (split-K weight gradients)
    EPI_QKV = 6,          // head-split q/k/v (+ transposed copies, + rel-pos biased queries)
    EPI_F32_BF16 = 7,     // outF = acc + bias and outH = bf16(same)
    EPI_GELU32 = 8,       // outH = 16-bit(h = acc + bias) (kept for backward), outF = fp32 gelu(h) (split-precision consumers)
};

struct GemmArgs {
    const bf16_t* A;
    const bf16_t* B;
    int M, N, K, lda, ldb, ldc, ksplit;
    float alpha;
    const float* bias;
    const float* resF;
    float* outF;
    bf16_t* outH;
    bf16_t* outH2;
    const bf16_t* auxH;
    // EPI_QKV
    bf16_t *q, *k, *v, *qt, *kt, *vt, *q2, *q2t;
    const float *pu, *pv;
    int seq, seq_pad, heads;
    int bwd_bf16; // f16 runs only: tensors that only the (bf16) backward consumes are written as bf16 straight away
    int stagger;  // v3: first-round workgroups start phase * stagger wall-clock ticks (10 ns) late, phase = 0..7
    int ncols;    // 128^2 kernel: output columns >= ncols are computed but not written (operands padded to the tile width)
};

#define TILE 128
#define BK 64

// Epilogue for one accumulator quad.  The MFMA operands are issued swapped (B fragment as the row operand), so the
// accumulator block holds C^T: a lane owns ONE output row m and a register quad holds 4 CONSECUTIVE columns n..n+3 ->
// 16-byte fp32 / 8-byte 16-bit vector stores instead of 4 scalar ones per quad.
// two packed IEEE halves -> two packed bf16
__device__ __forceinline__ unsigned h2x2_to_bf(unsigned p) { return pack2bf(h2f((bf16_t)(p & 0xFFFF)), h2f((bf16_t)(p >> 16))); }
template <bool F16>
__device__ __forceinline__ unsigned pack2_sel(float a, float b, int as_bf16) { return (F16 && !as_bf16) ? pack2<true>(a, b) : pack2<false>(a, b); }
template <bool F16>
__device__ __forceinline__ bf16_t to16_sel(float a, int as_bf16) { return (F16 && !as_bf16) ? to_16<true>(a) : to_16<false>(a); }

template <int EPI, bool F16>
__device__ __forceinline__ void epilogue_quad(const GemmArgs& g, int m, int n, const float v4[4]) {
    float b[4] = {0.f, 0.f, 0.f, 0.f};
    if (g.bias != nullptr) {
        const float4 bb = *reinterpret_cast<const float4*>(g.bias + n);
        b[0] = bb.x; b[1] = bb.y; b[2] = bb.z; b[3] = bb.w;
    }
    if (EPI == EPI_QKV) {
        const int D = g.heads * 64;
        const int which = n / D, hn = n - which * D, h = hn >> 6, d = hn & 63;
        const int bidx = m / g.seq, t = m - bidx * g.seq;
        const int bh = bidx * g.heads + h;
        bf16_t* row_dst = which == 0 ? g.q : (which == 1 ? g.k : g.v);
        bf16_t* tr_dst = which == 0 ? g.qt : (which == 1 ? g.kt : g.vt);
        float e1[4] = {0.f, 0.f, 0.f, 0.f}, e2[4] = {0.f, 0.f, 0.f, 0.f};
        if (which == 0 && g.pu != nullptr) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { e1[j] = g.pu[h * 64 + d + j]; e2[j] = g.pv[h * 64 + d + j]; }
        }
        float val[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) val[j] = v4[j] + b[j];
        // backward-only tensors (row-major V, every transposed copy except V^T) may be requested as bf16 (g.bwd_bf16)
        const int row_bf = g.bwd_bf16 && which == 2, tr_bf = g.bwd_bf16 && which != 2;
        uint2 pk;
        pk.x = pack2_sel<F16>(val[0] + e1[0], val[1] + e1[1], row_bf);
        pk.y = pack2_sel<F16>(val[2] + e1[2], val[3] + e1[3], row_bf);
        if (row_dst != nullptr) *reinterpret_cast<uint2*>(&row_dst[((size_t)bh * g.seq + t) * 64 + d]) = pk;
        if (tr_dst != nullptr) {
#pragma unroll
            for (int j = 0; j < 4; ++j) tr_dst[((size_t)bh * 64 + d + j) * g.seq_pad + t] = to16_sel<F16>(val[j] + e1[j], tr_bf);
        }
        if (which == 0 && g.q2 != nullptr) {
            pk.x = pack2<F16>(val[0] + e2[0], val[1] + e2[1]);
            pk.y = pack2<F16>(val[2] + e2[2], val[3] + e2[3]);
            *reinterpret_cast<uint2*>(&g.q2[((size_t)bh * g.seq + t) * 64 + d]) = pk;
            if (g.q2t != nullptr) {
#pragma unroll
                for (int j = 0; j < 4; ++j) g.q2t[((size_t)bh * 64 + d + j) * g.seq_pad + t] = to16_sel<F16>(val[j] + e2[j], g.bwd_bf16);
            }
        }
        return;
    }
    const size_t o = (size_t)m * g.ldc + n;
    if (EPI == EPI_F32) {
        *reinterpret_cast<float4*>(g.outF + o) =
            make_float4(v4[0] * g.alpha + b[0], v4[1] * g.alpha + b[1], v4[2] * g.alpha + b[2], v4[3] * g.alpha + b[3]);
    } else if (EPI == EPI_F32_RESID) {
        const float4 r = *reinterpret_cast<const float4*>(g.resF + o);
        *reinterpret_cast<float4*>(g.outF + o) = make_float4(r.x + v4[0] + b[0], r.y + v4[1] + b[1], r.z + v4[2] + b[2], r.w + v4[3] + b[3]);
    } else if (EPI == EPI_BF16) {
        uint2 pk;
        pk.x = pack2<F16>(v4[0] + b[0], v4[1] + b[1]);
        pk.y = pack2<F16>(v4[2] + b[2], v4[3] + b[3]);
        *reinterpret_cast<uint2*>(g.outH + o) = pk;
    } else if (EPI == EPI_GELU) {
        float h[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) h[j] = v4[j] + b[j];
        uint2 pk;
        if (g.outH != nullptr) {  // pre-activation is only kept when a backward will need it
            pk.x = pack2_sel<F16>(h[0], h[1], g.bwd_bf16);
            pk.y = pack2_sel<F16>(h[2], h[3], g.bwd_bf16);
            *reinterpret_cast<uint2*>(g.outH + o) = pk;
        }
        pk.x = pack2<F16>(gelu_fast(h[0]), gelu_fast(h[1]));
        pk.y = pack2<F16>(gelu_fast(h[2]), gelu_fast(h[3]));
        *reinterpret_cast<uint2*>(g.outH2 + o) = pk;
    } else if (EPI == EPI_DGELU) {
        const uint2 a = *reinterpret_cast<const uint2*>(g.auxH + o);
        const float h0 = to_f32<F16>((bf16_t)(a.x & 0xFFFF)), h1 = to_f32<F16>((bf16_t)(a.x >> 16));
        const float h2 = to_f32<F16>((bf16_t)(a.y & 0xFFFF)), h3 = to_f32<F16>((bf16_t)(a.y >> 16));
        uint2 pk;
        pk.x = pack2<F16>(v4[0] * gelu_fast_grad(h0), v4[1] * gelu_fast_grad(h1));
        pk.y = pack2<F16>(v4[2] * gelu_fast_grad(h2), v4[3] * gelu_fast_grad(h3));
        *reinterpret_cast<uint2*>(g.outH + o) = pk;
    } else if (EPI == EPI_ATOMIC) {
#pragma unroll
        for (int j = 0; j < 4; ++j) unsafeAtomicAdd(&g.outF[o + j], v4[j] * g.alpha);
    } else if (EPI == EPI_F32_BF16) {
        *reinterpret_cast<float4*>(g.outF + o) = make_float4(v4[0] + b[0], v4[1] + b[1], v4[2] + b[2], v4[3] + b[3]);
        uint2 pk;
        pk.x = pack2<F16>(v4[0] + b[0], v4[1] + b[1]);
        pk.y = pack2<F16>(v4[2] + b[2], v4[3] + b[3]);
        *reinterpret_cast<uint2*>(g.outH + o) = pk;
    } else if (EPI == EPI_GELU32) {
        float h[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) h[j] = v4[j] + b[j];
        uint2 pk;
        pk.x = pack2_sel<F16>(h[0], h[1], g.bwd_bf16);
        pk.y = pack2_sel<F16>(h[2], h[3], g.bwd_bf16);
        *reinterpret_cast<uint2*>(g.outH + o) = pk;
        *reinterpret_cast<float4*>(g.outF + o) = make_float4(gelu_erf(h[0]), gelu_erf(h[1]), gelu_erf(h[2]), gelu_erf(h[3]));
    }
}

// MFMA phase over one 64-deep K tile for a wave's 2x2 accumulator blocks, with the LDS->register fragment loads of
// k-step s+1 issued BEFORE the four MFMAs of k-step s (register double buffering), so ds_read latency hides under MFMA
// issue instead of serialising "4 reads - wait - 4 MFMAs" per k-step.
template <bool F16, bool SWAP>
__device__ __forceinline__ void mfma_tile(const unsigned char* la, const unsigned char* lb, const int (&arow)[2],
                                          const int (&brow)[2], int lg, f32x16_t (&acc)[2][2]) {
    s16x8_t af[2][2], bfr[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        af[0][i] = *reinterpret_cast<const s16x8_t*>(la + arow[i] * 128 + ((lg ^ ((arow[i] >> 1) & 7)) << 4));
        bfr[0][i] = *reinterpret_cast<const s16x8_t*>(lb + brow[i] * 128 + ((lg ^ ((brow[i] >> 1) & 7)) << 4));
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int cur = s & 1, nxt = cur ^ 1;
        if (s < 3) {
            const int ch = 2 * (s + 1) + lg;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                af[nxt][i] = *reinterpret_cast<const s16x8_t*>(la + arow[i] * 128 + ((ch ^ ((arow[i] >> 1) & 7)) << 4));
                bfr[nxt][i] = *reinterpret_cast<const s16x8_t*>(lb + brow[i] * 128 + ((ch ^ ((brow[i] >> 1) & 7)) << 4));
            }
        }
        __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ahead of the MFMAs (else the reads are sunk onto reused VGPRs)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i][j] = SWAP ? mfma32t<F16>(bfr[cur][j], af[cur][i], acc[i][j]) : mfma32t<F16>(af[cur][i], bfr[cur][j], acc[i][j]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int EPI, bool F16, bool GLDS>
__global__ __launch_bounds__(256) void gemm_nt_kernel(const GemmArgs g) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2][2][TILE * BK * 2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int ntn = g.N / TILE, ntm = (g.M + TILE - 1) / TILE, nwg = ntm * ntn;
    // XCD-contiguous linear tile id, then "grouped" order inside it: 8 tile-rows x all tile-columns form a group whose
    // A panel (8 x 192 KB at K = 768) and B panel stay resident in the XCD's 4 MB L2 while its ~64 concurrent
    // workgroups sweep it (otherwise every tile-row re-streams the whole weight matrix: ~64 FLOP/B, bandwidth-bound).
    const int t = xcd_remap(blockIdx.x, nwg);
    const int group_size = 8 * ntn, gid = t / group_size, first_m = gid * 8;
    const int gm = (ntm - first_m) < 8 ? (ntm - first_m) : 8;
    const int tin = t - gid * group_size;
    const int m0 = (first_m + tin % gm) * TILE, n0 = (tin / gm) * TILE;
    const int ktiles = g.K / BK;
    const int kt_begin = (int)(((long long)blockIdx.y * ktiles) / g.ksplit);
    const int kt_end = (int)(((long long)(blockIdx.y + 1) * ktiles) / g.ksplit);

    // staging: thread -> 16-B chunk c of rows r0 + 32 i (i = 0..3) of both operand tiles.  Named registers (not arrays
    // captured by lambdas): arrays end up in scratch and every prefetch then stalls on vmcnt(0) + scratch_store.
    const int c = tid & 7, r0 = tid >> 3;
    const int am_last = g.M - 1;
    const int am0 = (m0 + r0) < g.M ? (m0 + r0) : am_last, am1 = (m0 + r0 + 32) < g.M ? (m0 + r0 + 32) : am_last;
    const int am2 = (m0 + r0 + 64) < g.M ? (m0 + r0 + 64) : am_last, am3 = (m0 + r0 + 96) < g.M ? (m0 + r0 + 96) : am_last;
    const bf16_t* ap0 = g.A + (size_t)am0 * g.lda + c * 8;
    const bf16_t* ap1 = g.A + (size_t)am1 * g.lda + c * 8;
    const bf16_t* ap2 = g.A + (size_t)am2 * g.lda + c * 8;
    const bf16_t* ap3 = g.A + (size_t)am3 * g.lda + c * 8;
    const bf16_t* bp0 = g.B + (size_t)(n0 + r0) * g.ldb + c * 8;
    const bf16_t* bp1 = bp0 + (size_t)32 * g.ldb;
    const bf16_t* bp2 = bp0 + (size_t)64 * g.ldb;
    const bf16_t* bp3 = bp0 + (size_t)96 * g.ldb;
    const int lo0 = r0 * 128 + ((c ^ ((r0 >> 1) & 7)) << 4);
    const int lo1 = (r0 + 32) * 128 + ((c ^ (((r0 + 32) >> 1) & 7)) << 4);
    const int lo2 = (r0 + 64) * 128 + ((c ^ (((r0 + 64) >> 1) & 7)) << 4);
    const int lo3 = (r0 + 96) * 128 + ((c ^ (((r0 + 96) >> 1) & 7)) << 4);
    uint4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
#define GEMM_GLOAD(kt)                                                          \
    {                                                                           \
        const size_t ko = (size_t)(kt) * BK;                                    \
        ra0 = *reinterpret_cast<const uint4*>(ap0 + ko);                        \
        ra1 = *reinterpret_cast<const uint4*>(ap1 + ko);                        \
        ra2 = *reinterpret_cast<const uint4*>(ap2 + ko);                        \
        ra3 = *reinterpret_cast<const uint4*>(ap3 + ko);                        \
        rb0 = *reinterpret_cast<const uint4*>(bp0 + ko);                        \
        rb1 = *reinterpret_cast<const uint4*>(bp1 + ko);                        \
        rb2 = *reinterpret_cast<const uint4*>(bp2 + ko);                        \
        rb3 = *reinterpret_cast<const uint4*>(bp3 + ko);                        \
    }
#define GEMM_LSTORE(buf)                                                        \
    {                                                                           \
        *reinterpret_cast<uint4*>(&lds[buf][0][lo0]) = ra0;                     \
        *reinterpret_cast<uint4*>(&lds[buf][0][lo1]) = ra1;                     \
        *reinterpret_cast<uint4*>(&lds[buf][0][lo2]) = ra2;                     \
        *reinterpret_cast<uint4*>(&lds[buf][0][lo3]) = ra3;                     \
        *reinterpret_cast<uint4*>(&lds[buf][1][lo0]) = rb0;                     \
        *reinterpret_cast<uint4*>(&lds[buf][1][lo1]) = rb1;                     \
        *reinterpret_cast<uint4*>(&lds[buf][1][lo2]) = rb2;                     \
        *reinterpret_cast<uint4*>(&lds[buf][1][lo3]) = rb3;                     \
    }

    f32x16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment read offsets (row part), chunk part depends on the k-step
    const int lr = lane & 31, lg = lane >> 5;
    int arow[2], brow[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        arow[i] = wm * 64 + i * 32 + lr;
        brow[i] = wn * 64 + i * 32 + lr;
    }

// split-K weight gradients keep the un-swapped accumulator layout (lane = output column): their atomics then hit 2 cache
// lines per instruction instead of 64
#define GEMM_COMPUTE(buf) mfma_tile<F16, EPI != EPI_ATOMIC>(lds[buf][0], lds[buf][1], arow, brow, lg, acc)
    if (GLDS) {
        // Direct-to-LDS staging (global_load_lds_dwordx4): each wave-instruction DMAs 64 x 16 B = 8 tile rows straight
        // into LDS (lane-linear destination), no VGPR round trip and no ds_write pass.  The XOR swizzle therefore moves to
        // the per-lane SOURCE address: lane l fills physical chunk l & 7 of row 8 p + (l >> 3) with logical chunk
        // (l & 7) ^ ((row >> 1) & 7) -- the same involution the fragment reads apply.  Wave w owns pieces 4 w .. 4 w + 3.
        const int prow = lane >> 3, pch = lane & 7;
        const bf16_t* asrc[4];
        const bf16_t* bsrc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (wave * 4 + i) * 8 + prow;
            const int cl = pch ^ ((row >> 1) & 7);
            int am = m0 + row;
            am = am < g.M ? am : g.M - 1;
            asrc[i] = g.A + (size_t)am * g.lda + cl * 8;
            bsrc[i] = g.B + (size_t)(n0 + row) * g.ldb + cl * 8;
        }
#define GEMM_DMA(kt, buf)                                                                                                \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                      \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(asrc[i] + (size_t)(kt) * BK),   \
                                         (__attribute__((address_space(3))) void*)(&lds[buf][0][(wave * 4 + i) * 1024]), \
                                         16, 0, 0);                                                                      \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bsrc[i] + (size_t)(kt) * BK),   \
                                         (__attribute__((address_space(3))) void*)(&lds[buf][1][(wave * 4 + i) * 1024]), \
                                         16, 0, 0);                                                                      \
    }
        if (kt_begin < kt_end) GEMM_DMA(kt_begin, 0);
        __syncthreads();  // (drains the LDS-DMA: hipcc emits vmcnt(0) before a barrier while one is in flight)
        for (int kt = kt_begin; kt < kt_end; ++kt) {
            const int buf = (kt - kt_begin) & 1;
            if (kt + 1 < kt_end) GEMM_DMA(kt + 1, buf ^ 1);
            GEMM_COMPUTE(buf);
            __syncthreads();
        }
    } else {
        if (kt_begin < kt_end) {
            GEMM_GLOAD(kt_begin);
            GEMM_LSTORE(0);
        }
        __syncthreads();
        for (int kt = kt_begin; kt < kt_end; ++kt) {
            const int buf = (kt - kt_begin) & 1;
            // register-staged variant: prefetch -> MFMA -> LDS store, pinned so the loads are not sunk to their ds_writes
            GEMM_GLOAD(kt + 1 < kt_end ? kt + 1 : kt);
            __builtin_amdgcn_sched_barrier(0);
            GEMM_COMPUTE(buf);
            __builtin_amdgcn_sched_barrier(0);
            GEMM_LSTORE(buf ^ 1);
            __syncthreads();
        }
    }

    if (EPI == EPI_ATOMIC) {  // un-swapped: column = lane & 31, rows from the register index
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = n0 + wn * 64 + j * 32 + lr;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wm * 64 + i * 32 + mfma32_row(r, lg);
                    if (m < g.M) unsafeAtomicAdd(&g.outF[(size_t)m * g.ldc + n], acc[i][j][r] * g.alpha);
                }
            }
        return;
    }
    // accumulator block (i, j) holds C^T: column index (lane & 31) = row m of C, register rows = columns n of C
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + wm * 64 + i * 32 + lr;
        if (m >= g.M) continue;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = n0 + wn * 64 + j * 32 + 8 * q + 4 * lg;
                if (n >= g.ncols) continue;
                const float v4[4] = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                epilogue_quad<EPI, F16>(g, m, n, v4);
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// v3: 256 x 256 x 64 workgroup tile, 8 waves (2 x 4), 128 x 64 per wave (4 x 2 MFMA 32x32x16 blocks, 128 accumulator
// registers), two 64-KiB LDS stages filled by direct-to-LDS DMA one K tile ahead.
// Why (tools/ablate, MI355X): with 64x64 per-wave tiles the LDS is the shared bottleneck -- DMA fills and fragment reads
// together take as long as the MFMAs (MFMA-only 1571 TFLOP/s, DMA-only about the same, combined ~900).  This geometry
// needs 27 % fewer LDS read bytes and 50 % fewer LDS write bytes per FLOP and 6 ds_read_b128 per 8 MFMAs, and one K tile
// of MFMA work per wave pair (~2k cycles) covers the DMA latency.
// ---------------------------------------------------------------------------------------------------------------------
// ---- v3 staged epilogue ------------------------------------------------------------------------------------------------
// After the K loop each wave owns a private 17 KiB LDS region.  The accumulators (C^T layout: lane = row, register quad = 4
// columns) are written there as a row-major [rows][64] sub-tile (padded row stride: conflict-free), then read back so that
// 8 (16-bit) or 16 (fp32) consecutive lanes cover one full output row: every global store / residual load is a run of whole
// 128- / 256-byte rows instead of 64 different cache lines per instruction.
#define V3_WLDS 17408          // per-wave bytes: 128 rows x 136 B (16-bit) or 64 rows x 264 B (fp32, two passes)
#define V3_RS16 136
#define V3_RS32 264

// Per-lane column constants (bias, plus the rel-pos u / v vector for the q projection) for the 32 columns a lane owns in
// the C^T accumulator layout: fetched ONCE, before any store -- the output pointers may alias them as far as the compiler
// knows, so a load placed between stores costs a full `s_waitcnt vmcnt(0)` round trip each time (measured: 8 us / tile).
__device__ __forceinline__ void v3_col_consts(float (&bv)[2][4][4], const float* bias_n, const float* extra, int lg) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int col = j * 32 + 8 * q + 4 * lg;
            float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
            if (bias_n != nullptr) b = *reinterpret_cast<const float4*>(bias_n + col);
            if (extra != nullptr) {
                const float4 x = *reinterpret_cast<const float4*>(extra + col);
                b.x += x.x; b.y += x.y; b.z += x.z; b.w += x.w;
            }
            bv[j][q][0] = b.x; bv[j][q][1] = b.y; bv[j][q][2] = b.z; bv[j][q][3] = b.w;
        }
}

template <bool F16, int MODE>
__device__ __forceinline__ void v3_stage16(unsigned char* wl, const f32x16_t (&acc)[4][2], const float (&bv)[2][4][4], int lr, int lg) {
    // mode 0: acc + column constant   1: gelu_fast(acc + column constant)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int col = j * 32 + 8 * q + 4 * lg;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; e += 2) {
                    f32x2v x = {acc[i][j][4 * q + e] + bv[j][q][e], acc[i][j][4 * q + e + 1] + bv[j][q][e + 1]};
                    if (MODE == 1) x = gelu_fast2(x);
                    v[e] = x.x; v[e + 1] = x.y;
                }
                uint2 pk;
                pk.x = pack2<F16>(v[0], v[1]);
                pk.y = pack2<F16>(v[2], v[3]);
                *reinterpret_cast<uint2*>(wl + (i * 32 + lr) * V3_RS16 + col * 2) = pk;
            }
}
// 64 staged rows (accumulator blocks i0, i0 + 1) as fp32
__device__ __forceinline__ void v3_stage32(unsigned char* wl, const f32x16_t (&acc)[4][2], int i0, int lr, int lg) {
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int col = j * 32 + 8 * q + 4 * lg;
                const f32x16_t& a = acc[i0 + ii][j];
                *reinterpret_cast<float4*>(wl + (ii * 32 + lr) * V3_RS32 + col * 4) =
                    make_float4(a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]);
            }
}

// C-tile stores of the 256^2 kernel are non-temporal: the tile is not re-read by this kernel, and write-allocating it evicts
// the A/B panels that the neighbouring column tiles still need from the 4 MB L2 (+2..4 % on the model's shapes, 156 -> 154 ms
// per step; -DV3_NT_STORE=0 restores plain stores).
#ifndef V3_NT_STORE
#define V3_NT_STORE 1
#endif
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
template <class T>
__device__ __forceinline__ void v3_st(void* p, const T& v) {
#if V3_NT_STORE
    if constexpr (sizeof(T) == 16) {
        u32x4_t t;
        __builtin_memcpy(&t, &v, 16);
        __builtin_nontemporal_store(t, reinterpret_cast<u32x4_t*>(p));
    } else {
        u32x2_t t;
        __builtin_memcpy(&t, &v, 8);
        __builtin_nontemporal_store(t, reinterpret_cast<u32x2_t*>(p));
    }
#else
    *reinterpret_cast<T*>(p) = v;
#endif
}

// Side input of one batch (8 rows per lane, rows m0 + 4 u + lane/16): residual rows (fp32) or saved pre-activations (16-bit)
template <int EPI>
struct V3Side {
    float4 r[EPI == EPI_F32_RESID ? 8 : 1];
    uint2 a[EPI == EPI_DGELU ? 8 : 1];
};
template <int EPI>
__device__ __forceinline__ void v3_side_load(V3Side<EPI>& sd, const GemmArgs& g, int m0, int n, int lane) {
    if constexpr (EPI == EPI_F32_RESID || EPI == EPI_DGELU) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int m = m0 + u * 4 + (lane >> 4);
            const size_t o = (size_t)(m < g.M ? m : g.M - 1) * g.ldc + n;
            if constexpr (EPI == EPI_F32_RESID) sd.r[u] = *reinterpret_cast<const float4*>(g.resF + o);
            else sd.a[u] = *reinterpret_cast<const uint2*>(g.auxH + o);
        }
    }
}
// rows m0 .. m0 + 31 of the output = staged rows srow0 .. srow0 + 31
template <int EPI, bool F16>
__device__ __forceinline__ void v3_store_batch(const GemmArgs& g, const unsigned char* wl, const V3Side<EPI>& sd, const float4 b,
                                               int m0, int srow0, int n, int c4, int lane) {
    float4 vv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) vv[u] = *reinterpret_cast<const float4*>(wl + (srow0 + u * 4 + (lane >> 4)) * V3_RS32 + c4 * 16);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int m = m0 + u * 4 + (lane >> 4);
        if (m >= g.M) continue;
        float4 v = vv[u];
        const size_t o = (size_t)m * g.ldc + n;
        if constexpr (EPI == EPI_F32) {
            v3_st<float4>(g.outF + o, make_float4(v.x * g.alpha + b.x, v.y * g.alpha + b.y, v.z * g.alpha + b.z, v.w * g.alpha + b.w));
        } else if constexpr (EPI == EPI_F32_RESID) {
            const float4 r = sd.r[u];
            v3_st<float4>(g.outF + o, make_float4(r.x + v.x + b.x, r.y + v.y + b.y, r.z + v.z + b.z, r.w + v.w + b.w));
        } else if constexpr (EPI == EPI_F32_BF16) {
            v = make_float4(v.x + b.x, v.y + b.y, v.z + b.z, v.w + b.w);
            v3_st<float4>(g.outF + o, v);
            uint2 pk; pk.x = pack2<F16>(v.x, v.y); pk.y = pack2<F16>(v.z, v.w);
            v3_st<uint2>(g.outH + o, pk);
        } else if constexpr (EPI == EPI_GELU32) {
            v = make_float4(v.x + b.x, v.y + b.y, v.z + b.z, v.w + b.w);
            uint2 pk; pk.x = pack2_sel<F16>(v.x, v.y, g.bwd_bf16); pk.y = pack2_sel<F16>(v.z, v.w, g.bwd_bf16);
            v3_st<uint2>(g.outH + o, pk);
            v3_st<float4>(g.outF + o, make_float4(gelu_erf(v.x), gelu_erf(v.y), gelu_erf(v.z), gelu_erf(v.w)));
        } else if constexpr (EPI == EPI_DGELU) {
            const uint2 a = sd.a[u];
            const float h0 = to_f32<F16>((bf16_t)(a.x & 0xFFFF)), h1 = to_f32<F16>((bf16_t)(a.x >> 16));
            const float h2 = to_f32<F16>((bf16_t)(a.y & 0xFFFF)), h3 = to_f32<F16>((bf16_t)(a.y >> 16));
            uint2 pk;
            const f32x2v ga = gelu_fast_grad2(f32x2v{h0, h1}), gb = gelu_fast_grad2(f32x2v{h2, h3});
            pk.x = pack2<F16>(v.x * ga.x, v.y * ga.y);
            pk.y = pack2<F16>(v.z * gb.x, v.w * gb.y);
            v3_st<uint2>(g.outH + o, pk);
        }
    }
}

template <int EPI>
struct V3Consts {  // per-lane bias values, fetched before the K loop so their latency is off the epilogue's critical path
    static constexpr bool kStaged16 = (EPI == EPI_BF16 || EPI == EPI_GELU);  // QKV is at the VGPR cap: it loads late
    float bv[kStaged16 ? 2 : 1][kStaged16 ? 4 : 1][4];
};
template <int EPI>
__device__ __forceinline__ void v3_load_consts(V3Consts<EPI>& c, const GemmArgs& g, int nb, int lane) {
    if (EPI == EPI_QKV) {
        c.bv[0][0][0] = 0.f;
    } else if (V3Consts<EPI>::kStaged16) {
        float t[2][4][4];
        v3_col_consts(t, g.bias != nullptr ? g.bias + nb : nullptr, nullptr, lane >> 5);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) c.bv[V3Consts<EPI>::kStaged16 ? j : 0][V3Consts<EPI>::kStaged16 ? q : 0][e] = t[j][q][e];
    } else {
        float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
        if (g.bias != nullptr) b = *reinterpret_cast<const float4*>(g.bias + nb + (lane & 15) * 4);
        c.bv[0][0][0] = b.x; c.bv[0][0][1] = b.y; c.bv[0][0][2] = b.z; c.bv[0][0][3] = b.w;
    }
}

template <int EPI, bool F16>
__device__ __forceinline__ void v3_epilogue(const GemmArgs& g, const f32x16_t (&acc)[4][2], const V3Consts<EPI>& cc,
                                            unsigned char* wl, int mb, int nb, int lane) {
    // mb = first row of this wave's 128 x 64 sub-tile, nb = its first column
    const int lr = lane & 31, lg = lane >> 5;
    if constexpr (EPI == EPI_BF16 || EPI == EPI_GELU) {
        const float (&bv)[2][4][4] = cc.bv;
#pragma unroll
        for (int pass = 0; pass < (EPI == EPI_GELU ? 2 : 1); ++pass) {
            bf16_t* out = (EPI == EPI_GELU && pass == 1) ? g.outH2 : g.outH;
            if (out == nullptr) continue;
            if (EPI == EPI_GELU && pass == 1) v3_stage16<F16, 1>(wl, acc, bv, lr, lg);
            else if (EPI == EPI_GELU && F16 && g.bwd_bf16) v3_stage16<false, 0>(wl, acc, bv, lr, lg);  // pre-activation for the bf16 backward
            else v3_stage16<F16, 0>(wl, acc, bv, lr, lg);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int rb = 0; rb < 16; rb += 8) {
                uint4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const uint4*>(wl + ((rb + u) * 8 + (lane >> 3)) * V3_RS16 + (lane & 7) * 16);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int row = (rb + u) * 8 + (lane >> 3);
                    if (mb + row < g.M) v3_st<uint4>(out + (size_t)(mb + row) * g.ldc + nb + (lane & 7) * 8, v[u]);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        return;
    }
    if constexpr (EPI == EPI_QKV) {
        const int D = g.heads * 64;
        const int which = nb / D, h = (nb - which * D) >> 6;
        bf16_t* row_dst = which == 0 ? g.q : (which == 1 ? g.k : g.v);
        bf16_t* tr_dst = which == 0 ? g.qt : (which == 1 ? g.kt : g.vt);
        const int npass = (which == 0 && g.q2 != nullptr) ? 2 : 1;
        for (int pass = 0; pass < npass; ++pass) {
            const float* extra = nullptr;
            if (which == 0 && g.pu != nullptr) extra = (pass == 0 ? g.pu : g.pv) + h * 64;
            bf16_t* rd = pass == 0 ? row_dst : g.q2;
            bf16_t* td = pass == 0 ? tr_dst : g.q2t;
            float bv[2][4][4];
            v3_col_consts(bv, g.bias != nullptr ? g.bias + nb : nullptr, extra, lg);
            v3_stage16<F16, 0>(wl, acc, bv, lr, lg);
            // Tensors that only the bf16 backward reads (row-major V; Q^T, K^T, (q+v)^T) are emitted as bf16 when g.bwd_bf16 is
            // set: converted from the staged f16 tile on the way out (same double rounding as a later in-place conversion,
            // without the extra pass over HBM); V^T and the row-major q / k stay f16.
            const bool row_bf = F16 && g.bwd_bf16 && which == 2 && pass == 0;
            const bool tr_bf = F16 && g.bwd_bf16 && !(which == 2 && pass == 0);
            __builtin_amdgcn_wave_barrier();
            if (rd != nullptr) {  // (the row-major V is only needed by the backward: inference passes v = NULL)
#pragma unroll
                for (int rb = 0; rb < 16; rb += 8) {
                    uint4 v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const uint4*>(wl + ((rb + u) * 8 + (lane >> 3)) * V3_RS16 + (lane & 7) * 16);
                    if (row_bf) {
#pragma unroll
                        for (int u = 0; u < 8; ++u) v[u] = make_uint4(h2x2_to_bf(v[u].x), h2x2_to_bf(v[u].y), h2x2_to_bf(v[u].z), h2x2_to_bf(v[u].w));
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int m = mb + (rb + u) * 8 + (lane >> 3);
                        if (m < g.M) {
                            const int bidx = m / g.seq, t = m - bidx * g.seq;
                            v3_st<uint4>(rd + ((size_t)(bidx * g.heads + h) * g.seq + t) * 64 + (lane & 7) * 8, v[u]);
                        }
                    }
                }
            }
            if (td != nullptr && (g.seq & 1) == 0) {
                // transposed copy [bh][d][seq_pad]: a lane owns a PAIR of consecutive tokens (same clip: seq is even and
                // the pair starts on an even row) and one of two interleaved d columns -> 4-byte stores, 32 lanes = 128 B
                const int pr = (lane & 31) * 2, dsel = lane >> 5;
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int row = half * 64 + pr, m = mb + row;
                    if (m < g.M) {
                        const int bidx = m / g.seq, t = m - bidx * g.seq;
                        bf16_t* base = td + (size_t)(bidx * g.heads + h) * 64 * g.seq_pad + t;
                        const unsigned short* s0 = reinterpret_cast<const unsigned short*>(wl + row * V3_RS16);
                        const unsigned short* s1 = reinterpret_cast<const unsigned short*>(wl + (row + 1) * V3_RS16);
#pragma unroll 8
                        for (int dd = 0; dd < 32; ++dd) {
                            const int d = dd * 2 + dsel;
                            unsigned pk = (unsigned)s0[d] | ((unsigned)s1[d] << 16);
                            if (tr_bf) pk = h2x2_to_bf(pk);
                            *reinterpret_cast<unsigned*>(base + (size_t)d * g.seq_pad) = pk;
                        }
                    }
                }
            } else if (td != nullptr) {
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int row = half * 64 + lane, m = mb + row;
                    if (m < g.M) {
                        const int bidx = m / g.seq, t = m - bidx * g.seq;
                        bf16_t* base = td + (size_t)(bidx * g.heads + h) * 64 * g.seq_pad + t;
                        const unsigned short* src = reinterpret_cast<const unsigned short*>(wl + row * V3_RS16);
#pragma unroll 8
                        for (int d = 0; d < 64; ++d) base[(size_t)d * g.seq_pad] = tr_bf ? f2bf(h2f(src[d])) : src[d];
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        return;
    }
    if constexpr (EPI == EPI_ATOMIC) {
        // split-K weight gradients: stage the tile row-major, then one atomic per lane with the 64 lanes on 64 consecutive
        // columns (256 contiguous bytes per instruction)
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            v3_stage32(wl, acc, 2 * pass, lr, lg);
            __builtin_amdgcn_wave_barrier();
#pragma unroll 8
            for (int row = 0; row < 64; ++row) {
                const int m = mb + pass * 64 + row;
                const float v = *reinterpret_cast<const float*>(wl + row * V3_RS32 + lane * 4);
                if (m < g.M) unsafeAtomicAdd(&g.outF[(size_t)m * g.ldc + nb + lane], v * g.alpha);
            }
            __builtin_amdgcn_wave_barrier();
        }
        return;
    }
    // fp32-staged epilogues (two passes of 64 rows): EPI_F32, EPI_F32_RESID, EPI_F32_BF16, EPI_GELU32, EPI_DGELU
    // 16 lanes x 16 B = one 256-B fp32 row; the lane's 4 columns (and so its bias) are the same for every row.  Four batches
    // of 8 rows-per-lane; the residual / saved pre-activation of batch k+1 is requested before batch k is stored, so its
    // latency hides under the stores (loads placed between stores would each cost a full wait, see v3_col_consts).
    const int c4 = lane & 15, n = nb + c4 * 4;
    const float4 b = make_float4(cc.bv[0][0][0], cc.bv[0][0][1], cc.bv[0][0][2], cc.bv[0][0][3]);
    V3Side<EPI> s0, s1;
    v3_side_load<EPI>(s0, g, mb, n, lane);
    v3_stage32(wl, acc, 0, lr, lg);
    __builtin_amdgcn_wave_barrier();
    v3_side_load<EPI>(s1, g, mb + 32, n, lane);
    v3_store_batch<EPI, F16>(g, wl, s0, b, mb, 0, n, c4, lane);
    v3_side_load<EPI>(s0, g, mb + 64, n, lane);
    v3_store_batch<EPI, F16>(g, wl, s1, b, mb + 32, 32, n, c4, lane);
    __builtin_amdgcn_wave_barrier();
    v3_stage32(wl, acc, 2, lr, lg);
    __builtin_amdgcn_wave_barrier();
    v3_side_load<EPI>(s1, g, mb + 96, n, lane);
    v3_store_batch<EPI, F16>(g, wl, s0, b, mb + 64, 0, n, c4, lane);
    v3_store_batch<EPI, F16>(g, wl, s1, b, mb + 96, 32, n, c4, lane);
    __builtin_amdgcn_wave_barrier();
}

#define V3_T 256
#define V3_STAGE (64 * 1024)
#ifdef SED_GEMM_TRACE  // developer build only (tools/ablate/trace_v3.py): per-workgroup phase timestamps
__device__ unsigned long long* sed_trace_buf = nullptr;
__device__ int sed_trace_wave = 0;   // which wave of the workgroup reports its per-K-tile waits
extern "C" int sed_debug_set_gemm_trace(unsigned long long* p) {
    return hipMemcpyToSymbol(HIP_SYMBOL(sed_trace_buf), &p, sizeof(p)) == hipSuccess ? 0 : -1;
}
extern "C" int sed_debug_set_gemm_trace_wave(int w) {
    return hipMemcpyToSymbol(HIP_SYMBOL(sed_trace_wave), &w, sizeof(w)) == hipSuccess ? 0 : -1;
}
#define V3_TRACE(slot) do { if (tid == 0 && sed_trace_buf != nullptr) sed_trace_buf[(size_t)blockIdx.x * 8 + (slot)] = wall_clock64(); } while (0)
#else
#define V3_TRACE(slot) do {} while (0)
#endif
#define V3_LDS (8 * V3_WLDS > 2 * V3_STAGE ? 8 * V3_WLDS : 2 * V3_STAGE)
template <int EPI, bool F16>
__global__ __launch_bounds__(512) void gemm_nt_v3_kernel(const GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds3[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    V3_TRACE(0);
#ifdef SED_GEMM_TRACE
    if (tid == 0 && sed_trace_buf != nullptr) {
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        sed_trace_buf[(size_t)blockIdx.x * 8 + 7] = ((unsigned long long)xcc << 32) | hw;
    }
#endif
    const int ntn = g.N / V3_T, ntm = (g.M + V3_T - 1) / V3_T, nwg = ntm * ntn;
    if (g.stagger > 0 && blockIdx.x < 256) {
        // De-phase the first round: identical workgroups launched together reach their epilogues together and the
        // whole chip's C tiles hit HBM in one burst (measured: 8-25 us of store issue per tile vs ~2 us alone).
        const unsigned long long wait = (unsigned long long)(((blockIdx.x >> 3) & 7) * g.stagger);
        const unsigned long long t0 = wall_clock64();
        while (wall_clock64() - t0 < wait) __builtin_amdgcn_s_sleep(16);
    }
    const int t = xcd_remap(blockIdx.x, nwg);
    const int group_size = 4 * ntn, gid = t / group_size, first_m = gid * 4;
    const int gm = (ntm - first_m) < 4 ? (ntm - first_m) : 4;
    const int tin = t - gid * group_size;
    const int m0 = (first_m + tin % gm) * V3_T, n0 = (tin / gm) * V3_T;
    const int ktiles = g.K / BK;
    const int kt_begin = (int)(((long long)blockIdx.y * ktiles) / g.ksplit);
    const int kt_end = (int)(((long long)(blockIdx.y + 1) * ktiles) / g.ksplit);
    const int nk = kt_end - kt_begin;

    // DMA: 32 (A) + 32 (B) pieces of 8 rows x 128 B per stage; wave w owns pieces 8 w .. 8 w + 7 (waves 0-3: A, 4-7: B)
    const int prow = lane >> 3, pch = lane & 7;
    const bf16_t* src[8];
    int dst[8];
    const bool isB = wave >= 4;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int piece = (wave & 3) * 8 + i;
        const int row = piece * 8 + prow;
        const int cl = pch ^ ((row >> 1) & 7);
        int am = m0 + row;
        am = am < g.M ? am : g.M - 1;
        src[i] = isB ? g.B + (size_t)(n0 + row) * g.ldb + cl * 8 : g.A + (size_t)am * g.lda + cl * 8;
        dst[i] = (isB ? 32768 : 0) + piece * 1024;
    }
#define V3_DMA(kt, stage)                                                                                                 \
    _Pragma("unroll") for (int i = 0; i < 8; ++i)                                                                         \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + (size_t)(kt) * BK),     \
                                         (__attribute__((address_space(3))) void*)(lds3 + (stage) * V3_STAGE + dst[i]),   \
                                         16, 0, 0);
#define V3_DMA2(kt, stage, I0)                                                                                            \
    _Pragma("unroll") for (int i = (I0); i < (I0) + 2; ++i)                                                               \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + (size_t)(kt) * BK),     \
                                         (__attribute__((address_space(3))) void*)(lds3 + (stage) * V3_STAGE + dst[i]),   \
                                         16, 0, 0);
    f32x16_t acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int lr = lane & 31, lg = lane >> 5;
    int aoff[4], boff[2], aswz[4], bswz[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int r = wm * 128 + i * 32 + lr; aoff[i] = r * 128; aswz[i] = (r >> 1) & 7; }
#pragma unroll
    for (int j = 0; j < 2; ++j) { const int r = wn * 64 + j * 32 + lr; boff[j] = 32768 + r * 128; bswz[j] = (r >> 1) & 7; }

    V3Consts<EPI> cc;
    v3_load_consts<EPI>(cc, g, n0 + wn * 64, lane);
#ifndef V3_XBAR
#define V3_XBAR 1
#endif
#ifndef V3_DMA_SPREAD
#define V3_DMA_SPREAD 1   // +4-5 % on every shape of tools/gemm_bench.py (ablation: the DMA's LDS writes cost the loop 22 %, the fragment reads 11 %)
#endif
#if V3_XBAR
    // The per-K-tile barrier sits BEFORE the last k-step's MFMAs instead of at the top of the tile: by then every wave has
    // finished reading the current stage (its k-step-3 fragments are in registers), so right after the barrier the stage can be
    // handed to the DMA of tile it+2, and the first fragments of tile it+1 (landed: vmcnt(0) + barrier) are requested from the
    // other stage -- both latencies run under the 8 MFMAs of k-step 3 instead of stalling the top of the next tile (a phase trace
    // showed the waves never wait for DMA data; they idle ~400 cycles per tile between barrier and first MFMA).
#ifdef SED_GEMM_TRACE
    unsigned long long tw = 0, tb = 0;   // (per-wave wait accounting exists for the V3_XBAR=0 loop only)
#endif
    s16x8_t af[2][4], bfr[2][2];
#define V3_FRAGS(SET, BASE, KS)                                                                                           \
    {                                                                                                                     \
        const int ch_ = 2 * (KS) + lg;                                                                                    \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                     \
            af[SET][i] = *reinterpret_cast<const s16x8_t*>((BASE) + aoff[i] + ((ch_ ^ aswz[i]) << 4));                    \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                     \
            bfr[SET][j] = *reinterpret_cast<const s16x8_t*>((BASE) + boff[j] + ((ch_ ^ bswz[j]) << 4));                   \
    }
    if (nk > 0) {
        V3_DMA(kt_begin, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        V3_TRACE(1);
#if V3_DMA_SPREAD
        if (nk > 1) { V3_DMA2(kt_begin + 1, 1, 0); }
#else
        if (nk > 1) { V3_DMA(kt_begin + 1, 1); }
#endif
        V3_FRAGS(0, lds3, 0);
#ifdef V3_ABLATE
        V3_FRAGS(1, lds3, 1);
#endif
    }
    for (int it = 0; it < nk; ++it) {
        const unsigned char* base = lds3 + (it & 1) * V3_STAGE;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int cur = s & 1, nxt = cur ^ 1;
#ifdef V3_ABLATE   // timing experiments only (results are wrong): bit 0 = no DMA, bit 1 = no fragment reads, bit 2 = no barrier
            if (s < 3) {
                if (!(V3_ABLATE & 2)) { V3_FRAGS(nxt, base, s + 1); }
            } else if (it + 1 < nk) {
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                if (!(V3_ABLATE & 4)) __builtin_amdgcn_s_barrier();
                if (!(V3_ABLATE & 1) && it + 2 < nk) { V3_DMA(kt_begin + it + 2, it & 1); }
                if (!(V3_ABLATE & 2)) { V3_FRAGS(nxt, lds3 + ((it + 1) & 1) * V3_STAGE, 0); }
            }
            if (V3_ABLATE & 2) {
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(af[nxt][i]));
#pragma unroll
                for (int j = 0; j < 2; ++j) asm volatile("" : "+v"(bfr[nxt][j]));
            }
#else
            if (s < 3) {
#if V3_DMA_SPREAD
                // the DMA of tile it+1 is spread over the k-steps (2 of this wave's 8 pieces each) instead of one burst of 64 KiB
                // into the LDS right when every wave starts reading fragments
                if (it + 1 < nk) { V3_DMA2(kt_begin + it + 1, (it + 1) & 1, 2 * s + 2); }
#endif
                V3_FRAGS(nxt, base, s + 1);
            } else if (it + 1 < nk) {
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // next tile landed; my reads of this stage are done
                __builtin_amdgcn_s_barrier();                                 // ... for every wave
#if V3_DMA_SPREAD
                if (it + 2 < nk) { V3_DMA2(kt_begin + it + 2, it & 1, 0); }
#else
                if (it + 2 < nk) { V3_DMA(kt_begin + it + 2, it & 1); }       // this stage is free again
#endif
                V3_FRAGS(nxt, lds3 + ((it + 1) & 1) * V3_STAGE, 0);
            }
#endif
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mfma32t<F16>(bfr[cur][j], af[cur][i], acc[i][j]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#undef V3_FRAGS
#else
    if (nk > 0) { V3_DMA(kt_begin, 0); }
#ifdef SED_GEMM_TRACE
    unsigned long long tw = 0, tb = 0;
#endif
    for (int it = 0; it < nk; ++it) {
        const int stage = it & 1;
#ifdef SED_GEMM_TRACE
        const unsigned long long c0 = __builtin_readcyclecounter();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long c1 = __builtin_readcyclecounter();
        __builtin_amdgcn_s_barrier();
        const unsigned long long c2 = __builtin_readcyclecounter();
        if (it > 0) { tw += c1 - c0; tb += c2 - c1; }
#else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // tile `it` landed (all waves); the other stage is no longer being read
#endif
        if (it == 0) V3_TRACE(1);
        if (it + 1 < nk) { V3_DMA(kt_begin + it + 1, stage ^ 1); }
        const unsigned char* base = lds3 + stage * V3_STAGE;
        s16x8_t af[2][4], bfr[2][2];
#pragma unroll
        for (int i = 0; i < 4; ++i) af[0][i] = *reinterpret_cast<const s16x8_t*>(base + aoff[i] + ((lg ^ aswz[i]) << 4));
#pragma unroll
        for (int j = 0; j < 2; ++j) bfr[0][j] = *reinterpret_cast<const s16x8_t*>(base + boff[j] + ((lg ^ bswz[j]) << 4));
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int cur = s & 1, nxt = cur ^ 1;
            if (s < 3) {
                const int ch = 2 * (s + 1) + lg;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    af[nxt][i] = *reinterpret_cast<const s16x8_t*>(base + aoff[i] + ((ch ^ aswz[i]) << 4));
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    bfr[nxt][j] = *reinterpret_cast<const s16x8_t*>(base + boff[j] + ((ch ^ bswz[j]) << 4));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mfma32t<F16>(bfr[cur][j], af[cur][i], acc[i][j]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#endif
    __builtin_amdgcn_s_barrier();  // every wave is done reading the operand stages: LDS becomes the per-wave C staging area
    V3_TRACE(2);
#ifdef SED_GEMM_TRACE
    if (lane == 0 && wave == sed_trace_wave && sed_trace_buf != nullptr) { sed_trace_buf[(size_t)blockIdx.x * 8 + 5] = tw; sed_trace_buf[(size_t)blockIdx.x * 8 + 6] = tb; }
#endif
    v3_epilogue<EPI, F16>(g, acc, cc, lds3 + wave * V3_WLDS, m0 + wm * 128, n0 + wn * 64, lane);
    V3_TRACE(3);
#ifdef SED_GEMM_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    V3_TRACE(4);
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// v7: 256 x 256 workgroup tile, FOUR waves (2 x 2) with 128 x 128 per wave (16 MFMA blocks = 256 accumulator registers in the
// AGPR half of the unified file, one wave per SIMD), K consumed in 32-deep tiles through FOUR 32 KiB LDS stages.
// Why: v3 keeps one K tile in flight, so an iteration can never be shorter than one DMA round trip (~2 us under load = its
// measured iteration time; halving the LDS fragment traffic alone -- the same 4-wave geometry at BK = 64 -- changed nothing).
// Here two tiles are in flight while two have landed: tile it is being multiplied, tile it+1 is already visible (its first
// fragments are pre-read under tile it's MFMAs, so no LDS latency is exposed after the barrier), tiles it+2 and it+3 fly.
// Counted `s_waitcnt vmcnt(8)` (8 DMA instructions per wave per tile) + raw s_barrier per tile.
// LDS rows are 64 B: 16-B chunk c of row r is stored at chunk c ^ ((r >> 2) & 3) (conflict-free for the ds_read_b128 lane groups).
// ---------------------------------------------------------------------------------------------------------------------
#define V7_BK 32
#define V7_STAGE (32 * 1024)
#define V7_NST 4
#define V7_LDS (V7_NST * V7_STAGE)
template <int EPI, bool F16>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_nt_v7_kernel(const GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds3[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int ntn = g.N / V3_T, ntm = (g.M + V3_T - 1) / V3_T, nwg = ntm * ntn;
    const int t = xcd_remap(blockIdx.x, nwg);
    const int group_size = 4 * ntn, gid = t / group_size, first_m = gid * 4;
    const int gm = (ntm - first_m) < 4 ? (ntm - first_m) : 4;
    const int tin = t - gid * group_size;
    const int m0 = (first_m + tin % gm) * V3_T, n0 = (tin / gm) * V3_T;
    const int ktiles = g.K / V7_BK;
    const int kt_begin = (int)(((long long)blockIdx.y * ktiles) / g.ksplit);
    const int kt_end = (int)(((long long)(blockIdx.y + 1) * ktiles) / g.ksplit);
    const int nk = kt_end - kt_begin;

    // DMA: 16 (A) + 16 (B) pieces of 16 rows x 64 B per stage; waves 0,1 fetch A pieces, waves 2,3 B pieces (8 each)
    const int prow = lane >> 2, pch = lane & 3;
    const bool isB = wave >= 2;
    const bf16_t* const opbase = isB ? g.B : g.A;
    unsigned src[8];  // element offsets (the launcher checks they fit 32 bits)
    const int dst0 = (isB ? 16384 : 0) + (wave & 1) * 8 * 1024;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = ((wave & 1) * 8 + i) * 16 + prow;
        const int cl = pch ^ ((row >> 2) & 3);
        int am = m0 + row;
        am = am < g.M ? am : g.M - 1;
        src[i] = (unsigned)((isB ? (n0 + row) * g.ldb : am * g.lda) + cl * 8 + kt_begin * V7_BK);
    }
#define V7_DMA(kt)                                                                                                        \
    _Pragma("unroll") for (int i = 0; i < 8; ++i)                                                                         \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(opbase + src[i] + (kt) * V7_BK), \
                                         (__attribute__((address_space(3))) void*)(lds3 + ((kt) & 3) * V7_STAGE + dst0 + i * 1024), \
                                         16, 0, 0);
    f32x16_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int lr = lane & 31, lg = lane >> 5;
    int aoff[4], boff[4], aswz[4], bswz[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int r = wm * 128 + i * 32 + lr; aoff[i] = r * 64; aswz[i] = (r >> 2) & 3; }
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int r = wn * 128 + j * 32 + lr; boff[j] = 16384 + r * 64; bswz[j] = (r >> 2) & 3; }
#define V7_FRAGS(dstA, dstB, kt, s)                                                                                       \
    {                                                                                                                     \
        const unsigned char* fb = lds3 + ((kt) & 3) * V7_STAGE;                                                           \
        const int ch = 2 * (s) + lg;                                                                                      \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) dstA[i] = *reinterpret_cast<const s16x8_t*>(fb + aoff[i] + ((ch ^ aswz[i]) << 4)); \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) dstB[j] = *reinterpret_cast<const s16x8_t*>(fb + boff[j] + ((ch ^ bswz[j]) << 4)); \
    }
#define V7_MFMA(fa, fb_)                                                                                                  \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                         \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) acc[i][j] = mfma32t<F16>(fb_[j], fa[i], acc[i][j]);

    s16x8_t a0[4], b0[4], a1[4], b1[4];
    if (nk > 0) {
        V7_DMA(0);
        if (nk > 1) { V7_DMA(1); }
        if (nk > 2) { V7_DMA(2); }
        if (nk > 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if (nk > 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        V7_FRAGS(a0, b0, 0, 0);
    }
    for (int it = 0; it < nk; ++it) {
        // tile it + 1 must be visible before this iteration pre-reads its first fragments; tile it + 2 may stay in flight
        if (it + 2 < nk) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // ... for every wave's pieces; also: nobody still reads stage (it - 1) & 3
        if (it + 3 < nk) { V7_DMA(it + 3); }
        V7_FRAGS(a1, b1, it, 1);
        __builtin_amdgcn_sched_barrier(0);
        V7_MFMA(a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        if (it + 1 < nk) { V7_FRAGS(a0, b0, it + 1, 0); }
        __builtin_amdgcn_sched_barrier(0);
        V7_MFMA(a1, b1);
        __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_barrier();  // every wave is done reading the operand stages: LDS becomes the per-wave C staging area
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int nb = n0 + wn * 128 + h * 64;
        V3Consts<EPI> cc;
        v3_load_consts<EPI>(cc, g, nb, lane);
        f32x16_t a2[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i) { a2[i][0] = acc[i][2 * h]; a2[i][1] = acc[i][2 * h + 1]; }
        v3_epilogue<EPI, F16>(g, a2, cc, lds3 + wave * V3_WLDS, m0 + wm * 128, nb, lane);
    }
#undef V7_DMA
#undef V7_FRAGS
#undef V7_MFMA
}

// ---------------------------------------------------------------------------------------------------------------------
// v5 = v3's main loop as a PERSISTENT kernel (one workgroup per CU walks its XCD's share of the tiles).  What it buys: the next
// tile's first K-tile DMA is issued right after the last MFMA of the current tile, BEFORE the epilogue, so the ~3 us DMA
// prologue that every short-K tile paid (K = 768: 22 us main loop) runs under the epilogue's LDS staging and stores.  The
// epilogue therefore may only use LDS outside stage 0: per-wave staging shrinks to 8.5 KiB (64 rows of 16-bit / 32 rows of
// fp32 per pass) placed on stage 1 and the 27 KiB above the stages.
// ---------------------------------------------------------------------------------------------------------------------
#define V5_WLDS 8704
#define V5_LDS (V3_STAGE + 8 * V5_WLDS)
template <bool F16, int MODE>
__device__ __forceinline__ void v5_stage16(unsigned char* wl, const f32x16_t (&acc)[4][2], const float (&bv)[2][4][4], int p, int lr,
                                           int lg) {
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int col = j * 32 + 8 * q + 4 * lg;
                const f32x16_t& a = p == 0 ? acc[ii][j] : acc[2 + ii][j];
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float x = a[4 * q + e] + bv[j][q][e];
                    v[e] = MODE == 1 ? gelu_fast(x) : x;
                }
                uint2 pk;
                pk.x = pack2<F16>(v[0], v[1]);
                pk.y = pack2<F16>(v[2], v[3]);
                *reinterpret_cast<uint2*>(wl + (ii * 32 + lr) * V3_RS16 + col * 2) = pk;
            }
}
__device__ __forceinline__ void v5_stage32(unsigned char* wl, const f32x16_t (&a2)[2], int lr, int lg) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int col = j * 32 + 8 * q + 4 * lg;
            *reinterpret_cast<float4*>(wl + lr * V3_RS32 + col * 4) =
                make_float4(a2[j][4 * q], a2[j][4 * q + 1], a2[j][4 * q + 2], a2[j][4 * q + 3]);
        }
}

template <int EPI, bool F16>
__device__ __forceinline__ void v5_epilogue(const GemmArgs& g, const f32x16_t (&acc)[4][2], const V3Consts<EPI>& cc,
                                            unsigned char* wl, int mb, int nb, int lane) {
    const int lr = lane & 31, lg = lane >> 5;
    if constexpr (EPI == EPI_BF16 || EPI == EPI_GELU) {
        const float (&bv)[2][4][4] = cc.bv;
#pragma unroll
        for (int pass = 0; pass < (EPI == EPI_GELU ? 2 : 1); ++pass) {
            bf16_t* out = (EPI == EPI_GELU && pass == 1) ? g.outH2 : g.outH;
            if (out == nullptr) continue;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                if (EPI == EPI_GELU && pass == 1) v5_stage16<F16, 1>(wl, acc, bv, p, lr, lg);
                else v5_stage16<F16, 0>(wl, acc, bv, p, lr, lg);
                __builtin_amdgcn_wave_barrier();
                uint4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const uint4*>(wl + (u * 8 + (lane >> 3)) * V3_RS16 + (lane & 7) * 16);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int m = mb + p * 64 + u * 8 + (lane >> 3);
                    if (m < g.M) v3_st<uint4>(out + (size_t)m * g.ldc + nb + (lane & 7) * 8, v[u]);
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        return;
    }
    if constexpr (EPI == EPI_QKV) {
        const int D = g.heads * 64;
        const int which = nb / D, h = (nb - which * D) >> 6;
        bf16_t* row_dst = which == 0 ? g.q : (which == 1 ? g.k : g.v);
        bf16_t* tr_dst = which == 0 ? g.qt : (which == 1 ? g.kt : g.vt);
        const int npass = (which == 0 && g.q2 != nullptr) ? 2 : 1;
        for (int pass = 0; pass < npass; ++pass) {
            const float* extra = nullptr;
            if (which == 0 && g.pu != nullptr) extra = (pass == 0 ? g.pu : g.pv) + h * 64;
            bf16_t* rd = pass == 0 ? row_dst : g.q2;
            bf16_t* td = pass == 0 ? tr_dst : g.q2t;
            float bv[2][4][4];
            v3_col_consts(bv, g.bias != nullptr ? g.bias + nb : nullptr, extra, lg);
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                v5_stage16<F16, 0>(wl, acc, bv, p, lr, lg);
                __builtin_amdgcn_wave_barrier();
                if (rd != nullptr) {  // (the row-major V is only needed by the backward: inference passes v = NULL)
#pragma unroll
                    for (int ub = 0; ub < 8; ub += 4) {  // batches of 4: this epilogue sits at the VGPR cap
                        uint4 v[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            v[u] = *reinterpret_cast<const uint4*>(wl + ((ub + u) * 8 + (lane >> 3)) * V3_RS16 + (lane & 7) * 16);
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int m = mb + p * 64 + (ub + u) * 8 + (lane >> 3);
                            if (m < g.M) {
                                const int bidx = m / g.seq, tt = m - bidx * g.seq;
                                v3_st<uint4>(rd + ((size_t)(bidx * g.heads + h) * g.seq + tt) * 64 + (lane & 7) * 8, v[u]);
                            }
                        }
                    }
                }
                if (td != nullptr && (g.seq & 1) == 0) {
                    // transposed copy [bh][d][seq_pad]: a lane owns a PAIR of consecutive tokens (same clip: seq is even, the
                    // pair starts on an even row) and one of two interleaved d columns -> 4-byte stores, 32 lanes = 128 B
                    const int pr = (lane & 31) * 2, dsel = lane >> 5;
                    const int m = mb + p * 64 + pr;
                    if (m < g.M) {
                        const int bidx = m / g.seq, t = m - bidx * g.seq;
                        bf16_t* base = td + (size_t)(bidx * g.heads + h) * 64 * g.seq_pad + t;
                        const unsigned short* s0 = reinterpret_cast<const unsigned short*>(wl + pr * V3_RS16);
                        const unsigned short* s1 = reinterpret_cast<const unsigned short*>(wl + (pr + 1) * V3_RS16);
#pragma unroll 8
                        for (int dd = 0; dd < 32; ++dd) {
                            const int d = dd * 2 + dsel;
                            *reinterpret_cast<unsigned*>(base + (size_t)d * g.seq_pad) = (unsigned)s0[d] | ((unsigned)s1[d] << 16);
                        }
                    }
                } else if (td != nullptr) {
                    const int m = mb + p * 64 + lane;
                    if (m < g.M) {
                        const int bidx = m / g.seq, t = m - bidx * g.seq;
                        bf16_t* base = td + (size_t)(bidx * g.heads + h) * 64 * g.seq_pad + t;
                        const unsigned short* src = reinterpret_cast<const unsigned short*>(wl + lane * V3_RS16);
#pragma unroll 8
                        for (int d = 0; d < 64; ++d) base[(size_t)d * g.seq_pad] = src[d];
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        return;
    }
    if constexpr (EPI == EPI_ATOMIC) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v5_stage32(wl, acc[i], lr, lg);
            __builtin_amdgcn_wave_barrier();
#pragma unroll 8
            for (int row = 0; row < 32; ++row) {
                const int m = mb + i * 32 + row;
                const float v = *reinterpret_cast<const float*>(wl + row * V3_RS32 + lane * 4);
                if (m < g.M) unsafeAtomicAdd(&g.outF[(size_t)m * g.ldc + nb + lane], v * g.alpha);
            }
            __builtin_amdgcn_wave_barrier();
        }
        return;
    }
    // fp32-staged epilogues, four passes of 32 rows; the side input of pass i + 1 is requested before pass i is stored
    const int c4 = lane & 15, n = nb + c4 * 4;
    const float4 b = make_float4(cc.bv[0][0][0], cc.bv[0][0][1], cc.bv[0][0][2], cc.bv[0][0][3]);
    V3Side<EPI> s0, s1;
    v3_side_load<EPI>(s0, g, mb, n, lane);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v5_stage32(wl, acc[i], lr, lg);
        __builtin_amdgcn_wave_barrier();
        if (i < 3) v3_side_load<EPI>((i & 1) ? s0 : s1, g, mb + 32 * (i + 1), n, lane);
        v3_store_batch<EPI, F16>(g, wl, (i & 1) ? s1 : s0, b, mb + 32 * i, 0, n, c4, lane);
        __builtin_amdgcn_wave_barrier();
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// v8 (experiment, SED_GEMM_V8=1): TWO workgroups per CU.  The 256 x 256 / 8-wave kernel above owns its CU alone (246 VGPRs x 2 waves
// per SIMD, 128 KB of LDS), so a tile's epilogue -- stores, GELU, residual reads -- never overlaps another tile's MFMAs; MFMA-pipe
// busy is 34 % on the K = 768 shapes.  Here a workgroup is 4 waves (one per SIMD) on a 128 x 256 tile with the same 128 x 64
// accumulator block per wave and the same staged epilogue; one 48 KB operand stage (A 128 rows | B 256 rows of 128 B) that doubles
// as the epilogue staging area (4 x 17 KB) keeps the footprint at 68 KB, so two workgroups are co-resident and cover each other's
// DMA waits, barriers and epilogues.
// ---------------------------------------------------------------------------------------------------------------------
#define V8_LDS (4 * V3_WLDS)
template <int EPI, bool F16>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_nt_v8_kernel(const GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds3[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wn = wave;
    const int ntn = g.N / V3_T, ntm = (g.M + 127) / 128, nwg = ntm * ntn;
    if (g.stagger > 0 && blockIdx.x >= 256 && blockIdx.x < 512) {
        // the second workgroup of every CU starts late, so that the two co-resident workgroups are in different phases (one in its
        // K loop while the other stores): launched together they would run in lock step and reach their epilogues together
        const unsigned long long t0 = wall_clock64();
        while (wall_clock64() - t0 < (unsigned long long)g.stagger) __builtin_amdgcn_s_sleep(32);
    }
    const int t = xcd_remap(blockIdx.x, nwg);
    const int group_size = 8 * ntn, gid = t / group_size, first_m = gid * 8;
    const int gm = (ntm - first_m) < 8 ? (ntm - first_m) : 8;
    const int tin = t - gid * group_size;
    const int m0 = (first_m + tin % gm) * 128, n0 = (tin / gm) * V3_T;
    const int nk = g.K / BK;
    // DMA: 16 (A) + 32 (B) pieces of 8 rows x 128 B; wave w owns pieces 12 w .. 12 w + 11
    const int prow = lane >> 3, pch = lane & 7;
    const bf16_t* src[12];
    int dst[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        const int p = wave * 12 + i;
        const bool isB = p >= 16;
        const int q = isB ? p - 16 : p;
        const int row = q * 8 + prow;
        const int cl = pch ^ ((row >> 1) & 7);
        int am = m0 + row;
        am = am < g.M ? am : g.M - 1;
        src[i] = isB ? g.B + (size_t)(n0 + row) * g.ldb + cl * 8 : g.A + (size_t)am * g.lda + cl * 8;
        dst[i] = (isB ? 16384 : 0) + q * 1024;
    }
#define V8_DMA(kt)                                                                                                        \
    _Pragma("unroll") for (int i = 0; i < 12; ++i)                                                                        \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + (size_t)(kt) * BK),     \
                                         (__attribute__((address_space(3))) void*)(lds3 + dst[i]), 16, 0, 0);
    f32x16_t acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int lr = lane & 31, lg = lane >> 5;
    int aoff[4], boff[2], aswz[4], bswz[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int r = i * 32 + lr; aoff[i] = r * 128; aswz[i] = (r >> 1) & 7; }
#pragma unroll
    for (int j = 0; j < 2; ++j) { const int r = wn * 64 + j * 32 + lr; boff[j] = 16384 + r * 128; bswz[j] = (r >> 1) & 7; }
    V3Consts<EPI> cc;
    v3_load_consts<EPI>(cc, g, n0 + wn * 64, lane);
    s16x8_t af[2][4], bfr[2][2];
#define V8_FRAGS(SET, KS)                                                                                                 \
    {                                                                                                                     \
        const int ch_ = 2 * (KS) + lg;                                                                                    \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                     \
            af[SET][i] = *reinterpret_cast<const s16x8_t*>(lds3 + aoff[i] + ((ch_ ^ aswz[i]) << 4));                      \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                     \
            bfr[SET][j] = *reinterpret_cast<const s16x8_t*>(lds3 + boff[j] + ((ch_ ^ bswz[j]) << 4));                     \
    }
    if (nk > 0) {
        V8_DMA(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        V8_FRAGS(0, 0);
    }
    for (int it = 0; it < nk; ++it) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int cur = s & 1, nxt = cur ^ 1;
            if (s < 3) {
                V8_FRAGS(nxt, s + 1);
            } else if (it + 1 < nk) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // my last fragments of this tile are in registers
                __builtin_amdgcn_s_barrier();                         // ... and everybody's: the stage can be overwritten
                V8_DMA(it + 1);                                       // lands under the 8 MFMAs below (and the other workgroup)
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mfma32t<F16>(bfr[cur][j], af[cur][i], acc[i][j]);
            __builtin_amdgcn_sched_barrier(0);
            if (s == 3 && it + 1 < nk) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                V8_FRAGS(nxt, 0);
            }
        }
    }
#undef V8_FRAGS
#undef V8_DMA
    __builtin_amdgcn_s_barrier();  // every wave is done reading the operand stage: LDS becomes the per-wave C staging area
    v3_epilogue<EPI, F16>(g, acc, cc, lds3 + wave * V3_WLDS, m0, n0 + wn * 64, lane);
}

template <int EPI, bool F16>
__global__ __launch_bounds__(512) void gemm_nt_v5_kernel(const GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds3[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int ntn = g.N / V3_T, ntm = (g.M + V3_T - 1) / V3_T, nwg = ntm * ntn;
    // workgroup w sits on XCD w % 8 (round-robin dispatch) and walks that XCD's contiguous share of the tile list, so the 32
    // workgroups of an XCD work on neighbouring tiles (4 tile-rows x 8 tile-columns) at any time
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
    const int tpx = (nwg + 7) >> 3;
    const int t_hi = (xcd + 1) * tpx < nwg ? (xcd + 1) * tpx : nwg;
    int t = xcd * tpx + slot;
    if (t >= t_hi) return;
    const int ktiles = g.K / BK;
    const int kt_begin = (int)(((long long)blockIdx.y * ktiles) / g.ksplit);
    const int kt_end = (int)(((long long)(blockIdx.y + 1) * ktiles) / g.ksplit);
    const int nk = kt_end - kt_begin;
    const int group_size = 4 * ntn;
    const int prow = lane >> 3, pch = lane & 7;
    const bool isB = wave >= 4;
    unsigned src[8];  // element offsets from the operand base (32-bit: the launcher checks the operands are < 2^31 elements)
    const bf16_t* const opbase = isB ? g.B : g.A;
    int dst[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) dst[i] = (isB ? 32768 : 0) + ((wave & 3) * 8 + i) * 1024;
    int m0, n0;
#define V5_TILE(tt)                                                                                                        \
    {                                                                                                                      \
        const int gid = (tt) / group_size, first_m = gid * 4;                                                              \
        const int gm = (ntm - first_m) < 4 ? (ntm - first_m) : 4;                                                          \
        const int tin = (tt) - gid * group_size;                                                                           \
        m0 = (first_m + tin % gm) * V3_T;                                                                                  \
        n0 = (tin / gm) * V3_T;                                                                                            \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                                    \
            const int row = ((wave & 3) * 8 + i) * 8 + prow;                                                               \
            const int cl = pch ^ ((row >> 1) & 7);                                                                         \
            int am = m0 + row;                                                                                             \
            am = am < g.M ? am : g.M - 1;                                                                                  \
            src[i] = (unsigned)((isB ? (n0 + row) * g.ldb : am * g.lda) + cl * 8 + kt_begin * BK);                         \
        }                                                                                                                  \
    }
#define V5_DMA(kt, stage)                                                                                                 \
    _Pragma("unroll") for (int i = 0; i < 8; ++i)                                                                         \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(opbase + src[i] + (kt) * BK),    \
                                         (__attribute__((address_space(3))) void*)(lds3 + (stage) * V3_STAGE + dst[i]),   \
                                         16, 0, 0);
    const int lr = lane & 31, lg = lane >> 5;
    int aoff[4], boff[2], aswz[4], bswz[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int r = wm * 128 + i * 32 + lr; aoff[i] = r * 128; aswz[i] = (r >> 1) & 7; }
#pragma unroll
    for (int j = 0; j < 2; ++j) { const int r = wn * 64 + j * 32 + lr; boff[j] = 32768 + r * 128; bswz[j] = (r >> 1) & 7; }

    V5_TILE(t);
    if (nk > 0) { V5_DMA(0, 0); }
    while (true) {
        f32x16_t acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        for (int it = 0; it < nk; ++it) {
            const int stage = it & 1;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();  // tile `it` landed (all waves); the other stage (and the C staging) is idle
            if (it + 1 < nk) { V5_DMA(it + 1, stage ^ 1); }
            const unsigned char* base = lds3 + stage * V3_STAGE;
            s16x8_t af[2][4], bfr[2][2];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[0][i] = *reinterpret_cast<const s16x8_t*>(base + aoff[i] + ((lg ^ aswz[i]) << 4));
#pragma unroll
            for (int j = 0; j < 2; ++j) bfr[0][j] = *reinterpret_cast<const s16x8_t*>(base + boff[j] + ((lg ^ bswz[j]) << 4));
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int cur = s & 1, nxt = cur ^ 1;
                if (s < 3) {
                    const int ch = 2 * (s + 1) + lg;
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        af[nxt][i] = *reinterpret_cast<const s16x8_t*>(base + aoff[i] + ((ch ^ aswz[i]) << 4));
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        bfr[nxt][j] = *reinterpret_cast<const s16x8_t*>(base + boff[j] + ((ch ^ bswz[j]) << 4));
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = mfma32t<F16>(bfr[cur][j], af[cur][i], acc[i][j]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        const int mb = m0 + wm * 128, nb = n0 + wn * 64;
        V3Consts<EPI> cc;  // bias values: requested once the operand fragments are dead, they land during the barrier + DMA issue
        v3_load_consts<EPI>(cc, g, nb, lane);
        __builtin_amdgcn_s_barrier();  // every wave is done reading the operand stages
        const int tn = t + nslots;
        const bool more = tn < t_hi;
        if (more) {
            V5_TILE(tn);
            if (nk > 0) { V5_DMA(0, 0); }  // next tile's first K tile flies under this tile's epilogue
        }
        v5_epilogue<EPI, F16>(g, acc, cc, lds3 + V3_STAGE + wave * V5_WLDS, mb, nb, lane);
        if (!more) break;
        t = tn;
    }
#undef V5_TILE
#undef V5_DMA
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight-gradient GEMM in TN form: dW[m, n] += sum_t dY[t, m] * X[t, n] with BOTH operands read in their natural row-major
// [token][feature] layout -- no transposed operand copies in HBM (the NT kernels need dY^T and X^T: 6.4 ms/step of transposes).
// The contraction index is the slow dimension of both tiles, so the MFMA fragments (8 consecutive k per lane) are columns of the
// LDS image: read with `ds_read_b64_tr_b16`.  Semantics measured on gfx950 (tools/ablate/trread.hip): the 16 lanes of a group
// supply 16 addresses of 8-byte chunks, taken as a [4 rows r = a>>2][4 chunks c = a&3] grid = a 4 x 16 halfword matrix M;
// lane i of the group receives column i: {M[0][i], M[1][i], M[2][i], M[3][i]}.  With rows = 4 consecutive tokens and columns =
// 16 consecutive features, two such reads give the 8-token fragment of one feature per lane.
// LDS stage = [64 tokens][256 features] per operand (512-B rows), filled by DMA; 64-B unit u of token row k is stored at unit
// u ^ (k & 3), so that the four rows of a transposing read hit four different bank quarters.
// Same 256 x 256 x 64 tiling / 8 waves (128 x 64 per wave) / two stages as v3; split-K over the tokens, atomic accumulate.
// X may be IEEE half (saved forward activation): converted to bf16 in registers (the gradient-side MFMA is bf16).
// ---------------------------------------------------------------------------------------------------------------------
struct TnArgs {
    const bf16_t* A;  // dY [T, lda]  bf16
    const bf16_t* B;  // X  [T, ldb]  bf16 or f16
    float* C;         // dW [M, ldc]  fp32 (accumulated)
    float* ws;        // optional split-K workspace [ksplit][M][N] fp32 (plain stores + a reduce pass instead of atomics)
    float* dbias;     // optional: dbias[m] += sum_t dY[t, m] (the bias gradient of the same linear), taken from the dY fragments
    int M, N, T, lda, ldb, ldc, ksplit, b_f16;
};
template <int OFF>
__device__ __forceinline__ unsigned long long lds_tr(unsigned addr) {
    unsigned long long v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
__device__ __forceinline__ s16x8_t tn_frag(unsigned long long lo, unsigned long long hi, bool cvt_f16) {
    unsigned w[4] = {(unsigned)lo, (unsigned)(lo >> 32), (unsigned)hi, (unsigned)(hi >> 32)};
    if (cvt_f16) {
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = h2x2_to_bf(w[e]);
    }
    s16x8_t f;
    __builtin_memcpy(&f, w, 16);
    return f;
}
template <bool BF16_B>
__global__ __launch_bounds__(512) void gemm_tn_dw_kernel(const TnArgs g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds3[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int ntn = g.N / V3_T, ntm = g.M / V3_T, nwg = ntm * ntn;
    const int t = xcd_remap(blockIdx.x, nwg);
    const int m0 = (t % ntm) * V3_T, n0 = (t / ntm) * V3_T;
    const int ktiles = g.T / BK;
    const int kt_begin = (int)(((long long)blockIdx.y * ktiles) / g.ksplit);
    const int kt_end = (int)(((long long)(blockIdx.y + 1) * ktiles) / g.ksplit);
    const int nk = kt_end - kt_begin;
    // DMA: per operand 32 pieces of 2 token rows x 512 B; waves 0-3 fetch dY pieces, 4-7 X pieces (8 each)
    const bool isB = wave >= 4;
    const bf16_t* src[8];
    int dst[8];
    {
        const int krow = lane >> 5, pc = lane & 31;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int piece = (wave & 3) * 8 + i, k = piece * 2 + krow;
            const int col = (((pc >> 2) ^ (k & 3)) * 4 + (pc & 3)) * 8;   // logical feature offset stored at physical chunk pc
            src[i] = (isB ? g.B + (size_t)k * g.ldb + n0 : g.A + (size_t)k * g.lda + m0) + col + (size_t)kt_begin * BK * (isB ? g.ldb : g.lda);
            dst[i] = (isB ? 32768 : 0) + piece * 1024;
        }
    }
    const size_t kstride = (size_t)BK * (isB ? g.ldb : g.lda);
#define TN_DMA(kt, stage)                                                                                                 \
    _Pragma("unroll") for (int i = 0; i < 8; ++i)                                                                         \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + (size_t)(kt) * kstride), \
                                         (__attribute__((address_space(3))) void*)(lds3 + (stage) * V3_STAGE + dst[i]),   \
                                         16, 0, 0);
    f32x16_t acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // transposing-read lane addresses: group G = lane >> 4 (k half g = G >> 1, 16-feature half = G & 1), a = lane & 15
    const int a = lane & 15, G = lane >> 4, kg = G >> 1;
    unsigned aaddr[4], baddr[2];
    const unsigned lbase = (unsigned)(size_t)lds3;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int col = wm * 128 + i * 32 + (G & 1) * 16 + 4 * (a & 3);
        aaddr[i] = lbase + (8 * kg + (a >> 2)) * 512 + (((col >> 5) ^ (a >> 2)) << 6) + (col & 31) * 2;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = wn * 64 + j * 32 + (G & 1) * 16 + 4 * (a & 3);
        baddr[j] = lbase + 32768 + (8 * kg + (a >> 2)) * 512 + (((col >> 5) ^ (a >> 2)) << 6) + (col & 31) * 2;
    }
    // bias gradient: the workgroups of the first N tile also sum their dY fragments over the tokens -- wave (wm, wn) owns the 32
    // features of fragment i = wn (a lane holds 8 tokens of one feature); VALU work that rides under the MFMAs
    const bool do_bias = g.dbias != nullptr && n0 == 0;
    float colacc = 0.f;
    if (nk > 0) { TN_DMA(0, 0); }
    for (int it = 0; it < nk; ++it) {
        const int stage = it & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (it + 1 < nk) { TN_DMA(it + 1, stage ^ 1); }
        const unsigned so = stage * V3_STAGE;
#define TN_READS(S, al, ah, bl, bh)                                                                                       \
        {                                                                                                                  \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) { al[i] = lds_tr<(S) * 8192>(aaddr[i] + so); ah[i] = lds_tr<(S) * 8192 + 2048>(aaddr[i] + so); } \
            _Pragma("unroll") for (int j = 0; j < 2; ++j) { bl[j] = lds_tr<(S) * 8192>(baddr[j] + so); bh[j] = lds_tr<(S) * 8192 + 2048>(baddr[j] + so); } \
        }
        /* the compiler does not know the asm results are still in flight: every consumer is tied to the counted wait */       \
#define TN_WAIT(CNT, al, ah, bl, bh)                                                                                       \
        asm volatile("s_waitcnt lgkmcnt(" #CNT ")"                                                                         \
                     : "+v"(al[0]), "+v"(al[1]), "+v"(al[2]), "+v"(al[3]), "+v"(ah[0]), "+v"(ah[1]), "+v"(ah[2]), "+v"(ah[3]), \
                       "+v"(bl[0]), "+v"(bl[1]), "+v"(bh[0]), "+v"(bh[1])                                                  \
                     :: "memory");
#define TN_MFMA(al, ah, bl, bh)                                                                                            \
        {                                                                                                                  \
            s16x8_t af[4], bf_[2];                                                                                         \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) af[i] = tn_frag(al[i], ah[i], false);                            \
            _Pragma("unroll") for (int j = 0; j < 2; ++j) bf_[j] = tn_frag(bl[j], bh[j], !BF16_B);                         \
            _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                  \
                _Pragma("unroll") for (int j = 0; j < 2; ++j) acc[i][j] = mfma32t<false>(bf_[j], af[i], acc[i][j]);        \
            if (do_bias) {                                                                                                 \
                _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                              \
                    if (wn == i) {                                                                                         \
                        unsigned w4[4];                                                                                    \
                        __builtin_memcpy(w4, &af[i], 16);                                                                  \
                        _Pragma("unroll") for (int e = 0; e < 4; ++e)                                                      \
                            colacc += __uint_as_float(w4[e] << 16) + __uint_as_float(w4[e] & 0xffff0000u);                 \
                    }                                                                                                      \
            }                                                                                                              \
        }
        // fragment reads run one k-step ahead of the MFMAs (12 transposing reads per k-step; lgkmcnt counts them in order)
        unsigned long long pal[4], pah[4], pbl[2], pbh[2], qal[4], qah[4], qbl[2], qbh[2];
        TN_READS(0, pal, pah, pbl, pbh)
        TN_READS(1, qal, qah, qbl, qbh)
        TN_WAIT(12, pal, pah, pbl, pbh)
        TN_MFMA(pal, pah, pbl, pbh)
        TN_READS(2, pal, pah, pbl, pbh)
        TN_WAIT(12, qal, qah, qbl, qbh)
        TN_MFMA(qal, qah, qbl, qbh)
        TN_READS(3, qal, qah, qbl, qbh)
        TN_WAIT(12, pal, pah, pbl, pbh)
        TN_MFMA(pal, pah, pbl, pbh)
        TN_WAIT(0, qal, qah, qbl, qbh)
        TN_MFMA(qal, qah, qbl, qbh)
#undef TN_READS
#undef TN_WAIT
#undef TN_MFMA
    }
    if (do_bias) {
        colacc += __shfl_xor(colacc, 32, 64);      // the two token halves of the fragment
        if (lane < 32) unsafeAtomicAdd(&g.dbias[m0 + wm * 128 + wn * 32 + lane], colacc);
    }
    __builtin_amdgcn_s_barrier();
    // Split-K partial sums.  With a workspace: plain coalesced stores of the tile into ws[split] (a reduce pass adds the splits
    // into dW) -- one workgroup per CU cannot hide 64 Ki device-scope atomics behind anything (measured ~150-200 us per GEMM,
    // as long as the whole K loop).  Without: atomics straight into dW.
    unsigned char* wl = lds3 + wave * V3_WLDS;
    const int lr = lane & 31, lg = lane >> 5;
    const int mb = m0 + wm * 128, nb = n0 + wn * 64;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        v3_stage32(wl, acc, 2 * pass, lr, lg);
        __builtin_amdgcn_wave_barrier();
        if (g.ws != nullptr) {
            float* dstp = g.ws + ((size_t)blockIdx.y * g.M + mb + pass * 64) * g.N + nb + (lane & 15) * 4;
#pragma unroll
            for (int rb = 0; rb < 16; rb += 8) {
                float4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(wl + ((rb + u) * 4 + (lane >> 4)) * V3_RS32 + (lane & 15) * 16);
#pragma unroll
                for (int u = 0; u < 8; ++u) v3_st<float4>(dstp + (size_t)((rb + u) * 4 + (lane >> 4)) * g.N, v[u]);
            }
        } else {
#pragma unroll 8
            for (int row = 0; row < 64; ++row) {
                const float v = *reinterpret_cast<const float*>(wl + row * V3_RS32 + lane * 4);
                unsafeAtomicAdd(&g.C[(size_t)(mb + pass * 64 + row) * g.ldc + nb + lane], v);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
#undef TN_DMA
}

// dW[m, n] += sum_s ws[s][m][n]
__global__ __launch_bounds__(256) void tn_reduce_kernel(const float* __restrict__ ws, int ks, int M, int N, float* __restrict__ C,
                                                        int ldc) {
    const size_t total4 = (size_t)M * N / 4, plane4 = total4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
        float4 a = reinterpret_cast<const float4*>(ws)[i];
        for (int s2 = 1; s2 < ks; ++s2) {
            const float4 b = reinterpret_cast<const float4*>(ws)[(size_t)s2 * plane4 + i];
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        const size_t e = i * 4, m = e / N, n = e - m * N;
        float4* dst = reinterpret_cast<float4*>(C + m * ldc + n);
        float4 c = *dst;
        c.x += a.x; c.y += a.y; c.z += a.z; c.w += a.w;
        *dst = c;
    }
}

extern "C" int sed_gemm_dw_tn(const void* dY, const void* X, int x_f16, int T, int M, int N, int ldy, int ldx, float* dW,
                              int ldc, float* dbias, float* workspace, int64_t workspace_bytes, hipStream_t stream) {
    (void)hipGetLastError();
    if (T <= 0 || (T % BK) || (M % V3_T) || (N % V3_T) || (ldy % 8) || (ldx % 8) || (ldc % 4) || M <= 0 || N <= 0) return SED_ERR_ARG;
    TnArgs g;
    g.A = (const bf16_t*)dY; g.B = (const bf16_t*)X; g.C = dW; g.dbias = dbias;
    g.M = M; g.N = N; g.T = T; g.lda = ldy; g.ldb = ldx; g.ldc = ldc; g.b_f16 = x_f16;
    const int tiles = (M / V3_T) * (N / V3_T), ktiles = T / BK;
    // one workgroup per CU and ONE round: tiles * ks <= 256 (rounding the split count up instead costs a second, nearly empty
    // round -- 36 tiles x 8 splits = 288 workgroups took twice the time of 36 x 7)
    int ks = 256 / tiles;
    if (ks > ktiles / 16) ks = ktiles / 16;
    if (ks < 1) ks = 1;
    g.ksplit = ks;
    g.ws = (workspace != nullptr && workspace_bytes >= (int64_t)ks * M * N * 4) ? workspace : nullptr;
    static bool attr[2] = {false, false};
    dim3 grid(tiles, ks);
    if (x_f16) {
        if (!attr[1]) { (void)hipFuncSetAttribute((const void*)gemm_tn_dw_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, V3_LDS); attr[1] = true; }
        hipLaunchKernelGGL((gemm_tn_dw_kernel<false>), grid, dim3(512), V3_LDS, stream, g);
    } else {
        if (!attr[0]) { (void)hipFuncSetAttribute((const void*)gemm_tn_dw_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, V3_LDS); attr[0] = true; }
        hipLaunchKernelGGL((gemm_tn_dw_kernel<true>), grid, dim3(512), V3_LDS, stream, g);
    }
    if (g.ws != nullptr) {
        int blocks = (int)(((size_t)M * N / 4 + 255) / 256);
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(tn_reduce_kernel, dim3(blocks), dim3(256), 0, stream, g.ws, ks, M, N, dW, ldc);
    }
    return sed_check_launch();
}

template <int EPI>
static int launch_gemm(const GemmArgs& g, int f16, hipStream_t s) {
    if (g.M <= 0 || g.N % TILE != 0 || g.K % BK != 0 || g.ksplit < 1) return SED_ERR_ARG;
    if ((g.lda % 8) || (g.ldb % 8) || (g.ldc % 4)) return SED_ERR_ARG;
    dim3 grid(cdiv(g.M, TILE) * (g.N / TILE), g.ksplit);
    static const int glds = []() { const char* e = getenv("SED_GEMM_GLDS"); return (e == nullptr || e[0] != '0') ? 1 : 0; }();
    static const int v3 = []() { const char* e = getenv("SED_GEMM_V3"); return (e == nullptr || e[0] != '0') ? 1 : 0; }();
    // split-K dW through the 256^2 kernel measured slower in the train step (178 vs 171 ms): one workgroup per CU leaves the
    // long atomic epilogue uncovered, whereas the 128^2 kernel keeps a second workgroup's MFMAs running under it.  Opt-in.
    static const int v3dw = []() { const char* e = getenv("SED_GEMM_V3_DW"); return (e != nullptr && e[0] == '1') ? 1 : 0; }();
    const bool v3_ok = EPI == EPI_ATOMIC ? (v3dw && g.M >= 512 && g.K >= 32 * BK) : (g.M >= 1024 && g.ksplit == 1);
    if (v3 && g.N % V3_T == 0 && v3_ok) {
        int ks3 = 1;
        if (EPI == EPI_ATOMIC) {  // split-K weight gradients: one workgroup per CU, at least 16 K tiles per split
            const int tiles3 = cdiv(g.M, V3_T) * (g.N / V3_T), ktiles = g.K / BK;
            ks3 = 256 / tiles3;  // one round of workgroups (see sed_gemm_dw_tn)
            if (ks3 > ktiles / 16) ks3 = ktiles / 16;
            if (ks3 < 1) ks3 = 1;
        }
        dim3 grid3(cdiv(g.M, V3_T) * (g.N / V3_T), ks3);
        static const float stagger_us = []() { const char* e = getenv("SED_GEMM_STAGGER_US"); return e ? (float)atof(e) : 0.f; }();
        GemmArgs gs = g;
        gs.ksplit = ks3;
        gs.stagger = (int)(stagger_us * 100.f / 8.f);
        const GemmArgs& g = gs;
        // persistent variant: measured no better than the per-tile launch (fc1 0.303 vs 0.290 ms, step 151.1 vs 149.2 ms) -- the
        // hidden 3 us prologue is paid back by the smaller staging passes and the loss of dynamic tile balancing.  Opt-in.
        static const int v5 = []() { const char* e = getenv("SED_GEMM_V5"); return (e != nullptr && e[0] == '1') ? 1 : 0; }();
        const bool fits32 = (long long)g.M * g.lda < (1LL << 31) && (long long)g.N * g.ldb < (1LL << 31);
        // 4-wave / 128x128-per-wave / four 32-deep stages: measured slower than v3 (1009 vs 1089 TFLOP/s at 8192^3, fc1 0.344 vs
        // 0.301 ms): neither fewer LDS fragment reads nor two K tiles in flight move the ~2 us per 256x256x64 step.  Opt-in.
        static const int v8 = []() { const char* e = getenv("SED_GEMM_V8"); return (e != nullptr && e[0] == '1') ? 1 : 0; }();
        if (v8 && EPI != EPI_ATOMIC && (long long)g.M * g.lda < (1LL << 31) && (long long)g.N * g.ldb < (1LL << 31)) {
            dim3 grid8(cdiv(g.M, 128) * (g.N / V3_T));
            static bool attr8[2] = {false, false};
            static const float st8 = []() { const char* e = getenv("SED_V8_STAGGER_US"); return e ? (float)atof(e) : 0.f; }();
            GemmArgs g8 = g;
            g8.stagger = (int)(st8 * 100.f);     // wall-clock ticks of 10 ns
            const GemmArgs& g = g8;
            if (f16) {
                if (!attr8[1]) { (void)hipFuncSetAttribute((const void*)gemm_nt_v8_kernel<EPI, true>, hipFuncAttributeMaxDynamicSharedMemorySize, V8_LDS); attr8[1] = true; }
                hipLaunchKernelGGL((gemm_nt_v8_kernel<EPI, true>), grid8, dim3(256), V8_LDS, s, g);
            } else {
                if (!attr8[0]) { (void)hipFuncSetAttribute((const void*)gemm_nt_v8_kernel<EPI, false>, hipFuncAttributeMaxDynamicSharedMemorySize, V8_LDS); attr8[0] = true; }
                hipLaunchKernelGGL((gemm_nt_v8_kernel<EPI, false>), grid8, dim3(256), V8_LDS, s, g);
            }
            return sed_check_launch();
        }
        static const int v7 = []() { const char* e = getenv("SED_GEMM_V7"); return (e != nullptr && e[0] == '1') ? 1 : 0; }();
        if (v7 && fits32) {
            static bool attr7[2] = {false, false};
            if (f16) {
                if (!attr7[1]) { (void)hipFuncSetAttribute((const void*)gemm_nt_v7_kernel<EPI, true>, hipFuncAttributeMaxDynamicSharedMemorySize, V7_LDS); attr7[1] = true; }
                hipLaunchKernelGGL((gemm_nt_v7_kernel<EPI, true>), grid3, dim3(256), V7_LDS, s, g);
            } else {
                if (!attr7[0]) { (void)hipFuncSetAttribute((const void*)gemm_nt_v7_kernel<EPI, false>, hipFuncAttributeMaxDynamicSharedMemorySize, V7_LDS); attr7[0] = true; }
                hipLaunchKernelGGL((gemm_nt_v7_kernel<EPI, false>), grid3, dim3(256), V7_LDS, s, g);
            }
            return sed_check_launch();
        }
        if (v5 && fits32 && !g.bwd_bf16) {
            static const int ncu = []() {
                int dev = 0, n = 0;
                (void)hipGetDevice(&dev);
                (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
                return n >= 8 ? (n & ~7) : 256;
            }();
            const int nwg3 = (int)grid3.x;
            dim3 grid5(nwg3 < ncu ? ((nwg3 + 7) & ~7) : ncu, ks3);
            static bool attr5[2] = {false, false};
            if (f16) {
                if (!attr5[1]) { (void)hipFuncSetAttribute((const void*)gemm_nt_v5_kernel<EPI, true>, hipFuncAttributeMaxDynamicSharedMemorySize, V5_LDS); attr5[1] = true; }
                hipLaunchKernelGGL((gemm_nt_v5_kernel<EPI, true>), grid5, dim3(512), V5_LDS, s, g);
            } else {
                if (!attr5[0]) { (void)hipFuncSetAttribute((const void*)gemm_nt_v5_kernel<EPI, false>, hipFuncAttributeMaxDynamicSharedMemorySize, V5_LDS); attr5[0] = true; }
                hipLaunchKernelGGL((gemm_nt_v5_kernel<EPI, false>), grid5, dim3(512), V5_LDS, s, g);
            }
            return sed_check_launch();
        }
        static bool attr3[2] = {false, false};
        if (f16) {
            if (!attr3[1]) { (void)hipFuncSetAttribute((const void*)gemm_nt_v3_kernel<EPI, true>, hipFuncAttributeMaxDynamicSharedMemorySize, V3_LDS); attr3[1] = true; }
            hipLaunchKernelGGL((gemm_nt_v3_kernel<EPI, true>), grid3, dim3(512), V3_LDS, s, g);
        } else {
            if (!attr3[0]) { (void)hipFuncSetAttribute((const void*)gemm_nt_v3_kernel<EPI, false>, hipFuncAttributeMaxDynamicSharedMemorySize, V3_LDS); attr3[0] = true; }
            hipLaunchKernelGGL((gemm_nt_v3_kernel<EPI, false>), grid3, dim3(512), V3_LDS, s, g);
        }
        return sed_check_launch();
    }
    if (glds) {
        if (f16) hipLaunchKernelGGL((gemm_nt_kernel<EPI, true, true>), grid, dim3(256), 0, s, g);
        else hipLaunchKernelGGL((gemm_nt_kernel<EPI, false, true>), grid, dim3(256), 0, s, g);
    } else {
        if (f16) hipLaunchKernelGGL((gemm_nt_kernel<EPI, true, false>), grid, dim3(256), 0, s, g);
        else hipLaunchKernelGGL((gemm_nt_kernel<EPI, false, false>), grid, dim3(256), 0, s, g);
    }
    return sed_check_launch();
}

static int gemm_nt_impl(const void* A, const void* B, int M, int N, int K, int lda, int ldb, int epi, const float* bias,
                        const float* resF, float* outF, void* outH, void* outH2, const void* auxH, int ldc, float alpha,
                        int ksplit, int f16, int ncols, hipStream_t stream) {
    (void)hipGetLastError();
    GemmArgs g = {};
    g.A = (const bf16_t*)A; g.B = (const bf16_t*)B;
    g.ncols = ncols;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ksplit = ksplit > 0 ? ksplit : 1;
    g.alpha = alpha; g.bias = bias; g.resF = resF; g.outF = outF; g.outH = (bf16_t*)outH; g.outH2 = (bf16_t*)outH2;
    g.auxH = (const bf16_t*)auxH;
    g.bwd_bf16 = (f16 & 2) ? 1 : 0;
    f16 &= 1;
    if (g.bwd_bf16 && !f16) return SED_ERR_ARG;
    switch (epi) {
        case EPI_F32: return launch_gemm<EPI_F32>(g, f16, stream);
        case EPI_F32_RESID: return launch_gemm<EPI_F32_RESID>(g, f16, stream);
        case EPI_BF16: return launch_gemm<EPI_BF16>(g, f16, stream);
        case EPI_GELU: return launch_gemm<EPI_GELU>(g, f16, stream);
        case EPI_DGELU: return launch_gemm<EPI_DGELU>(g, f16, stream);
        case EPI_ATOMIC: return launch_gemm<EPI_ATOMIC>(g, f16, stream);
        case EPI_F32_BF16: return launch_gemm<EPI_F32_BF16>(g, f16, stream);
        case EPI_GELU32: return launch_gemm<EPI_GELU32>(g, f16, stream);
        default: return SED_ERR_ARG;
    }
}
extern "C" int sed_gemm_nt(const void* A, const void* B, int M, int N, int K, int lda, int ldb, int epi,
                           const float* bias, const float* resF, float* outF, void* outH, void* outH2,
                           const void* auxH, int ldc, float alpha, int ksplit, int f16, hipStream_t stream) {
    return gemm_nt_impl(A, B, M, N, K, lda, ldb, epi, bias, resF, outF, outH, outH2, auxH, ldc, alpha, ksplit, f16, N, stream);
}
// same GEMM with a narrow result: the operands are padded to N (multiple of 128) but only the first ncols (multiple of 4) output
// columns exist in memory (row stride ldc >= ncols); bias / residual / outputs are indexed like the narrow matrix.  128^2 kernel only.
extern "C" int sed_gemm_nt_cols(const void* A, const void* B, int M, int N, int K, int lda, int ldb, int epi,
                                const float* bias, const float* resF, float* outF, void* outH, void* outH2,
                                const void* auxH, int ldc, float alpha, int f16, int ncols, hipStream_t stream) {
    if (ncols <= 0 || ncols > N || (ncols % 4) || N % 256 == 0 || epi == EPI_ATOMIC) return SED_ERR_ARG;
    return gemm_nt_impl(A, B, M, N, K, lda, ldb, epi, bias, resF, outF, outH, outH2, auxH, ldc, alpha, 1, f16, ncols, stream);
}

extern "C" int sed_gemm_qkv(const void* A, const void* W, const float* bias, int M, int K, int heads, int seq,
                            int seq_pad, void* q, void* k, void* v, void* qt, void* kt, void* vt, void* q2,
                            void* q2t, const float* pos_u, const float* pos_v, int f16, hipStream_t stream) {
    (void)hipGetLastError();
    GemmArgs g = {};
    g.A = (const bf16_t*)A; g.B = (const bf16_t*)W;
    g.M = M; g.N = 3 * heads * 64; g.K = K; g.lda = K; g.ldb = K; g.ldc = g.N; g.ksplit = 1; g.alpha = 1.f;
    g.ncols = g.N;
    g.bias = bias;
    g.q = (bf16_t*)q; g.k = (bf16_t*)k; g.v = (bf16_t*)v; g.qt = (bf16_t*)qt; g.kt = (bf16_t*)kt; g.vt = (bf16_t*)vt;
    g.q2 = (bf16_t*)q2; g.q2t = (bf16_t*)q2t; g.pu = pos_u; g.pv = pos_v;
    g.seq = seq; g.seq_pad = seq_pad; g.heads = heads;
    if (seq <= 0 || (seq_pad % 64) || M % seq) return SED_ERR_ARG;
    g.bwd_bf16 = (f16 & 2) ? 1 : 0;
    f16 &= 1;
    if (g.bwd_bf16 && !f16) return SED_ERR_ARG;
    return launch_gemm<EPI_QKV>(g, f16, stream);
}

// ---------------------------------------------------------------------------------------------------
// Layout helpers around the GEMM: casts, transposes (with zero padding of the reduction dim), column sums.
// ---------------------------------------------------------------------------------------------------
// in fp32 [R, C] -> out bf16 [R, C]  (grid-stride, float4 in / 8 B out)
template <bool F16>
__global__ void cast_f32_bf16_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, size_t n4) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n4; i += stride) {
        const float4 v = reinterpret_cast<const float4*>(in)[i];
        uint2 p;
        p.x = pack2<F16>(v.x, v.y);
        p.y = pack2<F16>(v.z, v.w);
        reinterpret_cast<uint2*>(out)[i] = p;
    }
}

extern "C" int sed_cast_f32_bf16(const float* in, void* out, int64_t n, int f16, hipStream_t stream) {
    (void)hipGetLastError();
    if (n % 4) return SED_ERR_ARG;
    const size_t n4 = n / 4;
    int blocks = (int)((n4 + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    if (f16) hipLaunchKernelGGL(cast_f32_bf16_kernel<true>, dim3(blocks), dim3(256), 0, stream, in, (bf16_t*)out, n4);
    else hipLaunchKernelGGL(cast_f32_bf16_kernel<false>, dim3(blocks), dim3(256), 0, stream, in, (bf16_t*)out, n4);
    return sed_check_launch();
}

// Transpose [R, C] (fp32 or bf16 in) -> bf16 out^T [C, Rpad] (rows R..Rpad-1 written as zeros), optionally also
// the straight bf16 copy [R, C] and the fp32 column sums (atomicAdd into colsum[C]) -- one pass over the input.
// 64x64 tiles through LDS; 256 threads.
// kinds: 0 = bf16, 1 = f32 (input only), 2 = f16
__device__ __forceinline__ float load_kind(const void* p, size_t i, int kind) {
    if (kind == 1) return ((const float*)p)[i];
    const bf16_t h = ((const bf16_t*)p)[i];
    return kind == 2 ? h2f(h) : bf2f(h);
}
__device__ __forceinline__ bf16_t store_kind(float v, int kind) { return kind == 2 ? f2h(v) : f2bf(v); }

// 64 x 64 tiles through an fp32 LDS tile; 16-byte global loads along input rows and 16-byte stores along output rows
// (requires C % 64 == 0, which holds for every feature width on this path; R is arbitrary, rows R..Rpad-1 become zeros).
__global__ __launch_bounds__(256) void transpose_kernel(const void* __restrict__ in, int in_kind, int R, int C, int ldin,
                                                        bf16_t* __restrict__ outT, int Rpad, int outT_kind,
                                                        bf16_t* __restrict__ outS, int outS_kind,
                                                        float* __restrict__ colsum) {
    __shared__ float tile[64][65];
    const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
    const int t = threadIdx.x;
    {
        const int row = t >> 2, cg = (t & 3) * 16, r = r0 + row;
        float v[16];
        if (r < R) {
            if (in_kind == 1) {
                const float4* src = reinterpret_cast<const float4*>((const float*)in + (size_t)r * ldin + c0 + cg);
#pragma unroll
                for (int k = 0; k < 4; ++k) { const float4 f = src[k]; v[4 * k] = f.x; v[4 * k + 1] = f.y; v[4 * k + 2] = f.z; v[4 * k + 3] = f.w; }
            } else {
                const uint4* src = reinterpret_cast<const uint4*>((const bf16_t*)in + (size_t)r * ldin + c0 + cg);
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const uint4 u = src[k];
                    const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const bf16_t lo = (bf16_t)(w[e] & 0xFFFF), hi = (bf16_t)(w[e] >> 16);
                        v[8 * k + 2 * e] = in_kind == 2 ? h2f(lo) : bf2f(lo);
                        v[8 * k + 2 * e + 1] = in_kind == 2 ? h2f(hi) : bf2f(hi);
                    }
                }
            }
            if (outS != nullptr) {
                unsigned pk[8];
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    pk[e] = (unsigned)store_kind(v[2 * e], outS_kind) | ((unsigned)store_kind(v[2 * e + 1], outS_kind) << 16);
                uint4* dst = reinterpret_cast<uint4*>(outS + (size_t)r * C + c0 + cg);
                dst[0] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
                dst[1] = make_uint4(pk[4], pk[5], pk[6], pk[7]);
            }
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] = 0.f;
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) tile[row][cg + e] = v[e];
    }
    __syncthreads();
    if (colsum != nullptr && t < 64) {
        float s = 0.f;
#pragma unroll 8
        for (int i = 0; i < 64; ++i) s += tile[i][t];
        unsafeAtomicAdd(&colsum[c0 + t], s);
    }
    if (outT == nullptr) return;
    {
        const int oc = t >> 2, seg = (t & 3) * 16;
        if (r0 + seg < Rpad) {
            unsigned pk[8];
#pragma unroll
            for (int e = 0; e < 8; ++e)
                pk[e] = (unsigned)store_kind(tile[seg + 2 * e][oc], outT_kind) |
                        ((unsigned)store_kind(tile[seg + 2 * e + 1][oc], outT_kind) << 16);
            uint4* dst = reinterpret_cast<uint4*>(outT + (size_t)(c0 + oc) * Rpad + r0 + seg);
            dst[0] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
            dst[1] = make_uint4(pk[4], pk[5], pk[6], pk[7]);
        }
    }
}

extern "C" int sed_transpose_to_bf16(const void* in, int in_kind, int R, int C, int ldin, void* outT, int Rpad,
                                     int outT_kind, void* outS, int outS_kind, float* colsum, hipStream_t stream) {
    (void)hipGetLastError();
    if (Rpad < R || in_kind < 0 || in_kind > 2 || (C % 64) || (Rpad % 16) || (ldin % 8)) return SED_ERR_ARG;
    dim3 grid(cdiv(C, 64), cdiv(Rpad, 64));
    hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, stream, in, in_kind, R, C, ldin, (bf16_t*)outT, Rpad,
                       outT_kind, (bf16_t*)outS, outS_kind, colsum);
    return sed_check_launch();
}

// All weight images of a model in ONE launch.  Per step the engine rebuilds, from the fp32 masters, the straight 16-bit image of
// every GEMM weight, its transposed bf16 image (backward operand) and, for the split-precision layers, the [hi | hi | lo] f16
// image: ~130 launches of 5-20 us for student + teacher.  desc: n_desc x 8 int64 {in fp32 [R, C], outT bf16 [C, R] or 0,
// outS [R, C] or 0, split f16 [R, 3C] or 0, R, C, outS kind (0 bf16 / 2 f16), first tile index}; one workgroup per 64 x 64 tile.
__global__ __launch_bounds__(256) void weight_images_kernel(const long long* __restrict__ desc, int n_desc) {
    __shared__ float tile[64][65];
    int lo = 0, hi = n_desc - 1;
    while (lo < hi) {   // last descriptor whose first tile <= blockIdx.x
        const int mid = (lo + hi + 1) >> 1;
        if (desc[(size_t)mid * 8 + 7] <= (long long)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const long long* d = desc + (size_t)lo * 8;
    const float* in = reinterpret_cast<const float*>(d[0]);
    bf16_t* outT = reinterpret_cast<bf16_t*>(d[1]);
    bf16_t* outS = reinterpret_cast<bf16_t*>(d[2]);
    bf16_t* outP = reinterpret_cast<bf16_t*>(d[3]);
    const int R = (int)d[4], C = (int)d[5], skind = (int)d[6];
    const int tl = blockIdx.x - (int)d[7], tx = C / 64;
    const int c0 = (tl % tx) * 64, r0 = (tl / tx) * 64;
    const int t = threadIdx.x;
    {
        const int row = t >> 2, cg = (t & 3) * 16, r = r0 + row;
        float v[16];
        if (r < R) {
            const float4* src = reinterpret_cast<const float4*>(in + (size_t)r * C + c0 + cg);
#pragma unroll
            for (int k = 0; k < 4; ++k) { const float4 f = src[k]; v[4 * k] = f.x; v[4 * k + 1] = f.y; v[4 * k + 2] = f.z; v[4 * k + 3] = f.w; }
            if (outS != nullptr) {
                unsigned pk[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) pk[e] = (unsigned)store_kind(v[2 * e], skind) | ((unsigned)store_kind(v[2 * e + 1], skind) << 16);
                uint4* dst = reinterpret_cast<uint4*>(outS + (size_t)r * C + c0 + cg);
                dst[0] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
                dst[1] = make_uint4(pk[4], pk[5], pk[6], pk[7]);
            }
            if (outP != nullptr) {
                unsigned ph[8], pl[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const bf16_t h0 = f2h(v[2 * e]), h1 = f2h(v[2 * e + 1]);
                    ph[e] = (unsigned)h0 | ((unsigned)h1 << 16);
                    pl[e] = (unsigned)f2h(v[2 * e] - h2f(h0)) | ((unsigned)f2h(v[2 * e + 1] - h2f(h1)) << 16);
                }
                bf16_t* prow = outP + (size_t)r * 3 * C + c0 + cg;
                uint4* d0 = reinterpret_cast<uint4*>(prow);
                uint4* d1 = reinterpret_cast<uint4*>(prow + C);
                uint4* d2 = reinterpret_cast<uint4*>(prow + 2 * C);
                d0[0] = make_uint4(ph[0], ph[1], ph[2], ph[3]); d0[1] = make_uint4(ph[4], ph[5], ph[6], ph[7]);
                d1[0] = make_uint4(ph[0], ph[1], ph[2], ph[3]); d1[1] = make_uint4(ph[4], ph[5], ph[6], ph[7]);
                d2[0] = make_uint4(pl[0], pl[1], pl[2], pl[3]); d2[1] = make_uint4(pl[4], pl[5], pl[6], pl[7]);
            }
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] = 0.f;
        }
        if (outT != nullptr) {
#pragma unroll
            for (int e = 0; e < 16; ++e) tile[row][cg + e] = v[e];
        }
    }
    if (outT == nullptr) return;    // uniform per workgroup
    __syncthreads();
    {
        const int oc = t >> 2, seg = (t & 3) * 16;
        if (r0 + seg < R) {
            unsigned pk[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) pk[e] = pack2bf(tile[seg + 2 * e][oc], tile[seg + 2 * e + 1][oc]);
            uint4* dst = reinterpret_cast<uint4*>(outT + (size_t)(c0 + oc) * R + r0 + seg);
            dst[0] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
            dst[1] = make_uint4(pk[4], pk[5], pk[6], pk[7]);
        }
    }
}
extern "C" int sed_weight_images(const int64_t* desc, int n_desc, int total_tiles, hipStream_t stream) {
    (void)hipGetLastError();
    if (n_desc <= 0 || total_tiles <= 0) return SED_ERR_ARG;
    hipLaunchKernelGGL(weight_images_kernel, dim3(total_tiles), dim3(256), 0, stream, (const long long*)desc, n_desc);
    return sed_check_launch();
}

// Split-precision operand images (f16 hi + f16 lo carries ~22 significand bits): a GEMM over the concatenated reduction
// dimension [A_hi | A_lo | A_hi] . [W_hi | W_hi | W_lo]^T accumulates A_hi W_hi + A_lo W_hi + A_hi W_lo in fp32 inside the
// ordinary MFMA kernel.  Used for the context-network GEMMs, whose operand rounding dominates the posterior error.
// in fp32 [M, K] -> out f16 [M, 3K]; mode 0: [hi | lo | hi] (activations), mode 1: [hi | hi | lo] (weights)
__global__ void split3_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, size_t M, int K, int mode) {
    const size_t total = M * (size_t)(K / 2);
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const size_t m = idx / (K / 2);
        const int k = (int)(idx - m * (K / 2)) * 2;
        const float2 v = *reinterpret_cast<const float2*>(in + m * K + k);
        const bf16_t h0 = f2h(v.x), h1 = f2h(v.y);
        const bf16_t l0 = f2h(v.x - h2f(h0)), l1 = f2h(v.y - h2f(h1));
        const unsigned hi = (unsigned)h0 | ((unsigned)h1 << 16), lo = (unsigned)l0 | ((unsigned)l1 << 16);
        unsigned* row = reinterpret_cast<unsigned*>(out + m * (size_t)(3 * K));
        row[k / 2] = hi;
        row[(K + k) / 2] = mode == 0 ? lo : hi;
        row[(2 * K + k) / 2] = mode == 0 ? hi : lo;
    }
}
extern "C" int sed_split3_f16(const float* in, void* out, int64_t M, int K, int mode, hipStream_t stream) {
    (void)hipGetLastError();
    if (K % 2 || M <= 0) return SED_ERR_ARG;
    size_t total = (size_t)M * (K / 2);
    int blocks = (int)((total + 255) / 256);
    blocks = blocks > 4096 ? 4096 : blocks;
    hipLaunchKernelGGL(split3_kernel, dim3(blocks), dim3(256), 0, stream, in, (bf16_t*)out, (size_t)M, K, mode);
    return sed_check_launch();
}

// in-place f16 -> bf16 conversion of saved forward operands that the backward consumes as bf16 MFMA operands
__global__ void f16_to_bf16_kernel(bf16_t* __restrict__ p, size_t n8) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
        uint4 v = reinterpret_cast<uint4*>(p)[i];
        unsigned* w = reinterpret_cast<unsigned*>(&v);
#pragma unroll
        for (int k = 0; k < 4; ++k) w[k] = pack2bf(h2f((bf16_t)(w[k] & 0xFFFF)), h2f((bf16_t)(w[k] >> 16)));
        reinterpret_cast<uint4*>(p)[i] = v;
    }
}
extern "C" int sed_f16_to_bf16_inplace(void* p, int64_t n, hipStream_t stream) {
    (void)hipGetLastError();
    if (n % 8) return SED_ERR_ARG;
    size_t n8 = n / 8;
    int blocks = (int)((n8 + 255) / 256);
    blocks = blocks > 4096 ? 4096 : (blocks < 1 ? 1 : blocks);
    hipLaunchKernelGGL(f16_to_bf16_kernel, dim3(blocks), dim3(256), 0, stream, (bf16_t*)p, n8);
    return sed_check_launch();
}

// Small-M fp32 linear: out[m, n] = sum_k a[m, k] w[n, k] + b[n]   (one wave per output element row-chunk).
// Used for the batch-independent / per-clip tiny projections (AT-head query, out_proj, 768->10 classifier).
__global__ void small_linear_kernel(const float* __restrict__ a, const float* __restrict__ w,
                                    const float* __restrict__ b, float* __restrict__ out, int M, int N, int K,
                                    int act) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= M * N) return;
    const int m = wave / N, n = wave - m * N;
    float s = 0.f;
    for (int k = lane; k < K; k += 64) s += a[(size_t)m * K + k] * w[(size_t)n * K + k];
    s = wave_sum(s);
    if (lane == 0) {
        s += (b != nullptr ? b[n] : 0.f);
        if (act == 1) s = sigmoidf_(s);
        out[(size_t)m * N + n] = s;
    }
}

extern "C" int sed_small_linear(const float* a, const float* w, const float* b, float* out, int M, int N, int K,
                                int act, hipStream_t stream) {
    (void)hipGetLastError();
    const int64_t waves = (int64_t)M * N;
    hipLaunchKernelGGL(small_linear_kernel, dim3(cdiv(waves * 64, 256)), dim3(256), 0, stream, a, w, b, out, M, N,
                       K, act);
    return sed_check_launch();
}
