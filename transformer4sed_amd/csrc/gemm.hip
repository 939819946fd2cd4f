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
#include <algorithm>
#include <atomic>

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
    int group_m;  // 256^2 kernel: tile rows per L2 group (see gemm_nt_pp_kernel)
    int persist;  // 256^2 kernel: one workgroup per CU walks the tiles blockIdx.x, blockIdx.x + gridDim.x, ...
    int ncols;    // 128^2 kernel: output columns >= ncols are computed but not written (operands padded to the tile width)
    // Row-group bias (nullable): row m additionally gets gbias[(m / gb_rows) * N + n] -- one bias row per clip (gb_rows = tokens per
    // clip, M % gb_rows == 0, gb_rows >= 128).  Carries the weight-rounding correction of the evaluation-mode encoder (engine.py
    // `_wcorr_bias`): mean activation of the clip x the part of the fp32 weight its f16 image dropped.
    const float* gbias;
    int gb_rows;
    // Two-term weights (256^2 kernel, evaluation-mode encoder): B rows are [f16(W) | f16(W - f16(W))] over K = 2 * k_wrap * 64 and the
    // A panel (k_wrap K tiles wide) is walked twice -- the fp32 weight to ~2^-19 against the same f16 activations.  0 = off.
    int k_wrap;
    // ... and, when the B rows are a split-precision weight image [hi | hi | lo] over 3 * k_wrap K tiles (context network), the number of B
    // K tiles to skip once the A panel has wrapped: the walk then multiplies A . hi^T and A . lo^T (b_skip = k_wrap), dropping the a_lo
    // term of the three-term product.
    int b_skip;
    // Two-term weights with the lo product on the fp8 matrix path (256^2 kernel, evaluation-mode encoder; k8 != 0): A rows are
    // [K f16 | K e4m3] (the activation and its OCP-e4m3 image, scaled by 2^-2: sed_fp8_tail or the producing kernels), B rows
    // [f16(W) | e4m3(2^s (W - f16(W)))] (sed_weight_two_term_f8); both with a row pitch of 3 K bytes.  The K walk is K / 64 f16 tiles followed
    // by k8 = K / 128 fp8 tiles of the same 128 bytes per row: same DMA, same LDS image, same fragment reads -- the 32 bytes a lane holds
    // of a row are the operand of ONE v_mfma_scale_f32_16x16x128_f8f6f4 instead of two v_mfma_f32_16x16x32_f16 (the k order inside a
    // tile is the same permutation for both operands).  f8_scale = the e8m0 byte 128 - s / 2 replicated four times, used as BOTH operands' block
    // scale (s even): the hardware scales the lo product back by 2^-(s - 2).  W_lo is 2^-12 of the result and e4m3 keeps 2^-4 of either factor: the lo product to ~2^-15 of the
    // result for half of an f16 pass (`profiles/r4_fp8_mfma_probe.txt`).
    int k8;
    int f8_scale;
    // ... and the 16-bit output of such a GEMM (EPI_GELU's outH2: the fc1 activation, fc2's A operand) can carry its own e4m3 image:
    // rows [N f16 | N e4m3], ldc >= 3N / 2 halfs
    int out_e4m3;
    // Persistent 256^2 kernel: workgroups with an odd (blockIdx.x >> 3) start `stagger` ticks of the 100 MHz real-time counter late, so
    // that their store phases fall into the other half's main loops instead of all 256 CUs hitting HBM in lock-step.  0 = off.
    int stagger;
    // LayerNorm folded into the GEMMs around it (256^2 kernel, no-grad f16 passes; engine.py `_encoder_fwd`):
    //   producer (EPI_F32_RESID): besides the fp32 residual stream it writes the f16 image of the new stream to outH and, per row and
    //     64-column slice, the partial sums (sum x, sum x^2) to rowpart [M][N / 64][2];
    //   consumer (EPI_QKV / EPI_GELU): A is that f16 image of the RAW stream, B the f16 image of gamma (.) W; row m of the result is
    //     rstd[m] * (acc - mean[m] * colS[n]) + bias[n] with rowstat [M][2] = (mean, rstd), colS[n] = sum_k B[n, k], and `bias` already
    //     holding beta . W^T + b  ->  exactly Linear(LayerNorm(x)) without the normalised tensor ever existing.
    float* rowpart;
    const float* rowstat;
    const float* colS;
    // producer, split-plane residual stream: the stream lives as two f16 planes hi + lo (~22 significand bits) instead of fp32 -- the hi
    // plane IS the next GEMM's A operand, so the producer moves 8 bytes per element (read hi, lo; write hi, lo) instead of 10 (fp32
    // read-modify-write + the f16 image).  res_lo != NULL: the residual comes as auxH (hi) + res_lo; out_lo != NULL: the result leaves as
    // outH (hi) + out_lo and outF is not written.
    const bf16_t* res_lo;
    bf16_t* out_lo;
    // ... with an 8-bit lo plane (lo8 != 0; res_lo / out_lo then point at [M][ldc] BYTES): the stream value is hi * (1 + (q - 128) * 2^-18),
    // q = the byte -- the f16 rounding residual |x - hi| <= ulp(hi) / 2 <= |hi| 2^-11 in steps of |hi| 2^-18, i.e. the stream to ~2^-19
    // relative (the f16 lo plane: ~2^-22; an f16 stream: 2^-12) for 6 instead of 8 bytes per element through a producer.
    int lo8;
    // ... and with lo8 the hi plane (outH / auxH of the producer) is slab-major as well: [ldc / 64][M][64] f16 -- a wave's 128 rows x 64
    // columns are one contiguous 16 KB run, and for the consumer GEMMs (a_slab != 0: A is such a plane with K / 64 slabs) a K tile of
    // 256 rows is one contiguous 32 KB run instead of 256 pieces of 128 bytes, 1536 bytes apart.
    int a_slab;
    // ... and the 16-bit output of the fused-GELU consumer (outH2) can leave slab-major as well (c_slab != 0: [N / 64][M][64]) -- the fc1
    // activation of a folded block is read by nobody but the fc2 producer, as its slab-major A operand.
    int c_slab;
    // Persistent 256^2 kernel, dynamic tile walk: 8 per-XCD tile counters + 1 exit counter, one 64-byte line each (int index 16 x).  A
    // workgroup takes the next tile of ITS XCD's contiguous tile range from counter blockIdx.x & 7 (so the L2 grouping of the static walk
    // is kept) instead of the fixed blockIdx.x + k gridDim.x: a workgroup that becomes resident late -- or only after the others have
    // drained the range, when communication kernels hold CUs -- costs nothing instead of a whole column of tiles.  NULL = static walk.
    int* tile_ctr;
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
    if (g.gbias != nullptr) {
        const float4 bb = *reinterpret_cast<const float4*>(g.gbias + (size_t)(m / g.gb_rows) * g.N + n);
        b[0] += bb.x; b[1] += bb.y; b[2] += bb.z; b[3] += bb.w;
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

template <int EPI, bool F16>
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
    // Direct-to-LDS staging (global_load_lds_dwordx4): each wave-instruction DMAs 64 x 16 B = 8 tile rows straight into LDS
    // (lane-linear destination), no VGPR round trip and no ds_write pass.  The XOR swizzle therefore sits on the per-lane SOURCE
    // address: lane l fills physical chunk l & 7 of row 8 p + (l >> 3) with logical chunk (l & 7) ^ ((row >> 1) & 7) -- the same
    // involution the fragment reads apply.  Wave w owns pieces 4 w .. 4 w + 3.
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
// 256 x 256 x 64 workgroup tile, 8 waves (2 x 4), 128 x 64 per wave, "ping-pong" K loop on v_mfma_f32_16x16x32.
//
// Why this shape of loop (measured on MI355X, tools/ablate/pp_lab.hip, mfma_bench.hip, dma_bench.hip):
//  * The chip is POWER-limited in a dense GEMM on real data: the clock settles at 1.35-1.65 GHz (2.4 GHz on all-zero
//    operands), so TFLOP/s is set by energy per FLOP, not by issue slots.  v_mfma_f32_16x16x32 sustains ~1800 TFLOP/s on
//    random operands, v_mfma_f32_32x32x16 ~1550 (twice the accumulator read/write traffic per FLOP): the whole kernel moved
//    from 1270 to 1440 TFLOP/s at 8192^3 by changing the MFMA shape alone.  Fragment reads cost ~6 %, the DMA ~13 % of the energy.
//  * The two wave rows run half a phase apart.  A K tile is four phases; in each, a wave computes one 64 x 32 quadrant of its
//    sub-tile (4 x 2 blocks x 2 k-steps = 16 back-to-back MFMAs) between two workgroup barriers, and before the first of them
//    does its memory work: 2 direct-to-LDS DMA pieces of a later K tile and the fragment reads the coming quadrant needs (A half:
//    8 ds_read_b128, B half: 4).  Waves 4-7 execute one extra barrier up front, so on every SIMD one wave is in its MFMA section
//    while its partner issues DMA / LDS reads: 92 % MFMA-pipe occupancy in cycles, against two waves that stall on the same
//    `s_waitcnt lgkmcnt(0)` at the same time in a lock-step loop.
//  * Reads per phase are balanced 8/4/8/4: the next tile's B half 0 is read in P4 into the registers that B half 1 vacated after
//    P3, so the two B register sets swap roles every K tile (two tiles per loop trip).
// DMA plan for K tile t+1 (16 pieces of 8 rows x 128 B per slot, 2 per wave): B half-0 rows at P2(t-1), A half-0 rows at P3(t-1),
// B half-1 rows at P4(t-1), A half-1 rows at P1(t) -- each at least two barriers after the last read of the region it overwrites --
// retired by ONE counted `s_waitcnt vmcnt(4)` at P3(t) ahead of that phase's first barrier (the 4 youngest pieces belong to tile
// t+2); the first read of tile t+1 (its B half 0) is a phase later, in P4(t).
// Operands come through buffer descriptors: rows past M read as zeros, 32-bit lane offsets + an SGPR K offset.
// LDS: A stage s at s * 32 KiB, B stage s at 64 KiB + s * 32 KiB -> every fragment read is `vaddr + imm16`; the address registers
// flip bit 15 per K tile.  128-byte rows, 16-B chunk c of row r stored at chunk c ^ ((r >> 1) & 7) (applied on the DMA source
// address): conflict-free for the 16 x 32 fragment reads (lane & 15 = row, lane >> 4 = chunk inside the k-step).
// Accumulators: acc[i][j], i = 0..7 (16-row blocks), j = 0..3 (16-column blocks); the MFMA operands are issued swapped (B fragment
// first), so a block holds C^T: lane & 15 = row, register r = column 4 (lane >> 4) + r of the block's 16 weight rows -> a lane owns 4
// CONSECUTIVE columns of one row per block.  Which 16 output columns a block's weight rows are is the DMA's choice (round 4): PP_COL /
// PP_COL32 below place them so that a lane's blocks form 16-byte runs that the epilogues store straight from the registers.
// ---------------------------------------------------------------------------------------------------------------------
// ---- staged epilogue (what is left of it after round 4: the fp32 + 16-bit outputs, the head split with transposed copies, the
//      LayerNorm-fold producers at the ends of a folded run, the evaluation-mode residual GEMMs; everything else is LDS-free) -----------
// After the K loop each wave owns a private 17 KiB LDS region.  The accumulators are written there as a row-major [rows][64]
// sub-tile (padded row stride), then read back so that 8 (16-bit) or 16 (fp32) consecutive lanes cover one full output row: every
// global store / residual load is a run of whole 128- / 256-byte rows instead of 64 different cache lines per instruction.
#define V3_WLDS 17408          // per-wave bytes: 128 rows x 136 B (16-bit) or 64 rows x 264 B (fp32, two passes)
#define V3_RS16 136
#define V3_RS32 264
#define V3_T 256
#define V3_STAGE (64 * 1024)
#define V3_LDS (8 * V3_WLDS > 2 * V3_STAGE ? 8 * V3_WLDS : 2 * V3_STAGE)

// Per-lane column constants (bias, plus the rel-pos u / v vector for the q projection) for the 16 columns a lane owns: fetched ONCE,
// before any store -- the output pointers may alias them as far as the compiler knows, so a load placed between stores costs a full
// `s_waitcnt vmcnt(0)` round trip each time (measured: 8 us / tile).
// Column order of a wave's 128 x 64 sub-tile (round 4).  MFMA block j, lane group lq = lane >> 4, register r used to be column
// 16 j + 4 lq + r: a lane's 16 values of a row were four separate 8-byte runs, and the tile had to go through LDS to become whole rows.
// The B operand's DMA now places weight row PP_COL(j, lq) + r at LDS row 16 j + 4 lq + r of its 64-row group (a permutation of address
// bits 2..4 on the DMA SOURCE side; fragment reads and swizzle unchanged), so the lane holds columns 8 lq .. 8 lq + 7 (blocks 0, 1)
// and 32 + 8 lq .. 32 + 8 lq + 7 (blocks 2, 3): two 16-byte runs of 16-bit outputs, and the four lane groups of a row make 64
// contiguous bytes per store instruction -- the 16-bit epilogues store straight from the accumulators, no LDS round trip.
__device__ __forceinline__ constexpr int PP_COL(int j, int lq) { return 8 * lq + 32 * (j >> 1) + 4 * (j & 1); }
// LDS row (within the B stage) -> weight row of the tile
__device__ __forceinline__ int pp_brow_src(int row) { return (row & ~0x1C) | ((row & 0x10) >> 2) | ((row & 0x0C) << 1); }
// Column order of the kernel variants whose epilogue stores fp32 straight from the accumulators (EPI_F32, the plain EPI_F32_RESID):
// block j, lane group lq -> columns 32 (j & 1) + 16 (j >> 1) + 4 lq .. + 3.  After the lane-pair swap (pp_pair_swap32) a lane holds, of
// one row, blocks 2 hk and 2 hk + 1 (hk = l15 >> 3): float4 number 4 hk + lq of the row's first and of its second 128 bytes -- each of the
// two 16-byte stores (or residual loads) per row is then 8 lanes x 16 B = 128 contiguous bytes, 8 rows per instruction.
__device__ __forceinline__ constexpr int PP_COL32(int j, int lq) { return 32 * (j & 1) + 16 * (j >> 1) + 4 * lq; }
__device__ __forceinline__ int pp_brow_src32(int row) { return (row & ~0x30) | ((row & 0x10) << 1) | ((row & 0x20) >> 1); }
__device__ __forceinline__ void pp_col_consts(float (&bv)[4][4], const float* bias_n, const float* extra, int lq) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int col = PP_COL(j, lq);
        float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bias_n != nullptr) b = *reinterpret_cast<const float4*>(bias_n + col);
        if (extra != nullptr) {
            const float4 x = *reinterpret_cast<const float4*>(extra + col);
            b.x += x.x; b.y += x.y; b.z += x.z; b.w += x.w;
        }
        bv[j][0] = b.x; bv[j][1] = b.y; bv[j][2] = b.z; bv[j][3] = b.w;
    }
}

// Row-group bias of a wave's 128-row sub-tile (rows mb .. mb + 127): with gb_rows >= 128 it touches at most two groups; rows below
// `bnd` (relative to mb) belong to the first, the rest to the second.
__device__ __forceinline__ int gb_split(const GemmArgs& g, int mb, const float*& rowA, const float*& rowB) {
    const int last = g.M / g.gb_rows - 1;
    int gA = mb / g.gb_rows;
    gA = gA < last ? gA : last;
    const int gB = gA < last ? gA + 1 : last;
    rowA = g.gbias + (size_t)gA * g.N;
    rowB = g.gbias + (size_t)gB * g.N;
    return (gA + 1) * g.gb_rows - mb;
}
// C-tile stores of the 256^2 kernels are non-temporal: the tile is not re-read by this kernel, and write-allocating it evicts
// the A/B panels that the neighbouring column tiles still need from the 4 MB L2 (+2..4 % on the model's shapes).
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
template <class T>
__device__ __forceinline__ void v3_st(void* p, const T& v) {
    if constexpr (sizeof(T) == 16) {
        u32x4_t t;
        __builtin_memcpy(&t, &v, 16);
        __builtin_nontemporal_store(t, reinterpret_cast<u32x4_t*>(p));
    } else {
        u32x2_t t;
        __builtin_memcpy(&t, &v, 8);
        __builtin_nontemporal_store(t, reinterpret_cast<u32x2_t*>(p));
    }
}

// ---- 16-bit epilogue sinks: where a lane's 8 consecutive 16-bit outputs of row block i, column half k (PP_COL order: columns
// 32 k + 8 lq .. + 7 of the wave's 64) go.  Global sinks store 16 bytes per lane straight from the accumulators; the LDS sink keeps
// the staged form for the head-split variants that also write transposed copies.
// A store instruction of 64 lanes x 16 B should cover 8 rows x 128 contiguous bytes (for the head-split q / k / v a wave's 128 token rows
// of one head are ONE contiguous 16 KB run), not 16 rows x 64: lanes l15 and l15 ^ 8 of a 16-lane DPP row swap one half each
// (`row_ror:8`), after which lanes l15 < 8 hold the low 64 bytes and lanes l15 >= 8 the high 64 bytes of rows 16 i + (l15 & 7)
// (first store) and 16 i + 8 + (l15 & 7) (second store).  12 VALU operations per 16-row block instead of the LDS round trip.
__device__ __forceinline__ unsigned pp_ror8(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xF, 0xF, true);      // row_ror:8
}
__device__ __forceinline__ void pp_pair_swap(uint4& a0, uint4& a1, bool lo) {
    // in: a0 / a1 = this lane's row, column halves 0 / 1.  out: a0 = (row l15 & 7, half l15 >> 3), a1 = (row 8 + (l15 & 7), half l15 >> 3)
    // (component-wise selects: a conditional between the two structs becomes a pointer select into scratch memory)
    const unsigned rx = pp_ror8(lo ? a1.x : a0.x), ry = pp_ror8(lo ? a1.y : a0.y), rz = pp_ror8(lo ? a1.z : a0.z), rw = pp_ror8(lo ? a1.w : a0.w);
    a0 = make_uint4(lo ? a0.x : rx, lo ? a0.y : ry, lo ? a0.z : rz, lo ? a0.w : rw);
    a1 = make_uint4(lo ? rx : a1.x, lo ? ry : a1.y, lo ? rz : a1.z, lo ? rw : a1.w);
}
struct PPSinkLds {
    unsigned char* wl; int l15, lq;
    __device__ __forceinline__ void half(int i, int k, const uint4& v) const {
        unsigned char* d = wl + (i * 16 + l15) * V3_RS16 + (32 * k + 8 * lq) * 2;      // (136-byte rows: two 8-byte writes)
        *reinterpret_cast<uint2*>(d) = make_uint2(v.x, v.y);
        *reinterpret_cast<uint2*>(d + 8) = make_uint2(v.z, v.w);
    }
    __device__ __forceinline__ void row(int i, const uint4& a0, const uint4& a1) const { half(i, 0, a0); half(i, 1, a1); }
};
template <bool TAIL = false>
struct PPSinkRowsT {         // row-major [M][ldc] output
    bf16_t* base;            // row mb + (l15 & 7), column nb + 32 (l15 >> 3) + 8 lq
    bf16_t* hbase;           // row mb + l15, column nb + 8 lq   (un-swapped halves)
    int ld8;                 // 8 * ldc
    int mrem, hrem;          // rows left from base's / hbase's row
    bool lo;
    // TAIL (F8 kernels): rows are [N f16 | N e4m3] and `tail` > 0 is the byte distance from a lane's 8 f16 outputs to their 8 e4m3 images
    // (GemmArgs.out_e4m3: the next GEMM's A operand with both halves written here); 0 = no image.  What the image costs is its bytes
    // (+50 % on the epilogue's writes: 70 us of 510 at M = 105 k, N = 3072), not its conversion: with MODE.FP16_OVFL = 1 the conversion
    // saturates by itself (tools/ablate/cvt_fp8_probe.hip) and the packed clamps can go -- 4 instructions instead of 12 per 8 values --
    // for no measurable change (726 vs 723 us).
    int tail;
    __device__ __forceinline__ void st16(bf16_t* p, const uint4& v) const {
        v3_st<uint4>(p, v);
        if constexpr (TAIL) {
#ifndef ABL_NO_TAIL      // (timing ablation, tools/ablate/build_variant.sh)
            if (tail > 0) v3_st<uint2>(reinterpret_cast<unsigned char*>(p) + tail, make_uint2(e4m3x4_of_h4(v.x, v.y), e4m3x4_of_h4(v.z, v.w)));
#endif
        }
    }
    __device__ __forceinline__ void half(int i, int k, const uint4& v) const {
        if (16 * i < hrem) v3_st<uint4>(hbase + (size_t)(2 * i) * ld8 + 32 * k, v);
    }
    __device__ __forceinline__ void row(int i, uint4 a0, uint4 a1) const {
        pp_pair_swap(a0, a1, lo);
        if (16 * i < mrem) st16(base + (size_t)(2 * i) * ld8, a0);
        if (16 * i + 8 < mrem) st16(base + (size_t)(2 * i + 1) * ld8, a1);
    }
};
typedef PPSinkRowsT<false> PPSinkRows;
struct PPSinkHeads {         // head-split q / k / v: [clip * heads + h][seq][64]
    bf16_t* dst;             // + h * seq * 64 + 32 (l15 >> 3) + 8 lq already applied
    int b0, t0, seq, hs, mrem;   // clip / token of row mb + (l15 & 7); hs = heads * seq; rows left from there
    bool lo;
    __device__ __forceinline__ void st(int r, const uint4& v) const {      // r = row offset from this lane's first row (a multiple of 8)
        if (r >= mrem) return;
        int t = t0 + r, b = b0;
        if (seq >= 128) { if (t >= seq) { t -= seq; ++b; } }
        else { const int q = t / seq; t -= q * seq; b += q; }
        v3_st<uint4>(dst + ((size_t)b * hs + t) * 64, v);
    }
    __device__ __forceinline__ void row(int i, uint4 a0, uint4 a1) const {
        pp_pair_swap(a0, a1, lo);
        st(16 * i, a0);
        st(16 * i + 8, a1);
    }
};
template <bool F16, int MODE>
__device__ __forceinline__ uint4 pp_pack8(f32x2v a0, f32x2v a1, f32x2v b0, f32x2v b1) {
    if (MODE == 1) { a0 = gelu_fast2(a0); a1 = gelu_fast2(a1); b0 = gelu_fast2(b0); b1 = gelu_fast2(b1); }
    return make_uint4(pack2<F16>(a0.x, a0.y), pack2<F16>(a1.x, a1.y), pack2<F16>(b0.x, b0.y), pack2<F16>(b1.x, b1.y));
}

// per-row choice between two sets of column constants (bv + group-bias row A for rows < bnd, + row B otherwise), kept as
// base = bv + A and delta = B - A (32 registers, as many as one 32-column half of both sets used to take): all four column blocks of a
// row block are then at hand together and the row leaves through the pair-swapped whole-row stores like every other 16-bit epilogue
template <bool F16, int MODE, class Sink>
__device__ __forceinline__ void pp_stage16_gb(const Sink& sink, const f32x4_t (&acc)[8][4], const float (&bv)[4][4], const float* rowA,
                                              const float* rowB, int bnd, int l15, int lq) {
    float base[4][4], delta[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float4 a4 = *reinterpret_cast<const float4*>(rowA + PP_COL(j, lq)), b4 = *reinterpret_cast<const float4*>(rowB + PP_COL(j, lq));
        base[j][0] = bv[j][0] + a4.x; base[j][1] = bv[j][1] + a4.y; base[j][2] = bv[j][2] + a4.z; base[j][3] = bv[j][3] + a4.w;
        delta[j][0] = b4.x - a4.x; delta[j][1] = b4.y - a4.y; delta[j][2] = b4.z - a4.z; delta[j][3] = b4.w - a4.w;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const bool second = i * 16 + l15 >= bnd;
        uint4 o[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            f32x2v x[2][2];
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int j = 2 * k + jj;
                const f32x4_t& a = acc[i][j];
                x[jj][0] = f32x2v{a[0] + base[j][0] + (second ? delta[j][0] : 0.f), a[1] + base[j][1] + (second ? delta[j][1] : 0.f)};
                x[jj][1] = f32x2v{a[2] + base[j][2] + (second ? delta[j][2] : 0.f), a[3] + base[j][3] + (second ? delta[j][3] : 0.f)};
            }
            o[k] = pp_pack8<F16, MODE>(x[0][0], x[0][1], x[1][0], x[1][1]);
        }
        sink.row(i, o[0], o[1]);
    }
}

// (the head-split epilogue's form: column constants per 32-column half -- 16 registers live at a time, it has none to spare -- halves
//  handed to the sink one at a time)
template <bool F16, int MODE, class Sink>
__device__ __forceinline__ void pp_stage16_gb_k(const Sink& sink, const f32x4_t (&acc)[8][4], const float (&bv)[4][4], const float* rowA,
                                                const float* rowB, int bnd, int l15, int lq) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        float cA[2][4], cB[2][4];
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int j = 2 * k + jj;
            const float4 a4 = *reinterpret_cast<const float4*>(rowA + PP_COL(j, lq)), b4 = *reinterpret_cast<const float4*>(rowB + PP_COL(j, lq));
            cA[jj][0] = bv[j][0] + a4.x; cA[jj][1] = bv[j][1] + a4.y; cA[jj][2] = bv[j][2] + a4.z; cA[jj][3] = bv[j][3] + a4.w;
            cB[jj][0] = bv[j][0] + b4.x; cB[jj][1] = bv[j][1] + b4.y; cB[jj][2] = bv[j][2] + b4.z; cB[jj][3] = bv[j][3] + b4.w;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const bool second = i * 16 + l15 >= bnd;
            f32x2v x[2][2];
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const f32x4_t& a = acc[i][2 * k + jj];
                x[jj][0] = f32x2v{a[0] + (second ? cB[jj][0] : cA[jj][0]), a[1] + (second ? cB[jj][1] : cA[jj][1])};
                x[jj][1] = f32x2v{a[2] + (second ? cB[jj][2] : cA[jj][2]), a[3] + (second ? cB[jj][3] : cA[jj][3])};
            }
            sink.half(i, k, pp_pack8<F16, MODE>(x[0][0], x[0][1], x[1][0], x[1][1]));
        }
    }
}

// LayerNorm-folded form: x = rstd[row] * (acc - mean[row] * sv[col]) + bv[col]  (rows mb + 16 i + l15, statistics clipped to the last row)
// (BVP: the column constants are fetched per 32-column half from `bias_n` instead of living in 16 registers -- the head-split epilogue
//  has no room for them beside the accumulators)
template <bool F16, int MODE, bool BVP = false, int RB = 8, class Sink>
__device__ __forceinline__ void pp_stage16_ln(const Sink& sink, const f32x4_t (&acc)[8][4], const float (&bv)[4][4], const float* colS_n,
                                              const float* rowstat, int mb, int M, int l15, int lq, const float* bias_n = nullptr) {
    // every global load of the epilogue is issued up front (8 row statistics, the column vectors): one exposed round trip instead of
    // one per column block -- the K loop's fragment registers are free by now
    float2 st[8];
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        const int m = mb + i * 16 + l15;
        st[i] = *reinterpret_cast<const float2*>(rowstat + 2 * (size_t)(m < M ? m : M - 1));
    }
    if constexpr (BVP) {      // head-split epilogue: the bias vector comes from `bias_n` (no per-lane constants were kept over the K loop)
        float4 sn[4], bn[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            sn[j] = *reinterpret_cast<const float4*>(colS_n + PP_COL(j, lq));
            bn[j] = *reinterpret_cast<const float4*>(bias_n + PP_COL(j, lq));
        }
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const float rs = st[i].y, tm = -st[i].x * st[i].y;
            uint4 o[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                f32x2v x[2][2];
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int j = 2 * k + jj;
                    const f32x4_t& a = acc[i][j];
                    x[jj][0] = f32x2v{__builtin_fmaf(a[0], rs, __builtin_fmaf(tm, sn[j].x, bn[j].x)), __builtin_fmaf(a[1], rs, __builtin_fmaf(tm, sn[j].y, bn[j].y))};
                    x[jj][1] = f32x2v{__builtin_fmaf(a[2], rs, __builtin_fmaf(tm, sn[j].z, bn[j].z)), __builtin_fmaf(a[3], rs, __builtin_fmaf(tm, sn[j].w, bn[j].w))};
                }
                o[k] = pp_pack8<F16, MODE>(x[0][0], x[0][1], x[1][0], x[1][1]);
            }
            sink.row(i, o[0], o[1]);
        }
        return;
    }
    float4 s4[4], b4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        s4[j] = *reinterpret_cast<const float4*>(colS_n + PP_COL(j, lq));
        b4[j] = make_float4(bv[j][0], bv[j][1], bv[j][2], bv[j][3]);
    }
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        const float rs = st[i].y, tm = -st[i].x * st[i].y;
        uint4 o[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            f32x2v x[2][2];
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int j = 2 * k + jj;
                x[jj][0] = f32x2v{__builtin_fmaf(acc[i][j][0], rs, __builtin_fmaf(tm, s4[j].x, b4[j].x)),
                                  __builtin_fmaf(acc[i][j][1], rs, __builtin_fmaf(tm, s4[j].y, b4[j].y))};
                x[jj][1] = f32x2v{__builtin_fmaf(acc[i][j][2], rs, __builtin_fmaf(tm, s4[j].z, b4[j].z)),
                                  __builtin_fmaf(acc[i][j][3], rs, __builtin_fmaf(tm, s4[j].w, b4[j].w))};
            }
            o[k] = pp_pack8<F16, MODE>(x[0][0], x[0][1], x[1][0], x[1][1]);
        }
        sink.row(i, o[0], o[1]);
    }
}

template <bool F16, int MODE, int RB = 8, class Sink>
__device__ __forceinline__ void pp_stage16(const Sink& sink, const f32x4_t (&acc)[8][4], const float (&bv)[4][4]) {
    // mode 0: acc + column constant   1: gelu_fast(acc + column constant)
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        uint4 o[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            f32x2v x[2][2];
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int j = 2 * k + jj;
                x[jj][0] = f32x2v{acc[i][j][0] + bv[j][0], acc[i][j][1] + bv[j][1]};
                x[jj][1] = f32x2v{acc[i][j][2] + bv[j][2], acc[i][j][3] + bv[j][3]};
            }
            o[k] = pp_pack8<F16, MODE>(x[0][0], x[0][1], x[1][0], x[1][1]);
        }
        sink.row(i, o[0], o[1]);
    }
}
// 64 staged rows (accumulator blocks 4 p .. 4 p + 3) as fp32
// (PERM: the accumulators are in PP_COL column order -- the NT ping-pong kernel; the TN kernel's are in plain order)
template <int RB = 8, bool PERM = false>
__device__ __forceinline__ void pp_stage32(unsigned char* wl, const f32x4_t (&acc)[8][4], int p, int l15, int lq) {
#pragma unroll
    for (int ii = 0; ii < 4; ++ii)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (4 * p + ii >= RB) continue;
            const f32x4_t& a = acc[4 * p + ii][j];
            *reinterpret_cast<float4*>(wl + (ii * 16 + l15) * V3_RS32 + (PERM ? PP_COL(j, lq) : j * 16 + 4 * lq) * 4) = make_float4(a[0], a[1], a[2], a[3]);
        }
}
// Byte plane layout: slab-major [N / 64][M][64] -- the 64 bytes of row m in 64-column slab s sit at (s * M + m) * 64, so a wave's
// 128 rows x 64 columns are ONE contiguous 8 KB run (row-major they are 128 pieces of 64 bytes, 768 bytes apart).  Only the producers
// touch this plane, so the layout is theirs to choose.
__device__ __forceinline__ size_t lo8_off(const GemmArgs& g, int m, int n) { return ((size_t)(n >> 6) * g.M + m) * 64 + (n & 63); }
// element offset of (m, n) in a plane of the producer: slab-major with the byte lo plane (GemmArgs.lo8), row-major otherwise
__device__ __forceinline__ size_t hi_off(const GemmArgs& g, int m, int n) { return g.lo8 ? lo8_off(g, m, n) : (size_t)m * g.ldc + n; }
// 8-bit lo plane of the split residual stream (GemmArgs.lo8): x = hi + (q - 128) * hi * 2^-18 = hi * (1 + (q - 128) 2^-18)
__device__ __forceinline__ float lo8_decode(float hf, unsigned q) {
    return __builtin_fmaf(__builtin_fmaf((float)q, 0x1p-18f, -0x1p-11f), hf, hf);
}
__device__ __forceinline__ unsigned lo8_encode4(float x0, float x1, float x2, float x3, unsigned pk0, unsigned pk1) {
    // (hi = 0: the quotient is inf / NaN and the byte arbitrary -- the decoder multiplies it by |hi| = 0)
    const float h0 = h2f((bf16_t)(pk0 & 0xFFFF)), h1 = h2f((bf16_t)(pk0 >> 16)), h2 = h2f((bf16_t)(pk1 & 0xFFFF)), h3 = h2f((bf16_t)(pk1 >> 16));
    const float q0 = __builtin_rintf(__builtin_fmaf((x0 - h0) * __builtin_amdgcn_rcpf(h0), 0x1p18f, 128.f));
    const float q1 = __builtin_rintf(__builtin_fmaf((x1 - h1) * __builtin_amdgcn_rcpf(h1), 0x1p18f, 128.f));
    const float q2 = __builtin_rintf(__builtin_fmaf((x2 - h2) * __builtin_amdgcn_rcpf(h2), 0x1p18f, 128.f));
    const float q3 = __builtin_rintf(__builtin_fmaf((x3 - h3) * __builtin_amdgcn_rcpf(h3), 0x1p18f, 128.f));
    unsigned r = __builtin_amdgcn_cvt_pk_u8_f32(q0, 0, 0);      // saturating float -> byte, inserted at byte 0..3
    r = __builtin_amdgcn_cvt_pk_u8_f32(q1, 1, r);
    r = __builtin_amdgcn_cvt_pk_u8_f32(q2, 2, r);
    return __builtin_amdgcn_cvt_pk_u8_f32(q3, 3, r);
}

// Side input of one batch (8 rows per lane, rows m0 + 4 u + lane/16): residual rows (fp32) or saved pre-activations (16-bit)
template <int EPI>
struct V3Side {
    float4 r[EPI == EPI_F32_RESID ? 8 : 1];
    uint2 a[EPI == EPI_DGELU ? 8 : 1];
};
// SM (side mode of the LayerNorm-fold producer): 0 = not a producer, 1 = fp32 residual, 2 = f16 hi + f16 lo planes, 3 = f16 hi + byte lo.
// A compile-time mode (pp_epilogue dispatches on the launch's pointers): with the three load forms as run-time branches of one function
// their results meet in register copies right behind the loads, and every batch waits for its own round trip.
template <int EPI, int SM = 0>
__device__ __forceinline__ void v3_side_load(V3Side<EPI>& sd, const GemmArgs& g, int m0, int n, int lane) {
#ifdef ABL_NO_SIDE
    if constexpr (SM >= 2) return;
#endif
    if constexpr (EPI == EPI_F32_RESID || EPI == EPI_DGELU) {
        if constexpr (SM >= 2) {
            // split-plane residual: the RAW plane words are parked in sd.r and decoded where they are used (v3_residual) -- arithmetic
            // on a loaded value right here makes the compiler wait for every row's loads before it issues the next row's (8 exposed
            // round trips per batch: what the "HBM-saturated" 30 us producer epilogue of rounds 3-4 really was)
            if constexpr (SM == 3) {      // f16 hi + byte lo
                const unsigned char* lo8p = reinterpret_cast<const unsigned char*>(g.res_lo);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int m = m0 + u * 4 + (lane >> 4);
                    const size_t o = (size_t)(m < g.M ? m : g.M - 1) * g.ldc + n;
                    const u32x2_t h = __builtin_nontemporal_load(reinterpret_cast<const u32x2_t*>(g.auxH + lo8_off(g, m < g.M ? m : g.M - 1, n)));
                    const unsigned l = __builtin_nontemporal_load(reinterpret_cast<const unsigned*>(lo8p + lo8_off(g, m < g.M ? m : g.M - 1, n)));
                    sd.r[u] = make_float4(__uint_as_float(h[0]), __uint_as_float(h[1]), __uint_as_float(l), 0.f);
                }
                return;
            }
            if constexpr (SM == 2) {      // f16 hi + f16 lo
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int m = m0 + u * 4 + (lane >> 4);
                    const size_t o = (size_t)(m < g.M ? m : g.M - 1) * g.ldc + n;
                    const u32x2_t h = __builtin_nontemporal_load(reinterpret_cast<const u32x2_t*>(g.auxH + o));
                    const u32x2_t l = __builtin_nontemporal_load(reinterpret_cast<const u32x2_t*>(g.res_lo + o));
                    sd.r[u] = make_float4(__uint_as_float(h[0]), __uint_as_float(h[1]), __uint_as_float(l[0]), __uint_as_float(l[1]));
                }
                return;
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int m = m0 + u * 4 + (lane >> 4);
            const size_t o = (size_t)(m < g.M ? m : g.M - 1) * g.ldc + n;
            // read-once side inputs: non-temporal, like the tile stores
            if constexpr (EPI == EPI_F32_RESID) {
                const f32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(g.resF + o));
                sd.r[u] = make_float4(v[0], v[1], v[2], v[3]);
            } else {
                const u32x2_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x2_t*>(g.auxH + o));
                sd.a[u] = make_uint2(v[0], v[1]);
            }
        }
    }
}
// rows m0 .. m0 + 31 of the output = staged rows srow0 .. srow0 + 31
// 16-lane (one staged row) sum through DPP: quad butterflies, then the two mirrors
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));   // row_mirror
    return v;
}
template <int EPI, bool F16, int SM = 0>
__device__ __forceinline__ void v3_store_batch(const GemmArgs& g, const unsigned char* wl, const V3Side<EPI>& sd, const float4 b0,
                                               int m0, int srow0, int n, int c4, int lane, const float4 bB = make_float4(0.f, 0.f, 0.f, 0.f),
                                               int mbnd = 0x7fffffff, int mend = 0x7fffffff) {
    // (b0 already contains the first group's row-group bias; rows m >= mbnd take bB instead -- see pp_epilogue)
    float4 vv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) vv[u] = *reinterpret_cast<const float4*>(wl + (srow0 + u * 4 + (lane >> 4)) * V3_RS32 + c4 * 16);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int m = m0 + u * 4 + (lane >> 4);
        if (m >= g.M || m >= mend) continue;      // (mend: first row past a 112-row wave sub-tile)
        float4 v = vv[u];
        const float4 b = m >= mbnd ? bB : b0;
        const size_t o = (size_t)m * g.ldc + n;
        if constexpr (EPI == EPI_F32) {
            v3_st<float4>(g.outF + o, make_float4(v.x * g.alpha + b.x, v.y * g.alpha + b.y, v.z * g.alpha + b.z, v.w * g.alpha + b.w));
        } else if constexpr (EPI == EPI_F32_RESID) {
            float4 r = sd.r[u];
            if constexpr (SM >= 2) {      // split-plane residual: sd.r holds the raw plane words (v3_side_load)
                {
                    const unsigned h0 = __float_as_uint(r.x), h1 = __float_as_uint(r.y), l0 = __float_as_uint(r.z), l1 = __float_as_uint(r.w);
                    if constexpr (SM == 3)
                        r = make_float4(lo8_decode(h2f((bf16_t)(h0 & 0xFFFF)), l0 & 0xFF), lo8_decode(h2f((bf16_t)(h0 >> 16)), (l0 >> 8) & 0xFF),
                                        lo8_decode(h2f((bf16_t)(h1 & 0xFFFF)), (l0 >> 16) & 0xFF), lo8_decode(h2f((bf16_t)(h1 >> 16)), l0 >> 24));
                    else
                        r = make_float4(h2f((bf16_t)(h0 & 0xFFFF)) + h2f((bf16_t)(l0 & 0xFFFF)), h2f((bf16_t)(h0 >> 16)) + h2f((bf16_t)(l0 >> 16)),
                                        h2f((bf16_t)(h1 & 0xFFFF)) + h2f((bf16_t)(l1 & 0xFFFF)), h2f((bf16_t)(h1 >> 16)) + h2f((bf16_t)(l1 >> 16)));
                }
            }
            if constexpr (SM >= 2) asm volatile("" : "+v"(r.x), "+v"(r.y), "+v"(r.z), "+v"(r.w));      // (the decoded value is ONE operand: no re-association into it)
#ifdef ABL_NO_SIDE
            r = make_float4(0.f, 0.f, 0.f, 0.f);
#endif
            const float4 x = make_float4(r.x + v.x + b.x, r.y + v.y + b.y, r.z + v.z + b.z, r.w + v.w + b.w);
            if constexpr (SM == 0) v3_st<float4>(g.outF + o, x);
            if constexpr (SM >= 1) {      // LayerNorm-fold producer: f16 image of the stream + this 64-column slice's (sum, sum of squares) per row
                uint2 pk; pk.x = pack2<true>(x.x, x.y); pk.y = pack2<true>(x.z, x.w);
#ifdef ABL_NO_LO
                if (false) {
#else
                if (g.out_lo != nullptr && g.lo8) {
#endif
                    __builtin_nontemporal_store(lo8_encode4(x.x, x.y, x.z, x.w, pk.x, pk.y), reinterpret_cast<unsigned*>(reinterpret_cast<unsigned char*>(g.out_lo) + lo8_off(g, m, n)));
#ifdef ABL_NO_LO
                } else if (false) {
#else
                } else if (g.out_lo != nullptr) {      // split-plane stream: lo = f16(x - hi)
#endif
                    uint2 pl;
                    pl.x = pack2<true>(x.x - h2f((bf16_t)(pk.x & 0xFFFF)), x.y - h2f((bf16_t)(pk.x >> 16)));
                    pl.y = pack2<true>(x.z - h2f((bf16_t)(pk.y & 0xFFFF)), x.w - h2f((bf16_t)(pk.y >> 16)));
                    v3_st<uint2>(g.out_lo + o, pl);
                } else {
                    v3_st<float4>(g.outF + o, x);
                }
#ifndef ABL_NO_HI
                *reinterpret_cast<uint2*>(g.outH + hi_off(g, m, n)) = pk;      // (a plain store: the next GEMM reads this image right away)
#endif
                // (all 16 lanes of a row take this path together: m is uniform across them)
#ifndef ABL_NO_STATS
                const float s1 = row16_sum((x.x + x.y) + (x.z + x.w));
                const float s2 = row16_sum((x.x * x.x + x.y * x.y) + (x.z * x.z + x.w * x.w));
                if (c4 == 0) *reinterpret_cast<float2*>(g.rowpart + ((size_t)m * (g.N >> 6) + (n >> 6)) * 2) = make_float2(s1, s2);
#endif
            }
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

// LayerNorm-fold producer on the split-plane stream with the byte lo plane, planes in -> planes out (every proj / fc2 of a folded run
// except its first and last): straight from the accumulators, no LDS.  The staged form of this epilogue cost 33-40 us per 256^2 tile
// against a 14 us K = 768 main loop (tools/producer_ablate.sh: no single part -- residual loads, hi stores, row sums -- was worth more
// than 4 us of it): ~160 narrow memory instructions (4 / 8 bytes per lane) and ~80 vector instructions per lane and row.  Here a lane
// owns 16 values of a row (PP_COL order: two runs of 8 columns), so the planes move as 16-byte (hi) / 8-byte (lo) accesses -- 72 memory
// instructions per wave and tile --, every load is issued before the first store (a load behind a store waits for that store's
// acknowledgement: the counter retires in order), and the row sums are one 16-value sum per lane + two cross-lane steps.
template <int RB>
__device__ __forceinline__ void pp_epilogue_lo8_direct(const GemmArgs& g, f32x4_t (&acc)[8][4], int mb, int nb, int lane) {
    const int lq = lane >> 4;
    int lrow = lane & 15;
    asm volatile("" : "+v"(lrow));      // (see PPSinkRows: keeps the per-lane addresses out of the persistent loop's live registers)
    const unsigned char* lo_in = reinterpret_cast<const unsigned char*>(g.res_lo);
    unsigned char* lo_out = reinterpret_cast<unsigned char*>(g.out_lo);
    // row blocks in four groups of two: two groups are requested before any store, the next one each time a group has been computed --
    // its loads queue behind that group's stores, one group of arithmetic ahead of their use -- into the registers it vacated
    // (128 accumulators leave room for two groups in flight)
    constexpr int NG = (RB + 1) / 2;
    u32x4_t hw[NG][2][2];
    u32x2_t lw[NG][2][2];
    const int mfirst = mb + lrow;
    const bool lo = lrow < 8;
    const int r8 = lrow & 7;
    const size_t ocol = (size_t)nb + 32 * (lrow >> 3) + 8 * lq;
    // Loads in the pair-swapped layout too (rows 16 i + r8 and 16 i + 8 + r8, column half l15 >> 3): an instruction then covers 8 rows x
    // 128 B of the hi plane / 8 rows x 64 B of the byte plane -- whole lines.  With a lane reading its own row's two halves by two
    // instructions (16 rows x 64 / 32 B pieces) the launches fetched 1.5x their algorithmic bytes and the byte plane cost as many bytes
    // as the f16 one (FETCH_SIZE, profiles/r4_producer_epilogue.txt).  pp_pair_swap (an involution) brings a lane's own row back.
    auto side = [&](int i, u32x4_t (&hw)[2], u32x2_t (&lw)[2]) {
        const int mA = mb + 16 * i + r8, mB = mA + 8;
#ifdef ABL_NO_SIDE
        hw[0] = hw[1] = u32x4_t{0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u}; lw[0] = lw[1] = u32x2_t{(unsigned)mA, 0x80808080u}; return;
#endif
        hw[0] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(g.auxH + lo8_off(g, mA < g.M ? mA : g.M - 1, (int)ocol)));
        hw[1] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(g.auxH + lo8_off(g, mB < g.M ? mB : g.M - 1, (int)ocol)));
        lw[0] = __builtin_nontemporal_load(reinterpret_cast<const u32x2_t*>(lo_in + lo8_off(g, mA < g.M ? mA : g.M - 1, (int)ocol)));
        lw[1] = __builtin_nontemporal_load(reinterpret_cast<const u32x2_t*>(lo_in + lo8_off(g, mB < g.M ? mB : g.M - 1, (int)ocol)));
    };
    float bv[4][4];
    pp_col_consts(bv, g.bias != nullptr ? g.bias + nb : nullptr, nullptr, lq);
#pragma unroll
    for (int i = 0; i < 2; ++i) side(i, hw[0][i], lw[0][i]);
    // bias into the accumulators while the first half's rows are in flight; its registers are free before the second half is requested
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{acc[i][j][0] + bv[j][0], acc[i][j][1] + bv[j][1], acc[i][j][2] + bv[j][2], acc[i][j][3] + bv[j][3]};
#pragma unroll
    for (int i = 0; i < 2; ++i) side(2 + i, hw[1][i], lw[1][i]);
    auto row_block = [&](int i, const u32x4_t (&hws)[2], const u32x2_t (&lws)[2]) {
        // un-swap the loaded words: this lane's own row, column halves 0 / 1
        uint4 hw[2] = {make_uint4(hws[0][0], hws[0][1], hws[0][2], hws[0][3]), make_uint4(hws[1][0], hws[1][1], hws[1][2], hws[1][3])};
        pp_pair_swap(hw[0], hw[1], lo);
        uint2 lw[2];
        {
            const unsigned rx = pp_ror8(lo ? lws[1][0] : lws[0][0]), ry = pp_ror8(lo ? lws[1][1] : lws[0][1]);
            lw[0] = make_uint2(lo ? lws[0][0] : rx, lo ? lws[0][1] : ry);
            lw[1] = make_uint2(lo ? rx : lws[1][0], lo ? ry : lws[1][1]);
        }
        uint4 oh[2];
        uint2 ol[2];
        f32x2v s1v = {0.f, 0.f}, s2v = {0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            float x[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const unsigned hword = (e >> 1) == 0 ? hw[k].x : ((e >> 1) == 1 ? hw[k].y : ((e >> 1) == 2 ? hw[k].z : hw[k].w));
                const unsigned lword = (e >> 2) == 0 ? lw[k].x : lw[k].y;
                const float hf = h2f((bf16_t)((e & 1) ? (hword >> 16) : (hword & 0xFFFF)));
                // residual = hi (1 + (q - 128) 2^-18) = hi * (q 2^-18 + (1 - 2^-11)), added to the accumulator by the same fma
                const float t = __builtin_fmaf((float)((lword >> (8 * (e & 3))) & 0xFF), 0x1p-18f, 1.0f - 0x1p-11f);
                x[e] = __builtin_fmaf(hf, t, acc[i][2 * k + (e >> 2)][e & 3]);
            }
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                const f32x2v xv = {x[e], x[e + 1]};
                s1v += xv;
                s2v = xv * xv + s2v;
            }
            oh[k] = make_uint4(pack2<true>(x[0], x[1]), pack2<true>(x[2], x[3]), pack2<true>(x[4], x[5]), pack2<true>(x[6], x[7]));
#ifdef ABL_NO_LO
            ol[k] = make_uint2(oh[k].x, oh[k].y);
#else
            // byte of the new value: (x / hi' - 1) 2^18 + 128, with 1 - hi' / x for x / hi' - 1 (they differ by the square of a number
            // below 2^-11: 1/16 of a step) -- one reciprocal of the fp32 value, one fma on the f16 hi' itself, one fma, round, convert
            unsigned ob[2] = {0u, 0u};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const unsigned hword = e < 2 ? oh[k].x : (e < 4 ? oh[k].y : (e < 6 ? oh[k].z : oh[k].w));
                const float hn = h2f((bf16_t)((e & 1) ? (hword >> 16) : (hword & 0xFFFF)));
                const float eps = __builtin_fmaf(-hn, __builtin_amdgcn_rcpf(x[e]), 1.0f);
                ob[e >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf(eps, 0x1p18f, 128.f), e & 3, ob[e >> 2]);      // (rounds to nearest even, saturates: tools/ablate/cvt_u8_probe.hip)
            }
            ol[k] = make_uint2(ob[0], ob[1]);
#endif
        }
        float s1 = s1v.x + s1v.y, s2 = s2v.x + s2v.y;
        // this 64-column slice's (sum, sum of squares) of row 16 i + l15: the four lane groups hold 16 columns each
#ifndef ABL_NO_STATS
        s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
        s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
        const int mrow = mfirst + 16 * i;
        if (lq == 0 && mrow < g.M) *reinterpret_cast<float2*>(g.rowpart + ((size_t)mrow * (g.N >> 6) + (nb >> 6)) * 2) = make_float2(s1, s2);
#else
        if (s1 + s2 == 12345.678f) g.rowpart[0] = s1;
#endif
        // lane pairs swap one half each: stores of whole 128-byte (hi) / 64-byte (lo) row segments, rows 16 i + r8 and 16 i + 8 + r8
        pp_pair_swap(oh[0], oh[1], lo);
        {
            const unsigned rx = pp_ror8(lo ? ol[1].x : ol[0].x), ry = pp_ror8(lo ? ol[1].y : ol[0].y);
            ol[0] = make_uint2(lo ? ol[0].x : rx, lo ? ol[0].y : ry);
            ol[1] = make_uint2(lo ? rx : ol[1].x, lo ? ry : ol[1].y);
        }
#ifdef ABL_NO_ST
        const int mA = (oh[0].x ^ oh[1].y ^ ol[0].x ^ ol[1].y) == 0x12345u ? mb + 16 * i + r8 : g.M;
#else
        const int mA = mb + 16 * i + r8;
#endif
        if (mA < g.M) {
            *reinterpret_cast<uint4*>(g.outH + lo8_off(g, mA, (int)ocol)) = oh[0];      // (a plain store: the next GEMM reads this image right away)
#ifndef ABL_NO_LO
            v3_st<uint2>(lo_out + lo8_off(g, mA, (int)ocol), ol[0]);
#endif
        }
        if (mA + 8 < g.M) {
            *reinterpret_cast<uint4*>(g.outH + lo8_off(g, mA + 8, (int)ocol)) = oh[1];
#ifndef ABL_NO_LO
            v3_st<uint2>(lo_out + lo8_off(g, mA + 8, (int)ocol), ol[1]);
#endif
        }
    };
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
            if (2 * gi + i < RB) row_block(2 * gi + i, hw[gi][i], lw[gi][i]);
        if (gi + 2 < NG) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
                if (2 * (gi + 2) + i < RB) side(2 * (gi + 2) + i, hw[gi + 2][i], lw[gi + 2][i]);
        }
    }
}

// GELU' epilogue (dX of fc2 times gelu'(saved pre-activation), student backward) straight from the accumulators: the pre-activation rows
// are fetched up front in the pair-swapped whole-line layout (16 bytes per lane, 8 rows x 128 bytes per instruction -- the staged form
// read and wrote 8 bytes per lane, 128 memory instructions per wave and tile instead of 32), un-swapped in registers, and the products
// leave the same way.  No LDS.
template <bool F16, int RB>
__device__ __forceinline__ void pp_epilogue_dgelu_direct(const GemmArgs& g, const f32x4_t (&acc)[8][4], int mb, int nb, int lane) {
    const int lq = lane >> 4;
    int lrow = lane & 15;
    asm volatile("" : "+v"(lrow));
    const bool lo = lrow < 8;
    const int r8 = lrow & 7;
    const size_t ocol = (size_t)nb + 32 * (lrow >> 3) + 8 * lq;
    u32x4_t hw[RB][2];
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        const int mA = mb + 16 * i + r8, mB = mA + 8;
        hw[i][0] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(g.auxH + (size_t)(mA < g.M ? mA : g.M - 1) * g.ldc + ocol));
        hw[i][1] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(g.auxH + (size_t)(mB < g.M ? mB : g.M - 1) * g.ldc + ocol));
    }
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        uint4 h[2] = {make_uint4(hw[i][0][0], hw[i][0][1], hw[i][0][2], hw[i][0][3]), make_uint4(hw[i][1][0], hw[i][1][1], hw[i][1][2], hw[i][1][3])};
        pp_pair_swap(h[0], h[1], lo);      // -> this lane's own row, column halves 0 / 1
        uint4 o[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            unsigned ow[4];
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const unsigned hword = w == 0 ? h[k].x : (w == 1 ? h[k].y : (w == 2 ? h[k].z : h[k].w));
                const f32x2v ga = gelu_fast_grad2(f32x2v{to_f32<F16>((bf16_t)(hword & 0xFFFF)), to_f32<F16>((bf16_t)(hword >> 16))});
                const f32x4_t& a = acc[i][2 * k + (w >> 1)];
                ow[w] = pack2<F16>(a[2 * (w & 1)] * ga.x, a[2 * (w & 1) + 1] * ga.y);
            }
            o[k] = make_uint4(ow[0], ow[1], ow[2], ow[3]);
        }
        pp_pair_swap(o[0], o[1], lo);
        const int mA = mb + 16 * i + r8;
        if (mA < g.M) v3_st<uint4>(g.outH + (size_t)mA * g.ldc + ocol, o[0]);
        if (mA + 8 < g.M) v3_st<uint4>(g.outH + (size_t)(mA + 8) * g.ldc + ocol, o[1]);
    }
}

// fp32 pair swap: a0 / a1 = this lane's row, blocks (0, 1) / (2, 3) as 8 floats each -> (row l15 & 7, blocks 2 hk, 2 hk + 1), (row 8 + (l15 & 7), same blocks)
__device__ __forceinline__ void pp_pair_swap32(float (&a0)[8], float (&a1)[8], bool lo) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float r = __uint_as_float(pp_ror8(__float_as_uint(lo ? a1[e] : a0[e])));
        a0[e] = lo ? a0[e] : r;
        a1[e] = lo ? r : a1[e];
    }
}
// fp32 output (EPI_F32: acc * alpha + bias; plain EPI_F32_RESID: residual + acc + bias, in place or not) straight from the accumulators:
// residual rows requested two row-block groups ahead in the swapped layout, no LDS.  The staged form cost 18 (fp32 out) / 29 us (read-
// modify-write) per 256^2 tile.
template <int EPI, int RB>
__device__ __forceinline__ void pp_epilogue_f32_direct(const GemmArgs& g, const f32x4_t (&acc)[8][4], int mb, int nb, int lane) {
    constexpr bool RES = EPI == EPI_F32_RESID;
    const int lq = lane >> 4;
    int lrow = lane & 15;
    asm volatile("" : "+v"(lrow));
    const bool lo = lrow < 8;
    const int r8 = lrow & 7;
    const size_t ocol = (size_t)nb + 16 * (lrow >> 3) + 4 * lq;      // + 32 for the second float4
    constexpr int NG = (RB + 1) / 2;
    f32x4_t rw[RES ? NG : 1][2][4];      // [group][row block of the group][row A / row B x first / second float4]
    auto side = [&](int i, f32x4_t (&r)[4]) {
        const int mA = mb + 16 * i + r8, mB = mA + 8;
        const float* pA = g.resF + (size_t)(mA < g.M ? mA : g.M - 1) * g.ldc + ocol;
        const float* pB = g.resF + (size_t)(mB < g.M ? mB : g.M - 1) * g.ldc + ocol;
        r[0] = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(pA));
        r[1] = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(pA + 32));
        r[2] = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(pB));
        r[3] = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(pB + 32));
    };
    float bv[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
        if (g.bias != nullptr) b = *reinterpret_cast<const float4*>(g.bias + nb + PP_COL32(j, lq));
        bv[j][0] = b.x; bv[j][1] = b.y; bv[j][2] = b.z; bv[j][3] = b.w;
    }
    if constexpr (RES) {
#pragma unroll
        for (int gi = 0; gi < 2 && gi < NG; ++gi)
#pragma unroll
            for (int i = 0; i < 2; ++i)
                if (2 * gi + i < RB) side(2 * gi + i, rw[gi][i]);
    }
    const float alpha = RES ? 1.f : g.alpha;
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
            const int i = 2 * gi + ii;
            if (i >= RB) continue;
            float a0[8], a1[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                a0[e] = __builtin_fmaf(acc[i][e >> 2][e & 3], alpha, bv[e >> 2][e & 3]);
                a1[e] = __builtin_fmaf(acc[i][2 + (e >> 2)][e & 3], alpha, bv[2 + (e >> 2)][e & 3]);
            }
            pp_pair_swap32(a0, a1, lo);
            if constexpr (RES) {      // ((product + bias) + residual: the variants that still stage add in another order -- last-bit differences)
                const f32x4_t (&r)[4] = rw[gi][ii];
#pragma unroll
                for (int e = 0; e < 8; ++e) { a0[e] += r[e >> 2][e & 3]; a1[e] += r[2 + (e >> 2)][e & 3]; }
            }
            const int mA = mb + 16 * i + r8;
            if (mA < g.M) {
                float* p = g.outF + (size_t)mA * g.ldc + ocol;
                v3_st<float4>(p, make_float4(a0[0], a0[1], a0[2], a0[3]));
                v3_st<float4>(p + 32, make_float4(a0[4], a0[5], a0[6], a0[7]));
            }
            if (mA + 8 < g.M) {
                float* p = g.outF + (size_t)(mA + 8) * g.ldc + ocol;
                v3_st<float4>(p, make_float4(a1[0], a1[1], a1[2], a1[3]));
                v3_st<float4>(p + 32, make_float4(a1[4], a1[5], a1[6], a1[7]));
            }
        }
        if constexpr (RES) {
            if (gi + 2 < NG) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    if (2 * (gi + 2) + i < RB) side(2 * (gi + 2) + i, rw[gi + 2][i]);
            }
        }
    }
}

template <int EPI>
struct V3Consts {  // per-lane bias values, fetched before the K loop so their latency is off the epilogue's critical path
    static constexpr bool kStaged16 = (EPI == EPI_BF16 || EPI == EPI_GELU);
    float bv[kStaged16 ? 4 : 1][4];
};
template <int EPI>
__device__ __forceinline__ void v3_load_consts(V3Consts<EPI>& c, const GemmArgs& g, int nb, int lane) {
    if constexpr (EPI == EPI_QKV) {
        c.bv[0][0] = 0.f;
    } else if constexpr (V3Consts<EPI>::kStaged16) {
        pp_col_consts(c.bv, g.bias != nullptr ? g.bias + nb : nullptr, nullptr, lane >> 4);
    } else {
        float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
        if (g.bias != nullptr) b = *reinterpret_cast<const float4*>(g.bias + nb + (lane & 15) * 4);
        c.bv[0][0] = b.x; c.bv[0][1] = b.y; c.bv[0][2] = b.z; c.bv[0][3] = b.w;
    }
}

template <int EPI, bool F16, int GBM, int RB = 8, bool F8 = false>
__device__ __forceinline__ void pp_epilogue(const GemmArgs& g, const f32x4_t (&acc)[8][4], const V3Consts<EPI>& cc,
                                            unsigned char* wl, int mb, int nb, int lane, unsigned long long* gxt = nullptr) {
    // mb = first row of this wave's 128 x 64 sub-tile, nb = its first column
    constexpr bool GB = GBM == 1;   // 1: row-group bias, 2: two-term weights (plain epilogue), 3: folded LayerNorm (producer / consumer by EPI)
    constexpr bool LN = GBM == 3 || (GBM >= 5 && GBM <= 7);      // (5: producer reading f16 + f16 planes; 6 / 7: f16 + byte planes, writing planes / fp32)
    constexpr bool ROWS_ONLY = GBM >= 3;      // head-split epilogue: row-major q / k / v only (3: folded LayerNorm, 4: the plain encoder form)
    const int l15 = lane & 15, lq = lane >> 4;
    if constexpr (EPI == EPI_BF16 || EPI == EPI_GELU) {
        // 16-byte stores straight from the accumulators (PP_COL column order): no staging, no wave barrier
#pragma unroll
        for (int pass = 0; pass < (EPI == EPI_GELU ? 2 : 1); ++pass) {
            bf16_t* out = (EPI == EPI_GELU && pass == 1) ? g.outH2 : g.outH;
            if (out == nullptr) continue;
            // (opaque to the optimiser: otherwise the per-lane output address is hoisted out of the persistent tile loop as a 64-bit value,
            //  spilled around it in the register-tight variants and reloaded -- behind a vmcnt(0) -- at every tile's store phase)
            int lrow = l15;
            asm volatile("" : "+v"(lrow));
            const int r8 = lrow & 7, hk = lrow >> 3;
            // (slab-major output: the wave's 64 columns are slab nb / 64, rows 128 bytes apart)
            const int ldo = g.c_slab ? 64 : g.ldc;
            bf16_t* const ob = g.c_slab ? out + (size_t)(nb >> 6) * g.M * 64 : out + nb;
            const PPSinkRowsT<F8> sink{ob + (size_t)(mb + r8) * ldo + 32 * hk + 8 * lq, ob + (size_t)(mb + lrow) * ldo + 8 * lq,
                                       8 * ldo, g.M - mb - r8, g.M - mb - lrow, lrow < 8,
                                       (F8 && g.out_e4m3) ? 2 * g.N - (nb + 32 * hk + 8 * lq) : 0};
            if constexpr (GB) {      // evaluation-mode encoder only (no saved pre-activation: one pass)
                const float *rA, *rB;
                const int bnd = gb_split(g, mb, rA, rB);
                if (EPI == EPI_GELU && pass == 1) pp_stage16_gb<F16, 1>(sink, acc, cc.bv, rA + nb, rB + nb, bnd, l15, lq);
                else pp_stage16_gb<F16, 0>(sink, acc, cc.bv, rA + nb, rB + nb, bnd, l15, lq);
            } else if constexpr (LN) {      // consumer: Linear(LayerNorm(x)) from the raw stream's product (no saved pre-activation either)
                if (EPI == EPI_GELU && pass == 1) pp_stage16_ln<F16, 1, false, RB>(sink, acc, cc.bv, g.colS + nb, g.rowstat, mb, g.M, l15, lq);
                else pp_stage16_ln<F16, 0, false, RB>(sink, acc, cc.bv, g.colS + nb, g.rowstat, mb, g.M, l15, lq);
            } else
            if (EPI == EPI_GELU && pass == 1) pp_stage16<F16, 1, RB>(sink, acc, cc.bv);
            else if (EPI == EPI_GELU && F16 && g.bwd_bf16) pp_stage16<false, 0, RB>(sink, acc, cc.bv);  // pre-activation for the bf16 backward
            else pp_stage16<F16, 0, RB>(sink, acc, cc.bv);
#ifdef GX_TRACE
            if (gxt != nullptr) gxt[4] = __builtin_amdgcn_s_memrealtime();
#endif
        }
        return;
    }
    if constexpr (EPI == EPI_QKV) {
        const int D = g.heads * 64;
        const int which = nb / D, h = (nb - which * D) >> 6;
        bf16_t* row_dst = which == 0 ? g.q : (which == 1 ? g.k : g.v);
        if constexpr (ROWS_ONLY) {
            // (folded-LayerNorm consumer / plain encoder form: row-major q / k / v only -- no second biased q, no transposed copies.)
            // Stored straight from the accumulators: a lane's 8 columns of a half are 16 contiguous bytes of one head row.
            if (row_dst == nullptr) return;      // (the row-major V is only needed by the backward: inference passes v = NULL)
            int lrow = l15;
            asm volatile("" : "+v"(lrow));
            const int m_first = mb + (lrow & 7), b0 = m_first / g.seq;
            const PPSinkHeads sink{row_dst + (size_t)h * g.seq * 64 + 32 * (lrow >> 3) + 8 * lq, b0, m_first - b0 * g.seq, g.seq, g.heads * g.seq,
                                   g.M - m_first, lrow < 8};
            if constexpr (LN) {
                float bv[4][4];
                pp_stage16_ln<F16, 0, true, RB>(sink, acc, bv, g.colS + nb, g.rowstat, mb, g.M, l15, lq, g.bias + nb);
            } else {
                float bv[4][4];
                pp_col_consts(bv, g.bias != nullptr ? g.bias + nb : nullptr, (which == 0 && g.pu != nullptr) ? g.pu + h * 64 : nullptr, lq);
                pp_stage16<F16, 0, RB>(sink, acc, bv);
            }
            return;
        }
        bf16_t* tr_dst = which == 0 ? g.qt : (which == 1 ? g.kt : g.vt);
        const PPSinkLds lsink{wl, l15, lq};
        const int npass = (which == 0 && g.q2 != nullptr) ? 2 : 1;
        for (int pass = 0; pass < npass; ++pass) {
            const float* extra = nullptr;
            if (which == 0 && g.pu != nullptr) extra = (pass == 0 ? g.pu : g.pv) + h * 64;
            bf16_t* rd = pass == 0 ? row_dst : g.q2;
            bf16_t* td = pass == 0 ? tr_dst : g.q2t;
            float bv[4][4];
            pp_col_consts(bv, g.bias != nullptr ? g.bias + nb : nullptr, extra, lq);
            if constexpr (GB) {
                const float *rA, *rB;
                const int bnd = gb_split(g, mb, rA, rB);
                pp_stage16_gb_k<F16, 0>(lsink, acc, bv, rA + nb, rB + nb, bnd, l15, lq);
            } else {
                pp_stage16<F16, 0, RB>(lsink, acc, bv);
            }
            // Tensors that only the bf16 backward reads (row-major V; Q^T, K^T, (q+v)^T) are emitted as bf16 when g.bwd_bf16 is
            // set: converted from the staged f16 tile on the way out (same double rounding as a later in-place conversion,
            // without the extra pass over HBM); V^T and the row-major q / k stay f16.
            // (a row-major V that the forward itself consumes -- no V^T requested: the encoder's attention transposes it in LDS -- stays f16)
            const bool row_bf = F16 && g.bwd_bf16 && which == 2 && pass == 0 && g.vt != nullptr;
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
                        if (m < g.M && (rb + u) * 8 + (lane >> 3) < 16 * RB) {
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
                    if (m < g.M && row < 16 * RB) {
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
                    if (m < g.M && row < 16 * RB) {
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
    if constexpr ((EPI == EPI_F32 || EPI == EPI_F32_RESID) && GBM == 0) {
        pp_epilogue_f32_direct<EPI, RB>(g, acc, mb, nb, lane);
        return;
    }
    if constexpr (EPI == EPI_DGELU && GBM == 0) {
        pp_epilogue_dgelu_direct<F16, RB>(g, acc, mb, nb, lane);
        return;
    }
    if constexpr (EPI == EPI_F32_RESID && GBM == 6) {      // byte planes in -> byte planes out: straight from the accumulators
        pp_epilogue_lo8_direct<RB>(g, const_cast<f32x4_t (&)[8][4]>(acc), mb, nb, lane);
        return;
    }
    // fp32-staged epilogues (two passes of 64 rows): EPI_F32, EPI_F32_RESID, EPI_F32_BF16, EPI_GELU32, EPI_DGELU
    // 16 lanes x 16 B = one 256-B fp32 row; the lane's 4 columns (and so its bias) are the same for every row.  Four batches
    // of 8 rows-per-lane; the residual / saved pre-activation of batch k+1 is requested before batch k is stored, so its
    // latency hides under the stores (loads placed between stores would each cost a full wait, see pp_col_consts).
    const int c4 = lane & 15, n = nb + c4 * 4;
    float4 b = make_float4(cc.bv[0][0], cc.bv[0][1], cc.bv[0][2], cc.bv[0][3]);
    float4 bB = b;
    int mbnd = 0x7fffffff;
    if constexpr (GB) {
        const float *rA, *rB;
        mbnd = mb + gb_split(g, mb, rA, rB);
        const float4 xa = *reinterpret_cast<const float4*>(rA + n), xb = *reinterpret_cast<const float4*>(rB + n);
        bB = make_float4(b.x + xb.x, b.y + xb.y, b.z + xb.z, b.w + xb.w);
        b = make_float4(b.x + xa.x, b.y + xa.y, b.z + xa.z, b.w + xa.w);
    }
    const int mend = mb + 16 * RB;
    // (the producer's residual form is part of the kernel variant: GBM 3 = fp32, 5 = f16 + f16 planes, 6 = f16 + byte planes -- see v3_side_load)
    constexpr int SM = (LN && EPI == EPI_F32_RESID) ? (GBM >= 6 ? 3 : (GBM == 5 ? 2 : 1)) : 0;
#ifdef ABL_NO_EPI
    if constexpr (SM >= 2) return;
#endif
    V3Side<EPI> s0, s1;
    v3_side_load<EPI, SM>(s0, g, mb, n, lane);
    pp_stage32<RB, true>(wl, acc, 0, l15, lq);
    __builtin_amdgcn_wave_barrier();
    v3_side_load<EPI, SM>(s1, g, mb + 32, n, lane);
    v3_store_batch<EPI, F16, SM>(g, wl, s0, b, mb, 0, n, c4, lane, bB, mbnd, mend);
    v3_side_load<EPI, SM>(s0, g, mb + 64, n, lane);
    v3_store_batch<EPI, F16, SM>(g, wl, s1, b, mb + 32, 32, n, c4, lane, bB, mbnd, mend);
    __builtin_amdgcn_wave_barrier();
    pp_stage32<RB, true>(wl, acc, 1, l15, lq);
    __builtin_amdgcn_wave_barrier();
    v3_side_load<EPI, SM>(s1, g, mb + 96, n, lane);
    v3_store_batch<EPI, F16, SM>(g, wl, s0, b, mb + 64, 0, n, c4, lane, bB, mbnd, mend);
    v3_store_batch<EPI, F16, SM>(g, wl, s1, b, mb + 96, 32, n, c4, lane, bB, mbnd, mend);
    __builtin_amdgcn_wave_barrier();
}

template <bool F16> __device__ __forceinline__ f32x4_t mfma16t(s16x8_t a, s16x8_t b, f32x4_t c) {
    if (F16)
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

typedef __attribute__((address_space(3))) void* lds_ptr_t;
// RB = 16-row blocks per wave row: 8 -> 256-row tiles; 7 -> 224-row tiles (the eighth block's reads, MFMAs and stores are skipped; its
// LDS rows are still filled so that every wave keeps the same DMA count for the counted waits).  With M = 38080 tokens on 256 CUs the
// N = 768 / 2304 GEMMs are 447 / 1341 tiles of 256 rows = 1.75 / 5.24 rounds, i.e. 2 / 6 rounds with 13 % of the last ones empty; as
// 510 / 1530 tiles of 224 rows they are 1.99 / 5.98 rounds of tiles that are 12.5 % shorter -- launch_gemm picks the cheaper height.
typedef int i32x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ i32x8_t cat32(s16x8_t a, s16x8_t b) {
    typedef int i32x4_t __attribute__((ext_vector_type(4)));
    const i32x4_t x = __builtin_bit_cast(i32x4_t, a), y = __builtin_bit_cast(i32x4_t, b);
    return i32x8_t{x[0], x[1], x[2], x[3], y[0], y[1], y[2], y[3]};
}
// The F8 variants pin their accumulators: with the compiler's untied MFMA forms (vdst != srcC allowed) and 250 of 256 registers in use
// the allocator moved accumulator tuples between K tiles and spilled some (scratch reloads behind s_waitcnt vmcnt(0): the end of the DMA
// pipeline).  Tied inline-asm forms leave it nothing to move.  Back-to-back accumulation into the same vdst needs no wait states; the
// epilogue's first VALU read of an accumulator is kept >= 18 cycles behind the last MFMA by hand (pp kernel, after the K loop).
__device__ __forceinline__ void mfma16_f16_tied(f32x4_t& c, s16x8_t a, s16x8_t b) {
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma128_e4m3_tied(f32x4_t& c, i32x8_t a, i32x8_t b, int sc) {
    asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]" : "+v"(c) : "v"(a), "v"(b), "v"(sc));
}
template <int EPI, bool F16, int GB = 0, int RB = 8, bool F8 = false>
__global__ __launch_bounds__(512) void gemm_nt_pp_kernel(const GemmArgs g) {
    constexpr int TM = 32 * RB;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds3[];
#ifdef GX_TRACE
    const unsigned long long gx_top = __builtin_amdgcn_s_memrealtime();
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int ntn = g.N / V3_T, ntm = (g.M + TM - 1) / TM, nwg = ntm * ntn;
    // Persistent form: gridDim.x = number of CUs, every workgroup walks the tiles blockIdx.x, blockIdx.x + gridDim.x, ... (gridDim.x is
    // a multiple of 8, so a workgroup's tiles stay on its XCD and the L2 grouping below is unchanged).  Measured per 256^2 tile with
    // one launch per tile: 1.6 us between the last store of a workgroup and the first instruction of the next one on that CU plus
    // 0.5 us of kernel-argument / address setup (tools/epi_gaps.py) against a 17 us K = 768 main loop.
    const int tstep = g.persist ? (int)gridDim.x : nwg;
    if (g.stagger > 0) {
        // (stagger >> 16 = number of phase groups P, default 2; workgroup i of an XCD starts (i mod P) / P * ticks late)
        const int P = (g.stagger >> 16) ? (g.stagger >> 16) : 2, ticks = g.stagger & 0xFFFF;
        const unsigned long long dly = (unsigned long long)(((blockIdx.x >> 3) % P) * ticks / P) * (P == 2 ? 2 : 1);
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__builtin_amdgcn_s_memrealtime() - t0 < dly) __builtin_amdgcn_s_sleep(16);
    }
    // Dynamic walk (g.tile_ctr): the logical tiles tl with tl & 7 == x are XCD x's contiguous range (xcd_remap); index k of that range
    // comes from the per-XCD counter.  The counter for tile i + 1 is read at the top of tile i (one lane) and handed to the other waves
    // through one LDS word after the prologue barrier, so the ~1 us of a device-scope atomic is never waited for.
    __shared__ int s_next_tile;
    int* const dyn_ctr = g.tile_ctr != nullptr ? g.tile_ctr + 16 * (blockIdx.x & 7) : nullptr;
    int tl = blockIdx.x;
    if (dyn_ctr != nullptr) {
        if (tid == 0) s_next_tile = atomicAdd(dyn_ctr, 1);
        __syncthreads();
        tl = __builtin_amdgcn_readfirstlane(s_next_tile) * 8 + (blockIdx.x & 7);
    }
    // Epilogues that store straight from the accumulators leave the LDS alone: the next tile's first operand stage is requested BEFORE
    // the epilogue (its DMA lands under the stores) and the end-of-tile workgroup barrier goes away.
    constexpr bool DIRECT_EPI = (EPI == EPI_BF16 || EPI == EPI_GELU || (EPI == EPI_QKV && GB >= 3));
    constexpr bool F32L = (EPI == EPI_F32 || EPI == EPI_F32_RESID) && GB == 0;      // fp32 straight from the accumulators: PP_COL32 column order
    auto tile_mn = [&](int tl_, int& m0_, int& n0_) {
        const int t = xcd_remap(tl_, nwg);
        const int GM = g.group_m;
        const int group_size = GM * ntn, gid = t / group_size, first_m = gid * GM;
        const int gm = (ntm - first_m) < GM ? (ntm - first_m) : GM;
        const int tin = t - gid * group_size;
        m0_ = (first_m + tin % gm) * TM;
        n0_ = (tin / gm) * V3_T;
    };
    bool primed = false;      // this tile's K-tile-0 DMA was issued by the previous tile
    int m0n = 0, n0n = 0;
    while (tl < nwg) {
    int m0, n0;
    if (primed) { m0 = m0n; n0 = n0n; } else tile_mn(tl, m0, n0);
    const int nk = g.K / BK;

    const int rows_a = (g.M - m0) < TM ? (g.M - m0) : TM;
    // (slab-major A, GemmArgs.a_slab: row pitch 128 bytes, K tile kt at kt * M * 128; rows past M read the next slab -- they only reach
    //  accumulator rows that are never stored -- and nothing is read past the last slab's end)
    const bool a_slab = (GB == 1 || GB == 2 || GB == 8) ? false : g.a_slab != 0;      // (the evaluation-mode variants have no register to spare for it)
    const int a_ld2 = a_slab ? 128 : g.lda * 2, a_kst = a_slab ? g.M * 128 : BK * 2;
    const __amdgpu_buffer_rsrc_t ra = a_slab
        ? __builtin_amdgcn_make_buffer_rsrc((void*)(g.A + (size_t)m0 * 64), 0, (unsigned)((size_t)(g.K / BK) * g.M * 128 - (size_t)m0 * 128), 0x00020000)
#ifdef ABL_A_ALIAS   /* experiment (tools/ablate): every row panel reads the first 1024 rows of A -- the operand then lives in L2, results are wrong */
        : __builtin_amdgcn_make_buffer_rsrc((void*)(g.A + (size_t)(m0 % 1024) * g.lda), 0, rows_a * g.lda * 2, 0x00020000);
#else
        : __builtin_amdgcn_make_buffer_rsrc((void*)(g.A + (size_t)m0 * g.lda), 0, rows_a * g.lda * 2, 0x00020000);
#endif
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)(g.B + (size_t)n0 * g.ldb), 0, V3_T * g.ldb * 2, 0x00020000);
    // DMA pieces of this wave: slot piece index p = 2 wave + e.  Slot 0 = A rows of half 0 (wave-row * 128 + 0..63), slot 3 = A rows
    // of half 1, slot 1 = B rows of half 0 (wave-column * 64 + 0..31), slot 2 = B rows of half 1
    const int prow = lane >> 3, pch = lane & 7;
    int vo[4][2];      // lane byte offsets into the A / B row panels, [slot][e]
    int ld_[4][2];     // LDS byte offsets (stage bit 15 added at issue)
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
            // LDS row wm * 128 + r of the A stage holds tile row wm * 16 RB + r (r >= 16 RB: a row nobody reads)
            const int grow = is_b ? (F32L ? pp_brow_src32(row) : pp_brow_src(row)) : (row >> 7) * (16 * RB) + (row & 127);
            vo[sl][e] = grow * (is_b ? g.ldb * 2 : a_ld2) + cl * 16;
            ld_[sl][e] = (is_b ? 65536 : 0) + rows[sl] * 128;
        }
    }
#define PP_DMA_R(RA_, RB_, SL, KT)                                                                                        \
    {                                                                                                                     \
        const int kt_ = ((GB == 2 || GB == 8) && !F8 && (KT) >= g.k_wrap) ? (((SL) == 0 || (SL) == 3) ? (KT) - g.k_wrap : (KT) + g.b_skip) : (KT); \
        const int so_ = kt_ * (((SL) == 1 || (SL) == 2) ? BK * 2 : a_kst), st_ = ((KT) & 1) << 15;                        \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(((SL) == 1 || (SL) == 2) ? RB_ : RA_, (lds_ptr_t)(lds3 + st_ + ld_[SL][0]), 16, vo[SL][0], so_, 0, 0); \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(((SL) == 1 || (SL) == 2) ? RB_ : RA_, (lds_ptr_t)(lds3 + st_ + ld_[SL][1]), 16, vo[SL][1], so_, 0, 0); \
    }
#define PP_DMA(SL, KT) PP_DMA_R(ra, rb, SL, KT)
    f32x4_t acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    // fragment addresses: every block's rows are (lane & 15) + a multiple of 16, so one swizzle serves all of them
    const int l15 = lane & 15, lq = lane >> 4, sw = (l15 >> 1) & 7;
    int aaddr[2], baddr[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int c = ((4 * ks + lq) ^ sw) << 4;
        aaddr[ks] = wm * 16384 + l15 * 128 + c;
        baddr[ks] = 65536 + wn * 8192 + l15 * 128 + c;
    }
    V3Consts<EPI> cc;
    v3_load_consts<EPI>(cc, g, n0 + wn * 64, lane);

    s16x8_t fa[4][2], fb[2][2][2];   // fa[ii][ks]: 4 row blocks of the current A half; fb[set][jj][ks]: 2 column blocks of a B half
#define PP_RD_A(IH)                                                                                                       \
    _Pragma("unroll") for (int ii = 0; ii < ((IH) == 1 ? RB - 4 : 4); ++ii)                                               \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                                  \
            fa[ii][ks] = *reinterpret_cast<const s16x8_t*>(lds3 + aaddr[ks] + (4 * (IH) + ii) * 2048);
#define PP_RD_B(SET, JH, XOR)                                                                                             \
    _Pragma("unroll") for (int jj = 0; jj < 2; ++jj)                                                                      \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                                  \
            fb[SET][jj][ks] = *reinterpret_cast<const s16x8_t*>(lds3 + (baddr[ks] ^ (XOR)) + (2 * (JH) + jj) * 2048);
#define PP_MFMA(IH, JH, SET)                                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                                                    \
    __builtin_amdgcn_s_barrier();                                                                                         \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                                    \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                                      \
        _Pragma("unroll") for (int ii = 0; ii < ((IH) == 1 ? RB - 4 : 4); ++ii)                                           \
            _Pragma("unroll") for (int jj = 0; jj < 2; ++jj) {                                                            \
                if constexpr (F8) mfma16_f16_tied(acc[4 * (IH) + ii][2 * (JH) + jj], fb[SET][jj][ks], fa[ii][ks]);           \
                else acc[4 * (IH) + ii][2 * (JH) + jj] = mfma16t<F16>(fb[SET][jj][ks], fa[ii][ks], acc[4 * (IH) + ii][2 * (JH) + jj]); \
            }                                                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                                                    \
    __builtin_amdgcn_s_barrier();                                                                                         \
    __builtin_amdgcn_sched_barrier(0);
    // fp8 K tile (F8): the two 16-byte fragments of a row block are one 32-byte e4m3 operand
#define PP_MFMA8(IH, JH, SET)                                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                                                    \
    __builtin_amdgcn_s_barrier();                                                                                         \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                                    \
    _Pragma("unroll") for (int ii = 0; ii < ((IH) == 1 ? RB - 4 : 4); ++ii)                                               \
        _Pragma("unroll") for (int jj = 0; jj < 2; ++jj)                                                                  \
            mfma128_e4m3_tied(acc[4 * (IH) + ii][2 * (JH) + jj], cat32(fb[SET][jj][0], fb[SET][jj][1]), cat32(fa[ii][0], fa[ii][1]), f8s); \
    __builtin_amdgcn_sched_barrier(0);                                                                                    \
    __builtin_amdgcn_s_barrier();                                                                                         \
    __builtin_amdgcn_sched_barrier(0);
    // one K tile; X = B register set holding this tile's half 0, Y = the other set
#define PP_TILE_(MF, T_, X, Y)                                                                                            \
    if ((T_) + 1 < nk) PP_DMA(3, (T_) + 1)                                                                                \
    PP_RD_A(0)                                                                                                            \
    MF(0, 0, X)                                                                                                           \
    if ((T_) + 2 < nk) PP_DMA(1, (T_) + 2)                                                                                \
    PP_RD_B(Y, 1, 0)                                                                                                      \
    MF(0, 1, Y)                                                                                                           \
    if ((T_) + 2 < nk) { PP_DMA(0, (T_) + 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }                           \
    else { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }                                                             \
    PP_RD_A(1)                                                                                                            \
    MF(1, 1, Y)                                                                                                           \
    if ((T_) + 2 < nk) PP_DMA(2, (T_) + 2)                                                                                \
    if ((T_) + 1 < nk) { PP_RD_B(Y, 0, 0x8000) }                                                                          \
    MF(1, 0, X)                                                                                                           \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) { aaddr[ks] ^= 0x8000; baddr[ks] ^= 0x8000; }
#define PP_TILE(T_, X, Y) PP_TILE_(PP_MFMA, T_, X, Y)
#define PP_TILE8(T_, X, Y) PP_TILE_(PP_MFMA8, T_, X, Y)

#ifdef GX_TRACE
    unsigned long long gx_t[8];
    gx_t[0] = __builtin_amdgcn_s_memrealtime();
#define GX_STAMP(i) gx_t[i] = __builtin_amdgcn_s_memrealtime();
#else
#define GX_STAMP(i)
#endif
    // prologue: all of tile 0; then the three slots of tile 1 that the steady state would have issued during tile -1
    int dyn_nxt = 0;
    if (dyn_ctr != nullptr && tid == 0) dyn_nxt = atomicAdd(dyn_ctr, 1);     // next tile's index: returns under the prologue's DMA
    if (!primed) { PP_DMA(1, 0) PP_DMA(0, 0) PP_DMA(2, 0) PP_DMA(3, 0) }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (dyn_ctr != nullptr && tid == 0) s_next_tile = dyn_nxt;     // (everybody read the previous value before this barrier; read again after the tile's last one)
    GX_STAMP(1)
    if (nk > 1) { PP_DMA(1, 1) PP_DMA(0, 1) PP_DMA(2, 1) }
    if (wm == 1) __builtin_amdgcn_s_barrier();   // second wave row: half a phase behind
    PP_RD_B(0, 0, 0)
    int it = 0;
    if constexpr (F8) {
        const int f8s = g.f8_scale;
        const int nk16 = nk - g.k8;          // even (K % 128 == 0)
        for (; it < nk16; it += 2) {
            PP_TILE(it, 0, 1)
            PP_TILE(it + 1, 1, 0)
        }
        for (; it + 1 < nk; it += 2) {
            PP_TILE8(it, 0, 1)
            PP_TILE8(it + 1, 1, 0)
        }
        if (it < nk) { PP_TILE8(it, 0, 1) }
        asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");      // last (16-pass) MFMA -> the epilogue's accumulator reads
    } else {
    for (; it + 1 < nk; it += 2) {
        PP_TILE(it, 0, 1)
        PP_TILE(it + 1, 1, 0)
    }
    if (it < nk) { PP_TILE(it, 0, 1) }
    }
    GX_STAMP(2)
    if (wm == 0) __builtin_amdgcn_s_barrier();   // pairs with the last barrier of waves 4-7: nobody reads the stages any more
    GX_STAMP(3)
    // next tile (the K loop's last barrier is behind every wave: the operand stages are free and s_next_tile holds the next index)
    int tl_next = tl + tstep;
    primed = false;
    if constexpr (DIRECT_EPI) {
        if (dyn_ctr != nullptr) tl_next = __builtin_amdgcn_readfirstlane(s_next_tile) * 8 + (blockIdx.x & 7);
        if (tl_next < nwg) {
            tile_mn(tl_next, m0n, n0n);
            const int rows_n = (g.M - m0n) < TM ? (g.M - m0n) : TM;
            const __amdgpu_buffer_rsrc_t ran = a_slab
                ? __builtin_amdgcn_make_buffer_rsrc((void*)(g.A + (size_t)m0n * 64), 0, (unsigned)((size_t)(g.K / BK) * g.M * 128 - (size_t)m0n * 128), 0x00020000)
                : __builtin_amdgcn_make_buffer_rsrc((void*)(g.A + (size_t)m0n * g.lda), 0, rows_n * g.lda * 2, 0x00020000);
            const __amdgpu_buffer_rsrc_t rbn = __builtin_amdgcn_make_buffer_rsrc((void*)(g.B + (size_t)n0n * g.ldb), 0, V3_T * g.ldb * 2, 0x00020000);
            PP_DMA_R(ran, rbn, 1, 0) PP_DMA_R(ran, rbn, 0, 0) PP_DMA_R(ran, rbn, 2, 0) PP_DMA_R(ran, rbn, 3, 0)
            primed = true;
        }
    }
#undef PP_DMA
#undef PP_DMA_R
#undef PP_RD_A
#undef PP_RD_B
#undef PP_MFMA
#undef PP_MFMA8
#undef PP_TILE
#undef PP_TILE8
#undef PP_TILE_
#ifdef GX_TRACE
    pp_epilogue<EPI, F16, GB, RB, F8>(g, acc, cc, lds3 + wave * V3_WLDS, m0 + wm * (16 * RB), n0 + wn * 64, lane, gx_t);
    GX_STAMP(5)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    GX_STAMP(6)
    if constexpr (EPI == EPI_BF16 || EPI == EPI_GELU) {
        // developer trace: stamps of every wave of workgroups 0, 100 and 300 behind the output matrix (tools/epi_trace.py)
        const int slot = blockIdx.x == 0 ? 0 : (blockIdx.x == 100 ? 1 : (blockIdx.x == 300 ? 2 : -1));
        if (slot >= 0 && lane == 0) {
            unsigned long long* tr = reinterpret_cast<unsigned long long*>((EPI == EPI_GELU ? g.outH2 : g.outH) + (size_t)g.M * g.ldc) + (slot * 8 + wave) * 8;
#pragma unroll
            for (int i = 0; i < 7; ++i) tr[i] = gx_t[i];
        }
        // every workgroup: (hardware id, first instruction, entry stamp, last store acknowledged) of wave 0 -> occupancy gaps per CU
        if (wave == 0 && lane == 0) {
            unsigned long long* wr = reinterpret_cast<unsigned long long*>((EPI == EPI_GELU ? g.outH2 : g.outH) + (size_t)g.M * g.ldc) + 256 + 4 * blockIdx.x;
            unsigned hw, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            wr[0] = ((unsigned long long)xcc << 32) | hw;
            wr[1] = gx_top; wr[2] = gx_t[0]; wr[3] = gx_t[6];
        }
    }
#else
    pp_epilogue<EPI, F16, GB, RB, F8>(g, acc, cc, lds3 + wave * V3_WLDS, m0 + wm * (16 * RB), n0 + wn * 64, lane);
#endif
    if constexpr (DIRECT_EPI) {
        tl = tl_next;                        // (no barrier: nothing of this tile is left in the LDS)
    } else if (dyn_ctr != nullptr) {
        __syncthreads();                     // (also orders the epilogue's staging reads before the next tile's DMA)
        tl = __builtin_amdgcn_readfirstlane(s_next_tile) * 8 + (blockIdx.x & 7);
    } else {
        tl += tstep;
        if (tl < nwg) __syncthreads();       // the staging areas overlap the operand stages the next tile's DMA is about to fill
    }
    }
    if (dyn_ctr != nullptr && tid == 0) {
        // last workgroup out re-arms the slot for its next user (every tile-counter atomic of a workgroup has returned before its exit atomic)
        int* const base = g.tile_ctr;
        if (atomicAdd(base + 16 * 8, 1) == (int)gridDim.x - 1) {
#pragma unroll
            for (int x = 0; x <= 8; ++x) atomicExch(base + 16 * x, 0);
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// Weight-gradient GEMM in TN form: dW[m, n] += sum_t dY[t, m] * X[t, n] with BOTH operands read in their natural row-major
// [token][feature] layout -- no transposed operand copies in HBM (the NT kernels need dY^T and X^T: 6.4 ms/step of transposes).
// The contraction index is the slow dimension of both tiles, so the MFMA fragments (8 consecutive tokens per lane) are columns of
// the LDS image: read with `ds_read_b64_tr_b16`.  Semantics measured on gfx950 (tools/ablate/trread.hip): the 16 lanes of a group
// supply 16 addresses of 8-byte chunks, taken as a [4 rows r = a>>2][4 chunks c = a&3] grid = a 4 x 16 halfword matrix M;
// lane i of the group receives column i: {M[0][i], M[1][i], M[2][i], M[3][i]}.  With rows = 4 consecutive tokens and columns =
// 16 consecutive features, two such reads give the 8-token slice of one feature per lane -- exactly the v_mfma_f32_16x16x32
// operand (lane & 15 = feature, lane >> 4 = token group of 8): lane group G reads tokens 8 G .. 8 G + 7 of the 32-token k-step.
// LDS stage = [64 tokens][256 features] per operand (512-B rows), filled by DMA; the 64-B unit u of token row k is stored at unit
// u ^ (k & 3) and its 32-B halves are swapped when (k >> 3) & 1, so that the 32 lanes of a read cycle (groups G, G + 1: token rows
// 8 G + r and 8 G + 8 + r, r = 0..3, 32 bytes each) cover four whole 64-B units = all 64 banks.
// Same 256 x 256 x 64 tiling / 8 waves (128 x 64 per wave) / two stages as the NT kernel; split-K over the tokens.
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
// LDS stage: per operand four PANELS of [64 tokens][64 features] (128-byte rows, 8 KiB each) -- a DMA piece is 8 token rows of one
// panel, so pieces (and the DMA slots built from them) follow the feature halves the ping-pong phases consume, exactly like the row
// halves of the NT kernel.  Inside a row the 32-byte block b (16 features) of token row r sits at block b ^ (r & 3) ^ ((r >> 3) & 1):
// the 32 lanes of one transposing-read cycle (two 16-lane groups: token rows 8 G + r and 8 G + 8 + r, r = 0..3, 32 bytes each) then
// cover all 64 banks.  A stage s at s * 32 KiB, B stage s at 64 KiB + s * 32 KiB.
// K loop: the ping-pong schedule of gemm_nt_pp_kernel (two wave rows half a phase apart, four 64 x 32 quadrant phases per 64-token
// K tile, B half 0 of the next tile read in P4).  DMA slots (2 pieces per wave each): A0 = A panels {0, 2}, A1 = {1, 3}, B01, B23.
// A piece spans both feature halves of the waves that read it, so a B slot is free only after P2 and the issue order differs from the
// NT kernel: P1(t): B23(t+1), P2(t): A1(t+1), P3(t): A0(t+2), P4(t): B01(t+2); `vmcnt(4)` at P3(t) retires A0 / B01 / B23 of tile
// t+1 one phase ahead of their first read (B half 0 in P4(t)), `vmcnt(6)` at P1(t) retires A1(t) two phases ahead of its read.
template <bool BF16_B>
__global__ __launch_bounds__(512) void gemm_tn_dw_kernel(const TnArgs g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds3[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    // One linear workgroup index, split-major and XCD-contiguous: the tiles of one token split run side by side on ONE XCD, so the
    // split's dY / X rows go through that L2 once for all of them (9 x 3 tiles: fabric reads 663 -> ~250 MB per launch; with the tile
    // index spread over the XCDs every dY panel was fetched by three L2s and every X panel by up to nine).
    // M, N multiples of 64: the last tile of a dimension may be partly valid.  Its out-of-range operand columns are whatever lies behind
    // them in memory (the next token's row; nothing past the buffer end) -- an output element depends on its own operand columns only,
    // and the out-of-range rows / columns of the tile are not stored.
    const int ntn = (g.N + V3_T - 1) / V3_T, ntm = (g.M + V3_T - 1) / V3_T, ntiles = ntm * ntn;
    const int L = xcd_remap(blockIdx.x, ntiles * g.ksplit);
    const int split = L / ntiles, t = L - split * ntiles;
    const int m0 = (t % ntm) * V3_T, n0 = (t / ntm) * V3_T;
    const int ktiles = (g.T + BK - 1) / BK;
    const int kt_begin = (int)(((long long)split * ktiles) / g.ksplit);
    const int kt_end = (int)(((long long)(split + 1) * ktiles) / g.ksplit);
    const int nk = kt_end - kt_begin;
    const int mw = (g.M - m0) < V3_T ? (g.M - m0) : V3_T, nw = (g.N - n0) < V3_T ? (g.N - n0) : V3_T;     // valid tile extent
    // token rows of this split; a ragged last K tile (T not a multiple of 64) ends the buffer descriptors at the last real token, the
    // rows behind it read as zeros and add nothing to dW or to the bias sums (round 4: the tail used to be two transposes + an NT GEMM)
    const int trows = (g.T - kt_begin * BK) < nk * BK ? (g.T - kt_begin * BK) : nk * BK;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)(g.A + (size_t)kt_begin * BK * g.lda + m0), 0,
                                                                        (trows - 1) * g.lda * 2 + mw * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)(g.B + (size_t)kt_begin * BK * g.ldb + n0), 0,
                                                                        (trows - 1) * g.ldb * 2 + nw * 2, 0x00020000);
    // DMA pieces of this wave: slot piece index pi = 2 wave + e -> panel list[pi >> 3], token rows 8 (pi & 7) .. + 7
    const int prow = lane >> 3, pc = lane & 7;
    int vo[4][2], ld_[4][2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int pi = 2 * wave + e, q = pi & 7, hi = pi >> 3;
        const int panels[4] = {2 * hi, hi, 2 + hi, 2 * hi + 1};      // slots: A0 {0,2}, B01 {0,1}, B23 {2,3}, A1 {1,3}
        const int r = 8 * q + prow;
        const int lc = ((((pc >> 1) ^ (r & 3) ^ ((r >> 3) & 1)) & 3) << 1) | (pc & 1);      // logical 16-B chunk held by physical chunk pc
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) {
            const bool is_b = (sl == 1 || sl == 2);
            vo[sl][e] = r * (is_b ? g.ldb : g.lda) * 2 + (panels[sl] * 64 + lc * 8) * 2;
            ld_[sl][e] = (is_b ? 65536 : 0) + panels[sl] * 8192 + q * 1024;
        }
    }
    const int kstride_a = BK * g.lda * 2, kstride_b = BK * g.ldb * 2;
#define TN_DMA(SL, KT)                                                                                                    \
    {                                                                                                                     \
        const bool b_ = ((SL) == 1 || (SL) == 2);                                                                         \
        const int so_ = (KT) * (b_ ? kstride_b : kstride_a), st_ = ((KT) & 1) << 15;                                      \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(b_ ? rb : ra, (lds_ptr_t)(lds3 + st_ + ld_[SL][0]), 16, vo[SL][0], so_, 0, 0); \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(b_ ? rb : ra, (lds_ptr_t)(lds3 + st_ + ld_[SL][1]), 16, vo[SL][1], so_, 0, 0); \
    }
    f32x4_t acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    // transposing-read lane addresses: group G = lane >> 4 reads token rows 8 G + (a >> 2) (+ 4 for the second read), a = lane & 15
    const int a = lane & 15, G = lane >> 4;
    const unsigned lbase = (unsigned)(size_t)lds3;
    const unsigned rowb = (unsigned)((8 * G + (a >> 2)) * 128 + 8 * (a & 3));
    const int sx = ((a >> 2) & 3) ^ (G & 1);
    unsigned aaddr[4], baddr[4];        // one per 32-byte block (16 features) of a 64-feature panel
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        aaddr[u] = lbase + (2 * wm) * 8192 + rowb + ((u ^ sx) << 5);
        baddr[u] = lbase + 65536 + wn * 8192 + rowb + ((u ^ sx) << 5);
    }
    const bool do_bias = g.dbias != nullptr && n0 == 0;
    float colacc[2] = {0.f, 0.f};
    unsigned long long al[4][2], ah[4][2], bl[2][2][2], bh[2][2][2];   // A half: [ii][ks]; B: [set][jj][ks]; lo / hi = tokens +0..3 / +4..7
#define TN_RD_A(H)                                                                                                        \
    _Pragma("unroll") for (int ii = 0; ii < 4; ++ii) {                                                                    \
        al[ii][0] = lds_tr<(H) * 8192>(aaddr[ii]); ah[ii][0] = lds_tr<(H) * 8192 + 512>(aaddr[ii]);                       \
        al[ii][1] = lds_tr<(H) * 8192 + 4096>(aaddr[ii]); ah[ii][1] = lds_tr<(H) * 8192 + 4608>(aaddr[ii]);               \
    }
#define TN_RD_B(SET, JH, XOR)                                                                                             \
    _Pragma("unroll") for (int jj = 0; jj < 2; ++jj) {                                                                    \
        bl[SET][jj][0] = lds_tr<0>(baddr[2 * (JH) + jj] ^ (XOR)); bh[SET][jj][0] = lds_tr<512>(baddr[2 * (JH) + jj] ^ (XOR)); \
        bl[SET][jj][1] = lds_tr<4096>(baddr[2 * (JH) + jj] ^ (XOR)); bh[SET][jj][1] = lds_tr<4608>(baddr[2 * (JH) + jj] ^ (XOR)); \
    }
    // the compiler does not know the asm read results are in flight: the fragments are tied to the wait that retires them
#define TN_TIE_A() asm volatile("" : "+v"(al[0][0]), "+v"(al[0][1]), "+v"(al[1][0]), "+v"(al[1][1]), "+v"(al[2][0]), "+v"(al[2][1]), "+v"(al[3][0]), "+v"(al[3][1]), \
                                   "+v"(ah[0][0]), "+v"(ah[0][1]), "+v"(ah[1][0]), "+v"(ah[1][1]), "+v"(ah[2][0]), "+v"(ah[2][1]), "+v"(ah[3][0]), "+v"(ah[3][1]) :: "memory");
#define TN_TIE_B(SET) asm volatile("" : "+v"(bl[SET][0][0]), "+v"(bl[SET][0][1]), "+v"(bl[SET][1][0]), "+v"(bl[SET][1][1]), \
                                        "+v"(bh[SET][0][0]), "+v"(bh[SET][0][1]), "+v"(bh[SET][1][0]), "+v"(bh[SET][1][1]) :: "memory");
#define TN_SYNC_IN()                                                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                                    \
    __builtin_amdgcn_s_barrier();                                                                                         \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#define TN_MFMA(IH, JH, SET, BIAS)                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                                    \
    {                                                                                                                     \
        s16x8_t bf_[2][2];                                                                                                \
        _Pragma("unroll") for (int jj = 0; jj < 2; ++jj)                                                                  \
            _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) bf_[jj][ks] = tn_frag(bl[SET][jj][ks], bh[SET][jj][ks], !BF16_B); \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                                  \
            _Pragma("unroll") for (int ii = 0; ii < 4; ++ii) {                                                            \
                const s16x8_t af = tn_frag(al[ii][ks], ah[ii][ks], false);                                                \
                _Pragma("unroll") for (int jj = 0; jj < 2; ++jj)                                                          \
                    acc[4 * (IH) + ii][2 * (JH) + jj] = mfma16t<false>(bf_[jj][ks], af, acc[4 * (IH) + ii][2 * (JH) + jj]); \
                if ((BIAS) && do_bias && ((4 * (IH) + ii) >> 1) == wn) {                                                  \
                    unsigned w4[4];                                                                                       \
                    __builtin_memcpy(w4, &af, 16);                                                                        \
                    _Pragma("unroll") for (int e = 0; e < 4; ++e)                                                         \
                        colacc[ii & 1] += __uint_as_float(w4[e] << 16) + __uint_as_float(w4[e] & 0xffff0000u);            \
                }                                                                                                         \
            }                                                                                                             \
    }                                                                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                                                    \
    __builtin_amdgcn_s_barrier();                                                                                         \
    __builtin_amdgcn_sched_barrier(0);
    // one K tile; X = B register set holding this tile's half 0, Y = the other set
#define TN_TILE(T_, X, Y)                                                                                                 \
    if ((T_) + 1 < nk) { TN_DMA(2, (T_) + 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }                           \
    else { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }                                                             \
    TN_RD_A(0)                                                                                                            \
    TN_SYNC_IN() TN_TIE_A() TN_TIE_B(X)                                                                                   \
    TN_MFMA(0, 0, X, true)                                                                                                \
    if ((T_) + 1 < nk) TN_DMA(3, (T_) + 1)                                                                                \
    TN_RD_B(Y, 1, 0)                                                                                                      \
    TN_SYNC_IN() TN_TIE_B(Y)                                                                                              \
    TN_MFMA(0, 1, Y, false)                                                                                               \
    if ((T_) + 2 < nk) { TN_DMA(0, (T_) + 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }                           \
    else if ((T_) + 1 < nk) { asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }                                          \
    TN_RD_A(1)                                                                                                            \
    TN_SYNC_IN() TN_TIE_A()                                                                                               \
    TN_MFMA(1, 1, Y, true)                                                                                                \
    if ((T_) + 2 < nk) TN_DMA(1, (T_) + 2)                                                                                \
    if ((T_) + 1 < nk) { TN_RD_B(Y, 0, 0x8000) }                                                                          \
    TN_SYNC_IN() TN_TIE_B(Y)                                                                                              \
    TN_MFMA(1, 0, X, false)                                                                                               \
    _Pragma("unroll") for (int u = 0; u < 4; ++u) { aaddr[u] ^= 0x8000; baddr[u] ^= 0x8000; }

    if (nk > 0) {
        TN_DMA(0, 0) TN_DMA(1, 0) TN_DMA(2, 0) TN_DMA(3, 0)
        if (nk > 1) { TN_DMA(0, 1) TN_DMA(1, 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
        else { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    }
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();   // second wave row: half a phase behind
    if (nk > 0) { TN_RD_B(0, 0, 0) }
    int it = 0;
    for (; it + 1 < nk; it += 2) {
        TN_TILE(it, 0, 1)
        TN_TILE(it + 1, 1, 0)
    }
    if (it < nk) { TN_TILE(it, 0, 1) }
    if (wm == 0) __builtin_amdgcn_s_barrier();   // pairs with the last barrier of waves 4-7
#undef TN_DMA
#undef TN_RD_A
#undef TN_RD_B
#undef TN_TIE_A
#undef TN_TIE_B
#undef TN_SYNC_IN
#undef TN_MFMA
#undef TN_TILE
    if (do_bias) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            float c = colacc[e];
            c += __shfl_xor(c, 16, 64);      // the four 8-token groups of the fragment
            c += __shfl_xor(c, 32, 64);
            if (lane < 16 && wm * 128 + (2 * wn + e) * 16 + lane < mw) unsafeAtomicAdd(&g.dbias[m0 + wm * 128 + (2 * wn + e) * 16 + lane], c);
        }
    }
    __builtin_amdgcn_s_barrier();    // both wave rows are past their last operand read: LDS becomes the per-wave staging area
    // Split-K partial sums.  With a workspace: plain coalesced stores of the tile into ws[split] (a reduce pass adds the splits
    // into dW) -- one workgroup per CU cannot hide 64 Ki device-scope atomics behind anything (measured ~150-200 us per GEMM,
    // as long as the whole K loop).  Without: atomics straight into dW.
    unsigned char* wl = lds3 + wave * V3_WLDS;
    const int l15 = lane & 15, lq = lane >> 4;
    const int mb = m0 + wm * 128, nb = n0 + wn * 64;
    if (mb >= g.M || nb >= g.N) return;      // (64-granular validity: each 64 x 64 pass of a wave's sub-tile is entirely in or entirely out)
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        if (mb + pass * 64 >= g.M) break;
        pp_stage32(wl, acc, pass, l15, lq);
        __builtin_amdgcn_wave_barrier();
        if (g.ws != nullptr) {
            float* dstp = g.ws + ((size_t)split * g.M + mb + pass * 64) * g.N + nb + (lane & 15) * 4;
#pragma unroll
            for (int rb2 = 0; rb2 < 16; rb2 += 8) {
                float4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(wl + ((rb2 + u) * 4 + (lane >> 4)) * V3_RS32 + (lane & 15) * 16);
#pragma unroll
                for (int u = 0; u < 8; ++u) v3_st<float4>(dstp + (size_t)((rb2 + u) * 4 + (lane >> 4)) * g.N, v[u]);
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

// ... for small outputs (a few thousand float4s: the [64, k] gradient images of the PMAM CNN reduce up to 256 splits with 16 blocks' worth of
// threads, each walking all splits one dependent load after the other -- 64 us): 32 float4s per block, the splits dealt to 8 thread groups
__global__ __launch_bounds__(256) void tn_reduce_small_kernel(const float* __restrict__ ws, int ks, int M, int N, float* __restrict__ C,
                                                              int ldc) {
    __shared__ float4 red[8][32];
    const size_t total4 = (size_t)M * N / 4;
    const size_t i = (size_t)blockIdx.x * 32 + (threadIdx.x & 31);
    const int grp = threadIdx.x >> 5;
    float4 a = {0.f, 0.f, 0.f, 0.f};
    if (i < total4)
        for (int s2 = grp; s2 < ks; s2 += 8) {
            const float4 b = reinterpret_cast<const float4*>(ws)[(size_t)s2 * total4 + i];
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
    red[grp][threadIdx.x & 31] = a;
    __syncthreads();
    if (grp == 0 && i < total4) {
#pragma unroll
        for (int g2 = 1; g2 < 8; ++g2) { const float4 b = red[g2][threadIdx.x]; a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
        const size_t e = i * 4, m = e / N, n = e - m * N;
        float4* dst = reinterpret_cast<float4*>(C + m * ldc + n);
        float4 c = *dst;
        c.x += a.x; c.y += a.y; c.z += a.z; c.w += a.w;
        *dst = c;
    }
}

// Counter slots of the dynamic tile walk: a ring in device memory (zero at module load, every launch leaves its slot zeroed again), one
// slot per launch so that GEMMs running concurrently on different streams never share counters.  A slot comes round again after
// DYN_SLOTS launches -- far more than a stream queue holds.
#define DYN_SLOTS 2048
#define DYN_SLOT_INTS (16 * 9)
__device__ int g_tile_counters[DYN_SLOTS * DYN_SLOT_INTS];
static int* dyn_counter_slot() {
    static int* base[64] = {};
    static std::atomic<unsigned> next{0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    if (base[dev] == nullptr) {
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_tile_counters)) != hipSuccess) return nullptr;
        base[dev] = (int*)p;
    }
    return base[dev] + (size_t)(next.fetch_add(1) % DYN_SLOTS) * DYN_SLOT_INTS;
}
// Number of CUs the persistent GEMM kernels size their grids for.  Default: all of them.  A data-parallel job whose collectives run
// beside the backward (ddp.py) reserves the communication kernels' CUs by lowering it (sed_gemm_set_cu_budget / SED_GEMM_CUS): a
// persistent workgroup that cannot become resident beside a communication kernel would otherwise start only after another one has exited.
static std::atomic<int> g_cu_budget{0};
static int cu_budget(int ncu_device) {
    int b = g_cu_budget.load();
    if (b <= 0) {
        const char* e = getenv("SED_GEMM_CUS");
        b = e ? atoi(e) : 0;
    }
    if (b <= 0 || b > ncu_device) b = ncu_device;
    return b < 8 ? 8 : b;
}
// Measurement aid (tools/cu_steal.py): `n` single-wave workgroups, one per CU (81 KiB of LDS each: two cannot share a CU), that hold their
// CUs for `usec` microseconds doing nothing -- a stand-in for the RCCL kernels of a data-parallel step, whose workgroups keep a persistent
// GEMM workgroup (the whole register file of a CU) from becoming resident next to them.
__global__ __launch_bounds__(64) void hold_cus_kernel(unsigned long long ticks, int* sink) {
    extern __shared__ unsigned char hold_lds[];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
    if (sink != nullptr && threadIdx.x == 1000) sink[0] = hold_lds[0];
}
extern "C" int sed_debug_hold_cus(int n_cus, int usec, hipStream_t stream) {
    (void)hipGetLastError();
    if (n_cus <= 0 || usec <= 0) return SED_ERR_ARG;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)hold_cus_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 81 * 1024); attr = true; }
    hipLaunchKernelGGL(hold_cus_kernel, dim3(n_cus), dim3(64), 81 * 1024, stream, (unsigned long long)usec * 100ull, (int*)nullptr);
    return sed_check_launch();
}
// Test aid: number of non-zero words in the counter ring of the current device after a device synchronisation (must be 0 whenever no GEMM
// is in flight: every launch re-arms its slot).  Returns the count (>= 0) or a negative error code.
extern "C" int sed_debug_tile_counters_dirty(int reserved) {
    (void)reserved;
    if (hipDeviceSynchronize() != hipSuccess) return SED_ERR_LAUNCH;
    static int host[DYN_SLOTS * DYN_SLOT_INTS];
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(g_tile_counters), sizeof(host)) != hipSuccess) return SED_ERR_LAUNCH;
    int dirty = 0;
    for (int i = 0; i < DYN_SLOTS * DYN_SLOT_INTS; ++i) dirty += host[i] != 0;
    return dirty;
}
extern "C" int sed_gemm_set_cu_budget(int n_cus) {
    g_cu_budget.store(n_cus > 0 ? n_cus : 0);
    return SED_OK;
}

extern "C" int sed_gemm_dw_tn(const void* dY, const void* X, int x_f16, int T, int M, int N, int ldy, int ldx, float* dW,
                              int ldc, float* dbias, float* workspace, int64_t workspace_bytes, hipStream_t stream) {
    (void)hipGetLastError();
    if (T <= 0 || (M % 64) || (N % 64) || (ldy % 8) || (ldx % 8) || (ldc % 4) || M <= 0 || N <= 0) return SED_ERR_ARG;
    TnArgs g;
    g.A = (const bf16_t*)dY; g.B = (const bf16_t*)X; g.C = dW; g.dbias = dbias;
    g.M = M; g.N = N; g.T = T; g.lda = ldy; g.ldb = ldx; g.ldc = ldc; g.b_f16 = x_f16;
    const int tiles = cdiv(M, V3_T) * cdiv(N, V3_T), ktiles = cdiv(T, BK);
    // one workgroup per CU and ONE round: tiles * ks <= 256 (rounding the split count up instead costs a second, nearly empty
    // round -- 36 tiles x 8 splits = 288 workgroups took twice the time of 36 x 7)
    static int ncu_dev_tn = 0;
    if (ncu_dev_tn == 0) {
        int dev = 0, n = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        ncu_dev_tn = n;
    }
    int ks = cu_budget(ncu_dev_tn) / tiles;      // (the CUs a data-parallel job leaves to the GEMMs: see cu_budget)
    if (ks >= 8) ks &= ~7;        // whole splits per XCD (the kernel lays the splits out XCD-contiguously)
    if (ks > ktiles / 16) ks = ktiles / 16;
    if (ks < 1) ks = 1;
    g.ksplit = ks;
    g.ws = (workspace != nullptr && workspace_bytes >= (int64_t)ks * M * N * 4) ? workspace : nullptr;
    static bool attr[2] = {false, false};
    dim3 grid(tiles * ks);
    if (x_f16) {
        if (!attr[1]) { (void)hipFuncSetAttribute((const void*)gemm_tn_dw_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, V3_LDS); attr[1] = true; }
        hipLaunchKernelGGL((gemm_tn_dw_kernel<false>), grid, dim3(512), V3_LDS, stream, g);
    } else {
        if (!attr[0]) { (void)hipFuncSetAttribute((const void*)gemm_tn_dw_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, V3_LDS); attr[0] = true; }
        hipLaunchKernelGGL((gemm_tn_dw_kernel<true>), grid, dim3(512), V3_LDS, stream, g);
    }
    if (g.ws != nullptr) {
        const size_t total4 = (size_t)M * N / 4;
        if (total4 < 32768 && ks >= 16) {
            hipLaunchKernelGGL(tn_reduce_small_kernel, dim3((unsigned)((total4 + 31) / 32)), dim3(256), 0, stream, g.ws, ks, M, N, dW, ldc);
        } else {
            int blocks = (int)((total4 + 255) / 256);
            if (blocks > 2048) blocks = 2048;
            hipLaunchKernelGGL(tn_reduce_kernel, dim3(blocks), dim3(256), 0, stream, g.ws, ks, M, N, dW, ldc);
        }
    }
    return sed_check_launch();
}

template <int EPI>
static int launch_gemm(const GemmArgs& g, int f16, hipStream_t s) {
    if (g.M <= 0 || g.N % TILE != 0 || g.K % BK != 0 || g.ksplit < 1) return SED_ERR_ARG;
    if ((g.lda % 8) || (g.ldb % 8) || (g.ldc % 4)) return SED_ERR_ARG;
    // 256^2 kernel: every forward / dX GEMM of the model (N % 256 == 0, M >= 1024).  The split-K weight-gradient GEMMs of the NT
    // form stay on the 128^2 kernel (two workgroups per CU cover its atomic epilogue).
    if constexpr (EPI != EPI_ATOMIC) if (g.N % V3_T == 0 && g.M >= 1024 && g.ksplit == 1) {
        if ((long long)g.lda * 2 * V3_T >= (1LL << 31) || (long long)g.ldb * 2 * V3_T >= (1LL << 31)) return SED_ERR_ARG;  // 32-bit panel offsets
        // L2 grouping: 4 tile rows x all tile columns per group, groups XCD-contiguous.  Swept 2..32 on the model's shapes
        // (tools/gemm_l2.py): fabric reads stay at 1.4-2.3x (N = 768) / ~5x (N = 3072) of the algorithmic operand bytes for every
        // height -- column tiles of one row panel drift apart by more than the ~2 K-steps a line survives in the 4 MB L2 and re-read it
        // from the Infinity Cache -- while the time is best at 4 (fc2 1186 vs 998 TFLOP/s at 2): the counter does not track the time.
        GemmArgs gg = g;
        gg.group_m = 4;
        const int persist_env = 1;      // (round 6: SED_GEMM_PERSIST=0, one workgroup per tile, lost its A/B in round 3 and is gone)
        // SED_GEMM_RB (read per launch): 7 / 8 force a tile height (A/B and the bit-exactness test), 0 = choose by the round count below.
        // Default 8: alone on the GPU the 224-row form wins 3-4 % on the N = 768 shapes, but inside the train step the teacher's and the
        // weight-gradient streams fill the last round's idle CUs anyway and the shorter tiles cost 0.25 % (104.03 vs 103.77 ms, 3 A/B pairs).
        const char* rb_s = getenv("SED_GEMM_RB");
        const int rb_env = rb_s ? atoi(rb_s) : 8;
        static int ncu_probe = 0;
        if (ncu_probe == 0) {
            int dev = 0, n = 0;
            (void)hipGetDevice(&dev);
            if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
            ncu_probe = n;
        }
        const int ncu_dev = ncu_probe;
        const int ncu = cu_budget(ncu_dev) & ~7;   // whole XCD rounds: blockIdx.x & 7 must stay the XCD of every tile a workgroup walks
        // Tile height: rounds of workgroups x rows per tile is what the launch costs; 224-row tiles win when they fill the last round
        // that 256-row tiles leave mostly empty (M = 38080: 2 x 256 vs 2 x 224 for N = 768, 6 x 256 vs 6 x 224 for N = 2304).
        const int ntn = g.N / V3_T;
        const long long t8 = (long long)cdiv(g.M, 256) * ntn, t7 = (long long)cdiv(g.M, 224) * ntn;
        const long long c8 = ((t8 + ncu - 1) / ncu) * 256, c7 = ((t7 + ncu - 1) / ncu) * 224;
        const bool use7 = g.gbias == nullptr && g.k_wrap == 0 && g.k8 == 0 && (rb_env == 7 || (rb_env == 0 && c7 * 100 < c8 * 97));
        dim3 grid3((unsigned)(use7 ? t7 : t8), 1);
        gg.persist = (persist_env && (int)grid3.x > ncu) ? 1 : 0;
        if (gg.persist) grid3.x = ncu;
        {
            // SED_GEMM_DYN (read per launch; default 1): dynamic tile walk of the persistent form, 0 = the static walk (bit-identical results)
            const char* dy = getenv("SED_GEMM_DYN");
            gg.tile_ctr = (gg.persist && !(dy && atoi(dy) == 0)) ? dyn_counter_slot() : nullptr;
        }
        gg.stagger = 0;      // (the staggered-start experiment of rounds 3 / 4 showed no effect; its environment switches are gone, the field stays reserved)
        const GemmArgs& g = gg;
        if (g.rowpart != nullptr || g.rowstat != nullptr) {
            // folded LayerNorm: producer (residual epilogue) or consumer (head-split / fused-GELU epilogue); f16, no other special mode
            if constexpr (EPI == EPI_F32_RESID || EPI == EPI_GELU || EPI == EPI_QKV) {
                constexpr bool producer = EPI == EPI_F32_RESID;
                if (!f16 || g.gbias != nullptr || g.k_wrap != 0 || (g.N & 63)) return SED_ERR_ARG;
                if (producer ? (g.rowpart == nullptr || g.outH == nullptr || g.rowstat != nullptr)
                             : (g.rowstat == nullptr || g.colS == nullptr || g.rowpart != nullptr)) return SED_ERR_ARG;
                static bool attrl[8] = {false, false, false, false, false, false, false, false};
#define PP_LN_LAUNCH(GBV, RBV, SLOT)                                                                                     \
                {                                                                                                         \
                    if (!attrl[SLOT]) { (void)hipFuncSetAttribute((const void*)gemm_nt_pp_kernel<EPI, true, GBV, RBV>, hipFuncAttributeMaxDynamicSharedMemorySize, V3_LDS); attrl[SLOT] = true; } \
                    hipLaunchKernelGGL((gemm_nt_pp_kernel<EPI, true, GBV, RBV>), grid3, dim3(512), V3_LDS, s, g);          \
                }
                if constexpr (producer) {
                    if (g.res_lo != nullptr && g.lo8 && g.out_lo != nullptr) { if (use7) PP_LN_LAUNCH(6, 7, 4) else PP_LN_LAUNCH(6, 8, 5) }
                    else if (g.res_lo != nullptr && g.lo8) { if (use7) PP_LN_LAUNCH(7, 7, 6) else PP_LN_LAUNCH(7, 8, 7) }
                    else if (g.res_lo != nullptr) { if (use7) PP_LN_LAUNCH(5, 7, 2) else PP_LN_LAUNCH(5, 8, 3) }
                    else { if (use7) PP_LN_LAUNCH(3, 7, 0) else PP_LN_LAUNCH(3, 8, 1) }
                } else {
                    if (use7) PP_LN_LAUNCH(3, 7, 0) else PP_LN_LAUNCH(3, 8, 1)
                }
#undef PP_LN_LAUNCH
                return sed_check_launch();
            } else {
                return SED_ERR_ARG;
            }
        }
        if (g.k8 != 0) {
            // two-term weights, lo product on the fp8 matrix path (GemmArgs.k8): the two-term variants' epilogues behind a K walk of f16
            // tiles followed by e4m3 tiles
            if constexpr (EPI == EPI_F32_RESID || EPI == EPI_GELU || EPI == EPI_QKV) {
                if (!f16 || g.gbias != nullptr || g.k_wrap != 0 || g.a_slab || ((g.K / BK - g.k8) & 1) || g.k8 >= g.K / BK) return SED_ERR_ARG;
                static bool attr9 = false;
                if constexpr (EPI == EPI_QKV) {
                    if (g.qt != nullptr || g.kt != nullptr || g.vt != nullptr || g.q2 != nullptr || g.q2t != nullptr || g.pu != nullptr) return SED_ERR_ARG;
                    if (!attr9) { (void)hipFuncSetAttribute((const void*)gemm_nt_pp_kernel<EPI, true, 8, 8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, V3_LDS); attr9 = true; }
                    hipLaunchKernelGGL((gemm_nt_pp_kernel<EPI, true, 8, 8, true>), grid3, dim3(512), V3_LDS, s, g);
                } else {
                    if (!attr9) { (void)hipFuncSetAttribute((const void*)gemm_nt_pp_kernel<EPI, true, 2, 8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, V3_LDS); attr9 = true; }
                    hipLaunchKernelGGL((gemm_nt_pp_kernel<EPI, true, 2, 8, true>), grid3, dim3(512), V3_LDS, s, g);
                }
                return sed_check_launch();
            } else {
                return SED_ERR_ARG;
            }
        }
        if (g.gbias != nullptr || g.k_wrap != 0) {
            // row-group bias / two-term weights: evaluation-mode encoder GEMMs only (f16 operands; residual, fused-GELU and head-split epilogues)
            if constexpr (EPI == EPI_F32_RESID || EPI == EPI_GELU || EPI == EPI_QKV) {
                if (!f16 || (g.gbias != nullptr && g.k_wrap != 0)) return SED_ERR_ARG;
                static bool attrg[2] = {false, false};
                if constexpr (EPI == EPI_QKV) {
                    // two-term weights, row-major q / k / v only (the evaluation passes of the encoder): mode 8 = the two-term K walk with the
                    // LDS-free head-split epilogue of modes 3 / 4
                    if (g.k_wrap != 0 && g.qt == nullptr && g.kt == nullptr && g.vt == nullptr && g.q2 == nullptr && g.q2t == nullptr && g.pu == nullptr) {
                        static bool attr8 = false;
                        if (!attr8) { (void)hipFuncSetAttribute((const void*)gemm_nt_pp_kernel<EPI, true, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, V3_LDS); attr8 = true; }
                        hipLaunchKernelGGL((gemm_nt_pp_kernel<EPI, true, 8>), grid3, dim3(512), V3_LDS, s, g);
                        return sed_check_launch();
                    }
                }
                if constexpr (EPI == EPI_GELU) if (g.out_e4m3) {
                    // row-group bias + the e4m3 image of the output (fc1 of the evaluation-mode encoder in its fp8 form: f16 weights + mean
                    // correction, the activation leaves as fc2's two-image A operand): the F8 kernel with no fp8 K tiles
                    if (g.k_wrap != 0 || ((g.K / BK) & 1)) return SED_ERR_ARG;
                    static bool attrge = false;
                    if (!attrge) { (void)hipFuncSetAttribute((const void*)gemm_nt_pp_kernel<EPI, true, 1, 8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, V3_LDS); attrge = true; }
                    hipLaunchKernelGGL((gemm_nt_pp_kernel<EPI, true, 1, 8, true>), grid3, dim3(512), V3_LDS, s, g);
                    return sed_check_launch();
                }
                if (g.k_wrap != 0) {
                    if (!attrg[1]) { (void)hipFuncSetAttribute((const void*)gemm_nt_pp_kernel<EPI, true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, V3_LDS); attrg[1] = true; }
                    hipLaunchKernelGGL((gemm_nt_pp_kernel<EPI, true, 2>), grid3, dim3(512), V3_LDS, s, g);
                } else {
                    if (!attrg[0]) { (void)hipFuncSetAttribute((const void*)gemm_nt_pp_kernel<EPI, true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, V3_LDS); attrg[0] = true; }
                    hipLaunchKernelGGL((gemm_nt_pp_kernel<EPI, true, 1>), grid3, dim3(512), V3_LDS, s, g);
                }
                return sed_check_launch();
            } else {
                return SED_ERR_ARG;
            }
        }
        if constexpr (EPI == EPI_QKV) {
            // the encoder's form of the head split -- row-major q / k / v, no biased second q, no transposed copies (its attention kernels
            // transpose in LDS): a kernel without those paths (mode 4), 3 % faster than carrying them as run-time branches
            if (g.qt == nullptr && g.kt == nullptr && g.vt == nullptr && g.q2 == nullptr && g.q2t == nullptr && g.pu == nullptr) {
                static bool attrq[2][2] = {{false, false}, {false, false}};
#define SED_PP_LAUNCH_Q(F, RBV)                                                                                            \
                {                                                                                                          \
                    if (!attrq[F][RBV - 7]) {                                                                              \
                        (void)hipFuncSetAttribute((const void*)gemm_nt_pp_kernel<EPI, F, 4, RBV>, hipFuncAttributeMaxDynamicSharedMemorySize, V3_LDS); \
                        attrq[F][RBV - 7] = true;                                                                          \
                    }                                                                                                      \
                    hipLaunchKernelGGL((gemm_nt_pp_kernel<EPI, F, 4, RBV>), grid3, dim3(512), V3_LDS, s, g);               \
                }
                if (f16) { if (use7) SED_PP_LAUNCH_Q(true, 7) else SED_PP_LAUNCH_Q(true, 8) }
                else { if (use7) SED_PP_LAUNCH_Q(false, 7) else SED_PP_LAUNCH_Q(false, 8) }
#undef SED_PP_LAUNCH_Q
                return sed_check_launch();
            }
        }
        static bool attrp[2][2] = {{false, false}, {false, false}};
#define SED_PP_LAUNCH(F, RBV)                                                                                              \
        {                                                                                                                  \
            if (!attrp[F][RBV - 7]) {                                                                                      \
                (void)hipFuncSetAttribute((const void*)gemm_nt_pp_kernel<EPI, F, 0, RBV>, hipFuncAttributeMaxDynamicSharedMemorySize, V3_LDS); \
                attrp[F][RBV - 7] = true;                                                                                  \
            }                                                                                                              \
            hipLaunchKernelGGL((gemm_nt_pp_kernel<EPI, F, 0, RBV>), grid3, dim3(512), V3_LDS, s, g);                   \
        }
        if (f16) { if (use7) SED_PP_LAUNCH(true, 7) else SED_PP_LAUNCH(true, 8) }
        else { if (use7) SED_PP_LAUNCH(false, 7) else SED_PP_LAUNCH(false, 8) }
#undef SED_PP_LAUNCH
        return sed_check_launch();
    }
    if (g.k_wrap != 0 || g.k8 != 0 || g.rowpart != nullptr || g.rowstat != nullptr) return SED_ERR_ARG;   // 256^2-kernel-only modes (N % 256 == 0, M >= 1024)
    dim3 grid(cdiv(g.M, TILE) * (g.N / TILE), g.ksplit);
    if (f16) hipLaunchKernelGGL((gemm_nt_kernel<EPI, true>), grid, dim3(256), 0, s, g);
    else hipLaunchKernelGGL((gemm_nt_kernel<EPI, false>), grid, dim3(256), 0, s, g);
    return sed_check_launch();
}

static int gemm_nt_impl(const void* A, const void* B, int M, int N, int K, int lda, int ldb, int epi, const float* bias,
                        const float* resF, float* outF, void* outH, void* outH2, const void* auxH, int ldc, float alpha,
                        int ksplit, int f16, int ncols, hipStream_t stream, const float* gbias = nullptr, int gb_rows = 0, int two_term = 0,
                        int f8_scale = 0, int out_e4m3 = 0) {
    (void)hipGetLastError();
    GemmArgs g = {};
    if (out_e4m3 && ((two_term != 3 && gbias == nullptr) || epi != EPI_GELU || outH != nullptr || ldc < N + N / 2)) return SED_ERR_ARG;
    g.out_e4m3 = out_e4m3;
    if (two_term == 3) {      // A [M, K f16 | K e4m3] against B [N, K f16 | K e4m3]: K / 64 f16 tiles + K / 128 fp8 tiles
        if (K % 128 || epi == EPI_ATOMIC || epi == EPI_DGELU || gbias != nullptr || ksplit > 1) return SED_ERR_ARG;
        g.k8 = K / 128; g.f8_scale = f8_scale;
        K += K / 2;
    } else if (two_term) {      // A [M, K] against B [N, 2K]
        if (K % BK || epi == EPI_ATOMIC || epi == EPI_DGELU || gbias != nullptr || ksplit > 1) return SED_ERR_ARG;
        g.k_wrap = K / BK;
        K *= 2;
    }
    if (gbias != nullptr && (gb_rows < 128 || M % gb_rows || epi == EPI_ATOMIC || epi == EPI_DGELU)) return SED_ERR_ARG;
    g.gbias = gbias; g.gb_rows = gb_rows;
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
extern "C" int sed_gemm_nt_gb(const void* A, const void* B, int M, int N, int K, int lda, int ldb, int epi,
                              const float* bias, const float* resF, float* outF, void* outH, void* outH2,
                              const void* auxH, int ldc, float alpha, int f16, const float* gbias, int gb_rows, hipStream_t stream) {
    return gemm_nt_impl(A, B, M, N, K, lda, ldb, epi, bias, resF, outF, outH, outH2, auxH, ldc, alpha, 1, f16, N, stream, gbias, gb_rows);
}
// ... fused GELU with the e4m3 image of the result in the same rows (outH2 [M][N f16 | N e4m3], ldc >= 3N / 2): the A operand of a following
// sed_gemm_nt_w2f8
extern "C" int sed_gemm_nt_gb_e4m3(const void* A, const void* B, int M, int N, int K, int lda, int ldb, const float* bias, void* outH2,
                                   int ldc, const float* gbias, int gb_rows, hipStream_t stream) {
    if (gbias == nullptr || N % 256 || M < 1024) return SED_ERR_ARG;
    return gemm_nt_impl(A, B, M, N, K, lda, ldb, EPI_GELU, bias, nullptr, nullptr, nullptr, outH2, nullptr, ldc, 1.f, 1, 1, N, stream, gbias, gb_rows,
                        0, 0, 1);
}
// Two-term weights: A [M, K] (f16) against B [N, 2K] = [f16(W) | f16(W - f16(W))] (sed_weight_two_term_f16); A is read twice, the
// result is A . W^T with W good to ~2^-19 relative.  256^2 kernel only (N % 256 == 0, M >= 1024), epilogues EPI_F32_RESID / EPI_GELU.
extern "C" int sed_gemm_nt_w2(const void* A, const void* B, int M, int N, int K, int lda, int ldb, int epi,
                              const float* bias, const float* resF, float* outF, void* outH, void* outH2,
                              int ldc, int f16, hipStream_t stream) {
    if (!(f16 & 1) || N % 256 || M < 1024) return SED_ERR_ARG;
    return gemm_nt_impl(A, B, M, N, K, lda, ldb, epi, bias, resF, outF, outH, outH2, nullptr, ldc, 1.f, 1, f16, N, stream, nullptr, 0, 1);
}
// ... with the lo product on the fp8 matrix path: A [M][K f16 | K e4m3] (row pitch lda halfs >= 3K / 2; the e4m3 half is the activation
// times 2^-2: sed_fp8_tail), B [N][K f16 | K e4m3] = sed_weight_two_term_f8's image with its scale exponent s; f8_exp = s.  K % 128 == 0.
static int f8_scale_word(int s) {       // both operands' scale registers are the same one: e8m0 byte of 2^-(s - 2) / 2, replicated; s even
    const int e = 128 - s / 2;
    return (s & 1) || e < 1 || e > 254 ? -1 : e * 0x01010101;
}
// out_e4m3 != 0 (epi 3, outH NULL): outH2 rows are [N f16 | N e4m3] (ldc >= 3N / 2) -- the activation and its e4m3 image, the A operand
// of the next sed_gemm_nt_w2f8
extern "C" int sed_gemm_nt_w2f8(const void* A, const void* B, int M, int N, int K, int lda, int ldb, int epi,
                                const float* bias, const float* resF, float* outF, void* outH, void* outH2,
                                int ldc, int out_e4m3, int f8_exp, hipStream_t stream) {
    if (N % 256 || M < 1024 || f8_scale_word(f8_exp) < 0) return SED_ERR_ARG;
    return gemm_nt_impl(A, B, M, N, K, lda, ldb, epi, bias, resF, outF, outH, outH2, nullptr, ldc, 1.f, 1, 1, N, stream, nullptr, 0, 3,
                        f8_scale_word(f8_exp), out_e4m3);
}
// LayerNorm folded into the two GEMMs around it (no-grad f16 passes; 256^2 kernel only: N % 256 == 0, M >= 1024).
// producer = sed_gemm_nt with EPI_F32_RESID that ALSO writes x16 [M, N] (f16 image of the new residual stream) and rowpart [M][N / 64][2]
// (per-row partial sum / sum of squares of each 64-column slice); sed_ln_fold_stats turns rowpart into rowstat [M][2] = (mean, rstd);
// consumer = EPI_GELU GEMM whose A operand is x16 (the RAW stream), B the f16 image of gamma (.) W (sed_ln_fold_weight), with
// out[m, n] = act(rstd[m] * (acc - mean[m] * colS[n]) + colC[n]).
static int gemm_nt_lnp_impl(const void* A, const void* B, int M, int N, int K, int lda, int ldb, const float* bias, const float* resF,
                            const void* res_hi, const void* res_lo, float* outF, void* x16, void* out_lo, float* rowpart, int ldc, int lo8,
                            hipStream_t stream) {
    (void)hipGetLastError();
    if (N % 256 || M < 1024 || K % BK || x16 == nullptr || rowpart == nullptr || ldc != N) return SED_ERR_ARG;
    // residual: fp32 resF, or the planes res_hi + res_lo; result: fp32 outF + f16 image x16, or the planes x16 (hi) + out_lo
    if ((res_lo != nullptr) != (res_hi != nullptr) || (res_lo == nullptr && resF == nullptr) || (out_lo == nullptr && outF == nullptr)) return SED_ERR_ARG;
    GemmArgs g = {};
    g.A = (const bf16_t*)A; g.B = (const bf16_t*)B; g.ncols = N;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ksplit = 1; g.alpha = 1.f;
    g.bias = bias; g.resF = resF; g.outF = outF; g.outH = (bf16_t*)x16; g.rowpart = rowpart;
    g.auxH = (const bf16_t*)res_hi; g.res_lo = (const bf16_t*)res_lo; g.out_lo = (bf16_t*)out_lo; g.lo8 = lo8;
    if (lo8 && lda == 64 && K > 64) { g.a_slab = 1; g.lda = K; }      // A as [K / 64][M][64] (what sed_gemm_nt_lnc8 writes with ldc = 64)
    // (slab-major operands are addressed through one 32-bit buffer range / K-tile offset: the whole plane has to stay below 2 GiB)
    if (g.a_slab && (size_t)M * K * 2 >= (1ull << 31)) return SED_ERR_ARG;
    return launch_gemm<EPI_F32_RESID>(g, 1, stream);
}
extern "C" int sed_gemm_nt_lnp(const void* A, const void* B, int M, int N, int K, int lda, int ldb, const float* bias, const float* resF,
                               const void* res_hi, const void* res_lo, float* outF, void* x16, void* out_lo, float* rowpart, int ldc,
                               hipStream_t stream) {
    return gemm_nt_lnp_impl(A, B, M, N, K, lda, ldb, bias, resF, res_hi, res_lo, outF, x16, out_lo, rowpart, ldc, 0, stream);
}
// ... with the lo plane as bytes (GemmArgs.lo8): res_lo / out_lo are [M][ldc] uint8
extern "C" int sed_gemm_nt_lnp8(const void* A, const void* B, int M, int N, int K, int lda, int ldb, const float* bias, const float* resF,
                                const void* res_hi, const void* res_lo8, float* outF, void* x16, void* out_lo8, float* rowpart, int ldc,
                                hipStream_t stream) {
    return gemm_nt_lnp_impl(A, B, M, N, K, lda, ldb, bias, resF, res_hi, res_lo8, outF, x16, out_lo8, rowpart, ldc, 1, stream);
}
static int gemm_nt_lnc_impl(const void* A, const void* B, int M, int N, int K, int lda, int ldb, const float* colC, const float* colS,
                            const float* rowstat, void* outH2, int ldc, int a_slab, hipStream_t stream) {
    (void)hipGetLastError();
    if (N % 256 || M < 1024 || K % BK || colS == nullptr || rowstat == nullptr || outH2 == nullptr) return SED_ERR_ARG;
    GemmArgs g = {};
    g.A = (const bf16_t*)A; g.B = (const bf16_t*)B; g.ncols = N;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ksplit = 1; g.alpha = 1.f;
    g.bias = colC; g.outH2 = (bf16_t*)outH2; g.colS = colS; g.rowstat = rowstat; g.a_slab = a_slab;
    if (a_slab && (lda != K || (size_t)M * K * 2 >= (1ull << 31))) return SED_ERR_ARG;
    if (a_slab && ldc == 64 && N > 64) { g.c_slab = 1; g.ldc = N; }      // output as [N / 64][M][64]
    return launch_gemm<EPI_GELU>(g, 1, stream);
}
extern "C" int sed_gemm_nt_lnc(const void* A, const void* B, int M, int N, int K, int lda, int ldb, const float* colC, const float* colS,
                               const float* rowstat, void* outH2, int ldc, hipStream_t stream) {
    return gemm_nt_lnc_impl(A, B, M, N, K, lda, ldb, colC, colS, rowstat, outH2, ldc, 0, stream);
}
// ... fed with the slab-major hi plane sed_gemm_nt_lnp8 writes ([K / 64][M][64] f16)
extern "C" int sed_gemm_nt_lnc8(const void* A, const void* B, int M, int N, int K, int lda, int ldb, const float* colC, const float* colS,
                                const float* rowstat, void* outH2, int ldc, hipStream_t stream) {
    return gemm_nt_lnc_impl(A, B, M, N, K, lda, ldb, colC, colS, rowstat, outH2, ldc, 1, stream);
}
// same GEMM with a narrow result: the operands are padded to N (multiple of 128) but only the first ncols (multiple of 4) output
// columns exist in memory (row stride ldc >= ncols); bias / residual / outputs are indexed like the narrow matrix.  128^2 kernel only.
extern "C" int sed_gemm_nt_cols(const void* A, const void* B, int M, int N, int K, int lda, int ldb, int epi,
                                const float* bias, const float* resF, float* outF, void* outH, void* outH2,
                                const void* auxH, int ldc, float alpha, int f16, int ncols, hipStream_t stream) {
    if (ncols <= 0 || ncols > N || (ncols % 4) || N % 256 == 0 || epi == EPI_ATOMIC) return SED_ERR_ARG;
    return gemm_nt_impl(A, B, M, N, K, lda, ldb, epi, bias, resF, outF, outH, outH2, auxH, ldc, alpha, 1, f16, ncols, stream);
}

static int gemm_qkv_impl(const void* A, const void* W, const float* bias, int M, int K, int heads, int seq,
                         int seq_pad, void* q, void* k, void* v, void* qt, void* kt, void* vt, void* q2,
                         void* q2t, const float* pos_u, const float* pos_v, int f16, const float* gbias, int gb_rows, hipStream_t stream, int two_term = 0,
                         int f8_scale = 0);
// the context network's in_proj on TWO of the three split-precision terms: A = plain f16 activations [M, K], W = the split weight image
// [N, 3K] = [hi | hi | lo] (sed_weight_images); computes A . (hi + lo)^T -- the weight to ~2^-22, the activation rounded once.  Which terms
// the posteriors need per GEMM: tools/err_sim.py (SIM_DEC_TERMS=1): in_proj is insensitive to the activation's lo part (logit error
// 3.96e-4 -> 5.4e-4) and sensitive to the weight's (1.9e-3).  All outputs of sed_gemm_qkv.
extern "C" int sed_gemm_qkv_w2s(const void* A, const void* Wsplit, const float* bias, int M, int K, int heads, int seq, int seq_pad, void* q,
                                void* k, void* v, void* qt, void* kt, void* vt, void* q2, void* q2t, const float* pos_u, const float* pos_v,
                                int f16, hipStream_t stream) {
    if (!(f16 & 1) || M < 1024) return SED_ERR_ARG;
    return gemm_qkv_impl(A, Wsplit, bias, M, K, heads, seq, seq_pad, q, k, v, qt, kt, vt, q2, q2t, pos_u, pos_v, f16, nullptr, 0, stream, 2);
}
extern "C" int sed_gemm_qkv(const void* A, const void* W, const float* bias, int M, int K, int heads, int seq,
                            int seq_pad, void* q, void* k, void* v, void* qt, void* kt, void* vt, void* q2,
                            void* q2t, const float* pos_u, const float* pos_v, int f16, hipStream_t stream) {
    return gemm_qkv_impl(A, W, bias, M, K, heads, seq, seq_pad, q, k, v, qt, kt, vt, q2, q2t, pos_u, pos_v, f16, nullptr, 0, stream);
}
extern "C" int sed_gemm_qkv_gb(const void* A, const void* W, const float* bias, int M, int K, int heads, int seq,
                               int seq_pad, void* q, void* k, void* v, int f16, const float* gbias, int gb_rows, hipStream_t stream) {
    return gemm_qkv_impl(A, W, bias, M, K, heads, seq, seq_pad, q, k, v, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, f16,
                         gbias, gb_rows, stream);
}
// folded-LayerNorm consumer (see sed_gemm_nt_lnc): A = f16 image of the raw stream, W = f16 image of gamma (.) W_qkv; inference outputs only
static int gemm_qkv_lnc_impl(const void* A, const void* W, const float* colC, const float* colS, const float* rowstat, int M, int K,
                             int heads, int seq, int seq_pad, void* q, void* k, void* v, int a_slab, hipStream_t stream) {
    (void)hipGetLastError();
    if (M < 1024 || K % BK || colS == nullptr || rowstat == nullptr || seq <= 0 || (seq_pad % 64) || M % seq) return SED_ERR_ARG;
    GemmArgs g = {};
    g.A = (const bf16_t*)A; g.B = (const bf16_t*)W;
    g.M = M; g.N = 3 * heads * 64; g.K = K; g.lda = K; g.ldb = K; g.ldc = g.N; g.ksplit = 1; g.alpha = 1.f; g.ncols = g.N;
    g.bias = colC; g.colS = colS; g.rowstat = rowstat;
    g.q = (bf16_t*)q; g.k = (bf16_t*)k; g.v = (bf16_t*)v;
    g.seq = seq; g.seq_pad = seq_pad; g.heads = heads; g.a_slab = a_slab;
    if (a_slab && (size_t)M * K * 2 >= (1ull << 31)) return SED_ERR_ARG;
    return launch_gemm<EPI_QKV>(g, 1, stream);
}
extern "C" int sed_gemm_qkv_lnc(const void* A, const void* W, const float* colC, const float* colS, const float* rowstat, int M, int K,
                                int heads, int seq, int seq_pad, void* q, void* k, void* v, hipStream_t stream) {
    return gemm_qkv_lnc_impl(A, W, colC, colS, rowstat, M, K, heads, seq, seq_pad, q, k, v, 0, stream);
}
extern "C" int sed_gemm_qkv_lnc8(const void* A, const void* W, const float* colC, const float* colS, const float* rowstat, int M, int K,
                                 int heads, int seq, int seq_pad, void* q, void* k, void* v, hipStream_t stream) {
    return gemm_qkv_lnc_impl(A, W, colC, colS, rowstat, M, K, heads, seq, seq_pad, q, k, v, 1, stream);
}
// two-term weights (see sed_gemm_nt_w2): W is [3 * heads * 64, 2K]; inference outputs only
extern "C" int sed_gemm_qkv_w2(const void* A, const void* W, const float* bias, int M, int K, int heads, int seq,
                               int seq_pad, void* q, void* k, void* v, int f16, hipStream_t stream) {
    if (!(f16 & 1) || M < 1024) return SED_ERR_ARG;
    return gemm_qkv_impl(A, W, bias, M, K, heads, seq, seq_pad, q, k, v, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, f16,
                         nullptr, 0, stream, 1);
}
// ... lo product on the fp8 matrix path (see sed_gemm_nt_w2f8): A [M][K f16 | K e4m3], W [3 * heads * 64][K f16 | K e4m3], both with
// a row pitch of 3K / 2 halfs
extern "C" int sed_gemm_qkv_w2f8(const void* A, const void* W, const float* bias, int M, int K, int heads, int seq,
                                 int seq_pad, void* q, void* k, void* v, int f8_exp, hipStream_t stream) {
    if (M < 1024 || K % 128 || f8_scale_word(f8_exp) < 0) return SED_ERR_ARG;
    return gemm_qkv_impl(A, W, bias, M, K, heads, seq, seq_pad, q, k, v, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 1,
                         nullptr, 0, stream, 3, f8_scale_word(f8_exp));
}
static int gemm_qkv_impl(const void* A, const void* W, const float* bias, int M, int K, int heads, int seq,
                         int seq_pad, void* q, void* k, void* v, void* qt, void* kt, void* vt, void* q2,
                         void* q2t, const float* pos_u, const float* pos_v, int f16, const float* gbias, int gb_rows, hipStream_t stream, int two_term,
                         int f8_scale) {
    (void)hipGetLastError();
    GemmArgs g = {};
    if (gbias != nullptr && (gb_rows < 128 || M % gb_rows)) return SED_ERR_ARG;
    g.gbias = gbias; g.gb_rows = gb_rows;
    g.A = (const bf16_t*)A; g.B = (const bf16_t*)W;
    g.M = M; g.N = 3 * heads * 64; g.K = K; g.lda = K; g.ldb = K; g.ldc = g.N; g.ksplit = 1; g.alpha = 1.f;
    if (two_term == 3) {
        if (K % 128 || gbias != nullptr) return SED_ERR_ARG;
        g.k8 = K / 128; g.f8_scale = f8_scale; g.K = K + K / 2; g.lda = g.K; g.ldb = g.K;
    } else if (two_term) {
        if (K % BK || gbias != nullptr) return SED_ERR_ARG;
        g.k_wrap = K / BK; g.K = 2 * K; g.ldb = 2 * K;
        if (two_term == 2) { g.ldb = 3 * K; g.b_skip = K / BK; }      // W = [hi | hi | lo]: walk hi, then lo
    }
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
// image: ~130 launches of 5-20 us for student + teacher.  desc: n_desc x 16 int64 {in fp32, outT bf16 [C, R] or 0,
// outS [R, C] or 0, split f16 [R, 3C] or 0, R, C, outS kind (0 bf16 / 2 f16), first tile index, gather plan int32 [R, C] or 0,
// scale plan fp32 [R, C] or 0, LoRA A fp32 [r, C] or 0, LoRA B fp32 [R, r], r, LoRA scaling (float bits), 0, 0}; one workgroup per
// 64 x 64 tile.  Image element (i, j) = in[plan ? plan[i C + j] : i C + j] * (scale ? scale[i C + j] : 1) + s sum_k B[i, k] A[k, j]:
// the PMAM model's padded context-network / CNN images (a gather of the master with zeros on the padding, round 4) and its merged
// LoRA weights W + s B A (src/models/lora/layers.py:148-151) without an fp32 copy of either.
#define WIMG_DESC 16
__global__ __launch_bounds__(256) void weight_images_kernel(const long long* __restrict__ desc, int n_desc) {
    __shared__ float tile[64][65];
    int lo = 0, hi = n_desc - 1;
    while (lo < hi) {   // last descriptor whose first tile <= blockIdx.x
        const int mid = (lo + hi + 1) >> 1;
        if (desc[(size_t)mid * WIMG_DESC + 7] <= (long long)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const long long* d = desc + (size_t)lo * WIMG_DESC;
    const float* in = reinterpret_cast<const float*>(d[0]);
    bf16_t* outT = reinterpret_cast<bf16_t*>(d[1]);
    bf16_t* outS = reinterpret_cast<bf16_t*>(d[2]);
    bf16_t* outP = reinterpret_cast<bf16_t*>(d[3]);
    const int R = (int)d[4], C = (int)d[5], skind = (int)d[6];
    const int tl = blockIdx.x - (int)d[7], tx = C / 64;
    const int c0 = (tl % tx) * 64, r0 = (tl / tx) * 64;
    const int* gplan = reinterpret_cast<const int*>(d[8]);
    const float* gscale = reinterpret_cast<const float*>(d[9]);
    const float* la = reinterpret_cast<const float*>(d[10]);
    const float* lb = reinterpret_cast<const float*>(d[11]);
    const int lr = (int)d[12];
    const float ls = __int_as_float((int)d[13]);
    const int t = threadIdx.x;
    {
        const int row = t >> 2, cg = (t & 3) * 16, r = r0 + row;
        float v[16];
        if (r < R) {
            const size_t e0 = (size_t)r * C + c0 + cg;
            if (gplan != nullptr) {      // (uniform per workgroup)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int4 ix = reinterpret_cast<const int4*>(gplan + e0)[k];
                    v[4 * k] = in[ix.x]; v[4 * k + 1] = in[ix.y]; v[4 * k + 2] = in[ix.z]; v[4 * k + 3] = in[ix.w];
                }
            } else {
                const float4* src = reinterpret_cast<const float4*>(in + e0);
#pragma unroll
                for (int k = 0; k < 4; ++k) { const float4 f = src[k]; v[4 * k] = f.x; v[4 * k + 1] = f.y; v[4 * k + 2] = f.z; v[4 * k + 3] = f.w; }
            }
            if (gscale != nullptr) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float4 f = reinterpret_cast<const float4*>(gscale + e0)[k];
                    v[4 * k] *= f.x; v[4 * k + 1] *= f.y; v[4 * k + 2] *= f.z; v[4 * k + 3] *= f.w;
                }
            }
            if (la != nullptr) {         // + s B A, accumulated as sed_lora_merge does (k ascending, one fma with s at the end)
                float acc[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[e] = 0.f;
                for (int k = 0; k < lr; ++k) {
                    const float bv = lb[(size_t)r * lr + k];
                    const float4* arow = reinterpret_cast<const float4*>(la + (size_t)k * C + c0 + cg);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 a = arow[q];
                        acc[4 * q] = fmaf(bv, a.x, acc[4 * q]); acc[4 * q + 1] = fmaf(bv, a.y, acc[4 * q + 1]);
                        acc[4 * q + 2] = fmaf(bv, a.z, acc[4 * q + 2]); acc[4 * q + 3] = fmaf(bv, a.w, acc[4 * q + 3]);
                    }
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) v[e] = fmaf(ls, acc[e], v[e]);
            }
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

// ---------------------------------------------------------------------------------------------------------------------
// Weight-rounding correction of the evaluation-mode encoder (engine.py `_wcorr_bias`).  The f16 image of an fp32 weight drops
// W_lo = W - f16(W); the dropped product x . W_lo^T has a part that is the SAME for every token of a clip -- mean_t(x) . W_lo^T -- which
// no later averaging (frequency pooling, attention) reduces, and which is most of the posterior error the image costs (measured
// with tools/err_sim.py: logit error of the f16 weight images 1.6e-3 -> 0.7e-3).  It is a [clips, K] x [K, N] product: the two kernels
// below make its operands (per-clip token means of the 16-bit activation; f16 image of 2^11 W_lo), the ordinary GEMM forms it and
// the main GEMM adds it as a row-group bias.
// x [groups * rows, K] 16-bit -> mean [groups, K] 16-bit (same kind); one workgroup per (group, 256-column chunk), fp32 sums
// `step` > 1: every step-th token only -- an estimate of the mean (the correction needs it to a few per cent; the part of the dropped
// product that varies from token to token is left uncorrected anyway), for 1 / step of the extra pass over the activation
template <bool F16>
__global__ __launch_bounds__(256) void group_colmean_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ out, int rows, int K, int ld, int step) {
    __shared__ float part[8][256];
    const int grp = blockIdx.x, c0 = blockIdx.y * 256;
    // thread = (row slice of 8, 32 column groups of 8 columns = 16 bytes)
    const int cg = threadIdx.x & 31, sl = threadIdx.x >> 5;
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const bf16_t* base = x + (size_t)grp * rows * ld + c0 + cg * 8;
    const int nvis = (rows + step - 1) / step;
    for (int r = sl * step; r < rows; r += 8 * step) {
        const uint4 u = *reinterpret_cast<const uint4*>(base + (size_t)r * ld);
        const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            a[2 * e] += to_f32<F16>((bf16_t)(w[e] & 0xFFFF));
            a[2 * e + 1] += to_f32<F16>((bf16_t)(w[e] >> 16));
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) part[sl][cg * 8 + e] = a[e];
    __syncthreads();
    const int c = threadIdx.x;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += part[i][c];
    out[(size_t)grp * K + c0 + c] = to_16<F16>(s / (float)nvis);
}
extern "C" int sed_group_colmean_ld(const void* x, void* out, int groups, int rows, int K, int ld, int step, int f16, hipStream_t stream) {
    (void)hipGetLastError();
    if (groups <= 0 || rows <= 0 || K % 256 || step < 1 || ld < K || ld % 8) return SED_ERR_ARG;
    if (f16) hipLaunchKernelGGL(group_colmean_kernel<true>, dim3(groups, K / 256), dim3(256), 0, stream, (const bf16_t*)x, (bf16_t*)out, rows, K, ld, step);
    else hipLaunchKernelGGL(group_colmean_kernel<false>, dim3(groups, K / 256), dim3(256), 0, stream, (const bf16_t*)x, (bf16_t*)out, rows, K, ld, step);
    return sed_check_launch();
}
extern "C" int sed_group_colmean(const void* x, void* out, int groups, int rows, int K, int step, int f16, hipStream_t stream) {
    return sed_group_colmean_ld(x, out, groups, rows, K, K, step, f16, stream);
}
// out = f16(scale * (w - f16(w)))  -- what the straight f16 image of an fp32 weight dropped (scale 2^11 keeps it a normal number)
__global__ void weight_residual_kernel(const float* __restrict__ w, bf16_t* __restrict__ out, size_t n4, float scale) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = reinterpret_cast<const float4*>(w)[i];
        uint2 p;
        p.x = (unsigned)f2h(scale * (v.x - h2f(f2h(v.x)))) | ((unsigned)f2h(scale * (v.y - h2f(f2h(v.y)))) << 16);
        p.y = (unsigned)f2h(scale * (v.z - h2f(f2h(v.z)))) | ((unsigned)f2h(scale * (v.w - h2f(f2h(v.w)))) << 16);
        reinterpret_cast<uint2*>(out)[i] = p;
    }
}
// e4m3 (OCP) images for the fp8 lo product (GemmArgs.k8); pack4_e4m3: common.h
// rows of [K f16 | K e4m3] (pitch ld halfs): the e4m3 half = 2^-2 x the f16 half
__global__ __launch_bounds__(256) void fp8_tail_kernel(bf16_t* __restrict__ x, int M, int K, int ld) {
    const int per_row = K / 8;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < (size_t)M * per_row; i += (size_t)gridDim.x * 256) {
        const size_t m = i / per_row;
        const int c = (int)(i - m * per_row);
        bf16_t* row = x + m * ld;
        const uint4 v = *reinterpret_cast<const uint4*>(row + 8 * c);
        uint2 o;
        o.x = e4m3x4_of_h4(v.x, v.y);
        o.y = e4m3x4_of_h4(v.z, v.w);
        *reinterpret_cast<uint2*>(reinterpret_cast<unsigned char*>(row + K) + 8 * c) = o;
    }
}
extern "C" int sed_fp8_tail(void* x, int M, int K, int ld, hipStream_t stream) {
    (void)hipGetLastError();
    if (M <= 0 || K % 8 || ld % 8 || ld < K + K / 2) return SED_ERR_ARG;
    const size_t n = (size_t)M * (K / 8);
    hipLaunchKernelGGL(fp8_tail_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 8192)), dim3(256), 0, stream, (bf16_t*)x, M, K, ld);
    return sed_check_launch();
}
// fp32 weight [N, K] -> rows [f16(W) | e4m3(2^s (W - f16(W)))] (3 K bytes per row)
__global__ __launch_bounds__(256) void weight_two_term_f8_kernel(const float* __restrict__ w, unsigned char* __restrict__ out, size_t N, int K, float scale) {
    const int per_row = K / 4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < N * per_row; i += (size_t)gridDim.x * 256) {
        const size_t n = i / per_row;
        const int c = (int)(i - n * per_row);
        const float4 v = *reinterpret_cast<const float4*>(w + n * K + 4 * c);
        const _Float16 h0 = (_Float16)v.x, h1 = (_Float16)v.y, h2 = (_Float16)v.z, h3 = (_Float16)v.w;
        unsigned char* row = out + n * (size_t)(3 * K);
        typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
        *reinterpret_cast<f16x4_t*>(row + 8 * c) = f16x4_t{h0, h1, h2, h3};
        *reinterpret_cast<unsigned*>(row + 2 * K + 4 * c) =
            pack4_e4m3(scale * (v.x - (float)h0), scale * (v.y - (float)h1), scale * (v.z - (float)h2), scale * (v.w - (float)h3));
    }
}
extern "C" int sed_weight_two_term_f8(const float* w, void* out, int64_t N, int K, int s, hipStream_t stream) {
    (void)hipGetLastError();
    if (N <= 0 || K % 8 || s < -100 || s > 100) return SED_ERR_ARG;
    const size_t n = (size_t)N * (K / 4);
    hipLaunchKernelGGL(weight_two_term_f8_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 8192)), dim3(256), 0, stream, w,
                       (unsigned char*)out, (size_t)N, K, ldexpf(1.f, s));
    return sed_check_launch();
}
extern "C" int sed_weight_residual_f16(const float* w, void* out, int64_t n, float scale, hipStream_t stream) {
    (void)hipGetLastError();
    if (n <= 0 || n % 4) return SED_ERR_ARG;
    const size_t n4 = n / 4;
    int blocks = (int)((n4 + 255) / 256);
    blocks = blocks > 4096 ? 4096 : blocks;
    hipLaunchKernelGGL(weight_residual_kernel, dim3(blocks), dim3(256), 0, stream, w, (bf16_t*)out, n4, scale);
    return sed_check_launch();
}

// rowpart [M][S][2] (per-row partial sums of S column slices) -> rowstat [M][2] = (mean, 1 / sqrt(var + eps)) over D columns
__global__ void ln_fold_stats_kernel(const float* __restrict__ rowpart, float* __restrict__ rowstat, int M, int S, float invD, float eps) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    float s1 = 0.f, s2 = 0.f;
    for (int i = 0; i < S; ++i) {
        const float2 p = *reinterpret_cast<const float2*>(rowpart + ((size_t)m * S + i) * 2);
        s1 += p.x; s2 += p.y;
    }
    const float mean = s1 * invD;
    float var = s2 * invD - mean * mean;
    var = var > 0.f ? var : 0.f;
    *reinterpret_cast<float2*>(rowstat + 2 * (size_t)m) = make_float2(mean, 1.0f / sqrtf(var + eps));
}
extern "C" int sed_ln_fold_stats(const float* rowpart, float* rowstat, int M, int S, int D, float eps, hipStream_t stream) {
    (void)hipGetLastError();
    if (M <= 0 || S <= 0 || D <= 0) return SED_ERR_ARG;
    hipLaunchKernelGGL(ln_fold_stats_kernel, dim3((M + 255) / 256), dim3(256), 0, stream, rowpart, rowstat, M, S, 1.0f / (float)D, eps);
    return sed_check_launch();
}
// Weight side of the fold: W16[n, k] = f16(gamma[k] W[n, k]); colS[n] = sum_k W16[n, k] (of the ROUNDED values: the mean component the
// GEMM accumulates is then cancelled exactly); colC[n] = sum_k beta[k] W[n, k] + bias[n].  One wave per output row.
__global__ __launch_bounds__(256) void ln_fold_weight_kernel(const float* __restrict__ W, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, const float* __restrict__ bias,
                                                             bf16_t* __restrict__ W16, float* __restrict__ colS, float* __restrict__ colC,
                                                             int N, int K) {
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (n >= N) return;
    float s = 0.f, c = 0.f;
    for (int k = lane * 4; k < K; k += 256) {
        const float4 w = *reinterpret_cast<const float4*>(W + (size_t)n * K + k), gm = *reinterpret_cast<const float4*>(gamma + k),
                     bt = *reinterpret_cast<const float4*>(beta + k);
        const bf16_t h0 = f2h(w.x * gm.x), h1 = f2h(w.y * gm.y), h2 = f2h(w.z * gm.z), h3 = f2h(w.w * gm.w);
        uint2 pk;
        pk.x = (unsigned)h0 | ((unsigned)h1 << 16);
        pk.y = (unsigned)h2 | ((unsigned)h3 << 16);
        *reinterpret_cast<uint2*>(W16 + (size_t)n * K + k) = pk;
        s += (h2f(h0) + h2f(h1)) + (h2f(h2) + h2f(h3));
        c += (bt.x * w.x + bt.y * w.y) + (bt.z * w.z + bt.w * w.w);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); c += __shfl_xor(c, o, 64); }
    if (lane == 0) { colS[n] = s; colC[n] = c + (bias != nullptr ? bias[n] : 0.f); }
}
extern "C" int sed_ln_fold_weight(const float* W, const float* gamma, const float* beta, const float* bias, void* W16, float* colS,
                                  float* colC, int N, int K, hipStream_t stream) {
    (void)hipGetLastError();
    if (N <= 0 || K <= 0 || (K % 4)) return SED_ERR_ARG;
    hipLaunchKernelGGL(ln_fold_weight_kernel, dim3((N + 3) / 4), dim3(256), 0, stream, W, gamma, beta, bias, (bf16_t*)W16, colS, colC, N, K);
    return sed_check_launch();
}

// Split-precision operand images (f16 hi + f16 lo carries ~22 significand bits): a GEMM over the concatenated reduction
// dimension [A_hi | A_lo | A_hi] . [W_hi | W_hi | W_lo]^T accumulates A_hi W_hi + A_lo W_hi + A_hi W_lo in fp32 inside the
// ordinary MFMA kernel.  Used for the context-network GEMMs, whose operand rounding dominates the posterior error.
// in fp32 [M, K] -> out f16 [M, 3K]; mode 0: [hi | lo | hi] (activations), mode 1: [hi | hi | lo] (weights);
// mode 2: out f16 [M, 2K] = [hi | lo], the two-term weight image of sed_gemm_nt_w2 / sed_gemm_qkv_w2
__global__ void split3_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, size_t M, int K, int mode) {
    const size_t total = M * (size_t)(K / 2);
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const size_t m = idx / (K / 2);
        const int k = (int)(idx - m * (K / 2)) * 2;
        const float2 v = *reinterpret_cast<const float2*>(in + m * K + k);
        const bf16_t h0 = f2h(v.x), h1 = f2h(v.y);
        const bf16_t l0 = f2h(v.x - h2f(h0)), l1 = f2h(v.y - h2f(h1));
        const unsigned hi = (unsigned)h0 | ((unsigned)h1 << 16), lo = (unsigned)l0 | ((unsigned)l1 << 16);
        unsigned* row = reinterpret_cast<unsigned*>(out + m * (size_t)((mode == 2 ? 2 : 3) * K));
        row[k / 2] = hi;
        row[(K + k) / 2] = mode == 1 ? hi : lo;
        if (mode != 2) row[(2 * K + k) / 2] = mode == 0 ? hi : lo;
    }
}
extern "C" int sed_split3_f16(const float* in, void* out, int64_t M, int K, int mode, hipStream_t stream) {
    (void)hipGetLastError();
    if (K % 2 || M <= 0 || mode < 0 || mode > 2) return SED_ERR_ARG;
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
