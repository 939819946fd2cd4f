// DASM (open-vocabulary sound event detection, BASELINE.json config #5) -- the query decoder and the dual-stream head, forward and backward, gfx950.
//
// Replaces src/models/detect_any_sound/at_adapter.py:7-50 (nn.TransformerDecoder of cross-attention-first layers over the backbone's
// patch tokens) and src/models/detect_any_sound/detect_any_sound.py:283-322, 362-389 (query projector, at_head, sed_head,
// mask_embedding MLP, einsum('bqc,bct->bqt'), sigmoid(x / temp) * at_out, pad mask, clamp, linear-softmax pooling).
//
// Everything on the query side is small (Q = 10 .. a few hundred queries per clip against 1188 patch tokens / 1000 frames) and sits
// directly in front of a sigmoid with temperature 0.1 .. 0.5, i.e. it is precision-critical: the kernels here take and return fp32 and keep
// fp32-level products (exact fp32 MFMA products in the GEMM; since round 6 three-term split-precision products, 2^-22 / 2^-17, in the
// attention, see "Round 6 (second half)" below) with fp32 accumulation --
//   * sed_gemm_f32_nt     C = act(A . B^T + bias) (+ residual), batched, on the fp32-input matrix instruction v_mfma_f32_32x32x2_f32
//                         (exact fp32 products and accumulation at the fp32 vector rate, without one VALU instruction per FMA and
//                         with a 64 x 64 output tile fed from LDS: MI355X_MICROARCH.md "FP32-input MFMA");
//   * sed_xattn_f32_fwd   softmax(q k^T / sqrt(dh) + mask) v for query counts that differ from the key count (cross attention over the
//                         patch tokens, self attention among the queries with the open-vocabulary mask), 32 queries per workgroup, four
//                         waves splitting the key tiles;
//   * sed_dasm_head_fwd   the dual-stream finish on the [B, T, Q] logits: sigmoid / temperature, times the clip-level tagging
//                         probability, pad mask, clamp, transposed store [B, Q, T], linear-softmax pooling.
#include "common.h"
#include "../../include/sed_hip.h"

// ---------------------------------------------------------------------------------------------------------------------
// fp32 GEMM:  C[z][m][n] (+)= drop(act(sum_k A(z, m, k) B(z, n, k) + bias[n])) (+ R[z][m][n])
//   A(m, k) = transA ? A[k lda + m] : A[m lda + k]        B(n, k) = transB ? B[k ldb + n] : B[n ldb + k]
// which covers the three products of a Linear under autograd -- y = x W^T (NT: transA 0, transB 0), dx = dy W (transB 1),
// dW = dy^T x (transA 1, transB 1; split over the tokens with fp32 atomics into the zero-initialised / accumulating gradient) -- and the
// einsum('bqc,bct->bqt') of the dual-stream head with its two gradients (batched).
// 256 threads = 4 waves, 64 x 64 tile, wave (wm, wn) owns a 32 x 32 block = ONE accumulator tile of v_mfma_f32_32x32x2_f32;
// K tiles of 32 staged k-major in LDS ([k][m], row pitch 65 floats: the 4-scalar writes of a float4 and the 32-lane
// fragment reads are both bank-conflict free), next tile's global loads in flight during the 16 MFMAs of the current one.
// SAFE: operands whose rows are not float4-addressable (leading dimension or contraction length not a multiple of 4, a one-column
// tagging head) are read element by element with bounds checks; everything else takes float4 loads.
// ---------------------------------------------------------------------------------------------------------------------
#define F32_BK 32
#define F32_LD 65
typedef __attribute__((ext_vector_type(16))) float f32x16;

// counter-based dropout bits: keep(seed, site, element index) -- the forward and the backward kernels of one dropout site evaluate the same
// function (no mask tensor is stored), `sed_dropout_f32` dumps it for the tests' CPU oracle.  One splitmix64 finaliser serves FOUR
// consecutive elements (its four 16-bit fields against a 16-bit threshold: p to 1.5e-5): the attention kernels hold four consecutive keys
// in four consecutive accumulator registers, so a lane hashes once per register quad (the per-element hash was ~150 issue cycles beside
// the 64 of the MFMA it sits next to).  (Round 6, with the products on the 16-bit pipe: replacing the finaliser by one multiply-add changes the
// three attention kernels by 3-4 % -- 354 -> 342, 289 -> 279, 256 -> 247 us at the dasm_train shape: a cheaper hash is not worth its risk.)
__device__ __forceinline__ unsigned long long drop_hash4(unsigned long long seed, unsigned sid, unsigned long long idx4) {
    unsigned long long z = idx4 + (seed ^ ((unsigned long long)sid << 48)) * 0x9E3779B97F4A7C15ull + 0xD1B54A32D192ED03ull * (sid + 1u);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ bool drop_keep_of(unsigned long long z, unsigned lane4, unsigned thr16) { return ((unsigned)(z >> (16u * lane4)) & 0xffffu) >= thr16; }
__device__ __forceinline__ bool drop_keep(unsigned long long seed, unsigned sid, unsigned long long idx, unsigned thr16) {
    return drop_keep_of(drop_hash4(seed, sid, idx >> 2), (unsigned)idx & 3u, thr16);
}
static inline unsigned drop_thr24(float p) { return p <= 0.f ? 0u : (unsigned)(p * 65536.0f + 0.5f); }      // (16-bit threshold; name kept)

struct GemmF32Args {
    const float *A, *B, *bias, *R;
    float *C, *pre;
    int M, N, K, lda, ldb, ldc, ksplit, act, accumulate;
    long long sA, sB, sC;
    unsigned drop_thr, drop_sid;
    float drop_scale;
    unsigned long long drop_seed;
};

template <bool TA, bool TB, bool SAFE>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmF32Args g) {
    __shared__ float As[F32_BK * F32_LD], Bs[F32_BK * F32_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int z = blockIdx.z / g.ksplit, ks = blockIdx.z - z * g.ksplit;
    const int M = g.M, N = g.N;
    const float* A = g.A + (size_t)z * g.sA;
    const float* B = g.B + (size_t)z * g.sB;
    float* C = g.C + (size_t)z * g.sC;
    // contraction range of this split (multiples of the K tile)
    const int kchunk = ((g.K + g.ksplit - 1) / g.ksplit + F32_BK - 1) / F32_BK * F32_BK;
    const int kbeg = ks * kchunk, kend = min(g.K, kbeg + kchunk);
    if (kbeg >= kend && g.accumulate) return;
    float4 ra[2], rb[2];
    // loaders.  k-contiguous operand: thread -> (row = tid / 8 (+ 32), k quad = tid % 8); row-contiguous operand (transposed):
    // thread -> (k = tid / 8, row quad = tid % 8 (+ 8)): LDS element [k][row] either way
    const int l_hi = tid >> 3, l_lo = (tid & 7) * 4;
    auto load4 = [&](const float* base, int ld, bool trans, int row0, int rows, int k0, int i) -> float4 {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!trans) {
            const int r = row0 + l_hi + 32 * i, k = k0 + l_lo;
            if (r < rows) {
                const float* p = base + (size_t)r * ld + k;
                if (!SAFE) { if (k < kend) v = *reinterpret_cast<const float4*>(p); }      // (K and the split boundaries are multiples of 4 here)
                else {
                    if (k + 0 < kend) v.x = p[0];
                    if (k + 1 < kend) v.y = p[1];
                    if (k + 2 < kend) v.z = p[2];
                    if (k + 3 < kend) v.w = p[3];
                }
            }
        } else {
            const int k = k0 + l_hi, r = row0 + l_lo + 32 * i;
            if (k < kend) {
                const float* p = base + (size_t)k * ld + r;
                if (!SAFE && r + 3 < rows) { v = *reinterpret_cast<const float4*>(p); }
                else {
                    if (r + 0 < rows) v.x = p[0];
                    if (r + 1 < rows) v.y = p[1];
                    if (r + 2 < rows) v.z = p[2];
                    if (r + 3 < rows) v.w = p[3];
                }
            }
        }
        return v;
    };
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            ra[i] = load4(A, g.lda, TA, m0, M, k0, i);
            rb[i] = load4(B, g.ldb, TB, n0, N, k0, i);
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (!TA) {
                const int r = l_hi + 32 * i;
                As[(l_lo + 0) * F32_LD + r] = ra[i].x; As[(l_lo + 1) * F32_LD + r] = ra[i].y; As[(l_lo + 2) * F32_LD + r] = ra[i].z; As[(l_lo + 3) * F32_LD + r] = ra[i].w;
            } else {
                float* d = As + l_hi * F32_LD + l_lo + 32 * i;
                d[0] = ra[i].x; d[1] = ra[i].y; d[2] = ra[i].z; d[3] = ra[i].w;
            }
            if (!TB) {
                const int r = l_hi + 32 * i;
                Bs[(l_lo + 0) * F32_LD + r] = rb[i].x; Bs[(l_lo + 1) * F32_LD + r] = rb[i].y; Bs[(l_lo + 2) * F32_LD + r] = rb[i].z; Bs[(l_lo + 3) * F32_LD + r] = rb[i].w;
            } else {
                float* d = Bs + l_hi * F32_LD + l_lo + 32 * i;
                d[0] = rb[i].x; d[1] = rb[i].y; d[2] = rb[i].z; d[3] = rb[i].w;
            }
        }
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int fa = (lane >> 5) * F32_LD + wm * 32 + (lane & 31), fb = (lane >> 5) * F32_LD + wn * 32 + (lane & 31);
    if (kbeg < kend) gload(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += F32_BK) {
        lstore();
        __syncthreads();
        if (k0 + F32_BK < kend) gload(k0 + F32_BK);
#pragma unroll
        for (int kk = 0; kk < F32_BK; kk += 2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[kk * F32_LD + fa], Bs[kk * F32_LD + fb], acc, 0, 0, 0);
        __syncthreads();
    }
    const int n = n0 + wn * 32 + (lane & 31);
    if (n >= N) return;
    if (g.accumulate) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * 32 + mfma32_row(r, lane >> 5);
            if (m < M) unsafeAtomicAdd(&C[(size_t)m * g.ldc + n], acc[r]);
        }
        return;
    }
    const float* R = g.R != nullptr ? g.R + (size_t)z * g.sC : nullptr;
    float* pre = g.pre != nullptr ? g.pre + (size_t)z * g.sC : nullptr;
    const float bn = g.bias != nullptr ? g.bias[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 32 + mfma32_row(r, lane >> 5);
        if (m >= M) continue;
        float v = acc[r] + bn;
        if (pre != nullptr) pre[(size_t)m * g.ldc + n] = v;
        if (g.act == 1) v = gelu_fast(v);
        if (g.act == 2) v = fmaxf(v, 0.f);
        if (g.drop_thr != 0u) v = drop_keep(g.drop_seed, g.drop_sid, ((unsigned long long)z * M + m) * N + n, g.drop_thr) ? v * g.drop_scale : 0.f;
        if (R != nullptr) v += R[(size_t)m * g.ldc + n];
        C[(size_t)m * g.ldc + n] = v;
    }
}

extern "C" int sed_gemm_f32(const float* A, const float* B, const float* bias, const float* R, float* C, float* pre, int M, int N, int K,
                            int lda, int ldb, int ldc, int transA, int transB, int batch, int64_t strideA, int64_t strideB, int64_t strideC,
                            int act, int accumulate, int ksplit, float drop_p, int64_t drop_seed, int drop_site, hipStream_t stream) {
    (void)hipGetLastError();
    if (M <= 0 || N <= 0 || K <= 0 || batch < 1 || ksplit < 1 || (int64_t)batch * ksplit > 65535 || act < 0 || act > 2 || !(drop_p >= 0.f && drop_p < 1.f))
        return SED_ERR_ARG;
    if (accumulate && (bias != nullptr || R != nullptr || pre != nullptr || act != 0 || drop_p > 0.f)) return SED_ERR_ARG;
    if (ksplit > 1 && !accumulate) return SED_ERR_ARG;
    // float4 path: both operands 16-byte addressable along their contiguous dimension; an operand whose contiguous dimension is the
    // contraction needs its length to be a multiple of 4 (the tail of the last K tile is masked per float4), a row-contiguous
    // (transposed) operand takes any contraction length -- the token count of a weight gradient
    const bool vec = !(((uintptr_t)A | (uintptr_t)B) & 15) && !((lda | ldb) & 3) && !((strideA | strideB) & 3) && ((transA && transB) || (K & 3) == 0);
    GemmF32Args g;
    g.A = A; g.B = B; g.bias = bias; g.R = R; g.C = C; g.pre = pre;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ksplit = ksplit; g.act = act; g.accumulate = accumulate;
    g.sA = strideA; g.sB = strideB; g.sC = strideC;
    g.drop_thr = drop_thr24(drop_p); g.drop_sid = (unsigned)drop_site; g.drop_scale = 1.0f / (1.0f - drop_p); g.drop_seed = (unsigned long long)drop_seed;
    const dim3 grid(cdiv(N, 64), cdiv(M, 64), batch * ksplit);
#define GEMM_F32_LAUNCH(TA_, TB_)                                                                                       \
    {                                                                                                                   \
        if (vec) hipLaunchKernelGGL((gemm_f32_kernel<TA_, TB_, false>), grid, dim3(256), 0, stream, g);                 \
        else hipLaunchKernelGGL((gemm_f32_kernel<TA_, TB_, true>), grid, dim3(256), 0, stream, g);                      \
    }
    if (!transA && !transB) GEMM_F32_LAUNCH(false, false)
    else if (!transA && transB) GEMM_F32_LAUNCH(false, true)
    else if (transA && !transB) GEMM_F32_LAUNCH(true, false)
    else GEMM_F32_LAUNCH(true, true)
#undef GEMM_F32_LAUNCH
    return sed_check_launch();
}

extern "C" int sed_gemm_f32_nt(const float* A, const float* B, const float* bias, const float* R, float* C, int M, int N, int K, int lda,
                               int ldb, int ldc, int batch, int64_t strideA, int64_t strideB, int64_t strideC, int act,
                               hipStream_t stream) {
    return sed_gemm_f32(A, B, bias, R, C, nullptr, M, N, K, lda, ldb, ldc, 0, 0, batch, strideA, strideB, strideC, act, 0, 1, 0.f, 0, 0, stream);
}

// ---------------------------------------------------------------------------------------------------------------------
// attention with Nq != Nk, fp32, on v_mfma_f32_32x32x2_f32:   O[b, i, h*DH + d] = sum_j softmax_j(q_i . k_j / sqrt(DH) + mask_ij) v_j[d]
// One workgroup per (32 queries, head, clip); its four waves take every fourth 32-key tile each and merge their partial (max, sum,
// output) triples through LDS at the end (log-sum-exp combination).  Per tile and wave, in the "swapped" form of the encoder attention
// (a lane owns one query column of every accumulator: lane = (query q = lane & 31, half g = lane >> 5)):
//   S^T[key, q]  = sum_d K[key, d] Q[q, d]     DH / 2 MFMAs; A = K rows out of the wave-private LDS tile (row pitch DH + 4 floats), B = the
//                                              query row held in DH / 2 registers (pre-scaled by log2(e) / sqrt(DH))
//   online softmax along the 16 accumulator registers + one cross-half shuffle
//   O^T[d, q]   += sum_key V^T[d, key] P^T[key, q]     DH / 2 MFMAs; B = the S^T accumulator registers themselves (register r of the two
//                                              halves holds exactly the key pair (row(r, 0), row(r, 1)) of one k step), A = V rows from LDS
// fp32 operands and accumulation throughout (the matrix instruction runs at the fp32 vector rate, but without an LDS operand read and a
// VALU issue slot per FMA: the first version of this kernel, one lane per query with broadcast K / V reads, spent 128 FMA instructions and
// 32 ds_read_b128 per key and wave).  Q / K / V rows are addressed through their own leading dimensions (packed in_proj outputs are read
// in place); mask [Nq, Nk] bytes, non-zero = not allowed.
// ---------------------------------------------------------------------------------------------------------------------
#define XA_WAVES 4
#define XA_KT 32
// Round 6 (second half): every product of the three attention kernels as a THREE-TERM split-precision product on the 16-bit matrix pipe,
//   x y ~ xh yh + xl yh + xh yl,   xh = rn16(x), xl = rn16(x - xh),   fp32 accumulation (v_mfma_f32_32x32x16_{f16,bf16}: same accumulator
// layout as the fp32 instruction, a lane half supplies EIGHT consecutive k values instead of one) -- 12 x 32 issue cycles per 32 x 32 x 64
// product instead of 32 x 64, for ~16 vector instructions per operand octet.  The dropped term xl yl is 2^-22 of the product with IEEE-half
// pairs (scores and probabilities: q, k, v, P -- O(1) values) and 2^-17 with bf16 pairs (everything that carries a gradient: dO, dS, the
// dropped probabilities they meet, and the tile rows of the products that contract over the streamed rows -- bf16 keeps the fp32 exponent
// range, loss-scaled gradients of 1e-8 survive).  The forward's S and the backward's recomputed S use the same operand split and the same
// accumulation order, so P is reproduced exactly in the query-stationary pass.  Range: the half pairs need |q| / sqrt(dh), |k|, |v| < 65504
// (LayerNorm-ed activations through a Linear: O(1 .. 100)); below 6e-5 a half hi term is subnormal and the pair carries less than 22 bits of a
// value that no longer matters beside O(1) neighbours.  The fp32-input-MFMA form these kernels had until then
// (v_mfma_f32_32x32x2_f32, exact fp32 products) measured 334 / 457 / 955 us for forward / query-stationary / key-stationary backward at the
// dasm_train shape (24 clips, 407 queries over 1188 tokens) against 258 / 303 / 630 now; it is in the history (commit 661841a), not in the tree.
template <bool F16> __device__ __forceinline__ void xa_split8(const float (&x)[8], s16x8_t& hi, s16x8_t& lo) {
    unsigned h[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        h[i] = pack2<F16>(x[2 * i], x[2 * i + 1]);
        l[i] = pack2<F16>(x[2 * i] - to_f32<F16>((bf16_t)(h[i] & 0xffffu)), x[2 * i + 1] - to_f32<F16>((bf16_t)(h[i] >> 16)));
    }
    hi = __builtin_bit_cast(s16x8_t, make_uint4(h[0], h[1], h[2], h[3]));
    lo = __builtin_bit_cast(s16x8_t, make_uint4(l[0], l[1], l[2], l[3]));
}
// c += (ah + al) (bh + bl) without the al bl term; A rows / B columns as in mfma32t
template <bool F16> __device__ __forceinline__ f32x16_t xa_mfma3(s16x8_t ah, s16x8_t al, s16x8_t bh, s16x8_t bl, f32x16_t c) {
    c = mfma32t<F16>(ah, bh, c);
    c = mfma32t<F16>(al, bh, c);
    return mfma32t<F16>(ah, bl, c);
}
// eight consecutive floats of a tile row (16-byte aligned) / eight elements of one tile COLUMN at the rows a lane's accumulator registers
// 8 m .. 8 m + 7 stand for
__device__ __forceinline__ void xa_row8(const float* p, float (&x)[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
}
template <int LDK_> __device__ __forceinline__ void xa_col8(const float* tile, int m, int lg, int col, float (&x)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = tile[mfma32_row(8 * m + i, lg) * LDK_ + col];
}
// A wave's next pair of tile half-rows on its way from global memory to LDS: N4 = DH / 8 float4 of each tile per lane.  Named members, not
// arrays: a loop-carried float4 array that is assigned under a condition is left in scratch memory by the compiler (global load ->
// scratch store -> scratch load -> LDS store, with the load's latency waited out at once -- measured in the first version of the prefetch).
struct XaTileRegs { f32x4_t a0, a1, a2, a3, a4, a5, a6, a7, b0, b1, b2, b3, b4, b5, b6, b7; };      // (native vectors: a HIP float4 assignment from global memory is a memcpy the optimiser does not split)
template <int N4> __device__ __forceinline__ void xa_regs_gload(XaTileRegs& r, const float* g0, const float* g1) {
    const f32x4_t *p0 = reinterpret_cast<const f32x4_t*>(g0), *p1 = reinterpret_cast<const f32x4_t*>(g1);
    r.a0 = p0[0]; r.a1 = p0[1]; r.a2 = p0[2]; r.a3 = p0[3];
    r.b0 = p1[0]; r.b1 = p1[1]; r.b2 = p1[2]; r.b3 = p1[3];
    if constexpr (N4 == 8) {
        r.a4 = p0[4]; r.a5 = p0[5]; r.a6 = p0[6]; r.a7 = p0[7];
        r.b4 = p1[4]; r.b5 = p1[5]; r.b6 = p1[6]; r.b7 = p1[7];
    }
}
template <int N4> __device__ __forceinline__ void xa_regs_lstore(const XaTileRegs& r, float* t0, float* t1) {
    f32x4_t *q0 = reinterpret_cast<f32x4_t*>(t0), *q1 = reinterpret_cast<f32x4_t*>(t1);
    q0[0] = r.a0; q0[1] = r.a1; q0[2] = r.a2; q0[3] = r.a3;
    q1[0] = r.b0; q1[1] = r.b1; q1[2] = r.b2; q1[3] = r.b3;
    if constexpr (N4 == 8) {
        q0[4] = r.a4; q0[5] = r.a5; q0[6] = r.a6; q0[7] = r.a7;
        q1[4] = r.b4; q1[5] = r.b5; q1[6] = r.b6; q1[7] = r.b7;
    }
}
struct XaDrop {      // attention-probability dropout of one site (torch.nn.MultiheadAttention(dropout=p) in train mode); thr 0 = off
    unsigned thr, sid;
    float scale;
    unsigned long long seed;
};
template <int DH, bool MASK, bool TRAIN>
__global__ __launch_bounds__(64 * XA_WAVES, 2) void xattn_f32_fwd_kernel(const float* __restrict__ Q, const float* __restrict__ Kp,
                                                                      const float* __restrict__ Vp, float* __restrict__ O,
                                                                      const unsigned char* __restrict__ mask, int Nq, int Nk, int ldq, int ldk,
                                                                      int ldv, int ldo, long long q_bstride, float* __restrict__ lse_out,
                                                                      const XaDrop dr) {
    constexpr int LDK = DH + 4, NDB = DH / 32, MS = DH + 3;      // merge pitch: ODD -- the 32 query slots of a wave then sit in 32 different banks (DH + 2 was 2-way)
    extern __shared__ __attribute__((aligned(16))) float xa_lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), h = blockIdx.y, b = blockIdx.z;
    const int lq = lane & 31, lg = lane >> 5;
    float* Ks = xa_lds + wave * (2 * XA_KT * LDK);
    float* Vs = Ks + XA_KT * LDK;
    const int qi = blockIdx.x * 32 + lq;
    const int qc = qi < Nq ? qi : Nq - 1;
    const float sc = 1.4426950408889634f * rsqrtf((float)DH);
    // B operand of the score product: lane half lg holds d = 16 jb + 8 lg .. + 7 of its query's (pre-scaled) row as an IEEE-half hi / lo pair
    s16x8_t qh[DH / 16], ql[DH / 16];
    {
        const float* qp = Q + (size_t)b * q_bstride + (size_t)qc * ldq + h * DH + 8 * lg;
#pragma unroll
        for (int jb = 0; jb < DH / 16; ++jb) {
            float x[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = qp[16 * jb + i] * sc;
            xa_split8<true>(x, qh[jb], ql[jb]);
        }
    }
    f32x16 o[NDB];
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const float* kb = Kp + (size_t)b * Nk * ldk + h * DH;
    const float* vb = Vp + (size_t)b * Nk * ldv + h * DH;
    // tile loader: lane -> (key row lane & 31, half row lane >> 5): DH / 2 floats = DH / 8 float4 of K and of V.  (Eight consecutive lanes -- one
    // ds_write_b128 group -- are eight ROWS at 4 banks each = all 32 banks; with (row lane >> 1, half lane & 1) the two halves of a row, 32 floats
    // apart, met in the same banks: every tile store 2-way conflicted.)
    const int trow = lane & 31, thalf = (lane >> 5) * (DH / 2);
    // The wave's NEXT tile is in flight (in registers) while the current one is computed: with the products on the 16-bit pipe a tile is
    // ~1500 issue cycles, less than one trip to L2 / HBM -- loaded at the top of its own iteration, as the fp32 form did, the kernel waits.
    XaTileRegs tr;
#define XA_FWD_GLOAD(J0_)                                                                                                                  \
    {                                                                                                                                      \
        const int j_ = ((J0_) + trow) < Nk ? ((J0_) + trow) : Nk - 1;      /* rows past the end: their scores are masked, their V rows meet p = 0 */ \
        xa_regs_gload<DH / 8>(tr, kb + (size_t)j_ * ldk + thalf, vb + (size_t)j_ * ldv + thalf);                                            \
    }
    if (wave * XA_KT < Nk) XA_FWD_GLOAD(wave * XA_KT)
    for (int j0 = wave * XA_KT; j0 < Nk; j0 += XA_KT * XA_WAVES) {
        {
            __builtin_amdgcn_wave_barrier();      // (the tile buffers are private to the wave: the previous tile's reads are behind us in program order)
            xa_regs_lstore<DH / 8>(tr, Ks + trow * LDK + thalf, Vs + trow * LDK + thalf);
            __builtin_amdgcn_wave_barrier();
            if (j0 + XA_KT * XA_WAVES < Nk) XA_FWD_GLOAD(j0 + XA_KT * XA_WAVES)
        }
#undef XA_FWD_GLOAD
        // ---- S^T[key, q]: A = K[key = lq][2 j + lg]
        f32x16 st;
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
        for (int jb = 0; jb < DH / 16; ++jb) {      // ONE accumulator chain, k blocks in order: the backward recomputes S exactly like this
            float kx[8];
            xa_row8(Ks + lq * LDK + 16 * jb + 8 * lg, kx);
            s16x8_t kh, kl;
            xa_split8<true>(kx, kh, kl);
            st = xa_mfma3<true>(kh, kl, qh[jb], ql[jb], st);
        }
        // ---- online softmax over the lane's 16 keys (register r <-> key j0 + mfma32_row(r, lg)) and the other half's 16
        float cmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = j0 + mfma32_row(r, lg);
            bool dead = j >= Nk;
            if (MASK) dead = dead || mask[(size_t)qc * Nk + (j < Nk ? j : Nk - 1)] != 0;
            st[r] = dead ? -INFINITY : st[r];
            cmax = fmaxf(cmax, st[r]);
        }
        cmax = fmaxf(cmax, __shfl_xor(cmax, 32, 64));
        const float m_new = fmaxf(m_run, cmax);
        // (a tile whose keys are all masked for this query while nothing has been seen yet: m_new = -inf, alpha = 1, p = 0)
        const float alpha = m_new == -INFINITY ? 1.f : exp2f(m_run - m_new);
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            st[r] = m_new == -INFINITY ? 0.f : exp2f(st[r] - m_new);
            psum += st[r];
        }
        psum += __shfl_xor(psum, 32, 64);
        l_run = l_run * alpha + psum;
        m_run = m_new;
        if (TRAIN && dr.thr != 0u) {      // dropout acts on the normalised probabilities: the denominator above keeps every key
            const unsigned long long rowbase = (((unsigned long long)b * gridDim.y + h) * Nq + qc) * (unsigned long long)Nk;
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {      // registers 4 rq .. 4 rq + 3 = keys j0 + 8 rq + 4 lg + (0 .. 3)
                const int jq = j0 + 8 * rq + 4 * lg;
                if ((Nk & 3) == 0) {               // (a row's keys start on a multiple of 4 of the flat index: one hash per quad)
                    const unsigned long long z = drop_hash4(dr.seed, dr.sid, (rowbase + jq) >> 2);
#pragma unroll
                    for (int i = 0; i < 4; ++i) st[4 * rq + i] = (jq + i < Nk && drop_keep_of(z, i, dr.thr)) ? st[4 * rq + i] * dr.scale : 0.f;
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) st[4 * rq + i] = (jq + i < Nk && drop_keep(dr.seed, dr.sid, rowbase + jq + i, dr.thr)) ? st[4 * rq + i] * dr.scale : 0.f;
                }
            }
        }
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
        // ---- O^T[d, q] += V^T[d, key] P^T[key, q]: k step r = the key pair (row(r, 0), row(r, 1)); A = V[row(r, lg)][32 db + lq], B = st[r]
        // (k block m = the keys of the lane's accumulator registers 8 m .. 8 m + 7: B = those registers as a half pair, A = V[key][32 db + lq])
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            float px[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) px[i] = st[8 * m + i];
            s16x8_t ph, pl;
            xa_split8<true>(px, ph, pl);
#pragma unroll
            for (int db = 0; db < NDB; ++db) {
                float vx[8];
                xa_col8<LDK>(Vs, m, lg, lq + 32 * db, vx);
                s16x8_t vh, vl;
                xa_split8<true>(vx, vh, vl);
                o[db] = xa_mfma3<true>(vh, vl, ph, pl, o[db]);
            }
        }
    }
    // ---- merge the four waves' partial results: slot [wave][query][DH + 2]; lane (q, g) owns d = 32 db + mfma32_row(r, g)
    __syncthreads();
    {
        float* mg = xa_lds + (size_t)(wave * 32 + lq) * MS;
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) mg[32 * db + mfma32_row(r, lg)] = o[db][r];
        if (lg == 0) { mg[DH] = m_run; mg[DH + 1] = l_run; }
    }
    __syncthreads();
    // every wave finishes a quarter of the output columns of the 32 queries: lane -> (query lq, column group)
    {
        constexpr int CW = DH / (2 * XA_WAVES);            // columns per lane: DH / 8
        const int d0 = (wave * 2 + lg) * CW;
        float mx = -INFINITY;
#pragma unroll
        for (int w = 0; w < XA_WAVES; ++w) mx = fmaxf(mx, xa_lds[(size_t)(w * 32 + lq) * MS + DH]);
        float l = 0.f, acc[CW];
#pragma unroll
        for (int d = 0; d < CW; ++d) acc[d] = 0.f;
#pragma unroll
        for (int w = 0; w < XA_WAVES; ++w) {
            const float* pw = xa_lds + (size_t)(w * 32 + lq) * MS;
            const float f = pw[DH] == -INFINITY ? 0.f : exp2f(pw[DH] - mx);
            l = fmaf(pw[DH + 1], f, l);
#pragma unroll
            for (int d = 0; d < CW; ++d) acc[d] = fmaf(pw[d0 + d], f, acc[d]);
        }
        if (qi < Nq) {
            const float inv = 1.0f / l;      // (a query with every key masked: 0 / 0 = NaN, like torch's softmax over an all -inf row)
            if (TRAIN && lse_out != nullptr && wave == 0 && lg == 0)      // log2-domain log-sum-exp of the scaled scores: what the backward re-normalises with
                lse_out[((size_t)b * gridDim.y + h) * Nq + qi] = mx + log2f(l);
            float* op = O + ((size_t)b * Nq + qi) * ldo + h * DH + d0;
#pragma unroll
            for (int d = 0; d < CW; d += 4) *reinterpret_cast<float4*>(op + d) = make_float4(acc[d] * inv, acc[d + 1] * inv, acc[d + 2] * inv, acc[d + 3] * inv);
        }
    }
}

static int xattn_fwd_launch(const float* Q, const float* K, const float* V, float* O, const uint8_t* mask, int B, int H, int Nq, int Nk,
                            int head_dim, int ldq, int ldk, int ldv, int ldo, int64_t q_batch_stride, float* lse, float drop_p, int64_t seed, int site,
                            bool train, hipStream_t stream) {
    (void)hipGetLastError();
    if (B <= 0 || H <= 0 || Nq <= 0 || Nk <= 0 || (head_dim != 32 && head_dim != 64) || ((ldq | ldk | ldv | ldo) & 3) || B > 65535 || H > 65535 ||
        !(drop_p >= 0.f && drop_p < 1.f))
        return SED_ERR_ARG;
    if (((uintptr_t)Q | (uintptr_t)K | (uintptr_t)V | (uintptr_t)O) & 15) return SED_ERR_ARG;
    XaDrop dr;
    dr.thr = drop_thr24(drop_p); dr.sid = (unsigned)site; dr.scale = 1.0f / (1.0f - drop_p); dr.seed = (unsigned long long)seed;
    const dim3 grid(cdiv(Nq, 32), H, B);
#define XATTN_LAUNCH(DH_, MK_, TR_)                                                                                              \
    {                                                                                                                            \
        const int tile_ = XA_WAVES * 2 * XA_KT * (DH_ + 4) * 4, merge_ = XA_WAVES * 32 * (DH_ + 3) * 4;                          \
        const int lds_ = tile_ > merge_ ? tile_ : merge_;                                                                        \
        static bool attr_ = false;                                                                                               \
        if (!attr_) { (void)hipFuncSetAttribute((const void*)xattn_f32_fwd_kernel<DH_, MK_, TR_>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_); attr_ = true; } \
        hipLaunchKernelGGL((xattn_f32_fwd_kernel<DH_, MK_, TR_>), grid, dim3(64 * XA_WAVES), lds_, stream, Q, K, V, O, mask, Nq, Nk, ldq, ldk, ldv, ldo, \
                           (long long)q_batch_stride, lse, dr);                                                                  \
    }
#define XATTN_PICK(TR_)                                                                                                          \
    if (head_dim == 64) { if (mask != nullptr) XATTN_LAUNCH(64, true, TR_) else XATTN_LAUNCH(64, false, TR_) }                   \
    else { if (mask != nullptr) XATTN_LAUNCH(32, true, TR_) else XATTN_LAUNCH(32, false, TR_) }
    if (train) { XATTN_PICK(true) } else { XATTN_PICK(false) }
#undef XATTN_PICK
#undef XATTN_LAUNCH
    return sed_check_launch();
}
extern "C" int sed_xattn_f32_fwd(const float* Q, const float* K, const float* V, float* O, const uint8_t* mask, int B, int H, int Nq, int Nk,
                                 int head_dim, int ldq, int ldk, int ldv, int ldo, int64_t q_batch_stride, hipStream_t stream) {
    return xattn_fwd_launch(Q, K, V, O, mask, B, H, Nq, Nk, head_dim, ldq, ldk, ldv, ldo, q_batch_stride, nullptr, 0.f, 0, 0, false, stream);
}
extern "C" int sed_xattn_f32_fwd_train(const float* Q, const float* K, const float* V, float* O, const uint8_t* mask, float* lse, int B, int H,
                                       int Nq, int Nk, int head_dim, int ldq, int ldk, int ldv, int ldo, int64_t q_batch_stride, float drop_p,
                                       int64_t drop_seed, int drop_site, hipStream_t stream) {
    if (lse == nullptr) return SED_ERR_ARG;
    return xattn_fwd_launch(Q, K, V, O, mask, B, H, Nq, Nk, head_dim, ldq, ldk, ldv, ldo, q_batch_stride, lse, drop_p, drop_seed, drop_site, true,
                            stream);
}

// ---------------------------------------------------------------------------------------------------------------------
// backward of the attention above (autograd of torch.nn.MultiheadAttention's softmax(q k^T / sqrt(dh) + mask) -> dropout -> . v), fp32 on the
// same matrix instruction, two launches of one kernel body:
//   MODE 0  query-stationary: a lane owns one query column, the four waves stream the key tiles;  dQ = (P o (dP - D)) K / sqrt(dh), and the
//           row constants D_i = dO_i . O_i go to `Dq` for the second launch;
//   MODE 1  key-stationary: a lane owns one key column, the waves stream (Q, dO) tiles;  dV = Pd^T dO,  dK = (P o (dP - D))^T Q / sqrt(dh).
// P is recomputed from the saved log-sum-exp (log2 domain), Pd = P o keep / (1 - p) with the dropout bits of the forward re-evaluated
// (drop_keep); dP = (dO V^T) o keep / (1 - p).  Per streamed 32-row tile and wave: DH MFMAs for S and dO V^T (A = the tile's rows out of
// wave-private LDS, B = the stationary row held in DH / 2 registers), then the accumulator registers themselves are the B operand of the
// products that contract over the streamed rows (register r of the two lane halves = one k step, as in the forward's P V).
// The partial sums of the four waves are added through LDS at the end.
// ---------------------------------------------------------------------------------------------------------------------
#define XA_BWD_NW(MODE_) ((MODE_) == 0 ? XA_WAVES : 1)
template <int DH, bool MASK, int MODE>
__global__ __launch_bounds__(64 * XA_BWD_NW(MODE), 2) void xattn_f32_bwd_kernel(const float* __restrict__ Q, const float* __restrict__ Kp, const float* __restrict__ Vp,
                                                                      const float* __restrict__ O, const float* __restrict__ dO,
                                                                      const float* __restrict__ lse, float* __restrict__ Dq, float* __restrict__ dQ,
                                                                      float* __restrict__ dK, float* __restrict__ dV,
                                                                      const unsigned char* __restrict__ mask, int Nq, int Nk, int ldq, int ldk, int ldv,
                                                                      int ldo, int lddq, int lddk, int lddv, long long q_bstride, const XaDrop dr) {
    // MW: floats per column in the final merge; MS = its slot pitch, ODD: with pitch DH / 2 DH the 32 columns of a wave all fell into ONE bank --
    // every merge access 32-way conflicted, 50 % (MODE 0) and 79 % (MODE 1) of the kernels' LDS cycles (profiles/r6_dasm_pmc_fp32attn.json) for a
    // step that runs once per workgroup beside only 3-4 tiles per wave
    constexpr int LDK = DH + 4, NDB = DH / 32, WS = 2 * XA_KT * LDK + 64, MW = MODE == 0 ? DH : 2 * DH, MS = MW + 1;
    // Waves per workgroup.  MODE 0 has 13 column blocks x 38 tiles per (clip, head) at the training shape: four waves split the tiles and merge.
    // MODE 1 has 38 column blocks x 13 tiles: ONE wave per workgroup walks them all -- no cross-wave merge, no workgroup barrier, and the
    // stationary rows are loaded and split once per column block instead of once per wave (432 -> see profiles/r6_dasm_xattn_split16.txt).
    constexpr int NWV = XA_BWD_NW(MODE);
    extern __shared__ __attribute__((aligned(16))) float xa_lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), h = blockIdx.y, b = blockIdx.z, Hn = gridDim.y;
    const int lq = lane & 31, lg = lane >> 5;
    float* T0 = xa_lds + wave * WS;           // MODE 0: K tile,  MODE 1: Q tile
    float* T1 = T0 + XA_KT * LDK;             // MODE 0: V tile,  MODE 1: dO tile
    float* rowc = T1 + XA_KT * LDK;           // MODE 1: [0 .. 31] lse of the tile's queries, [32 .. 63] their D
    const int Ncol = MODE == 0 ? Nq : Nk, Nrow = MODE == 0 ? Nk : Nq;
    const int c = blockIdx.x * 32 + lq;
    const int cc = c < Ncol ? c : Ncol - 1;
    const float sc = 1.4426950408889634f * rsqrtf((float)DH);
    const float* qb = Q + (size_t)b * q_bstride + h * DH;
    const float* kb = Kp + (size_t)b * Nk * ldk + h * DH;
    const float* vb = Vp + (size_t)b * Nk * ldv + h * DH;
    const float* ob = O + (size_t)b * Nq * ldo + h * DH;
    const float* dob = dO + (size_t)b * Nq * ldo + h * DH;
    const size_t statbase = ((size_t)b * Hn + h) * Nq;
    float lse_c = 0.f, D_c = 0.f;
    // stationary operands, lane half lg = elements 16 jb + 8 lg .. + 7 of the column's row: fa (score product, IEEE-half pair, pre-scaled like the
    // forward's) and fb (dO V^T product, bf16 pair)
    s16x8_t fah[DH / 16], fal[DH / 16], fbh[DH / 16], fbl[DH / 16];
    {
        const float *pa, *pb;
        if (MODE == 0) { pa = qb + (size_t)cc * ldq + 8 * lg; pb = dob + (size_t)cc * ldo + 8 * lg; }
        else { pa = kb + (size_t)cc * ldk + 8 * lg; pb = vb + (size_t)cc * ldv + 8 * lg; }
        const float* op = ob + (size_t)cc * ldo + 8 * lg;
        float part = 0.f;
#pragma unroll
        for (int jb = 0; jb < DH / 16; ++jb) {
            float xa[8], xb[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                xa[i] = pa[16 * jb + i] * sc;
                xb[i] = pb[16 * jb + i];
                if (MODE == 0) part = fmaf(xb[i], op[16 * jb + i], part);
            }
            xa_split8<true>(xa, fah[jb], fal[jb]);
            xa_split8<false>(xb, fbh[jb], fbl[jb]);
        }
        if (MODE == 0) {
            D_c = part + __shfl_xor(part, 32, 64);
            lse_c = lse[statbase + cc];
            if (wave == 0 && lg == 0 && c < Nq && Dq != nullptr) Dq[statbase + c] = D_c;
        }
    }
    f32x16 acc0[NDB], acc1[NDB];      // MODE 0: acc0 = dQ^T;  MODE 1: acc0 = dK^T, acc1 = dV^T   ([d, column])
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[db][r] = 0.f; acc1[db][r] = 0.f; }
    const int trow = lane & 31, thalf = (lane >> 5) * (DH / 2);      // (see the forward's tile loader)
    const unsigned long long bh = (unsigned long long)b * Hn + h;
    // The wave's next tile travels in registers while the current one is computed, as in the forward -- in the query-stationary pass only: the
    // key-stationary one (two output accumulator sets, 256 registers) has no 64 registers to spare; with the prefetch it spilled 57 of them
    // through scratch memory, whose loads queue behind the prefetch in the same counter (630 -> 838 us; MODE 0: 343 -> 303, forward 287 -> 258).
    constexpr bool PF = MODE == 0;
    XaTileRegs tr;
    float stat = 0.f;
#define XA_BWD_GLOAD(R0_)                                                                                                                  \
    {                                                                                                                                      \
        const int rr_ = ((R0_) + trow) < Nrow ? ((R0_) + trow) : Nrow - 1;                                                                  \
        const float *p0_, *p1_;                                                                                                            \
        if (MODE == 0) {                                                                                                                   \
            p0_ = kb + (size_t)rr_ * ldk + thalf;                                                          \
            p1_ = vb + (size_t)rr_ * ldv + thalf;                                                          \
        } else {                                                                                                                           \
            p0_ = qb + (size_t)rr_ * ldq + thalf;                                                          \
            p1_ = dob + (size_t)rr_ * ldo + thalf;                                                         \
        }                                                                                                                                  \
        xa_regs_gload<DH / 8>(tr, p0_, p1_);                                                                                                \
        if (MODE == 1) {                                                                                                                   \
            const int ri_ = ((R0_) + lq) < Nq ? ((R0_) + lq) : Nq - 1;                                                                      \
            stat = lg == 0 ? lse[statbase + ri_] : Dq[statbase + ri_];                                                                      \
        }                                                                                                                                  \
    }
    if (PF && wave * XA_KT < Nrow) XA_BWD_GLOAD(wave * XA_KT)
    for (int r0 = wave * XA_KT; r0 < Nrow; r0 += XA_KT * NWV) {
        {
            if (!PF) XA_BWD_GLOAD(r0)
            __builtin_amdgcn_wave_barrier();
            xa_regs_lstore<DH / 8>(tr, T0 + trow * LDK + thalf, T1 + trow * LDK + thalf);
            if (MODE == 1) rowc[lane] = stat;
            __builtin_amdgcn_wave_barrier();
            if (PF && r0 + XA_KT * NWV < Nrow) XA_BWD_GLOAD(r0 + XA_KT * NWV)
        }
#undef XA_BWD_GLOAD
        f32x16 st, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { st[r] = 0.f; dp[r] = 0.f; }
        {
#pragma unroll
            for (int jb = 0; jb < DH / 16; ++jb) {      // S exactly as the forward computes it (one chain, k blocks in order); dP on bf16 pairs
                float ax[8], bx[8];
                xa_row8(T0 + lq * LDK + 16 * jb + 8 * lg, ax);
                xa_row8(T1 + lq * LDK + 16 * jb + 8 * lg, bx);
                s16x8_t ah, al, bh_, bl_;
                xa_split8<true>(ax, ah, al);
                xa_split8<false>(bx, bh_, bl_);
                st = xa_mfma3<true>(ah, al, fah[jb], fal[jb], st);
                dp = xa_mfma3<false>(bh_, bl_, fbh[jb], fbl[jb], dp);
            }
        }
        // dropout bits of the tile's 16 elements of this lane, as a mask (bit r = keep).  Four consecutive keys share one hash (drop_hash4):
        // in MODE 0 they are four consecutive registers of the lane; in MODE 1 (a lane = one key, registers = queries) the four lanes of a
        // quad hold the four keys of a group -- each lane hashes the rows r = 4 i + (lane & 3) and the quad exchanges them (DPP quad_perm)
        unsigned keepm = 0xffffu;
        if (dr.thr != 0u) {
            keepm = 0u;
            if ((Nk & 3) == 0) {
                if (MODE == 0) {
#pragma unroll
                    for (int rq = 0; rq < 4; ++rq) {
                        const int jq = r0 + 8 * rq + 4 * lg;
                        const unsigned long long z = drop_hash4(dr.seed, dr.sid, ((bh * Nq + cc) * (unsigned long long)Nk + (jq < Nk ? jq : 0)) >> 2);
#pragma unroll
                        for (int i = 0; i < 4; ++i) keepm |= (drop_keep_of(z, i, dr.thr) ? 1u : 0u) << (4 * rq + i);
                    }
                } else {
                    unsigned zlo[4], zhi[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int rr = r0 + mfma32_row(4 * i + (lane & 3), lg);
                        const unsigned long long z = drop_hash4(dr.seed, dr.sid, ((bh * Nq + (rr < Nq ? rr : Nq - 1)) * (unsigned long long)Nk + (cc & ~3)) >> 2);
                        zlo[i] = (unsigned)z; zhi[i] = (unsigned)(z >> 32);
                    }
                    const unsigned l4 = (unsigned)cc & 3u;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
#define XA_QB(K_) (((unsigned long long)(unsigned)__builtin_amdgcn_mov_dpp((int)zhi[i], (K_) * 0x55, 0xf, 0xf, true) << 32) | \
                   (unsigned)__builtin_amdgcn_mov_dpp((int)zlo[i], (K_) * 0x55, 0xf, 0xf, true))
                        keepm |= (drop_keep_of(XA_QB(0), l4, dr.thr) ? 1u : 0u) << (4 * i + 0);
                        keepm |= (drop_keep_of(XA_QB(1), l4, dr.thr) ? 1u : 0u) << (4 * i + 1);
                        keepm |= (drop_keep_of(XA_QB(2), l4, dr.thr) ? 1u : 0u) << (4 * i + 2);
                        keepm |= (drop_keep_of(XA_QB(3), l4, dr.thr) ? 1u : 0u) << (4 * i + 3);
#undef XA_QB
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ri = r0 + mfma32_row(r, lg);
                    const int qi = MODE == 0 ? cc : (ri < Nq ? ri : Nq - 1), kj = MODE == 0 ? (ri < Nk ? ri : Nk - 1) : cc;
                    keepm |= (drop_keep(dr.seed, dr.sid, (bh * Nq + qi) * (unsigned long long)Nk + kj, dr.thr) ? 1u : 0u) << r;
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rl = mfma32_row(r, lg), ri = r0 + rl;
            const int qi = MODE == 0 ? cc : (ri < Nq ? ri : Nq - 1), kj = MODE == 0 ? (ri < Nk ? ri : Nk - 1) : cc;
            bool dead = ri >= Nrow;
            if (MASK) dead = dead || mask[(size_t)qi * Nk + kj] != 0;
            const float l2 = MODE == 0 ? lse_c : rowc[rl];
            const float Dr = MODE == 0 ? D_c : rowc[32 + rl];
            const float pr = dead ? 0.f : exp2f(st[r] - l2);
            float dpv = dp[r], pd = pr;
            if (dr.thr != 0u) {
                const bool keep = (keepm >> r) & 1u;
                dpv = keep ? dpv * dr.scale : 0.f;
                pd = keep ? pr * dr.scale : 0.f;
            }
            st[r] = pr * (dpv - Dr);      // dS (before the 1 / sqrt(dh) of the score scale)
            dp[r] = pd;                   // dropped probabilities: B operand of dV
        }
        // products that contract over the streamed rows, bf16 pairs: k block m = the rows of the lane's accumulator registers 8 m .. 8 m + 7
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            float sx[8], px[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) { sx[i] = st[8 * m + i]; px[i] = dp[8 * m + i]; }
            s16x8_t sh, sl, ph, pl;
            xa_split8<false>(sx, sh, sl);
            if (MODE == 1) xa_split8<false>(px, ph, pl);
#pragma unroll
            for (int db = 0; db < NDB; ++db) {
                float ax[8];
                xa_col8<LDK>(T0, m, lg, lq + 32 * db, ax);
                s16x8_t ah, al;
                xa_split8<false>(ax, ah, al);
                acc0[db] = xa_mfma3<false>(ah, al, sh, sl, acc0[db]);     // dQ += dS K   /  dK += dS^T Q
                if (MODE == 1) {
                    xa_col8<LDK>(T1, m, lg, lq + 32 * db, ax);
                    xa_split8<false>(ax, ah, al);
                    acc1[db] = xa_mfma3<false>(ah, al, ph, pl, acc1[db]);      // dV += Pd^T dO
                }
            }
        }
    }
    // ---- add the four waves' partial sums: slot [wave][column][MS]; lane (column, g) owns d = 32 db + mfma32_row(r, g)
    __syncthreads();
    {
        float* mg = xa_lds + (size_t)(wave * 32 + lq) * MS;
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                mg[32 * db + mfma32_row(r, lg)] = acc0[db][r];
                if (MODE == 1) mg[DH + 32 * db + mfma32_row(r, lg)] = acc1[db][r];
            }
    }
    __syncthreads();
    {
        constexpr int CW = MW / (2 * NWV);
        const int d0 = (wave * 2 + lg) * CW;
        float out[CW];
#pragma unroll
        for (int d = 0; d < CW; ++d) out[d] = 0.f;
#pragma unroll
        for (int w = 0; w < NWV; ++w) {
            const float* pw = xa_lds + (size_t)(w * 32 + lq) * MS + d0;
#pragma unroll
            for (int d = 0; d < CW; ++d) out[d] += pw[d];
        }
        if (c < Ncol) {
            const float rs = rsqrtf((float)DH);
            float* dst;
            float f;
            if (MODE == 0) { dst = dQ + ((size_t)b * Nq + c) * lddq + h * DH + d0; f = rs; }
            else if (d0 < DH) { dst = dK + ((size_t)b * Nk + c) * lddk + h * DH + d0; f = rs; }
            else { dst = dV + ((size_t)b * Nk + c) * lddv + h * DH + (d0 - DH); f = 1.f; }
#pragma unroll
            for (int d = 0; d < CW; d += 4) *reinterpret_cast<float4*>(dst + d) = make_float4(out[d] * f, out[d + 1] * f, out[d + 2] * f, out[d + 3] * f);
        }
    }
}

extern "C" int sed_xattn_f32_bwd(const float* Q, const float* K, const float* V, const float* O, const float* dO, const float* lse, float* Dq,
                                 float* dQ, float* dK, float* dV, const uint8_t* mask, int B, int H, int Nq, int Nk, int head_dim, int ldq, int ldk,
                                 int ldv, int ldo, int lddq, int lddk, int lddv, int64_t q_batch_stride, float drop_p, int64_t drop_seed,
                                 int drop_site, hipStream_t stream) {
    (void)hipGetLastError();
    if (B <= 0 || H <= 0 || Nq <= 0 || Nk <= 0 || (head_dim != 32 && head_dim != 64) || ((ldq | ldk | ldv | ldo | lddq | lddk | lddv) & 3) || B > 65535 ||
        H > 65535 || !(drop_p >= 0.f && drop_p < 1.f) || lse == nullptr || Dq == nullptr)
        return SED_ERR_ARG;
    if (((uintptr_t)Q | (uintptr_t)K | (uintptr_t)V | (uintptr_t)O | (uintptr_t)dO | (uintptr_t)dQ | (uintptr_t)dK | (uintptr_t)dV) & 15) return SED_ERR_ARG;
    XaDrop dr;
    dr.thr = drop_thr24(drop_p); dr.sid = (unsigned)drop_site; dr.scale = 1.0f / (1.0f - drop_p); dr.seed = (unsigned long long)drop_seed;
#define XBWD_LAUNCH(DH_, MK_, MODE_)                                                                                             \
    {                                                                                                                            \
        const int nw_ = XA_BWD_NW(MODE_);                                                                                        \
        const int tile_ = nw_ * (2 * XA_KT * (DH_ + 4) + 64) * 4, merge_ = nw_ * 32 * (((MODE_) == 0 ? DH_ : 2 * DH_) + 1) * 4;   \
        const int lds_ = tile_ > merge_ ? tile_ : merge_;                                                                        \
        static bool attr_ = false;                                                                                               \
        if (!attr_) { (void)hipFuncSetAttribute((const void*)xattn_f32_bwd_kernel<DH_, MK_, MODE_>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_); attr_ = true; } \
        hipLaunchKernelGGL((xattn_f32_bwd_kernel<DH_, MK_, MODE_>), dim3(cdiv((MODE_) == 0 ? Nq : Nk, 32), H, B), dim3(64 * nw_), lds_, stream, \
                           Q, K, V, O, dO, lse, Dq, dQ, dK, dV, mask, Nq, Nk, ldq, ldk, ldv, ldo, lddq, lddk, lddv, (long long)q_batch_stride, dr); \
    }
#define XBWD_PICK(MODE_)                                                                                                         \
    if (head_dim == 64) { if (mask != nullptr) XBWD_LAUNCH(64, true, MODE_) else XBWD_LAUNCH(64, false, MODE_) }                 \
    else { if (mask != nullptr) XBWD_LAUNCH(32, true, MODE_) else XBWD_LAUNCH(32, false, MODE_) }
    XBWD_PICK(0)
    XBWD_PICK(1)
#undef XBWD_PICK
#undef XBWD_LAUNCH
    return sed_check_launch();
}

// ---------------------------------------------------------------------------------------------------------------------
// dual-stream finish (detect_any_sound.py:394-404): logits [B, T, Q] -> strong [B, Q, T] = clamp(pad ? 0 : sigmoid(logit / temp) * at[b, q],
// 1e-7, 1) through a 32 x 32 LDS transpose (coalesced on both sides), at[b, q] = sigmoid(at_logit[b, q]);
// weak [B, Q] = clamp(sum_t s^2 / sum_t s, 1e-7, 1).
// at_logit == nullptr, clamp_strong == 0: the closed-set classifier head for any class count (passt_cnn.py:74-86, passt_sed.py:285-296:
// strong = sigmoid(logit / temp), pad mask, NO clamp on the frame posteriors, same pooling) -- the 407 AudioSet-Strong classes of
// recipes/audioset_strong/base/passt_cnn; the 10-class DESED head keeps its dedicated kernels (norm_elem.hip).
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dasm_head_kernel(const float* __restrict__ logits, const float* __restrict__ at_logit,
                                                        const unsigned char* __restrict__ pad, float inv_temp, float* __restrict__ strong,
                                                        float* __restrict__ at_out, int T, int Qn, int clamp_strong) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, t0 = blockIdx.y * 32, q0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int t = t0 + ty + 8 * i, qn = q0 + tx;
        float v = 0.f;
        if (t < T && qn < Qn) {
            const float a = at_logit != nullptr ? sigmoidf_(at_logit[(size_t)b * Qn + qn]) : 1.0f;      // (no tagging stream: the closed-set head)
            const bool masked = pad != nullptr && pad[(size_t)b * T + t] != 0;
            v = masked ? 0.f : sigmoidf_(logits[((size_t)b * T + t) * Qn + qn] * inv_temp) * a;
            if (clamp_strong) v = fminf(fmaxf(v, 1e-7f), 1.0f);
            if (t == 0 && at_out != nullptr) at_out[(size_t)b * Qn + qn] = a;
        }
        tile[ty + 8 * i][tx] = v;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int qn = q0 + ty + 8 * i, t = t0 + tx;
        if (t < T && qn < Qn) strong[((size_t)b * Qn + qn) * T + t] = tile[tx][ty + 8 * i];
    }
}
__global__ __launch_bounds__(256) void dasm_weak_kernel(const float* __restrict__ strong, float* __restrict__ weak, int T) {
    __shared__ float ra[4], rb[4];
    const float* s = strong + (size_t)blockIdx.x * T;
    float a = 0.f, bs = 0.f;
    for (int t = threadIdx.x; t < T; t += 256) { const float v = s[t]; a += v * v; bs += v; }
    a = wave_sum(a); bs = wave_sum(bs);
    if ((threadIdx.x & 63) == 0) { ra[threadIdx.x >> 6] = a; rb[threadIdx.x >> 6] = bs; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float w = (ra[0] + ra[1] + ra[2] + ra[3]) / (rb[0] + rb[1] + rb[2] + rb[3]);
        weak[blockIdx.x] = (w != w) ? w : fminf(fmaxf(w, 1e-7f), 1.0f);
    }
}
extern "C" int sed_dasm_head_fwd(const float* logits, const float* at_logit, const uint8_t* pad_mask, float temp, float* strong, float* weak,
                                 float* at_out, int B, int T, int Q, int clamp_strong, hipStream_t stream) {
    (void)hipGetLastError();
    if (B <= 0 || T <= 0 || Q <= 0 || B > 65535 || !(temp > 0.f)) return SED_ERR_ARG;
    hipLaunchKernelGGL(dasm_head_kernel, dim3(cdiv(Q, 32), cdiv(T, 32), B), dim3(256), 0, stream, logits, at_logit, pad_mask, 1.0f / temp, strong,
                       at_out, T, Q, clamp_strong);
    hipLaunchKernelGGL(dasm_weak_kernel, dim3(B * Q), dim3(256), 0, stream, (const float*)strong, weak, T);
    return sed_check_launch();
}

// ---------------------------------------------------------------------------------------------------------------------
// backward of the dual-stream finish.  Given d strong [B, Q, T], d weak [B, Q], d at_out [B, Q] (each nullable):
//   weak = clamp(S2 / S1, 1e-7, 1), S1 = sum_t s, S2 = sum_t s^2      ->  ds_t += dweak (2 s_t S1 - S2) / S1^2  inside the clamp
//   s = clamp(pad ? 0 : sigma(l / temp) a, 1e-7, 1), a = sigmoid(at_logit)   ->  dl = ds a sigma (1 - sigma) / temp,  da = sum_t ds sigma
//   (torch.clamp passes the gradient where min <= x <= max; `sed_out[pad_mask] = 0` cuts it on padded frames, detect_any_sound.py:382-388)
//   d at_logit = (da + d at_out) a (1 - a)
// at_logit == nullptr: the closed-set head (strong = sigmoid(logit / temp), passt_cnn.py:74-86 with any class count): a = 1.
// scratch: 3 B Q floats.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dasm_head_bwd_prep_kernel(const float* __restrict__ strong, const float* __restrict__ dweak,
                                                                 float* __restrict__ scratch, int T) {
    __shared__ float ra[4], rb[4];
    const float* s = strong + (size_t)blockIdx.x * T;
    float a = 0.f, bs = 0.f;
    for (int t = threadIdx.x; t < T; t += 256) { const float v = s[t]; a += v * v; bs += v; }
    a = wave_sum(a); bs = wave_sum(bs);
    if ((threadIdx.x & 63) == 0) { ra[threadIdx.x >> 6] = a; rb[threadIdx.x >> 6] = bs; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float S2 = (ra[0] + ra[1]) + (ra[2] + ra[3]), S1 = (rb[0] + rb[1]) + (rb[2] + rb[3]);
        const float w = S2 / S1;
        float c1 = 0.f, c0 = 0.f;
        if (dweak != nullptr && w >= 1e-7f && w <= 1.0f) {
            const float g = dweak[blockIdx.x];
            c1 = 2.f * g / S1;
            c0 = -g * S2 / (S1 * S1);
        }
        scratch[3 * (size_t)blockIdx.x] = c1;
        scratch[3 * (size_t)blockIdx.x + 1] = c0;
        scratch[3 * (size_t)blockIdx.x + 2] = 0.f;
    }
}
__global__ __launch_bounds__(256) void dasm_head_bwd_kernel(const float* __restrict__ logits, const float* __restrict__ at_logit,
                                                            const unsigned char* __restrict__ pad, float inv_temp, const float* __restrict__ strong,
                                                            const float* __restrict__ dstrong, float* __restrict__ scratch,
                                                            float* __restrict__ dlogits, int T, int Qn, int clamp_strong) {
    __shared__ float ts[32][33], tg[32][33], red[8][32];
    const int b = blockIdx.z, t0 = blockIdx.y * 32, q0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
    // strong / d strong tiles, read along t
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int qn = q0 + ty + 8 * i, t = t0 + tx;
        float sv = 0.f, gv = 0.f;
        if (t < T && qn < Qn) {
            const size_t o = ((size_t)b * Qn + qn) * T + t;
            sv = strong[o];
            const float* cf = scratch + 3 * ((size_t)b * Qn + qn);
            gv = (dstrong != nullptr ? dstrong[o] : 0.f) + cf[0] * sv + cf[1];
        }
        ts[ty + 8 * i][tx] = sv;
        tg[ty + 8 * i][tx] = gv;
    }
    __syncthreads();
    float da = 0.f;
    const int qn = q0 + tx;
    const float a = (qn < Qn && at_logit != nullptr) ? sigmoidf_(at_logit[(size_t)b * Qn + qn]) : 1.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int t = t0 + ty + 8 * i;
        if (t < T && qn < Qn) {
            const size_t o = ((size_t)b * T + t) * Qn + qn;
            const bool masked = pad != nullptr && pad[(size_t)b * T + t] != 0;
            const float sg = sigmoidf_(logits[o] * inv_temp);
            const float x = sg * a;                       // value before the clamp
            const float g = (!masked && (!clamp_strong || (x >= 1e-7f && x <= 1.0f))) ? tg[tx][ty + 8 * i] : 0.f;
            dlogits[o] = g * a * sg * (1.f - sg) * inv_temp;
            da += g * sg;
        }
    }
    if (at_logit != nullptr) {
        red[ty][tx] = da;
        __syncthreads();
        if (ty == 0 && qn < Qn) {
            float v = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) v += red[k][tx];
            unsafeAtomicAdd(&scratch[3 * ((size_t)b * Qn + qn) + 2], v);
        }
    }
}
__global__ __launch_bounds__(256) void dasm_head_bwd_at_kernel(const float* __restrict__ at_logit, const float* __restrict__ dat_out,
                                                               const float* __restrict__ scratch, float* __restrict__ dat_logit, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float a = sigmoidf_(at_logit[i]);
    dat_logit[i] = (scratch[3 * (size_t)i + 2] + (dat_out != nullptr ? dat_out[i] : 0.f)) * a * (1.f - a);
}
extern "C" int sed_dasm_head_bwd(const float* logits, const float* at_logit, const uint8_t* pad_mask, float temp, const float* strong,
                                 const float* dstrong, const float* dweak, const float* dat_out, float* dlogits, float* dat_logit, float* scratch,
                                 int B, int T, int Q, int clamp_strong, hipStream_t stream) {
    (void)hipGetLastError();
    if (B <= 0 || T <= 0 || Q <= 0 || B > 65535 || !(temp > 0.f) || scratch == nullptr || dlogits == nullptr) return SED_ERR_ARG;
    if ((at_logit == nullptr) != (dat_logit == nullptr)) return SED_ERR_ARG;
    hipLaunchKernelGGL(dasm_head_bwd_prep_kernel, dim3(B * Q), dim3(256), 0, stream, strong, dweak, scratch, T);
    hipLaunchKernelGGL(dasm_head_bwd_kernel, dim3(cdiv(Q, 32), cdiv(T, 32), B), dim3(256), 0, stream, logits, at_logit, pad_mask, 1.0f / temp, strong,
                       dstrong, scratch, dlogits, T, Q, clamp_strong);
    if (at_logit != nullptr)
        hipLaunchKernelGGL(dasm_head_bwd_at_kernel, dim3(cdiv(B * Q, 256)), dim3(256), 0, stream, at_logit, dat_out, (const float*)scratch, dat_logit,
                           B * Q);
    return sed_check_launch();
}

// ---------------------------------------------------------------------------------------------------------------------
// elementwise helpers of the query decoder's backward
// ---------------------------------------------------------------------------------------------------------------------
// out = dy o keep / (1 - p) o gelu'(pre)   (h = dropout(gelu(pre)): the FFN of nn.TransformerDecoderLayer, the MLPs of the heads with p = 0)
__global__ __launch_bounds__(256) void gelu_bwd_f32_kernel(const float* __restrict__ dy, const float* __restrict__ pre, float* __restrict__ out,
                                                           long long n, unsigned thr, unsigned sid, float scale, unsigned long long seed) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        float g = dy[i];
        if (thr != 0u) g = drop_keep(seed, sid, (unsigned long long)i, thr) ? g * scale : 0.f;
        out[i] = g * gelu_fast_grad(pre[i]);
    }
}
extern "C" int sed_gelu_bwd_f32(const float* dy, const float* pre, float* out, int64_t n, float drop_p, int64_t drop_seed, int drop_site,
                                hipStream_t stream) {
    (void)hipGetLastError();
    if (n <= 0 || !(drop_p >= 0.f && drop_p < 1.f)) return SED_ERR_ARG;
    const int blocks = (int)((n + 255) / 256 > 8192 ? 8192 : (n + 255) / 256);
    hipLaunchKernelGGL(gelu_bwd_f32_kernel, dim3(blocks), dim3(256), 0, stream, dy, pre, out, (long long)n, drop_thr24(drop_p), (unsigned)drop_site,
                       1.0f / (1.0f - drop_p), (unsigned long long)drop_seed);
    return sed_check_launch();
}
// out = x o keep / (1 - p): the gradient through a dropout site whose forward ran in a GEMM epilogue; mask_u8 (nullable): the bits themselves
__global__ __launch_bounds__(256) void dropout_f32_kernel(const float* __restrict__ x, float* __restrict__ out, unsigned char* __restrict__ mask_u8,
                                                          long long n, unsigned thr, unsigned sid, float scale, unsigned long long seed) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const bool keep = thr == 0u || drop_keep(seed, sid, (unsigned long long)i, thr);
        if (out != nullptr) out[i] = keep ? x[i] * scale : 0.f;
        if (mask_u8 != nullptr) mask_u8[i] = keep ? 1 : 0;
    }
}
extern "C" int sed_dropout_f32(const float* x, float* out, uint8_t* mask_u8, int64_t n, float drop_p, int64_t drop_seed, int drop_site,
                               hipStream_t stream) {
    (void)hipGetLastError();
    if (n <= 0 || !(drop_p >= 0.f && drop_p < 1.f) || (out != nullptr && x == nullptr)) return SED_ERR_ARG;
    const int blocks = (int)((n + 255) / 256 > 8192 ? 8192 : (n + 255) / 256);
    hipLaunchKernelGGL(dropout_f32_kernel, dim3(blocks), dim3(256), 0, stream, x, out, mask_u8, (long long)n, drop_thr24(drop_p), (unsigned)drop_site,
                       1.0f / (1.0f - drop_p), (unsigned long long)drop_seed);
    return sed_check_launch();
}
// out = drop(act(x)) (+ res): the elementwise tail of a Linear whose product ran on the 16-bit matrix pipe (dasm.py, large query counts):
// act 1 = GELU; dropout bits of element index i as in sed_gemm_f32's epilogue
__global__ __launch_bounds__(256) void act_drop_res_f32_kernel(const float* __restrict__ x, const float* __restrict__ res, float* __restrict__ out,
                                                               long long n, int act, unsigned thr, unsigned sid, float scale, unsigned long long seed) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        float v = x[i];
        if (act == 1) v = gelu_fast(v);
        if (thr != 0u) v = drop_keep(seed, sid, (unsigned long long)i, thr) ? v * scale : 0.f;
        if (res != nullptr) v += res[i];
        out[i] = v;
    }
}
extern "C" int sed_act_drop_res_f32(const float* x, const float* res, float* out, int64_t n, int act, float drop_p, int64_t drop_seed, int drop_site,
                                    hipStream_t stream) {
    (void)hipGetLastError();
    if (n <= 0 || act < 0 || act > 1 || !(drop_p >= 0.f && drop_p < 1.f)) return SED_ERR_ARG;
    const int blocks = (int)((n + 255) / 256 > 8192 ? 8192 : (n + 255) / 256);
    hipLaunchKernelGGL(act_drop_res_f32_kernel, dim3(blocks), dim3(256), 0, stream, x, res, out, (long long)n, act, drop_thr24(drop_p), (unsigned)drop_site,
                       1.0f / (1.0f - drop_p), (unsigned long long)drop_seed);
    return sed_check_launch();
}
// out[c] += sum_r x[r ld + c]: bias gradients (column sums over the tokens), the gradient of the shared queries (sum over the clips)
__global__ __launch_bounds__(256) void colsum_f32_kernel(const float* __restrict__ x, float* __restrict__ out, int rows, int cols, long long ld,
                                                         int rows_per_block) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    float a = 0.f;
    for (int r = r0; r < r1; ++r) a += x[(size_t)r * ld + c];
    unsafeAtomicAdd(&out[c], a);
}
extern "C" int sed_colsum_f32(const float* x, float* out, int rows, int cols, int64_t ld, hipStream_t stream) {
    (void)hipGetLastError();
    if (rows <= 0 || cols <= 0 || ld < cols) return SED_ERR_ARG;
    int rpb = 64;
    while (cdiv(rows, rpb) > 65535) rpb *= 2;
    hipLaunchKernelGGL(colsum_f32_kernel, dim3(cdiv(cols, 256), cdiv(rows, rpb)), dim3(256), 0, stream, x, out, rows, cols, (long long)ld, rpb);
    return sed_check_launch();
}

// ---------------------------------------------------------------------------------------------------------------------
// supervised losses of the AudioSet-Strong trainers (src/functional/loss/__init__.py:18-68 through loss_function_factory), mean over all
// elements, with the gradient with respect to the prediction in the same pass:
//   kind 0  BCELoss / AsymmetricalFocalLoss(gamma, zeta) / AslLoss(rp, rn, margin):
//           -[(1 - p)^gp t max(log p, -100) + pm^gn (1 - t) max(log(1 - pm), -100)],  pm = max(p - margin, 0)
//           (gp = gn = margin = 0 is torch.nn.BCELoss, whose backward divides by max(p (1 - p), 1e-12))
//   kind 1  MSELoss
// loss[0] += sum / n (zeroed by the caller); grad = d loss / d p (nullable).
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sup_loss_kernel(const float* __restrict__ pr, const float* __restrict__ tg, float* __restrict__ loss,
                                                       float* __restrict__ grad, long long n, int kind, float gp, float gn, float margin) {
    __shared__ float red[4];
    const float inv_n = 1.0f / (float)n;
    float acc = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float p = pr[i], t = tg[i];
        float l, g;
        if (kind == 1) {
            const float d = p - t;
            l = d * d;
            g = 2.f * d;
        } else if (gp == 0.f && gn == 0.f && margin == 0.f) {
            l = -(t * fmaxf(logf(p), -100.f) + (1.f - t) * fmaxf(log1pf(-p), -100.f));
            g = (p - t) / fmaxf(p * (1.f - p), 1e-12f);
        } else {
            const float pm = fmaxf(p - margin, 0.f);
            const float lp = logf(p), lq = log1pf(-pm);
            const float L1 = fmaxf(lp, -100.f), L2 = fmaxf(lq, -100.f);
            const float w1 = gp == 0.f ? 1.f : powf(1.f - p, gp), w2 = gn == 0.f ? 1.f : powf(pm, gn);
            l = -(w1 * t * L1 + w2 * (1.f - t) * L2);
            // d/dp of the first term: -gp (1 - p)^(gp - 1) t L1 + (1 - p)^gp t / p (inside the log clamp)
            const float d1 = (gp == 0.f ? 0.f : -gp * powf(1.f - p, gp - 1.f) * t * L1) + (lp > -100.f ? w1 * t / p : 0.f);
            // second term through pm (d pm / d p = 1 where p > margin): gn pm^(gn - 1) (1 - t) L2 - pm^gn (1 - t) / (1 - pm)
            float d2 = 0.f;
            if (p - margin > 0.f) d2 = (gn == 0.f ? 0.f : gn * powf(pm, gn - 1.f) * (1.f - t) * L2) - (lq > -100.f ? w2 * (1.f - t) / (1.f - pm) : 0.f);
            g = -(d1 + d2);
        }
        acc += l;
        if (grad != nullptr) grad[i] = g * inv_n;
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) unsafeAtomicAdd(loss, ((red[0] + red[1]) + (red[2] + red[3])) * inv_n);
}
extern "C" int sed_sup_loss(const float* pred, const float* target, float* loss, float* grad, int64_t n, int kind, float gamma_pos, float gamma_neg,
                            float margin, hipStream_t stream) {
    (void)hipGetLastError();
    if (n <= 0 || kind < 0 || kind > 1 || loss == nullptr) return SED_ERR_ARG;
    const int blocks = (int)((n + 1023) / 1024 > 1024 ? 1024 : (n + 1023) / 1024);
    hipLaunchKernelGGL(sup_loss_kernel, dim3(blocks), dim3(256), 0, stream, pred, target, loss, grad, (long long)n, kind, gamma_pos, gamma_neg, margin);
    return sed_check_launch();
}
