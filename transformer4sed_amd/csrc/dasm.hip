// DASM (open-vocabulary sound event detection, BASELINE.json config #5) -- the query decoder and the dual-stream head, forward, gfx950.
//
// Replaces src/models/detect_any_sound/at_adapter.py:7-50 (nn.TransformerDecoder of cross-attention-first layers over the backbone's
// patch tokens) and src/models/detect_any_sound/detect_any_sound.py:283-322, 362-389 (query projector, at_head, sed_head,
// mask_embedding MLP, einsum('bqc,bct->bqt'), sigmoid(x / temp) * at_out, pad mask, clamp, linear-softmax pooling).
//
// Everything on the query side is small (Q = 10 .. a few hundred queries per clip against 1188 patch tokens / 1000 frames) and sits
// directly in front of a sigmoid with temperature 0.1 .. 0.5, i.e. it is precision-critical, not throughput-critical: the kernels here
// compute in fp32 throughout --
//   * sed_gemm_f32_nt     C = act(A . B^T + bias) (+ residual), batched, on the fp32-input matrix instruction v_mfma_f32_32x32x2_f32
//                         (exact fp32 products and accumulation at the fp32 vector rate, without one VALU instruction per FMA and
//                         with a 64 x 64 output tile fed from LDS: MI355X_MICROARCH.md "FP32-input MFMA");
//   * sed_xattn_f32_fwd   softmax(q k^T / sqrt(dh) + mask) v for query counts that differ from the key count (cross attention over the
//                         patch tokens, self attention among the queries with the open-vocabulary mask), both products on the same
//                         fp32 matrix instruction, 32 queries per workgroup, four waves splitting the key tiles;
//   * sed_dasm_head_fwd   the dual-stream finish on the [B, T, Q] logits: sigmoid / temperature, times the clip-level tagging
//                         probability, pad mask, clamp, transposed store [B, Q, T], linear-softmax pooling.
#include "common.h"
#include "../../include/sed_hip.h"

// ---------------------------------------------------------------------------------------------------------------------
// fp32 GEMM, NT form: C[z][m][n] = act(sum_k A[z][m][k] B[z][n][k] + bias[n]) (+ R[z][m][n])
// 256 threads = 4 waves, 64 x 64 tile, wave (wm, wn) owns a 32 x 32 block = ONE accumulator tile of v_mfma_f32_32x32x2_f32;
// K tiles of 32 staged k-major in LDS ([k][m], row pitch 65 floats: the 4-scalar transposing writes of a float4 and the 32-lane
// fragment reads are both bank-conflict free), next tile's global loads in flight during the 16 MFMAs of the current one.
// ---------------------------------------------------------------------------------------------------------------------
#define F32_BK 32
#define F32_LD 65
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int ACT>
__global__ __launch_bounds__(256) void gemm_f32_nt_kernel(const float* __restrict__ A, const float* __restrict__ B, const float* __restrict__ bias,
                                                          const float* __restrict__ R, float* __restrict__ C, int M, int N, int K, int lda,
                                                          int ldb, int ldc, long long sA, long long sB, long long sC) {
    __shared__ float As[F32_BK * F32_LD], Bs[F32_BK * F32_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    A += (size_t)blockIdx.z * sA;
    B += (size_t)blockIdx.z * sB;
    C += (size_t)blockIdx.z * sC;
    if (R != nullptr) R += (size_t)blockIdx.z * sC;
    // loader: thread -> (row = tid / 8 (+ 32), k quad = tid % 8) of both operand tiles
    const int lrow = tid >> 3, lk = (tid & 7) * 4;
    float4 ra[2], rb[2];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = m0 + lrow + 32 * i, n = n0 + lrow + 32 * i;
            ra[i] = m < M ? *reinterpret_cast<const float4*>(A + (size_t)m * lda + k0 + lk) : make_float4(0.f, 0.f, 0.f, 0.f);
            rb[i] = n < N ? *reinterpret_cast<const float4*>(B + (size_t)n * ldb + k0 + lk) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = lrow + 32 * i;
            As[(lk + 0) * F32_LD + r] = ra[i].x; As[(lk + 1) * F32_LD + r] = ra[i].y; As[(lk + 2) * F32_LD + r] = ra[i].z; As[(lk + 3) * F32_LD + r] = ra[i].w;
            Bs[(lk + 0) * F32_LD + r] = rb[i].x; Bs[(lk + 1) * F32_LD + r] = rb[i].y; Bs[(lk + 2) * F32_LD + r] = rb[i].z; Bs[(lk + 3) * F32_LD + r] = rb[i].w;
        }
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int fa = (lane >> 5) * F32_LD + wm * 32 + (lane & 31), fb = (lane >> 5) * F32_LD + wn * 32 + (lane & 31);
    gload(0);
    for (int k0 = 0; k0 < K; k0 += F32_BK) {
        lstore();
        __syncthreads();
        if (k0 + F32_BK < K) gload(k0 + F32_BK);
#pragma unroll
        for (int ks = 0; ks < F32_BK; ks += 2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[ks * F32_LD + fa], Bs[ks * F32_LD + fb], acc, 0, 0, 0);
        __syncthreads();
    }
    const int n = n0 + wn * 32 + (lane & 31);
    if (n >= N) return;
    const float bn = bias != nullptr ? bias[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 32 + mfma32_row(r, lane >> 5);
        if (m >= M) continue;
        float v = acc[r] + bn;
        if (ACT == 1) v = gelu_fast(v);
        if (ACT == 2) v = fmaxf(v, 0.f);
        if (R != nullptr) v += R[(size_t)m * ldc + n];
        C[(size_t)m * ldc + n] = v;
    }
}

extern "C" int sed_gemm_f32_nt(const float* A, const float* B, const float* bias, const float* R, float* C, int M, int N, int K, int lda,
                               int ldb, int ldc, int batch, int64_t strideA, int64_t strideB, int64_t strideC, int act,
                               hipStream_t stream) {
    (void)hipGetLastError();
    if (M <= 0 || N <= 0 || K <= 0 || (K % F32_BK) != 0 || (lda & 3) || (ldb & 3) || batch < 1 || batch > 65535 || act < 0 || act > 2)
        return SED_ERR_ARG;
    if ((((uintptr_t)A | (uintptr_t)B) & 15) || ((strideA | strideB) & 3)) return SED_ERR_ARG;
    const dim3 grid(cdiv(N, 64), cdiv(M, 64), batch);
    if (act == 0)
        hipLaunchKernelGGL(gemm_f32_nt_kernel<0>, grid, dim3(256), 0, stream, A, B, bias, R, C, M, N, K, lda, ldb, ldc, (long long)strideA,
                           (long long)strideB, (long long)strideC);
    else if (act == 1)
        hipLaunchKernelGGL(gemm_f32_nt_kernel<1>, grid, dim3(256), 0, stream, A, B, bias, R, C, M, N, K, lda, ldb, ldc, (long long)strideA,
                           (long long)strideB, (long long)strideC);
    else
        hipLaunchKernelGGL(gemm_f32_nt_kernel<2>, grid, dim3(256), 0, stream, A, B, bias, R, C, M, N, K, lda, ldb, ldc, (long long)strideA,
                           (long long)strideB, (long long)strideC);
    return sed_check_launch();
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
template <int DH, bool MASK>
__global__ __launch_bounds__(64 * XA_WAVES) void xattn_f32_fwd_kernel(const float* __restrict__ Q, const float* __restrict__ Kp,
                                                                      const float* __restrict__ Vp, float* __restrict__ O,
                                                                      const unsigned char* __restrict__ mask, int Nq, int Nk, int ldq, int ldk,
                                                                      int ldv, int ldo, long long q_bstride) {
    constexpr int LDK = DH + 4, NDB = DH / 32, MS = DH + 2;
    extern __shared__ __attribute__((aligned(16))) float xa_lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), h = blockIdx.y, b = blockIdx.z;
    const int lq = lane & 31, lg = lane >> 5;
    float* Ks = xa_lds + wave * (2 * XA_KT * LDK);
    float* Vs = Ks + XA_KT * LDK;
    const int qi = blockIdx.x * 32 + lq;
    const int qc = qi < Nq ? qi : Nq - 1;
    const float sc = 1.4426950408889634f * rsqrtf((float)DH);
    // B operand of the score product: Q[q][2 j + g], j = 0 .. DH / 2 - 1
    float qf[DH / 2];
    {
        const float* qp = Q + (size_t)b * q_bstride + (size_t)qc * ldq + h * DH + lg;
#pragma unroll
        for (int j = 0; j < DH / 2; ++j) qf[j] = qp[2 * j] * sc;
    }
    f32x16 o[NDB];
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const float* kb = Kp + (size_t)b * Nk * ldk + h * DH;
    const float* vb = Vp + (size_t)b * Nk * ldv + h * DH;
    // tile loader: lane -> (key row lane >> 1, half row lane & 1): DH / 2 floats = DH / 8 float4 of K and of V
    const int trow = lane >> 1, thalf = (lane & 1) * (DH / 2);
    for (int j0 = wave * XA_KT; j0 < Nk; j0 += XA_KT * XA_WAVES) {
        {
            const int j = (j0 + trow) < Nk ? (j0 + trow) : Nk - 1;      // (rows past the end: their scores are masked, their V rows meet p = 0)
            const float4* kr = reinterpret_cast<const float4*>(kb + (size_t)j * ldk + thalf);
            const float4* vr = reinterpret_cast<const float4*>(vb + (size_t)j * ldv + thalf);
            float4 kreg[DH / 8], vreg[DH / 8];
#pragma unroll
            for (int d = 0; d < DH / 8; ++d) { kreg[d] = kr[d]; vreg[d] = vr[d]; }
            __builtin_amdgcn_wave_barrier();      // (the tile buffers are private to the wave: the previous tile's reads are behind us in program order)
#pragma unroll
            for (int d = 0; d < DH / 8; ++d) {
                *reinterpret_cast<float4*>(Ks + trow * LDK + thalf + 4 * d) = kreg[d];
                *reinterpret_cast<float4*>(Vs + trow * LDK + thalf + 4 * d) = vreg[d];
            }
            __builtin_amdgcn_wave_barrier();
        }
        // ---- S^T[key, q]: A = K[key = lq][2 j + lg]
        f32x16 st;
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
        for (int j = 0; j < DH / 2; ++j) st = __builtin_amdgcn_mfma_f32_32x32x2f32(Ks[lq * LDK + 2 * j + lg], qf[j], st, 0, 0, 0);
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
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
        // ---- O^T[d, q] += V^T[d, key] P^T[key, q]: k step r = the key pair (row(r, 0), row(r, 1)); A = V[row(r, lg)][32 db + lq], B = st[r]
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float* vrow = Vs + mfma32_row(r, lg) * LDK + lq;
#pragma unroll
            for (int db = 0; db < NDB; ++db) o[db] = __builtin_amdgcn_mfma_f32_32x32x2f32(vrow[32 * db], st[r], o[db], 0, 0, 0);
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
            float* op = O + ((size_t)b * Nq + qi) * ldo + h * DH + d0;
#pragma unroll
            for (int d = 0; d < CW; d += 4) *reinterpret_cast<float4*>(op + d) = make_float4(acc[d] * inv, acc[d + 1] * inv, acc[d + 2] * inv, acc[d + 3] * inv);
        }
    }
}

extern "C" int sed_xattn_f32_fwd(const float* Q, const float* K, const float* V, float* O, const uint8_t* mask, int B, int H, int Nq, int Nk,
                                 int head_dim, int ldq, int ldk, int ldv, int ldo, int64_t q_batch_stride, hipStream_t stream) {
    (void)hipGetLastError();
    if (B <= 0 || H <= 0 || Nq <= 0 || Nk <= 0 || (head_dim != 32 && head_dim != 64) || ((ldq | ldk | ldv | ldo) & 3) || B > 65535 || H > 65535)
        return SED_ERR_ARG;
    if (((uintptr_t)Q | (uintptr_t)K | (uintptr_t)V | (uintptr_t)O) & 15) return SED_ERR_ARG;
    const dim3 grid(cdiv(Nq, 32), H, B);
#define XATTN_LAUNCH(DH_, MK_)                                                                                                   \
    {                                                                                                                            \
        const int tile_ = XA_WAVES * 2 * XA_KT * (DH_ + 4) * 4, merge_ = XA_WAVES * 32 * (DH_ + 2) * 4;                          \
        const int lds_ = tile_ > merge_ ? tile_ : merge_;                                                                        \
        static bool attr_ = false;                                                                                               \
        if (!attr_) { (void)hipFuncSetAttribute((const void*)xattn_f32_fwd_kernel<DH_, MK_>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_); attr_ = true; } \
        hipLaunchKernelGGL((xattn_f32_fwd_kernel<DH_, MK_>), grid, dim3(64 * XA_WAVES), lds_, stream, Q, K, V, O, mask, Nq, Nk, ldq, ldk, ldv, ldo, \
                           (long long)q_batch_stride);                                                                           \
    }
    if (head_dim == 64) { if (mask != nullptr) XATTN_LAUNCH(64, true) else XATTN_LAUNCH(64, false) }
    else { if (mask != nullptr) XATTN_LAUNCH(32, true) else XATTN_LAUNCH(32, false) }
#undef XATTN_LAUNCH
    return sed_check_launch();
}

// ---------------------------------------------------------------------------------------------------------------------
// dual-stream finish (detect_any_sound.py:394-404): logits [B, T, Q] -> strong [B, Q, T] = clamp(pad ? 0 : sigmoid(logit / temp) * at[b, q],
// 1e-7, 1) through a 32 x 32 LDS transpose (coalesced on both sides), at[b, q] = sigmoid(at_logit[b, q]);
// weak [B, Q] = clamp(sum_t s^2 / sum_t s, 1e-7, 1).
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dasm_head_kernel(const float* __restrict__ logits, const float* __restrict__ at_logit,
                                                        const unsigned char* __restrict__ pad, float inv_temp, float* __restrict__ strong,
                                                        float* __restrict__ at_out, int T, int Qn) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, t0 = blockIdx.y * 32, q0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int t = t0 + ty + 8 * i, qn = q0 + tx;
        float v = 0.f;
        if (t < T && qn < Qn) {
            const float a = sigmoidf_(at_logit[(size_t)b * Qn + qn]);
            const bool masked = pad != nullptr && pad[(size_t)b * T + t] != 0;
            v = masked ? 0.f : sigmoidf_(logits[((size_t)b * T + t) * Qn + qn] * inv_temp) * a;
            v = fminf(fmaxf(v, 1e-7f), 1.0f);
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
                                 float* at_out, int B, int T, int Q, hipStream_t stream) {
    (void)hipGetLastError();
    if (B <= 0 || T <= 0 || Q <= 0 || B > 65535 || !(temp > 0.f)) return SED_ERR_ARG;
    hipLaunchKernelGGL(dasm_head_kernel, dim3(cdiv(Q, 32), cdiv(T, 32), B), dim3(256), 0, stream, logits, at_logit, pad_mask, 1.0f / temp, strong,
                       at_out, T, Q);
    hipLaunchKernelGGL(dasm_weak_kernel, dim3(B * Q), dim3(256), 0, stream, (const float*)strong, weak, T);
    return sed_check_launch();
}
