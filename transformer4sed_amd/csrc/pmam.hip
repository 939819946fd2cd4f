// HBM-bound kernels of the PMAM variant of the model path: LoRA weight merge, width-generic
// LayerNorm / masking (the 384-wide context network), the CNN branch around its im2col GEMMs (3x3 patch gather, BatchNorm affine,
// ContextGating + dropout + average pooling), the per-frame attention pooling over the 12 frequency tokens and the
// projector merge (two linear interpolations + learned mixing weight).  All fp32 math; 16-bit only as GEMM-operand outputs.
#include "common.h"
#include "../../include/sed_hip.h"

static inline int grid_for(size_t work, int threads = 256, int cap = 8192) {
    size_t b = (work + threads - 1) / threads;
    if (b > (size_t)cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}
__device__ __forceinline__ bf16_t cvt16(float v, int f16) { return f16 ? f2h(v) : f2bf(v); }
__device__ __forceinline__ float ld16(bf16_t v, int f16) { return f16 ? h2f(v) : bf2f(v); }

// ---------------------------------------------------------------------------------------------------
// LoRA:  W_eff = W + scaling * B A   (src/models/lora/layers.py:120-133, the eval-mode merge; also the operand image the
// train-mode forward multiplies with -- W x + s B (A x) = (W + s B A) x)
// ---------------------------------------------------------------------------------------------------
__global__ void lora_merge_kernel(const float* __restrict__ W, const float* __restrict__ A, const float* __restrict__ Bm,
                                  float scaling, float* __restrict__ out, int n_out, int k_in, int r) {
    const size_t total = (size_t)n_out * (k_in / 4);
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int k4 = (int)(idx % (k_in / 4));
        const int n = (int)(idx / (k_in / 4));
        float4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int j = 0; j < r; ++j) {
            const float b = Bm[(size_t)n * r + j];
            const float4 a = reinterpret_cast<const float4*>(A + (size_t)j * k_in)[k4];
            acc.x = fmaf(b, a.x, acc.x); acc.y = fmaf(b, a.y, acc.y); acc.z = fmaf(b, a.z, acc.z); acc.w = fmaf(b, a.w, acc.w);
        }
        float4 w = reinterpret_cast<const float4*>(W)[idx];
        w.x = fmaf(scaling, acc.x, w.x); w.y = fmaf(scaling, acc.y, w.y); w.z = fmaf(scaling, acc.z, w.z); w.w = fmaf(scaling, acc.w, w.w);
        reinterpret_cast<float4*>(out)[idx] = w;
    }
}
extern "C" int sed_lora_merge(const float* W, const float* A, const float* Bm, float scaling, float* out, int n_out,
                              int k_in, int r, hipStream_t stream) {
    (void)hipGetLastError();
    if (n_out <= 0 || k_in <= 0 || (k_in % 4) || r <= 0) return SED_ERR_ARG;
    hipLaunchKernelGGL(lora_merge_kernel, dim3(grid_for((size_t)n_out * (k_in / 4))), dim3(256), 0, stream, W, A, Bm, scaling, out,
                       n_out, k_in, r);
    return sed_check_launch();
}

// ---------------------------------------------------------------------------------------------------
// LayerNorm of any width D (multiple of 128, <= 1024):  y = LN(in_scale * x) * gamma + beta.  One wave per row, float2 per
// lane per 128-column group, two-pass statistics in registers.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ln_fwd_any_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float eps, float in_scale,
                                                         bf16_t* __restrict__ y16, float* __restrict__ y32, float* __restrict__ mean,
                                                         float* __restrict__ rstd, int M, int D, int f16) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int G = D / 128;
    float2 v[8];
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g)
        if (g < G) {
            v[g] = reinterpret_cast<const float2*>(x + (size_t)row * D)[lane + 64 * g];
            v[g].x *= in_scale; v[g].y *= in_scale;
            s += v[g].x + v[g].y;
        }
    const float mu = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g)
        if (g < G) { v[g].x -= mu; v[g].y -= mu; q += v[g].x * v[g].x + v[g].y * v[g].y; }
    const float rs = rsqrtf(wave_sum(q) / (float)D + eps);
    if (lane == 0) {
        if (mean != nullptr) mean[row] = mu;
        if (rstd != nullptr) rstd[row] = rs;
    }
#pragma unroll
    for (int g = 0; g < 8; ++g)
        if (g < G) {
            const float2 ga = reinterpret_cast<const float2*>(gamma)[lane + 64 * g], be = reinterpret_cast<const float2*>(beta)[lane + 64 * g];
            const float a = v[g].x * rs * ga.x + be.x, b = v[g].y * rs * ga.y + be.y;
            if (y32 != nullptr) reinterpret_cast<float2*>(y32 + (size_t)row * D)[lane + 64 * g] = make_float2(a, b);
            if (y16 != nullptr)
                reinterpret_cast<unsigned*>(y16 + (size_t)row * D)[lane + 64 * g] = (unsigned)cvt16(a, f16) | ((unsigned)cvt16(b, f16) << 16);
        }
}
extern "C" int sed_ln_fwd_any(const float* x, const float* gamma, const float* beta, float eps, float in_scale, void* y16,
                              float* y32, float* mean, float* rstd, int M, int D, int f16, hipStream_t stream) {
    (void)hipGetLastError();
    if (M <= 0 || D <= 0 || (D % 128) || D > 1024) return SED_ERR_ARG;
    hipLaunchKernelGGL(ln_fwd_any_kernel, dim3(cdiv(M, 4)), dim3(256), 0, stream, x, gamma, beta, eps, in_scale, (bf16_t*)y16, y32, mean,
                       rstd, M, D, f16);
    return sed_check_launch();
}

// MlmModule.setence_mask application for rows of any width C (multiple of 4); see sed_mlm_apply.
__global__ void mlm_apply_c_kernel(const float* __restrict__ x, const float* __restrict__ mask_token,
                                   const unsigned char* __restrict__ action, const int* __restrict__ src_idx,
                                   float* __restrict__ out, int rows, int C4) {
    const size_t total = (size_t)rows * C4;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int d4 = (int)(idx % C4);
        const int row = (int)(idx / C4);
        const unsigned char a = action[row];
        float4 v;
        if (a == 1) v = reinterpret_cast<const float4*>(mask_token)[d4];
        else if (a == 2) v = reinterpret_cast<const float4*>(x)[(size_t)src_idx[row] * C4 + d4];
        else v = reinterpret_cast<const float4*>(x)[idx];
        reinterpret_cast<float4*>(out)[idx] = v;
    }
}
extern "C" int sed_mlm_apply_c(const float* x, const float* mask_token, const uint8_t* action, const int* src_idx, float* out,
                               int rows, int C, hipStream_t stream) {
    (void)hipGetLastError();
    if (rows <= 0 || C <= 0 || (C % 4)) return SED_ERR_ARG;
    hipLaunchKernelGGL(mlm_apply_c_kernel, dim3(grid_for((size_t)rows * (C / 4))), dim3(256), 0, stream, x, mask_token, action, src_idx,
                       out, rows, C / 4);
    return sed_check_launch();
}

// ---------------------------------------------------------------------------------------------------
// CNN branch (src/models/cnn/base.py:62-113).  Activations are NHWC 16-bit [B, H = time, W = freq, Cp] with the channel
// count padded to a multiple of 64 (zeros), so that every convolution is one im2col gather + one NT GEMM.
// ---------------------------------------------------------------------------------------------------
// layer 0: mel [B, 128, T] fp32 -> col [B*T*128, 64]: 9 taps (time-major: tap = 3 * (dt + 1) + (df + 1)) + 55 zero columns
__global__ void conv0_im2col_kernel(const float* __restrict__ mel, bf16_t* __restrict__ col, int B, int T, int f16) {
    const size_t total = (size_t)B * T * 128;
    for (size_t m = (size_t)blockIdx.x * blockDim.x + threadIdx.x; m < total; m += (size_t)gridDim.x * blockDim.x) {
        const int f = (int)(m % 128);
        const int t = (int)((m / 128) % T);
        const int b = (int)(m / ((size_t)128 * T));
        bf16_t v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = 0;
#pragma unroll
        for (int dt = -1; dt <= 1; ++dt)
#pragma unroll
            for (int df = -1; df <= 1; ++df) {
                const int tt = t + dt, ff = f + df;
                float x = 0.f;
                if (tt >= 0 && tt < T && ff >= 0 && ff < 128) x = mel[((size_t)b * 128 + ff) * T + tt];
                v[3 * (dt + 1) + (df + 1)] = cvt16(x, f16);
            }
        uint4* dst = reinterpret_cast<uint4*>(col + m * 64);
        uint4 p0, p1;
        p0.x = (unsigned)v[0] | ((unsigned)v[1] << 16); p0.y = (unsigned)v[2] | ((unsigned)v[3] << 16);
        p0.z = (unsigned)v[4] | ((unsigned)v[5] << 16); p0.w = (unsigned)v[6] | ((unsigned)v[7] << 16);
        p1.x = v[8]; p1.y = 0; p1.z = 0; p1.w = 0;
        const uint4 z = {0, 0, 0, 0};
        dst[0] = p0; dst[1] = p1;
#pragma unroll
        for (int i = 2; i < 8; ++i) dst[i] = z;
    }
}
extern "C" int sed_conv0_im2col(const float* mel, void* col, int B, int T, int f16, hipStream_t stream) {
    (void)hipGetLastError();
    if (B <= 0 || T <= 0) return SED_ERR_ARG;
    hipLaunchKernelGGL(conv0_im2col_kernel, dim3(grid_for((size_t)B * T * 128)), dim3(256), 0, stream, mel, (bf16_t*)col, B, T, f16);
    return sed_check_launch();
}
// generic layer: X [B, H, W, Cp] -> col [B*H*W, Kp], column tap * C + c (tap as above, c < C), zeros up to Kp; 16-byte chunks
__global__ void conv3x3_im2col_kernel(const bf16_t* __restrict__ X, bf16_t* __restrict__ col, int B, int H, int W, int C, int Cp,
                                      int Kp) {
    const int chunks = Kp / 8;
    const size_t total = (size_t)B * H * W * chunks;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(idx % chunks);
        const size_t m = idx / chunks;
        const int k0 = j * 8;
        uint4 v = {0, 0, 0, 0};
        if (k0 < 9 * C) {
            const int tap = k0 / C, c = k0 - tap * C;
            const int w = (int)(m % W), h = (int)((m / W) % H);
            const int hh = h + tap / 3 - 1, ww = w + tap % 3 - 1;
            if (hh >= 0 && hh < H && ww >= 0 && ww < W) {
                const size_t b = m / ((size_t)W * H);
                v = *reinterpret_cast<const uint4*>(X + ((b * H + hh) * W + ww) * Cp + c);
            }
        }
        reinterpret_cast<uint4*>(col)[idx] = v;
    }
}
extern "C" int sed_conv3x3_im2col(const void* X, void* col, int B, int H, int W, int C, int Cp, int Kp, hipStream_t stream) {
    (void)hipGetLastError();
    if (B <= 0 || H <= 0 || W <= 0 || (C % 8) || (Cp % 8) || (Kp % 8) || Kp < 9 * C || Cp < C) return SED_ERR_ARG;
    hipLaunchKernelGGL(conv3x3_im2col_kernel, dim3(grid_for((size_t)B * H * W * (Kp / 8), 256, 16384)), dim3(256), 0, stream,
                       (const bf16_t*)X, (bf16_t*)col, B, H, W, C, Cp, Kp);
    return sed_check_launch();
}
// BatchNorm as a per-channel affine (eval: running statistics; train: the batch statistics folded by the host into a, b):
// Z16[m, c] = Y[m, c] * a[c] + b[c] for c < C, zero for C <= c < Cp.  Operand of the ContextGating GEMM.
__global__ void bn_act_kernel(const float* __restrict__ Y, int ldy, const float* __restrict__ a, const float* __restrict__ b,
                              bf16_t* __restrict__ Z, size_t M, int C, int Cp, int f16) {
    const int c4n = Cp / 4;
    const size_t total = M * c4n;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx % c4n) * 4;
        const size_t m = idx / c4n;
        uint2 p = {0, 0};
        if (c < C) {
            const float4 y = *reinterpret_cast<const float4*>(Y + m * ldy + c);
            const float4 aa = *reinterpret_cast<const float4*>(a + c), bb = *reinterpret_cast<const float4*>(b + c);
            p.x = (unsigned)cvt16(fmaf(y.x, aa.x, bb.x), f16) | ((unsigned)cvt16(fmaf(y.y, aa.y, bb.y), f16) << 16);
            p.y = (unsigned)cvt16(fmaf(y.z, aa.z, bb.z), f16) | ((unsigned)cvt16(fmaf(y.w, aa.w, bb.w), f16) << 16);
        }
        *reinterpret_cast<uint2*>(Z + m * Cp + c) = p;
    }
}
extern "C" int sed_bn_act(const float* Y, int ldy, const float* a, const float* b, void* Z, int64_t M, int C, int Cp, int f16,
                          hipStream_t stream) {
    (void)hipGetLastError();
    if (M <= 0 || (C % 4) || (Cp % 4) || Cp < C || (ldy % 4)) return SED_ERR_ARG;
    hipLaunchKernelGGL(bn_act_kernel, dim3(grid_for((size_t)M * (Cp / 4), 256, 16384)), dim3(256), 0, stream, Y, ldy, a, b, (bf16_t*)Z,
                       (size_t)M, C, Cp, f16);
    return sed_check_launch();
}
// ContextGating + dropout + average pooling:  out[b, ho, wo, c] = mean over the (ph x pw) window of z * sigmoid(l) * keep,
// z = Y * a + b (BatchNorm output, fp32), l = L (gate logits), keep = mask * drop_scale (mask NULL: 1).  16-bit NHWC output with
// channel pad Cpo (zeros) and/or fp32 [rows, C].
__global__ void cg_pool_kernel(const float* __restrict__ Y, int ldy, const float* __restrict__ a, const float* __restrict__ b,
                               const float* __restrict__ L, int ldl, const unsigned char* __restrict__ mask, float drop_scale,
                               bf16_t* __restrict__ out16, float* __restrict__ out32, int B, int H, int W, int C, int Cpo, int ph,
                               int pw, int f16) {
    const int Ho = H / ph, Wo = W / pw, c4n = Cpo / 4;
    const size_t total = (size_t)B * Ho * Wo * c4n;
    const float inv = 1.0f / (float)(ph * pw);
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx % c4n) * 4;
        const size_t mo = idx / c4n;
        float4 acc = {0.f, 0.f, 0.f, 0.f};
        if (c < C) {
            const int wo = (int)(mo % Wo), ho = (int)((mo / Wo) % Ho);
            const size_t bi = mo / ((size_t)Wo * Ho);
            const float4 aa = *reinterpret_cast<const float4*>(a + c), bb = *reinterpret_cast<const float4*>(b + c);
            for (int i = 0; i < ph; ++i)
                for (int j = 0; j < pw; ++j) {
                    const size_t m = (bi * H + (size_t)ho * ph + i) * W + (size_t)wo * pw + j;
                    const float4 y = *reinterpret_cast<const float4*>(Y + m * ldy + c);
                    const float4 l = *reinterpret_cast<const float4*>(L + m * ldl + c);
                    float4 k = {1.f, 1.f, 1.f, 1.f};
                    if (mask != nullptr) {
                        const uchar4 mk = *reinterpret_cast<const uchar4*>(mask + m * C + c);
                        k.x = mk.x ? drop_scale : 0.f; k.y = mk.y ? drop_scale : 0.f; k.z = mk.z ? drop_scale : 0.f; k.w = mk.w ? drop_scale : 0.f;
                    }
                    acc.x += fmaf(y.x, aa.x, bb.x) * sigmoidf_(l.x) * k.x;
                    acc.y += fmaf(y.y, aa.y, bb.y) * sigmoidf_(l.y) * k.y;
                    acc.z += fmaf(y.z, aa.z, bb.z) * sigmoidf_(l.z) * k.z;
                    acc.w += fmaf(y.w, aa.w, bb.w) * sigmoidf_(l.w) * k.w;
                }
            acc.x *= inv; acc.y *= inv; acc.z *= inv; acc.w *= inv;
            if (out32 != nullptr) *reinterpret_cast<float4*>(out32 + mo * C + c) = acc;
        }
        if (out16 != nullptr) {
            uint2 p;
            p.x = (unsigned)cvt16(acc.x, f16) | ((unsigned)cvt16(acc.y, f16) << 16);
            p.y = (unsigned)cvt16(acc.z, f16) | ((unsigned)cvt16(acc.w, f16) << 16);
            *reinterpret_cast<uint2*>(out16 + mo * Cpo + c) = p;
        }
    }
}
extern "C" int sed_cg_pool(const float* Y, int ldy, const float* a, const float* b, const float* L, int ldl,
                           const uint8_t* mask, float drop_scale, void* out16, float* out32, int B, int H, int W, int C,
                           int Cpo, int ph, int pw, int f16, hipStream_t stream) {
    (void)hipGetLastError();
    if (B <= 0 || ph <= 0 || pw <= 0 || (H % ph) || (W % pw) || (C % 4) || (Cpo % 4) || Cpo < C || (ldy % 4) || (ldl % 4))
        return SED_ERR_ARG;
    hipLaunchKernelGGL(cg_pool_kernel, dim3(grid_for((size_t)B * (H / ph) * (W / pw) * (Cpo / 4), 256, 16384)), dim3(256), 0, stream, Y,
                       ldy, a, b, L, ldl, mask, drop_scale, (bf16_t*)out16, out32, B, H, W, C, Cpo, ph, pw, f16);
    return sed_check_launch();
}

// ---------------------------------------------------------------------------------------------------
// `attention` frequency pooling (src/models/pooling.py:37-51 with 6 heads of 128, passt_sed.py:211-215): for every (clip, time
// column) one learned query attends over the 12 frequency tokens.  kv [B*N, 1536] = (k | v) projections of the out_norm'ed tokens
// (rows b*N + 2 + f*tp + t), q [768] the projected query.  One wave per head, six waves per (b, t).
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(384) void fpool_attn_fwd_kernel(const bf16_t* __restrict__ kv, const float* __restrict__ q,
                                                             bf16_t* __restrict__ out16, float* __restrict__ out32,
                                                             float* __restrict__ probs, int N, int tp, int f16) {
    const int lane = threadIdx.x & 63, h = threadIdx.x >> 6;
    const int b = blockIdx.x / tp, t = blockIdx.x % tp;
    const float2 qq = reinterpret_cast<const float2*>(q + h * 128)[lane];
    float s[12];
    float mx = -3.0e38f;
#pragma unroll
    for (int f = 0; f < 12; ++f) {
        const bf16_t* row = kv + ((size_t)b * N + 2 + (size_t)f * tp + t) * 1536 + h * 128;
        const unsigned kk = reinterpret_cast<const unsigned*>(row)[lane];
        const float d = qq.x * ld16((bf16_t)(kk & 0xffff), f16) + qq.y * ld16((bf16_t)(kk >> 16), f16);
        s[f] = wave_sum(d) * 0.08838834764831845f;   // 1 / sqrt(128)
        mx = fmaxf(mx, s[f]);
    }
    float den = 0.f;
#pragma unroll
    for (int f = 0; f < 12; ++f) { s[f] = __expf(s[f] - mx); den += s[f]; }
    const float inv = 1.0f / den;
    float ox = 0.f, oy = 0.f;
#pragma unroll
    for (int f = 0; f < 12; ++f) {
        const float p = s[f] * inv;
        const bf16_t* row = kv + ((size_t)b * N + 2 + (size_t)f * tp + t) * 1536 + 768 + h * 128;
        const unsigned vv = reinterpret_cast<const unsigned*>(row)[lane];
        ox = fmaf(p, ld16((bf16_t)(vv & 0xffff), f16), ox);
        oy = fmaf(p, ld16((bf16_t)(vv >> 16), f16), oy);
        if (probs != nullptr && lane == 0) probs[((size_t)blockIdx.x * 6 + h) * 12 + f] = p;
    }
    const size_t o = (size_t)blockIdx.x * 768 + h * 128;
    if (out16 != nullptr) reinterpret_cast<unsigned*>(out16 + o)[lane] = (unsigned)cvt16(ox, f16) | ((unsigned)cvt16(oy, f16) << 16);
    if (out32 != nullptr) reinterpret_cast<float2*>(out32 + o)[lane] = make_float2(ox, oy);
}
extern "C" int sed_fpool_attn_fwd(const void* kv, const float* q, void* out16, float* out32, float* probs, int B, int N, int tp,
                                  int f16, hipStream_t stream) {
    (void)hipGetLastError();
    if (B <= 0 || tp <= 0 || N != 2 + 12 * tp) return SED_ERR_ARG;
    hipLaunchKernelGGL(fpool_attn_fwd_kernel, dim3(B * tp), dim3(384), 0, stream, (const bf16_t*)kv, q, (bf16_t*)out16, out32, probs, N,
                       tp, f16);
    return sed_check_launch();
}

// ---------------------------------------------------------------------------------------------------
// Projector merge (passt_cnn.py:48-62, with both projections applied BEFORE the interpolations -- linear maps commute with the
// interpolation weights, which sum to one): out[b, j] = lerp_r1(pad(P1))[j] + mw * lerp_r2(P2)[j];  F.interpolate(mode='linear',
// align_corners=False) index arithmetic as in sed_interp_fwd.  P1 [B, tp1, C] is extended by `pad1` copies of its last frame.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void lerp_idx(int j, int ratio, int tin, int& i0, int& i1, float& lam) {
    float src = ((float)j + 0.5f) / (float)ratio - 0.5f;
    src = src < 0.f ? 0.f : src;
    i0 = (int)src;
    i1 = i0 + 1 < tin ? i0 + 1 : tin - 1;
    lam = src - (float)i0;
}
__global__ void pmam_merge_kernel(const float* __restrict__ P1, const float* __restrict__ P2, const float* __restrict__ mw,
                                  float* __restrict__ out, int B, int tp1, int pad1, int r1, int tp2, int r2, int C4) {
    const int T = (tp1 + pad1) * r1;
    const size_t total = (size_t)B * T * C4;
    const float w = mw[0];
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int d4 = (int)(idx % C4);
        const int j = (int)((idx / C4) % T);
        const size_t b = idx / ((size_t)C4 * T);
        int a0, a1, c0, c1;
        float la, lc;
        lerp_idx(j, r1, tp1 + pad1, a0, a1, la);
        a0 = a0 < tp1 ? a0 : tp1 - 1;
        a1 = a1 < tp1 ? a1 : tp1 - 1;
        lerp_idx(j, r2, tp2, c0, c1, lc);
        const float4 x0 = reinterpret_cast<const float4*>(P1)[(b * tp1 + a0) * C4 + d4], x1 = reinterpret_cast<const float4*>(P1)[(b * tp1 + a1) * C4 + d4];
        const float4 y0 = reinterpret_cast<const float4*>(P2)[(b * tp2 + c0) * C4 + d4], y1 = reinterpret_cast<const float4*>(P2)[(b * tp2 + c1) * C4 + d4];
        float4 o;
        o.x = ((1.f - la) * x0.x + la * x1.x) + w * ((1.f - lc) * y0.x + lc * y1.x);
        o.y = ((1.f - la) * x0.y + la * x1.y) + w * ((1.f - lc) * y0.y + lc * y1.y);
        o.z = ((1.f - la) * x0.z + la * x1.z) + w * ((1.f - lc) * y0.z + lc * y1.z);
        o.w = ((1.f - la) * x0.w + la * x1.w) + w * ((1.f - lc) * y0.w + lc * y1.w);
        reinterpret_cast<float4*>(out)[idx] = o;
    }
}
extern "C" int sed_pmam_merge(const float* P1, const float* P2, const float* mw, float* out, int B, int tp1, int pad1, int r1,
                              int tp2, int r2, int C, hipStream_t stream) {
    (void)hipGetLastError();
    if (B <= 0 || (C % 4) || (tp1 + pad1) * r1 != tp2 * r2) return SED_ERR_ARG;
    hipLaunchKernelGGL(pmam_merge_kernel, dim3(grid_for((size_t)B * tp2 * r2 * (C / 4))), dim3(256), 0, stream, P1, P2, mw, out, B, tp1,
                       pad1, r1, tp2, r2, C / 4);
    return sed_check_launch();
}
