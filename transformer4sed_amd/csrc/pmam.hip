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
// layer 0 with 16 filters, direct (round 4): Y[m, c] = bias[c] + sum_tap Wc[c, tap] patch(m, tap) on the fp32 spectrogram -- 144 FMAs per
// pixel instead of a [pixels, 64] 16-bit patch matrix (393 MB at batch 24) and a [pixels x 64] . [64 -> 128] GEMM over it.  Pixel m =
// (b, t, f), tap = 3 (dt + 1) + (df + 1) as above; Wc = conv0.weight [16, 1, 3, 3] as it lies.
__global__ __launch_bounds__(256) void conv0_fwd16_kernel(const float* __restrict__ mel, const float* __restrict__ Wc,
                                                          const float* __restrict__ bias, float* __restrict__ Y, int B, int T,
                                                          float* __restrict__ s1, float* __restrict__ s2) {
    __shared__ float wl[9][16], bl[16];
    __shared__ float sred[4][32];
    if (threadIdx.x < 144) wl[threadIdx.x % 9][threadIdx.x / 9] = Wc[threadIdx.x];
    if (threadIdx.x < 16) bl[threadIdx.x] = bias[threadIdx.x];
    __syncthreads();
    float t1[16], t2[16];      // (s1 != nullptr: the BatchNorm batch statistics sums, sed_colstats mode 0, from the values in registers)
#pragma unroll
    for (int c = 0; c < 16; ++c) { t1[c] = 0.f; t2[c] = 0.f; }
    const size_t total = (size_t)B * T * 128;
    for (size_t m = (size_t)blockIdx.x * blockDim.x + threadIdx.x; m < total; m += (size_t)gridDim.x * blockDim.x) {
        const int f = (int)(m % 128);
        const int t = (int)((m / 128) % T);
        const size_t b = m / ((size_t)128 * T);
        float acc[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) acc[c] = bl[c];
#pragma unroll
        for (int dt = -1; dt <= 1; ++dt)
#pragma unroll
            for (int df = -1; df <= 1; ++df) {
                const int tt = t + dt, ff = f + df;
                float x = 0.f;
                if (tt >= 0 && tt < T && ff >= 0 && ff < 128) x = mel[(b * 128 + ff) * T + tt];
                const int tap = 3 * (dt + 1) + (df + 1);
#pragma unroll
                for (int c = 0; c < 16; ++c) acc[c] = fmaf(wl[tap][c], x, acc[c]);
            }
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4*>(Y + m * 16 + 4 * q) = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
        if (s1 != nullptr) {
#pragma unroll
            for (int c = 0; c < 16; ++c) { t1[c] += acc[c]; t2[c] = fmaf(acc[c], acc[c], t2[c]); }
        }
    }
    if (s1 != nullptr) {
#pragma unroll
        for (int c = 0; c < 16; ++c) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { t1[c] += __shfl_xor(t1[c], o, 64); t2[c] += __shfl_xor(t2[c], o, 64); }
        }
        if ((threadIdx.x & 63) == 0) {
#pragma unroll
            for (int c = 0; c < 16; ++c) { sred[threadIdx.x >> 6][c] = t1[c]; sred[threadIdx.x >> 6][16 + c] = t2[c]; }
        }
        __syncthreads();
        if (threadIdx.x < 32) {
            const float v = (sred[0][threadIdx.x] + sred[1][threadIdx.x]) + (sred[2][threadIdx.x] + sred[3][threadIdx.x]);
            unsafeAtomicAdd((threadIdx.x < 16 ? s1 : s2) + (threadIdx.x & 15), v);
        }
    }
}
extern "C" int sed_conv0_fwd16(const float* mel, const float* Wc, const float* bias, float* Y, int B, int T, float* s1, float* s2,
                               hipStream_t stream) {
    (void)hipGetLastError();
    if (B <= 0 || T <= 0 || (s1 == nullptr) != (s2 == nullptr)) return SED_ERR_ARG;
    hipLaunchKernelGGL(conv0_fwd16_kernel, dim3(grid_for((size_t)B * T * 128, 256, 4096)), dim3(256), 0, stream, mel, Wc, bias, Y, B, T, s1, s2);
    return sed_check_launch();
}
// ... and its weight / bias gradient: dW[c, tap] += sum_m dY[m, c] patch(m, tap), dbias[c] += sum_m dY[m, c], as the streaming reduction of
// sed_small_dw with the patches gathered from the spectrogram (lane = (pixel slot, 4 taps): four lanes per pixel, 16 pixels per wave trip).
__global__ __launch_bounds__(256) void conv0_dw16_kernel(const bf16_t* __restrict__ dY, int ldy, const float* __restrict__ mel,
                                                         float* __restrict__ dW, int ldw, float* __restrict__ dbias, int B, int T,
                                                         int rows_per_wg) {
    __shared__ float red[4][16][16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cg = lane & 3, rs = lane >> 2;
    const long long M = (long long)B * T * 128;
    const long long m_begin = (long long)blockIdx.x * rows_per_wg, m_end = (m_begin + rows_per_wg) < M ? (m_begin + rows_per_wg) : M;
    float acc[16][4], bacc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        bacc[i] = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[i][e] = 0.f;
    }
    for (long long m0 = m_begin + (long long)wave * 32; m0 < m_end; m0 += 128) {      // two pixels per lane and trip
        float xv[2][4];
        uint4 dr[2][2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const long long m = m0 + 16 * u + rs;
            const bool ok = m < m_end;
            const int f = (int)(m % 128), t = (int)((m / 128) % T);
            const long long b = m / ((long long)128 * T);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int tap = 4 * cg + e, tt = t + tap / 3 - 1, ff = f + tap % 3 - 1;
                xv[u][e] = (ok && tap < 9 && tt >= 0 && tt < T && ff >= 0 && ff < 128) ? mel[(b * 128 + ff) * T + tt] : 0.f;
            }
            dr[u][0] = ok ? reinterpret_cast<const uint4*>(dY + m * ldy)[0] : make_uint4(0u, 0u, 0u, 0u);
            dr[u][1] = ok ? reinterpret_cast<const uint4*>(dY + m * ldy)[1] : make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const unsigned w[4] = {dr[u][q].x, dr[u][q].y, dr[u][q].z, dr[u][q].w};
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const float d0 = __uint_as_float(w[p] << 16), d1 = __uint_as_float(w[p] & 0xffff0000u);
                    const int i = 8 * q + 2 * p;
                    bacc[i] += d0; bacc[i + 1] += d1;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc[i][e] = fmaf(d0, xv[u][e], acc[i][e]);
                        acc[i + 1][e] = fmaf(d1, xv[u][e], acc[i + 1][e]);
                    }
                }
            }
    }
    for (int off = 4; off < 64; off <<= 1) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            bacc[i] += __shfl_xor(bacc[i], off, 64);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][e] += __shfl_xor(acc[i][e], off, 64);
        }
    }
    if (rs == 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) red[wave][i][4 * cg + e] = acc[i][e];
    }
    __syncthreads();
    {
        const int i = threadIdx.x >> 4, j = threadIdx.x & 15;
        if (j < 9) unsafeAtomicAdd(dW + (size_t)i * ldw + j, (red[0][i][j] + red[1][i][j]) + (red[2][i][j] + red[3][i][j]));
    }
    if (dbias != nullptr) {
        __syncthreads();
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) red[wave][0][i] = bacc[i];
        }
        __syncthreads();
        if (threadIdx.x < 16) unsafeAtomicAdd(dbias + threadIdx.x, (red[0][0][threadIdx.x] + red[1][0][threadIdx.x]) + (red[2][0][threadIdx.x] + red[3][0][threadIdx.x]));
    }
}
extern "C" int sed_conv0_dw16(const void* dY, int ldy, const float* mel, float* dW, int ldw, float* dbias, int B, int T, hipStream_t stream) {
    (void)hipGetLastError();
    if (B <= 0 || T <= 0 || (ldy % 8) || ldy < 16 || ldw < 9) return SED_ERR_ARG;
    const long long M = (long long)B * T * 128;
    long long rows = (M + 1023) / 1024;
    rows = (rows + 127) / 128 * 128;
    const int slabs = (int)((M + rows - 1) / rows);
    hipLaunchKernelGGL(conv0_dw16_kernel, dim3(slabs), dim3(256), 0, stream, (const bf16_t*)dY, ldy, mel, dW, ldw, dbias, B, T, (int)rows);
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

// The same for a 16-filter layer with the gate Linear inside (round 4): l = W_g z + b_g by 256 FMAs per pixel on the fp32 BatchNorm output --
// no 64-column 16-bit image of z, no [pixels x 64] . [64 -> 128] gate GEMM, no fp32 logits round trip.  One thread per OUTPUT pixel, the 16 x 16
// weight from LDS (every lane reads the same words: broadcasts).  Optional side outputs for the backward: the logits L [pixels, 16] fp32 and
// the 16-bit image Z [pixels, 16] of z (operand of the gate's weight gradient).
__global__ __launch_bounds__(256) void cg_gate16_pool_kernel(const float* __restrict__ Y, int ldy, const float* __restrict__ a,
                                                             const float* __restrict__ b, const float* __restrict__ Wg,
                                                             const float* __restrict__ bg, const unsigned char* __restrict__ mask,
                                                             float drop_scale, float* __restrict__ Lout, bf16_t* __restrict__ Zout,
                                                             bf16_t* __restrict__ out16, float* __restrict__ out32, int B, int H, int W,
                                                             int Cpo, int ph, int pw, int f16) {
    __shared__ __attribute__((aligned(16))) float wg[16][16];
    __shared__ float ab[3][16];
    wg[threadIdx.x >> 4][threadIdx.x & 15] = Wg[threadIdx.x];
    if (threadIdx.x < 16) { ab[0][threadIdx.x] = a[threadIdx.x]; ab[1][threadIdx.x] = b[threadIdx.x]; ab[2][threadIdx.x] = bg[threadIdx.x]; }
    __syncthreads();
    const int Ho = H / ph, Wo = W / pw;
    const size_t total = (size_t)B * Ho * Wo;
    const float inv = 1.0f / (float)(ph * pw);
    for (size_t mo = (size_t)blockIdx.x * blockDim.x + threadIdx.x; mo < total; mo += (size_t)gridDim.x * blockDim.x) {
        const int wo = (int)(mo % Wo), ho = (int)((mo / Wo) % Ho);
        const size_t bi = mo / ((size_t)Wo * Ho);
        float acc[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) acc[c] = 0.f;
        for (int i = 0; i < ph; ++i)
            for (int j = 0; j < pw; ++j) {
                const size_t m = (bi * H + (size_t)ho * ph + i) * W + (size_t)wo * pw + j;
                float z[16], l[16];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 y = *reinterpret_cast<const float4*>(Y + m * ldy + 4 * q);
                    z[4 * q] = fmaf(y.x, ab[0][4 * q], ab[1][4 * q]); z[4 * q + 1] = fmaf(y.y, ab[0][4 * q + 1], ab[1][4 * q + 1]);
                    z[4 * q + 2] = fmaf(y.z, ab[0][4 * q + 2], ab[1][4 * q + 2]); z[4 * q + 3] = fmaf(y.w, ab[0][4 * q + 3], ab[1][4 * q + 3]);
                }
#pragma unroll
                for (int c = 0; c < 16; ++c) {
                    float s = ab[2][c];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 w = *reinterpret_cast<const float4*>(&wg[c][4 * q]);
                        s = fmaf(w.x, z[4 * q], s); s = fmaf(w.y, z[4 * q + 1], s); s = fmaf(w.z, z[4 * q + 2], s); s = fmaf(w.w, z[4 * q + 3], s);
                    }
                    l[c] = s;
                }
                if (Lout != nullptr) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        *reinterpret_cast<float4*>(Lout + m * 16 + 4 * q) = make_float4(l[4 * q], l[4 * q + 1], l[4 * q + 2], l[4 * q + 3]);
                }
                if (Zout != nullptr) {
                    unsigned pk[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) pk[e] = (unsigned)cvt16(z[2 * e], f16) | ((unsigned)cvt16(z[2 * e + 1], f16) << 16);
                    uint4* zd = reinterpret_cast<uint4*>(Zout + m * 16);
                    zd[0] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
                    zd[1] = make_uint4(pk[4], pk[5], pk[6], pk[7]);
                }
                uint4 mk4 = make_uint4(0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u);
                if (mask != nullptr) mk4 = *reinterpret_cast<const uint4*>(mask + m * 16);
                const unsigned mw[4] = {mk4.x, mk4.y, mk4.z, mk4.w};
#pragma unroll
                for (int c = 0; c < 16; ++c) {
                    const float keep = ((mw[c >> 2] >> (8 * (c & 3))) & 0xffu) ? (mask != nullptr ? drop_scale : 1.0f) : 0.f;
                    acc[c] += z[c] * sigmoidf_(l[c]) * keep;
                }
            }
        if (out32 != nullptr) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<float4*>(out32 + mo * 16 + 4 * q) = make_float4(acc[4 * q] * inv, acc[4 * q + 1] * inv, acc[4 * q + 2] * inv, acc[4 * q + 3] * inv);
        }
        if (out16 != nullptr) {
            unsigned pk[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) pk[e] = (unsigned)cvt16(acc[2 * e] * inv, f16) | ((unsigned)cvt16(acc[2 * e + 1] * inv, f16) << 16);
            uint4* od = reinterpret_cast<uint4*>(out16 + mo * Cpo);
            od[0] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
            od[1] = make_uint4(pk[4], pk[5], pk[6], pk[7]);
            for (int c8 = 2; c8 < Cpo / 8; ++c8) od[c8] = make_uint4(0u, 0u, 0u, 0u);       // channel padding of the NHWC operand
        }
    }
}
extern "C" int sed_cg_gate16_pool(const float* Y, int ldy, const float* a, const float* b, const float* Wg, const float* bg,
                                  const uint8_t* mask, float drop_scale, float* Lout, void* Zout, void* out16, float* out32, int B, int H,
                                  int W, int Cpo, int ph, int pw, int f16, hipStream_t stream) {
    (void)hipGetLastError();
    if (B <= 0 || ph <= 0 || pw <= 0 || (H % ph) || (W % pw) || (Cpo % 8) || Cpo < 16 || (ldy % 4) || ldy < 16) return SED_ERR_ARG;
    hipLaunchKernelGGL(cg_gate16_pool_kernel, dim3(grid_for((size_t)B * (H / ph) * (W / pw), 256, 16384)), dim3(256), 0, stream, Y, ldy, a, b,
                       Wg, bg, mask, drop_scale, Lout, (bf16_t*)Zout, (bf16_t*)out16, out32, B, H, W, Cpo, ph, pw, f16);
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
    float src = (float)(1.0 / (double)ratio) * ((float)j + 0.5f) - 0.5f;   // torch: scale * (dst + 0.5) - 0.5, scale = (float)(1 / scale_factor)
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

// ===================================================================================================
// Backward kernels of the PMAM path
// ===================================================================================================
// LayerNorm backward of any width (see layernorm_bwd_kernel): dx = in_scale * rstd * (dyg - mean(dyg) - xhat * mean(dyg * xhat)),
// dyg = dy * gamma; dx is ADDED into dx_acc when accumulate != 0; dgamma / dbeta (+=, nullable).
__global__ __launch_bounds__(256) void ln_bwd_any_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         const float* __restrict__ gamma, float in_scale, float* __restrict__ dx_acc,
                                                         int accumulate, float* __restrict__ dgamma, float* __restrict__ dbeta, int M,
                                                         int D) {
    __shared__ float red[4][1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, G = D / 128;
    float2 g[8], pg[8], pb[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        pg[k] = make_float2(0.f, 0.f); pb[k] = make_float2(0.f, 0.f);
        if (k < G) g[k] = reinterpret_cast<const float2*>(gamma)[lane + 64 * k];
    }
    for (int row = blockIdx.x * 4 + wave; row < M; row += gridDim.x * 4) {
        const float mu = mean[row], rs = rstd[row];
        float2 d[8], xh[8];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (k < G) {
                const float2 dv = reinterpret_cast<const float2*>(dy + (size_t)row * D)[lane + 64 * k];
                const float2 xv = reinterpret_cast<const float2*>(x + (size_t)row * D)[lane + 64 * k];
                xh[k] = make_float2((xv.x * in_scale - mu) * rs, (xv.y * in_scale - mu) * rs);
                pg[k].x += dv.x * xh[k].x; pg[k].y += dv.y * xh[k].y;
                pb[k].x += dv.x; pb[k].y += dv.y;
                d[k] = make_float2(dv.x * g[k].x, dv.y * g[k].y);
                s1 += d[k].x + d[k].y;
                s2 += d[k].x * xh[k].x + d[k].y * xh[k].y;
            }
        s1 = wave_sum(s1) / (float)D;
        s2 = wave_sum(s2) / (float)D;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (k < G) {
                float2* o = reinterpret_cast<float2*>(dx_acc + (size_t)row * D) + lane + 64 * k;
                float2 v = make_float2(in_scale * rs * (d[k].x - s1 - xh[k].x * s2), in_scale * rs * (d[k].y - s1 - xh[k].y * s2));
                if (accumulate) { const float2 old = *o; v.x += old.x; v.y += old.y; }
                *o = v;
            }
    }
    if (dgamma == nullptr) return;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (k < G) reinterpret_cast<float2*>(red[wave])[lane + 64 * k] = pass == 0 ? pg[k] : pb[k];
        __syncthreads();
        for (int c = threadIdx.x; c < D; c += 256)
            unsafeAtomicAdd(pass == 0 ? &dgamma[c] : &dbeta[c], red[0][c] + red[1][c] + red[2][c] + red[3][c]);
        __syncthreads();
    }
}
extern "C" int sed_ln_bwd_any(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                              float in_scale, float* dx, int accumulate, float* dgamma, float* dbeta, int M, int D,
                              hipStream_t stream) {
    (void)hipGetLastError();
    if (M <= 0 || D <= 0 || (D % 128) || D > 1024 || (dgamma == nullptr) != (dbeta == nullptr)) return SED_ERR_ARG;
    int blocks = cdiv(M, 4);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(ln_bwd_any_kernel, dim3(blocks), dim3(256), 0, stream, dy, x, mean, rstd, gamma, in_scale, dx, accumulate, dgamma,
                       dbeta, M, D);
    return sed_check_launch();
}

// backward of sed_mlm_apply_c: dx (zero-initialised) receives kept rows and the scatter of 'copy' rows, dtoken the 'mask' rows
__global__ __launch_bounds__(256) void mlm_apply_bwd_c_kernel(const float* __restrict__ dout, const unsigned char* __restrict__ action,
                                                              const int* __restrict__ src_idx, float* __restrict__ dx,
                                                              float* __restrict__ dtoken, int rows, int C) {
    // a workgroup walks a slab of rows with one thread per column (C <= 1024): the 'mask' rows (most of the masked frames) are summed
    // in registers and leave as ONE atomic per column and workgroup instead of one per element on the same C addresses
    float tok[4] = {0.f, 0.f, 0.f, 0.f};
    const int per = (rows + gridDim.x - 1) / gridDim.x, r0 = blockIdx.x * per, r1 = r0 + per < rows ? r0 + per : rows;
    for (int row = r0; row < r1; ++row) {
        const unsigned char a = action[row];
        const int src = a == 2 ? src_idx[row] : row;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int d = threadIdx.x + 256 * u;
            if (d >= C) break;
            const float g = dout[(size_t)row * C + d];
            if (a == 1) tok[u] += g;
            else unsafeAtomicAdd(&dx[(size_t)src * C + d], g);   // 'copy' rows scatter onto rows that also keep their own gradient
        }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int d = threadIdx.x + 256 * u;
        if (d < C && tok[u] != 0.f) unsafeAtomicAdd(&dtoken[d], tok[u]);
    }
}
extern "C" int sed_mlm_apply_bwd_c(const float* dout, const uint8_t* action, const int* src_idx, float* dx_zeroed, float* dtoken,
                                   int rows, int C, hipStream_t stream) {
    (void)hipGetLastError();
    if (rows <= 0 || C <= 0 || C > 1024) return SED_ERR_ARG;
    int blocks = rows < 2048 ? rows : 2048;
    hipLaunchKernelGGL(mlm_apply_bwd_c_kernel, dim3(blocks), dim3(256), 0, stream, dout, action, src_idx, dx_zeroed, dtoken, rows, C);
    return sed_check_launch();
}

// backward of sed_pmam_merge: dP1 [B, tp1, C], dP2 [B, tp2, C] (gather over the output frames each input frame feeds) and
// dmw[0] += sum(g * lerp_r2(P2)).  grid.y: 0 -> dP1, 1 -> dP2 (+ dmw).
__global__ void pmam_merge_bwd_kernel(const float* __restrict__ g, const float* __restrict__ P2, const float* __restrict__ mw,
                                      float* __restrict__ dP1, float* __restrict__ dP2, float* __restrict__ dmw, int B, int tp1,
                                      int pad1, int r1, int tp2, int r2, int C4) {
    const int T = (tp1 + pad1) * r1;
    const bool second = blockIdx.y == 1;
    const int tin = second ? tp2 : tp1, ratio = second ? r2 : r1, tlen = second ? tp2 : tp1 + pad1;
    const size_t total = (size_t)B * tin * C4;
    const float w = mw[0];
    double dw_part = 0.0;      // (the merge weight's gradient is a sum of ~10^6 products that cancel to ~10^-5 of their size: fp64 partials)
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int d4 = (int)(idx % C4);
        const int i = (int)((idx / C4) % tin);
        const size_t b = idx / ((size_t)C4 * tin);
        int jlo = (i - 2) * ratio, jhi = (i == tin - 1) ? T - 1 : (i + 2) * ratio;
        jlo = jlo < 0 ? 0 : jlo;
        jhi = jhi > T - 1 ? T - 1 : jhi;
        float4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int j = jlo; j <= jhi; ++j) {
            int i0, i1;
            float lam;
            lerp_idx(j, ratio, tlen, i0, i1, lam);
            i0 = i0 < tin ? i0 : tin - 1;
            i1 = i1 < tin ? i1 : tin - 1;
            float wt = 0.f;
            if (i0 == i) wt += 1.f - lam;
            if (i1 == i) wt += lam;
            if (wt != 0.f) {
                const float4 gv = reinterpret_cast<const float4*>(g)[(b * T + j) * C4 + d4];
                acc.x += wt * gv.x; acc.y += wt * gv.y; acc.z += wt * gv.z; acc.w += wt * gv.w;
            }
        }
        if (second) {
            const float4 p = reinterpret_cast<const float4*>(P2)[idx];
            dw_part += ((double)acc.x * p.x + (double)acc.y * p.y) + ((double)acc.z * p.z + (double)acc.w * p.w);    // sum_j g lerp(P2) = sum_i (lerp^T g)_i P2_i
            acc.x *= w; acc.y *= w; acc.z *= w; acc.w *= w;
            reinterpret_cast<float4*>(dP2)[idx] = acc;
        } else {
            reinterpret_cast<float4*>(dP1)[idx] = acc;
        }
    }
    if (second && dmw != nullptr) {
        __shared__ double wsum[4];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) dw_part += __shfl_xor(dw_part, o, 64);
        if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = dw_part;
        __syncthreads();
        if (threadIdx.x == 0) unsafeAtomicAdd(dmw, (float)((wsum[0] + wsum[1]) + (wsum[2] + wsum[3])));
    }
}
extern "C" int sed_pmam_merge_bwd(const float* g, const float* P2, const float* mw, float* dP1, float* dP2, float* dmw, int B,
                                  int tp1, int pad1, int r1, int tp2, int r2, int C, hipStream_t stream) {
    (void)hipGetLastError();
    if (B <= 0 || (C % 4) || (tp1 + pad1) * r1 != tp2 * r2) return SED_ERR_ARG;
    hipLaunchKernelGGL(pmam_merge_bwd_kernel, dim3(grid_for((size_t)B * tp2 * (C / 4), 256, 2048), 2), dim3(256), 0, stream, g, P2, mw,
                       dP1, dP2, dmw, B, tp1, pad1, r1, tp2, r2, C / 4);
    return sed_check_launch();
}

// ---------------------------------------------------------------------------------------------------
// Column statistics over the rows of fp32 matrices: s1[c] += sum_m A[m, c] (* B[m, c] when B != NULL ... ) -- the BatchNorm
// batch statistics and its backward sums.  mode 0: s1 = sum A, s2 = sum A^2.  mode 1: s1 = sum A, s2 = sum A * (B * a + b)
// (A = dZ, B = Y: the BatchNorm backward sums with xhat folded into the affine a, b).
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void colstats_kernel(const float* __restrict__ A, int lda, const float* __restrict__ Bm, int ldb,
                                                       const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ s1,
                                                       float* __restrict__ s2, size_t M, int C, int mode) {
    // a block covers Cw = min(C, 64) columns (grid.x column groups) and 256 / Cw rows per iteration (grid.y row slabs), so that
    // narrow matrices (16 / 32 channels) still use every lane; partials are combined through LDS before the atomics
    const int Cw = C < 64 ? C : 64, rpi = 256 / Cw;
    const int cl = threadIdx.x % Cw, rsub = threadIdx.x / Cw;
    const int c = blockIdx.x * 64 + cl;
    __shared__ float r1[256], r2[256];
    float t1 = 0.f, t2 = 0.f;
    if (c < C && rsub < rpi) {
        const float aa = mode ? a[c] : 0.f, bb = mode ? b[c] : 0.f;
        for (size_t m = (size_t)blockIdx.y * rpi + rsub; m < M; m += (size_t)gridDim.y * rpi) {
            const float v = A[m * lda + c];
            t1 += v;
            t2 += mode ? v * fmaf(Bm[m * ldb + c], aa, bb) : v * v;
        }
    }
    r1[threadIdx.x] = t1; r2[threadIdx.x] = t2;
    __syncthreads();
    if (rsub == 0 && c < C) {
        float u1 = 0.f, u2 = 0.f;
        for (int j = 0; j < rpi; ++j) { u1 += r1[j * Cw + cl]; u2 += r2[j * Cw + cl]; }
        unsafeAtomicAdd(&s1[c], u1);
        unsafeAtomicAdd(&s2[c], u2);
    }
}
extern "C" int sed_colstats(const float* A, int lda, const float* Bm, int ldb, const float* a, const float* b, float* s1,
                            float* s2, int64_t M, int C, int mode, hipStream_t stream) {
    (void)hipGetLastError();
    if (M <= 0 || C <= 0 || (C < 64 && (256 % C)) || (mode && (Bm == nullptr || a == nullptr || b == nullptr))) return SED_ERR_ARG;
    const int rpi = 256 / (C < 64 ? C : 64);
    int slabs = (int)((M + (size_t)rpi * 16 - 1) / ((size_t)rpi * 16));
    if (slabs > 2048) slabs = 2048;
    if (slabs < 1) slabs = 1;
    hipLaunchKernelGGL(colstats_kernel, dim3(cdiv(C, 64), slabs), dim3(256), 0, stream, A, lda, Bm, ldb, a, b, s1, s2, (size_t)M, C, mode);
    return sed_check_launch();
}

// backward of sed_cg_pool: dout (fp32 [B, Ho, Wo, C]) -> dzd [M, C] fp32 = up * sigmoid(l)  (direct path of z) and
// dL16 [M, ldl16] bf16 = up * z * s * (1 - s)  (gate logits; columns C..ldl16-1 zero), up = dout / (ph pw) * keep.
__global__ void cg_pool_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ Y, int ldy, const float* __restrict__ a,
                                   const float* __restrict__ b, const float* __restrict__ L, int ldl,
                                   const unsigned char* __restrict__ mask, float drop_scale, float* __restrict__ dzd, int ldz,
                                   bf16_t* __restrict__ dL16, int ldl16, int B, int H, int W, int C, int ph, int pw) {
    const int Ho = H / ph, Wo = W / pw, c4n = ldl16 / 4;
    const size_t total = (size_t)B * H * W * c4n;
    const float inv = 1.0f / (float)(ph * pw);
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx % c4n) * 4;
        const size_t m = idx / c4n;
        uint2 p = {0, 0};
        if (c < C) {
            const int w = (int)(m % W), h = (int)((m / W) % H);
            const size_t bi = m / ((size_t)W * H);
            const size_t mo = (bi * Ho + h / ph) * Wo + w / pw;
            const float4 g = *reinterpret_cast<const float4*>(dout + mo * C + c);
            const float4 y = *reinterpret_cast<const float4*>(Y + m * ldy + c), l = *reinterpret_cast<const float4*>(L + m * ldl + c);
            const float4 aa = *reinterpret_cast<const float4*>(a + c), bb = *reinterpret_cast<const float4*>(b + c);
            float4 k = {inv, inv, inv, inv};
            if (mask != nullptr) {
                const uchar4 mk = *reinterpret_cast<const uchar4*>(mask + m * C + c);
                k.x = mk.x ? inv * drop_scale : 0.f; k.y = mk.y ? inv * drop_scale : 0.f;
                k.z = mk.z ? inv * drop_scale : 0.f; k.w = mk.w ? inv * drop_scale : 0.f;
            }
            const float sx = sigmoidf_(l.x), sy = sigmoidf_(l.y), sz = sigmoidf_(l.z), sw = sigmoidf_(l.w);
            const float ux = g.x * k.x, uy = g.y * k.y, uz = g.z * k.z, uw = g.w * k.w;
            *reinterpret_cast<float4*>(dzd + m * ldz + c) = make_float4(ux * sx, uy * sy, uz * sz, uw * sw);
            p.x = pack2bf(ux * fmaf(y.x, aa.x, bb.x) * sx * (1.f - sx), uy * fmaf(y.y, aa.y, bb.y) * sy * (1.f - sy));
            p.y = pack2bf(uz * fmaf(y.z, aa.z, bb.z) * sz * (1.f - sz), uw * fmaf(y.w, aa.w, bb.w) * sw * (1.f - sw));
        }
        else if (c < ldz) *reinterpret_cast<float4*>(dzd + m * ldz + c) = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<uint2*>(dL16 + m * ldl16 + c) = p;
    }
}
extern "C" int sed_cg_pool_bwd(const float* dout, const float* Y, int ldy, const float* a, const float* b, const float* L, int ldl,
                               const uint8_t* mask, float drop_scale, float* dzd, int ldz, void* dL16, int ldl16, int B, int H,
                               int W, int C, int ph, int pw, hipStream_t stream) {
    (void)hipGetLastError();
    if (B <= 0 || ph <= 0 || pw <= 0 || (H % ph) || (W % pw) || (C % 4) || (ldl16 % 4) || ldl16 < C || (ldy % 4) || (ldl % 4) || (ldz % 4) || ldz < C || ldz > ldl16)
        return SED_ERR_ARG;
    hipLaunchKernelGGL(cg_pool_bwd_kernel, dim3(grid_for((size_t)B * H * W * (ldl16 / 4), 256, 16384)), dim3(256), 0, stream, dout, Y, ldy, a,
                       b, L, ldl, mask, drop_scale, dzd, ldz, (bf16_t*)dL16, ldl16, B, H, W, C, ph, pw);
    return sed_check_launch();
}
// backward of sed_cg_gate16_pool (16 filters): as sed_cg_pool_bwd, with the gate Linear's input gradient added in the same pass --
// dz[m, c] = up * sigmoid(l) + sum_i dl[m, i] Wg[i, c] (fp32 dl; the NT GEMM dz += dL16 . Wg it replaces read a bf16 dl) -- and dL16 as a
// 16-column bf16 image (operand of the gate's weight gradient, sed_small_dw).  One thread per input pixel.
__global__ __launch_bounds__(256) void cg_gate16_pool_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ Y, int ldy,
                                                                 const float* __restrict__ a, const float* __restrict__ b,
                                                                 const float* __restrict__ L, const float* __restrict__ Wg,
                                                                 const unsigned char* __restrict__ mask, float drop_scale,
                                                                 float* __restrict__ dz, bf16_t* __restrict__ dL16, int B, int H, int W,
                                                                 int ph, int pw, const float* __restrict__ ah, const float* __restrict__ bh,
                                                                 float* __restrict__ s1, float* __restrict__ s2) {
    __shared__ __attribute__((aligned(16))) float wg[16][16];
    __shared__ float ab[4][16];
    __shared__ float sred[4][32];
    wg[threadIdx.x >> 4][threadIdx.x & 15] = Wg[threadIdx.x];
    if (threadIdx.x < 16) {
        ab[0][threadIdx.x] = a[threadIdx.x]; ab[1][threadIdx.x] = b[threadIdx.x];
        ab[2][threadIdx.x] = s1 != nullptr ? ah[threadIdx.x] : 0.f; ab[3][threadIdx.x] = s1 != nullptr ? bh[threadIdx.x] : 0.f;
    }
    __syncthreads();
    // (s1 != nullptr: also the BatchNorm backward sums of this layer, s1[c] += sum dz, s2[c] += sum dz * xhat with xhat = Y ah + bh --
    //  sed_colstats mode 1 -- from the values in registers)
    float t1[16], t2[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) { t1[c] = 0.f; t2[c] = 0.f; }
    const int Ho = H / ph, Wo = W / pw;
    const size_t total = (size_t)B * H * W;
    const float inv = 1.0f / (float)(ph * pw);
    for (size_t m = (size_t)blockIdx.x * blockDim.x + threadIdx.x; m < total; m += (size_t)gridDim.x * blockDim.x) {
        const int w = (int)(m % W), h = (int)((m / W) % H);
        const size_t bi = m / ((size_t)W * H);
        const size_t mo = (bi * Ho + h / ph) * Wo + w / pw;
        uint4 mk4 = make_uint4(0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u);
        if (mask != nullptr) mk4 = *reinterpret_cast<const uint4*>(mask + m * 16);
        const unsigned mw[4] = {mk4.x, mk4.y, mk4.z, mk4.w};
        const float kscale = mask != nullptr ? inv * drop_scale : inv;
        float dzv[16], dl[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 g = *reinterpret_cast<const float4*>(dout + mo * 16 + 4 * q);
            const float4 y = *reinterpret_cast<const float4*>(Y + m * ldy + 4 * q);
            const float4 l = *reinterpret_cast<const float4*>(L + m * 16 + 4 * q);
            const float gg[4] = {g.x, g.y, g.z, g.w}, yy[4] = {y.x, y.y, y.z, y.w}, ll[4] = {l.x, l.y, l.z, l.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = 4 * q + e;
                const float keep = ((mw[q] >> (8 * e)) & 0xffu) ? kscale : 0.f;
                const float sg = sigmoidf_(ll[e]), up = gg[e] * keep;
                dzv[c] = up * sg;
                dl[c] = up * fmaf(yy[e], ab[0][c], ab[1][c]) * sg * (1.f - sg);
            }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 wv = *reinterpret_cast<const float4*>(&wg[i][4 * q]);
                dzv[4 * q] = fmaf(dl[i], wv.x, dzv[4 * q]); dzv[4 * q + 1] = fmaf(dl[i], wv.y, dzv[4 * q + 1]);
                dzv[4 * q + 2] = fmaf(dl[i], wv.z, dzv[4 * q + 2]); dzv[4 * q + 3] = fmaf(dl[i], wv.w, dzv[4 * q + 3]);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4*>(dz + m * 16 + 4 * q) = make_float4(dzv[4 * q], dzv[4 * q + 1], dzv[4 * q + 2], dzv[4 * q + 3]);
        if (s1 != nullptr) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 y = *reinterpret_cast<const float4*>(Y + m * ldy + 4 * q);      // (L1 hit: read above)
                const float yy[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int c = 4 * q + e;
                    t1[c] += dzv[c];
                    t2[c] = fmaf(dzv[c], fmaf(yy[e], ab[2][c], ab[3][c]), t2[c]);
                }
            }
        }
        unsigned pk[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) pk[e] = pack2bf(dl[2 * e], dl[2 * e + 1]);
        uint4* dd = reinterpret_cast<uint4*>(dL16 + m * 16);
        dd[0] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
        dd[1] = make_uint4(pk[4], pk[5], pk[6], pk[7]);
    }
    if (s1 != nullptr) {
#pragma unroll
        for (int c = 0; c < 16; ++c) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { t1[c] += __shfl_xor(t1[c], o, 64); t2[c] += __shfl_xor(t2[c], o, 64); }
        }
        if ((threadIdx.x & 63) == 0) {
#pragma unroll
            for (int c = 0; c < 16; ++c) { sred[threadIdx.x >> 6][c] = t1[c]; sred[threadIdx.x >> 6][16 + c] = t2[c]; }
        }
        __syncthreads();
        if (threadIdx.x < 32) {
            const float v = (sred[0][threadIdx.x] + sred[1][threadIdx.x]) + (sred[2][threadIdx.x] + sred[3][threadIdx.x]);
            unsafeAtomicAdd((threadIdx.x < 16 ? s1 : s2) + (threadIdx.x & 15), v);
        }
    }
}
extern "C" int sed_cg_gate16_pool_bwd(const float* dout, const float* Y, int ldy, const float* a, const float* b, const float* L,
                                      const float* Wg, const uint8_t* mask, float drop_scale, float* dz, void* dL16, int B, int H, int W,
                                      int ph, int pw, const float* ah, const float* bh, float* s1, float* s2, hipStream_t stream) {
    (void)hipGetLastError();
    if (B <= 0 || ph <= 0 || pw <= 0 || (H % ph) || (W % pw) || (ldy % 4) || ldy < 16) return SED_ERR_ARG;
    if ((s1 == nullptr) != (s2 == nullptr) || (s1 != nullptr && (ah == nullptr || bh == nullptr))) return SED_ERR_ARG;
    hipLaunchKernelGGL(cg_gate16_pool_bwd_kernel, dim3(grid_for((size_t)B * H * W, 256, 4096)), dim3(256), 0, stream, dout, Y, ldy, a, b, L, Wg,
                       mask, drop_scale, dz, (bf16_t*)dL16, B, H, W, ph, pw, ah, bh, s1, s2);
    return sed_check_launch();
}
// BatchNorm backward (batch statistics), given the column sums s1 = sum dz, s2 = sum dz * xhat (xhat = Y * ah + bh with
// ah = rstd, bh = -mean * rstd):  dY = gamma * rstd * (dz - s1 / M - xhat * s2 / M) -> bf16 [M, ldo] (columns C.. zero).
__global__ void bn_bwd_kernel(const float* __restrict__ dz, int ldz, const float* __restrict__ Y, int ldy, const float* __restrict__ ah,
                              const float* __restrict__ bh, const float* __restrict__ gamma, const float* __restrict__ s1,
                              const float* __restrict__ s2, bf16_t* __restrict__ dY, int ldo, size_t M, int C) {
    const int c4n = ldo / 4;
    const size_t total = M * c4n;
    const float invM = 1.0f / (float)M;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx % c4n) * 4;
        const size_t m = idx / c4n;
        uint2 p = {0, 0};
        if (c < C) {
            float o[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float xh = fmaf(Y[m * ldy + c + u], ah[c + u], bh[c + u]);
                o[u] = gamma[c + u] * ah[c + u] * (dz[m * ldz + c + u] - s1[c + u] * invM - xh * s2[c + u] * invM);
            }
            p.x = pack2bf(o[0], o[1]);
            p.y = pack2bf(o[2], o[3]);
        }
        *reinterpret_cast<uint2*>(dY + m * ldo + c) = p;
    }
}
extern "C" int sed_bn_bwd(const float* dz, int ldz, const float* Y, int ldy, const float* ah, const float* bh, const float* gamma,
                          const float* s1, const float* s2, void* dY, int ldo, int64_t M, int C, hipStream_t stream) {
    (void)hipGetLastError();
    if (M <= 0 || (C % 4) || (ldo % 4) || ldo < C) return SED_ERR_ARG;
    hipLaunchKernelGGL(bn_bwd_kernel, dim3(grid_for((size_t)M * (ldo / 4), 256, 16384)), dim3(256), 0, stream, dz, ldz, Y, ldy, ah, bh, gamma,
                       s1, s2, (bf16_t*)dY, ldo, (size_t)M, C);
    return sed_check_launch();
}
// col2im: dX[b, h, w, c] = sum over the 9 taps of dcol[(b, h - dh, w - dw), tap * C + c]  (gather form of the im2col transpose)
__global__ void col2im3x3_kernel(const bf16_t* __restrict__ dcol, int Kp, float* __restrict__ dX, int B, int H, int W, int C) {
    const int c4n = C / 4;
    const size_t total = (size_t)B * H * W * c4n;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx % c4n) * 4;
        const size_t m = idx / c4n;
        const int w = (int)(m % W), h = (int)((m / W) % H);
        const size_t b = m / ((size_t)W * H);
        float4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int hh = h - (tap / 3 - 1), ww = w - (tap % 3 - 1);   // the output pixel whose tap `tap` reads (h, w)
            if (hh < 0 || hh >= H || ww < 0 || ww >= W) continue;
            const uint2 v = *reinterpret_cast<const uint2*>(dcol + ((b * H + hh) * W + ww) * Kp + tap * C + c);
            acc.x += bf2f((bf16_t)(v.x & 0xffff)); acc.y += bf2f((bf16_t)(v.x >> 16));
            acc.z += bf2f((bf16_t)(v.y & 0xffff)); acc.w += bf2f((bf16_t)(v.y >> 16));
        }
        *reinterpret_cast<float4*>(dX + m * C + c) = acc;
    }
}
extern "C" int sed_col2im3x3(const void* dcol, int Kp, float* dX, int B, int H, int W, int C, hipStream_t stream) {
    (void)hipGetLastError();
    if (B <= 0 || (C % 4) || Kp < 9 * C) return SED_ERR_ARG;
    hipLaunchKernelGGL(col2im3x3_kernel, dim3(grid_for((size_t)B * H * W * (C / 4), 256, 16384)), dim3(256), 0, stream, (const bf16_t*)dcol,
                       Kp, dX, B, H, W, C);
    return sed_check_launch();
}

// backward of sed_fpool_attn_fwd: dout [B*tp, 768] fp32 -> dkv bf16 [B*N, 1536] (the 12 token rows of the column; cls / dist rows
// of each clip zeroed by the t == 0 workgroups), dq [768] (+=).
__global__ __launch_bounds__(384) void fpool_attn_bwd_kernel(const bf16_t* __restrict__ kv, const float* __restrict__ q,
                                                             const float* __restrict__ probs, const float* __restrict__ dout,
                                                             bf16_t* __restrict__ dkv, float* __restrict__ dq, int N, int tp, int f16) {
    const int lane = threadIdx.x & 63, h = threadIdx.x >> 6;
    const int b = blockIdx.x / tp, t = blockIdx.x % tp;
    const float2 qq = reinterpret_cast<const float2*>(q + h * 128)[lane];
    const float2 go = reinterpret_cast<const float2*>(dout + (size_t)blockIdx.x * 768 + h * 128)[lane];
    float p[12], dp[12];
    float dot = 0.f;
#pragma unroll
    for (int f = 0; f < 12; ++f) {
        p[f] = probs[((size_t)blockIdx.x * 6 + h) * 12 + f];
        const bf16_t* row = kv + ((size_t)b * N + 2 + (size_t)f * tp + t) * 1536 + 768 + h * 128;
        const unsigned vv = reinterpret_cast<const unsigned*>(row)[lane];
        dp[f] = wave_sum(go.x * ld16((bf16_t)(vv & 0xffff), f16) + go.y * ld16((bf16_t)(vv >> 16), f16));
        dot += p[f] * dp[f];
    }
    float dqx = 0.f, dqy = 0.f;
#pragma unroll
    for (int f = 0; f < 12; ++f) {
        const float ds = p[f] * (dp[f] - dot) * 0.08838834764831845f;
        const size_t r = ((size_t)b * N + 2 + (size_t)f * tp + t) * 1536 + h * 128;
        const unsigned kk = reinterpret_cast<const unsigned*>(kv + r)[lane];
        dqx = fmaf(ds, ld16((bf16_t)(kk & 0xffff), f16), dqx);
        dqy = fmaf(ds, ld16((bf16_t)(kk >> 16), f16), dqy);
        reinterpret_cast<unsigned*>(dkv + r)[lane] = pack2bf(ds * qq.x, ds * qq.y);
        reinterpret_cast<unsigned*>(dkv + r + 768)[lane] = pack2bf(p[f] * go.x, p[f] * go.y);
    }
    unsafeAtomicAdd(&dq[h * 128 + 2 * lane], dqx);
    unsafeAtomicAdd(&dq[h * 128 + 2 * lane + 1], dqy);
    if (t == 0)
        for (int i = threadIdx.x; i < 2 * 1536 / 2; i += 384) reinterpret_cast<unsigned*>(dkv + (size_t)b * N * 1536)[i] = 0;
}
extern "C" int sed_fpool_attn_bwd(const void* kv, const float* q, const float* probs, const float* dout, void* dkv, float* dq,
                                  int B, int N, int tp, int f16, hipStream_t stream) {
    (void)hipGetLastError();
    if (B <= 0 || tp <= 0 || N != 2 + 12 * tp) return SED_ERR_ARG;
    hipLaunchKernelGGL(fpool_attn_bwd_kernel, dim3(B * tp), dim3(384), 0, stream, (const bf16_t*)kv, q, probs, dout, (bf16_t*)dkv, dq, N, tp,
                       f16);
    return sed_check_launch();
}

// LoRA gradients from the gradient of the merged weight (dW [n_out, k_in] fp32):  dB += s dW A^T  [n_out, r],
// dA += s B^T dW  [r, k_in]   (r <= 16).  grid.y 0: one wave per row of dW (dB); grid.y 1: one thread per column (dA), 32-row slabs.
__global__ __launch_bounds__(256) void lora_grad_kernel(const float* __restrict__ dW, const float* __restrict__ A,
                                                        const float* __restrict__ Bm, float s, float* __restrict__ dA,
                                                        float* __restrict__ dB, int n_out, int k_in, int r) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (blockIdx.y == 0) {
        // dB: one wave per row of dW; the row's loads are issued eight at a time (both halves of this kernel were latency-bound loops of
        // dependent loads: 86 us per launch in the step for 2-9 MB of dW)
        // (wide rows, k_in > 1024: the four waves of a block share ONE row, a quarter of its columns each)
        __shared__ float part[4][16];
        const bool wide = k_in > 1024;
        const int kq = wide ? ((k_in / 4 + 63) / 64) * 64 : k_in, kb = wide ? wave * kq : 0, ke = wide ? (kb + kq < k_in ? kb + kq : k_in) : k_in;
        for (int n = wide ? blockIdx.x : blockIdx.x * 4 + wave; n < n_out; n += wide ? gridDim.x : gridDim.x * 4) {
            float acc[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[j] = 0.f;
            const float* row = dW + (size_t)n * k_in;
            for (int k0 = kb; k0 < ke; k0 += 512) {
                float g[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int k = k0 + 64 * u + lane; g[u] = k < ke ? row[k] : 0.f; }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int k = k0 + 64 * u + lane;
                    if (k < ke) {
#pragma unroll
                        for (int j = 0; j < 16; ++j)
                            if (j < r) acc[j] = fmaf(g[u], A[(size_t)j * k_in + k], acc[j]);
                    }
                }
            }
            if (wide) {
#pragma unroll
                for (int j = 0; j < 16; ++j)
                    if (j < r) {
                        const float v = wave_sum(acc[j]);
                        if (lane == 0) part[wave][j] = v;
                    }
                __syncthreads();
                if (threadIdx.x < r) dB[(size_t)n * r + threadIdx.x] += s * (part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x]);
                __syncthreads();
                continue;
            }
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if (j < r) {
                    const float v = wave_sum(acc[j]);
                    if (lane == 0) dB[(size_t)n * r + j] += s * v;
                }
        }
    } else {
        // dA: block = (slab of 32 dW rows, 256 columns); a thread owns one column, walks the slab's rows eight loads at a time (coalesced
        // across the block) and adds its r partial sums with one atomic each
        const int cblocks = (k_in + 255) / 256;
        const int slab = blockIdx.x / cblocks, k = (blockIdx.x - slab * cblocks) * 256 + threadIdx.x;
        const int n0 = slab * 32;
        if (k >= k_in || n0 >= n_out) return;
        float acc[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = 0.f;
        for (int nn = n0; nn < n0 + 32; nn += 8) {
            float g[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) g[u] = nn + u < n_out ? dW[(size_t)(nn + u) * k_in + k] : 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int n = nn + u < n_out ? nn + u : n_out - 1;
#pragma unroll
                for (int j = 0; j < 16; ++j)
                    if (j < r) acc[j] = fmaf(g[u], Bm[(size_t)n * r + j], acc[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (j < r) unsafeAtomicAdd(&dA[(size_t)j * k_in + k], s * acc[j]);
    }
}
extern "C" int sed_lora_grad(const float* dW, const float* A, const float* Bm, float scaling, float* dA, float* dB, int n_out,
                             int k_in, int r, hipStream_t stream) {
    (void)hipGetLastError();
    if (n_out <= 0 || k_in <= 0 || r <= 0 || r > 16) return SED_ERR_ARG;
    const int nb_a = ((n_out + 31) / 32) * ((k_in + 255) / 256), nb_b = k_in > 1024 ? n_out : (n_out + 3) / 4;
    hipLaunchKernelGGL(lora_grad_kernel, dim3(nb_a > nb_b ? nb_a : nb_b, 2), dim3(256), 0, stream, dW, A, Bm, scaling, dA, dB, n_out, k_in, r);
    return sed_check_launch();
}

// ---------------------------------------------------------------------------------------------------
// Prototype-similarity loss of the PMAM trainer (recipes/desed/pmam/train.py:82-87, 100-106): for every selected frame
//   z_c = 2 leaky_relu_0.2(cos(logit, proto_c)) - 1,  p_c = sigmoid(z_c / T),  loss = mean over (selected frames x classes) BCE(p, y)
// protos [C, 768] row-normalised by the caller side (F.normalize of the GMM means, train.py:31), labels [B, C, T], sel [B*T] bytes.
// One wave per frame; dlogit (nullable) receives d loss / d logit for selected rows, zeros elsewhere.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void proto_bce_kernel(const float* __restrict__ logit, const float* __restrict__ protos,
                                                        const float* __restrict__ labels, const unsigned char* __restrict__ sel,
                                                        float inv_count, const int* __restrict__ n_dev, float inv_temp,
                                                        float* __restrict__ loss, float* __restrict__ dlogit, float* __restrict__ post,
                                                        int rows, int T, int C) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (n_dev != nullptr) inv_count = 1.0f / (fmaxf((float)n_dev[0], 1.0f) * (float)C);   // count produced on the device: no host sync
    float lsum = 0.f;
    for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
        float4 x[3], gacc[3];
        if (!sel[row]) {
            if (dlogit != nullptr)
                for (int i = 0; i < 3; ++i) reinterpret_cast<float4*>(dlogit + (size_t)row * 768)[lane + 64 * i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (post != nullptr && lane < C) post[(size_t)row * C + lane] = 0.f;
            continue;
        }
        float nn = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            x[i] = reinterpret_cast<const float4*>(logit + (size_t)row * 768)[lane + 64 * i];
            nn += x[i].x * x[i].x + x[i].y * x[i].y + x[i].z * x[i].z + x[i].w * x[i].w;
            gacc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const float nrm = fmaxf(sqrtf(wave_sum(nn)), 1e-12f), rn = 1.0f / nrm;
        const int b = row / T, t = row - b * T;
        float sum_gc_cos = 0.f;
        for (int c = 0; c < C; ++c) {
            float4 pr[3];
            float d = 0.f;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                pr[i] = reinterpret_cast<const float4*>(protos + (size_t)c * 768)[lane + 64 * i];
                d += x[i].x * pr[i].x + x[i].y * pr[i].y + x[i].z * pr[i].z + x[i].w * pr[i].w;
            }
            const float cs = wave_sum(d) * rn;
            const float slope = cs > 0.f ? 2.f : 0.4f;
            const float z = (cs > 0.f ? cs : 0.2f * cs) * 2.f - 1.f;
            const float p = sigmoidf_(z * inv_temp);
            const float y = labels[((size_t)b * C + c) * T + t];
            if (post != nullptr && lane == 0) post[(size_t)row * C + c] = p;
            // torch BCELoss clamps each log at -100
            const float lp = fmaxf(__logf(p), -100.f), lq = fmaxf(__logf(1.f - p), -100.f);
            lsum += -(y * lp + (1.f - y) * lq);
            const float gc = (p - y) * inv_temp * slope * inv_count;      // d loss / d cos_c
            sum_gc_cos += gc * cs;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                gacc[i].x = fmaf(gc, pr[i].x, gacc[i].x); gacc[i].y = fmaf(gc, pr[i].y, gacc[i].y);
                gacc[i].z = fmaf(gc, pr[i].z, gacc[i].z); gacc[i].w = fmaf(gc, pr[i].w, gacc[i].w);
            }
        }
        if (dlogit != nullptr) {
            // d cos_c / d x = (proto_c - cos_c * xhat) / |x|
            const float k = sum_gc_cos * rn;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                float4 o;
                o.x = (gacc[i].x - k * x[i].x) * rn; o.y = (gacc[i].y - k * x[i].y) * rn;
                o.z = (gacc[i].z - k * x[i].z) * rn; o.w = (gacc[i].w - k * x[i].w) * rn;
                reinterpret_cast<float4*>(dlogit + (size_t)row * 768)[lane + 64 * i] = o;
            }
        }
    }
    if (lane == 0 && lsum != 0.f) unsafeAtomicAdd(loss, lsum * inv_count);
}
extern "C" int sed_proto_bce(const float* logit, const float* protos, const float* labels, const uint8_t* sel, int n_selected,
                             const int* n_selected_dev, float temperature, float* loss, float* dlogit, float* post, int B, int T,
                             int C, int D, hipStream_t stream) {
    (void)hipGetLastError();
    if (B <= 0 || T <= 0 || C <= 0 || C > 64 || D != 768 || (n_selected <= 0 && n_selected_dev == nullptr) || temperature <= 0.f)
        return SED_ERR_ARG;
    const int rows = B * T;
    int blocks = cdiv(rows, 4);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(proto_bce_kernel, dim3(blocks), dim3(256), 0, stream, logit, protos, labels, sel,
                       n_selected > 0 ? 1.0f / ((float)n_selected * (float)C) : 0.f, n_selected_dev, 1.0f / temperature, loss, dlogit, post,
                       rows, T, C);
    return sed_check_launch();
}

// ---------------------------------------------------------------------------------------------------
// Transpose of a NARROW 16-bit matrix: in [R, ldin] (first C <= 64 columns valid, C % 8 == 0) -> outT bf16 [C, Rpad], rows
// R..Rpad-1 zero, optional fp32 column sums (+=).  One thread per input row: the 64 x 64 tile transpose moves 64 columns even when
// 16 hold data, which for the [3.07 M, 16] operands of the first CNN layers was 4x the traffic at a quarter of the bandwidth.
// ---------------------------------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(256) void transpose_narrow_kernel(const bf16_t* __restrict__ in, int in_f16, int R, int ldin,
                                                               bf16_t* __restrict__ outT, int Rpad, float* __restrict__ colsum) {
    __shared__ float red[4][C];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc[C];
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = 0.f;
    for (int m = blockIdx.x * 256 + threadIdx.x; m < Rpad; m += gridDim.x * 256) {
#pragma unroll
        for (int c0 = 0; c0 < C; c0 += 8) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = 0.f;
            if (m < R) {
                const uint4 u = *reinterpret_cast<const uint4*>(in + (size_t)m * ldin + c0);
                const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[2 * e] = ld16((bf16_t)(w[e] & 0xffff), in_f16);
                    v[2 * e + 1] = ld16((bf16_t)(w[e] >> 16), in_f16);
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                outT[(size_t)(c0 + e) * Rpad + m] = f2bf(v[e]);
                acc[c0 + e] += v[e];
            }
        }
    }
    if (colsum == nullptr) return;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const float s = wave_sum(acc[c]);
        if (lane == 0) red[wave][c] = s;
    }
    __syncthreads();
    if (threadIdx.x < C) unsafeAtomicAdd(&colsum[threadIdx.x], (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]));
}
extern "C" int sed_transpose_narrow(const void* in, int in_f16, int R, int C, int ldin, void* outT, int Rpad, float* colsum,
                                    hipStream_t stream) {
    (void)hipGetLastError();
    if (R <= 0 || (C != 8 && C != 16 && C != 24 && C != 32) || (ldin % 8) || ldin < C || Rpad < R) return SED_ERR_ARG;
    int blocks = cdiv(Rpad, 256);
    if (blocks > 2048) blocks = 2048;
#define SED_TN(CC) hipLaunchKernelGGL(transpose_narrow_kernel<CC>, dim3(blocks), dim3(256), 0, stream, (const bf16_t*)in, in_f16, R, ldin, \
                                      (bf16_t*)outT, Rpad, colsum)
    if (C == 8) SED_TN(8); else if (C == 16) SED_TN(16); else if (C == 24) SED_TN(24); else SED_TN(32);
#undef SED_TN
    return sed_check_launch();
}

// ---------------------------------------------------------------------------------------------------
// Batched small-tensor plumbing of the PMAM step (round 4).  The padded fp32 vectors the kernels read (in_proj bias / pos_bias_u / v with
// 32-wide heads sitting in 64-wide slots, convolution and gate biases in 128-wide tiles) are gathers of the masters, and the gradients of
// every padded image go back to the masters through the same plan: ~230 torch index_select / mul / add_ launches per step as one launch
// each way.  desc: n_desc x 8 int64.
//   sed_gather_f32      {src fp32, plan int32 [n] (image element -> source element), scale fp32 [n] or 0, dst fp32 [n], n, first block, 0, 0}
//                       dst[e] = src[plan[e]] * scale[e]
//   sed_scatter_add_f32 {gimg fp32, plan int32 [n] or 0 (identity), scale fp32 [n] or 0, gmaster fp32, n, first block, C | ld_i << 32, ld_j}
//                       image element e = (i, j) = (e / C, e % C) read at gimg[i ld_i + j ld_j];  gmaster[plan[e]] += scale[e] * value
//                       for every e with scale[e] != 0 (the padding has scale 0; a plan is injective on the rest: no atomics)
// one workgroup per 256 elements.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ const long long* small_desc(const long long* desc, int n_desc) {
    int lo = 0, hi = n_desc - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (desc[(size_t)mid * 8 + 5] <= (long long)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    return desc + (size_t)lo * 8;
}
__global__ __launch_bounds__(256) void gather_f32_kernel(const long long* __restrict__ desc, int n_desc) {
    const long long* d = small_desc(desc, n_desc);
    const float* src = reinterpret_cast<const float*>(d[0]);
    const int* plan = reinterpret_cast<const int*>(d[1]);
    const float* scale = reinterpret_cast<const float*>(d[2]);
    float* dst = reinterpret_cast<float*>(d[3]);
    const long long e = (long long)(blockIdx.x - (int)d[5]) * 256 + threadIdx.x;
    if (e >= d[4]) return;
    const float v = src[plan[e]];
    dst[e] = scale != nullptr ? v * scale[e] : v;
}
extern "C" int sed_gather_f32(const int64_t* desc, int n_desc, int total_blocks, hipStream_t stream) {
    (void)hipGetLastError();
    if (n_desc <= 0 || total_blocks <= 0) return SED_ERR_ARG;
    hipLaunchKernelGGL(gather_f32_kernel, dim3(total_blocks), dim3(256), 0, stream, (const long long*)desc, n_desc);
    return sed_check_launch();
}
__global__ __launch_bounds__(256) void scatter_add_f32_kernel(const long long* __restrict__ desc, int n_desc) {
    const long long* d = small_desc(desc, n_desc);
    const float* gimg = reinterpret_cast<const float*>(d[0]);
    const int* plan = reinterpret_cast<const int*>(d[1]);
    const float* scale = reinterpret_cast<const float*>(d[2]);
    float* gm = reinterpret_cast<float*>(d[3]);
    const long long e = (long long)(blockIdx.x - (int)d[5]) * 256 + threadIdx.x;
    if (e >= d[4]) return;
    float sc = 1.0f;
    if (scale != nullptr) {
        sc = scale[e];
        if (sc == 0.0f) return;
    }
    const int C = (int)(d[6] & 0xffffffffll);
    const long long ld_i = d[6] >> 32, ld_j = d[7];
    const long long i = e / C, j = e - i * C;
    gm[plan != nullptr ? (long long)plan[e] : e] += sc * gimg[i * ld_i + j * ld_j];
}
extern "C" int sed_scatter_add_f32(const int64_t* desc, int n_desc, int total_blocks, hipStream_t stream) {
    (void)hipGetLastError();
    if (n_desc <= 0 || total_blocks <= 0) return SED_ERR_ARG;
    hipLaunchKernelGGL(scatter_add_f32_kernel, dim3(total_blocks), dim3(256), 0, stream, (const long long*)desc, n_desc);
    return sed_check_launch();
}

// ---------------------------------------------------------------------------------------------------
// BatchNorm2d statistics -> the per-channel affines the CNN kernels use (src/models/cnn/base.py:72-75; torch BatchNorm semantics).
//   train (s1 / s2 = column sums of Y and Y^2 over the M rows, sed_colstats mode 0):  mean = s1 / M, var = max(s2 / M - mean^2, 0);
//         running_mean = (1 - mom) running_mean + mom mean,  running_var = (1 - mom) running_var + mom var M / (M - 1)
//   eval  (s1 == nullptr): mean / var = the running statistics, left untouched
//   a = gamma rstd, b = beta - mean a   (Z = Y a + b);   ah = rstd, bh = -mean rstd   (xhat = Y ah + bh, backward)
// replaces eleven single-purpose torch launches per layer.
// ---------------------------------------------------------------------------------------------------
__global__ void bn_finalize_kernel(const float* __restrict__ s1, const float* __restrict__ s2, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float* __restrict__ run_mean, float* __restrict__ run_var, float inv_m,
                                   float unbias, float mom, float keep, float eps, float* __restrict__ a, float* __restrict__ b, float* __restrict__ ah,
                                   float* __restrict__ bh, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float mean, var;
    if (s1 != nullptr) {
        mean = s1[c] * inv_m;
        var = fmaxf(s2[c] * inv_m - mean * mean, 0.f);
        run_mean[c] = run_mean[c] * keep + mom * mean;
        run_var[c] = run_var[c] * keep + mom * (var * unbias);
    } else {
        mean = run_mean[c];
        var = run_var[c];
    }
    const float rstd = 1.0f / sqrtf(var + eps);
    const float av = gamma[c] * rstd;
    a[c] = av;
    b[c] = beta[c] - mean * av;
    ah[c] = rstd;
    bh[c] = -mean * rstd;
}
extern "C" int sed_bn_finalize(const float* s1, const float* s2, const float* gamma, const float* beta, float* run_mean, float* run_var,
                               int64_t M, int C, double momentum, double eps, float* a, float* b, float* ah, float* bh, hipStream_t stream) {
    (void)hipGetLastError();
    if (C <= 0 || M <= 1 || (s1 == nullptr) != (s2 == nullptr)) return SED_ERR_ARG;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, 128)), dim3(128), 0, stream, s1, s2, gamma, beta, run_mean, run_var,
                       (float)(1.0 / (double)M), (float)((double)M / (double)(M - 1)), (float)momentum, (float)(1.0 - momentum), (float)eps, a, b,
                       ah, bh, C);
    return sed_check_launch();
}

// ---------------------------------------------------------------------------------------------------
// LoRA gradients without the gradient of the merged weight (round 4).  For y = x W^T + s (x A^T) B^T (src/models/lora/layers.py:148-151;
// A [r, in], B [out, r]) autograd gives  dB = s dy^T (x A^T),  dA = (s dy B)^T x:  four skinny products over the M tokens, each a single
// pass over x or dy, instead of the full dW = dy^T x (2 M out in flops per layer, the split-K reduce and the projection of dW):
//   sed_lora_rowproj    out[M, r] = scale * X[M, K] . Wm        (u = x A^T;  du = s dy B)       MFMA 16x16x32, X fragments straight from HBM
//   sed_lora_colreduce  G += scale * sum_m P[m, :] (x) Y[m, :]    (dB from dy, u;  dA from x, du)  lane = 4 columns, P by scalar loads
// ---------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) _Float16 pm_f16x8_t;
template <bool F16> __device__ __forceinline__ f32x4_t pm_mfma16(s16x8_t a, s16x8_t b, f32x4_t c) {
    if (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(pm_f16x8_t, a), __builtin_bit_cast(pm_f16x8_t, b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
// Wm fp32 [r, K] (w_kxr = 0) or [K, r] (1) -> 16-bit image in LDS, row j at j * (2 K + 16) bytes (the 16-byte skew spreads the 16 rows of
// a fragment read over the banks); a wave owns 16 token rows at a time: A operand = lane (row l & 15, k chunk l >> 4) 16-byte loads from X.
template <bool F16>
__global__ __launch_bounds__(256) void lora_rowproj_kernel(const bf16_t* __restrict__ X, int ldx, const float* __restrict__ Wm, int w_kxr, int M,
                                                           int K, int r, float scale, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char pm_wl[];
    const int rows_w = r <= 8 ? 8 : 16, stride = K * 2 + 16;
    // W image: batches of independent loads (a load -> convert -> ds_write chain per element made this prologue, not the stream over X,
    // the longest part of the kernel: 96 dependent round trips at K = 3072)
    if (w_kxr && r == 8) {      // Wm [K, 8]: one thread per k, its row as two 16-byte loads
        for (int k = threadIdx.x; k < K; k += 1024) {
            float4 v[4][2];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    v[q][h] = (k + 256 * q < K) ? reinterpret_cast<const float4*>(Wm + (size_t)(k + 256 * q) * 8)[h] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (k + 256 * q < K) {
                    unsigned char* d = pm_wl + (k + 256 * q) * 2;
                    *reinterpret_cast<bf16_t*>(d) = to_16<F16>(v[q][0].x);
                    *reinterpret_cast<bf16_t*>(d + stride) = to_16<F16>(v[q][0].y);
                    *reinterpret_cast<bf16_t*>(d + 2 * stride) = to_16<F16>(v[q][0].z);
                    *reinterpret_cast<bf16_t*>(d + 3 * stride) = to_16<F16>(v[q][0].w);
                    *reinterpret_cast<bf16_t*>(d + 4 * stride) = to_16<F16>(v[q][1].x);
                    *reinterpret_cast<bf16_t*>(d + 5 * stride) = to_16<F16>(v[q][1].y);
                    *reinterpret_cast<bf16_t*>(d + 6 * stride) = to_16<F16>(v[q][1].z);
                    *reinterpret_cast<bf16_t*>(d + 7 * stride) = to_16<F16>(v[q][1].w);
                }
        }
    } else if (w_kxr) {  // Wm [K, r]: one thread per k, its r values contiguous
        for (int k = threadIdx.x; k < K; k += 1024) {
            float v[4][16];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int j = 0; j < 16; ++j) v[q][j] = (j < r && k + 256 * q < K) ? Wm[(size_t)(k + 256 * q) * r + j] : 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (k + 256 * q < K)
#pragma unroll
                    for (int j = 0; j < 16; ++j)
                        if (j < rows_w) *reinterpret_cast<bf16_t*>(pm_wl + j * stride + (k + 256 * q) * 2) = to_16<F16>(v[q][j]);
        }
    } else {             // Wm [r, K]: four consecutive k per thread
        const int nq = K / 4;
        for (int j = 0; j < rows_w; ++j)
            for (int q0 = threadIdx.x; q0 < nq; q0 += 2048) {
                float4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    v[u] = (j < r && q0 + 256 * u < nq) ? reinterpret_cast<const float4*>(Wm + (size_t)j * K)[q0 + 256 * u] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (q0 + 256 * u < nq) {
                        uint2 pk;
                        pk.x = pack2<F16>(v[u].x, v[u].y);
                        pk.y = pack2<F16>(v[u].z, v[u].w);
                        *reinterpret_cast<uint2*>(pm_wl + j * stride + (q0 + 256 * u) * 8) = pk;
                    }
            }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 15, kc = lane >> 4;
    const bool has_w = n < rows_w;
    // K is walked 64 columns per pair of MFMA steps: lane group kc owns columns 16 kc .. 16 kc + 15 of them (32 contiguous bytes: the four
    // groups of a row cover one 128-byte line), the first eight in the even step, the other eight in the odd one; the W fragments follow.
    // A workgroup owns 16 token rows at a time, its four waves a quarter of K each (one tile per wave left most waves of the grid with a
    // single tile and K / 256 dependent memory round trips: 72 us for 175 MB); the partial tiles meet in LDS.
    __shared__ float part[4][16][17];
    const unsigned char* wr = pm_wl + (has_w ? n : 0) * stride + kc * 32;
    const s16x8_t zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    const int kq = ((K / 64 + 3) / 4) * 64, kb = wave * kq, ke = (kb + kq) < K ? (kb + kq) : K;
    for (int tile = blockIdx.x; tile * 16 < M; tile += gridDim.x) {
        const int row0 = tile * 16;
        int row = row0 + n;
        row = row < M ? row : M - 1;
        const bf16_t* xr = X + (size_t)row * ldx + kc * 16;
        f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
        for (int k0 = kb; k0 < ke; k0 += 512) {       // sixteen 32-wide K steps per trip, their loads issued together
            s16x8_t xa[16];
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (k0 + 64 * (u >> 1) < ke) xa[u] = *reinterpret_cast<const s16x8_t*>(xr + k0 + 64 * (u >> 1) + 8 * (u & 1));
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (k0 + 64 * (u >> 1) < ke) {
                    const s16x8_t wb = has_w ? *reinterpret_cast<const s16x8_t*>(wr + (k0 + 64 * (u >> 1) + 8 * (u & 1)) * 2) : zero8;
                    acc = pm_mfma16<F16>(xa[u], wb, acc);
                }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) part[wave][4 * kc + i][n] = acc[i];       // D[m = 4 (lane >> 4) + i][n = lane & 15]
        __syncthreads();
        {
            const int ro = threadIdx.x >> 4, col = threadIdx.x & 15;
            if (col < r && row0 + ro < M)
                out[(size_t)(row0 + ro) * r + col] = scale * ((part[0][ro][col] + part[1][ro][col]) + (part[2][ro][col] + part[3][ro][col]));
        }
        __syncthreads();
    }
}
extern "C" int sed_lora_rowproj(const void* X, int x_f16, int M, int K, int ldx, const float* Wm, int w_kxr, int r, float scale, float* out,
                                hipStream_t stream) {
    (void)hipGetLastError();
    if (M <= 0 || K <= 0 || (K % 64) || (ldx % 8) || ldx < K || r <= 0 || r > 16) return SED_ERR_ARG;
    const int lds = (r <= 8 ? 8 : 16) * (K * 2 + 16);
    if (lds > 152 * 1024) return SED_ERR_ARG;      // (4.3 KiB of static LDS beside the W image)
    int blocks = cdiv(M, 16);
    if (blocks > 768) blocks = 768;      // (three 49 KiB workgroups per CU at K = 3072: the W image is built once per resident workgroup)
    static bool attr[2] = {false, false};
    if (x_f16) {
        if (!attr[1]) { (void)hipFuncSetAttribute((const void*)lora_rowproj_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024); attr[1] = true; }
        hipLaunchKernelGGL((lora_rowproj_kernel<true>), dim3(blocks), dim3(256), lds, stream, (const bf16_t*)X, ldx, Wm, w_kxr, M, K, r, scale, out);
    } else {
        if (!attr[0]) { (void)hipFuncSetAttribute((const void*)lora_rowproj_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024); attr[0] = true; }
        hipLaunchKernelGGL((lora_rowproj_kernel<false>), dim3(blocks), dim3(256), lds, stream, (const bf16_t*)X, ldx, Wm, w_kxr, M, K, r, scale, out);
    }
    return sed_check_launch();
}
// grid (column groups of 256, token slabs); the four waves of a workgroup take every fourth token of the slab, their partial sums meet in
// LDS and leave as coalesced atomics: G [C, r] (g_cxr = 1: dB) or [r, C] (0: dA).  r <= 8.
template <bool F16>
__global__ __launch_bounds__(256) void lora_colreduce_kernel(const bf16_t* __restrict__ Y, int ldy, const float* __restrict__ P, int M, int C, int r,
                                                             float scale, float* __restrict__ G, int g_cxr, int rows_per_wg) {
    __shared__ float red[4][8][256];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c0 = blockIdx.x * 256, c = c0 + 4 * lane;
    const bool live = c < C;
    const int m_begin = blockIdx.y * rows_per_wg, m_end = (m_begin + rows_per_wg) < M ? (m_begin + rows_per_wg) : M;
    float acc[8][4];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[j][e] = 0.f;
    for (int m0 = m_begin + wave; m0 < m_end; m0 += 32) {      // eight rows of this wave per trip (m0, m0 + 4, ...): 4 KiB in flight
        uint2 y[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int m = m0 + 4 * u;
            y[u] = (live && m < m_end) ? *reinterpret_cast<const uint2*>(Y + (size_t)m * ldy + c) : make_uint2(0u, 0u);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int m = m0 + 4 * u;
            if (m >= m_end) break;       // (wave-uniform)
            const float* pr = P + (size_t)m * r;
            float yv[4];
            if (F16) {
                yv[0] = h2f((bf16_t)(y[u].x & 0xffffu)); yv[1] = h2f((bf16_t)(y[u].x >> 16));
                yv[2] = h2f((bf16_t)(y[u].y & 0xffffu)); yv[3] = h2f((bf16_t)(y[u].y >> 16));
            } else {
                yv[0] = __uint_as_float(y[u].x << 16); yv[1] = __uint_as_float(y[u].x & 0xffff0000u);
                yv[2] = __uint_as_float(y[u].y << 16); yv[3] = __uint_as_float(y[u].y & 0xffff0000u);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (j < r) {
                    const float pj = pr[j];
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[j][e] = fmaf(pj, yv[e], acc[j][e]);
                }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) red[wave][j][4 * lane + e] = acc[j][e];
    __syncthreads();
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int e = t + 256 * i;
        const int col = g_cxr ? (e >> 3) : (e & 255), j = g_cxr ? (e & 7) : (e >> 8);
        if (j < r && c0 + col < C) {
            const float v = (red[0][j][col] + red[1][j][col]) + (red[2][j][col] + red[3][j][col]);
            unsafeAtomicAdd(G + (g_cxr ? (size_t)(c0 + col) * r + j : (size_t)j * C + c0 + col), scale * v);
        }
    }
}
extern "C" int sed_lora_colreduce(const void* Y, int y_f16, int M, int C, int ldy, const float* P, int r, float scale, float* G, int g_cxr,
                                  hipStream_t stream) {
    (void)hipGetLastError();
    if (M <= 0 || C <= 0 || (C % 4) || (ldy % 4) || ldy < C || r <= 0 || r > 8) return SED_ERR_ARG;
    const int cgs = cdiv(C, 256);
    int slabs = 1024 / cgs;      // one round of four workgroups per CU
    if (slabs < 1) slabs = 1;
    int rows = cdiv(M, slabs);
    rows = (rows + 31) / 32 * 32;
    slabs = cdiv(M, rows);
    if (y_f16)
        hipLaunchKernelGGL((lora_colreduce_kernel<true>), dim3(cgs, slabs), dim3(256), 0, stream, (const bf16_t*)Y, ldy, P, M, C, r, scale, G, g_cxr, rows);
    else
        hipLaunchKernelGGL((lora_colreduce_kernel<false>), dim3(cgs, slabs), dim3(256), 0, stream, (const bf16_t*)Y, ldy, P, M, C, r, scale, G, g_cxr, rows);
    return sed_check_launch();
}

// ---------------------------------------------------------------------------------------------------
// Dropout keep-masks of the CNN branch (nn.Dropout(conv_dropout) after every ContextGating, src/models/cnn/base.py:88-89), all layers in one
// launch: mask[e] = 1 with probability 1 - p.  Counter-based generator (splitmix64 finaliser of seed + 4-element group index, 16 bits per
// element: p is resolved to 1 / 65536) -- torch's Philox stream cannot be reproduced from outside torch, and only the distribution is part
// of the model.  Replaces rand -> compare -> byte cast (three launches and 9 bytes of traffic per element, per layer).
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dropout_mask_kernel(unsigned char* __restrict__ mask, long long n4, unsigned thr, unsigned long long seed) {
    for (long long g = (long long)blockIdx.x * 256 + threadIdx.x; g < n4; g += (long long)gridDim.x * 256) {
        unsigned long long z = seed + (unsigned long long)(g + 1) * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        const unsigned lo = (unsigned)z, hi = (unsigned)(z >> 32);
        const unsigned m = ((lo & 0xffffu) >= thr ? 1u : 0u) | ((lo >> 16) >= thr ? 0x100u : 0u) | ((hi & 0xffffu) >= thr ? 0x10000u : 0u) |
                           ((hi >> 16) >= thr ? 0x1000000u : 0u);
        reinterpret_cast<unsigned*>(mask)[g] = m;
    }
}
extern "C" int sed_dropout_mask(uint8_t* mask, int64_t n, float p, int64_t seed, hipStream_t stream) {
    (void)hipGetLastError();
    if (n <= 0 || (n % 4) || p < 0.f || p >= 1.f) return SED_ERR_ARG;
    const unsigned thr = (unsigned)(p * 65536.0f + 0.5f);
    hipLaunchKernelGGL(dropout_mask_kernel, dim3(grid_for((size_t)(n / 4))), dim3(256), 0, stream, mask, (long long)(n / 4), thr,
                       (unsigned long long)seed);
    return sed_check_launch();
}

// ---------------------------------------------------------------------------------------------------
// Weight gradient of a 16- or 32-filter layer (round 4): dW[i, j] += sum_m dY[m, i] X[m, j], dbias[i] += sum_m dY[m, i] for i < NV,
// j < k <= 512, over millions of rows -- the gate / convolution gradients of the first CNN layers.  Through the 256 x 256-tile TN GEMM
// such an output costs a full tile's K loop (~1.95 us per 64 rows whatever the width: 365 us + a 64 us reduce for 3 M rows, 2.2 TB/s on
// its own operands); here it is a streaming reduction: lane = (row slot, 4 columns), the row's NV values of dY by two / four 16-byte
// loads shared by the lanes of the slot, NV x 4 packed FMAs per row and lane, partial sums through shuffles and LDS, one round of atomics.
// ---------------------------------------------------------------------------------------------------
template <int NV, bool XF16>
__global__ __launch_bounds__(256) void small_dw_kernel(const bf16_t* __restrict__ dY, int ldy, const bf16_t* __restrict__ X, int ldx, int k,
                                                       float* __restrict__ dW, int ldw, float* __restrict__ dbias, long long M, int G,
                                                       int rows_per_wg) {
    extern __shared__ float sdw_red[];        // [4 waves][NV / HALVES... ][4 G]: see the combine below
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cg = lane & (G - 1), rs = lane / G, RS = 64 / G;
    const int c0 = blockIdx.y * 4 * G + 4 * cg;
    const bool col_ok = c0 < k;              // (k is a multiple of 4)
    const long long m_begin = (long long)blockIdx.x * rows_per_wg, m_end = (m_begin + rows_per_wg) < M ? (m_begin + rows_per_wg) : M;
    float acc[NV][4], bacc[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        bacc[i] = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[i][e] = 0.f;
    }
    const int step = 4 * RS;                 // rows per wave and trip (this wave: rows m0 + rs, + RS, ...), the four waves interleaved
    for (long long m0 = m_begin + (long long)wave * step; m0 < m_end; m0 += 4 * step) {
        uint2 xr[4];
        uint4 dr[4][NV / 8];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long m = m0 + u * RS + rs;
            const bool ok = m < m_end;
            xr[u] = (ok && col_ok) ? *reinterpret_cast<const uint2*>(X + m * ldx + c0) : make_uint2(0u, 0u);
#pragma unroll
            for (int q = 0; q < NV / 8; ++q) dr[u][q] = ok ? reinterpret_cast<const uint4*>(dY + m * ldy)[q] : make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float xv[4];
            if (XF16) {
                xv[0] = h2f((bf16_t)(xr[u].x & 0xffffu)); xv[1] = h2f((bf16_t)(xr[u].x >> 16));
                xv[2] = h2f((bf16_t)(xr[u].y & 0xffffu)); xv[3] = h2f((bf16_t)(xr[u].y >> 16));
            } else {
                xv[0] = __uint_as_float(xr[u].x << 16); xv[1] = __uint_as_float(xr[u].x & 0xffff0000u);
                xv[2] = __uint_as_float(xr[u].y << 16); xv[3] = __uint_as_float(xr[u].y & 0xffff0000u);
            }
#pragma unroll
            for (int q = 0; q < NV / 8; ++q) {
                const unsigned w[4] = {dr[u][q].x, dr[u][q].y, dr[u][q].z, dr[u][q].w};
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const float d0 = __uint_as_float(w[p] << 16), d1 = __uint_as_float(w[p] & 0xffff0000u);
                    const int i = 8 * q + 2 * p;
                    bacc[i] += d0; bacc[i + 1] += d1;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc[i][e] = fmaf(d0, xv[e], acc[i][e]);
                        acc[i + 1][e] = fmaf(d1, xv[e], acc[i + 1][e]);
                    }
                }
            }
        }
    }
    // row slots of a wave -> slot 0 (lanes 0 .. G - 1)
    for (int off = G; off < 64; off <<= 1) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            bacc[i] += __shfl_xor(bacc[i], off, 64);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][e] += __shfl_xor(acc[i][e], off, 64);
        }
    }
    // the four waves of the workgroup through LDS, 8 rows of dW at a time ([4][8][4 G] floats <= 32 KiB), then one atomic per element
    const int W4 = 4 * G;
    for (int ib = 0; ib < NV; ib += 8) {
        __syncthreads();
        if (rs == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = 0.f;
#pragma unroll
                    for (int ii = 0; ii < NV; ++ii) v = (ii == ib + i) ? acc[ii][e] : v;      // (static register indexing)
                    sdw_red[(wave * 8 + i) * W4 + 4 * cg + e] = v;
                }
        }
        __syncthreads();
        for (int idx = threadIdx.x; idx < 8 * W4; idx += 256) {
            const int i = idx / W4, j = idx - i * W4;
            const int col = blockIdx.y * W4 + j;
            if (col < k) {
                const float v = (sdw_red[i * W4 + j] + sdw_red[(8 + i) * W4 + j]) + (sdw_red[(16 + i) * W4 + j] + sdw_red[(24 + i) * W4 + j]);
                unsafeAtomicAdd(dW + (size_t)(ib + i) * ldw + col, v);
            }
        }
    }
    if (dbias != nullptr && blockIdx.y == 0) {
        __syncthreads();
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < NV; ++i) sdw_red[wave * NV + i] = bacc[i];
        }
        __syncthreads();
        if (threadIdx.x < NV)
            unsafeAtomicAdd(dbias + threadIdx.x, (sdw_red[threadIdx.x] + sdw_red[NV + threadIdx.x]) + (sdw_red[2 * NV + threadIdx.x] + sdw_red[3 * NV + threadIdx.x]));
    }
}
extern "C" int sed_small_dw(const void* dY, int ldy, int n, const void* X, int x_f16, int ldx, int k, float* dW, int ldw, float* dbias,
                            int64_t M, hipStream_t stream) {
    (void)hipGetLastError();
    if ((n != 16 && n != 32) || k <= 0 || (k % 4) || k > 512 || (ldy % 8) || ldy < n || (ldx % 4) || ldx < k || ldw < k || M <= 0) return SED_ERR_ARG;
    int G = 1;
    while (G < 64 && 4 * G < k) G <<= 1;           // lanes per row: 4 columns each, a power of two
    const int cblocks = cdiv(k, 4 * G);
    int slabs = 1024 / cblocks;                    // ~ four workgroups per CU
    if (slabs < 1) slabs = 1;
    const int gran = 16 * (64 / G);                // rows per workgroup and trip
    long long rows = (M + slabs - 1) / slabs;
    rows = (rows + gran - 1) / gran * gran;
    slabs = (int)((M + rows - 1) / rows);
    const size_t lds = (size_t)32 * 4 * G * sizeof(float);
#define SED_SDW(NVV, F) hipLaunchKernelGGL((small_dw_kernel<NVV, F>), dim3(slabs, cblocks), dim3(256), lds, stream, (const bf16_t*)dY, ldy, \
                                           (const bf16_t*)X, ldx, k, dW, ldw, dbias, (long long)M, G, (int)rows)
    if (n == 16) { if (x_f16) SED_SDW(16, true); else SED_SDW(16, false); }
    else { if (x_f16) SED_SDW(32, true); else SED_SDW(32, false); }
#undef SED_SDW
    return sed_check_launch();
}
