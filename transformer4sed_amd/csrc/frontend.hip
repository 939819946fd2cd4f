// Log-mel frontend, spectrogram augmentation and median-filter post-processing (HBM-bound kernels), gfx950.
//
// Frontend replaces PasstFeatureExtractor (src/models/passt/passt_feature_extraction.py:46-94): per-clip max-abs
// normalisation, pre-emphasis, centred/reflect-padded STFT (n_fft 1024, hop 320, symmetric Hann 800), power,
// Kaldi mel filterbank, log, affine -- fused into ONE pass over the waveform: no [B,513,1000,2] complex tensor and
// no power spectrogram ever reach HBM (the reference writes 6 MB/clip of intermediates).  Each workgroup
// produces 8 frames as 4 pairs; a pair of real frames is transformed with one 1024-point complex radix-4
// Stockham FFT in LDS (frame a -> real part, frame b -> imaginary part), the mel filterbank is applied from a
// banded table (each triangle touches a contiguous bin range), and the 128x8 output tile is written as 16-byte
// rows.  Algorithmic HBM bytes: 1.28 MB read + 0.512 MB written per clip.
#include "common.h"
#include "../../include/sed_hip.h"

#define NFFT 1024
#define HOP 320
#define WINLEN 800
#define WINOFF 112
#define NMEL 128
#define NBIN 513
#define FR_PER_WG 8

__global__ void zero_u32_kernel(unsigned* __restrict__ p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0u;
}

__global__ void wav_absmax_kernel(const float* __restrict__ wav, unsigned* __restrict__ maxbits, int L) {
    const int b = blockIdx.y;
    const float* w = wav + (size_t)b * L;
    float m = 0.f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < L; i += gridDim.x * blockDim.x) m = fmaxf(m, fabsf(w[i]));
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) atomicMax(&maxbits[b], __float_as_uint(m));  // non-negative floats order as uints
}

// pre-emphasised, reflect-padded signal sample n of the padded axis (n = 0 .. Ly + 1023), Ly = L - 1
__device__ __forceinline__ float ypad_at(const float* __restrict__ w, int n, int Ly, float inv) {
    int m = n - NFFT / 2;
    m = m < 0 ? -m : m;
    m = m > Ly - 1 ? 2 * (Ly - 1) - m : m;
    return w[m + 1] / inv - 0.97f * (w[m] / inv);
}

__global__ __launch_bounds__(256) void logmel_kernel(const float* __restrict__ wav, const unsigned* __restrict__ maxbits,
                                                     const float* __restrict__ window,  // [800] symmetric Hann
                                                     const float2* __restrict__ twiddle,  // [1024] exp(-2 pi i k / 1024)
                                                     const float* __restrict__ melw,      // [128, 513] dense
                                                     const int* __restrict__ mel_range,   // [128, 2] first bin, end bin
                                                     float* __restrict__ out, int L, int T, int do_log) {
    __shared__ float xr[2][NFFT], xi[2][NFFT];
    __shared__ float pw[2][NBIN + 3];
    __shared__ float ostage[NMEL][FR_PER_WG];
    const int tid = threadIdx.x, b = blockIdx.y, t0 = blockIdx.x * FR_PER_WG;
    const float* w = wav + (size_t)b * L;
    const float denom = __uint_as_float(maxbits[b]) + 1e-10f;
    const int Ly = L - 1;
    for (int pair = 0; pair < FR_PER_WG / 2; ++pair) {
        const int ta = t0 + 2 * pair, tb = ta + 1;
        // windowed frames -> complex input (zero outside the 800-sample window support)
        for (int n = tid; n < NFFT; n += 256) {
            float a = 0.f, c = 0.f;
            if (n >= WINOFF && n < WINOFF + WINLEN) {
                const float wn = window[n - WINOFF];
                if (ta < T) a = wn * ypad_at(w, HOP * ta + n, Ly, denom);
                if (tb < T) c = wn * ypad_at(w, HOP * tb + n, Ly, denom);
            }
            xr[0][n] = a;
            xi[0][n] = c;
        }
        __syncthreads();
        int cur = 0;
#pragma unroll
        for (int Ns = 1; Ns < NFFT; Ns *= 4) {
            const int j = tid, k = j & (Ns - 1);
            float ur[4], ui[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float re = xr[cur][j + q * 256], im = xi[cur][j + q * 256];
                if (q == 0 || Ns == 1) { ur[q] = re; ui[q] = im; }
                else {
                    const float2 tw = twiddle[q * k * (256 / Ns)];
                    ur[q] = re * tw.x - im * tw.y;
                    ui[q] = re * tw.y + im * tw.x;
                }
            }
            const float v0r = ur[0] + ur[2], v0i = ui[0] + ui[2], v1r = ur[0] - ur[2], v1i = ui[0] - ui[2];
            const float v2r = ur[1] + ur[3], v2i = ui[1] + ui[3];
            const float v3r = ui[1] - ui[3], v3i = -(ur[1] - ur[3]);  // (u1 - u3) * (-i)
            const int j0 = ((j / Ns) * Ns * 4) + k;
            const int nxt = cur ^ 1;
            xr[nxt][j0] = v0r + v2r;          xi[nxt][j0] = v0i + v2i;
            xr[nxt][j0 + Ns] = v1r + v3r;     xi[nxt][j0 + Ns] = v1i + v3i;
            xr[nxt][j0 + 2 * Ns] = v0r - v2r; xi[nxt][j0 + 2 * Ns] = v0i - v2i;
            xr[nxt][j0 + 3 * Ns] = v1r - v3r; xi[nxt][j0 + 3 * Ns] = v1i - v3i;
            __syncthreads();
            cur = nxt;
        }
        // split the two real spectra: Xa = (Z[k] + conj Z[N-k]) / 2, Xb = (Z[k] - conj Z[N-k]) / (2i); power
        for (int k = tid; k < NBIN; k += 256) {
            const int kn = (NFFT - k) & (NFFT - 1);
            const float zr = xr[cur][k], zi = xi[cur][k], yr = xr[cur][kn], yi = -xi[cur][kn];
            const float ar = 0.5f * (zr + yr), ai = 0.5f * (zi + yi);
            const float dr = zr - yr, di = zi - yi;           // (Z - conj Zn)
            const float br = 0.5f * di, bi = -0.5f * dr;      // divided by 2i
            pw[0][k] = ar * ar + ai * ai;
            pw[1][k] = br * br + bi * bi;
        }
        __syncthreads();
        {
            const int m = tid & 127, which = tid >> 7;
            const int k0 = mel_range[2 * m], k1 = mel_range[2 * m + 1];
            const float* wrow = melw + (size_t)m * NBIN;
            float acc = 0.f;
            for (int k = k0; k < k1; ++k) acc += wrow[k] * pw[which][k];
            ostage[m][2 * pair + which] = do_log ? (__logf(acc + 1e-5f) + 4.5f) / 5.0f : acc;
        }
        __syncthreads();
    }
    // 128 x 8 tile -> global: thread (m, half) writes 4 consecutive frames
    {
        const int m = tid >> 1, half = tid & 1, t = t0 + 4 * half;
        float* dst = out + ((size_t)b * NMEL + m) * T + t;
        if (t + 3 < T && (T & 3) == 0) {
            *reinterpret_cast<float4*>(dst) = make_float4(ostage[m][4 * half], ostage[m][4 * half + 1],
                                                          ostage[m][4 * half + 2], ostage[m][4 * half + 3]);
        } else {
            for (int i = 0; i < 4; ++i) if (t + i < T) dst[i] = ostage[m][4 * half + i];
        }
    }
}

extern "C" int sed_logmel_fwd(const float* wav, float* out, uint32_t* maxbits_tmp, const float* window,
                              const float* twiddle, const float* melw, const int* mel_range, int B, int L, int T,
                              int do_log, hipStream_t stream) {
    (void)hipGetLastError();
    if (B <= 0 || T != 1 + (L - 1) / HOP || L < NFFT) return SED_ERR_ARG;
    hipLaunchKernelGGL(zero_u32_kernel, dim3(cdiv(B, 256)), dim3(256), 0, stream, maxbits_tmp, B);
    hipLaunchKernelGGL(wav_absmax_kernel, dim3(64, B), dim3(256), 0, stream, wav, maxbits_tmp, L);
    hipLaunchKernelGGL(logmel_kernel, dim3(cdiv(T, FR_PER_WG), B), dim3(256), 0, stream, wav, maxbits_tmp, window,
                       (const float2*)twiddle, melw, mel_range, out, L, T, do_log);
    return sed_check_launch();
}

// ---------------------------------------------------------------------------------------------------
// Augmentation (src/preprocess/data_aug.py), all random draws are inputs:
//  roll_mix : out[b,f,t] = c_b in[b,f,(t - s_b) mod T] + (1-c_b) in[p_b,f,(t - s_{p_b}) mod T]   (frame_shift + mixup)
//  warp_filt: out[b,i,t] = in[b,k_i,t] + lam_i (in[b,k_i+1,t] - in[b,k_i,t]) + add[b,i]            (freq_nonlinear + filt_aug)
// ---------------------------------------------------------------------------------------------------
__global__ void roll_mix_kernel(const float* __restrict__ in, float* __restrict__ out, const int* __restrict__ shift,
                                const int* __restrict__ perm, const float* __restrict__ cmix, int B, int F, int T,
                                int clamp01) {
    const size_t total = (size_t)B * F * T;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int t = (int)(idx % T);
        const size_t bf = idx / T;
        const int f = (int)(bf % F), b = (int)(bf / F);
        int ts = (t - shift[b]) % T;
        ts = ts < 0 ? ts + T : ts;
        float v = in[((size_t)b * F + f) * T + ts];
        if (perm != nullptr && cmix != nullptr) {
            const int p = perm[b];
            const float c = cmix[2 * b], c1 = cmix[2 * b + 1];
            if (p != b || c != 1.0f) {
                int tp = (t - shift[p]) % T;
                tp = tp < 0 ? tp + T : tp;
                v = c * v + c1 * in[((size_t)p * F + f) * T + tp];
            }
        }
        if (clamp01) v = fminf(fmaxf(v, 0.f), 1.f);
        out[idx] = v;
    }
}
extern "C" int sed_roll_mix(const float* in, float* out, const int* shift, const int* perm, const float* cmix, int B,
                            int F, int T, int clamp01, hipStream_t stream) {
    (void)hipGetLastError();
    hipLaunchKernelGGL(roll_mix_kernel, dim3(2048), dim3(256), 0, stream, in, out, shift, perm, cmix, B, F, T, clamp01);
    return sed_check_launch();
}
__global__ void warp_filt_kernel(const float* __restrict__ in, float* __restrict__ out, const int* __restrict__ kidx,
                                 const float* __restrict__ lam, const float* __restrict__ add, int B, int F, int T) {
    const int T4 = T / 4;
    const size_t total = (size_t)B * F * T4;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int t4 = (int)(idx % T4);
        const size_t bf = idx / T4;
        const int i = (int)(bf % F), b = (int)(bf / F);
        float4 a, c;
        float l = 0.f;
        if (kidx != nullptr) {
            const int k = kidx[i];
            l = lam[i];
            a = reinterpret_cast<const float4*>(in + ((size_t)b * F + k) * T)[t4];
            c = reinterpret_cast<const float4*>(in + ((size_t)b * F + k + 1) * T)[t4];
        } else {
            a = reinterpret_cast<const float4*>(in + ((size_t)b * F + i) * T)[t4];
            c = a;
        }
        const float ad = add != nullptr ? add[b * F + i] : 0.f;
        float4 o;
        o.x = a.x + l * (c.x - a.x) + ad; o.y = a.y + l * (c.y - a.y) + ad;
        o.z = a.z + l * (c.z - a.z) + ad; o.w = a.w + l * (c.w - a.w) + ad;
        reinterpret_cast<float4*>(out + ((size_t)b * F + i) * T)[t4] = o;
    }
}
extern "C" int sed_warp_filt(const float* in, float* out, const int* kidx, const float* lam, const float* add, int B,
                             int F, int T, hipStream_t stream) {
    (void)hipGetLastError();
    if (T % 4) return SED_ERR_ARG;
    hipLaunchKernelGGL(warp_filt_kernel, dim3(2048), dim3(256), 0, stream, in, out, kidx, lam, add, B, F, T);
    return sed_check_launch();
}

// ---------------------------------------------------------------------------------------------------
// Median / max filter along time per class, both reference semantics, bit-exact (compare/select only).
//   mode 0: src/postprocess/filter.py:4-36   even size -> size+1, replicate padding, true median
//   mode 1: scipy.ndimage.median_filter as called at src/codec/decoder.py:91: window [i - k/2, i - k/2 + k),
//           symmetric (edge-inclusive reflect) padding, element of rank k/2
//   mode 2: scipy.ndimage.maximum_filter (decoder.py:94), same window/padding
// in/out [B, T, C]; one workgroup per (b, c).  Optional per-(b,c) multiplier applied first (soft weak mask,
// decoder.py:80) or hard zeroing (decoder.py:22-23) via `scale` [B, C].
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void median_filter_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                            const int* __restrict__ sizes, const float* __restrict__ scale,
                                                            int T, int C, int mode) {
    extern __shared__ float col[];  // [T]
    const int b = blockIdx.x / C, c = blockIdx.x - b * C;
    const float sc = scale != nullptr ? scale[b * C + c] : 1.0f;
    for (int t = threadIdx.x; t < T; t += 256) {
        const float v = in[((size_t)b * T + t) * C + c];
        col[t] = scale != nullptr ? v * sc : v;
    }
    __syncthreads();
    int k = sizes[c];
    if (mode == 0 && (k & 1) == 0) k += 1;
    const int lo = k / 2, rank = k / 2;
    for (int i = threadIdx.x; i < T; i += 256) {
        float res = 0.f;
        if (mode == 2) {
            float m = -INFINITY;
            for (int j = 0; j < k; ++j) {
                int idx = i - lo + j;
                idx = idx < 0 ? -idx - 1 : (idx >= T ? 2 * T - idx - 1 : idx);
                m = fmaxf(m, col[idx]);
            }
            res = m;
        } else {
            for (int a = 0; a < k; ++a) {
                int ia = i - lo + a;
                if (mode == 0) ia = ia < 0 ? 0 : (ia >= T ? T - 1 : ia);
                else ia = ia < 0 ? -ia - 1 : (ia >= T ? 2 * T - ia - 1 : ia);
                const float va = col[ia];
                int lt = 0, le = 0;
                for (int j = 0; j < k; ++j) {
                    int ij = i - lo + j;
                    if (mode == 0) ij = ij < 0 ? 0 : (ij >= T ? T - 1 : ij);
                    else ij = ij < 0 ? -ij - 1 : (ij >= T ? 2 * T - ij - 1 : ij);
                    const float vj = col[ij];
                    lt += vj < va;
                    le += vj <= va;
                }
                if (lt <= rank && rank < le) { res = va; break; }
            }
        }
        out[((size_t)b * T + i) * C + c] = res;
    }
}
extern "C" int sed_median_filter(const float* in, float* out, const int* sizes, const float* scale, int B, int T, int C,
                                 int mode, hipStream_t stream) {
    (void)hipGetLastError();
    if (mode < 0 || mode > 2 || T > 12288) return SED_ERR_ARG;
    hipLaunchKernelGGL(median_filter_kernel, dim3(B * C), dim3(256), T * sizeof(float), stream, in, out, sizes, scale, T,
                       C, mode);
    return sed_check_launch();
}

// ---------------------------------------------------------------------------------------------------
// Polyphase FIR resampler (src/utils/resample.py:10-14 does this offline with librosa; definition here = scipy.signal.resample_poly):
//   y[n] = sum_m x[m] h[(n + n_pre_remove) * down - m * up - n_pre_pad],   h = Kaiser-windowed sinc taps * up (host-built)
// One thread per output sample; taps in LDS.  HBM-bound: 4 B read (reused ~taps/up times from cache) + 4 B written per sample.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void resample_poly_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                            const float* __restrict__ h, int L, int Lout, int up, int down,
                                                            int ntaps, int n_pre_pad, int n_pre_remove) {
    extern __shared__ float taps[];
    for (int i = threadIdx.x; i < ntaps; i += blockDim.x) taps[i] = h[i];
    __syncthreads();
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= Lout) return;
    const float* xb = x + (size_t)blockIdx.y * L;
    const long long c = (long long)(n + n_pre_remove) * down - n_pre_pad;   // tap index k = c - m * up must lie in [0, ntaps)
    long long m_hi = c / up;                                                  // k >= 0
    if (m_hi > L - 1) m_hi = L - 1;
    long long m_lo = (c - (ntaps - 1) + up - 1) / up;                         // k <= ntaps - 1
    if (c - (ntaps - 1) < 0) m_lo = 0;
    if (m_lo < 0) m_lo = 0;
    float acc = 0.f;
    for (long long m = m_lo; m <= m_hi; ++m) acc += xb[m] * taps[(int)(c - m * up)];
    y[(size_t)blockIdx.y * Lout + n] = acc;
}
extern "C" int sed_resample_poly(const float* x, float* y, const float* h, int B, int L, int Lout, int up, int down, int ntaps,
                                 int n_pre_pad, int n_pre_remove, hipStream_t stream) {
    (void)hipGetLastError();
    if (B <= 0 || L <= 0 || Lout <= 0 || up < 1 || down < 1 || ntaps < 1 || ntaps > 8192) return SED_ERR_ARG;
    hipLaunchKernelGGL(resample_poly_kernel, dim3(cdiv(Lout, 256), B), dim3(256), ntaps * sizeof(float), stream, x, y, h, L, Lout,
                       up, down, ntaps, n_pre_pad, n_pre_remove);
    return sed_check_launch();
}
