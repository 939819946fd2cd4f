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
#include <stdlib.h>

#include "common.h"
#include "../../include/sed_hip.h"

#define NFFT 1024
#define HOP 320
#define WINLEN 800
#define WINOFF 112
#define NMEL 128
#define NBIN 513
#ifndef FR_PER_WG
#define FR_PER_WG 8      // frames per workgroup (multiple of 8).  More frames per workgroup amortise the setup but measured slower (B = 32: 100 us at 8, 119 at 16, 131 at 32, 136 at 64): the kernel is bound by barrier latency and wants many workgroups
#endif
#define FR_TILE 8        // frames per staged output tile

__global__ void zero_u32_kernel(unsigned* __restrict__ p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0u;
}

// |max| per clip: 16-byte loads, 8 independent loads in flight per lane (the scalar 4-byte version ran at 0.55 TB/s)
#define ABSMAX_PARTS 32      // partial maxima per clip (grid.x of wav_absmax_kernel); sed_logmel_fwd's scratch holds B x 32 words
__global__ __launch_bounds__(256) void wav_absmax_kernel(const float* __restrict__ wav, unsigned* __restrict__ maxbits, int L) {
    const int b = blockIdx.y;
    const float* w = wav + (size_t)b * L;
    float m = 0.f;
    const int lead = (int)((16 - ((size_t)w & 15)) & 15) >> 2;       // scalar elements before the first 16-byte boundary (odd L)
    const int L4 = (L - lead) >> 2;
    const f32x4_t* w4 = reinterpret_cast<const f32x4_t*>(w + lead);
    const int stride = gridDim.x * blockDim.x;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 7 * stride < L4; i += 8 * stride) {
        f32x4_t v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = w4[i + u * stride];      // plain loads: the log-mel kernel re-reads the clip from the caches
#pragma unroll
        for (int u = 0; u < 8; ++u) m = fmaxf(fmaxf(m, fmaxf(fabsf(v[u][0]), fabsf(v[u][1]))), fmaxf(fabsf(v[u][2]), fabsf(v[u][3])));
    }
    for (; i < L4; i += stride) {
        const f32x4_t v = w4[i];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
    }
    if (blockIdx.x == 0) {      // the unaligned head and the < 4-element tail
        if ((int)threadIdx.x < lead) m = fmaxf(m, fabsf(w[threadIdx.x]));
        const int t0 = lead + 4 * L4;
        if (t0 + (int)threadIdx.x < L) m = fmaxf(m, fabsf(w[t0 + threadIdx.x]));
    }
    // ONE atomic per workgroup: the B result words share a cache line, and device-scope atomics on one line retire at ~9 ns each
    // (measured: 10 112 atomics = 90 us for a 41 MB read)
    __shared__ float wmax[4];
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    // one word per (clip, workgroup): partial maxima, reduced by the log-mel kernel's workgroups (no zero-fill launch, no atomics: the B
    // atomic targets shared a cache line and retired at ~9 ns each)
    if (threadIdx.x == 0) maxbits[(size_t)b * ABSMAX_PARTS + blockIdx.x] = __float_as_uint(fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3])));
}

// pre-emphasised, reflect-padded signal sample n of the padded axis (n = 0 .. Ly + 1023), Ly = L - 1, from the raw samples
// (the 1 / (max + 1e-10) normalisation is applied by the caller: it commutes with the pre-emphasis up to rounding)
__device__ __forceinline__ int ypad_index(int n, int Ly) {
    int m = n - NFFT / 2;
    m = m < 0 ? -m : m;
    return m > Ly - 1 ? 2 * (Ly - 1) - m : m;
}

#define MEL_CSR_CAP 1280      // non-zero filterbank weights held in LDS (every bin lies under at most two triangles: <= 1026 + edges)
// complex helpers on packed fp32 pairs (v_pk_mul / v_pk_fma / v_pk_add): (re, im)
__device__ __forceinline__ f32x2v cmul_tw(f32x2v z, f32x2v tw, f32x2v twp) {   // z * tw, twp = (-tw.y, tw.x)
    return f32x2v{z.x, z.x} * tw + f32x2v{z.y, z.y} * twp;
}
__device__ __forceinline__ f32x2v mul_neg_i(f32x2v d) { return f32x2v{d.y, -d.x}; }

// One workgroup = FR_PER_WG frames as pairs; a pair of real frames is one 1024-point complex radix-4 Stockham FFT in LDS.  Everything a
// pair needs besides its samples lives on chip for the whole workgroup: the window taps and the 12 twiddle factors of a lane in
// registers, the non-zero filterbank weights as a CSR image in LDS; the samples of the next pair are requested before the current
// pair's FFT.  The arithmetic is packed fp32 on (re, im) pairs.
__global__ __launch_bounds__(256) void logmel_kernel(const float* __restrict__ wav, const unsigned* __restrict__ maxbits,
                                                     const float* __restrict__ window,  // [800] symmetric Hann
                                                     const float2* __restrict__ twiddle,  // [1024] exp(-2 pi i k / 1024)
                                                     const float* __restrict__ melw,      // [128, 513] dense
                                                     const int* __restrict__ mel_range,   // [128, 2] first bin, end bin
                                                     float* __restrict__ out, int L, int T, int do_log) {
    __shared__ f32x2v z[2][NFFT];
    __shared__ float pw[2][NBIN + 3];
    __shared__ float ostage[NMEL][FR_TILE];
    __shared__ float wcsr[MEL_CSR_CAP];
    __shared__ int moff[NMEL + 1];
    const int tid = threadIdx.x, b = blockIdx.y, t0 = blockIdx.x * FR_PER_WG;
    const float* w = wav + (size_t)b * L;
    float cmax = 0.f;
    for (int i = 0; i < ABSMAX_PARTS; ++i) cmax = fmaxf(cmax, __uint_as_float(maxbits[(size_t)b * ABSMAX_PARTS + i]));      // (uniform: scalar loads)
    const float rinv = 1.0f / (cmax + 1e-10f);
    const int Ly = L - 1;
    // ---- per-lane constants
    float wv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int n = tid + 256 * q;
        wv[q] = (n >= WINOFF && n < WINOFF + WINLEN) ? window[n - WINOFF] * rinv : 0.f;     // window tap x clip normalisation
    }
    f32x2v tw[4][3], twp[4][3];      // stages Ns = 4, 16, 64, 256 (the first stage has none)
#pragma unroll
    for (int st = 0; st < 4; ++st) {
        const int Ns = 4 << (2 * st), k = tid & (Ns - 1);
#pragma unroll
        for (int q = 1; q < 4; ++q) {
            const float2 t = twiddle[q * k * (256 / Ns)];
            tw[st][q - 1] = f32x2v{t.x, t.y};
            twp[st][q - 1] = f32x2v{-t.y, t.x};
        }
    }
    // ---- filterbank as CSR in LDS
    const int mm = tid & 127, which = tid >> 7;
    const int k0 = mel_range[2 * mm], k1 = mel_range[2 * mm + 1];
    {   // exclusive prefix sum of the 128 band widths: inclusive shuffle scan inside waves 0 and 1, wave 1 adds wave 0's total
        int incl = k1 - k0;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int up = __shfl_up(incl, o, 64);
            if ((tid & 63) >= o) incl += up;
        }
        if (tid == 63) moff[NMEL] = incl;          // total of the first 64 bands, parked in the last slot for a moment
        __syncthreads();
        const int base = (tid >= 64 && tid < NMEL) ? moff[NMEL] : 0;
        __syncthreads();
        if (tid < NMEL) moff[tid + 1] = incl + base;
        if (tid == 0) moff[0] = 0;
        __syncthreads();
    }
    const int off = moff[mm], nnz = moff[NMEL];
    const bool csr = nnz <= MEL_CSR_CAP;
    if (csr)
        for (int k = k0 + which; k < k1; k += 2) wcsr[off + k - k0] = melw[(size_t)mm * NBIN + k];
    // ---- samples of pair 0
    float sa[4][2], sb[4][2];        // [q][x[m], x[m + 1]] of frames a and b
    auto fetch = [&](int pair) {
        const int ta = t0 + 2 * pair, tb = ta + 1;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int n = tid + 256 * q;
            sa[q][0] = sa[q][1] = sb[q][0] = sb[q][1] = 0.f;
            if (n >= WINOFF && n < WINOFF + WINLEN) {
                if (ta < T) { const int m = ypad_index(HOP * ta + n, Ly); sa[q][0] = w[m]; sa[q][1] = w[m + 1]; }
                if (tb < T) { const int m = ypad_index(HOP * tb + n, Ly); sb[q][0] = w[m]; sb[q][1] = w[m + 1]; }
            }
        }
    };
    fetch(0);
    for (int pair = 0; pair < FR_PER_WG / 2; ++pair) {
        // windowed, pre-emphasised frames -> complex input (frame a real part, frame b imaginary part)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            z[0][tid + 256 * q] = f32x2v{wv[q] * (sa[q][1] - 0.97f * sa[q][0]), wv[q] * (sb[q][1] - 0.97f * sb[q][0])};
        __syncthreads();
        const bool last = pair + 1 == FR_PER_WG / 2 || t0 + 2 * pair + 2 >= T;      // workgroup-uniform
        if (!last) fetch(pair + 1);                          // in flight during the FFT
        int cur = 0;
#ifndef FE_ABL_NO_FFT      // (timing ablations, tools/ablate/build_variant.sh: results are not a spectrogram)
#pragma unroll
        for (int st = 0; st < 5; ++st) {
            const int Ns = 1 << (2 * st);
            const int j = tid, k = j & (Ns - 1);
            // LDS slot swizzle of the stage buffers: a stage's scattered writes (index bits: low 2 st bits from j, then q, then the
            // rest of j) put only 3-4 of a half-wave's 5 varying lane bits into the 5 bank-selecting index bits; the missing ones sit in
            // index bits >= 5 and are XORed back in.  The next stage reads linearly (j + 256 q), for which any XOR by bits >= 5 stays
            // conflict-free.  g(i) for the buffer WRITTEN by stage st: st 0: (i >> 5) & 3; st 1: ((i >> 5) & 3) << 2; st 2: ((i >> 6) & 1) << 4.
#define FFT_SWZ(ST, I) ((ST) == 0 ? ((I) ^ (((I) >> 5) & 3)) : (ST) == 1 ? ((I) ^ ((((I) >> 5) & 3) << 2)) : (ST) == 2 ? ((I) ^ ((((I) >> 6) & 1) << 4)) : (I))
            f32x2v u[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) u[q] = st == 0 ? z[cur][j + q * 256] : z[cur][FFT_SWZ(st - 1, j + q * 256)];
            if (st > 0) {
#pragma unroll
                for (int q = 1; q < 4; ++q) u[q] = cmul_tw(u[q], tw[st > 0 ? st - 1 : 0][q - 1], twp[st > 0 ? st - 1 : 0][q - 1]);
            }
            const f32x2v v0 = u[0] + u[2], v1 = u[0] - u[2], v2 = u[1] + u[3], v3 = mul_neg_i(u[1] - u[3]);
            const int j0 = ((j / Ns) * Ns * 4) + k;
            const int nxt = cur ^ 1;
            z[nxt][FFT_SWZ(st, j0)] = v0 + v2;
            z[nxt][FFT_SWZ(st, j0 + Ns)] = v1 + v3;
            z[nxt][FFT_SWZ(st, j0 + 2 * Ns)] = v0 - v2;
            z[nxt][FFT_SWZ(st, j0 + 3 * Ns)] = v1 - v3;
            __syncthreads();
            cur = nxt;
        }
#endif
        // split the two real spectra: Xa = (Z[k] + conj Z[N-k]) / 2, Xb = (Z[k] - conj Z[N-k]) / (2i); power
        for (int k = tid; k < NBIN; k += 256) {
            const int kn = (NFFT - k) & (NFFT - 1);
            const f32x2v zk = z[cur][k], zn = z[cur][kn];
            const float yr = zn.x, yi = -zn.y;
            const float ar = 0.5f * (zk.x + yr), ai = 0.5f * (zk.y + yi);
            const float dr = zk.x - yr, di = zk.y - yi;       // (Z - conj Zn)
            const float br = 0.5f * di, bi = -0.5f * dr;      // divided by 2i
            pw[0][k] = ar * ar + ai * ai;
            pw[1][k] = br * br + bi * bi;
        }
        __syncthreads();
        {
            float acc = 0.f;
#ifdef FE_ABL_NO_MEL
            acc = pw[which][k0];
#else
            if (csr) {
                const float* wr = wcsr + off;
                const float* pp = pw[which] + k0;
                const int n = k1 - k0;
                int i = 0;
                for (; i + 3 < n; i += 4) acc += wr[i] * pp[i] + wr[i + 1] * pp[i + 1] + wr[i + 2] * pp[i + 2] + wr[i + 3] * pp[i + 3];
                for (; i < n; ++i) acc += wr[i] * pp[i];
            } else {
                const float* wrow = melw + (size_t)mm * NBIN;
                for (int k = k0; k < k1; ++k) acc += wrow[k] * pw[which][k];
            }
#endif
            ostage[mm][(2 * pair + which) & (FR_TILE - 1)] = do_log ? (__logf(acc + 1e-5f) + 4.5f) / 5.0f : acc;
        }
        __syncthreads();
        // 128 x 8 tile -> global every fourth pair (and after the last one): thread (m, half) writes 4 consecutive frames.  The tile is
        // next written in the following pair's filterbank step, five barriers on.
        if ((pair & 3) == 3 || last) {
            const int m = tid >> 1, half = tid & 1, t = t0 + FR_TILE * (pair >> 2) + 4 * half;
            float* dst = out + ((size_t)b * NMEL + m) * T + t;
            if (t + 3 < T && (T & 3) == 0) {
                *reinterpret_cast<float4*>(dst) = make_float4(ostage[m][4 * half], ostage[m][4 * half + 1],
                                                              ostage[m][4 * half + 2], ostage[m][4 * half + 3]);
            } else {
                for (int i = 0; i < 4; ++i) if (t + i < T) dst[i] = ostage[m][4 * half + i];
            }
        }
        if (last) break;
    }
}

// ---------------------------------------------------------------------------------------------------
// Round 6: the same transform with ONE WAVE per frame pair and no workgroup barrier inside the FFT.  The 1024-point complex FFT of a pair
// (frame a = real part, frame b = imaginary part) is three register stages, n = 64 a + b, k = c + 16 (g + 4 h):
//   A  lane b:        Y[b][c]   = W1024^(b c)  sum_a x[64 a + b] W16^(a c)            16-point DFT in registers, lane twiddles in registers
//   B  lane (c, fq):  T[c][f][g] = W64^(f g)   sum_e Y[16 e + f][c] W4^(e g)           f = fq + 4 fi: four 4-point DFTs
//   C  lane (c, g):   X[c + 16 g + 64 h]     = sum_f T[c][f][g] W16^(f h)             16-point DFT in registers
// with two exchanges through a WAVE-PRIVATE 8.5 KB LDS buffer between them (a wave's DS operations execute in order: no barrier), the
// spectrum back through the same buffer for the conjugate-pair split, power -> filterbank (CSR weights in LDS, shared by the workgroup's
// four waves) -> one staged [128][16] output tile per workgroup.  Workgroup barriers: one after the CSR build, one before the tile goes
// out.  The round-5 kernel (radix-4 Stockham in LDS, 256 threads per pair) spent ~9 barriers per pair and measured latency-bound at every
// frames-per-workgroup setting; `logmel_kernel` above is kept as the reference of this one's test.
// ---------------------------------------------------------------------------------------------------
#ifndef FRW
#define FRW 16                 // frames per workgroup: 8 pairs, two per wave (measured at B = 32: 83.5 us at 16, 105.9 at 32, 97.5 at 64)
#endif
#define ZW 1280                // complex elements of a wave's buffer: the spectrum sits at SPOS(k) = k + 4 (k >> 4) < 1280 (exchanges need 16 ZS = 1088)
#define SPOS(K) ((K) + 4 * ((K) >> 4))      // the stage-C writes k = c + 16 g + 64 h land on distinct 8-byte bank pairs; consecutive k stay consecutive inside 16-blocks
#define OSP (NMEL + 1)         // row pitch of the transposed output tile [frame][mel]
#define FEW_LDS_BYTES (4 * ZW * 8 + FRW * OSP * 4 + MEL_CSR_CAP * 4 + (NMEL + 1) * 4)
#ifndef FEW_BOUNDS
#define FEW_BOUNDS __launch_bounds__(256, 2)      // two workgroups per CU: at most 256 registers (the compiler took 270 and one wave per SIMD when left alone: 78 -> 110 us)
#endif
#define ZS 68                  // row pitch (complex elements) of the exchange layouts: 2-way bank conflicts at worst
__device__ __forceinline__ f32x2v cmulc(f32x2v z, float cr, float ci) {      // z * (cr + i ci), compile-time constant
    return f32x2v{z.x, z.x} * f32x2v{cr, ci} + f32x2v{z.y, z.y} * f32x2v{-ci, cr};
}
__device__ __forceinline__ void dft4(f32x2v& a0, f32x2v& a1, f32x2v& a2, f32x2v& a3) {     // X[g] = sum_e a[e] (-i)^(e g)
    const f32x2v v0 = a0 + a2, v1 = a0 - a2, v2 = a1 + a3, v3 = mul_neg_i(a1 - a3);
    a0 = v0 + v2; a1 = v1 + v3; a2 = v0 - v2; a3 = v1 - v3;
}
// 16-point DFT in place; output X[c] is left at position (c >> 2) + 4 (c & 3)
__device__ __forceinline__ void dft16(f32x2v (&x)[16]) {
#pragma unroll
    for (int a0 = 0; a0 < 4; ++a0) dft4(x[a0], x[a0 + 4], x[a0 + 8], x[a0 + 12]);        // position a0 + 4 c1 = u[a0][c1]
    constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, H = 0.70710678118654752f;
    x[1 + 4] = cmulc(x[1 + 4], C1, -S1);   x[1 + 8] = cmulc(x[1 + 8], H, -H);      x[1 + 12] = cmulc(x[1 + 12], S1, -C1);      // W16^1, ^2, ^3
    x[2 + 4] = cmulc(x[2 + 4], H, -H);     x[2 + 8] = mul_neg_i(x[2 + 8]);         x[2 + 12] = cmulc(x[2 + 12], -H, -H);       // W16^2, ^4, ^6
    x[3 + 4] = cmulc(x[3 + 4], S1, -C1);   x[3 + 8] = cmulc(x[3 + 8], -H, -H);     x[3 + 12] = cmulc(x[3 + 12], -C1, S1);      // W16^3, ^6, ^9
#pragma unroll
    for (int c1 = 0; c1 < 4; ++c1) dft4(x[4 * c1], x[4 * c1 + 1], x[4 * c1 + 2], x[4 * c1 + 3]);   // position c0 + 4 c1 = X[c1 + 4 c0]
}
#define DFT16_POS(C) (((C) >> 2) + 4 * ((C) & 3))

__global__ FEW_BOUNDS void logmel_wave_kernel(const float* __restrict__ wav, const unsigned* __restrict__ maxbits,
                                                          const float* __restrict__ window, const float2* __restrict__ twiddle,
                                                          const float* __restrict__ melw, const int* __restrict__ mel_range,
                                                          float* __restrict__ out, int L, int T, int do_log) {
    // dynamic LDS (FEW_LDS_BYTES = 48.5 KB: three workgroups per CU): exchange buffers (also the power spectra), output tile, CSR weights, band offsets
    extern __shared__ __attribute__((aligned(16))) unsigned char few_lds[];
    f32x2v (*zb)[ZW] = reinterpret_cast<f32x2v (*)[ZW]>(few_lds);
    float (*ostage)[OSP] = reinterpret_cast<float (*)[OSP]>(few_lds + 4 * ZW * 8);      // [frame][mel]: a wave's 64 bands go to 64 consecutive words
    float* wcsr = reinterpret_cast<float*>(few_lds + 4 * ZW * 8 + FRW * OSP * 4);
    int* moff = reinterpret_cast<int*>(wcsr + MEL_CSR_CAP);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.y, t0 = blockIdx.x * FRW;
    const float* w = wav + (size_t)b * L;
    float cmax = 0.f;
    for (int i = 0; i < ABSMAX_PARTS; ++i) cmax = fmaxf(cmax, __uint_as_float(maxbits[(size_t)b * ABSMAX_PARTS + i]));      // (uniform: scalar loads)
    const float rinv = 1.0f / (cmax + 1e-10f);
    const int Ly = L - 1;
    // ---- filterbank as CSR in LDS (as in logmel_kernel)
    {
        const int mm = tid & 127, which = tid >> 7;
        const int k0 = mel_range[2 * mm], k1 = mel_range[2 * mm + 1];
        int incl = k1 - k0;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int up = __shfl_up(incl, o, 64);
            if ((tid & 63) >= o) incl += up;
        }
        if (tid == 63) moff[NMEL] = incl;
        __syncthreads();
        const int base = (tid >= 64 && tid < NMEL) ? moff[NMEL] : 0;
        __syncthreads();
        if (tid < NMEL) moff[tid + 1] = incl + base;
        if (tid == 0) moff[0] = 0;
        __syncthreads();
        if (moff[NMEL] <= MEL_CSR_CAP) {
            const int off = moff[mm];
            for (int k = k0 + which; k < k1; k += 2) wcsr[off + k - k0] = melw[(size_t)mm * NBIN + k];
        }
    }
    const bool csr = moff[NMEL] <= MEL_CSR_CAP;
    // ---- per-lane constants: window taps of n = 64 a + lane, stage-A twiddles W1024^(lane c), stage-B twiddles W64^(f g)
    float wv[16];
#pragma unroll
    for (int a = 0; a < 16; ++a) {
        const int n = 64 * a + lane;
        wv[a] = (n >= WINOFF && n < WINOFF + WINLEN) ? window[n - WINOFF] * rinv : 0.f;
    }
    f32x2v twa[15], twa_p[15];
#pragma unroll
    for (int c = 1; c < 16; ++c) {
        const float2 t = twiddle[lane * c];
        twa[c - 1] = f32x2v{t.x, t.y};
        twa_p[c - 1] = f32x2v{-t.y, t.x};
    }
    const int cB = lane >> 2, fq = lane & 3;      // stage B: lane = (c, fq); stage C: lane = (c, g) with g = fq
    f32x2v twb[4][3], twb_p[4][3];
#pragma unroll
    for (int fi = 0; fi < 4; ++fi)
#pragma unroll
        for (int g = 1; g < 4; ++g) {
            const float2 t = twiddle[16 * (fq + 4 * fi) * g];
            twb[fi][g - 1] = f32x2v{t.x, t.y};
            twb_p[fi][g - 1] = f32x2v{-t.y, t.x};
        }
    // the two bands of this lane (mel = lane, lane + 64)
    int mk0[2], mk1[2], mo[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        mk0[i] = mel_range[2 * (lane + 64 * i)];
        mk1[i] = mel_range[2 * (lane + 64 * i) + 1];
    }
    __syncthreads();      // CSR image complete
#pragma unroll
    for (int i = 0; i < 2; ++i) mo[i] = moff[lane + 64 * i];
    f32x2v* z = zb[wave];
    for (int pi = 0; pi < FRW / 8; ++pi) {
        const int pair = wave + 4 * pi, ta = t0 + 2 * pair, tb = ta + 1;
        if (ta >= T) break;                       // (wave-uniform)
        // ---- windowed, pre-emphasised samples: x[a] = (frame a, frame b) at n = 64 a + lane
        f32x2v x[16];
        x[0] = f32x2v{0.f, 0.f};
        x[15] = f32x2v{0.f, 0.f};
        const int base = HOP * ta - NFFT / 2;       // sample index of n = 0 of frame a on the un-padded axis
#ifdef FEW_ABL_NOSAMPLES      // (timing ablations, tools/ablate/build_variant.sh: results are not a spectrogram)
#pragma unroll
        for (int a = 1; a < 15; ++a) x[a] = f32x2v{wv[a], wv[a]};
#else
        if (base + 64 >= 0 && base + HOP + 15 * 64 <= Ly - 1 && tb < T) {
            // interior pair (all but the first two and last two frames of a clip): frame b is frame a moved by HOP = 5 x 64 samples, and
            // y[n] = w[n + 1] - 0.97 w[n] takes its second operand from the next lane -- 19 coalesced row loads per lane instead of 112 scalar
            // ones with reflected indices (the loads were 41 of the kernel's 96 us)
            float raw[20];
#pragma unroll
            for (int r = 1; r < 20; ++r) raw[r] = w[base + 64 * r + lane];
            float d[20];
#pragma unroll
            for (int r = 1; r < 20; ++r) {
                float nx = __shfl_down(raw[r], 1, 64);
                const float first_next = r < 19 ? __shfl(raw[r < 19 ? r + 1 : r], 0, 64) : 0.f;      // (row 19 is used by lanes < 16 only: taps beyond n = 911 are zero)
                nx = lane == 63 ? first_next : nx;
                d[r] = nx - 0.97f * raw[r];
            }
#pragma unroll
            for (int a = 1; a < 15; ++a) x[a] = f32x2v{wv[a] * d[a], wv[a] * d[a + 5]};
        } else {
#pragma unroll
            for (int a = 1; a < 15; ++a) {
                const int n = 64 * a + lane;
                float va = 0.f, vb = 0.f;
                if (n >= WINOFF && n < WINOFF + WINLEN) {
                    const int ma = ypad_index(HOP * ta + n, Ly);
                    va = w[ma + 1] - 0.97f * w[ma];
                    if (tb < T) { const int mb = ypad_index(HOP * tb + n, Ly); vb = w[mb + 1] - 0.97f * w[mb]; }
                }
                x[a] = f32x2v{wv[a] * va, wv[a] * vb};
            }
        }
#endif
#ifndef FEW_ABL_NOFFT
        // ---- stage A
        dft16(x);
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            f32x2v y = x[DFT16_POS(c)];
            if (c > 0) y = cmul_tw(y, twa[c - 1], twa_p[c - 1]);
            z[ZS * c + lane] = y;                                           // Y[b = lane][c]
        }
        __builtin_amdgcn_wave_barrier();
        // ---- stage B: lane (c, fq), f = fq + 4 fi: 4-point DFT over e of Y[16 e + f][c], then W64^(f g)
        f32x2v tq[16];
#pragma unroll
        for (int fi = 0; fi < 4; ++fi) {
            const int f = fq + 4 * fi;
            f32x2v e0 = z[ZS * cB + f], e1 = z[ZS * cB + 16 + f], e2 = z[ZS * cB + 32 + f], e3 = z[ZS * cB + 48 + f];
            dft4(e0, e1, e2, e3);
            tq[4 * fi] = e0;
            tq[4 * fi + 1] = cmul_tw(e1, twb[fi][0], twb_p[fi][0]);
            tq[4 * fi + 2] = cmul_tw(e2, twb[fi][1], twb_p[fi][1]);
            tq[4 * fi + 3] = cmul_tw(e3, twb[fi][2], twb_p[fi][2]);
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int fi = 0; fi < 4; ++fi)
#pragma unroll
            for (int g = 0; g < 4; ++g) z[ZS * cB + 17 * g + fq + 4 * fi] = tq[4 * fi + g];      // T[c][f][g] at c ZS + 17 g + f
        __builtin_amdgcn_wave_barrier();
        // ---- stage C: lane (c, g): 16-point DFT over f
#pragma unroll
        for (int f = 0; f < 16; ++f) x[f] = z[ZS * cB + 17 * fq + f];
        __builtin_amdgcn_wave_barrier();
        dft16(x);
#pragma unroll
        for (int h = 0; h < 16; ++h) z[SPOS(cB + 16 * fq + 64 * h)] = x[DFT16_POS(h)];           // X[k], k = c + 16 g + 64 h, at SPOS(k)
        __builtin_amdgcn_wave_barrier();
#else
#pragma unroll
        for (int h = 0; h < 16; ++h) z[lane + 64 * h] = x[h];
        __builtin_amdgcn_wave_barrier();
#endif
        // ---- split the two real spectra and take the power: bins lane + 64 i (i = 0 .. 7) and 512.  IN PLACE: z[k] becomes (|Xa[k]|^2,
        // |Xb[k]|^2) -- an iteration reads z[k] and z[1024 - k] >= 513 (never written) before it writes z[k], and a wave's DS operations
        // execute in order; the filterbank then gets both frames' powers of a bin from one 8-byte read
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int k = lane + 64 * i;
            if (i < 8 || lane == 0) {
                const f32x2v zk = z[SPOS(k)], zn = z[SPOS((NFFT - k) & (NFFT - 1))];
                const float yr = zn.x, yi = -zn.y;
                const float ar = 0.5f * (zk.x + yr), ai = 0.5f * (zk.y + yi);
                const float dr = zk.x - yr, di = zk.y - yi;
                const float br = 0.5f * di, bi = -0.5f * dr;
                z[SPOS(k)] = f32x2v{ar * ar + ai * ai, br * br + bi * bi};
            }
        }
        __builtin_amdgcn_wave_barrier();
        // ---- filterbank: bands lane and lane + 64 of both frames
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float acc0 = 0.f, acc1 = 0.f;
            const int k0 = mk0[i], n = mk1[i] - k0;
#ifdef FEW_ABL_NOMEL
            if (true) { acc0 = z[SPOS(k0)].x; acc1 = z[SPOS(k0)].y; } else
#endif
            if (csr) {
                // (a float4 form over rows padded to whole groups of four bins measured SLOWER: 113.8 against 83.1 us per call at B = 32)
                const float* wr = wcsr + mo[i];
                f32x2v acc = {0.f, 0.f};
                for (int j = 0; j < n; ++j) { const float wgt = wr[j]; acc = f32x2v{wgt, wgt} * z[SPOS(k0 + j)] + acc; }
                acc0 = acc.x; acc1 = acc.y;
            } else {
                const float* wrow = melw + (size_t)(lane + 64 * i) * NBIN + k0;
                for (int j = 0; j < n; ++j) { const float wgt = wrow[j]; acc0 = fmaf(wgt, z[SPOS(k0 + j)].x, acc0); acc1 = fmaf(wgt, z[SPOS(k0 + j)].y, acc1); }
            }
            ostage[2 * pair][lane + 64 * i] = do_log ? (__logf(acc0 + 1e-5f) + 4.5f) / 5.0f : acc0;
            ostage[2 * pair + 1][lane + 64 * i] = do_log ? (__logf(acc1 + 1e-5f) + 4.5f) / 5.0f : acc1;
        }
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    // ---- [FRW][128] tile -> global: thread (m, half) writes FRW / 2 consecutive frames of mel row m
    {
        const int m = tid >> 1, half = tid & 1, t = t0 + (FRW / 2) * half;
        float* dst = out + ((size_t)b * NMEL + m) * T + t;
#pragma unroll
        for (int i = 0; i < FRW / 2; i += 4) {
            const int f = (FRW / 2) * half + i;
            if (t + i + 3 < T && (T & 3) == 0)
                *reinterpret_cast<float4*>(dst + i) = make_float4(ostage[f][m], ostage[f + 1][m], ostage[f + 2][m], ostage[f + 3][m]);
            else
                for (int j = 0; j < 4; ++j) if (t + i + j < T) dst[i + j] = ostage[f + j][m];
        }
    }
}

extern "C" int sed_logmel_fwd(const float* wav, float* out, uint32_t* maxbits_tmp, const float* window,
                              const float* twiddle, const float* melw, const int* mel_range, int B, int L, int T,
                              int do_log, hipStream_t stream) {
    (void)hipGetLastError();
    if (B <= 0 || T != 1 + (L - 1) / HOP || L < NFFT) return SED_ERR_ARG;
    // |max| per clip as ABSMAX_PARTS partial maxima (every slot is written: workgroups past the clip's end write 0)
    hipLaunchKernelGGL(wav_absmax_kernel, dim3(ABSMAX_PARTS, B), dim3(256), 0, stream, wav, maxbits_tmp, L);
    // do_log bit 1 (test aid): the round-5 kernel, four waves per frame pair with the FFT in LDS -- the reference the wave-per-pair kernel
    // is checked against bit for bit in structure (same window, pre-emphasis, split, filterbank order per band)
    if (do_log & 2)
        hipLaunchKernelGGL(logmel_kernel, dim3(cdiv(T, FR_PER_WG), B), dim3(256), 0, stream, wav, maxbits_tmp, window,
                           (const float2*)twiddle, melw, mel_range, out, L, T, do_log & 1);
    else
    {
        static bool attr = false;
        if (!attr) { (void)hipFuncSetAttribute((const void*)logmel_wave_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, FEW_LDS_BYTES); attr = true; }
        hipLaunchKernelGGL(logmel_wave_kernel, dim3(cdiv(T, FRW), B), dim3(256), FEW_LDS_BYTES, stream, wav, maxbits_tmp, window,
                           (const float2*)twiddle, melw, mel_range, out, L, T, do_log & 1);
    }
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
// The augmentation branches no shipped config turns on (choice[1], choice[2], time_mask), HBM-bound like the rest:
//  mask_box : x[b, f0:f1, t0:t1] = value in place            (time_mask data_aug.py:93-108; FrequencyMasking :136-140)
//  add_noise: out = x + noise * std_b(x) / snr_b            (data_aug.py:195-204; std over (F, T) per clip, unbiased like torch.std)
// ---------------------------------------------------------------------------------------------------
__global__ void mask_box_kernel(float* __restrict__ x, int B, int F, int T, int f0, int f1, int t0, int t1, float value) {
    const int wf = f1 - f0, wt = t1 - t0;
    const size_t total = (size_t)B * wf * wt;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int t = (int)(idx % wt);
        const size_t r = idx / wt;
        const int f = (int)(r % wf), b = (int)(r / wf);
        x[((size_t)b * F + f0 + f) * T + t0 + t] = value;
    }
}
extern "C" int sed_mask_box(float* x, int B, int F, int T, int f0, int f1, int t0, int t1, float value, hipStream_t stream) {
    (void)hipGetLastError();
    if (f0 < 0 || f1 > F || t0 < 0 || t1 > T) return SED_ERR_ARG;
    if (f1 <= f0 || t1 <= t0 || B <= 0) return SED_OK;      // an empty slice, like the reference's python slicing
    const size_t total = (size_t)B * (f1 - f0) * (t1 - t0);
    hipLaunchKernelGGL(mask_box_kernel, dim3((unsigned)min((size_t)2048, (total + 255) / 256)), dim3(256), 0, stream, x, B, F, T, f0, f1,
                       t0, t1, value);
    return sed_check_launch();
}

#define NOISE_CHUNKS 64
// per (clip, chunk): count, mean and M2 = sum (x - mean)^2 of the chunk (two passes over registers: no cancellation)
__global__ __launch_bounds__(256) void clip_stats_kernel(const float* __restrict__ x, float* __restrict__ part, int n) {
    const int b = blockIdx.y, c = blockIdx.x;
    const int per = (n + NOISE_CHUNKS - 1) / NOISE_CHUNKS;
    const int lo = c * per, hi = min(n, lo + per);
    const float* xb = x + (size_t)b * n;
    __shared__ float red[8];
    float s = 0.f;
    for (int i = lo + threadIdx.x; i < hi; i += 256) s += xb[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    const int cnt = max(hi - lo, 0);
    const float mean = cnt > 0 ? (red[0] + red[1] + red[2] + red[3]) / (float)cnt : 0.f;
    float m2 = 0.f;
    for (int i = lo + threadIdx.x; i < hi; i += 256) { const float d = xb[i] - mean; m2 += d * d; }
    m2 = wave_sum(m2);
    if ((threadIdx.x & 63) == 0) red[4 + (threadIdx.x >> 6)] = m2;
    __syncthreads();
    if (threadIdx.x == 0) {
        float* o = part + ((size_t)b * NOISE_CHUNKS + c) * 3;
        o[0] = (float)cnt; o[1] = mean; o[2] = red[4] + red[5] + red[6] + red[7];
    }
}
__global__ __launch_bounds__(256) void add_noise_kernel(const float* __restrict__ x, const float* __restrict__ noise,
                                                        const float* __restrict__ snr_lin, const float* __restrict__ part,
                                                        float* __restrict__ out, int n) {
    const int b = blockIdx.y;
    // Chan's pairwise combination of the chunk statistics, in double (64 terms; every thread does it: no barrier)
    double cn = 0.0, cm = 0.0, cM2 = 0.0;
    for (int c = 0; c < NOISE_CHUNKS; ++c) {
        const float* p = part + ((size_t)b * NOISE_CHUNKS + c) * 3;
        const double k = p[0], mu = p[1], m2 = p[2];
        if (k > 0.0) {
            const double d = mu - cm, tot = cn + k;
            cM2 += m2 + d * d * cn * k / tot;
            cm += d * k / tot;
            cn = tot;
        }
    }
    const float sigma = (float)sqrt(cM2 / (cn - 1.0)) / snr_lin[b];
    const size_t base = (size_t)b * n;
    const int n4 = n / 4;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n4; i += gridDim.x * 256) {
        const float4 a = reinterpret_cast<const float4*>(x + base)[i], z = reinterpret_cast<const float4*>(noise + base)[i];
        float4 o;
        o.x = a.x + z.x * sigma; o.y = a.y + z.y * sigma; o.z = a.z + z.z * sigma; o.w = a.w + z.w * sigma;
        reinterpret_cast<float4*>(out + base)[i] = o;
    }
}
extern "C" int sed_add_noise(const float* x, const float* noise, const float* snr_lin, float* part, float* out, int B, int n,
                             hipStream_t stream) {
    (void)hipGetLastError();
    if (n % 4 || n < 2 || B <= 0) return SED_ERR_ARG;
    hipLaunchKernelGGL(clip_stats_kernel, dim3(NOISE_CHUNKS, B), dim3(256), 0, stream, x, part, n);
    hipLaunchKernelGGL(add_noise_kernel, dim3(64, B), dim3(256), 0, stream, x, noise, snr_lin, part, out, n);
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
// The same median (modes 0 and 1) for long windows -- the evaluation path filters with 32 / 128 frames (train.py:221-227: median_window
// / 156 * 1000), where the rank search above is O(k^2) per output (4.5 ms per launch, 7 % of a validation batch).  Order statistics
// on GLOBAL ranks instead: the padded column (T + k - 1 entries, padding entries kept as entries of their own, so every window is a
// contiguous run of distinct entries) gets a total order once (rank = entries smaller, ties by position: n^2 / 256 compares per
// thread); a window is then a bitmap over ranks, its median the (k/2 + 1)-th set bit, and sliding the window by one output clears one bit
// and sets one.  The result is the same element the rank search returns (an input value, never an arithmetic combination): bit-exact.
#define MEDR_MAXN 2048
__global__ __launch_bounds__(256) void median_filter_rank_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                                 const int* __restrict__ sizes, const float* __restrict__ scale,
                                                                 int T, int C, int mode, int W) {
    extern __shared__ unsigned char medr_lds[];
    float* vp = reinterpret_cast<float*>(medr_lds);                                  // [MEDR_MAXN] padded column
    float* byrank = vp + MEDR_MAXN;                                                   // [MEDR_MAXN] value of rank r
    unsigned short* rk = reinterpret_cast<unsigned short*>(byrank + MEDR_MAXN);      // [MEDR_MAXN] rank of padded entry p
    unsigned* bm = reinterpret_cast<unsigned*>(rk + MEDR_MAXN);                      // [256][W] per-thread window bitmaps (W odd)
    const int b = blockIdx.x / C, c = blockIdx.x - b * C, tid = threadIdx.x;
    const float sc = scale != nullptr ? scale[b * C + c] : 1.0f;
    int k = sizes[c];
    if (mode == 0 && (k & 1) == 0) k += 1;
    const int lo = k / 2, rank = k / 2, n = T + k - 1;
    if (n > MEDR_MAXN || n > 32 * W) __builtin_trap();      // a window larger than the caller's bound: fail loudly, never silently
    for (int p = tid; p < n; p += 256) {
        int s_ = p - lo;
        if (mode == 0) s_ = s_ < 0 ? 0 : (s_ >= T ? T - 1 : s_);
        else s_ = s_ < 0 ? -s_ - 1 : (s_ >= T ? 2 * T - s_ - 1 : s_);
        const float v = in[((size_t)b * T + s_) * C + c];
        vp[p] = scale != nullptr ? v * sc : v;
    }
    __syncthreads();
    for (int p = tid; p < n; p += 256) {
        const float v = vp[p];
        int r = 0;
        for (int q = 0; q < n; ++q) {
            const float u = vp[q];
            r += (u < v) || (u == v && q < p);
        }
        rk[p] = (unsigned short)r;
        byrank[r] = v;
    }
    __syncthreads();
    const int chunk = (T + 255) / 256, i0 = tid * chunk;
    if (i0 >= T) return;
    unsigned* my = bm + tid * W;
    for (int w = 0; w < W; ++w) my[w] = 0u;
    for (int j = 0; j < k; ++j) { const int r = rk[i0 + j]; my[r >> 5] |= 1u << (r & 31); }
    const int i1 = i0 + chunk < T ? i0 + chunk : T;
    for (int i = i0; i < i1; ++i) {
        int cum = 0, w = 0;
        unsigned x = my[0];
        int cnt = __popc(x);
        while (cum + cnt <= rank) { cum += cnt; x = my[++w]; cnt = __popc(x); }
        for (int t = rank - cum; t > 0; --t) x &= x - 1;          // drop the lower set bits: the wanted one becomes the lowest
        out[((size_t)b * T + i) * C + c] = byrank[32 * w + __ffs(x) - 1];
        if (i + 1 < i1) {
            const int r0 = rk[i], r1 = rk[i + k];
            my[r0 >> 5] &= ~(1u << (r0 & 31));
            my[r1 >> 5] |= 1u << (r1 & 31);
        }
    }
}
static int median_filter_impl(const float* in, float* out, const int* sizes, const float* scale, int B, int T, int C, int mode,
                              int max_size, hipStream_t stream) {
    (void)hipGetLastError();
    if (mode < 0 || mode > 2 || T > 12288 || max_size < 0) return SED_ERR_ARG;
    // max_size: the caller's bound on the window sizes (they live on the device); 0 = unknown -> the rank-search kernel
    if (mode != 2 && max_size >= 24 && T >= 256 && T + max_size + 1 <= MEDR_MAXN) {
        int W = (T + max_size + 1 + 31) / 32;
        W |= 1;                                     // odd row stride: the 64 lanes' bitmap words fall into different banks
        const size_t lds = (size_t)MEDR_MAXN * (4 + 4 + 2) + (size_t)256 * W * 4;
        static size_t attr = 0;
        if (lds > attr) { (void)hipFuncSetAttribute((const void*)median_filter_rank_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = lds; }
        hipLaunchKernelGGL(median_filter_rank_kernel, dim3(B * C), dim3(256), lds, stream, in, out, sizes, scale, T, C, mode, W);
        return sed_check_launch();
    }
    hipLaunchKernelGGL(median_filter_kernel, dim3(B * C), dim3(256), T * sizeof(float), stream, in, out, sizes, scale, T,
                       C, mode);
    return sed_check_launch();
}
extern "C" int sed_median_filter(const float* in, float* out, const int* sizes, const float* scale, int B, int T, int C,
                                 int mode, hipStream_t stream) {
    return median_filter_impl(in, out, sizes, scale, B, T, C, mode, 0, stream);
}
// the same filter with the caller's bound on the (device-resident) window sizes: long windows take the rank kernel
extern "C" int sed_median_filter_k(const float* in, float* out, const int* sizes, const float* scale, int B, int T, int C,
                                   int mode, int max_size, hipStream_t stream) {
    return median_filter_impl(in, out, sizes, scale, B, T, C, mode, max_size, stream);
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

// The same filter straight from 16-bit PCM file bodies (the input pipeline's form, data.WavBatchStream): x [B, L] int16 as read from
// the RIFF data chunks (zero padded to L by the host), lens[b] = samples the file really holds.  The int16 -> fp32 scaling by 2^-15
// (libsndfile's convention, what the reference's reader yields) happens in the tap loop, outputs at or past ceil(lens[b] up / down) are
// zero -- the offline tool resamples the file first and pad_wav zero-pads the result afterwards, so the filter's ringing past the
// file's end never reaches the model.  H2D traffic is the int16 16 kHz body: a quarter of the fp32 32 kHz clip.
__global__ __launch_bounds__(256) void resample_poly_pcm16_kernel(const short* __restrict__ x, const int* __restrict__ lens,
                                                                  float* __restrict__ y, const float* __restrict__ h, int L, int Lout,
                                                                  int up, int down, int ntaps, int n_pre_pad, int n_pre_remove) {
    extern __shared__ float taps[];
    for (int i = threadIdx.x; i < ntaps; i += blockDim.x) taps[i] = h[i] * (1.0f / 32768.0f);
    __syncthreads();
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= Lout) return;
    int len = lens[blockIdx.y];
    len = len < L ? len : L;
    const long long nvalid = ((long long)len * up + down - 1) / down;
    float acc = 0.f;
    if (n < nvalid) {
        const short* xb = x + (size_t)blockIdx.y * L;
        const long long c = (long long)(n + n_pre_remove) * down - n_pre_pad;
        long long m_hi = c / up;
        if (m_hi > len - 1) m_hi = len - 1;
        long long m_lo = (c - (ntaps - 1) + up - 1) / up;
        if (c - (ntaps - 1) < 0) m_lo = 0;
        if (m_lo < 0) m_lo = 0;
        for (long long m = m_lo; m <= m_hi; ++m) acc += (float)xb[m] * taps[(int)(c - m * up)];
    }
    y[(size_t)blockIdx.y * Lout + n] = acc;
}
extern "C" int sed_resample_poly_pcm16(const int16_t* x, const int* lens, float* y, const float* h, int B, int L, int Lout, int up,
                                       int down, int ntaps, int n_pre_pad, int n_pre_remove, hipStream_t stream) {
    (void)hipGetLastError();
    if (B <= 0 || L <= 0 || Lout <= 0 || up < 1 || down < 1 || ntaps < 1 || ntaps > 8192) return SED_ERR_ARG;
    hipLaunchKernelGGL(resample_poly_pcm16_kernel, dim3(cdiv(Lout, 256), B), dim3(256), ntaps * sizeof(float), stream,
                       (const short*)x, lens, y, h, L, Lout, up, down, ntaps, n_pre_pad, n_pre_remove);
    return sed_check_launch();
}
