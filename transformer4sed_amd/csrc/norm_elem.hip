// HBM-bound kernels of the MAT-SED model path: LayerNorm fwd/bwd, patch im2col / token assembly, frequency
// pooling + x10 linear interpolation (+ sliding-window merge), classifier/pooling heads, attention-pooling
// AT head, MLM masking and loss, fused AdamW + EMA.  All fp32 math; bf16 only as GEMM-operand outputs.
// Wave64: one wavefront per 768-wide row, 12 channels per lane held as 3 x float4, shuffle reductions.
#include <stdlib.h>
#include "common.h"
#include "../../include/sed_hip.h"

extern "C" int sed_abi_version(int) { return SED_HIP_ABI_VERSION; }

#define DM 768
#define NV 3  // float4 per lane per row

struct Row { float4 v[NV]; };

__device__ __forceinline__ void row_load(Row& r, const float* p, int lane) {
#pragma unroll
    for (int i = 0; i < NV; ++i) r.v[i] = reinterpret_cast<const float4*>(p)[lane + 64 * i];
}
__device__ __forceinline__ void row_store(const Row& r, float* p, int lane) {
#pragma unroll
    for (int i = 0; i < NV; ++i) reinterpret_cast<float4*>(p)[lane + 64 * i] = r.v[i];
}
__device__ __forceinline__ void row_store_bf16(const Row& r, bf16_t* p, int lane, int f16) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        uint2 pk;
        pk.x = f16 ? pack2<true>(r.v[i].x, r.v[i].y) : pack2bf(r.v[i].x, r.v[i].y);
        pk.y = f16 ? pack2<true>(r.v[i].z, r.v[i].w) : pack2bf(r.v[i].z, r.v[i].w);
        reinterpret_cast<uint2*>(p)[lane + 64 * i] = pk;
    }
}
// Streaming variants: the activation rows these kernels walk are read once and written once; non-temporal accesses keep them from
// displacing each other in L2 (LayerNorm forward, cold input: 59.7 -> 33.7 us at M = 38080, 2.9 -> 5.2 TB/s; tools/ablate/ln_lab.hip).
typedef float f32x4nt __attribute__((ext_vector_type(4)));
typedef unsigned u32x2nt __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void row_load_nt(Row& r, const float* p, int lane) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const f32x4nt v = __builtin_nontemporal_load(reinterpret_cast<const f32x4nt*>(p) + lane + 64 * i);
        r.v[i] = make_float4(v[0], v[1], v[2], v[3]);
    }
}
__device__ __forceinline__ void row_store_nt(const Row& r, float* p, int lane) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const f32x4nt v = {r.v[i].x, r.v[i].y, r.v[i].z, r.v[i].w};
        __builtin_nontemporal_store(v, reinterpret_cast<f32x4nt*>(p) + lane + 64 * i);
    }
}
__device__ __forceinline__ void row_store_bf16_nt(const Row& r, bf16_t* p, int lane, int f16) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        u32x2nt pk;
        pk[0] = f16 ? pack2<true>(r.v[i].x, r.v[i].y) : pack2bf(r.v[i].x, r.v[i].y);
        pk[1] = f16 ? pack2<true>(r.v[i].z, r.v[i].w) : pack2bf(r.v[i].z, r.v[i].w);
        __builtin_nontemporal_store(pk, reinterpret_cast<u32x2nt*>(p) + lane + 64 * i);
    }
}
// split-precision operand image of a row: [hi | lo | hi] f16 over 3 * DM columns (what sed_split3_f16 makes in a pass of its own)
__device__ __forceinline__ void row_store_split3_nt(const Row& r, bf16_t* p, int lane) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const float v[4] = {r.v[i].x, r.v[i].y, r.v[i].z, r.v[i].w};
        bf16_t h[4], l[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { h[e] = f2h(v[e]); l[e] = f2h(v[e] - h2f(h[e])); }
        u32x2nt hi, lo;
        hi[0] = (unsigned)h[0] | ((unsigned)h[1] << 16); hi[1] = (unsigned)h[2] | ((unsigned)h[3] << 16);
        lo[0] = (unsigned)l[0] | ((unsigned)l[1] << 16); lo[1] = (unsigned)l[2] | ((unsigned)l[3] << 16);
        u32x2nt* q = reinterpret_cast<u32x2nt*>(p) + lane + 64 * i;
        __builtin_nontemporal_store(hi, q);
        __builtin_nontemporal_store(lo, q + DM / 4);
        __builtin_nontemporal_store(hi, q + 2 * (DM / 4));
    }
}
// rows [DM f16 | DM e4m3] (pitch 3 DM / 2 halfs): the A operand of the two-term GEMMs with the lo product on the fp8 path (gemm.hip
// GemmArgs.k8); the e4m3 half is 2^-2 x the f16-ROUNDED value (what sed_fp8_tail makes from the f16 half in a pass of its own)
__device__ __forceinline__ void row_store_f16_e4m3_nt(const Row& r, bf16_t* p, int lane) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const float v[4] = {r.v[i].x, r.v[i].y, r.v[i].z, r.v[i].w};
        bf16_t h[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) h[e] = f2h(v[e]);
        u32x2nt hi;
        hi[0] = (unsigned)h[0] | ((unsigned)h[1] << 16); hi[1] = (unsigned)h[2] | ((unsigned)h[3] << 16);
        __builtin_nontemporal_store(hi, reinterpret_cast<u32x2nt*>(p) + lane + 64 * i);
        __builtin_nontemporal_store(e4m3x4_of_h4(hi[0], hi[1]), reinterpret_cast<unsigned*>(p + DM) + lane + 64 * i);
    }
}
#define ROW_FOREACH(i, c) for (int i = 0; i < NV; ++i) for (int c = 0; c < 4; ++c)
__device__ __forceinline__ float& f4(float4& v, int c) { return reinterpret_cast<float*>(&v)[c]; }
__device__ __forceinline__ const float& f4(const float4& v, int c) { return reinterpret_cast<const float*>(&v)[c]; }

// ---------------------------------------------------------------------------------------------------
// LayerNorm   y = LN(in_scale * x) * gamma + beta      (passt.py:361-362,580; timm Block norms; out_norm)
// ---------------------------------------------------------------------------------------------------
// two rows per wave: both rows' loads are in flight before the first reduction starts (the kernel is a latency chain
// load -> 2 wave reductions -> store).  A persistent grid-stride variant (gamma / beta fetched once per wave) looked 2x faster in
// a warm-cache loop but is slower on cold input (70 vs 56 us at M = 38080, tools/ln_bench.py) -- in the step the input was just
// written with non-temporal stores, so the one-shot grid below is what ships.
template <int RW>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float eps, float in_scale,
                                                            bf16_t* __restrict__ y16, float* __restrict__ y32,
                                                            float* __restrict__ mean, float* __restrict__ rstd, int M, int f16,
                                                            bf16_t* __restrict__ y16b) {
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RW;
    if (row0 >= M) return;
    Row r[RW], g, b;
#pragma unroll
    for (int k = 0; k < RW; ++k) row_load_nt(r[k], x + (size_t)(row0 + k < M ? row0 + k : row0) * DM, lane);
    row_load(g, gamma, lane);
    row_load(b, beta, lane);
    float s[RW], mu[RW], q[RW], rs[RW];
#pragma unroll
    for (int k = 0; k < RW; ++k) {
        s[k] = 0.f;
#pragma unroll
        ROW_FOREACH(i, c) { f4(r[k].v[i], c) *= in_scale; s[k] += f4(r[k].v[i], c); }
    }
#pragma unroll
    for (int k = 0; k < RW; ++k) mu[k] = wave_sum(s[k]) * (1.0f / DM);
#pragma unroll
    for (int k = 0; k < RW; ++k) {
        q[k] = 0.f;
#pragma unroll
        ROW_FOREACH(i, c) { const float d = f4(r[k].v[i], c) - mu[k]; q[k] += d * d; }
    }
#pragma unroll
    for (int k = 0; k < RW; ++k) rs[k] = rsqrtf(wave_sum(q[k]) * (1.0f / DM) + eps);
#pragma unroll
    for (int k = 0; k < RW; ++k) {
        const int row = row0 + k;
        if (row >= M) break;
#pragma unroll
        ROW_FOREACH(i, c) f4(r[k].v[i], c) = (f4(r[k].v[i], c) - mu[k]) * rs[k] * f4(g.v[i], c) + f4(b.v[i], c);
        if (y16 != nullptr) {
            if (f16 == 4) row_store_split3_nt(r[k], y16 + (size_t)row * 3 * DM, lane);
            else if (f16 == 8) row_store_f16_e4m3_nt(r[k], y16 + (size_t)row * (DM + DM / 2), lane);
            else row_store_bf16_nt(r[k], y16 + (size_t)row * DM, lane, f16);
        }
        if (y16b != nullptr) row_store_bf16_nt(r[k], y16b + (size_t)row * DM, lane, 0);      // bf16 copy for the backward's weight gradient
        if (y32 != nullptr) row_store_nt(r[k], y32 + (size_t)row * DM, lane);
        if (lane == 0 && mean != nullptr) { mean[row] = mu[k]; rstd[row] = rs[k]; }
    }
}

extern "C" int sed_layernorm_fwd(const float* x, const float* gamma, const float* beta, float eps, float in_scale,
                                 void* y_bf16, float* y_f32, float* mean, float* rstd, int M, int D, int f16, hipStream_t stream) {
    (void)hipGetLastError();
    if (D != DM || M <= 0) return SED_ERR_ARG;
    // (two rows per wave: the 1- and 4-row variants measured slower on cold input, tools/ln_bench.py)
    hipLaunchKernelGGL(layernorm_fwd_kernel<2>, dim3(cdiv(M, 8)), dim3(256), 0, stream, x, gamma, beta, eps, in_scale,
                       (bf16_t*)y_bf16, y_f32, mean, rstd, M, f16, (bf16_t*)nullptr);
    return sed_check_launch();
}
// ... writing the result twice: IEEE half (the forward GEMM's operand) and bf16 (what the backward's weight-gradient GEMM multiplies with:
// its TN kernel otherwise converts the saved f16 fragments to bf16 in registers, 17-20 % of that launch -- tools/dw_xtype_bench.py)
extern "C" int sed_layernorm_fwd_dual(const float* x, const float* gamma, const float* beta, float eps, float in_scale,
                                      void* y_f16, void* y_bf16, float* mean, float* rstd, int M, int D, hipStream_t stream) {
    (void)hipGetLastError();
    if (D != DM || M <= 0 || y_f16 == nullptr || y_bf16 == nullptr) return SED_ERR_ARG;
    hipLaunchKernelGGL(layernorm_fwd_kernel<2>, dim3(cdiv(M, 8)), dim3(256), 0, stream, x, gamma, beta, eps, in_scale,
                       (bf16_t*)y_f16, (float*)nullptr, mean, rstd, M, 1, (bf16_t*)y_bf16);
    return sed_check_launch();
}

// dx = in_scale * rstd * (dyg - mean(dyg) - xhat * mean(dyg * xhat)),  dyg = dy * gamma.
// dx is ADDED into dx_acc when accumulate != 0 (residual-stream gradient), else stored.
// dgamma/dbeta: per-wave register partials over a grid-stride row loop -> LDS block reduce -> atomicAdd.
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma, float in_scale,
                                                            float* __restrict__ dx_acc, int accumulate,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta, int M,
                                                            float dy_scale, bf16_t* __restrict__ dx16 = nullptr) {
    __shared__ float red[4][DM];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    Row g, pg, pb;
    row_load(g, gamma, lane);
#pragma unroll
    ROW_FOREACH(i, c) { f4(pg.v[i], c) = 0.f; f4(pb.v[i], c) = 0.f; }
    for (int row = blockIdx.x * 4 + wave; row < M; row += gridDim.x * 4) {
        Row d, xr;
        row_load_nt(d, dy + (size_t)row * DM, lane);
        row_load_nt(xr, x + (size_t)row * DM, lane);
        const float mu = mean[row], rs = rstd[row];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        ROW_FOREACH(i, c) {
            const float xh = (f4(xr.v[i], c) * in_scale - mu) * rs;
            const float dyv = f4(d.v[i], c) * dy_scale;
            f4(pg.v[i], c) += dyv * xh;
            f4(pb.v[i], c) += dyv;
            const float dg = dyv * f4(g.v[i], c);
            f4(d.v[i], c) = dg;
            f4(xr.v[i], c) = xh;
            s1 += dg;
            s2 += dg * xh;
        }
        s1 = wave_sum(s1) * (1.0f / DM);
        s2 = wave_sum(s2) * (1.0f / DM);
        float* out = dx_acc + (size_t)row * DM;
        Row o;
        if (accumulate) row_load_nt(o, out, lane);
#pragma unroll
        ROW_FOREACH(i, c) {
            const float v = in_scale * rs * (f4(d.v[i], c) - s1 - f4(xr.v[i], c) * s2);
            f4(o.v[i], c) = accumulate ? f4(o.v[i], c) + v : v;
        }
        row_store_nt(o, out, lane);
        // bf16 image of the updated residual-stream gradient: the dY operand of the weight-gradient / dX GEMMs of the block below
        if (dx16 != nullptr) row_store_bf16(o, dx16 + (size_t)row * DM, lane, 0);
    }
    if (dgamma == nullptr) return;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const Row& p = pass == 0 ? pg : pb;
        row_store(p, red[wave], lane);
        __syncthreads();
        for (int c = threadIdx.x; c < DM; c += 256) {
            const float s = red[0][c] + red[1][c] + red[2][c] + red[3][c];
            unsafeAtomicAdd(pass == 0 ? &dgamma[c] : &dbeta[c], s);
        }
        __syncthreads();
    }
}

extern "C" int sed_layernorm_bwd(const float* dy, const float* x, const float* mean, const float* rstd,
                                 const float* gamma, float in_scale, float* dx, int accumulate, float* dgamma,
                                 float* dbeta, int M, int D, hipStream_t stream) {
    (void)hipGetLastError();
    if (D != DM || M <= 0) return SED_ERR_ARG;
    int blocks = cdiv(M, 4);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(blocks), dim3(256), 0, stream, dy, x, mean, rstd, gamma, in_scale, dx,
                       accumulate, dgamma, dbeta, M, 1.0f, (bf16_t*)nullptr);
    return sed_check_launch();
}
// the same backward that also writes dx16 [M, D] = bf16 image of the dx it leaves in `dx` (after the accumulation)
extern "C" int sed_layernorm_bwd_x16(const float* dy, const float* x, const float* mean, const float* rstd,
                                     const float* gamma, float in_scale, float* dx, int accumulate, float* dgamma,
                                     float* dbeta, void* dx16, int M, int D, hipStream_t stream) {
    (void)hipGetLastError();
    if (D != DM || M <= 0 || dx16 == nullptr) return SED_ERR_ARG;
    int blocks = cdiv(M, 4);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(blocks), dim3(256), 0, stream, dy, x, mean, rstd, gamma, in_scale, dx,
                       accumulate, dgamma, dbeta, M, 1.0f, (bf16_t*)dx16);
    return sed_check_launch();
}

// ---------------------------------------------------------------------------------------------------
// Patch embedding plumbing (passt.py:302-315, 503-569): im2col for the 16x16/stride-10 conv, token assembly.
// ---------------------------------------------------------------------------------------------------
// mel [B, 128, T] fp32 -> cols bf16 [B * 12 * tp, 256];  row (b, f, t), col 16 i + j = mel[b, 10 f + i, tstart + 10 t + j]
// (tstart selects a sliding-window slab without copying it)
__global__ void im2col_kernel(const float* __restrict__ mel, bf16_t* __restrict__ cols, int B, int T, int tstart,
                              int tp, int f16) {
    const size_t total = (size_t)B * 12 * tp * 128;  // pairs
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int pr = (int)(idx & 127);
        const size_t row = idx >> 7;
        const int t = (int)(row % tp), f = (int)((row / tp) % 12), b = (int)(row / ((size_t)tp * 12));
        const int i = pr >> 3, j = (pr & 7) * 2;
        const float* src = mel + ((size_t)b * 128 + 10 * f + i) * T + tstart + 10 * t + j;
        reinterpret_cast<unsigned*>(cols)[idx] = f16 ? pack2<true>(src[0], src[1]) : pack2bf(src[0], src[1]);
    }
}
extern "C" int sed_im2col(const float* mel, void* cols, int B, int T, int tstart, int tp, int f16, hipStream_t stream) {
    (void)hipGetLastError();
    if (tp < 1 || tstart < 0 || tstart + 10 * (tp - 1) + 16 > T) return SED_ERR_ARG;
    hipLaunchKernelGGL(im2col_kernel, dim3(2048), dim3(256), 0, stream, mel, (bf16_t*)cols, B, T, tstart, tp, f16);
    return sed_check_launch();
}
// d mel is never needed (the mel input is data).

// conv [B * 12 * tp, D] (bias already added) -> x [B, 2 + 12 tp, D] with the three positional tables.
// freq_pe [D, 12], time_pe [D, 99] in the reference's checkpoint layout ([1,D,12,1] / [1,D,1,99]).
// One workgroup per token position n: the position's additive row (cls / dist + new_pos, or time_pe + freq_pe -- a strided gather
// from the checkpoint layout) is fetched once into registers and reused for all B clips; rows move as float4.
__global__ __launch_bounds__(256) void assemble_tokens_kernel(const float* __restrict__ conv, const float* __restrict__ cls,
                                                              const float* __restrict__ dist, const float* __restrict__ new_pos,
                                                              const float* __restrict__ freq_pe, const float* __restrict__ time_pe,
                                                              int toffset, float* __restrict__ x, int B, int tp) {
    const int N = 2 + 12 * tp, n = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    Row pe;
    if (n < 2) {
        Row a, c;
        row_load(a, n == 0 ? cls : dist, lane);
        row_load(c, new_pos + n * DM, lane);
#pragma unroll
        ROW_FOREACH(i, k) f4(pe.v[i], k) = f4(a.v[i], k) + f4(c.v[i], k);
        for (int bi = wave; bi < B; bi += 4) row_store(pe, x + ((size_t)bi * N + n) * DM, lane);
        return;
    }
    const int p = n - 2, f = p / tp, t = p - f * tp;
#pragma unroll
    ROW_FOREACH(i, k) {
        const int d = 4 * (lane + 64 * i) + k;
        f4(pe.v[i], k) = time_pe[d * 99 + toffset + t] + freq_pe[d * 12 + f];
    }
    for (int bi = wave; bi < B; bi += 4) {
        Row r;
        row_load(r, conv + ((size_t)bi * 12 * tp + p) * DM, lane);
#pragma unroll
        ROW_FOREACH(i, k) f4(r.v[i], k) += f4(pe.v[i], k);
        row_store(r, x + ((size_t)bi * N + n) * DM, lane);
    }
}
extern "C" int sed_assemble_tokens(const float* conv, const float* cls, const float* dist, const float* new_pos,
                                   const float* freq_pe, const float* time_pe, int toffset, float* x, int B, int tp,
                                   hipStream_t stream) {
    (void)hipGetLastError();
    if (tp < 1 || toffset < 0 || toffset + tp > 99) return SED_ERR_ARG;
    hipLaunchKernelGGL(assemble_tokens_kernel, dim3(2 + 12 * tp), dim3(256), 0, stream, conv, cls, dist, new_pos, freq_pe,
                       time_pe, toffset, x, B, tp);
    return sed_check_launch();
}
// backward: dx [B, N, D] -> dconv bf16 rows (GEMM operand), d cls/dist/new_pos, d freq_pe, d time_pe (atomics)
__global__ void assemble_tokens_bwd_kernel(const float* __restrict__ dx, bf16_t* __restrict__ dconv,
                                           float* __restrict__ dcls, float* __restrict__ ddist,
                                           float* __restrict__ dnew_pos, float* __restrict__ dfreq,
                                           float* __restrict__ dtime, int toffset, int B, int tp) {
    // one block per (f or token-pair or time column, d-chunk, batch slice): partial sums over the slice's clips in registers,
    // then one atomic per output (the batch slices run in parallel: a single block per job walked B * tp rows back to back)
    const int N = 2 + 12 * tp;
    const int d = blockIdx.y * 256 + threadIdx.x;
    if (d >= DM) return;
    const int bper = (B + gridDim.z - 1) / gridDim.z, b0 = blockIdx.z * bper, b1 = (b0 + bper < B) ? b0 + bper : B;
    const int job = blockIdx.x;  // 0: cls/dist, 1..12: freq row f = job-1 (also writes dconv), 13..: time cols
    if (job == 0) {
        float a = 0.f, c = 0.f;
        for (int b = b0; b < b1; ++b) {
            a += dx[((size_t)b * N) * DM + d];
            c += dx[((size_t)b * N + 1) * DM + d];
        }
        // every table may be frozen on its own (null = no gradient wanted)
        if (dcls != nullptr) unsafeAtomicAdd(&dcls[d], a);
        if (ddist != nullptr) unsafeAtomicAdd(&ddist[d], c);
        if (dnew_pos != nullptr) { unsafeAtomicAdd(&dnew_pos[d], a); unsafeAtomicAdd(&dnew_pos[DM + d], c); }
    } else if (job <= 12) {
        const int f = job - 1;
        float a = 0.f;
        for (int b = b0; b < b1; ++b)
            for (int t = 0; t < tp; ++t) {
                const float v = dx[((size_t)b * N + 2 + f * tp + t) * DM + d];
                a += v;
                dconv[((size_t)b * 12 * tp + f * tp + t) * DM + d] = f2bf(v);
            }
        if (dfreq != nullptr) unsafeAtomicAdd(&dfreq[d * 12 + f], a);
    } else {
        const int t = job - 13;
        if (dtime == nullptr) return;
        float a = 0.f;
        for (int b = b0; b < b1; ++b)
            for (int f = 0; f < 12; ++f) a += dx[((size_t)b * N + 2 + f * tp + t) * DM + d];
        unsafeAtomicAdd(&dtime[d * 99 + toffset + t], a);
    }
}
extern "C" int sed_assemble_tokens_bwd(const float* dx, void* dconv, float* dcls, float* ddist, float* dnew_pos,
                                       float* dfreq, float* dtime, int toffset, int B, int tp, hipStream_t stream) {
    (void)hipGetLastError();
    hipLaunchKernelGGL(assemble_tokens_bwd_kernel, dim3(13 + tp, DM / 256, B < 8 ? B : 8), dim3(256), 0, stream, dx, (bf16_t*)dconv,
                       dcls, ddist, dnew_pos, dfreq, dtime, toffset, B, tp);
    return sed_check_launch();
}

// ---------------------------------------------------------------------------------------------------
// f_pool (passt_sed.py:199-218, 'mean_pool'):  pooled[b, t] = mean_f LN_out_norm(x[b, 2 + f tp + t])
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fpool_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps,
                                                        float* __restrict__ pooled, float* __restrict__ mean,
                                                        float* __restrict__ rstd, int B, int tp) {
    const int lane = threadIdx.x & 63;
    const int job = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (job >= B * tp) return;
    const int b = job / tp, t = job - b * tp, N = 2 + 12 * tp;
    Row g, bt, acc;
    row_load(g, gamma, lane);
    row_load(bt, beta, lane);
#pragma unroll
    ROW_FOREACH(i, c) f4(acc.v[i], c) = 0.f;
    for (int f = 0; f < 12; ++f) {
        const size_t tok = (size_t)b * N + 2 + f * tp + t;
        Row r;
        row_load_nt(r, x + tok * DM, lane);
        float s = 0.f;
#pragma unroll
        ROW_FOREACH(i, c) s += f4(r.v[i], c);
        const float mu = wave_sum(s) * (1.0f / DM);
        float q = 0.f;
#pragma unroll
        ROW_FOREACH(i, c) { const float d = f4(r.v[i], c) - mu; q += d * d; }
        const float rs = rsqrtf(wave_sum(q) * (1.0f / DM) + eps);
#pragma unroll
        ROW_FOREACH(i, c) f4(acc.v[i], c) += (f4(r.v[i], c) - mu) * rs * f4(g.v[i], c) + f4(bt.v[i], c);
        if (lane == 0 && mean != nullptr) { mean[tok] = mu; rstd[tok] = rs; }
    }
#pragma unroll
    ROW_FOREACH(i, c) f4(acc.v[i], c) *= (1.0f / 12.0f);
    row_store(acc, pooled + (size_t)job * DM, lane);
}
extern "C" int sed_fpool_fwd(const float* x, const float* gamma, const float* beta, float eps, float* pooled,
                             float* mean, float* rstd, int B, int tp, hipStream_t stream) {
    (void)hipGetLastError();
    hipLaunchKernelGGL(fpool_fwd_kernel, dim3(cdiv(B * tp, 4)), dim3(256), 0, stream, x, gamma, beta, eps, pooled, mean,
                       rstd, B, tp);
    return sed_check_launch();
}
// backward = LayerNorm backward of the 12 B tp token rows with dy = dpooled[b, t] / 12, dx accumulated into the
// residual-stream gradient at rows 2 + f tp + t.  Implemented by expanding dpooled to token rows (cheap) and
// calling the LN backward kernel with dy_scale = 1/12.
__global__ void fpool_expand_kernel(const float* __restrict__ dpooled, float* __restrict__ dtok, int B, int tp) {
    const int N = 2 + 12 * tp;
    const size_t total = (size_t)B * N * (DM / 4);
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int d4 = (int)(idx % (DM / 4));
        const size_t tok = idx / (DM / 4);
        const int n = (int)(tok % N), b = (int)(tok / N);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n >= 2) {
            const int t = (n - 2) % tp;
            v = reinterpret_cast<const float4*>(dpooled)[((size_t)b * tp + t) * (DM / 4) + d4];
        }
        reinterpret_cast<float4*>(dtok)[idx] = v;
    }
}
extern "C" int sed_fpool_bwd(const float* dpooled, const float* x, const float* mean, const float* rstd,
                             const float* gamma, float* dtok_tmp, float* dx_acc, float* dgamma, float* dbeta, int B,
                             int tp, hipStream_t stream) {
    (void)hipGetLastError();
    const int N = 2 + 12 * tp, M = B * N;
    hipLaunchKernelGGL(fpool_expand_kernel, dim3(2048), dim3(256), 0, stream, dpooled, dtok_tmp, B, tp);
    // rows 0,1 of every clip have dy = 0 (and mean/rstd never written there -> use finite placeholders: the host
    // zero-fills mean/rstd once); their contribution is exactly 0 * finite.
    int blocks = cdiv(M, 4);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(blocks), dim3(256), 0, stream, (const float*)dtok_tmp, x, mean, rstd,
                       gamma, 1.0f, dx_acc, 1, dgamma, dbeta, M, 1.0f / 12.0f);
    return sed_check_launch();
}

// ---------------------------------------------------------------------------------------------------
// x`ratio` linear interpolation along time, align_corners=False (passt_sed.py:13-34,258-259)
// in [B, tin, D] (+ `pad` replicated frames at the end) -> out [B, ratio (tin + pad), D]
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void interp_coeff(int j, int ratio, int tlen, int tin, int& i0, int& i1, float& lam) {
    // torch's area_pixel_compute_source_index: scale * (dst + 0.5) - 0.5 with scale = (float)(1.0 / scale_factor)
    float src = (float)(1.0 / (double)ratio) * ((float)j + 0.5f) - 0.5f;
    src = src < 0.f ? 0.f : src;
    i0 = (int)src;
    i1 = i0 + 1 < tlen ? i0 + 1 : tlen - 1;
    lam = src - (float)i0;
    i0 = i0 < tin ? i0 : tin - 1;  // replicated padding frames
    i1 = i1 < tin ? i1 : tin - 1;
}
__global__ void interp_fwd_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int tin, int pad,
                                  int ratio) {
    const int tlen = tin + pad, tout = tlen * ratio;
    const size_t total = (size_t)B * tout * (DM / 4);
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int d4 = (int)(idx % (DM / 4));
        const size_t row = idx / (DM / 4);
        const int j = (int)(row % tout), b = (int)(row / tout);
        int i0, i1; float lam;
        interp_coeff(j, ratio, tlen, tin, i0, i1, lam);
        const float4 a = reinterpret_cast<const float4*>(in)[((size_t)b * tin + i0) * (DM / 4) + d4];
        const float4 c = reinterpret_cast<const float4*>(in)[((size_t)b * tin + i1) * (DM / 4) + d4];
        float4 o;
        o.x = (1.f - lam) * a.x + lam * c.x; o.y = (1.f - lam) * a.y + lam * c.y;
        o.z = (1.f - lam) * a.z + lam * c.z; o.w = (1.f - lam) * a.w + lam * c.w;
        reinterpret_cast<float4*>(out)[idx] = o;
    }
}
extern "C" int sed_interp_fwd(const float* in, float* out, int B, int tin, int pad, int ratio, hipStream_t stream) {
    (void)hipGetLastError();
    hipLaunchKernelGGL(interp_fwd_kernel, dim3(2048), dim3(256), 0, stream, in, out, B, tin, pad, ratio);
    return sed_check_launch();
}
__global__ void interp_bwd_kernel(const float* __restrict__ dout, float* __restrict__ din, int B, int tin, int pad,
                                  int ratio) {
    const int tlen = tin + pad, tout = tlen * ratio;
    const size_t total = (size_t)B * tin * (DM / 4);
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int d4 = (int)(idx % (DM / 4));
        const size_t row = idx / (DM / 4);
        const int i = (int)(row % tin), b = (int)(row / tin);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        int jlo = (i - 1) * ratio - ratio, jhi = (i == tin - 1) ? tout - 1 : (i + 1) * ratio + ratio;
        jlo = jlo < 0 ? 0 : jlo;
        jhi = jhi > tout - 1 ? tout - 1 : jhi;
        for (int j = jlo; j <= jhi; ++j) {
            int i0, i1; float lam;
            interp_coeff(j, ratio, tlen, tin, i0, i1, lam);
            float w = 0.f;
            if (i0 == i) w += 1.f - lam;
            if (i1 == i) w += lam;
            if (w != 0.f) {
                const float4 g = reinterpret_cast<const float4*>(dout)[((size_t)b * tout + j) * (DM / 4) + d4];
                acc.x += w * g.x; acc.y += w * g.y; acc.z += w * g.z; acc.w += w * g.w;
            }
        }
        reinterpret_cast<float4*>(din)[idx] = acc;
    }
}
extern "C" int sed_interp_bwd(const float* dout, float* din, int B, int tin, int pad, int ratio, hipStream_t stream) {
    (void)hipGetLastError();
    hipLaunchKernelGGL(interp_bwd_kernel, dim3(1024), dim3(256), 0, stream, dout, din, B, tin, pad, ratio);
    return sed_check_launch();
}

// Sliding-window merge (encoder_slide_window.py:16-36 + passt_sed.py:266-271), windows folded into the batch:
// window w's pooled frames live at pooled_win + offs[w] * D as [B, tps[w], D] (the last window of a sweep can be one
// patch shorter than the others);
//   x[b, j] = (1 - mix) x[b, j] + mix * (sum_w interp(pooled_w[b])[j - left_w]) / cnt_j
// (cnt_j == 0 -> local part is 0, the reference's NaN -> 0).
__global__ void window_mix_kernel(const float* __restrict__ pooled_win, const int* __restrict__ lefts,
                                  const int* __restrict__ tps, const int* __restrict__ offs, int nW,
                                  float* __restrict__ x, float mix, int B, int T, int ratio) {
    const size_t total = (size_t)B * T * (DM / 4);
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int d4 = (int)(idx % (DM / 4));
        const size_t row = idx / (DM / 4);
        const int j = (int)(row % T), b = (int)(row / T);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        int cnt = 0;
        for (int w = 0; w < nW; ++w) {
            const int jj = j - lefts[w], tpw = tps[w];
            if (jj < 0 || jj >= tpw * ratio) continue;
            int i0, i1; float lam;
            interp_coeff(jj, ratio, tpw, tpw, i0, i1, lam);
            const float* base = pooled_win + ((size_t)offs[w] + (size_t)b * tpw) * DM;
            const float4 a = reinterpret_cast<const float4*>(base)[(size_t)i0 * (DM / 4) + d4];
            const float4 c = reinterpret_cast<const float4*>(base)[(size_t)i1 * (DM / 4) + d4];
            acc.x += (1.f - lam) * a.x + lam * c.x; acc.y += (1.f - lam) * a.y + lam * c.y;
            acc.z += (1.f - lam) * a.z + lam * c.z; acc.w += (1.f - lam) * a.w + lam * c.w;
            ++cnt;
        }
        const float inv = cnt > 0 ? 1.0f / (float)cnt : 0.f;
        float4 g = reinterpret_cast<float4*>(x)[idx];
        g.x = mix * (acc.x * inv) + (1.f - mix) * g.x; g.y = mix * (acc.y * inv) + (1.f - mix) * g.y;
        g.z = mix * (acc.z * inv) + (1.f - mix) * g.z; g.w = mix * (acc.w * inv) + (1.f - mix) * g.w;
        reinterpret_cast<float4*>(x)[idx] = g;
    }
}
extern "C" int sed_window_mix(const float* pooled_win, const int* lefts, const int* tps, const int* offs, int nW,
                              float* x, float mix, int B, int T, int ratio, hipStream_t stream) {
    (void)hipGetLastError();
    hipLaunchKernelGGL(window_mix_kernel, dim3(2048), dim3(256), 0, stream, pooled_win, lefts, tps, offs, nW, x, mix, B,
                       T, ratio);
    return sed_check_launch();
}

// backward of the merge: every packed source row (window w, clip b, pooled frame i) gathers the output frames its two interpolation
// taps reach, weighted by mix / cnt_j; the global branch gets (1 - mix) dx.  No atomics: one thread owns one (row, 4 channels).
__global__ void window_mix_bwd_kernel(const float* __restrict__ dx, const int* __restrict__ lefts, const int* __restrict__ tps,
                                      const int* __restrict__ offs, int nW, float* __restrict__ dpooled, float* __restrict__ dglobal,
                                      float mix, int B, int T, int ratio, int rows) {
    const size_t total = (size_t)rows * (DM / 4);
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int d4 = (int)(idx % (DM / 4));
        const int row = (int)(idx / (DM / 4));
        int w = 0;
        for (int k = 1; k < nW; ++k) w = (row >= offs[k]) ? ((offs[k] >= offs[w]) ? k : w) : w;   // the window whose row range holds `row`
        const int tpw = tps[w], rel = row - offs[w], b = rel / tpw, i = rel - b * tpw, left = lefts[w];
        int jlo = (i - 1) * ratio - ratio, jhi = (i + 1) * ratio + ratio;
        jlo = jlo < 0 ? 0 : jlo;
        jhi = jhi > tpw * ratio - 1 ? tpw * ratio - 1 : jhi;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int jj = jlo; jj <= jhi; ++jj) {
            const int j = left + jj;
            if (j >= T) break;
            int i0, i1; float lam;
            interp_coeff(jj, ratio, tpw, tpw, i0, i1, lam);
            float wgt = 0.f;
            if (i0 == i) wgt += 1.f - lam;
            if (i1 == i) wgt += lam;
            if (wgt == 0.f) continue;
            int cnt = 0;
            for (int k = 0; k < nW; ++k) { const int q = j - lefts[k]; cnt += (q >= 0 && q < tps[k] * ratio) ? 1 : 0; }
            wgt *= mix / (float)cnt;
            const float4 g = reinterpret_cast<const float4*>(dx)[((size_t)b * T + j) * (DM / 4) + d4];
            acc.x += wgt * g.x; acc.y += wgt * g.y; acc.z += wgt * g.z; acc.w += wgt * g.w;
        }
        reinterpret_cast<float4*>(dpooled)[idx] = acc;
    }
    const size_t tot2 = (size_t)B * T * (DM / 4);
    const float s = 1.f - mix;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < tot2; idx += (size_t)gridDim.x * blockDim.x) {
        const float4 g = reinterpret_cast<const float4*>(dx)[idx];
        reinterpret_cast<float4*>(dglobal)[idx] = make_float4(s * g.x, s * g.y, s * g.z, s * g.w);
    }
}
extern "C" int sed_window_mix_bwd(const float* dx, const int* lefts, const int* tps, const int* offs, int nW, float* dpooled_win,
                                  float* dglobal, float mix, int B, int T, int ratio, int rows, hipStream_t stream) {
    (void)hipGetLastError();
    if (nW <= 0 || rows <= 0) return SED_ERR_ARG;
    hipLaunchKernelGGL(window_mix_bwd_kernel, dim3(2048), dim3(256), 0, stream, dx, lefts, tps, offs, nW, dpooled_win, dglobal, mix, B,
                       T, ratio, rows);
    return sed_check_launch();
}

// ---------------------------------------------------------------------------------------------------
// MLM masking (mask.py:62-85) and masked MSE (mlm_passt/train.py:36-38)
// ---------------------------------------------------------------------------------------------------
// action[row]: 0 keep, 1 -> mask_token, 2 -> copy row src_idx[row] of the ORIGINAL sequence
__global__ void mlm_apply_kernel(const float* __restrict__ x, const float* __restrict__ mask_token,
                                 const unsigned char* __restrict__ action, const int* __restrict__ src_idx,
                                 float* __restrict__ out, int rows) {
    const size_t total = (size_t)rows * (DM / 4);
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int d4 = (int)(idx % (DM / 4));
        const int row = (int)(idx / (DM / 4));
        const unsigned char a = action[row];
        float4 v;
        if (a == 1) v = reinterpret_cast<const float4*>(mask_token)[d4];
        else if (a == 2) v = reinterpret_cast<const float4*>(x)[(size_t)src_idx[row] * (DM / 4) + d4];
        else v = reinterpret_cast<const float4*>(x)[idx];
        reinterpret_cast<float4*>(out)[idx] = v;
    }
}
extern "C" int sed_mlm_apply(const float* x, const float* mask_token, const uint8_t* action, const int* src_idx,
                             float* out, int rows, hipStream_t stream) {
    (void)hipGetLastError();
    hipLaunchKernelGGL(mlm_apply_kernel, dim3(2048), dim3(256), 0, stream, x, mask_token, action, src_idx, out, rows);
    return sed_check_launch();
}
// backward of the masking: dx[row] = (action == 0) * dout[row] + scatter-add of 'copy' rows; dmask_token = sum of
// 'mask' rows.  (Only reached when the masking is effective, see DESIGN.md reference quirk 15.)
__global__ void mlm_apply_bwd_kernel(const float* __restrict__ dout, const unsigned char* __restrict__ action,
                                     const int* __restrict__ src_idx, float* __restrict__ dx,
                                     float* __restrict__ dtoken, int rows) {
    const size_t total = (size_t)rows * DM;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int d = (int)(idx % DM);
        const int row = (int)(idx / DM);
        const unsigned char a = action[row];
        const float g = dout[idx];
        if (a == 0) unsafeAtomicAdd(&dx[idx], g);
        else if (a == 1) unsafeAtomicAdd(&dtoken[d], g);
        else unsafeAtomicAdd(&dx[(size_t)src_idx[row] * DM + d], g);
    }
}
extern "C" int sed_mlm_apply_bwd(const float* dout, const uint8_t* action, const int* src_idx, float* dx_zeroed,
                                 float* dtoken, int rows, hipStream_t stream) {
    (void)hipGetLastError();
    hipLaunchKernelGGL(mlm_apply_bwd_kernel, dim3(2048), dim3(256), 0, stream, dout, action, src_idx, dx_zeroed, dtoken,
                       rows);
    return sed_check_launch();
}

// loss = mean over masked rows x D of (target - pred)^2 ; dpred = 2 (pred - target) / n, dtarget = -dpred.
__global__ __launch_bounds__(256) void masked_mse_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                         const unsigned char* __restrict__ mask, float inv_n,
                                                         const int* __restrict__ n_dev, float* __restrict__ loss,
                                                         float* __restrict__ dpred, float* __restrict__ dtarget, int rows) {
    __shared__ float red[4];
    if (n_dev != nullptr) inv_n = 1.0f / (fmaxf((float)n_dev[0], 1.0f) * (float)DM);   // count produced on the device: no host sync
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc = 0.f;
    for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
        const bool m = mask[row] != 0;
        Row p, t;
        if (m) {
            row_load(p, pred + (size_t)row * DM, lane);
            row_load(t, target + (size_t)row * DM, lane);
        }
#pragma unroll
        ROW_FOREACH(i, c) {
            const float d = m ? f4(p.v[i], c) - f4(t.v[i], c) : 0.f;
            acc += d * d;
            f4(p.v[i], c) = 2.f * d * inv_n;
            f4(t.v[i], c) = -2.f * d * inv_n;
        }
        if (dpred != nullptr) row_store(p, dpred + (size_t)row * DM, lane);
        if (dtarget != nullptr) row_store(t, dtarget + (size_t)row * DM, lane);
    }
    acc = wave_sum(acc);
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) unsafeAtomicAdd(loss, (red[0] + red[1] + red[2] + red[3]) * inv_n);
}
extern "C" int sed_masked_mse(const float* pred, const float* target, const uint8_t* mask, int n_masked_rows,
                              const int* n_masked_rows_dev, float* loss_zeroed, float* dpred, float* dtarget, int rows,
                              hipStream_t stream) {
    (void)hipGetLastError();
    if (n_masked_rows <= 0 && n_masked_rows_dev == nullptr) return SED_ERR_ARG;
    const float inv_n = n_masked_rows > 0 ? 1.0f / ((float)n_masked_rows * (float)DM) : 0.f;
    hipLaunchKernelGGL(masked_mse_kernel, dim3(512), dim3(256), 0, stream, pred, target, mask, inv_n, n_masked_rows_dev, loss_zeroed,
                       dpred, dtarget, rows);
    return sed_check_launch();
}

// ---------------------------------------------------------------------------------------------------
// The six loss terms of the mean-teacher step and their gradients (recipes/desed/finetune/train.py:160-191) in one pass over the
// posteriors: torch.nn.BCELoss (log clamped at -100; gradient (p - y) / max(p (1 - p), 1e-12)) and MSELoss, mean reductions.
// sums[0..5] = un-normalised {bce_strong, bce_weak, bce_at, se_strong, se_weak, se_at}; a one-thread kernel turns them into the
// seven scalars the trainer logs.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float bce_term(float p, float y) {
    const float lp = fmaxf(logf(p), -100.f), lq = fmaxf(logf(1.f - p), -100.f);
    return -(y * lp + (1.f - y) * lq);
}
__device__ __forceinline__ float bce_grad(float p, float y) { return (p - y) / fmaxf((1.f - p) * p, 1e-12f); }
__global__ void zero8_kernel(float* p) { if (threadIdx.x < 8) p[threadIdx.x] = 0.f; }
__global__ __launch_bounds__(256) void sed_losses_kernel(const float* __restrict__ ss, const float* __restrict__ sw, const float* __restrict__ sa,
                                                         const float* __restrict__ ts, const float* __restrict__ ta,
                                                         const float* __restrict__ y, const float* __restrict__ yw, int B, int C, int T,
                                                         int strong_n, int weak_lo, int weak_n, float w_weak, float w_weak_cons, float w_at,
                                                         float w_cons, float* __restrict__ sums, float* __restrict__ ds,
                                                         float* __restrict__ dw, float* __restrict__ da) {
    __shared__ float red[4][6];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const size_t per_clip = (size_t)C * T, total = (size_t)B * per_clip;
    const float g_bce_s = strong_n > 0 ? 1.f / ((float)strong_n * (float)C * (float)T) : 0.f;
    const float g_mse_s = 2.f * w_cons / ((float)B * (float)C * (float)T);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / per_clip);
        const float p = ss[i], q = ts[i];
        float g = g_mse_s * (p - q);
        acc[3] += (p - q) * (p - q);
        if (b < strong_n) {
            const float t = y[i];
            acc[0] += bce_term(p, t);
            g += g_bce_s * bce_grad(p, t);
        }
        ds[i] = g;
    }
    if (blockIdx.x == 0) {       // clip-level terms: B x C values
        const float g_bce_w = weak_n > 0 ? 1.f / ((float)weak_n * (float)C) : 0.f, g_mse_w = 2.f * w_cons / ((float)B * (float)C);
        for (int i = threadIdx.x; i < B * C; i += blockDim.x) {
            const int b = i / C;
            const bool in_w = b >= weak_lo && b < weak_lo + weak_n;
            const float pw = sw[i], pa = sa[i], qa = ta[i];
            float gw = g_mse_w * w_weak_cons * (pw - qa), ga = g_mse_w * w_at * (pa - qa);
            acc[4] += (pw - qa) * (pw - qa);
            acc[5] += (pa - qa) * (pa - qa);
            if (in_w) {
                const float t = yw[i];
                acc[1] += bce_term(pw, t);
                acc[2] += bce_term(pa, t);
                gw += w_weak * g_bce_w * bce_grad(pw, t);
                ga += w_at * g_bce_w * bce_grad(pa, t);
            }
            dw[i] = gw;
            da[i] = ga;
        }
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const float v = wave_sum(acc[k]);
        if (lane == 0) red[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 6) unsafeAtomicAdd(&sums[threadIdx.x], red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}
__global__ void sed_losses_final_kernel(const float* __restrict__ sums, float* __restrict__ out, int B, int C, int T, int strong_n,
                                        int weak_n, float w_weak, float w_weak_cons, float w_at, float w_cons) {
    if (threadIdx.x != 0) return;
    const float nan = __int_as_float(0x7fc00000);     // torch: the mean over an empty selection is NaN
    const float l_strong = strong_n > 0 ? sums[0] / ((float)strong_n * C * T) : nan;
    const float l_weak = weak_n > 0 ? sums[1] / ((float)weak_n * C) : nan, l_at = weak_n > 0 ? sums[2] / ((float)weak_n * C) : nan;
    const float lc_strong = sums[3] / ((float)B * C * T), lc_weak = sums[4] / ((float)B * C), lc_at = sums[5] / ((float)B * C);
    out[0] = l_strong + w_weak * l_weak + (lc_strong + w_weak_cons * lc_weak + w_at * lc_at) * w_cons + l_at * w_at;
    out[1] = l_strong; out[2] = l_weak; out[3] = l_at; out[4] = lc_strong; out[5] = lc_weak; out[6] = lc_at; out[7] = 0.f;
}
extern "C" int sed_sed_losses(const float* s_strong, const float* s_weak, const float* s_at, const float* t_strong, const float* t_at,
                              const float* labels, const float* labels_weak, int B, int C, int T, int strong_n, int weak_lo, int weak_n,
                              float w_weak, float w_weak_cons, float w_at, float w_cons, float* scratch, float* out, float* d_strong,
                              float* d_weak, float* d_at, hipStream_t stream) {
    (void)hipGetLastError();
    if (B <= 0 || C <= 0 || T <= 0 || strong_n < 0 || strong_n > B || weak_lo < 0 || weak_n < 0 || weak_lo + weak_n > B) return SED_ERR_ARG;
    hipLaunchKernelGGL(zero8_kernel, dim3(1), dim3(64), 0, stream, scratch);
    int blocks = cdiv((int64_t)B * C * T, 256 * 4);
    blocks = blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks);
    hipLaunchKernelGGL(sed_losses_kernel, dim3(blocks), dim3(256), 0, stream, s_strong, s_weak, s_at, t_strong, t_at, labels, labels_weak, B, C,
                       T, strong_n, weak_lo, weak_n, w_weak, w_weak_cons, w_at, w_cons, scratch, d_strong, d_weak, d_at);
    hipLaunchKernelGGL(sed_losses_final_kernel, dim3(1), dim3(64), 0, stream, scratch, out, B, C, T, strong_n, weak_n, w_weak, w_weak_cons,
                       w_at, w_cons);
    return sed_check_launch();
}

// ---------------------------------------------------------------------------------------------------
// SED head (passt_sed.py:285-296): logits = W x + b, s = sigmoid(logit / temp), pad mask, linear-softmax pooling
// ---------------------------------------------------------------------------------------------------
#define NCLS_MAX 16
__global__ __launch_bounds__(256) void sed_head_fwd_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                           const float* __restrict__ bias, float inv_temp,
                                                           const unsigned char* __restrict__ pad_mask,
                                                           float* __restrict__ strong, int B, int T, int C) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B * T) return;
    const int b = row / T, t = row - b * T;
    Row r;
    row_load(r, x + (size_t)row * DM, lane);
    const bool masked = pad_mask != nullptr && pad_mask[row] != 0;
    for (int c = 0; c < C; ++c) {
        Row w;
        row_load(w, W + (size_t)c * DM, lane);
        float s = 0.f;
#pragma unroll
        ROW_FOREACH(i, k) s += f4(r.v[i], k) * f4(w.v[i], k);
        s = wave_sum(s);
        if (lane == 0) {
            const float p = masked ? 0.f : sigmoidf_((s + bias[c]) * inv_temp);
            strong[((size_t)b * C + c) * T + t] = p;
        }
    }
}
// weak[b, c] = clamp(sum s^2 / sum s, 1e-7, 1); sums saved for backward
__global__ __launch_bounds__(256) void weak_pool_kernel(const float* __restrict__ strong, float* __restrict__ weak,
                                                        float* __restrict__ sums, int T) {
    __shared__ float ra[4], rb[4];
    const float* s = strong + (size_t)blockIdx.x * T;
    float a = 0.f, bsum = 0.f;
    for (int t = threadIdx.x; t < T; t += 256) { const float v = s[t]; a += v * v; bsum += v; }
    a = wave_sum(a); bsum = wave_sum(bsum);
    if ((threadIdx.x & 63) == 0) { ra[threadIdx.x >> 6] = a; rb[threadIdx.x >> 6] = bsum; }
    __syncthreads();
    if (threadIdx.x == 0) {
        a = ra[0] + ra[1] + ra[2] + ra[3];
        bsum = rb[0] + rb[1] + rb[2] + rb[3];
        float w = a / bsum;
        // torch.clamp propagates NaN (a clip whose every frame is padded gives 0/0 in the reference, passt_sed.py:293-294);
        // fminf/fmaxf would silently turn it into 1e-7
        w = (w != w) ? w : fminf(fmaxf(w, 1e-7f), 1.0f);
        weak[blockIdx.x] = w;
        if (sums != nullptr) { sums[2 * blockIdx.x] = a; sums[2 * blockIdx.x + 1] = bsum; }
    }
}
extern "C" int sed_head_fwd(const float* x, const float* W, const float* bias, float temp, const uint8_t* pad_mask,
                            float* strong, float* weak, float* sums, int B, int T, int C, hipStream_t stream) {
    (void)hipGetLastError();
    if (C > NCLS_MAX) return SED_ERR_ARG;
    hipLaunchKernelGGL(sed_head_fwd_kernel, dim3(cdiv(B * T, 4)), dim3(256), 0, stream, x, W, bias, 1.0f / temp, pad_mask,
                       strong, B, T, C);
    hipLaunchKernelGGL(weak_pool_kernel, dim3(B * C), dim3(256), 0, stream, (const float*)strong, weak, sums, T);
    return sed_check_launch();
}
// backward: dlogit[b,t,c] = (dstrong[b,c,t] + dweak[b,c] * (2 s B - A) / B^2 [if unclamped]) * s (1 - s) / temp
// dx[row] = sum_c dlogit W[c];  dW[c] += sum_rows dlogit x[row];  db[c] += sum dlogit
#define HEAD_C 10
// A wave owns 16 consecutive rows (b, t): lane = (class group cg = lane >> 4, row rl = lane & 15) evaluates dlogit for classes
// cg, cg + 4, cg + 8 of its row -- 64-byte coalesced segments of `strong` / `dstrong` along t; the value for (row rl, class c) then
// lives in lane (c & 3) * 16 + rl, register c >> 2, and is broadcast with a uniform-index shuffle when the row is processed.
__device__ __forceinline__ void head_dlogits16(float (&mine)[3], const float* __restrict__ strong, const float* __restrict__ sums,
                                               const float* __restrict__ dstrong, const float* __restrict__ dweak, float inv_temp,
                                               int row, int nrows, int T, int lane) {
    const int cg = lane >> 4;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int c = cg + 4 * k;
        mine[k] = 0.f;
        if (c < HEAD_C && row < nrows) {
            const int b = row / T, t = row - b * T;
            const size_t si = ((size_t)b * HEAD_C + c) * T + t;
            const float s = strong[si];
            float g = dstrong != nullptr ? dstrong[si] : 0.f;
            if (dweak != nullptr) {
                const float A = sums[2 * (b * HEAD_C + c)], Bs = sums[2 * (b * HEAD_C + c) + 1];
                const float wv = A / Bs;
                if (wv > 1e-7f && wv < 1.0f) g += dweak[b * HEAD_C + c] * (2.f * s * Bs - A) / (Bs * Bs);
            }
            mine[k] = g * s * (1.f - s) * inv_temp;
        }
    }
}
__device__ __forceinline__ void head_dl_row(float (&dl)[HEAD_C], const float (&mine)[3], int rl) {
#pragma unroll
    for (int c = 0; c < HEAD_C; ++c) dl[c] = __shfl(mine[c >> 2], (c & 3) * 16 + rl, 64);
}
// Two sweeps over the rows inside one launch: (1) dx = sum_c dlogit_c W_c with the 10 classifier rows held in registers (one
// write of dx, nothing re-read), (2) dW_c += dlogit_c x with the 10 per-lane partial rows in registers (one read of x).  The
// class-by-class version re-read x and read-modify-wrote dx ten times (0.79 ms per step).
__global__ __launch_bounds__(256) void sed_head_bwd_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                           const float* __restrict__ strong,
                                                           const float* __restrict__ sums,
                                                           const float* __restrict__ dstrong,
                                                           const float* __restrict__ dweak, float inv_temp,
                                                           float* __restrict__ dx, float* __restrict__ dW,
                                                           float* __restrict__ db, int B, int T, int C) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nrows = B * T, nblk = (nrows + 15) / 16;
    {
        Row w[HEAD_C];
#pragma unroll
        for (int c = 0; c < HEAD_C; ++c) row_load(w[c], W + (size_t)c * DM, lane);
        for (int blk = blockIdx.x * 4 + wave; blk < nblk; blk += gridDim.x * 4) {
            float mine[3];
            head_dlogits16(mine, strong, sums, dstrong, dweak, inv_temp, blk * 16 + (lane & 15), nrows, T, lane);
            for (int rl = 0; rl < 16; ++rl) {
                const int row = blk * 16 + rl;
                if (row >= nrows) break;
                float dl[HEAD_C];
                head_dl_row(dl, mine, rl);
                Row o;
#pragma unroll
                ROW_FOREACH(i, k) {
                    float v = 0.f;
#pragma unroll
                    for (int c = 0; c < HEAD_C; ++c) v += dl[c] * f4(w[c].v[i], k);
                    f4(o.v[i], k) = v;
                }
                row_store(o, dx + (size_t)row * DM, lane);
            }
        }
    }
    if (dW == nullptr) return;
    Row pw[HEAD_C];
    float pb[HEAD_C];
#pragma unroll
    for (int c = 0; c < HEAD_C; ++c) {
        pb[c] = 0.f;
#pragma unroll
        ROW_FOREACH(i, k) f4(pw[c].v[i], k) = 0.f;
    }
    for (int blk = blockIdx.x * 4 + wave; blk < nblk; blk += gridDim.x * 4) {
        float mine[3];
        head_dlogits16(mine, strong, sums, dstrong, dweak, inv_temp, blk * 16 + (lane & 15), nrows, T, lane);
        for (int rl = 0; rl < 16; ++rl) {
            const int row = blk * 16 + rl;
            if (row >= nrows) break;
            float dl[HEAD_C];
            head_dl_row(dl, mine, rl);
            Row xr;
            row_load(xr, x + (size_t)row * DM, lane);
#pragma unroll
            for (int c = 0; c < HEAD_C; ++c) {
                pb[c] += dl[c];
#pragma unroll
                ROW_FOREACH(i, k) f4(pw[c].v[i], k) += dl[c] * f4(xr.v[i], k);
            }
        }
    }
    // 7680 + 10 addresses receive every workgroup's partial sums: combine the four waves in LDS first (the per-wave atomics of
    // 2048 waves onto the same addresses were the whole 0.38 ms of this kernel)
    __shared__ float red[4][DM];
    __shared__ float redb[4][HEAD_C];
#pragma unroll   // fully unrolled: a runtime index into pw[] would put the 10 partial rows into scratch memory
    for (int c = 0; c < HEAD_C; ++c) {
        row_store(pw[c], red[wave], lane);
        if (lane == 0) redb[wave][c] = pb[c];
        __syncthreads();
        for (int d = threadIdx.x; d < DM; d += 256)
            unsafeAtomicAdd(&dW[(size_t)c * DM + d], (red[0][d] + red[1][d]) + (red[2][d] + red[3][d]));
        __syncthreads();
    }
    if (threadIdx.x < HEAD_C) unsafeAtomicAdd(&db[threadIdx.x], (redb[0][threadIdx.x] + redb[1][threadIdx.x]) + (redb[2][threadIdx.x] + redb[3][threadIdx.x]));
}
extern "C" int sed_head_bwd(const float* x, const float* W, const float* strong, const float* sums,
                            const float* dstrong, const float* dweak, float temp, float* dx, float* dW, float* db,
                            int B, int T, int C, hipStream_t stream) {
    (void)hipGetLastError();
    if (C != HEAD_C) return SED_ERR_ARG;
    hipLaunchKernelGGL(sed_head_bwd_kernel, dim3(256), dim3(256), 0, stream, x, W, strong, sums, dstrong, dweak,
                       1.0f / temp, dx, dW, db, B, T, C);
    return sed_check_launch();
}

// ---------------------------------------------------------------------------------------------------
// AT head: single-query multi-head attention pooling (pooling.py:45-51).  kv bf16 [B, N, 2 D] (K | V), tokens 2..N-1
//   one workgroup per (b, h): scores over the P = N - 2 patch tokens, softmax, weighted V sum.
// ---------------------------------------------------------------------------------------------------
// 64-wide dot of one token row (64 consecutive 16-bit values = 8 x 16 B) with a vector held in LDS as float[64]
__device__ __forceinline__ float attnpool_dot64(const bf16_t* row, const float* vec, int f16) {
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const uint4 u = reinterpret_cast<const uint4*>(row)[c];
        const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bf16_t lo = (bf16_t)(w[e] & 0xFFFF), hi = (bf16_t)(w[e] >> 16);
            acc += (f16 ? h2f(lo) : bf2f(lo)) * vec[c * 8 + 2 * e] + (f16 ? h2f(hi) : bf2f(hi)) * vec[c * 8 + 2 * e + 1];
        }
    }
    return acc;
}
// Scores: one LANE per token (no cross-lane reduction per token); weighted V sum: one lane per channel, 8 independent token rows in
// flight per wave.  (The first version did one wave reduction per token: ~300 dependent round trips per wave, 320 us.)
__global__ __launch_bounds__(256) void attnpool_fwd_kernel(const bf16_t* __restrict__ kv, const float* __restrict__ q,
                                                           float* __restrict__ out, float* __restrict__ probs, int N,
                                                           int H, int f16) {
    extern __shared__ float sc[];  // [P]
    __shared__ float red[4];
    __shared__ float part[4][64];
    __shared__ float qs[64];
    const int b = blockIdx.x / H, h = blockIdx.x - b * H, P = N - 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bf16_t* base = kv + ((size_t)b * N + 2) * (2 * DM) + h * 64;
    if (threadIdx.x < 64) qs[threadIdx.x] = q[h * 64 + threadIdx.x];
    __syncthreads();
    float mx = -1e30f;
    for (int t = threadIdx.x; t < P; t += 256) {
        const float s = attnpool_dot64(base + (size_t)t * 2 * DM, qs, f16) * 0.125f;
        sc[t] = s;
        mx = fmaxf(mx, s);
    }
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.f;
    for (int t = threadIdx.x; t < P; t += 256) { const float e = __expf(sc[t] - mx); sc[t] = e; sum += e; }
    sum = wave_sum(sum);
    if (lane == 0) red[wave] = sum;
    __syncthreads();
    const float inv = 1.0f / (red[0] + red[1] + red[2] + red[3]);
    float acc = 0.f;
    const bf16_t* vbase = base + DM + lane;
    int t = wave;
    for (; t + 28 < P; t += 32) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const bf16_t x = vbase[(size_t)(t + 4 * u) * 2 * DM]; v[u] = f16 ? h2f(x) : bf2f(x); }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += sc[t + 4 * u] * v[u];
    }
    for (; t < P; t += 4) { const bf16_t x = vbase[(size_t)t * 2 * DM]; acc += sc[t] * (f16 ? h2f(x) : bf2f(x)); }
    part[wave][lane] = acc * inv;
    if (probs != nullptr)
        for (int tt = threadIdx.x; tt < P; tt += 256) probs[(size_t)blockIdx.x * P + tt] = sc[tt] * inv;
    __syncthreads();
    if (wave == 0) out[(size_t)b * DM + h * 64 + lane] = part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane];
}
extern "C" int sed_attnpool_fwd(const void* kv, const float* q, float* out, float* probs, int B, int N, int H,
                                int f16, hipStream_t stream) {
    (void)hipGetLastError();
    hipLaunchKernelGGL(attnpool_fwd_kernel, dim3(B * H), dim3(256), (N - 2) * sizeof(float), stream, (const bf16_t*)kv, q,
                       out, probs, N, H, f16);
    return sed_check_launch();
}
// backward: dout [B, D] -> dkv bf16 [B, N, 2D] (rows 0,1 zero), dq [D] (atomic over b)
__global__ __launch_bounds__(256) void attnpool_bwd_kernel(const bf16_t* __restrict__ kv, const float* __restrict__ q,
                                                           const float* __restrict__ probs,
                                                           const float* __restrict__ dout, bf16_t* __restrict__ dkv,
                                                           float* __restrict__ dq, int N, int H, int f16) {
    extern __shared__ float dp[];  // [P] -> dS
    __shared__ float red[4];
    __shared__ float part[4][64];
    __shared__ float gos[64];
    const int b = blockIdx.x / H, h = blockIdx.x - b * H, P = N - 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bf16_t* base = kv + ((size_t)b * N + 2) * (2 * DM) + h * 64;
    bf16_t* dbase = dkv + ((size_t)b * N + 2) * (2 * DM) + h * 64;
    const float* pr = probs + (size_t)blockIdx.x * P;
    if (threadIdx.x < 64) gos[threadIdx.x] = dout[(size_t)b * DM + h * 64 + threadIdx.x];
    __syncthreads();
    const float go = gos[lane], qd = q[h * 64 + lane];
    // dP[t] = dout . V[t]  (one lane per token), dot = sum_t p[t] dP[t]
    float dot = 0.f;
    for (int t = threadIdx.x; t < P; t += 256) {
        const float v = attnpool_dot64(base + (size_t)t * 2 * DM + DM, gos, f16);
        dp[t] = v;
        dot += v * pr[t];
    }
    dot = wave_sum(dot);
    if (lane == 0) red[wave] = dot;
    __syncthreads();
    dot = red[0] + red[1] + red[2] + red[3];
    for (int t = threadIdx.x; t < P; t += 256) dp[t] = pr[t] * (dp[t] - dot) * 0.125f;   // dS[t]
    __syncthreads();
    // dq[d] = sum_t dS[t] K[t][d];  dK[t][d] = dS[t] q[d];  dV[t][d] = p[t] dout[d]   (one lane per channel, 8 rows in flight)
    float dqa = 0.f;
    int t = wave;
    for (; t + 28 < P; t += 32) {
        float kx[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const bf16_t x = base[(size_t)(t + 4 * u) * 2 * DM + lane]; kx[u] = f16 ? h2f(x) : bf2f(x); }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float ds = dp[t + 4 * u];
            dqa += ds * kx[u];
            dbase[(size_t)(t + 4 * u) * 2 * DM + lane] = f2bf(ds * qd);
            dbase[(size_t)(t + 4 * u) * 2 * DM + DM + lane] = f2bf(pr[t + 4 * u] * go);
        }
    }
    for (; t < P; t += 4) {
        const bf16_t x = base[(size_t)t * 2 * DM + lane];
        const float ds = dp[t];
        dqa += ds * (f16 ? h2f(x) : bf2f(x));
        dbase[(size_t)t * 2 * DM + lane] = f2bf(ds * qd);
        dbase[(size_t)t * 2 * DM + DM + lane] = f2bf(pr[t] * go);
    }
    if (wave == 0) {  // cls/dist rows receive no gradient from the AT head
        bf16_t* z = dkv + ((size_t)b * N) * (2 * DM);
        z[h * 64 + lane] = 0; z[DM + h * 64 + lane] = 0;
        z[2 * DM + h * 64 + lane] = 0; z[2 * DM + DM + h * 64 + lane] = 0;
    }
    part[wave][lane] = dqa;
    __syncthreads();
    if (wave == 0) unsafeAtomicAdd(&dq[h * 64 + lane], part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane]);
}
extern "C" int sed_attnpool_bwd(const void* kv, const float* q, const float* probs, const float* dout, void* dkv,
                                float* dq, int B, int N, int H, int f16, hipStream_t stream) {
    (void)hipGetLastError();
    hipLaunchKernelGGL(attnpool_bwd_kernel, dim3(B * H), dim3(256), (N - 2) * sizeof(float), stream, (const bf16_t*)kv, q,
                       probs, dout, (bf16_t*)dkv, dq, N, H, f16);
    return sed_check_launch();
}

// small fp32 linear backward: out = act(a W^T + b);  given dout (w.r.t. the activated output when act == 1, with
// `out` supplied), produce da [M, K] (=), dW [N, K] (+=), db [N] (+=).  One thread per output element, coalesced over k.
__device__ __forceinline__ float slb_g(const float* dout, const float* out, size_t i, int act) {
    float g = dout[i];
    if (act == 1) { const float o = out[i]; g *= o * (1.f - o); }
    return g;
}
__global__ __launch_bounds__(256) void small_linear_bwd_kernel(const float* __restrict__ a, const float* __restrict__ w,
                                                               const float* __restrict__ out, const float* __restrict__ dout,
                                                               float* __restrict__ da, float* __restrict__ dw, float* __restrict__ db, int M,
                                                               int N, int K, int act, int nb_dw, int nb_da) {
    // blocks [0, nb_dw): one thread per dW element; [nb_dw, nb_dw + nb_da): one block per (row m, 32 columns of da), the N-long dot products
    // split eight ways over the block (one thread per element walked all N = 768 rows of W one dependent load after the other: 84-114 us
    // per launch for 25 K outputs); the rest: db
    __shared__ float red[8][32];
    if ((int)blockIdx.x < nb_dw) {
        const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
        if (idx >= (size_t)N * K || dw == nullptr) return;
        const int n = (int)(idx / K), k = (int)(idx - (size_t)n * K);
        float acc = 0.f;
        for (int m = 0; m < M; ++m) acc += slb_g(dout, out, (size_t)m * N + n, act) * a[(size_t)m * K + k];
        dw[idx] += acc;
    } else if ((int)blockIdx.x < nb_dw + nb_da) {
        const int t = blockIdx.x - nb_dw, kt = (K + 31) / 32;
        const int m = t / kt, k = (t - m * kt) * 32 + (threadIdx.x & 31), nc = threadIdx.x >> 5;
        const int chunk = (N + 7) / 8, n0 = nc * chunk, n1 = (n0 + chunk) < N ? (n0 + chunk) : N;
        float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
        if (k < K) {
            int n = n0;
            for (; n + 3 < n1; n += 4) {
                acc0 += slb_g(dout, out, (size_t)m * N + n, act) * w[(size_t)n * K + k];
                acc1 += slb_g(dout, out, (size_t)m * N + n + 1, act) * w[(size_t)(n + 1) * K + k];
                acc2 += slb_g(dout, out, (size_t)m * N + n + 2, act) * w[(size_t)(n + 2) * K + k];
                acc3 += slb_g(dout, out, (size_t)m * N + n + 3, act) * w[(size_t)(n + 3) * K + k];
            }
            for (; n < n1; ++n) acc0 += slb_g(dout, out, (size_t)m * N + n, act) * w[(size_t)n * K + k];
        }
        red[nc][threadIdx.x & 31] = (acc0 + acc1) + (acc2 + acc3);
        __syncthreads();
        if (nc == 0 && k < K)
            da[(size_t)m * K + k] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x])) +
                                    ((red[4][threadIdx.x] + red[5][threadIdx.x]) + (red[6][threadIdx.x] + red[7][threadIdx.x]));
    } else {
        const int n = (blockIdx.x - nb_dw - nb_da) * blockDim.x + threadIdx.x;
        if (n >= N || db == nullptr) return;
        float acc = 0.f;
        for (int m = 0; m < M; ++m) acc += slb_g(dout, out, (size_t)m * N + n, act);
        db[n] += acc;
    }
}
extern "C" int sed_small_linear_bwd(const float* a, const float* w, const float* out, const float* dout, float* da,
                                    float* dw, float* db, int M, int N, int K, int act, hipStream_t stream) {
    (void)hipGetLastError();
    if (M <= 0 || N <= 0 || K <= 0) return SED_ERR_ARG;
    const int nb_dw = dw != nullptr ? (int)cdiv((int64_t)N * K, 256) : 0, nb_da = da != nullptr ? M * ((K + 31) / 32) : 0;
    const int nb_db = db != nullptr ? (int)cdiv((int64_t)N, 256) : 0;
    if (nb_dw + nb_da + nb_db == 0) return SED_OK;
    hipLaunchKernelGGL(small_linear_bwd_kernel, dim3(nb_dw + nb_da + nb_db), dim3(256), 0, stream, a, w, out, dout, da, dw, db,
                       M, N, K, act, nb_dw, nb_da);
    return sed_check_launch();
}

// ---------------------------------------------------------------------------------------------------
// Fused AdamW (decoupled weight decay, torch.optim.AdamW semantics; recipes/desed/setting.py:254-258) + EMA teacher
// update (src/utils/scheduler.py:125-130) over a contiguous slice of the flat parameter arena.
// ---------------------------------------------------------------------------------------------------
__global__ void adamw_ema_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                 float* __restrict__ v, float* __restrict__ ema, size_t n4, float lr, float wd, float b1,
                                 float b2, float eps, float bc1, float bc2_sqrt, float ema_alpha, int do_adam) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 pp = reinterpret_cast<float4*>(p)[i];
        if (do_adam) {
            const float4 gg = reinterpret_cast<const float4*>(g)[i];
            float4 mm = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float& pc = f4(pp, c);
                const float gc = f4(gg, c);
                pc *= (1.f - lr * wd);
                const float mc = b1 * f4(mm, c) + (1.f - b1) * gc;
                const float vc = b2 * f4(vv, c) + (1.f - b2) * gc * gc;
                f4(mm, c) = mc; f4(vv, c) = vc;
                const float denom = sqrtf(vc) / bc2_sqrt + eps;
                pc -= (lr / bc1) * (mc / denom);
            }
            reinterpret_cast<float4*>(p)[i] = pp;
            reinterpret_cast<float4*>(m)[i] = mm;
            reinterpret_cast<float4*>(v)[i] = vv;
        }
        if (ema != nullptr) {
            float4 ee = reinterpret_cast<float4*>(ema)[i];
#pragma unroll
            for (int c = 0; c < 4; ++c) f4(ee, c) = f4(ee, c) * ema_alpha + f4(pp, c) * (1.f - ema_alpha);
            reinterpret_cast<float4*>(ema)[i] = ee;
        }
    }
}
extern "C" int sed_adamw_ema(float* p, const float* g, float* m, float* v, float* ema, int64_t n, float lr, float wd,
                             float beta1, float beta2, float eps, int step, float ema_alpha, int do_adam,
                             hipStream_t stream) {
    (void)hipGetLastError();
    if (n % 4) return SED_ERR_ARG;
    const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
    size_t n4 = n / 4;
    int blocks = (int)((n4 + 255) / 256);
    blocks = blocks > 8192 ? 8192 : (blocks < 1 ? 1 : blocks);
    hipLaunchKernelGGL(adamw_ema_kernel, dim3(blocks), dim3(256), 0, stream, p, g, m, v, ema, n4, lr, wd, beta1, beta2,
                       eps, bc1, sqrtf(bc2), ema_alpha, do_adam);
    return sed_check_launch();
}
