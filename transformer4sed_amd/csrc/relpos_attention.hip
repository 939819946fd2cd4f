// Transformer-XL relative-position multi-head attention for the MAT-SED context network (gfx950),
// forward + backward, without ever materialising the [B,12,T,2T-1] "BD" tensor of the reference.
//
// Replaces src/models/transformer/transformerXL.py:493-576 (linear_pos is a plain GEMM, see gemm.hip):
//   S[i,j] = ((q_i + u) . k_j  +  (q_i + v) . p_{j-i+T-1}) / sqrt(64),   softmax_j,   @ v
// `rel_shift` (transformerXL.py:254-297) is folded into the tile indexing: for a (32-query x 64-key) tile only
// the 95-row band p[j0-i0-31+T-1 ...] of the positional table matters; the band product is computed on the
// matrix cores as G^T[rho, q] and then *skewed* through a wave-private LDS buffer:
//   BD^T[key, q] = G^T[key - q + 31, q].
//
// Operands (head-split by gemm.hip EPI_QKV; all bf16):
//   Qu = q+u, Qv = q+v : [B*H, T, 64]      K, V : [B*H, T, 64]       Kt, Vt, Qut, Qvt : [B*H, 64, Tpad]
//   P  = linear_pos(pos_emb) head-split : [H, Rpad, 64],   Pt : [H, 64, Rpad]      (R = 2T-1 rows, Rpad % 64 == 0)
// Requires T % 8 == 0 (band columns of Pt are then 16-byte aligned); T = 1000 in MAT-SED.
#include <stdlib.h>
#include "common.h"
#include "../../include/sed_hip.h"

#include "attn_common.h"

#define BAND_ROWS 192

struct BandRegs { uint4 x0, x1, x2, x3, x4, x5; };

__device__ __forceinline__ uint4 band_load1(const bf16_t* Ph, int row, int R, int c) {
    row = row < 0 ? 0 : (row >= R ? R - 1 : row);
    return *reinterpret_cast<const uint4*>(Ph + (size_t)row * HD + c * 8);
}
// band rows [RB, RB + 192) of P_h [R][64] -> registers (thread: chunk c of rows (tid>>3) + 32 i)
__device__ __forceinline__ void band_gload(BandRegs& b, const bf16_t* Ph, int RB, int R, int tid) {
    const int r = RB + (tid >> 3), c = tid & 7;
    b.x0 = band_load1(Ph, r, R, c);
    b.x1 = band_load1(Ph, r + 32, R, c);
    b.x2 = band_load1(Ph, r + 64, R, c);
    b.x3 = band_load1(Ph, r + 96, R, c);
    b.x4 = band_load1(Ph, r + 128, R, c);
    b.x5 = band_load1(Ph, r + 160, R, c);
}
__device__ __forceinline__ void band_lstore(const BandRegs& b, unsigned char* dst, int tid) {
    const int r = tid >> 3, c = tid & 7;
    *reinterpret_cast<uint4*>(dst + k_off(r, c)) = b.x0;
    *reinterpret_cast<uint4*>(dst + k_off(r + 32, c)) = b.x1;
    *reinterpret_cast<uint4*>(dst + k_off(r + 64, c)) = b.x2;
    *reinterpret_cast<uint4*>(dst + k_off(r + 96, c)) = b.x3;
    *reinterpret_cast<uint4*>(dst + k_off(r + 128, c)) = b.x4;
    *reinterpret_cast<uint4*>(dst + k_off(r + 160, c)) = b.x5;
}
// transposed band image [64 d][192 rho] bf16 (384-B rows), 16-B chunks swizzled inside groups of 8
__device__ __forceinline__ int bt_off(int row, int ch16) { return row * 384 + ((ch16 ^ ((row >> 1) & 7)) << 4); }
// columns [CB, CB+192) of Pt_h [64][Rpad] -> LDS; CB % 8 == 0; out-of-range columns are clamped (never used
// by valid pairs).  64 rows x 24 chunks = 1536 chunks = 6 per thread.
__device__ __forceinline__ uint4 bandT_load1(const bf16_t* Pth, int CB, int Rpad, int idx) {
    const int row = idx / 24, ch = idx - row * 24;
    int col = CB + ch * 8;
    col = col < 0 ? 0 : (col + 8 > Rpad ? Rpad - 8 : col);
    return *reinterpret_cast<const uint4*>(Pth + (size_t)row * Rpad + col);
}
__device__ __forceinline__ void bandT_gload(BandRegs& b, const bf16_t* Pth, int CB, int Rpad, int tid) {
    b.x0 = bandT_load1(Pth, CB, Rpad, tid);
    b.x1 = bandT_load1(Pth, CB, Rpad, tid + 256);
    b.x2 = bandT_load1(Pth, CB, Rpad, tid + 512);
    b.x3 = bandT_load1(Pth, CB, Rpad, tid + 768);
    b.x4 = bandT_load1(Pth, CB, Rpad, tid + 1024);
    b.x5 = bandT_load1(Pth, CB, Rpad, tid + 1280);
}
__device__ __forceinline__ void bandT_lstore1(unsigned char* dst, int idx, uint4 v) {
    const int row = idx / 24, ch = idx - row * 24;
    *reinterpret_cast<uint4*>(dst + bt_off(row, ch)) = v;
}
__device__ __forceinline__ void bandT_lstore(const BandRegs& b, unsigned char* dst, int tid) {
    bandT_lstore1(dst, tid, b.x0);
    bandT_lstore1(dst, tid + 256, b.x1);
    bandT_lstore1(dst, tid + 512, b.x2);
    bandT_lstore1(dst, tid + 768, b.x3);
    bandT_lstore1(dst, tid + 1024, b.x4);
    bandT_lstore1(dst, tid + 1280, b.x5);
}

// ---------------------------------------------------------------------------------------------------
// forward, on v_mfma_f32_16x16x32 with 8 waves (2 per SIMD) per workgroup -- the structure of the dQ kernel below:
//   workgroup = 128 queries, wave = 16 queries x 64-key tiles, lane = (query column c = lane & 15, row group g = lane >> 4).
//   (The first version -- 4 waves x 32 queries on 32x32x16, 238 VGPRs and 128 KiB of LDS, i.e. ONE wave per SIMD with nothing to
//   hide its LDS round trips behind, G^T scattered with 48 ds_write_b32 and gathered with 32 ds_read_b32 per lane per tile -- ran at
//   0.11 of the MFMA peak.)
//   * P rows of the band live in a 256-row LDS ring filled by DMA (one 1-KiB piece per wave per tile), K rows and the V^T tile
//     ([64 d][64 keys], from the zero-padded transposed copy the in_proj epilogue writes) are single-buffered and prefetched
//     through 8 VGPRs per lane
//   * G [16 q][85] fp32 per wave: the band product of a tile; the four values a lane adds to one score accumulator are one aligned
//     16-byte read that enters the K Qu^T MFMA as its C operand -- the skew costs no VALU
//   * K rows are permuted inside each 32-key group (see the dQ kernel) so that lane group g's accumulator rows of two neighbouring
//     16-key blocks are the 8 consecutive keys 32 ks + 8 g .. + 7: P^T goes from the accumulators into the B operand of the
//     V^T P^T product without leaving the lane, its A operand is one 16-byte read of the V^T tile
// ---------------------------------------------------------------------------------------------------
template <bool F16> __device__ __forceinline__ f32x4_t mfma16x(s16x8_t a, s16x8_t b, f32x4_t c) {
    if (F16)
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
typedef __attribute__((address_space(3))) void* rp_lds_ptr_t;
#define zero4 (f32x4_t{0.f, 0.f, 0.f, 0.f})

#define FW16_WL 5440                       // per-wave LDS: G [16 q][85] fp32
// NW waves (16 NW queries) per workgroup; the ring holds the 16 NW + 63 band rows a tile touches plus the 64 the next one adds
#define FW16_RING(NW) (16 * (NW) + 128)
#define FW16_LDS(NW) (16384 + FW16_RING(NW) * 128 + (NW) * FW16_WL)
template <bool F16, bool O32, int NW>
__global__ __launch_bounds__(64 * NW) void relpos_fwd_kernel(const bf16_t* __restrict__ Qu, const bf16_t* __restrict__ Qv,
                                                         const bf16_t* __restrict__ K, const bf16_t* __restrict__ Vt,
                                                         const bf16_t* __restrict__ P, bf16_t* __restrict__ O,
                                                         bf16_t* __restrict__ Osplit, float* __restrict__ LSE, int T, int Tpad, int H, int Rpad) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds_fw[];
    unsigned char* lds_k = lds_fw;                      // K rows of the tile (permuted, see above)
    unsigned char* lds_vt = lds_fw + 8192;              // V^T tile [64 d][64 keys]
    unsigned char* lds_band = lds_fw + 16384;           // ring of P rows, slot = n & 255
    const int tid = threadIdx.x, lane = tid & 63, c = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int RING = FW16_RING(NW);
    float* gs = reinterpret_cast<float*>(lds_fw + 16384 + RING * 128 + wave * FW16_WL);
    const int bh = blockIdx.y, b = bh / H, h = bh - b * H;
    const int I0 = blockIdx.x * (16 * NW), q0 = I0 + wave * 16;
    const int R = 2 * T - 1;
    const size_t hb = (size_t)bh * T * HD, hbt = (size_t)bh * HD * Tpad;
    const int RB0 = T - 16 * NW - I0;      // global P row of band index n = 0 (multiple of 8: T % 8 == 0)
    // rows outside [0, R) of the head's P slab read as zeros (they only meet pairs that do not exist)
    const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void*)(P + (size_t)h * Rpad * HD), 0, R * HD * 2, 0x00020000);
    const int prow = lane >> 3, pch = lane & 7;
    auto dma_band = [&](int n0) {          // band rows n0 .. n0 + 7 (n0 % 8 == 0) -> ring
        const int s0 = n0 % RING;
        const int slot = s0 + prow;
        const int vo = (RB0 + n0 + prow) * (HD * 2) + ((pch ^ ((slot >> 1) & 7)) << 4);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rp, (rp_lds_ptr_t)(lds_band + s0 * 128), 16, vo, 0, 0, 0);
    };
    for (int n0 = 8 * wave; n0 < 16 * NW + 64; n0 += 8 * NW) dma_band(n0);      // rows tile 0 touches
    const int trow = (tid >> 3) & 63, tch = tid & 7;   // this thread's 16-byte chunk of a [64][64] tile (threads 0..511)
    const bool loader = tid < 512;
    uint4 pk = make_uint4(0, 0, 0, 0), pvt = make_uint4(0, 0, 0, 0);
    auto gload = [&](int t) {
        if (!loader) return;
        const int j0 = t * KVB;
        int kr = j0 + trow;
        kr = kr < T ? kr : T - 1;
        pk = *reinterpret_cast<const uint4*>(K + hb + (size_t)kr * HD + tch * 8);
        pvt = *reinterpret_cast<const uint4*>(Vt + hbt + (size_t)trow * Tpad + j0 + tch * 8);
    };
    const int krow_lds = (trow & 32) + (((trow >> 2) & 1) << 4) + (((trow >> 3) & 3) << 2) + (trow & 3);
    gload(0);
    int qrow = q0 + c;
    const bool qvalid = qrow < T;
    qrow = qvalid ? qrow : T - 1;
    s16x8_t quf[2], qvf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        quf[ks] = *reinterpret_cast<const s16x8_t*>(Qu + hb + (size_t)qrow * HD + 32 * ks + 8 * g);
        qvf[ks] = *reinterpret_cast<const s16x8_t*>(Qv + hb + (size_t)qrow * HD + 32 * ks + 8 * g);
    }
    if (loader) {
        *reinterpret_cast<uint4*>(lds_k + k_off(krow_lds, tch)) = pk;
        *reinterpret_cast<uint4*>(lds_vt + k_off(trow, tch)) = pvt;
    }
    f32x4_t o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = zero4;
    float m_run = -1e30f, l_run = 0.f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("" ::"v"(quf[0]), "v"(quf[1]), "v"(qvf[0]), "v"(qvf[1]));   // (arrived: no compiler wait inside the loop)
    __syncthreads();
    const int ntiles = (T + KVB - 1) / KVB;
    const int swc = (c >> 1) & 7;
    for (int t = 0; t < ntiles; ++t) {
        const int j0 = t * KVB;
        const bool more = t + 1 < ntiles;
        if (more) {   // tile t + 1: the 64 band rows it adds replace the slots tile t - 1 retired (one piece per wave 0..7)
            if (wave < 8) dma_band(64 * t + 16 * NW + 64 + 8 * wave);
            gload(t + 1);
        }
        const int nb = 64 * t + 16 * (NW - 1 - wave);    // this wave's band base
        // ---- G^T[rho, q] = P_band[rho, :] . Qv[q, :]  (5 blocks of 16 rho) -> wave-private LDS
#pragma unroll
        for (int blk = 0; blk < 5; ++blk) {
            const int slot = ((nb + 16 * blk) % RING) + c;
            const unsigned char* rowp = lds_band + slot * 128;
            const int sw = (slot >> 1) & 7;
            f32x4_t gacc = zero4;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                gacc = mfma16x<F16>(*reinterpret_cast<const s16x8_t*>(rowp + (((4 * ks + g) ^ sw) << 4)), qvf[ks], gacc);
#pragma unroll
            for (int r = 0; r < 4; ++r) gs[85 * c + 1 + 16 * blk + 4 * g + r] = gacc[r];
        }
        // ---- S^T = K Qu^T + skew(G^T): block kb, register r <-> key jj = 32 (kb >> 1) + 4 (kb & 1) + 8 g + r
        f32x4_t st[4];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) st[kb] = *reinterpret_cast<const f32x4_t*>(gs + 84 * c + 16 + 32 * (kb >> 1) + 4 * (kb & 1) + 8 * g);
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            const unsigned char* kp = lds_k + (16 * kb + c) * 128;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                st[kb] = mfma16x<F16>(*reinterpret_cast<const s16x8_t*>(kp + (((4 * ks + g) ^ swc) << 4)), quf[ks], st[kb]);
        }
        // barrier A: every wave has read the K rows of tile t -> the prefetched K rows go to LDS
        if (more) {
            __syncthreads();
            if (loader) *reinterpret_cast<uint4*>(lds_k + k_off(krow_lds, tch)) = pk;
        }
        // ---- online softmax of the lane's query over its 16 keys, combined over the four lane groups
        if (j0 + KVB > T) {   // keys >= T exist only in the last tile
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    st[kb][r] = (j0 + 32 * (kb >> 1) + 4 * (kb & 1) + 8 * g + r < T) ? st[kb][r] : -1e30f;
        }
        float mloc = max3_raw(st[0][0], st[0][1], st[0][2]);
        mloc = max3_raw(mloc, st[0][3], st[1][0]);
        mloc = max3_raw(mloc, st[1][1], st[1][2]);
        mloc = max3_raw(mloc, st[1][3], st[2][0]);
        mloc = max3_raw(mloc, st[2][1], st[2][2]);
        mloc = max3_raw(mloc, st[2][3], st[3][0]);
        mloc = max3_raw(mloc, st[3][1], st[3][2]);
        mloc = fmaxf(mloc, st[3][3]);
        mloc = fmaxf(mloc, __shfl_xor(mloc, 16, 64));
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
        const float m_new = fmaxf(m_run, mloc * SCALE_LOG2E);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        float psum = 0.f;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(st[kb][r], SCALE_LOG2E, -m_new));
                st[kb][r] = pv;
                psum += pv;
            }
        psum += __shfl_xor(psum, 16, 64);
        psum += __shfl_xor(psum, 32, 64);
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int db = 0; db < 4; ++db) o[db] *= alpha;
        // ---- O^T[d, q] += V^T[d, key] P^T[key, q]
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const uint4 pu = make_uint4(pack2<F16>(st[2 * ks][0], st[2 * ks][1]), pack2<F16>(st[2 * ks][2], st[2 * ks][3]),
                                        pack2<F16>(st[2 * ks + 1][0], st[2 * ks + 1][1]), pack2<F16>(st[2 * ks + 1][2], st[2 * ks + 1][3]));
            const s16x8_t pf = __builtin_bit_cast(s16x8_t, pu);
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                const unsigned char* rowp = lds_vt + (16 * db + c) * 128;   // (row >> 1) & 7 == swc for every 16-row block
                o[db] = mfma16x<F16>(*reinterpret_cast<const s16x8_t*>(rowp + (((4 * ks + g) ^ swc) << 4)), pf, o[db]);
            }
        }
        // barrier B: every wave has read V^T of tile t and this wave's band piece of tile t + 1 has landed
        if (more) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (loader) *reinterpret_cast<uint4*>(lds_vt + k_off(trow, tch)) = pvt;
        }
    }
    if (qvalid) {
        const float inv = 1.0f / l_run;
        const int q = q0 + c;
        if (O32) {
            float* orow = reinterpret_cast<float*>(O) + ((size_t)b * T + q) * (H * HD) + h * HD + 4 * g;
#pragma unroll
            for (int db = 0; db < 4; ++db)
                *reinterpret_cast<float4*>(orow + 16 * db) = make_float4(o[db][0] * inv, o[db][1] * inv, o[db][2] * inv, o[db][3] * inv);
            if (Osplit != nullptr) {    // split-precision operand image of the same values, [hi | lo | hi] over 3 H 64 columns (out_proj's A operand)
                const int Dm = H * HD;
                bf16_t* srow = Osplit + ((size_t)b * T + q) * (3 * Dm) + h * HD + 4 * g;
#pragma unroll
                for (int db = 0; db < 4; ++db) {
                    bf16_t hh[4], ll[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const float v = o[db][e] * inv; hh[e] = f2h(v); ll[e] = f2h(v - h2f(hh[e])); }
                    const uint2 hi = make_uint2((unsigned)hh[0] | ((unsigned)hh[1] << 16), (unsigned)hh[2] | ((unsigned)hh[3] << 16));
                    const uint2 lo = make_uint2((unsigned)ll[0] | ((unsigned)ll[1] << 16), (unsigned)ll[2] | ((unsigned)ll[3] << 16));
                    *reinterpret_cast<uint2*>(srow + 16 * db) = hi;
                    *reinterpret_cast<uint2*>(srow + Dm + 16 * db) = lo;
                    *reinterpret_cast<uint2*>(srow + 2 * Dm + 16 * db) = hi;
                }
            }
        } else {
            bf16_t* orow = O + ((size_t)b * T + q) * (H * HD) + h * HD + 4 * g;
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                uint2 pk2;
                pk2.x = pack2<F16>(o[db][0] * inv, o[db][1] * inv);
                pk2.y = pack2<F16>(o[db][2] * inv, o[db][3] * inv);
                *reinterpret_cast<uint2*>(orow + 16 * db) = pk2;
            }
        }
        if (g == 0 && LSE != nullptr) LSE[(size_t)bh * T + q] = m_run + log2f(l_run);
    }
}

template <bool F16, bool O32, int NW>
static void launch_relpos_fwd(const void* Qu, const void* Qv, const void* K, const void* Vt, const void* P, void* O, void* Os, float* LSE, int B,
                              int H, int T, int Tpad, int Rpad, hipStream_t stream) {
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)relpos_fwd_kernel<F16, O32, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, FW16_LDS(NW)); attr = true; }
    hipLaunchKernelGGL((relpos_fwd_kernel<F16, O32, NW>), dim3(cdiv(T, 16 * NW), B * H), dim3(64 * NW), FW16_LDS(NW), stream, (const bf16_t*)Qu,
                       (const bf16_t*)Qv, (const bf16_t*)K, (const bf16_t*)Vt, (const bf16_t*)P, (bf16_t*)O, (bf16_t*)Os, LSE, T, Tpad, H, Rpad);
}
extern "C" int sed_relpos_attn_fwd(const void* Qu, const void* Qv, const void* K, const void* Vt, const void* P,
                                   void* O, void* O_split, float* LSE, int B, int H, int T, int Tpad, int Rpad, int f16, int o_f32,
                                   hipStream_t stream) {
    (void)hipGetLastError();
    if (T <= 0 || (T % 8) || Tpad % 64 || Tpad < T || Rpad % 64 || Rpad < 2 * T - 1 || (O_split != nullptr && !o_f32)) return SED_ERR_ARG;
    // 16 waves (256 queries) per workgroup when the sequence is long enough to fill them: four waves per SIMD, K / V^T tiles staged once per
    // 256 queries (the 8-wave form serves short sequences)
    const bool big = T > 128;
#define SED_RP_FWD(F, O32) { if (big) launch_relpos_fwd<F, O32, 16>(Qu, Qv, K, Vt, P, O, O_split, LSE, B, H, T, Tpad, Rpad, stream); \
                             else launch_relpos_fwd<F, O32, 8>(Qu, Qv, K, Vt, P, O, O_split, LSE, B, H, T, Tpad, Rpad, stream); }
    if (f16 && o_f32) SED_RP_FWD(true, true)
    else if (f16) SED_RP_FWD(true, false)
    else if (o_f32) SED_RP_FWD(false, true)
    else SED_RP_FWD(false, false)
#undef SED_RP_FWD
    return sed_check_launch();
}

// ---------------------------------------------------------------------------------------------------
// backward kernel 1: dK, dV   (workgroup = 128 keys; loops over 64-query tiles; lane owns a key column)
// ---------------------------------------------------------------------------------------------------
template <bool SF16>
__global__ __launch_bounds__(256) void relpos_bwd_dkdv_kernel(
    const bf16_t* __restrict__ Qu, const bf16_t* __restrict__ Qut, const bf16_t* __restrict__ Qv,
    const bf16_t* __restrict__ K, const bf16_t* __restrict__ V, const bf16_t* __restrict__ P,
    const bf16_t* __restrict__ dOh, const bf16_t* __restrict__ dOt, const float* __restrict__ LSE,
    const float* __restrict__ Dv, bf16_t* __restrict__ dqkv, int T, int Tpad, int H, int Rpad) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[5][KVB * 128];  // Qu rows, Qv rows, dO rows, Qu^T, dO^T
    __shared__ __attribute__((aligned(16))) unsigned char lds_band[BAND_ROWS * 128];
    __shared__ __attribute__((aligned(16))) float lds_g[4][32 * 65];
    __shared__ __attribute__((aligned(16))) float lstat[2][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31, lg = lane >> 5;
    const int bh = blockIdx.y, b = bh / H, h = bh - b * H;
    const int J0 = blockIdx.x * 128, key0 = J0 + wave * 32;
    const int R = 2 * T - 1;
    const size_t hb = (size_t)bh * T * HD, hbt = (size_t)bh * HD * Tpad;
    const bf16_t* Ph = P + (size_t)h * Rpad * HD;
    int krow = key0 + lr;
    const bool key_valid_lane = krow < T;
    krow = krow < T ? krow : T - 1;
    s16x8_t kf[4], vf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        kf[s] = *reinterpret_cast<const s16x8_t*>(K + hb + (size_t)krow * HD + 16 * s + 8 * lg);
        vf[s] = *reinterpret_cast<const s16x8_t*>(V + hb + (size_t)krow * HD + 16 * s + 8 * lg);
    }
    f32x16_t dk[2], dv[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[i][r] = 0.f; dv[i][r] = 0.f; }
    float* gs = lds_g[wave];
    const int ntiles = (T + 63) / 64;
    // next tile's operands travel HBM -> registers while the current tile is being consumed, registers -> LDS afterwards
    TileRegs r0, r1, r2, r3, r4;
    BandRegs rb;
    float rstat = 0.f;
    auto gload = [&](int t) {
        const int i0 = t * 64;
        tile_gload(r0, Qu + hb, i0, T, HD, 0, tid);
        tile_gload(r1, Qv + hb, i0, T, HD, 0, tid);
        tile_gload(r2, dOh + hb, i0, T, HD, 0, tid);
        tile_gload(r3, Qut + hbt, 0, HD, Tpad, i0, tid);
        tile_gload(r4, dOt + hbt, 0, HD, Tpad, i0, tid);
        band_gload(rb, Ph, J0 - (i0 + 63) + T - 1, R, tid);
        if (tid < 128) {
            const int qi = i0 + (tid & 63);
            const float* src = (tid < 64) ? LSE : Dv;
            rstat = qi < T ? src[(size_t)bh * T + qi] : (tid < 64 ? 1e30f : 0.f);  // L2 = +big -> P = 0 for padded queries
        }
    };
    auto lstore = [&]() {
        tile_lstore_rows(r0, lds[0], tid);
        tile_lstore_rows(r1, lds[1], tid);
        tile_lstore_rows(r2, lds[2], tid);
        tile_lstore_cols(r3, lds[3], tid);
        tile_lstore_cols(r4, lds[4], tid);
        band_lstore(rb, lds_band, tid);
        if (tid < 128) lstat[tid >> 6][tid & 63] = rstat;
    };
    gload(0);
    lstore();
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        if (t + 1 < ntiles) gload(t + 1);
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            const int rowoff = 32 * (wave - qb + 1);
            // band product G[i, rho] (rows = queries, column = rho): A = Qv rows, B = band rows
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                f32x16_t g;
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    g = mfma32t<SF16>(lds_frag_rows(lds[1], 32 * qb + lr, 2 * s + lg),
                                      lds_frag_rows(lds_band, rowoff + 32 * blk + lr, 2 * s + lg), s == 0 ? zero16 : g);
#pragma unroll
                for (int r = 0; r < 16; ++r) gs[mfma32_row(r, lg) * 65 + 32 * blk + lr] = g[r];
            }
            f32x16_t s_, dp;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                s_ = mfma32t<SF16>(lds_frag_rows(lds[0], 32 * qb + lr, 2 * s + lg), kf[s], s == 0 ? zero16 : s_);
                dp = mfma32(lds_frag_rows(lds[2], 32 * qb + lr, 2 * s + lg), vf[s], s == 0 ? zero16 : dp);
            }
            __syncthreads();
            // (a lane whose key is >= T needs no masking: its columns only feed dK / dV rows that are never stored)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const int qq = 32 * qb + 8 * qd + 4 * lg;
                const f32x4_t l2 = *reinterpret_cast<const f32x4_t*>(&lstat[0][qq]);
                const f32x4_t dd = *reinterpret_cast<const f32x4_t*>(&lstat[1][qq]);
#pragma unroll
                for (int j = 0; j < 4; j += 2) {
                    const int r = 4 * qd + j, ii = mfma32_row(r, lg);
                    const f32x2_t bd = {gs[ii * 65 + lr - ii + 31], gs[(ii + 1) * 65 + lr - ii + 30]};
                    const f32x2_t c2 = {SCALE_LOG2E, SCALE_LOG2E}, nl = {-l2[j], -l2[j + 1]}, nd = {-dd[j], -dd[j + 1]};
                    f32x2_t x = {s_[r], s_[r + 1]}, d2 = {dp[r], dp[r + 1]};
                    x = __builtin_elementwise_fma(x + bd, c2, nl);
                    const f32x2_t pv = {__builtin_amdgcn_exp2f(x.x), __builtin_amdgcn_exp2f(x.y)};
                    d2 = pv * (d2 + nd);
                    s_[r] = pv.x; s_[r + 1] = pv.y;
                    dp[r] = d2.x; dp[r + 1] = d2.y;
                }
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const s16x8_t pf = pack_frag(s_, s), dsf = pack_frag(dp, s);
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    dv[db] = mfma32(pf, lds_frag_cols(lds[4], 32 * db + lr, 8 * qb + 4 * s + lg), dv[db]);
                    dk[db] = mfma32(dsf, lds_frag_cols(lds[3], 32 * db + lr, 8 * qb + 4 * s + lg), dk[db]);
                }
            }
            __syncthreads();
        }
        if (t + 1 < ntiles) {
            lstore();
            __syncthreads();
        }
    }
    const int ldq = 3 * H * HD;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = key0 + mfma32_row(r, lg);
            if (key < T) {
                bf16_t* row = dqkv + ((size_t)b * T + key) * ldq + h * HD + 32 * db + lr;
                row[H * HD] = f2bf(dk[db][r] * SCALE);
                row[2 * H * HD] = f2bf(dv[db][r]);
            }
        }
}

// ---------------------------------------------------------------------------------------------------
// backward kernel 2: dQ (= dQu + dQv), per-head sums for pos_bias_u / pos_bias_v, and dS^T for the dP kernel, on
// v_mfma_f32_16x16x32 with 8 waves (2 per SIMD) per workgroup.
//   workgroup = 128 queries, wave = 16 queries x 64-key tiles, lane = (query column c = lane & 15, row group g = lane >> 4).
//   146 VGPRs per wave (the first version, 4 waves x 32 queries on 32x32x16 MFMAs, needed 441, ran one wave per SIMD and had to park
//   its register prefetch in AGPRs behind an s_waitcnt vmcnt(0) right after issuing it); the positional band never passes through
//   registers:
//     * P rows (score recompute, S type) live in a 256-row LDS ring, P^T columns (bf16, dQv) in four 64-column panels; the 64 new
//       rows / columns a tile needs are DMA'd (buffer_load ... lds) into the slot the previous tile retired, one 1-KiB piece per wave
//     * K rows, V rows and K^T (64 d x 64 keys) are single-buffered and prefetched through 12 VGPRs per lane
//     * wave-private LDS: G [16 q][85] fp32 for the skew (read back as aligned 16-byte rows), dG^T [16 q][96 rho] bf16 (zero outside
//       the band, set once)
//   K / V rows are permuted inside each 32-key group so that lane group g's accumulator rows of two neighbouring 16-key blocks are the
//   8 consecutive keys 32 ks + 8 g .. + 7: dS^T goes from the accumulators to the B operand of the dQu contraction without leaving
//   the lane, and its A operand is one 16-byte read of the K^T tile.
// ---------------------------------------------------------------------------------------------------

#define DQ16_WL 8576                      // per-wave LDS: G [16 q][85] fp32 (5440 B) + dG^T [16 q][96 rho] bf16 (3072 B)
#define DQ16_LDS (24576 + 65536 + 8 * DQ16_WL)
template <bool SF16>
__global__ __launch_bounds__(512) void relpos_bwd_dq_kernel(
    const bf16_t* __restrict__ Qu, const bf16_t* __restrict__ Qv, const bf16_t* __restrict__ K,
    const bf16_t* __restrict__ Kt, const bf16_t* __restrict__ V, const bf16_t* __restrict__ P,
    const bf16_t* __restrict__ Pt, const bf16_t* __restrict__ dOh, const float* __restrict__ LSE,
    const float* __restrict__ Dv, bf16_t* __restrict__ dqkv, bf16_t* __restrict__ dSt, bf16_t* __restrict__ Pst, float* __restrict__ du,
    float* __restrict__ dvb, int T, int Tpad, int H, int Rpad) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds_dq[];
    unsigned char (*lds_kv)[KVB * 128] = reinterpret_cast<unsigned char (*)[KVB * 128]>(lds_dq);                    // K rows, V rows, K^T rows (d)
    unsigned char* lds_band = lds_dq + 3 * KVB * 128;                                                               // ring of P rows, slot = n & 255
    unsigned char (*lds_bandT)[64 * 128] = reinterpret_cast<unsigned char (*)[64 * 128]>(lds_dq + 24576 + 32768);  // P^T panels [64 d][64 rho]
    unsigned char (*lds_w)[DQ16_WL] = reinterpret_cast<unsigned char (*)[DQ16_WL]>(lds_dq + 24576 + 65536);        // per wave: G fp32 + dG^T bf16
    const int tid = threadIdx.x, lane = tid & 63, c = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // grid (query blocks, batch x head): with 8 query blocks the hardware's round-robin puts query block i of every head on XCD i, so
    // one L2 serves a fixed slice of the positional table (an XCD-contiguous order that keeps a head's K / V in one L2 instead
    // measured 18 % slower)
    const int bh = blockIdx.y, b = bh / H, h = bh - b * H;
    const int I0 = blockIdx.x * 128, q0 = I0 + wave * 16;
    const int R = 2 * T - 1;
    const size_t hb = (size_t)bh * T * HD, hbt = (size_t)bh * HD * Tpad;
    const int RB0 = T - 128 - I0;          // global P row of band index n = 0 (multiple of 8: T % 8 == 0)
    // rows outside [0, R) of the head's P slab read as zeros, columns of P^T outside the slab likewise (inside it they are finite
    // and only ever multiply exact zeros of dG^T)
    const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void*)(P + (size_t)h * Rpad * HD), 0, R * HD * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rpt = __builtin_amdgcn_make_buffer_rsrc((void*)(Pt + (size_t)h * HD * Rpad), 0, HD * Rpad * 2, 0x00020000);
    const int prow = lane >> 3, pch = lane & 7;
    // band piece of this wave for band rows n0 .. n0 + 7 (n0 % 8 == 0) and P^T piece for d rows 8 wave .. + 7, columns n0 .. n0 + 63
    auto dma_band = [&](int n0) {
        const int slot = (n0 & 255) + prow;
        const int vo = (RB0 + n0 + prow) * (HD * 2) + ((pch ^ ((slot >> 1) & 7)) << 4);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rp, (rp_lds_ptr_t)(lds_band + (n0 & 255) * 128), 16, vo, 0, 0, 0);
    };
    auto dma_bandT = [&](int n0) {
        const int d = 8 * wave + prow;
        const int vo = (d * Rpad + RB0 + n0) * 2 + ((pch ^ ((d >> 1) & 7)) << 4);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rpt, (rp_lds_ptr_t)(lds_bandT[(n0 >> 6) & 3] + 8 * wave * 128), 16, vo, 0, 0, 0);
    };
    // ---- prologue: the whole ring (256 rows / 4 panels), tile 0 of K / V / K^T, query-side fragments
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        dma_band(64 * i + 8 * wave);
        dma_bandT(64 * i);
    }
    const int trow = tid >> 3, tch = tid & 7;   // this thread's 16-byte chunk of a [64][64] tile
    uint4 pk, pv, pkt;
    auto gload = [&](int t) {
        const int j0 = t * KVB;
        int kr = j0 + trow;
        kr = kr < T ? kr : T - 1;
        pk = *reinterpret_cast<const uint4*>(K + hb + (size_t)kr * HD + tch * 8);
        pv = *reinterpret_cast<const uint4*>(V + hb + (size_t)kr * HD + tch * 8);
        pkt = *reinterpret_cast<const uint4*>(Kt + hbt + (size_t)trow * Tpad + j0 + tch * 8);
    };
    // K / V rows sit in LDS in the order the 16-row MFMA blocks want them: block kb = 2 (key >> 5) + ((key >> 2) & 1) holds the keys
    // 32 (kb >> 1) + 8 G + 4 (kb & 1) + r at row 4 G + r, so that lane group G's accumulator rows of blocks 2 ks and 2 ks + 1 together
    // are the 8 consecutive keys 32 ks + 8 G .. + 7 -- the K^T operand of the dQu contraction is then ONE 16-byte read per lane
    const int krow_lds = (trow & 32) + (((trow >> 2) & 1) << 4) + (((trow >> 3) & 3) << 2) + (trow & 3);
    auto lstore_kv = [&]() {
        const int offp = k_off(krow_lds, tch);
        *reinterpret_cast<uint4*>(lds_kv[0] + offp) = pk;
        *reinterpret_cast<uint4*>(lds_kv[1] + offp) = pv;
    };
    auto lstore_kt = [&]() { *reinterpret_cast<uint4*>(lds_kv[2] + k_off(trow, tch)) = pkt; };
    gload(0);
    int qrow = q0 + c;
    const bool qvalid = qrow < T;
    qrow = qvalid ? qrow : T - 1;
    s16x8_t quf[2], qvf[2], dof[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        quf[ks] = *reinterpret_cast<const s16x8_t*>(Qu + hb + (size_t)qrow * HD + 32 * ks + 8 * g);
        qvf[ks] = *reinterpret_cast<const s16x8_t*>(Qv + hb + (size_t)qrow * HD + 32 * ks + 8 * g);
        dof[ks] = *reinterpret_cast<const s16x8_t*>(dOh + hb + (size_t)qrow * HD + 32 * ks + 8 * g);
    }
    const float l2 = qvalid ? LSE[(size_t)bh * T + qrow] : __builtin_inff(), dd = Dv[(size_t)bh * T + qrow];
    unsigned char* wl = lds_w[wave];
    // G[q][rho] fp32 at word 85 q + 1 + rho: the four band values a lane adds to one accumulator (rho = jj - q + 15 .. + 3) start at
    // word 84 q + 16 + jj, a multiple of 4 -> one aligned 16-byte read
    float* gs = reinterpret_cast<float*>(wl);
    unsigned char* dgl = wl + 5440;                                // dG^T [16 q][96 rho] bf16
#pragma unroll
    for (int i = 0; i < 3; ++i) *reinterpret_cast<uint4*>(dgl + (i * 64 + lane) * 16) = make_uint4(0, 0, 0, 0);
    lstore_kv();
    lstore_kt();
    f32x4_t dqu[4], dqv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { dqu[i] = zero4; dqv[i] = zero4; }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // tell the compiler's waitcnt pass that the prologue loads have arrived (it would otherwise wait for them at their first use
    // inside the loop, behind -- and therefore also for -- the tile prefetch issued at the top of the iteration)
    asm volatile("" ::"v"(quf[0]), "v"(quf[1]), "v"(qvf[0]), "v"(qvf[1]), "v"(dof[0]), "v"(dof[1]), "v"(l2), "v"(dd));
    __syncthreads();
    const int ntiles = (T + KVB - 1) / KVB;
    // dS^T slab of this (batch, head): [Tpad keys][Tpad queries] bf16; the 128-query block can overhang Tpad (a multiple of 64)
    const __amdgpu_buffer_rsrc_t rds = __builtin_amdgcn_make_buffer_rsrc((void*)(dSt + (size_t)bh * Tpad * Tpad), 0, Tpad * Tpad * 2, 0x00020000);
    // P^T slab, same layout (nullable): with it the dK / dV of this layer are two plain contractions over the queries of the stored
    // dS^T and P^T (relpos_bwd_dkdv_stream_kernel) instead of a second recomputation of the scores
    const bool store_p = Pst != nullptr;
    const __amdgpu_buffer_rsrc_t rps = __builtin_amdgcn_make_buffer_rsrc((void*)((store_p ? Pst : dSt) + (size_t)bh * Tpad * Tpad), 0,
                                                                        store_p ? Tpad * Tpad * 2 : 0, 0x00020000);
    // slab stores: neighbouring query lanes (c, c ^ 1) trade halves so that every lane stores 4-byte words (two queries of one key row):
    // the even lane takes key rows r = 0, 1 of a block, the odd lane rows 2, 3 -- half the store instructions of 2-byte stores
    const bool odd = c & 1;
    const int dvo = (q0 + c < Tpad) ? ((8 * g + (odd ? 2 : 0)) * Tpad + q0 + (c & ~1)) * 2 : 0x7ffffff0;     // (Tpad is even: a pair is in or out together)
    for (int t = 0; t < ntiles; ++t) {
        const int j0 = t * KVB;
        const bool more = t + 1 < ntiles;
        if (more) {   // tile t + 1: band rows / P^T columns n in [64 t + 192, 64 t + 256) replace the slot tile t - 1 retired
            dma_band(64 * t + 192 + 8 * wave);
            dma_bandT(64 * t + 192);
            gload(t + 1);
        }
        const int nb = 64 * t + 16 * (7 - wave);    // this wave's band base
        // ---- G^T[rho, q] = P_band[rho, :] . Qv[q, :]  (5 blocks of 16 rho) -> wave-private LDS
        // (round 4: the operand fragments of a whole phase are requested before its first MFMA.  With two waves per SIMD -- the LDS footprint
        //  allows no more -- a read issued right before the MFMA that consumes it exposed the LDS latency ~30 times per tile; at this
        //  occupancy the kernel has 110 registers of headroom: 146 -> 217 VGPRs, 881 -> 852 us per launch)
        {
            s16x8_t bfr[5][2];
#pragma unroll
            for (int blk = 0; blk < 5; ++blk) {
                const int slot = ((nb + 16 * blk) & 255) + c;
                const unsigned char* rowp = lds_band + slot * 128;
                const int sw = (slot >> 1) & 7;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) bfr[blk][ks] = *reinterpret_cast<const s16x8_t*>(rowp + (((4 * ks + g) ^ sw) << 4));
            }
            __builtin_amdgcn_sched_barrier(0);
            f32x4_t gacc[5];
#pragma unroll
            for (int blk = 0; blk < 5; ++blk) {
                gacc[blk] = mfma16x<SF16>(bfr[blk][0], qvf[0], zero4);
                gacc[blk] = mfma16x<SF16>(bfr[blk][1], qvf[1], gacc[blk]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int blk = 0; blk < 5; ++blk)
#pragma unroll
                for (int r = 0; r < 4; ++r) gs[85 * c + 1 + 16 * blk + 4 * g + r] = gacc[blk][r];
        }
        // ---- S^T = K Qu^T + skew(G^T) (the skewed band term enters as the accumulator input), dP^T = V dO^T
        f32x4_t st[4], dp[4];   // block kb, register r <-> key jj = 32 (kb >> 1) + 4 (kb & 1) + 8 g + r
        const int swc = (c >> 1) & 7;
        {
            s16x8_t kfr[4][2], vfr[4][2];
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                const unsigned char* kp = lds_kv[0] + (16 * kb + c) * 128;
                const unsigned char* vp = lds_kv[1] + (16 * kb + c) * 128;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    kfr[kb][ks] = *reinterpret_cast<const s16x8_t*>(kp + (((4 * ks + g) ^ swc) << 4));
                    vfr[kb][ks] = *reinterpret_cast<const s16x8_t*>(vp + (((4 * ks + g) ^ swc) << 4));
                }
            }
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) st[kb] = *reinterpret_cast<const f32x4_t*>(gs + 84 * c + 16 + 32 * (kb >> 1) + 4 * (kb & 1) + 8 * g);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                dp[kb] = zero4;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    st[kb] = mfma16x<SF16>(kfr[kb][ks], quf[ks], st[kb]);
                    dp[kb] = mfma16x<false>(vfr[kb][ks], dof[ks], dp[kb]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // Two barriers per tile, half a tile apart: A -- every wave has read K / V of tile t, the prefetched K / V rows go to LDS;
        // B (end of tile) -- every wave has read K^T of tile t and the band pieces of tile t + 1 have landed, K^T goes to LDS and is
        // first read after the next A.
        if (more) {
            __syncthreads();
            lstore_kv();
        }
        // K^T (dQu) and P^T-panel (dQv) fragments of this tile: requested now, they arrive under the exponentials
        s16x8_t ktf[2][4], ptf[3][4];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int db = 0; db < 4; ++db)
                ktf[ks][db] = *reinterpret_cast<const s16x8_t*>(lds_kv[2] + (16 * db + c) * 128 + (((4 * ks + g) ^ swc) << 4));
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
            const int n = nb + 32 * ks + 8 * g;
            const unsigned char* pan = lds_bandT[(n >> 6) & 3];
            const int ch = (n & 63) >> 3;
#pragma unroll
            for (int db = 0; db < 4; ++db) ptf[ks][db] = *reinterpret_cast<const s16x8_t*>(pan + (16 * db + c) * 128 + ((ch ^ swc) << 4));
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- P = exp2(S c - LSE), dS^T = P (dP^T - D)
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float p = __builtin_amdgcn_exp2f(__builtin_fmaf(st[kb][r], SCALE_LOG2E, -l2));   // invalid query: LSE = +inf
                if (j0 + KVB > T) p = (j0 + 32 * (kb >> 1) + 4 * (kb & 1) + 8 * g + r < T) ? p : 0.f;   // last tile only
                dp[kb][r] = p * (dp[kb][r] - dd);
                st[kb][r] = p;
            }
        // ---- dS^T -> skewed dG^T image (wave-private) and -> global for the dP kernel
        {
            unsigned short* dg16 = reinterpret_cast<unsigned short*>(dgl) + c * 96 + 15 - c + 8 * g;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                const unsigned p01 = pack2bf(dp[kb][0], dp[kb][1]), p23 = pack2bf(dp[kb][2], dp[kb][3]);
                const int jb = 32 * (kb >> 1) + 4 * (kb & 1);
                dg16[jb + 0] = (unsigned short)(p01 & 0xffffu);
                dg16[jb + 1] = (unsigned short)(p01 >> 16);
                dg16[jb + 2] = (unsigned short)(p23 & 0xffffu);
                dg16[jb + 3] = (unsigned short)(p23 >> 16);
                // branch-free: a lane whose query column does not exist in the [Tpad][Tpad] slab stores out of the buffer's bounds
                const int so = (j0 + jb) * Tpad * 2;
                {
                    const unsigned recv = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(odd ? p01 : p23), 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
                    const unsigned lo = odd ? recv : p01, hi = odd ? p23 : recv;      // the even query's pair of keys, the odd query's
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_amdgcn_perm(hi, lo, 0x05040100u), rds, dvo, so, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_amdgcn_perm(hi, lo, 0x07060302u), rds, dvo, so + Tpad * 2, 0);
                }
                {   // (without a P^T slab the resource has zero records: every lane stores out of bounds)
                    const unsigned q01 = pack2bf(st[kb][0], st[kb][1]), q23 = pack2bf(st[kb][2], st[kb][3]);
                    const unsigned recv = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(odd ? q01 : q23), 0xB1, 0xF, 0xF, true);
                    const unsigned lo = odd ? recv : q01, hi = odd ? q23 : recv;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_amdgcn_perm(hi, lo, 0x05040100u), rps, dvo, so, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_amdgcn_perm(hi, lo, 0x07060302u), rps, dvo, so + Tpad * 2, 0);
                }
            }
        }
        // ---- dQu^T[d, q] += K^T[d, key] dS^T[key, q]
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const uint4 dsu = make_uint4(pack2bf(dp[2 * ks][0], dp[2 * ks][1]), pack2bf(dp[2 * ks][2], dp[2 * ks][3]),
                                         pack2bf(dp[2 * ks + 1][0], dp[2 * ks + 1][1]), pack2bf(dp[2 * ks + 1][2], dp[2 * ks + 1][3]));
            const s16x8_t dsf = __builtin_bit_cast(s16x8_t, dsu);
#pragma unroll
            for (int db = 0; db < 4; ++db) dqu[db] = mfma16x<false>(ktf[ks][db], dsf, dqu[db]);      // ((row >> 1) & 7 == swc for every 16-row block)
        }
        // ---- dQv^T[d, q] += P^T[d, rho] dG^T[rho, q]   (96 rho slots, the last 16 and the cells outside the band are zeros)
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
            const s16x8_t gf = *reinterpret_cast<const s16x8_t*>(dgl + c * 192 + 64 * ks + 16 * g);
#pragma unroll
            for (int db = 0; db < 4; ++db) dqv[db] = mfma16x<false>(ptf[ks][db], gf, dqv[db]);
        }
        if (more) {
            asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            // this wave's band pieces have landed (only the 8 dS^T and 8 P^T stores are younger)
            __syncthreads();
            lstore_kt();
        }
    }
    // ---- outputs: dq rows (lane = query, 4 consecutive d per block) and the pos_bias_u / pos_bias_v column sums
    if (qvalid) {
        bf16_t* row = dqkv + ((size_t)b * T + q0 + c) * (3 * H * HD) + h * HD + 4 * g;
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            uint2 o;
            o.x = pack2bf((dqu[db][0] + dqv[db][0]) * SCALE, (dqu[db][1] + dqv[db][1]) * SCALE);
            o.y = pack2bf((dqu[db][2] + dqv[db][2]) * SCALE, (dqu[db][3] + dqv[db][3]) * SCALE);
            *reinterpret_cast<uint2*>(row + 16 * db) = o;
        }
    }
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float su = dqu[db][r] * SCALE, sv = dqv[db][r] * SCALE;   // invalid queries contributed exact zeros
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) {
                su += __shfl_xor(su, o, 64);
                sv += __shfl_xor(sv, o, 64);
            }
            dqu[db][r] = su; dqv[db][r] = sv;
        }
    if (c == 0) {
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                unsafeAtomicAdd(&du[h * HD + 16 * db + 4 * g + r], dqu[db][r]);
                unsafeAtomicAdd(&dvb[h * HD + 16 * db + 4 * g + r], dqv[db][r]);
            }
    }
}

// ---------------------------------------------------------------------------------------------------
// backward kernel 1' (round 3): dK and dV from the dS^T / P^T slabs the dQ kernel stored -- two contractions over the queries,
//   dK^T[d, key] = scale * sum_q Qu^T[d, q] dS^T[key, q]        dV^T[d, key] = sum_q dO^T[d, q] P^T[key, q]
// with nothing recomputed (the first-generation kernel above recomputes the content scores, the band product, its skew, the
// exponentials and dP a second time: 796 us, 286 VGPRs).  A streaming kernel: per (batch, head) it reads the two [Tpad][Tpad] bf16 slabs
// once (1.6 GB per layer at B = 32) and is bound by that.
//   workgroup = 8 waves x 16 keys; MFMA rows = d (Qu^T / dO^T tiles [64 d][64 q] in LDS, shared by the waves), MFMA columns = keys:
//   lane (c, g) reads the 16 bytes q = 32 ks + 8 g .. + 7 of its key's slab row straight from global memory into the B operand, one
//   tile ahead.  Output lane = key column, 4 consecutive d per block -> 8-byte stores.
// ---------------------------------------------------------------------------------------------------
// (round 5: a rotated tile order per workgroup changes nothing -- not channel hot-spotting; three workgroups per CU instead of two
//  (80 VGPRs) run 515-537 us against 421-427: more bytes in flight do not help, the slab walk -- 128 bytes from each of 128 rows 2 KiB
//  apart per step -- is what the memory system delivers at 3.8 TB/s)
__global__ __launch_bounds__(512) void relpos_bwd_dkdv_stream_kernel(const bf16_t* __restrict__ Qut, const bf16_t* __restrict__ dOt,
                                                                     const bf16_t* __restrict__ dSt, const bf16_t* __restrict__ Pst,
                                                                     bf16_t* __restrict__ dqkv, int T, int Tpad, int H) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2][2][64 * 128];     // [stage][Qu^T | dO^T][d rows x 64 q]
    const int tid = threadIdx.x, lane = tid & 63, c = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bh = blockIdx.y, b = bh / H, h = bh - b * H;
    const int key = blockIdx.x * 128 + wave * 16 + c;
    const int krow = key < Tpad ? key : Tpad - 1;          // (a key block may overhang Tpad; keys >= T are not stored)
    const size_t hbt = (size_t)bh * HD * Tpad, sb = (size_t)bh * Tpad * Tpad + (size_t)krow * Tpad + 8 * g;
    const int trow = tid >> 3, tch = tid & 7;              // this thread's 16-byte chunk of a [64 d][64 q] tile
    const int ntiles = Tpad / 64;
    uint4 ta, tb;
    s16x8_t fs[2], fp[2];
    // (one tile of slab rows in flight per wave; two were slower: 424 vs 402 us)
    auto gload = [&](int t) {
        ta = *reinterpret_cast<const uint4*>(Qut + hbt + (size_t)trow * Tpad + 64 * t + 8 * tch);
        tb = *reinterpret_cast<const uint4*>(dOt + hbt + (size_t)trow * Tpad + 64 * t + 8 * tch);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            fs[ks] = __builtin_nontemporal_load(reinterpret_cast<const s16x8_t*>(dSt + sb + 64 * t + 32 * ks));
            fp[ks] = __builtin_nontemporal_load(reinterpret_cast<const s16x8_t*>(Pst + sb + 64 * t + 32 * ks));
        }
    };
    f32x4_t dk[4], dv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { dk[i] = zero4; dv[i] = zero4; }
    gload(0);
    *reinterpret_cast<uint4*>(lds[0][0] + k_off(trow, tch)) = ta;
    *reinterpret_cast<uint4*>(lds[0][1] + k_off(trow, tch)) = tb;
    __syncthreads();
    const int swc = (c >> 1) & 7;
    for (int t = 0; t < ntiles; ++t) {
        const int cur = t & 1;
        s16x8_t bs[2], bp[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) { bs[ks] = fs[ks]; bp[ks] = fp[ks]; }
        if (t + 1 < ntiles) gload(t + 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                const int ro = (16 * db + c) * 128 + (((4 * ks + g) ^ swc) << 4);
                dk[db] = mfma16x<false>(*reinterpret_cast<const s16x8_t*>(lds[cur][0] + ro), bs[ks], dk[db]);
                dv[db] = mfma16x<false>(*reinterpret_cast<const s16x8_t*>(lds[cur][1] + ro), bp[ks], dv[db]);
            }
        if (t + 1 < ntiles) {
            *reinterpret_cast<uint4*>(lds[cur ^ 1][0] + k_off(trow, tch)) = ta;     // the other stage: last read before the previous barrier
            *reinterpret_cast<uint4*>(lds[cur ^ 1][1] + k_off(trow, tch)) = tb;
            __syncthreads();
        }
    }
    if (key < T) {
        bf16_t* row = dqkv + ((size_t)b * T + key) * (3 * H * HD) + h * HD + 4 * g;
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            uint2 o;
            o.x = pack2bf(dk[db][0] * SCALE, dk[db][1] * SCALE);
            o.y = pack2bf(dk[db][2] * SCALE, dk[db][3] * SCALE);
            *reinterpret_cast<uint2*>(row + H * HD + 16 * db) = o;
            o.x = pack2bf(dv[db][0], dv[db][1]);
            o.y = pack2bf(dv[db][2], dv[db][3]);
            *reinterpret_cast<uint2*>(row + 2 * H * HD + 16 * db) = o;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// backward kernel 3: dP[r, h*64 + d] += scale * sum_{b, i} dS_b[i, i + r - (T-1)] * Qv_b[i, d]
//   grid (Rpad/64, H, Bsplit); each workgroup: one 64-row block of r, loops over its batch slice and all i tiles.
//   dS^T tile staged with a 33-word row stride so that the diagonal gather is bank-conflict free.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void relpos_bwd_dp_kernel(const bf16_t* __restrict__ dSt, const bf16_t* __restrict__ Qvt,
                                                            float* __restrict__ dP, int B, int T, int Tpad, int H,
                                                            int ldp, int nrb, int bsplit) {
    __shared__ __attribute__((aligned(16))) unsigned int lds_s[2][128 * 33];
    __shared__ __attribute__((aligned(16))) unsigned char lds_q[2][KVB * 128];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31, lg = lane >> 5;
    // XCD-contiguous order with the r block fastest: neighbouring r blocks of one (head, batch slice) read overlapping key rows of the
    // same dS^T tiles (128 rows for 64 diagonals), so they should share an L2
    const int L = xcd_remap(blockIdx.x, nrb * H * bsplit);
    const int xb = L % nrb, h = (L / nrb) % H, zb = L / (nrb * H);
    const int R0 = xb * 64;
    const int rb = wave >> 1, db = wave & 1;
    const int bper = (B + bsplit - 1) / bsplit;
    const int b_begin = zb * bper, b_end = min(B, b_begin + bper);
    f32x16_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // query tiles whose key rows jbase .. jbase + 127 (jbase = 64 t + R0 - (T - 1)) touch the score matrix: a contiguous range of t
    const int ntiles = (T + 63) / 64;
    int t_lo = (T - 1 - R0 - 127 + 63) >> 6, t_hi = (2 * T - 2 - R0) >> 6;   // jbase + 127 >= 0, jbase <= T - 1
    t_lo = t_lo < 0 ? 0 : t_lo;
    t_hi = t_hi > ntiles - 1 ? ntiles - 1 : t_hi;
    const int nt = t_hi - t_lo + 1;
    const int total = (nt > 0 && b_end > b_begin) ? nt * (b_end - b_begin) : 0;
    // the next (batch, query tile) pair travels HBM -> registers while the current one is multiplied, registers -> the other LDS stage
    // afterwards (one barrier per pair; the first version loaded, stored and multiplied in sequence and spent 77 % of its wave
    // cycles waiting)
    uint4 pv[4];
    TileRegs rq;
    auto gload = [&](int it) {
        const int bb = it / nt, t = t_lo + (it - bb * nt);
        const int bh = (b_begin + bb) * H + h, i0 = t * 64, jbase = i0 + R0 - (T - 1);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int j = jbase + (tid >> 3) + 32 * i;
            const int jc = j < 0 ? 0 : (j < T ? j : T - 1);
            const uint4 v = *reinterpret_cast<const uint4*>(dSt + ((size_t)bh * Tpad + jc) * Tpad + i0 + (tid & 7) * 8);
            const bool ok = j >= 0 && j < T;
            pv[i] = make_uint4(ok ? v.x : 0u, ok ? v.y : 0u, ok ? v.z : 0u, ok ? v.w : 0u);
        }
        tile_gload(rq, Qvt + (size_t)bh * HD * Tpad, 0, HD, Tpad, i0, tid);
    };
    auto lstore = [&](int st) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            unsigned int* d = &lds_s[st][((tid >> 3) + 32 * i) * 33 + (tid & 7) * 4];
            d[0] = pv[i].x; d[1] = pv[i].y; d[2] = pv[i].z; d[3] = pv[i].w;
        }
        tile_lstore_rows(rq, lds_q[st], tid);
    };
    if (total > 0) {
        gload(0);
        lstore(0);
    }
    __syncthreads();
    for (int it = 0; it < total; ++it) {
        const int st = it & 1;
        if (it + 1 < total) gload(it + 1);
        const unsigned short* ls16 = reinterpret_cast<const unsigned short*>(lds_s[st]);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            s16x8_t af;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int ii = 16 * s + 8 * lg + e, rr = 32 * rb + lr;
                af[e] = (short)ls16[(ii + rr) * 66 + ii];
            }
            acc = mfma32(af, lds_frag_rows(lds_q[st], 32 * db + lr, 2 * s + lg), acc);
        }
        if (it + 1 < total) lstore(st ^ 1);
        __syncthreads();
    }
    const int R = 2 * T - 1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int rr = R0 + 32 * rb + mfma32_row(r, lg);
        if (rr < R) unsafeAtomicAdd(&dP[(size_t)rr * ldp + h * HD + 32 * db + lr], acc[r] * SCALE);
    }
}

// the pre-pass (D = rowsum(dO * O), head-split dO copies) is shared with the encoder attention
extern "C" int sed_mhsa_bwd_prep(const void* dO, const void* O, float* Dtmp, void* dOh, void* dOt, int B, int H, int N,
                                 int Npad, int o_f16, hipStream_t stream);

extern "C" int sed_relpos_attn_bwd(const void* Qu, const void* Qut, const void* Qv, const void* Qvt, const void* K,
                                   const void* Kt, const void* V, const void* P, const void* Pt, const void* O,
                                   const void* dO, const float* LSE, float* Dtmp, void* dOh, void* dOt, void* dqkv,
                                   void* dSt, void* Pst, float* dP, float* du, float* dv, int B, int H, int T, int Tpad,
                                   int Rpad, int need_param_grads, int f16, int o_kind, hipStream_t stream) {
    (void)hipGetLastError();
    // f16 != 0: Qu, Qv, K, P (score recompute) are IEEE half; Qut, Qvt, Kt, V, Pt, dO are bf16.
    // o_kind: storage type of O (0 bf16, 1 f16, 2 f32).
    if (T <= 0 || (T % 8) || Tpad % 64 || Tpad < T || Rpad % 64 || Rpad < 2 * T - 1) return SED_ERR_ARG;
    int rc = sed_mhsa_bwd_prep(dO, O, Dtmp, dOh, dOt, B, H, T, Tpad, o_kind, stream);
    if (rc) return rc;
    dim3 grid(cdiv(T, 128), B * H);
    // Pst (nullable): a second [B H, Tpad, Tpad] bf16 slab, zero outside its valid region like dSt.  With it the dQ kernel also stores
    // P^T and dK / dV come from the streaming kernel; without it the first-generation kernel recomputes the scores for them.
#define SED_LAUNCH_RP(F)                                                                                               \
    if (Pst == nullptr)                                                                                                \
        hipLaunchKernelGGL(relpos_bwd_dkdv_kernel<F>, grid, dim3(256), 0, stream, (const bf16_t*)Qu, (const bf16_t*)Qut, \
                           (const bf16_t*)Qv, (const bf16_t*)K, (const bf16_t*)V, (const bf16_t*)P, (const bf16_t*)dOh, \
                           (const bf16_t*)dOt, LSE, Dtmp, (bf16_t*)dqkv, T, Tpad, H, Rpad);                            \
    hipLaunchKernelGGL(relpos_bwd_dq_kernel<F>, grid, dim3(512), DQ16_LDS, stream, (const bf16_t*)Qu, (const bf16_t*)Qv,    \
                       (const bf16_t*)K, (const bf16_t*)Kt, (const bf16_t*)V, (const bf16_t*)P, (const bf16_t*)Pt,     \
                       (const bf16_t*)dOh, LSE, Dtmp, (bf16_t*)dqkv, (bf16_t*)dSt, (bf16_t*)Pst, du, dv, T, Tpad, H, Rpad);
    if (f16) { SED_LAUNCH_RP(true) } else { SED_LAUNCH_RP(false) }
#undef SED_LAUNCH_RP
    if (Pst != nullptr)
        hipLaunchKernelGGL(relpos_bwd_dkdv_stream_kernel, grid, dim3(512), 0, stream, (const bf16_t*)Qut, (const bf16_t*)dOt,
                           (const bf16_t*)dSt, (const bf16_t*)Pst, (bf16_t*)dqkv, T, Tpad, H);
    if (need_param_grads) {
        int bsplit = B < 8 ? B : 8;
        hipLaunchKernelGGL(relpos_bwd_dp_kernel, dim3((Rpad / 64) * H * bsplit), dim3(256), 0, stream, (const bf16_t*)dSt,
                           (const bf16_t*)Qvt, dP, B, T, Tpad, H, H * HD, Rpad / 64, bsplit);
    }
    return sed_check_launch();
}
