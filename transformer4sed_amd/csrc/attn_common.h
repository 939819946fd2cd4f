// Shared LDS tile images / fragment helpers for the attention kernels (attention.hip, relpos_attention.hip).
#pragma once
#include "common.h"

#define HD 64
#define KVB 64
#define SCALE_LOG2E 0.18033688011112042f  // (1/sqrt(64)) * log2(e)
#define SCALE 0.125f

// LDS tile images: K-like tile [64 rows][64 bf16] (128-B rows), 16-B chunks swizzled with (row>>1)&7;
// V^T-like tile [64 rows][64 bf16], 8-B chunks swizzled with (row>>1)&15.
__device__ __forceinline__ int k_off(int row, int ch16) { return row * 128 + ((ch16 ^ ((row >> 1) & 7)) << 4); }
__device__ __forceinline__ int vt_off(int row, int ch8) { return row * 128 + ((ch8 ^ ((row >> 1) & 15)) << 3); }

__device__ __forceinline__ s16x8_t lds_frag_rows(const unsigned char* base, int row, int ch16) {
    return *reinterpret_cast<const s16x8_t*>(base + k_off(row, ch16));
}
// fragment whose 8 k-elements are {c..c+3, c+8..c+11} (4-element chunk index ch8 and ch8 + 2)
__device__ __forceinline__ s16x8_t lds_frag_cols(const unsigned char* base, int row, int ch8) {
    const s16x4_t lo = *reinterpret_cast<const s16x4_t*>(base + vt_off(row, ch8));
    const s16x4_t hi = *reinterpret_cast<const s16x4_t*>(base + vt_off(row, ch8 + 2));
    s16x8_t r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}
// Row-major V tile [64 keys][64 d] (128-B rows) read through the transposing LDS load: 16-B chunk c of key row r sits at chunk
// c ^ (((r >> 1) & 1) << 2), so the four key rows of one read cycle (32 lanes = 4 rows x 64 B) fall on four different bank quarters.
__device__ __forceinline__ int v_off(int row, int ch16) { return row * 128 + ((ch16 ^ (((row >> 1) & 1) << 2)) << 4); }
// V^T operand fragment for the 32 d rows of block db and the 8 keys {c..c+3, c+8..c+11}, c = 4 ch8 (the same key pattern as
// lds_frag_cols: it matches the accumulator rows the P^T operand comes from).  ds_read_b64_tr_b16: the 16 lanes of a group supply the
// addresses of a [4 key rows][4 x 8 B] block and lane i receives column i -- 4 consecutive keys of d = 16 (group & 1) + i.
typedef short s16x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ s16x8_t lds_frag_vt(const unsigned char* base, int db, int ch8, int lane) {
    const int a = lane & 15, g16 = (lane >> 4) & 1;
    const int row = 4 * ch8 + (a >> 2);                       // (row >> 1) & 1 == (a >> 3): the key blocks start on multiples of 4
    const unsigned char* p = base + row * 128 + ((db ^ (a >> 3)) << 6) + g16 * 32 + 8 * (a & 3);
    const s16x4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4v*)p);
    const s16x4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4v*)(p + 8 * 128));
    s16x8_t r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}
// Dual-use row-major tile [64 rows][64 x 16 bit]: the SAME LDS image serves 16-byte row fragments (A / B operand with the tile's rows
// as MFMA rows) and transposing reads (operand with the tile's COLUMNS as MFMA rows).  16-byte chunk c of row r sits at chunk
// c ^ fd(r), fd(r) = (((r >> 1) & 1) << 2) | ((r >> 2) & 3): a bit permutation of (r >> 1) & 7, so the 16 rows of one ds_read_b128
// lane group still hit 16 different bank quads, while the four rows of one transposing read differ in the 64-byte half exactly as in
// v_off (PMC: zero bank conflicts for both kinds of read).
__device__ __forceinline__ int fd_sw(int row) { return (((row >> 1) & 1) << 2) | ((row >> 2) & 3); }
__device__ __forceinline__ int d_off(int row, int ch16) { return row * 128 + ((ch16 ^ fd_sw(row)) << 4); }
__device__ __forceinline__ s16x8_t lds_frag_rows_d(const unsigned char* base, int row, int ch16) {
    return *reinterpret_cast<const s16x8_t*>(base + d_off(row, ch16));
}
// transposed fragment (same contract as lds_frag_vt) out of a dual-use tile: rows 4 ch8 + (a >> 2) (+ 8 for the second read)
__device__ __forceinline__ s16x8_t lds_frag_vt_d(const unsigned char* base, int db, int ch8, int lane) {
    const int a = lane & 15, g16 = (lane >> 4) & 1;
    const int row = 4 * ch8 + (a >> 2);
    const int cl = 2 * g16 + ((a & 3) >> 1), half = (a & 3) & 1;          // 16-byte chunk inside the 64-byte half, 8-byte half of it
    const int hi64 = db ^ ((a >> 3) & 1);                                 // (row >> 1) & 1 == (a >> 3) & 1
    const int m0 = ch8 & 3, m1 = (ch8 + 2) & 3;                           // (row >> 2) & 3 for the two reads
    const unsigned char* p0 = base + row * 128 + (((4 * hi64) | (cl ^ m0)) << 4) + half * 8;
    const unsigned char* p1 = base + (row + 8) * 128 + (((4 * hi64) | (cl ^ m1)) << 4) + half * 8;
    const s16x4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4v*)p0);
    const s16x4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4v*)p1);
    s16x8_t r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}
template <bool F16>
__device__ __forceinline__ s16x8_t pack_frag_t(const f32x16_t& p, int s) {  // registers 8s..8s+7 -> 8 x 16-bit
    const uint4 r = make_uint4(pack2<F16>(p[8 * s], p[8 * s + 1]), pack2<F16>(p[8 * s + 2], p[8 * s + 3]),
                               pack2<F16>(p[8 * s + 4], p[8 * s + 5]), pack2<F16>(p[8 * s + 6], p[8 * s + 7]));
    return __builtin_bit_cast(s16x8_t, r);
}
__device__ __forceinline__ s16x8_t pack_frag(const f32x16_t& p, int s) { return pack_frag_t<false>(p, s); }

// cooperative stage of one [64][64] bf16 tile (rows r0.., row stride `ld` elements) : 256 threads x 2 x 16 B
// (named members, not an array: keeps the staged tile in VGPRs instead of scratch)
struct TileRegs { uint4 a, b; };
__device__ __forceinline__ uint4 tile_gload1(const bf16_t* src, int row_first, int nrows_valid, int ld, int col_first,
                                             int r, int c) {
    int rr = row_first + r;
    rr = rr < nrows_valid ? rr : nrows_valid - 1;
    return *reinterpret_cast<const uint4*>(src + (size_t)rr * ld + col_first + c * 8);
}
__device__ __forceinline__ void tile_gload(TileRegs& t, const bf16_t* src, int row_first, int nrows_valid, int ld,
                                           int col_first, int tid) {
    t.a = tile_gload1(src, row_first, nrows_valid, ld, col_first, tid >> 3, tid & 7);
    t.b = tile_gload1(src, row_first, nrows_valid, ld, col_first, (tid >> 3) + 32, tid & 7);
}
__device__ __forceinline__ void tile_lstore_rows(const TileRegs& t, unsigned char* dst, int tid) {
    const int r = tid >> 3, c = tid & 7;
    *reinterpret_cast<uint4*>(dst + k_off(r, c)) = t.a;
    *reinterpret_cast<uint4*>(dst + k_off(r + 32, c)) = t.b;
}
__device__ __forceinline__ void tile_lstore_vrows(const TileRegs& t, unsigned char* dst, int tid) {
    const int r = tid >> 3, c = tid & 7;
    *reinterpret_cast<uint4*>(dst + v_off(r, c)) = t.a;
    *reinterpret_cast<uint4*>(dst + v_off(r + 32, c)) = t.b;
}
__device__ __forceinline__ void tile_lstore_drows(const TileRegs& t, unsigned char* dst, int tid) {
    const int r = tid >> 3, c = tid & 7;
    *reinterpret_cast<uint4*>(dst + d_off(r, c)) = t.a;
    *reinterpret_cast<uint4*>(dst + d_off(r + 32, c)) = t.b;
}
// IEEE half -> bf16 of one 16-byte chunk (the score recompute reads the saved f16 operand, the gradient-side MFMA wants bf16)
__device__ __forceinline__ uint4 h8_to_bf8(uint4 v) {
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
    unsigned o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
        o[i] = pack2bf(h2f((bf16_t)(w[i] & 0xffffu)), h2f((bf16_t)(w[i] >> 16)));
    return make_uint4(o[0], o[1], o[2], o[3]);
}
__device__ __forceinline__ void tile_lstore_cols(const TileRegs& t, unsigned char* dst, int tid) {
    const int r = tid >> 3, c = tid & 7;
    *reinterpret_cast<uint2*>(dst + vt_off(r, 2 * c)) = make_uint2(t.a.x, t.a.y);
    *reinterpret_cast<uint2*>(dst + vt_off(r, 2 * c + 1)) = make_uint2(t.a.z, t.a.w);
    *reinterpret_cast<uint2*>(dst + vt_off(r + 32, 2 * c)) = make_uint2(t.b.x, t.b.y);
    *reinterpret_cast<uint2*>(dst + vt_off(r + 32, 2 * c + 1)) = make_uint2(t.b.z, t.b.w);
}

// Make the compiler wait for (and mark as arrived) register fragments that were loaded from global memory before a
// pipelined loop; otherwise its waitcnt pass conservatively emits s_waitcnt vmcnt(0) at their first in-loop use, which
// also drains the next tile's prefetch every iteration.
__device__ __forceinline__ void pin_frags(const s16x8_t (&f)[4]) {
#pragma unroll
    for (int s = 0; s < 4; ++s) asm volatile("" ::"v"(f[s]));
}

// packed-fp32 helper type (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) and an all-zero accumulator: passing it as the C
// operand of the first MFMA of a chain makes the instruction read the inline constant 0 instead of 16 zeroed registers
typedef float f32x2_t __attribute__((ext_vector_type(2)));
#define zero16 (f32x16_t{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f})
