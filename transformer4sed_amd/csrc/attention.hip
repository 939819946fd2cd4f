// Fused multi-head self-attention (flash style) for the PaSST encoder, forward + backward, gfx950.
//
// Replaces src/models/passt/passt.py:335-341 (scores, softmax, attn @ v; 68 MB/clip/layer of materialised
// probabilities in the reference) and its autograd backward.  head_dim = 64, any sequence length
// (1190 tokens for the global pass, 602 for the 512-frame windows).
//
// Inputs come head-split from the qkv GEMM epilogue (gemm.hip, EPI_QKV):
//   Q, K, V : [B*H, N, 64] 16-bit,   Qt, Kt : [B*H, 64, Npad] (zero padded to a multiple of 64; backward only)
//
// Forward ("swapped" form so that a lane owns one query column of every accumulator):
//   S^T[key, q] = K . Q^T   -> online softmax along registers (+1 cross-half shuffle) ->
//   O^T[d, q]  += V^T[d, key] . P^T[key, q]     with P^T taken straight from the S^T accumulator registers and the V^T fragments
//   read out of the ROW-major V tile with ds_read_b64_tr_b16 (no transposed copy of V in HBM)
//   (the MFMA k-index permutation is chosen to match the accumulator row pattern, so no cross-lane traffic).
// 4 waves x 32 queries per workgroup, 64-key tiles, K and V^T tiles double buffered in XOR-swizzled LDS,
// next tile's global loads in flight during the MFMA phase.
#include <stdlib.h>
#include <string.h>
#include "common.h"
#include "../../include/sed_hip.h"

#include "attn_common.h"

// ---------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------
// One 64-key tile of the online softmax for a lane's query: st holds the raw scores S^T (2 blocks x 16 keys per lane; the
// other 32 keys of the tile live in lane ^ 32), and leaves P = exp2(c s - m) there.  The VALU work per score is the binding
// resource of this kernel (16 MFMAs = 512 matrix-pipe cycles per tile against 4 cycles per VALU instruction), so: max on the
// raw scores (v_max3), scale and max-subtract in one packed FMA, raw v_exp_f32, packed row sums, masking only when MASKED.

// single-instruction fp32 helpers the SLP vectoriser cannot fuse into v_pk_* forms
#ifdef ATT_PLAIN
__device__ __forceinline__ float sfma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ float sadd(float a, float b) { return a + b; }
__device__ __forceinline__ float smul(float a, float b) { return a * b; }
#else
__device__ __forceinline__ float sfma(float a, float b, float c) { float d; asm("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
__device__ __forceinline__ float sadd(float a, float b) { float d; asm("v_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ float smul(float a, float b) { float d; asm("v_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
#endif

template <bool MASKED>
__device__ __forceinline__ void softmax_tile(f32x16_t (&st)[2], f32x16_t (&o)[2], float& m_run, float& l_run, int j0, int N, int lg) {
    if (MASKED) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = j0 + 32 * kb + mfma32_row(r, lg);
                st[kb][r] = key < N ? st[kb][r] : -1e30f;
            }
    }
#ifdef ATT_ABL
    // timing ablations (results are NOT a softmax): 1 = no running max / rescale, 2 = also no exp, 3 = no vector work at all
    {
#if ATT_ABL == 3
        return;
#else
        const f32x2_t c2a = {SCALE_LOG2E, SCALE_LOG2E}, nm2a = {-8.f, -8.f};
        f32x2_t ps2a = {0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                f32x2_t x = {st[kb][r], st[kb][r + 1]};
                x = __builtin_elementwise_fma(x, c2a, nm2a);
#if ATT_ABL == 1
                f32x2_t pv = {__builtin_amdgcn_exp2f(x.x), __builtin_amdgcn_exp2f(x.y)};
#else
                f32x2_t pv = x;
#endif
                st[kb][r] = pv.x; st[kb][r + 1] = pv.y;
                ps2a += pv;
            }
        float psa = ps2a.x + ps2a.y;
        psa += __shfl_xor(psa, 32, 64);
        l_run += psa;
        m_run = 8.f;
        return;
#endif
    }
#endif
    float mloc = -1e30f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; r += 2) mloc = max3_raw(mloc, st[kb][r], st[kb][r + 1]);
    mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
    const float m_new = fmaxf(m_run, mloc * SCALE_LOG2E);   // SCALE_LOG2E > 0: max(c s) = c max(s)
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
#ifdef ATT_SCALAR
    // scalar (one element per instruction) form of the same arithmetic: packed fp32 operations issued beside MFMAs of other waves cost
    // more than their own issue time (MI355X_MICROARCH.md, "price of one filler beside MFMAs")
    float psum = 0.f, psum1 = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const float p0 = __builtin_amdgcn_exp2f(sfma(st[kb][r], SCALE_LOG2E, -m_new));
            const float p1 = __builtin_amdgcn_exp2f(sfma(st[kb][r + 1], SCALE_LOG2E, -m_new));
            st[kb][r] = p0; st[kb][r + 1] = p1;
            psum = sadd(psum, p0); psum1 = sadd(psum1, p1);
        }
    psum = sadd(psum, psum1);
#else
    const f32x2_t c2 = {SCALE_LOG2E, SCALE_LOG2E}, nm2 = {-m_new, -m_new};
    f32x2_t ps2 = {0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            f32x2_t x = {st[kb][r], st[kb][r + 1]};
            x = __builtin_elementwise_fma(x, c2, nm2);
            f32x2_t pv = {__builtin_amdgcn_exp2f(x.x), __builtin_amdgcn_exp2f(x.y)};
            st[kb][r] = pv.x; st[kb][r + 1] = pv.y;
            ps2 += pv;
        }
    float psum = ps2.x + ps2.y;
#endif
    psum += __shfl_xor(psum, 32, 64);
    l_run = l_run * alpha + psum;
    m_run = m_new;
    // (a wave-uniform "max did not move" skip costs more in register copies at the join than the 16 packed multiplies)
#ifdef ATT_SCALAR
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] = smul(o[i][r], alpha);
#else
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
#endif
}

// NQ = query blocks (of 32) per wave: 2 for the bulk of the sequence (256 queries per workgroup: every K / V^T fragment
// read from LDS feeds two MFMAs and each staged K/V tile serves twice as many queries), 1 for a short tail block.
#ifndef WPE
#define WPE 3  // 167 VGPRs: three waves per SIMD (four would spill)
#endif
// Output row of query q, head h (o_mode 0: token-major [B][N][H * 64]; 1: head-major [H][B * N][64], the slab-major A operand of the
// LayerNorm-fold proj GEMM, csrc/gemm.hip a_slab; 2: token-major rows [H * 64 f16 | H * 64 e4m3] of pitch 3 H * 64 / 2 halfs, the A operand
// of the two-term proj GEMM with the lo product on the fp8 path, gemm.hip GemmArgs.k8) -- pointer to the head's 64 columns.
__device__ __forceinline__ bf16_t* attn_out_row(bf16_t* O, int o_mode, int b, int h, int q, int N, int H, int B) {
    if (o_mode == 1) return O + ((size_t)h * B * N + (size_t)b * N + q) * HD;
    const int pitch = o_mode == 2 ? H * HD + H * HD / 2 : H * HD;
    return O + ((size_t)b * N + q) * pitch + h * HD;
}
// four consecutive outputs at column `col` of the head; tail_b >= 0: byte offset from the head's f16 columns to its e4m3 columns
template <bool F16>
__device__ __forceinline__ void attn_store4(bf16_t* orow, int col, float a, float b, float c, float d, int tail_b) {
    uint2 pk;
    pk.x = pack2<F16>(a, b);
    pk.y = pack2<F16>(c, d);
    *reinterpret_cast<uint2*>(orow + col) = pk;
    if (tail_b >= 0)      // e4m3 of 2^-2 x the f16-rounded values (what sed_fp8_tail makes from the f16 half)
        *reinterpret_cast<unsigned*>(reinterpret_cast<unsigned char*>(orow) + tail_b + col) = e4m3x4_of_h4(pk.x, pk.y);
}
// ---------------------------------------------------------------------------------------------------
// forward, K / V tiles staged by LDS-DMA (round 4).  Same arithmetic, tile images and fragment reads as the round-3 register-staged kernel (removed in round 6); what changes is how
// a tile reaches LDS: `buffer_load ... lds` (1 KiB = 8 rows per wave-instruction, 4 per wave and tile) instead of global -> 16 VGPRs ->
// ds_write_b128.  The register path cost the LDS as much as the fragment reads did (a ds_write_b128 moves its 5 source dwords at 2
// cycles each: ~13 cycles per KiB against 4 for a read) and its 16 staging registers.  The DMA writes lane l's 16 bytes at piece base
// + 16 l, so the XOR swizzles of the K image (chunk ^ (row >> 1) & 7) and of the V image (chunk ^ ((row >> 1) & 1) << 2) are applied on the
// SOURCE address; rows past the sequence end read zeros through the buffer descriptor (no clamped duplicates).
// ---------------------------------------------------------------------------------------------------
// XCD-aware order (round 4): workgroups are dispatched round-robin over the 8 XCDs in linear (x fastest) order, which puts the query / key
// blocks of one (clip, head) on up to 8 different L2s -- its K / V (305 KB at N = 1190) fetched through each of them, ~1 GB per forward launch
// at 220 us.  With this order XCD x walks the (clip, head) pairs = x (mod 8), all blocks of a pair back to back (needs B H % 8 == 0).
__device__ __forceinline__ void attn_xcd_order(int& bh, int& blk) {
    const int nq = gridDim.x, nbh = gridDim.y;
    bh = blockIdx.y; blk = blockIdx.x;
#ifndef ATT_NO_XCD      // (A/B build switch, tools/attn_variants.sh)
    if ((nbh & 7) == 0) {
        const int D = blockIdx.y * nq + blockIdx.x, slot = D >> 3;
        bh = (slot / nq) * 8 + (D & 7);
        blk = slot - (slot / nq) * nq;
    }
#endif
}
typedef __attribute__((address_space(3))) void* att_lds_ptr_t;
template <int OFF>
__device__ __forceinline__ unsigned long long att_tr(unsigned addr) {
    unsigned long long v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
__device__ __forceinline__ s16x8_t att_frag(unsigned long long lo, unsigned long long hi) {
    const unsigned w[4] = {(unsigned)lo, (unsigned)(lo >> 32), (unsigned)hi, (unsigned)(hi >> 32)};
    s16x8_t f;
    __builtin_memcpy(&f, w, 16);
    return f;
}
#ifndef WPE_DMA
#define WPE_DMA 3
#endif

template <bool F16>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE_DMA, WPE_DMA))) void mhsa_fwd_dma_kernel(
    const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K, const bf16_t* __restrict__ V, bf16_t* __restrict__ O,
    float* __restrict__ LSE, int N, int H, int o_slab) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2][2][KVB * 128];  // [buf][K | V]
    const int tid = threadIdx.x, lane = tid & 63, lr = lane & 31, lg = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bh, qblk;
    attn_xcd_order(bh, qblk);
    const int b = bh / H, h = bh - b * H;
    const int q0 = qblk * 128 + wave * 32;
    const bf16_t* Qb = Q + (size_t)bh * N * HD;
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void*)(K + (size_t)bh * N * HD), 0, N * HD * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)(V + (size_t)bh * N * HD), 0, N * HD * 2, 0x00020000);
    // this wave's two pieces of a tile: rows 8 p .. 8 p + 7, p = 2 wave + e; lane -> (row, stored chunk slot)
    int vok[2], vov[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int row = 8 * (2 * wave + e) + (lane >> 3), cs = lane & 7;
        vok[e] = row * 128 + ((cs ^ ((row >> 1) & 7)) << 4);
        vov[e] = row * 128 + ((cs ^ (((row >> 1) & 1) << 2)) << 4);
    }
#define ATT_DMA(T_, BUF_)                                                                                                            \
    {                                                                                                                                \
        const int so_ = (T_) * (KVB * 128);                                                                                          \
        _Pragma("unroll") for (int e = 0; e < 2; ++e) {                                                                              \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, (att_lds_ptr_t)(lds[BUF_][0] + (2 * wave + e) * 1024), 16, vok[e], so_, 0, 0); \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (att_lds_ptr_t)(lds[BUF_][1] + (2 * wave + e) * 1024), 16, vov[e], so_, 0, 0); \
        }                                                                                                                            \
    }
    s16x8_t qf[4];
    {
        int qrow = q0 + lr;
        qrow = qrow < N ? qrow : N - 1;
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const s16x8_t*>(Qb + (size_t)qrow * HD + 16 * s + 8 * lg);
    }
    f32x16_t o[2];
    float m_run = -1e30f, l_run = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
    const int ntiles = (N + KVB - 1) / KVB;
    // lane part of the V^T fragment address (lds_frag_vt): key row 4 lg + (a >> 2) of a 16-key block, d block 0 (block 1: ^ 64)
    const unsigned vaddr = (unsigned)(size_t)&lds[0][0][0] + (4 * lg + ((lane & 15) >> 2)) * 128 + (((lane & 15) >> 3) << 6) + ((lane >> 4) & 1) * 32 +
                           8 * (lane & 3);
    ATT_DMA(0, 0)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    pin_frags(qf);
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        const int buf = t & 1, j0 = t * KVB;
        if (t + 1 < ntiles) {
            if (buf) ATT_DMA(t + 1, 0) else ATT_DMA(t + 1, 1)
        }
        const unsigned char* lk = lds[buf][0];
        const unsigned char* lv = lds[buf][1];
        f32x16_t st[2];
        // (round 5: a short path for a last tile that ends in its first 32 keys -- N = 2 + 12 tp leaves 2 / 26 keys there at 386 / 602 tokens:
        //  one score block, softmax over it alone, 1-2 of the 4 P.V steps -- was bit-compatible and SLOWER: 167 instead of 144 VGPRs and a
        //  second tile body cost the other 18 tiles more than the last one saved, 209 -> 220 us at N = 1190, 788 -> 797 at 602.  Not kept.)
        if (q0 < N) {        // (wave-uniform) a wave whose 32 queries all lie past the sequence end only stages tiles and joins the barriers
#ifndef ATT_NO_KPRE
        {   // all eight K fragments requested before the first MFMA: their LDS latency is paid once per tile, not once per MFMA
            s16x8_t kf[2][4];
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) kf[kb][s] = lds_frag_rows(lk, 32 * kb + lr, 2 * s + lg);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) st[kb] = mfma32t<F16>(kf[kb][s], qf[s], s == 0 ? zero16 : st[kb]);
        }
#else
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int s = 0; s < 4; ++s)
                st[kb] = mfma32t<F16>(lds_frag_rows(lk, 32 * kb + lr, 2 * s + lg), qf[s], s == 0 ? zero16 : st[kb]);
#endif
        if (j0 + KVB > N) softmax_tile<true>(st, o, m_run, l_run, j0, N, lg);
        else softmax_tile<false>(st, o, m_run, l_run, j0, N, lg);
        // V^T fragments by transposing reads issued as inline asm: through the builtin the compiler's wait-count pass assumes a read may
        // alias the LDS-DMA in flight and puts `s_waitcnt vmcnt(0)` -- the NEXT tile's arrival -- in front of the first one.  Two fragment
        // sets, four steps (key blocks of 16): reads of step j + 2 are issued behind the MFMAs of step j, counted lgkmcnt waits.
        {
            const unsigned va = vaddr ^ (buf << 14);
            unsigned long long vl[2][2], vh[2][2];
#define ATT_RDV(SET, J)                                                                                                   \
            vl[SET][0] = att_tr<8192 + (J) * 2048>(va); vh[SET][0] = att_tr<8192 + (J) * 2048 + 1024>(va);                \
            vl[SET][1] = att_tr<8192 + (J) * 2048>(va ^ 64); vh[SET][1] = att_tr<8192 + (J) * 2048 + 1024>(va ^ 64);
#define ATT_PV(SET, J, WAIT)                                                                                              \
            {                                                                                                             \
                const s16x8_t pf = pack_frag_t<F16>(st[(J) >> 1], (J) & 1);                                               \
                asm volatile("s_waitcnt lgkmcnt(" #WAIT ")" : "+v"(vl[SET][0]), "+v"(vh[SET][0]), "+v"(vl[SET][1]), "+v"(vh[SET][1]) :: "memory"); \
                _Pragma("unroll") for (int db = 0; db < 2; ++db) o[db] = mfma32t<F16>(att_frag(vl[SET][db], vh[SET][db]), pf, o[db]); \
            }
            ATT_RDV(0, 0) ATT_RDV(1, 1)
            ATT_PV(0, 0, 4)
            ATT_RDV(0, 2)
            ATT_PV(1, 1, 4)
            ATT_RDV(1, 3)
            ATT_PV(0, 2, 4)
            ATT_PV(1, 3, 0)
#undef ATT_RDV
#undef ATT_PV
        }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of tile t + 1 have landed ...
        __syncthreads();                                      // ... and everybody's; nobody reads tile t any more
    }
#undef ATT_DMA
    // Output through a wave-private LDS tile (the K / V stages are free: everybody is behind the K loop's last barrier).  A lane's
    // accumulators are 8-byte pieces of ONE query row -- stored directly, an instruction covers 32 rows x 16 bytes; read back as 16-byte
    // chunks (lane -> row lane >> 3, chunk lane & 7) it covers 8 rows x 128 contiguous bytes, the e4m3 image (o_slab 2) 8 x 64.
    // (o_slab 1: head-major output [H][B * N][64]: a workgroup's 128 queries are one contiguous 16 KB run, and the rows are the slab-major A
    //  operand of the LayerNorm-fold proj GEMM, csrc/gemm.hip a_slab; token-major they are 128-byte pieces 1536 bytes apart)
    {
        unsigned char* ws = &lds[0][0][0] + wave * 8192;      // 32 rows x 144 bytes
        const float inv = 1.0f / l_run;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                uint2 pk;
                pk.x = pack2<F16>(o[db][4 * qd] * inv, o[db][4 * qd + 1] * inv);
                pk.y = pack2<F16>(o[db][4 * qd + 2] * inv, o[db][4 * qd + 3] * inv);
                *reinterpret_cast<uint2*>(ws + lr * 144 + (32 * db + 8 * qd + 4 * lg) * 2) = pk;
            }
        if (q0 + lr < N && lg == 0 && LSE != nullptr) LSE[(size_t)bh * N + q0 + lr] = m_run + log2f(l_run);  // log2 domain
        __builtin_amdgcn_wave_barrier();
        const int rr = lane >> 3, ch = lane & 7;
        const int tail_b = (H - h) * HD * 2 + h * HD;      // o_slab 2: bytes from the head's f16 columns to its e4m3 columns
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = 8 * i + rr, q = q0 + row;
            const uint4 v = *reinterpret_cast<const uint4*>(ws + row * 144 + ch * 16);
            if (q < N) {
                bf16_t* orow = attn_out_row(O, o_slab, b, h, q, N, H, (int)gridDim.y / H);
                *reinterpret_cast<uint4*>(orow + 8 * ch) = v;
                if (o_slab == 2)
                    *reinterpret_cast<uint2*>(reinterpret_cast<unsigned char*>(orow) + tail_b + 8 * ch) =
                        make_uint2(e4m3x4_of_h4(v.x, v.y), e4m3x4_of_h4(v.z, v.w));
            }
        }
    }
}

template <bool F16>
static void launch_mhsa_fwd(const void* Q, const void* K, const void* Vt, void* O, float* LSE, int B, int H, int N, int Npad, int o_slab,
                            hipStream_t stream) {
    // K / V tiles by LDS-DMA (needs N * 128 B < 2 GiB per head: always).  (The round-3 register-staged kernel and its SED_MHSA_FWD=reg
    // switch went in round 6: it had served as the A/B reference of the DMA form)
    (void)Npad;
    hipLaunchKernelGGL((mhsa_fwd_dma_kernel<F16>), dim3(cdiv(N, 128), B * H), dim3(256), 0, stream, (const bf16_t*)Q, (const bf16_t*)K,
                       (const bf16_t*)Vt, (bf16_t*)O, LSE, N, H, o_slab);
}

extern "C" int sed_mhsa_fwd(const void* Q, const void* K, const void* V, void* O, float* LSE, int B, int H, int N,
                            int Npad, int f16, hipStream_t stream) {
    (void)hipGetLastError();
    if (N <= 0 || Npad % 64 || Npad < N) return SED_ERR_ARG;
    // f16 bit 0: IEEE half operands (else bf16); bit 1: O head-major [H][B * N][64] instead of token-major [B][N][H * 64]; bit 2 (f16 only):
    // token-major rows [H * 64 f16 | H * 64 e4m3], pitch 3 H * 64 / 2 halfs (the A operand of sed_gemm_nt_w2f8)
    if ((f16 & 4) && (f16 & 3) != 1) return SED_ERR_ARG;
    const int o_mode = (f16 & 4) ? 2 : (f16 >> 1) & 1;
    if (f16 & 1) launch_mhsa_fwd<true>(Q, K, V, O, LSE, B, H, N, Npad, o_mode, stream);
    else launch_mhsa_fwd<false>(Q, K, V, O, LSE, B, H, N, Npad, o_mode, stream);
    return sed_check_launch();
}

// ---------------------------------------------------------------------------------------------------
// backward pre-pass: D[bh, q] = sum_d dO * O ; head-split copies dOh [BH, N, 64] and dOt [BH, 64, Npad]
// one wave per (b, q, h-pair): 64 lanes x 2 elements = 128 channels = 2 heads
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mhsa_bwd_prep_kernel(const bf16_t* __restrict__ dO, const void* __restrict__ O,
                                                            float* __restrict__ Dv, bf16_t* __restrict__ dOh,
                                                            bf16_t* __restrict__ dOt, int B, int N, int Npad, int H,
                                                            int o_f16) {
    // block handles 64 consecutive tokens of one (b, h): transposes through LDS
    __shared__ bf16_t tile[64][66];
    const int bh = blockIdx.y, b = bh / H, h = bh - b * H;
    const int t0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
        const int t = t0 + ty * 16 + i;
        float prod = 0.f;
        bf16_t g = 0;
        if (t < N) {
            const size_t idx = ((size_t)b * N + t) * (H * HD) + h * HD + tx;
            g = dO[idx];
            prod = bf2f(g) * (o_f16 == 2 ? ((const float*)O)[idx] : (o_f16 ? h2f(((const bf16_t*)O)[idx]) : bf2f(((const bf16_t*)O)[idx])));
            dOh[((size_t)bh * N + t) * HD + tx] = g;
        }
        tile[ty * 16 + i][tx] = g;
        prod = wave_sum(prod);
        if (tx == 0 && t < N) Dv[(size_t)bh * N + t] = prod;
    }
    __syncthreads();
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
        const int d = ty * 16 + i, t = t0 + tx;
        if (dOt != nullptr && t < Npad) dOt[((size_t)bh * HD + d) * Npad + t] = (t < N) ? tile[tx][d] : (bf16_t)0;
    }
}

// ---------------------------------------------------------------------------------------------------
// backward kernel 1: dK, dV.  Workgroup = 128 keys (4 waves x 32), loops over 64-query tiles.
//   S[q, key]  = Q . K^T  (A = Q rows from LDS, B = K fragments in registers)  -> lane owns a key column
//   P = exp2(S c - L2[q]);  dP[q, key] = dO . V^T ;  dS = P (dP - D[q])
//   dV[key, d] += P^T[key, q] dO[q, d]   (A from the P registers, B = dO^T fragments)
//   dK[key, d] += dS^T[key, q] Q[q, d] * scale   (B = Q^T fragments)
// The transposed operands come out of ROW-major LDS tiles through ds_read_b64_tr_b16 (round 2): no Q^T / dO^T copies in HBM or LDS
// (the first version staged them as separate column-swizzled tiles -- a third of its LDS cycles were bank conflicts).  Q is saved as
// IEEE half for the score recompute while the gradient-side MFMA is bf16: the bf16 image of the Q tile is made on the way into LDS.
// ---------------------------------------------------------------------------------------------------
template <bool SF16>
__global__ __launch_bounds__(256) void mhsa_bwd_dkdv_kernel(
    const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K, const bf16_t* __restrict__ V,
    const bf16_t* __restrict__ dO, const float* __restrict__ LSE, const float* __restrict__ Dv,
    bf16_t* __restrict__ dqkv, int N, int Npad, int H, int v_is_f16) {
    // LDS per stage: Q rows (S type, row fragments), Q rows bf16 (transposing reads; the same tile when the forward ran in bf16),
    // dO rows bf16 (dual use: row fragments for dP, transposing reads for dV), L2[64], D[64]
    __shared__ __attribute__((aligned(16))) unsigned char lds[2][3][KVB * 128];
    __shared__ __attribute__((aligned(16))) float lstat[2][2][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31, lg = lane >> 5;
    int bh, kblk;
    attn_xcd_order(bh, kblk);       // (the key blocks of a pair share its Q / dO tiles)
    const int b = bh / H, h = bh - b * H;
    const int key0 = kblk * 128 + wave * 32;
    const bool wave_live = __builtin_amdgcn_readfirstlane(key0) < N;
    const size_t hb = (size_t)bh * N * HD;

    int krow = key0 + lr;
    krow = krow < N ? krow : N - 1;
    s16x8_t kf[4], vf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        kf[s] = *reinterpret_cast<const s16x8_t*>(K + hb + (size_t)krow * HD + 16 * s + 8 * lg);
        const uint4 vraw = *reinterpret_cast<const uint4*>(V + hb + (size_t)krow * HD + 16 * s + 8 * lg);
        vf[s] = __builtin_bit_cast(s16x8_t, (SF16 && v_is_f16) ? h8_to_bf8(vraw) : vraw);   // saved V is f16: bf16 operand made here
    }
    f32x16_t dk[2], dv[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[i][r] = 0.f; dv[i][r] = 0.f; }

    const int ntiles = (N + 63) / 64;
    TileRegs rq, rdo;
    float rs = 0.f;
    auto gload = [&](int t) {
        const int i0 = t * 64;
        tile_gload(rq, Q + hb, i0, N, HD, 0, tid);
        tile_gload(rdo, dO + (size_t)b * N * (H * HD), i0, N, H * HD, h * HD, tid);   // token-major dO [B, N, H * 64]: no head-split copy
        if (tid < 128) {
            const int qi = i0 + (tid & 63);
            const float* src = (tid < 64) ? LSE : Dv;
            rs = qi < N ? src[(size_t)bh * N + qi] : (tid < 64 ? 1e30f : 0.f);  // L2 = +big -> P = 0 for padded queries
        }
    };
    auto lstore = [&](int buf) {
        if (SF16) {
            tile_lstore_rows(rq, lds[buf][0], tid);
            TileRegs rb;
            rb.a = h8_to_bf8(rq.a);
            rb.b = h8_to_bf8(rq.b);
            tile_lstore_drows(rb, lds[buf][1], tid);
        } else {
            tile_lstore_drows(rq, lds[buf][1], tid);
        }
        tile_lstore_drows(rdo, lds[buf][2], tid);
        if (tid < 128) lstat[buf][tid >> 6][tid & 63] = rs;
    };
    gload(0);
    lstore(0);
    pin_frags(kf);
    pin_frags(vf);
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        const int buf = t & 1;
        if (t + 1 < ntiles) gload(t + 1);
        const unsigned char* lq = lds[buf][1];
        const unsigned char* ldo = lds[buf][2];
        if (wave_live)       // (wave-uniform) a wave whose 32 keys all lie past the sequence end only stages tiles and joins the barriers
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {  // 32-query sub-blocks of the tile
            f32x16_t s_, dp;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const s16x8_t qfr = SF16 ? lds_frag_rows(lds[buf][0], 32 * qb + lr, 2 * s + lg) : lds_frag_rows_d(lq, 32 * qb + lr, 2 * s + lg);
                s_ = mfma32t<SF16>(qfr, kf[s], s == 0 ? zero16 : s_);
                dp = mfma32(lds_frag_rows_d(ldo, 32 * qb + lr, 2 * s + lg), vf[s], s == 0 ? zero16 : dp);
            }
            // rows of the accumulators are queries 32 qb + mfma32_row(r, lg); column = this lane's key.  A lane whose key is
            // >= N needs no masking here: its P / dS columns only feed the dK / dV rows of that key, which are never stored
            // (its K / V rows are clamped copies of the last real key, so everything stays finite).
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const int qq = 32 * qb + 8 * qd + 4 * lg;
                const f32x4_t l2 = *reinterpret_cast<const f32x4_t*>(&lstat[buf][0][qq]);
                const f32x4_t dd = *reinterpret_cast<const f32x4_t*>(&lstat[buf][1][qq]);
#pragma unroll
                for (int j = 0; j < 4; j += 2) {
                    const int r = 4 * qd + j;
                    const f32x2_t c2 = {SCALE_LOG2E, SCALE_LOG2E}, nl = {-l2[j], -l2[j + 1]}, nd = {-dd[j], -dd[j + 1]};
                    f32x2_t x = {s_[r], s_[r + 1]}, d2 = {dp[r], dp[r + 1]};
                    x = __builtin_elementwise_fma(x, c2, nl);
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
                    dv[db] = mfma32(pf, lds_frag_vt_d(ldo, db, 8 * qb + 4 * s + lg, lane), dv[db]);
                    dk[db] = mfma32(dsf, lds_frag_vt_d(lq, db, 8 * qb + 4 * s + lg, lane), dk[db]);
                }
            }
        }
        if (t + 1 < ntiles) lstore(buf ^ 1);
        __syncthreads();
    }
    // dk/dv accumulators: row = key (mfma32_row), column = d = 32 db + lr
    const int ldq = 3 * H * HD;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = key0 + mfma32_row(r, lg);
            if (key < N) {
                bf16_t* row = dqkv + ((size_t)b * N + key) * ldq + h * HD + 32 * db + lr;
                row[H * HD] = f2bf(dk[db][r] * SCALE);
                row[2 * H * HD] = f2bf(dv[db][r]);
            }
        }
}

// ---------------------------------------------------------------------------------------------------
// backward kernel 2: dQ.  Workgroup = 128 queries, loops over 64-key tiles (swapped form as in forward).
//   S^T[key, q] = K . Q^T ; dP^T[key, q] = V . dO^T ; dS^T = P^T (dP^T - D[q])
//   dQ^T[d, q] += K^T[d, key] dS^T[key, q]      (K^T fragments by transposing reads of the bf16 image of the K rows tile)
// ---------------------------------------------------------------------------------------------------
template <bool SF16>
__global__ __launch_bounds__(256) void mhsa_bwd_dq_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,
                                                          const bf16_t* __restrict__ V,
                                                          const bf16_t* __restrict__ dO, const bf16_t* __restrict__ O,
                                                          const float* __restrict__ LSE,
                                                          float* __restrict__ Dv, bf16_t* __restrict__ dqkv,
                                                          int N, int Npad, int H, int v_is_f16) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2][3][KVB * 128];  // K rows (S type), V rows, K rows bf16 (dual use)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31, lg = lane >> 5;
    int bh, qblk;
    attn_xcd_order(bh, qblk);
    const int b = bh / H, h = bh - b * H;
    const int q0 = qblk * 128 + wave * 32;
    const bool wave_live = __builtin_amdgcn_readfirstlane(q0) < N;
    const size_t hb = (size_t)bh * N * HD;
    int qrow = q0 + lr;
    const bool qvalid = qrow < N;
    qrow = qvalid ? qrow : N - 1;
    // dO and O rows of this lane's query straight from the token-major [B, N, H * 64] tensors (the pre-pass that made head-split
    // copies and D = rowsum(dO * O) is gone: D is one shuffle away here, and the dK/dV kernel, launched afterwards, reads it)
    const size_t trow = ((size_t)b * N + qrow) * (H * HD) + h * HD;
    s16x8_t qf[4], dof[4];
    float dd = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        qf[s] = *reinterpret_cast<const s16x8_t*>(Q + hb + (size_t)qrow * HD + 16 * s + 8 * lg);
        dof[s] = *reinterpret_cast<const s16x8_t*>(dO + trow + 16 * s + 8 * lg);
        const s16x8_t of = *reinterpret_cast<const s16x8_t*>(O + trow + 16 * s + 8 * lg);
#pragma unroll
        for (int e = 0; e < 8; ++e) dd += bf2f((bf16_t)dof[s][e]) * to_f32<SF16>((bf16_t)of[e]);
    }
    dd += __shfl_xor(dd, 32, 64);
    if (lg == 0 && qvalid) Dv[(size_t)bh * N + qrow] = dd;
    const float l2 = LSE[(size_t)bh * N + qrow];
    f32x16_t dq[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[i][r] = 0.f;
    const int ntiles = (N + KVB - 1) / KVB;
    TileRegs rk, rv;
    auto gload = [&](int t) {
        tile_gload(rk, K + hb, t * KVB, N, HD, 0, tid);
        tile_gload(rv, V + hb, t * KVB, N, HD, 0, tid);
    };
    auto lstore = [&](int buf) {
        if (SF16) {
            tile_lstore_rows(rk, lds[buf][0], tid);
            TileRegs rb;
            rb.a = h8_to_bf8(rk.a);
            rb.b = h8_to_bf8(rk.b);
            tile_lstore_drows(rb, lds[buf][2], tid);
        } else {
            tile_lstore_drows(rk, lds[buf][2], tid);
        }
        if (SF16 && v_is_f16) {   // the saved V is IEEE half: its bf16 operand image is made on the way into LDS
            TileRegs vb;
            vb.a = h8_to_bf8(rv.a);
            vb.b = h8_to_bf8(rv.b);
            tile_lstore_rows(vb, lds[buf][1], tid);
        } else {
            tile_lstore_rows(rv, lds[buf][1], tid);
        }
    };
    gload(0);
    lstore(0);
    pin_frags(qf);
    pin_frags(dof);
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        const int buf = t & 1, j0 = t * KVB;
        if (t + 1 < ntiles) gload(t + 1);
        if (wave_live)       // (wave-uniform) see mhsa_bwd_dkdv_kernel
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            f32x16_t st, dp;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const s16x8_t kfr = SF16 ? lds_frag_rows(lds[buf][0], 32 * kb + lr, 2 * s + lg) : lds_frag_rows_d(lds[buf][2], 32 * kb + lr, 2 * s + lg);
                st = mfma32t<SF16>(kfr, qf[s], s == 0 ? zero16 : st);
                dp = mfma32(lds_frag_rows(lds[buf][1], 32 * kb + lr, 2 * s + lg), dof[s], s == 0 ? zero16 : dp);
            }
            const f32x2_t c2 = {SCALE_LOG2E, SCALE_LOG2E}, nl = {-l2, -l2}, nd = {-dd, -dd};
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                f32x2_t x = {st[r], st[r + 1]}, d2 = {dp[r], dp[r + 1]};
                x = __builtin_elementwise_fma(x, c2, nl);
                f32x2_t pv = {__builtin_amdgcn_exp2f(x.x), __builtin_amdgcn_exp2f(x.y)};
                if (j0 + KVB > N) {  // keys >= N exist only in the last tile
                    const int key = j0 + 32 * kb + mfma32_row(r, lg);
                    pv.x = key < N ? pv.x : 0.f;
                    pv.y = key + 1 < N ? pv.y : 0.f;
                }
                d2 = pv * (d2 + nd);
                dp[r] = d2.x; dp[r + 1] = d2.y;
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const s16x8_t dsf = pack_frag(dp, s);
#pragma unroll
                for (int db = 0; db < 2; ++db)
                    dq[db] = mfma32(lds_frag_vt_d(lds[buf][2], db, 8 * kb + 4 * s + lg, lane), dsf, dq[db]);
            }
        }
        if (t + 1 < ntiles) lstore(buf ^ 1);
        __syncthreads();
    }
    if (qvalid) {
        bf16_t* row = dqkv + ((size_t)b * N + q0 + lr) * (3 * H * HD) + h * HD;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                uint2 pk;
                pk.x = pack2bf(dq[db][4 * qd] * SCALE, dq[db][4 * qd + 1] * SCALE);
                pk.y = pack2bf(dq[db][4 * qd + 2] * SCALE, dq[db][4 * qd + 3] * SCALE);
                *reinterpret_cast<uint2*>(row + 32 * db + 8 * qd + 4 * lg) = pk;
            }
    }
}

extern "C" int sed_mhsa_bwd_prep(const void* dO, const void* O, float* Dtmp, void* dOh, void* dOt, int B, int H, int N,
                                 int Npad, int o_f16, hipStream_t stream) {
    (void)hipGetLastError();
    if (N <= 0 || Npad % 64 || Npad < N) return SED_ERR_ARG;
    hipLaunchKernelGGL(mhsa_bwd_prep_kernel, dim3(Npad / 64, B * H), dim3(256), 0, stream, (const bf16_t*)dO,
                       O, Dtmp, (bf16_t*)dOh, (bf16_t*)dOt, B, N, Npad, H, o_f16);
    return sed_check_launch();
}

extern "C" int sed_mhsa_bwd(const void* Q, const void* K, const void* V, const void* O, const void* dO, const float* LSE,
                            float* Dtmp, void* dOh, void* dqkv, int B, int H, int N, int Npad, int f16, int v_f16,
                            hipStream_t stream) {
    (void)hipGetLastError();
    // f16 != 0: Q, K (score recompute) and O are IEEE half as written by the forward; dO is bf16; V is bf16, or IEEE half as saved by
    // the forward when v_f16 != 0 (converted inside the kernels).  dOh is unused since the kernels
    // read dO / O in their token-major layout (kept in the signature for ABI stability; may be NULL).
    (void)dOh;
    if (N <= 0 || Npad % 64 || Npad < N) return SED_ERR_ARG;
    dim3 grid(cdiv(N, 128), B * H);
    // dQ first: it also produces D = rowsum(dO * O), which the dK/dV kernel reads
#define SED_LAUNCH_BWD(F)                                                                                              \
    hipLaunchKernelGGL(mhsa_bwd_dq_kernel<F>, grid, dim3(256), 0, stream, (const bf16_t*)Q, (const bf16_t*)K,          \
                       (const bf16_t*)V, (const bf16_t*)dO, (const bf16_t*)O, LSE, Dtmp, (bf16_t*)dqkv, N, Npad, H,   \
                       v_f16);                                                                                         \
    hipLaunchKernelGGL(mhsa_bwd_dkdv_kernel<F>, grid, dim3(256), 0, stream, (const bf16_t*)Q, (const bf16_t*)K,        \
                       (const bf16_t*)V, (const bf16_t*)dO, LSE, Dtmp, (bf16_t*)dqkv, N, Npad, H, v_f16);
    if (f16) { SED_LAUNCH_BWD(true) } else { SED_LAUNCH_BWD(false) }
#undef SED_LAUNCH_BWD
    return sed_check_launch();
}
