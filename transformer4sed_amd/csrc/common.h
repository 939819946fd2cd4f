// Shared device helpers for the MAT-SED HIP kernels (gfx950 / CDNA4 only, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SED_OK 0
#define SED_ERR_ARG (-1)
#define SED_ERR_LAUNCH (-2)

typedef unsigned short bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((unsigned)h) << 16); }
// fp32 -> 16-bit conversions on the gfx950 packed converters (v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32, round-to-nearest-even): one VALU
// instruction per PAIR instead of the 4-instruction integer rounding sequence per element
typedef __attribute__((ext_vector_type(2))) float f32x2v_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2v_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2v_t;
__device__ __forceinline__ unsigned pack2bf(float a, float b) {
    const f32x2v_t v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2v_t));
}
__device__ __forceinline__ unsigned pack2h(float a, float b) {
    const f32x2v_t v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2v_t));
}
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
__device__ __forceinline__ float h2f(bf16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
__device__ __forceinline__ bf16_t f2h(float f) { return __builtin_bit_cast(bf16_t, (_Float16)f); }
// 16-bit operand type selected per launch: F16 = IEEE half (forward activations/weights: 11-bit significand keeps
// the frame posteriors within 1e-3 of the fp32 reference), otherwise bf16 (gradient operands: fp32 exponent range).
template <bool F16> __device__ __forceinline__ float to_f32(bf16_t h) { return F16 ? h2f(h) : bf2f(h); }
template <bool F16> __device__ __forceinline__ bf16_t to_16(float f) { return F16 ? f2h(f) : f2bf(f); }
template <bool F16> __device__ __forceinline__ unsigned pack2(float a, float b) { return F16 ? pack2h(a, b) : pack2bf(a, b); }
template <bool F16> __device__ __forceinline__ f32x16_t mfma32t(s16x8_t a, s16x8_t b, f32x16_t c) {
    if (F16)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x16_t mfma32(s16x8_t a, s16x8_t b, f32x16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
// Row of accumulator register r (0..15) of a 32x32 MFMA tile for lane group g = lane >> 5; column = lane & 31.
__device__ __forceinline__ int mfma32_row(int r, int g) { return (r & 3) + 8 * (r >> 2) + 4 * g; }

// max of three without the canonicalisation moves the compiler adds around fmaxf for possibly-signalling inputs
__device__ __forceinline__ float max3_raw(float a, float b, float c) {
    float d;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
    const float pdf = 0.39894228040143268f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}
// Round 5 form.  gelu(x) = max(x, 0) - r(|x|),  r(a) = a Q(a),  Q(a) = 0.5 erfc(a / sqrt 2) = 2^P(a): log2 of the normal upper tail is
// close to a parabola, a quintic P with P(0) = -1 pinned (so gelu(x) -> x / 2 as x -> 0) reproduces r to 5.3e-7 on [0, 6] (weighted
// minimax fit of a Q ln2 dP, /tmp-side script quoted in DESIGN section 3), |gelu error| <= 9.3e-7 over [-8, 8] evaluated in fp32 -- the
// level of the Abramowitz-Stegun 7.1.28 form it replaces ((1 + a1 t + ... + a6 t^6)^-16: 6 FMAs, 4 squarings, a reciprocal and 5
// more operations per element = 19 issue slots with the reciprocal counted as 3).  Per element now: 5 FMAs, ONE v_exp_f32, max(x, 0)
// and the final FMA = 10 slots (|x| is a source modifier).  No clamp: P falls monotonically beyond the fitted range (P(6) = -30.2,
// the leading coefficient is negative), so a 2^P(a) -> 0 like r does, and a NaN input still comes out as NaN.  The symmetry
// gelu(x) - gelu(-x) = x holds exactly.
typedef float f32x2v __attribute__((ext_vector_type(2)));
#define SED_GELU_C1 (-1.1510004997253418f)
#define SED_GELU_C2 (-0.45959582924842834f)
#define SED_GELU_C3 (-0.052146632224321365f)
#define SED_GELU_C4 (0.007198718376457691f)
#define SED_GELU_C5 (-0.0004881021159235388f)
__device__ __forceinline__ f32x2v gelu_tail2(f32x2v a) {   // Q(a) = upper-tail probability of a >= 0 
    f32x2v p = {SED_GELU_C5, SED_GELU_C5};
    p = p * a + SED_GELU_C4;
    p = p * a + SED_GELU_C3;
    p = p * a + SED_GELU_C2;
    p = p * a + SED_GELU_C1;
    p = p * a - 1.0f;
    return f32x2v{__builtin_amdgcn_exp2f(p.x), __builtin_amdgcn_exp2f(p.y)};
}
__device__ __forceinline__ f32x2v gelu_fast2(f32x2v x) {
    // Non-finite inputs: NaN -> NaN (a = NaN poisons the tail term), and +-inf -> NaN as well (inf * 2^-inf = inf * 0), where the erf form
    // gave +inf / -0.  An infinite pre-activation is an already diverged run, and NaN is the louder of the two signals; clamping |x| for the
    // tail term (one v_min per element in the hottest epilogue of the model) would also swallow NaN inputs (fmin(NaN, 14) = 14).  Kernel test:
    // tests/test_gpu_dasm_train.py::test_gemm_f32_all_forms_vs_torch.
    const f32x2v a = {fabsf(x.x), fabsf(x.y)};
    const f32x2v q = gelu_tail2(a);
    const f32x2v r = {fmaxf(x.x, 0.0f), fmaxf(x.y, 0.0f)};
    return r - a * q;
}
__device__ __forceinline__ float gelu_fast(float x) { return gelu_fast2(f32x2v{x, x}).x; }
__device__ __forceinline__ f32x2v gelu_fast_grad2(f32x2v x) {    // Phi(x) + x phi(x), Phi(x) = 0.5 + sign(x) (0.5 - Q(|x|))  (|Phi error| 2.6e-6)
    const f32x2v a = {fabsf(x.x), fabsf(x.y)};
    const f32x2v d = 0.5f - gelu_tail2(a);
    const f32x2v sd = {__builtin_copysignf(d.x, x.x), __builtin_copysignf(d.y, x.y)};
    const f32x2v q = x * x * -0.72134752044448170f;                 // -0.5 x^2 log2(e)
    const f32x2v e = {__builtin_amdgcn_exp2f(q.x), __builtin_amdgcn_exp2f(q.y)};
    return (x * 0.39894228040143268f) * e + sd + 0.5f;
}
__device__ __forceinline__ float gelu_fast_grad(float x) { return gelu_fast_grad2(f32x2v{x, x}).x; }
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

// XCD-aware remap of a linear workgroup id: consecutive logical tiles land on the same XCD (private L2).
// Bijective for any nwg (guide T1).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

static inline int sed_check_launch() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? SED_OK : -(1000 + (int)e);  // -(1000 + hipError_t): decoded by the Python binding
}
static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// four fp32 -> four OCP e4m3 bytes (little end first) for the fp8 lo product of the two-term GEMMs (gemm.hip GemmArgs.k8):
// v_cvt_pk_fp8_f32 rounds to nearest even; values are clamped to +-448 first (an out-of-range input would convert to NaN).
__device__ __forceinline__ unsigned pack4_e4m3(float a, float b, float c, float d) {
    const float lim = 448.f;
    a = __builtin_amdgcn_fmed3f(a, -lim, lim); b = __builtin_amdgcn_fmed3f(b, -lim, lim);
    c = __builtin_amdgcn_fmed3f(c, -lim, lim); d = __builtin_amdgcn_fmed3f(d, -lim, lim);
    int w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
    return (unsigned)w;
}
// The activation's e4m3 image is 2^-2 x the (f16) activation, clamped to the e4m3 range (|x| <= 1792): four packed halves -> four bytes.
// v_cvt_scalef32_pk_fp8_f16 converts src / scale with round-to-nearest-even and does NOT saturate (tools/ablate/cvt_fp8_probe.hip:
// 2000 / 4 -> 0x7f = NaN), hence the packed clamp in front.
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef short s16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned e4m3x4_of_h4(unsigned h01, unsigned h23) {
    const f16x2_t lim = {(_Float16)1792.f, (_Float16)1792.f};
    f16x2_t a = __builtin_bit_cast(f16x2_t, h01), b = __builtin_bit_cast(f16x2_t, h23);
    a = __builtin_elementwise_min(__builtin_elementwise_max(a, -lim), lim);
    b = __builtin_elementwise_min(__builtin_elementwise_max(b, -lim), lim);
    s16x2_t r = {0, 0};
    r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(r, a, 4.0f, false);
    r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(r, b, 4.0f, true);
    return __builtin_bit_cast(unsigned, r);
}
