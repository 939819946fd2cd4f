/* libsed_hip.so -- C ABI of the MI355X-native MAT-SED hot path (gfx950).
 *
 * The reference (cai525/Transformer4SED) is pure PyTorch and has no FFI of its own; the seam is the set of
 * Python call contracts of SURVEY.md section 8(b).  This header is the native boundary behind those contracts:
 * every entry point takes raw DEVICE pointers, sizes and a hipStream_t, never allocates or frees caller-visible
 * memory, keeps no static state, is stream-ordered and re-entrant, and returns 0 or a negative error code
 * (-1 bad argument, -2 HIP launch failure, -(1000 + hipError_t) for a failed launch with the HIP error code) which the Python binding
 * (transformer4sed_amd/_lib.py) turns into RuntimeError.  Each entry cites the reference code it replaces
 * (paths relative to the reference repository root).
 *
 * Conventions: f32 = float, 16-bit operands = raw uint16 bit patterns (void*): bf16, or IEEE half where an `f16` flag is
 * set (forward activations/weights; gradient-side operands are always bf16), row-major, "D" = 768 channels,
 * H = 12 heads x 64.  Sequence-padded buffers ("pad") must be zero-initialised once by the caller.
 */
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

/* ABI version.  Bumped whenever an EXISTING entry point changes its argument list (new entry points alone do not bump it; a changed
 * one normally gets a new suffixed name instead, like sed_median_filter_k).  History: 3 = round 3 (sed_relpos_attn_fwd gained O_split,
 * sed_relpos_attn_bwd gained Pst, both in place); 4 = round 4.  A binding compares sed_abi_version() with the header it was written
 * against before the first call (transformer4sed_amd/_lib.py does). */
#define SED_HIP_ABI_VERSION 7

#ifdef __cplusplus
extern "C" {
#endif

int sed_abi_version(int reserved);

/* ------------------------------------------------------------------ frontend / augment / post-process */
/* PasstFeatureExtractor.forward + .normalize (src/models/passt/passt_feature_extraction.py:46-94).
 * wav [B,L] f32 -> out [B,128,T] f32; window [800] symmetric Hann, twiddle [1024] float2 exp(-2 pi i k/1024),
 * melw [128,513] dense Kaldi bank (host-built per (fmin,fmax)), mel_range [128,2] non-zero bin range.  maxbits_tmp: scratch of 32 B words
 * (32 partial |max| values per clip).  do_log bit 0: (log(mel + 1e-5) + 4.5) / 5; bit 1 (test aid): the round-5 kernel. */
int sed_logmel_fwd(const float* wav, float* out, uint32_t* maxbits_tmp, const float* window, const float* twiddle,
                   const float* melw, const int* mel_range, int B, int L, int T, int do_log, hipStream_t stream);
/* Polyphase FIR resampler, the device counterpart of the offline tool src/utils/resample.py:10-14 (16 kHz -> 32 kHz): x [B,L] ->
 * y [B,Lout], taps h [ntaps] and alignment (n_pre_pad, n_pre_remove) as scipy.signal.resample_poly defines them (host-built). */
int sed_resample_poly(const float* x, float* y, const float* h, int B, int L, int Lout, int up, int down, int ntaps,
                      int n_pre_pad, int n_pre_remove, hipStream_t stream);
/* The same resampler on 16-bit PCM file bodies (round 5, the batched input pipeline data.WavBatchStream -- replaces the per-file
 * librosa.load + resample of src/preprocess/feats_extraction.py:7-12 / src/utils/resample.py:10-14): x [B,L] int16 zero padded,
 * lens [B] real sample counts; int16 -> fp32 by 2^-15 inside; y [B,Lout] with zeros from ceil(lens[b] up / down) on (pad_wav). */
int sed_resample_poly_pcm16(const int16_t* x, const int* lens, float* y, const float* h, int B, int L, int Lout, int up, int down,
                            int ntaps, int n_pre_pad, int n_pre_remove, hipStream_t stream);
/* frame_shift + mixup (src/preprocess/data_aug.py:11-28, 75-90); shift [B], perm [B] / cmix [B,2]={c,1-c} nullable */
int sed_roll_mix(const float* in, float* out, const int* shift, const int* perm, const float* cmix, int B, int F, int T,
                 int clamp01, hipStream_t stream);
/* freq_nonlinear (gather-lerp table) + filt_aug additive term (data_aug.py:150-192, 207-222) */
int sed_warp_filt(const float* in, float* out, const int* kidx, const float* lam, const float* add, int B, int F, int T,
                  hipStream_t stream);
/* In-place box fill x[:, f0:f1, t0:t1] = value on x [B,F,T]: time_mask (src/preprocess/data_aug.py:93-108: features 0 / 1e-4 and labels 0
 * over a time range) and torchaudio's FrequencyMasking as called at data_aug.py:136-140 (a frequency range, all clips, value 0).  An empty
 * range is a no-op (python slice semantics are resolved by the caller). */
int sed_mask_box(float* x, int B, int F, int T, int f0, int f1, int t0, int t1, float value, hipStream_t stream);
/* add_noise (data_aug.py:195-204): out[b] = x[b] + noise[b] * std(x[b]) / snr_lin[b], std unbiased over the n = F*T elements of a clip
 * (torch.std over dims (1,2)), snr_lin = 10^(snr_dB/20); noise = the caller's standard-normal draws.  part: scratch fp32 [B, 64, 3]. */
int sed_add_noise(const float* x, const float* noise, const float* snr_lin, float* part, float* out, int B, int n,
                  hipStream_t stream);
/* median_filter_torch (src/postprocess/filter.py:4-36, mode 0); scipy median / max filter as called at
 * src/codec/decoder.py:91,94 (modes 1, 2); optional per-(clip,class) weak-mask multiplier (decoder.py:22-23,80) */
int sed_median_filter(const float* in, float* out, const int* sizes, const float* scale, int B, int T, int C, int mode,
                      hipStream_t stream);
/* the same filters with the caller's bound `max_size` on the window sizes (which live on the device): with it, windows of 24 frames and more
 * (the evaluation path filters with 32 / 128 frames, recipes/desed/finetune/train.py:221-227) take an order-statistics kernel -- global
 * ranks of the padded column once, windows as bitmaps over ranks -- instead of the O(k^2) rank search; same element selected, bit-exact.
 * A window larger than max_size traps. */
int sed_median_filter_k(const float* in, float* out, const int* sizes, const float* scale, int B, int T, int C, int mode,
                        int max_size, hipStream_t stream);

/* ------------------------------------------------------------------ GEMM family (nn.Linear / conv / autograd GEMMs) */
/* C[M,N] = A[M,K] . B[N,K]^T, bf16 operands, fp32 accumulate, fused epilogue `epi`:
 * 0 outF=acc*alpha+bias | 1 outF=resF+acc+bias (resF may alias outF) | 2 outH=bf16(acc+bias) | 3 outH=h, outH2=gelu(h) | 4 outH=acc*gelu'(auxH)
 * `f16`: 0 = bf16 operands / outputs, 1 = IEEE half, 3 = half, but tensors only the bf16 backward consumes are written as bf16
 *        (epi 3 / 8: the pre-activation outH;  sed_gemm_qkv: row-major v, qt, kt, q2t).
 * 5 atomicAdd(outF, acc*alpha) (split-K; `ksplit` is a hint, the library re-derives it for the tile shape it dispatches) | 7 outF and outH | 8 outH=h, outF=gelu(h) fp32.  Replaces F.linear at src/models/passt/passt.py:271,274,
 * 332,342; src/models/transformer/transformerXL.py:382,493,584; src/models/passt/passt_sed.py:196; conv2d passt.py:307 */
int sed_gemm_nt(const void* A, const void* B, int M, int N, int K, int lda, int ldb, int epi, const float* bias,
                const float* resF, float* outF, void* outH, void* outH2, const void* auxH, int ldc, float alpha,
                int ksplit, int f16, hipStream_t stream);
/* sed_gemm_qkv (all its outputs) on two of the three split-precision terms: A plain f16 [M, K], Wsplit the split weight image [N, 3K] =
 * [hi | hi | lo]; result A . (hi + lo)^T.  The context network's in_proj (src/models/transformer/transformerXL.py:382-384), whose output is
 * insensitive to the activation's lo part (tools/err_sim.py).  256 x 256 kernel: M >= 1024. */
int sed_gemm_qkv_w2s(const void* A, const void* Wsplit, const float* bias, int M, int K, int heads, int seq, int seq_pad, void* q,
                     void* k, void* v, void* qt, void* kt, void* vt, void* q2, void* q2t, const float* pos_u, const float* pos_v,
                     int f16, hipStream_t stream);
/* Number of CUs the persistent GEMM kernels (256 x 256 tiles: every forward / dX GEMM; the TN weight-gradient kernel's split count)
 * size their grids for; 0 = all (default; the environment variable SED_GEMM_CUS is read when nothing was set).  The one process-wide
 * setting of the library: a data-parallel job whose RCCL kernels run beside the backward leaves their CUs out (ddp.py).  The persistent
 * kernels walk their tiles dynamically (per-XCD counters), so a workgroup that starts late only shortens the others' share. */
int sed_gemm_set_cu_budget(int n_cus);
/* Measurement aid: n_cus single-wave workgroups (one per CU) that idle for `usec` microseconds on `stream` -- the stand-in for
 * communication kernels in tools/cu_steal.py (profiles/r4_cu_steal.txt).  Not used by the product path. */
int sed_debug_hold_cus(int n_cus, int usec, hipStream_t stream);
/* Test aid: synchronises the device and returns the number of non-zero words in the dynamic-walk counter ring (0 when idle). */
int sed_debug_tile_counters_dirty(int reserved);
/* sed_gemm_nt / sed_gemm_qkv with a ROW-GROUP bias: row m additionally gets gbias[(m / gb_rows) * N + n] (fp32 [M / gb_rows, N]; gb_rows =
 * tokens per clip >= 128, M % gb_rows == 0; epilogues 0-3, 7, 8).  Carries the weight-rounding correction of the evaluation-mode encoder:
 * mean_t(x) . (W - f16(W))^T per clip, added to the F.linear results of src/models/passt/passt.py:332,342 and timm Mlp fc1 / fc2
 * (same reference lines as sed_gemm_nt) so that frame posteriors stay inside BASELINE.json's 1e-3 at the validation temperature. */
int sed_gemm_nt_gb(const void* A, const void* B, int M, int N, int K, int lda, int ldb, int epi, const float* bias,
                   const float* resF, float* outF, void* outH, void* outH2, const void* auxH, int ldc, float alpha, int f16,
                   const float* gbias, int gb_rows, hipStream_t stream);
int sed_gemm_qkv_gb(const void* A, const void* W, const float* bias, int M, int K, int heads, int seq, int seq_pad, void* q,
                    void* k, void* v, int f16, const float* gbias, int gb_rows, hipStream_t stream);
/* The same F.linear calls with TWO-TERM weights (evaluation-mode encoder, default): A [M, K] f16 against B [N, 2K] =
 * [f16(W) | f16(W - f16(W))] (sed_split3_f16 mode 2); the A panel is walked twice, so the result carries the fp32 weight to ~2^-19 at the
 * cost of twice the MFMA work.  256^2 kernel only: N % 256 == 0, M >= 1024, f16; epi 1 (fp32 + residual) or 3 (GELU, outH nullable). */
int sed_gemm_nt_w2(const void* A, const void* B, int M, int N, int K, int lda, int ldb, int epi, const float* bias,
                   const float* resF, float* outF, void* outH, void* outH2, int ldc, int f16, hipStream_t stream);
int sed_gemm_qkv_w2(const void* A, const void* W, const float* bias, int M, int K, int heads, int seq, int seq_pad, void* q,
                    void* k, void* v, int f16, hipStream_t stream);
/* ... with the lo product x . (W - f16(W))^T on the fp8 matrix path (v_mfma_scale_f32_16x16x128_f8f6f4): half of an f16 K pass for a
 * term that is 2^-12 of the result and needs ~4 significant bits of either factor.  Operand rows carry both images: A [M][K f16 | K e4m3]
 * (row pitch lda halfs >= 3K / 2; the e4m3 half = 2^-2 x the activation: sed_fp8_tail, or the producing kernels' fp8 flags), B [N][K f16 |
 * K e4m3] = sed_weight_two_term_f8(W, s) with f8_exp = s.  K % 128 == 0; otherwise like sed_gemm_nt_w2 / sed_gemm_qkv_w2. */
int sed_gemm_nt_w2f8(const void* A, const void* B, int M, int N, int K, int lda, int ldb, int epi, const float* bias,
                     const float* resF, float* outF, void* outH, void* outH2, int ldc, int out_e4m3, int f8_exp, hipStream_t stream);
/* (out_e4m3 != 0, epi 3 with outH NULL: outH2 rows are [N f16 | N e4m3], ldc >= 3N / 2 -- the fc1 activation leaves as fc2's A operand) */
int sed_gemm_qkv_w2f8(const void* A, const void* W, const float* bias, int M, int K, int heads, int seq, int seq_pad, void* q,
                      void* k, void* v, int f8_exp, hipStream_t stream);
/* sed_gemm_nt_gb with epi 3 (fused GELU, outH NULL) whose result leaves as rows [N f16 | N e4m3] (ldc >= 3N / 2): fc1 of the evaluation-mode
 * encoder on f16 weights + the per-clip mean correction, writing fc2's two-image A operand.  N % 256 == 0, M >= 1024, f16. */
int sed_gemm_nt_gb_e4m3(const void* A, const void* B, int M, int N, int K, int lda, int ldb, const float* bias, void* outH2, int ldc,
                        const float* gbias, int gb_rows, hipStream_t stream);
/* x rows [K f16 | K e4m3] of pitch ld halfs: fills the e4m3 half (OCP e4m3, round to nearest even, clamped to +-448) with 2^-2 x the f16 half */
int sed_fp8_tail(void* x, int M, int K, int ld, hipStream_t stream);
/* fp32 weight [N, K] -> out rows [f16(W) | e4m3(2^s (W - f16(W)))], 3K bytes each */
int sed_weight_two_term_f8(const float* w, void* out, int64_t N, int K, int s, hipStream_t stream);
/* LayerNorm FOLDED into the two Linear layers around it -- timm Block.forward `x = x + attn(norm1(x))`, `x = x + mlp(norm2(x))`
 * (src/models/passt/passt.py:360-363 with the F.linear calls of :332,342 and Mlp fc1 / fc2) for no-grad f16 passes (teacher, inference), so
 * that the normalised tensor never exists in memory:
 *   sed_gemm_nt_lnp   the residual Linear (proj / fc2): outF = residual + A . B^T + bias (fp32, may alias resF) AND x16 = its f16 image AND
 *                     rowpart [M][N / 64][2] = per-row (sum, sum of squares) of every 64-column slice
 *   sed_ln_fold_stats rowpart -> rowstat [M][2] = (mean, 1 / sqrt(var + eps)) over D = 64 S columns
 *   sed_ln_fold_weight W16 [N, K] = f16(gamma[k] W[n, k]), colS[n] = sum_k W16[n, k], colC[n] = sum_k beta[k] W[n, k] + bias[n]
 *   sed_gemm_nt_lnc / sed_gemm_qkv_lnc  the Linear AFTER the LayerNorm (fc1 + GELU; qkv with the head split), fed with x16 (the RAW
 *                     stream) and W16: out[m, n] = f(rstd[m] * (acc[m, n] - mean[m] * colS[n]) + colC[n])  ==  f(Linear(LayerNorm(x)))
 * 256^2 kernel only: N % 256 == 0, M >= 1024, K % 64 == 0, f16 operands. */
int sed_gemm_nt_lnp(const void* A, const void* B, int M, int N, int K, int lda, int ldb, const float* bias, const float* resF,
                    const void* res_hi, const void* res_lo, float* outF, void* x16, void* out_lo, float* rowpart, int ldc,
                    hipStream_t stream);
/* (sed_gemm_nt_lnp, split-plane stream: between two folded blocks the residual stream can live as two f16 planes hi + lo instead of fp32
 *  -- res_hi / res_lo != NULL: the residual is read as hi + lo instead of resF; out_lo != NULL: the result is written as x16 (hi) + out_lo
 *  and outF is left alone.  The hi plane is the next GEMM's A operand, so a producer moves 8 instead of 10 bytes per element.) */
/* (sed_gemm_nt_lnp8: the same with the lo plane as BYTES -- res_lo8 / out_lo8 are M * ldc uint8 in slab-major order [ldc / 64][M][64]
 *  (a plane private to this entry point: written and read only here) and the stream value is
 *  hi * (1 + (q - 128) * 2^-18): the f16 rounding residual (at most |hi| 2^-11) in 256 steps, the stream to ~2^-19 relative.  A producer
 *  then moves 6 bytes per element.  Planes written by one entry point must be read by the same one.) */
int sed_gemm_nt_lnp8(const void* A, const void* B, int M, int N, int K, int lda, int ldb, const float* bias, const float* resF,
                     const void* res_hi, const void* res_lo8, float* outF, void* x16, void* out_lo8, float* rowpart, int ldc,
                     hipStream_t stream);
/* (with sed_gemm_nt_lnp8 the hi plane -- res_hi in, x16 out -- is slab-major too, [ldc / 64][M][64] f16: a K tile of the consumer and a
 *  wave's patch of the producer are contiguous runs.  sed_gemm_nt_lnc8 / sed_gemm_qkv_lnc8 are the consumers that read it: same
 *  arguments as sed_gemm_nt_lnc / sed_gemm_qkv_lnc, lda = K.  sed_gemm_nt_lnc8 with ldc = 64 (N > 64) writes its output slab-major as
 *  well, [N / 64][M][64]; sed_gemm_nt_lnp8 with lda = 64 (K > 64) reads such an A operand: the fc1 activation of a folded block.) */
int sed_gemm_nt_lnc8(const void* A, const void* B, int M, int N, int K, int lda, int ldb, const float* colC, const float* colS,
                     const float* rowstat, void* outH2, int ldc, hipStream_t stream);
int sed_gemm_qkv_lnc8(const void* A, const void* W, const float* colC, const float* colS, const float* rowstat, int M, int K, int heads,
                      int seq, int seq_pad, void* q, void* k, void* v, hipStream_t stream);
int sed_ln_fold_stats(const float* rowpart, float* rowstat, int M, int S, int D, float eps, hipStream_t stream);
int sed_ln_fold_weight(const float* W, const float* gamma, const float* beta, const float* bias, void* W16, float* colS, float* colC,
                       int N, int K, hipStream_t stream);
int sed_gemm_nt_lnc(const void* A, const void* B, int M, int N, int K, int lda, int ldb, const float* colC, const float* colS,
                    const float* rowstat, void* outH2, int ldc, hipStream_t stream);
int sed_gemm_qkv_lnc(const void* A, const void* W, const float* colC, const float* colS, const float* rowstat, int M, int K, int heads,
                     int seq, int seq_pad, void* q, void* k, void* v, hipStream_t stream);
/* operands of that correction: per-clip token means of a 16-bit activation x [groups * rows, K] -> [groups, K] (K % 256 == 0; every
 * step-th token), and the f16 image of scale * (w - f16(w)) for an fp32 weight of n (multiple of 4) elements */
int sed_group_colmean(const void* x, void* out, int groups, int rows, int K, int step, int f16, hipStream_t stream);
/* ... over rows of pitch ld >= K elements (the f16 half of rows [K f16 | K e4m3]) */
int sed_group_colmean_ld(const void* x, void* out, int groups, int rows, int K, int ld, int step, int f16, hipStream_t stream);
int sed_weight_residual_f16(const float* w, void* out, int64_t n, float scale, hipStream_t stream);
/* sed_gemm_nt with a narrow result: A / B are padded to N (a multiple of 128, not of 256) but only the first ncols (multiple of 4)
 * output columns exist in memory -- bias [ncols], residual and outputs [M, ldc] with ldc >= ncols.  The 16/32/64-filter layers of
 * the PMAM CNN branch (src/models/cnn/base.py:62-70) produce their [pixels, filters] matrices this way. */
int sed_gemm_nt_cols(const void* A, const void* B, int M, int N, int K, int lda, int ldb, int epi, const float* bias,
                     const float* resF, float* outF, void* outH, void* outH2, const void* auxH, int ldc, float alpha, int f16,
                     int ncols, hipStream_t stream);
/* Weight gradient in TN form: dW[M,N] (fp32, ldc) += dY[T,M]^T . X[T,N], both operands row-major [token][feature] as the forward /
 * backward left them (dY bf16, X bf16 or IEEE half) -- the autograd of F.linear's weight (same reference lines as sed_gemm_nt).
 * M % 64 == 0, N % 64 == 0, any T > 0 (256 x 256 tiles; a partly valid last tile computes on whatever lies behind its rows and stores
 * only its valid part; token rows past T read as zeros); split-K over tokens chosen by the library.  `workspace` (caller-owned, optional): with >= (256 / tiles) * M * N * 4
 * bytes (tiles = ceil(M/256) * ceil(N/256)) the splits are stored there and reduced by a second launch; otherwise atomics.
 * dbias (nullable): dbias[M] += column sums of dY over the T tokens, i.e. the bias gradient of the same linear, for free. */
int sed_gemm_dw_tn(const void* dY, const void* X, int x_f16, int T, int M, int N, int ldy, int ldx, float* dW, int ldc,
                   float* dbias, float* workspace, int64_t workspace_bytes, hipStream_t stream);
/* qkv / in_proj GEMM with head-split epilogue: q,k,v [B*H,seq,64] (each nullable), transposed copies [B*H,64,seq_pad] (nullable),
 * optional rel-pos biased queries q = q+u, q2 = q+v (passt.py:332-333; transformerXL.py:382-384,497-503) */
int sed_gemm_qkv(const void* A, const void* W, const float* bias, int M, int K, int heads, int seq, int seq_pad, void* q,
                 void* k, void* v, void* qt, void* kt, void* vt, void* q2, void* q2t, const float* pos_u,
                 const float* pos_v, int f16, hipStream_t stream);
int sed_cast_f32_bf16(const float* in, void* out, int64_t n, int f16, hipStream_t stream);
int sed_f16_to_bf16_inplace(void* p, int64_t n, hipStream_t stream);
/* split-precision operand image: fp32 [M,K] -> f16 [M,3K]; mode 0 [hi|lo|hi] (activations), 1 [hi|hi|lo] (weights);
 * mode 2: f16 [M,2K] = [hi|lo], the two-term weight image of sed_gemm_nt_w2 / sed_gemm_qkv_w2 */
int sed_split3_f16(const float* in, void* out, int64_t M, int K, int mode, hipStream_t stream);
/* in [R,C] -> outT [C,Rpad] (zero padded; nullable), optional straight 16-bit copy and fp32 column sums (+=).
 * kinds: 0 bf16, 1 f32 (input only), 2 f16 */
int sed_transpose_to_bf16(const void* in, int in_kind, int R, int C, int ldin, void* outT, int Rpad, int outT_kind,
                          void* outS, int outS_kind, float* colsum, hipStream_t stream);
/* tiny fp32 linears (AT-head query / out_proj / 768->10 classifier; src/models/pooling.py:45-51, passt_sed.py:143) */
int sed_small_linear(const float* a, const float* w, const float* b, float* out, int M, int N, int K, int act,
                     hipStream_t stream);
int sed_small_linear_bwd(const float* a, const float* w, const float* out, const float* dout, float* da, float* dw,
                         float* db, int M, int N, int K, int act, hipStream_t stream);

/* ------------------------------------------------------------------ attention */
/* encoder MHSA (src/models/passt/passt.py:335-341), flash style; Q, K, V head-split [B*H, N, 64] 16-bit (V row-major: the kernel takes
   V^T out of its LDS tile with transposing reads); O [B,N,768] 16-bit, LSE [B*H,N] (log2 domain).
   f16: bit 0 = IEEE half operands (else bf16); bit 1 = O head-major [H][B*N][64] (the slab-major A operand of sed_gemm_nt_lnp8, lda = 64);
   bit 2 (with bit 0, without bit 1) = O rows [768 f16 | 768 e4m3], pitch 1152 halfs (the A operand of sed_gemm_nt_w2f8) */
int sed_mhsa_fwd(const void* Q, const void* K, const void* V, void* O, float* LSE, int B, int H, int N, int Npad,
                 int f16, hipStream_t stream);
int sed_mhsa_bwd_prep(const void* dO, const void* O, float* Dtmp, void* dOh, void* dOt, int B, int H, int N, int Npad,
                      int o_f16, hipStream_t stream);
/* dqkv [B*N, 3*768] bf16 (dq | dk | dv).  f16 != 0: Q, K, O are IEEE half (as the f16 forward wrote them, used for the
 * score recompute), the gradient-side operand dO is bf16; V is bf16, or with v_f16 != 0 the IEEE half tensor the forward saved (its bf16
 * operand image is made inside the kernels).  Transposed operands are taken out of row-major LDS tiles with
 * transposing reads (no Q^T / K^T / dO^T copies); dO and O are read in their token-major [B, N, 768] layout and D = rowsum(dO * O)
 * is produced by the dQ kernel into Dtmp [B*H, N] -- no pre-pass; dOh is unused (may be NULL). */
int sed_mhsa_bwd(const void* Q, const void* K, const void* V, const void* O, const void* dO, const float* LSE, float* Dtmp,
                 void* dOh, void* dqkv, int B, int H, int N, int Npad, int f16, int v_f16, hipStream_t stream);
/* Transformer-XL rel-pos attention (src/models/transformer/transformerXL.py:493-576 incl. rel_shift 254-297).  O [B, T, H 64] 16-bit, or
 * fp32 with o_f32; O_split (nullable, fp32 mode only): the [hi | lo | hi] f16 split-precision image [B T, 3 H 64] of the same values --
 * out_proj's A operand (transformerXL.py:584), written here instead of by a sed_split3_f16 pass over O */
int sed_relpos_attn_fwd(const void* Qu, const void* Qv, const void* K, const void* Vt, const void* P, void* O, void* O_split,
                        float* LSE, int B, int H, int T, int Tpad, int Rpad, int f16, int o_f32, hipStream_t stream);
int sed_relpos_attn_bwd(const void* Qu, const void* Qut, const void* Qv, const void* Qvt, const void* K, const void* Kt,
                        const void* V, const void* P, const void* Pt, const void* O, const void* dO, const float* LSE,
                        float* Dtmp, void* dOh, void* dOt, void* dqkv, void* dSt, void* Pst, float* dP, float* du, float* dv, int B,
                        int H, int T, int Tpad, int Rpad, int need_param_grads, int f16, int o_kind, hipStream_t stream);
/* (dSt, Pst: [B H, Tpad, Tpad] bf16 scratch slabs, ZERO outside the region the kernels write -- rows = keys, columns = queries.  The dQ
 *  kernel stores dS^T there for the positional-table gradient and, when Pst is given, P^T as well: dK and dV are then two contractions
 *  over the stored slabs (a streaming kernel) instead of a second recomputation of the scores; Pst = NULL keeps the recomputing kernel.) */

/* ------------------------------------------------------------------ norms / glue / heads / optimiser */
/* nn.LayerNorm over D=768 (passt.py:361-362,580; passt_sed.py:128; timm Block norms); y = LN(in_scale*x).
 * f16: 0 bf16 / 1 IEEE-half y_bf16 [M, D]; 4: y_bf16 is the split-precision operand image [M, 3 D] = [hi | lo | hi] f16 of the result
 * (the context-network GEMMs' A operand, transformerXL.py:31-35 -- what sed_split3_f16 would make from y_f32 in a second pass);
 * 8: y_bf16 rows are [D f16 | D e4m3] (pitch 3 D / 2 halfs), the A operand of sed_gemm_nt_w2f8 / sed_gemm_qkv_w2f8 */
int sed_layernorm_fwd(const float* x, const float* gamma, const float* beta, float eps, float in_scale, void* y_bf16,
                      float* y_f32, float* mean, float* rstd, int M, int D, int f16, hipStream_t stream);
/* ... with the result written twice, as IEEE half (y_f16: the forward GEMM's operand) and as bf16 (y_bf16: the saved operand of the backward's
 * weight-gradient GEMM, rounded once from fp32) */
int sed_layernorm_fwd_dual(const float* x, const float* gamma, const float* beta, float eps, float in_scale, void* y_f16, void* y_bf16,
                           float* mean, float* rstd, int M, int D, hipStream_t stream);
int sed_layernorm_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                      float in_scale, float* dx, int accumulate, float* dgamma, float* dbeta, int M, int D,
                      hipStream_t stream);
/* the same backward, also writing dx16 [M, D] = bf16 image of the residual-stream gradient it leaves in dx: the dY operand of the
 * weight-gradient and dX GEMMs of the block below (autograd of timm Block.forward, passt.py:360-363) without a separate cast pass */
int sed_layernorm_bwd_x16(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                          float in_scale, float* dx, int accumulate, float* dgamma, float* dbeta, void* dx16, int M, int D,
                          hipStream_t stream);
/* patch embedding plumbing (passt.py:302-315, 503-569) */
int sed_im2col(const float* mel, void* cols, int B, int T, int tstart, int tp, int f16, hipStream_t stream);
int sed_assemble_tokens(const float* conv, const float* cls, const float* dist, const float* new_pos,
                        const float* freq_pe, const float* time_pe, int toffset, float* x, int B, int tp,
                        hipStream_t stream);
int sed_assemble_tokens_bwd(const float* dx, void* dconv, float* dcls, float* ddist, float* dnew_pos, float* dfreq,
                            float* dtime, int toffset, int B, int tp, hipStream_t stream);
/* PaSST_SED.f_pool 'mean_pool' (passt_sed.py:199-218) */
int sed_fpool_fwd(const float* x, const float* gamma, const float* beta, float eps, float* pooled, float* mean,
                  float* rstd, int B, int tp, hipStream_t stream);
int sed_fpool_bwd(const float* dpooled, const float* x, const float* mean, const float* rstd, const float* gamma,
                  float* dtok_tmp, float* dx_acc, float* dgamma, float* dbeta, int B, int tp, hipStream_t stream);
/* InterpolateModule linear x ratio (+ replicated last frame) (passt_sed.py:13-34,258-259) */
int sed_interp_fwd(const float* in, float* out, int B, int tin, int pad, int ratio, hipStream_t stream);
int sed_interp_bwd(const float* dout, float* din, int B, int tin, int pad, int ratio, hipStream_t stream);
/* EncoderSlideWindow merge + global/local mix (src/models/encoder_slide_window.py:16-36; passt_sed.py:266-271) */
int sed_window_mix(const float* pooled_win, const int* lefts, const int* tps, const int* offs, int nW, float* x,
                   float mix, int B, int T, int ratio, hipStream_t stream);
/* its backward (autograd of encoder_slide_window.py:16-36 + the mix of passt_sed.py:266-271, student with encoder_win=True):
   dx [B, T, 768] -> dpooled_win (same packed layout as pooled_win, `rows` rows in all) and dglobal = (1 - mix) dx */
int sed_window_mix_bwd(const float* dx, const int* lefts, const int* tps, const int* offs, int nW, float* dpooled_win,
                       float* dglobal, float mix, int B, int T, int ratio, int rows, hipStream_t stream);
/* MlmModule.setence_mask application (src/models/transformer/mask.py:62-85) and masked MSE (mlm_passt/train.py:36-38) */
int sed_mlm_apply(const float* x, const float* mask_token, const uint8_t* action, const int* src_idx, float* out,
                  int rows, hipStream_t stream);
int sed_mlm_apply_bwd(const float* dout, const uint8_t* action, const int* src_idx, float* dx_zeroed, float* dtoken,
                      int rows, hipStream_t stream);
/* n_masked_rows_dev (nullable): the row count as a device int, read by the kernel instead of n_masked_rows (no host sync) */
int sed_masked_mse(const float* pred, const float* target, const uint8_t* mask, int n_masked_rows, const int* n_masked_rows_dev,
                   float* loss_zeroed, float* dpred, float* dtarget, int rows, hipStream_t stream);
/* The six mean-teacher loss terms of Trainer.train and their gradients in one call (recipes/desed/finetune/train.py:160-191):
   l_strong = BCE(s_strong[:strong_n], labels[:strong_n]), l_weak = BCE(s_weak[ws], labels_weak[ws]), l_at = BCE(s_at[ws], labels_weak[ws]),
   lc_strong = MSE(s_strong, t_strong), lc_weak = MSE(s_weak, t_at), lc_at = MSE(s_at, t_at)   (ws = clips weak_lo .. weak_lo + weak_n - 1),
   total = l_strong + w_weak l_weak + w_at l_at + w_cons (lc_strong + w_weak_cons lc_weak + w_at lc_at); torch.nn.BCELoss / MSELoss
   semantics (mean reduction, log clamped at -100, BCE gradient denominator clamped at 1e-12).
   out[8] = {total, l_strong, l_weak, l_at, lc_strong, lc_weak, lc_at, 0}; scratch[8] is zeroed by the call;
   d_strong [B,C,T], d_weak [B,C], d_at [B,C] = d total / d (student outputs). */
int sed_sed_losses(const float* s_strong, const float* s_weak, const float* s_at, const float* t_strong, const float* t_at,
                   const float* labels, const float* labels_weak, int B, int C, int T, int strong_n, int weak_lo, int weak_n,
                   float w_weak, float w_weak_cons, float w_at, float w_cons, float* scratch, float* out, float* d_strong,
                   float* d_weak, float* d_at, hipStream_t stream);
/* classifier + sigmoid(x/temp) + pad mask + linear-softmax pooling (passt_sed.py:285-296) */
int sed_head_fwd(const float* x, const float* W, const float* bias, float temp, const uint8_t* pad_mask, float* strong,
                 float* weak, float* sums, int B, int T, int C, hipStream_t stream);
int sed_head_bwd(const float* x, const float* W, const float* strong, const float* sums, const float* dstrong,
                 const float* dweak, float temp, float* dx, float* dW, float* db, int B, int T, int C,
                 hipStream_t stream);
/* AttentionPooling (src/models/pooling.py:45-51): 1-query MHA over the patch tokens; kv [B,N,1536] bf16 */
int sed_attnpool_fwd(const void* kv, const float* q, float* out, float* probs, int B, int N, int H, int f16,
                     hipStream_t stream);
int sed_attnpool_bwd(const void* kv, const float* q, const float* probs, const float* dout, void* dkv, float* dq, int B,
                     int N, int H, int f16, hipStream_t stream);
/* torch.optim.AdamW step (recipes/desed/setting.py:254-258) fused with update_ema (src/utils/scheduler.py:125-130) */
int sed_adamw_ema(float* p, const float* g, float* m, float* v, float* ema, int64_t n, float lr, float wd, float beta1,
                  float beta2, float eps, int step, float ema_alpha, int do_adam, hipStream_t stream);

/* every weight image of a model in one launch (what sed_transpose_to_bf16 + sed_split3_f16 produce per weight): desc is a device
 * table of n_desc x 16 int64 {fp32 master, transposed bf16 image [C, R] or 0, straight 16-bit image [R, C] or 0,
 * split-precision f16 image [R, 3C] = [hi | hi | lo] or 0, R (multiple of 16), C (multiple of 64), straight-image kind
 * (0 bf16, 2 f16), index of the weight's first 64 x 64 tile, gather plan int32 [R, C] or 0 (image element -> master element; without
 * one the master IS [R, C]), scale plan fp32 [R, C] or 0, LoRA A fp32 [r, C] or 0, LoRA B fp32 [R, r], r, LoRA scaling as float bits,
 * 0, 0}; image = gather(master) * scale + s B A (src/models/lora/layers.py:148-151: the train-mode LoRA linear W x + s B (A x) as one
 * weight; the plans: zero-padded context-network / CNN images of the PMAM model); total_tiles = sum of ceil(R / 64) * (C / 64). */
int sed_weight_images(const int64_t* desc, int n_desc, int total_tiles, hipStream_t stream);

/* ---------------------------------------------------------------------------------------------------------------------------
 * PMAM variant of the model path: PaSST_CNN.forward (src/models/cnn_transformer/passt_cnn.py:31-88).
 * ------------------------------------------------------------------------------------------------------------------------- */
/* LoRA merge W_eff = W + scaling * B A, fp32 [n_out, k_in] (src/models/lora/layers.py:120-153; r = rank, A [r, k_in], Bm [n_out, r]) */
int sed_lora_merge(const float* W, const float* A, const float* Bm, float scaling, float* out, int n_out, int k_in, int r,
                   hipStream_t stream);
/* LayerNorm of width D (multiple of 128, <= 1024): the 384-wide context network (src/models/transformer/transformerXL.py:31-35).
 * Same outputs as sed_layernorm_fwd (16-bit and/or fp32 result, per-row mean / rstd for the backward; all nullable). */
int sed_ln_fwd_any(const float* x, const float* gamma, const float* beta, float eps, float in_scale, void* y16, float* y32,
                   float* mean, float* rstd, int M, int D, int f16, hipStream_t stream);
/* MlmModule.setence_mask application on rows of width C (src/models/transformer/mask.py:62-85); see sed_mlm_apply */
int sed_mlm_apply_c(const float* x, const float* mask_token, const uint8_t* action, const int* src_idx, float* out, int rows,
                    int C, hipStream_t stream);
/* CNN branch (src/models/cnn/base.py:62-113): 3x3 / pad 1 patch gathers for the im2col GEMMs.  Layer 0 reads the mel
 * spectrogram [B, 128, T] (the reference's input.transpose(1, 2).unsqueeze(1), passt_cnn.py:51) into col [B*T*128, 64]
 * (9 taps + zeros); later layers read NHWC 16-bit activations X [B, H, W, Cp] into col [B*H*W, Kp], column = tap * C + c. */
int sed_conv0_im2col(const float* mel, void* col, int B, int T, int f16, hipStream_t stream);
/* first convolution with 16 filters, direct on the fp32 spectrogram (no patch matrix): Y [B*T*128, 16] fp32 = conv3x3(mel) + bias, Wc =
 * conv0.weight [16, 1, 3, 3] (kernel rows over time, columns over frequency); and its weight / bias gradient from dY bf16 [pixels, ldy]:
 * dW[c, tap] += sum dY[m, c] patch(m, tap) (row stride ldw >= 9), dbias[c] += sum dY[m, c] (nullable).  s1 / s2 (nullable pair): the
 * BatchNorm batch-statistics sums of Y (sed_colstats mode 0) accumulated by the same pass */
int sed_conv0_fwd16(const float* mel, const float* Wc, const float* bias, float* Y, int B, int T, float* s1, float* s2, hipStream_t stream);
int sed_conv0_dw16(const void* dY, int ldy, const float* mel, float* dW, int ldw, float* dbias, int B, int T, hipStream_t stream);
int sed_conv3x3_im2col(const void* X, void* col, int B, int H, int W, int C, int Cp, int Kp, hipStream_t stream);
/* BatchNorm2d(eps 1e-3) as the per-channel affine Z = Y * a + b (base.py:72-75): 16-bit operand of the ContextGating GEMM,
 * channels C..Cp-1 zero.  Y [M, ldy] fp32 is the convolution output. */
int sed_bn_act(const float* Y, int ldy, const float* a, const float* b, void* Z, int64_t M, int C, int Cp, int f16,
               hipStream_t stream);
/* ContextGating (base.py:19-30) + Dropout (mask [M, C] bytes, nullable; kept values * drop_scale) + AvgPool2d((ph, pw)):
 * out[b, ho, wo, c] = mean of (Y a + b) * sigmoid(L) * keep.  out16 NHWC with channel pad Cpo and/or out32 [rows, C]. */
int sed_cg_pool(const float* Y, int ldy, const float* a, const float* b, const float* L, int ldl, const uint8_t* mask,
                float drop_scale, void* out16, float* out32, int B, int H, int W, int C, int Cpo, int ph, int pw, int f16,
                hipStream_t stream);
/* 'attention' frequency pooling (src/models/pooling.py:37-51, 6 heads; passt_sed.py:211-215): kv [B*N, 1536] 16-bit (k | v of the
 * out_norm'ed tokens), q [768] projected query -> out [B*tp, 768] (16-bit and/or fp32), probs [B*tp, 6, 12] (nullable). */
int sed_fpool_attn_fwd(const void* kv, const float* q, void* out16, float* out32, float* probs, int B, int N, int tp, int f16,
                       hipStream_t stream);
/* projector merge (passt_cnn.py:57-62): out[b, j] = lerp_r1(pad1(P1))[j] + mw[0] * lerp_r2(P2)[j], F.interpolate(linear,
 * align_corners=False) arithmetic; P1 [B, tp1, C], P2 [B, tp2, C], (tp1 + pad1) * r1 == tp2 * r2 output frames. */
int sed_pmam_merge(const float* P1, const float* P2, const float* mw, float* out, int B, int tp1, int pad1, int r1, int tp2,
                   int r2, int C, hipStream_t stream);

/* --- backward of the PMAM path --- */
int sed_ln_bwd_any(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma, float in_scale,
                   float* dx, int accumulate, float* dgamma, float* dbeta, int M, int D, hipStream_t stream);
int sed_mlm_apply_bwd_c(const float* dout, const uint8_t* action, const int* src_idx, float* dx_zeroed, float* dtoken, int rows,
                        int C, hipStream_t stream);
/* dP1 / dP2 / d merge_weight of sed_pmam_merge (dmw += , nullable) */
int sed_pmam_merge_bwd(const float* g, const float* P2, const float* mw, float* dP1, float* dP2, float* dmw, int B, int tp1,
                       int pad1, int r1, int tp2, int r2, int C, hipStream_t stream);
/* LoRA gradients by skinny products (src/models/lora/layers.py:148-151: y = x W^T + s (x A^T) B^T, A [r, in], B [out, r]; autograd:
 * dB = s dy^T (x A^T), dA = (s dy B)^T x) -- one pass over x or dy each, the gradient of the merged weight is never formed:
 *   rowproj    out[M, r] fp32 = scale * X[M, K] (16-bit, row stride ldx) . Wm, Wm fp32 [r, K] (w_kxr 0) or [K, r] (1); K % 32 == 0, r <= 16
 *   colreduce  G += scale * sum_m P[m, j] Y[m, c], P fp32 [M, r], Y 16-bit [M, C] (row stride ldy); G fp32 [C, r] (g_cxr 1) or [r, C] (0);
 *              C % 4 == 0, r <= 8 */
int sed_lora_rowproj(const void* X, int x_f16, int M, int K, int ldx, const float* Wm, int w_kxr, int r, float scale, float* out,
                     hipStream_t stream);
int sed_lora_colreduce(const void* Y, int y_f16, int M, int C, int ldy, const float* P, int r, float scale, float* G, int g_cxr,
                       hipStream_t stream);
/* weight / bias gradient of a 16- or 32-filter layer as a streaming reduction (the autograd of the CNN branch's gate Linear and 3x3
 * convolution, src/models/cnn/base.py:19-30, 62-100, for its first layers): dW[i, j] += sum_m dY[m, i] X[m, j] (i < n in {16, 32},
 * j < k <= 512, k % 4 == 0), dbias[i] += sum_m dY[m, i] (nullable); dY bf16 [M, ldy], X 16-bit [M, ldx], dW fp32 row stride ldw */
int sed_small_dw(const void* dY, int ldy, int n, const void* X, int x_f16, int ldx, int k, float* dW, int ldw, float* dbias, int64_t M,
                 hipStream_t stream);
/* keep-masks of nn.Dropout(p) (src/models/cnn/base.py:88-89): mask[e] = 1 with probability 1 - p (resolved to 2^-16), n % 4 == 0;
 * counter-based generator keyed by `seed` (the values differ from torch's Philox stream; only the distribution belongs to the model) */
int sed_dropout_mask(uint8_t* mask, int64_t n, float p, int64_t seed, hipStream_t stream);
/* BatchNorm2d(eps, momentum) batch statistics -> per-channel affines (src/models/cnn/base.py:72-75, torch BatchNorm semantics):
 * s1 / s2 = sed_colstats(mode 0) sums over the M rows (train: running_mean / running_var updated in place, unbiased variance) or
 * nullptr (eval: the running statistics are used); a = gamma rstd, b = beta - mean a, ah = rstd, bh = -mean rstd. */
int sed_bn_finalize(const float* s1, const float* s2, const float* gamma, const float* beta, float* run_mean, float* run_var,
                    int64_t M, int C, double momentum, double eps, float* a, float* b, float* ah, float* bh, hipStream_t stream);
/* batched small gathers / scatter-adds between fp32 masters and the padded images the kernels use; desc n_desc x 8 int64, one
 * workgroup per 256 elements (first block per descriptor in field 5):
 *   gather       {src, plan int32 [n], scale fp32 [n] or 0, dst [n], n, first block, 0, 0}:  dst[e] = src[plan[e]] * scale[e]
 *   scatter_add  {gimg, plan int32 [n] or 0, scale fp32 [n] or 0, gmaster, n, first block, C | ld_i << 32, ld_j}:
 *                gmaster[plan[e]] += scale[e] * gimg[(e / C) ld_i + (e % C) ld_j] where scale[e] != 0 (plans are injective there) */
int sed_gather_f32(const int64_t* desc, int n_desc, int total_blocks, hipStream_t stream);
int sed_scatter_add_f32(const int64_t* desc, int n_desc, int total_blocks, hipStream_t stream);
/* column sums over the M rows (+= into s1, s2): mode 0: sum A, sum A^2 (BatchNorm batch statistics, base.py:72-75 in train mode);
 * mode 1: sum A, sum A * (Bm * a + b) (BatchNorm backward sums with xhat = Y * a + b) */
int sed_colstats(const float* A, int lda, const float* Bm, int ldb, const float* a, const float* b, float* s1, float* s2,
                 int64_t M, int C, int mode, hipStream_t stream);
/* sed_bn_act + the ContextGating Linear + sed_cg_pool for a 16-filter layer in one pass (base.py:19-30, 72-100): l = Wg z + bg on the fp32
 * z = Y a + b (Wg fp32 [16, 16], bg [16]); optional side outputs for the backward: the logits Lout [pixels, 16] fp32 and the 16-bit image
 * Zout [pixels, 16] of z; out16 NHWC with channel pad Cpo (multiple of 8) and / or out32 [rows, 16] as sed_cg_pool */
int sed_cg_gate16_pool(const float* Y, int ldy, const float* a, const float* b, const float* Wg, const float* bg, const uint8_t* mask,
                       float drop_scale, float* Lout, void* Zout, void* out16, float* out32, int B, int H, int W, int Cpo, int ph, int pw,
                       int f16, hipStream_t stream);
/* backward of sed_cg_gate16_pool: dz [M, 16] fp32 = direct path + Wg^T dl (the complete gradient of z), dL16 [M, 16] bf16 = dl;
 * s1 / s2 (nullable pair, with ah / bh): the BatchNorm backward sums of dz (sed_colstats mode 1) accumulated by the same pass */
int sed_cg_gate16_pool_bwd(const float* dout, const float* Y, int ldy, const float* a, const float* b, const float* L, const float* Wg,
                           const uint8_t* mask, float drop_scale, float* dz, void* dL16, int B, int H, int W, int ph, int pw,
                           const float* ah, const float* bh, float* s1, float* s2, hipStream_t stream);
/* backward of sed_cg_pool: dzd [M, ldz] fp32 (direct path into the BatchNorm output; columns C..ldz-1 zero), dL16 [M, ldl16]
 * bf16 (gate logits; columns C.. zero) */
int sed_cg_pool_bwd(const float* dout, const float* Y, int ldy, const float* a, const float* b, const float* L, int ldl,
                    const uint8_t* mask, float drop_scale, float* dzd, int ldz, void* dL16, int ldl16, int B, int H, int W, int C,
                    int ph, int pw, hipStream_t stream);
/* BatchNorm backward with batch statistics: dY bf16 [M, ldo] from dz, xhat = Y * ah + bh and the sed_colstats(mode 1) sums */
int sed_bn_bwd(const float* dz, int ldz, const float* Y, int ldy, const float* ah, const float* bh, const float* gamma,
               const float* s1, const float* s2, void* dY, int ldo, int64_t M, int C, hipStream_t stream);
/* transpose of sed_conv3x3_im2col: dcol bf16 [B*H*W, Kp] -> dX fp32 [B, H, W, C] */
int sed_col2im3x3(const void* dcol, int Kp, float* dX, int B, int H, int W, int C, hipStream_t stream);
int sed_fpool_attn_bwd(const void* kv, const float* q, const float* probs, const float* dout, void* dkv, float* dq, int B, int N,
                       int tp, int f16, hipStream_t stream);
/* LoRA factor gradients from the merged-weight gradient: dB += s dW A^T, dA += s B^T dW (src/models/lora/layers.py:148-151) */
int sed_lora_grad(const float* dW, const float* A, const float* Bm, float scaling, float* dA, float* dB, int n_out, int k_in,
                  int r, hipStream_t stream);
/* prototype-similarity BCE of the PMAM trainer (recipes/desed/pmam/train.py:82-87, 100-106): loss[0] += mean BCE over the
 * `sel`ected frames x C classes; dlogit [B*T, 768] (nullable) = d loss / d logit; post [B*T, C] (nullable) = posteriors.
 * protos [C, 768] are the row-normalised GMM means (train.py:31); labels [B, C, T].  n_selected_dev (nullable): the number of
 * selected frames as a device int, read by the kernel instead of n_selected (no host sync). */
int sed_proto_bce(const float* logit, const float* protos, const float* labels, const uint8_t* sel, int n_selected,
                  const int* n_selected_dev, float temperature, float* loss, float* dlogit, float* post, int B, int T, int C, int D,
                  hipStream_t stream);

/* transpose of a narrow 16-bit matrix (bf16 or, in_f16 != 0, IEEE half): in [R, ldin] with C = 8, 16, 24 or 32 valid columns
 * -> outT bf16 [C, Rpad] (rows R..Rpad-1 zero) + optional fp32 column sums (+=): the weight-gradient operands of the 16/32-filter
 * CNN layers (src/models/cnn/base.py:62-70) */
int sed_transpose_narrow(const void* in, int in_f16, int R, int C, int ldin, void* outT, int Rpad, float* colsum,
                         hipStream_t stream);

/* ------------------------------------------------------------------ DASM query decoder + dual-stream head (round 5, forward) */
/* fp32 GEMM on the fp32-input matrix instruction (exact fp32 products / accumulation): C[z][M,N] = act(A[z][M,K] . B[z][N,K]^T + bias[N])
 * (+ R[z][M,N], same ldc / stride as C; nullable), z < batch with element strides (0 = shared operand); act 0 none, 1 GELU, 2 ReLU;
 * K % 32 == 0, lda / ldb / strides % 4 == 0, 16-byte aligned A / B.  The precision-critical small linears of the DASM head: nn.Linear in
 * src/models/detect_any_sound/detect_any_sound.py:76-78,125,138,401-416, the in_proj / out_proj / linear1 / linear2 of
 * at_adapter.py:36-45, and the per-clip einsum('bqc,bct->bqt') of detect_any_sound.py:394. */
int sed_gemm_f32_nt(const float* A, const float* B, const float* bias, const float* R, float* C, int M, int N, int K, int lda, int ldb,
                    int ldc, int batch, int64_t strideA, int64_t strideB, int64_t strideC, int act, hipStream_t stream);
/* softmax(q k^T / sqrt(head_dim) + mask) v with Nq != Nk, fp32 (torch.nn.MultiheadAttention inside nn.TransformerDecoderLayer,
 * at_adapter.py:24-32: cross attention of the queries over the patch tokens, self attention among the queries): Q rows
 * [b * q_batch_stride + i * ldq + h * head_dim] (q_batch_stride 0 = the same queries for every clip), K / V rows [(b * Nk + j) * ld + h *
 * head_dim], O [B, Nq, ldo]; mask [Nq, Nk] bytes, non-zero = not allowed (nullable); head_dim 32 or 64. */
int sed_xattn_f32_fwd(const float* Q, const float* K, const float* V, float* O, const uint8_t* mask, int B, int H, int Nq, int Nk,
                      int head_dim, int ldq, int ldk, int ldv, int ldo, int64_t q_batch_stride, hipStream_t stream);
/* dual-stream finish (detect_any_sound.py:317-319, 394-404): logits [B,T,Q], at_logit [B,Q], pad_mask [B,T] (nullable) ->
 * at_out = sigmoid(at_logit) [B,Q] (nullable), strong [B,Q,T] = clamp(pad ? 0 : sigmoid(logit / temp) * at_out, 1e-7, 1),
 * weak [B,Q] = clamp(sum_t strong^2 / sum_t strong, 1e-7, 1).  at_logit == NULL with clamp_strong == 0: the closed-set classifier head for
 * any class count, strong = pad ? 0 : sigmoid(logit / temp) without a clamp (src/models/cnn_transformer/passt_cnn.py:74-86). */
int sed_dasm_head_fwd(const float* logits, const float* at_logit, const uint8_t* pad_mask, float temp, float* strong, float* weak,
                      float* at_out, int B, int T, int Q, int clamp_strong, hipStream_t stream);

/* ---- DASM / AudioSet-Strong TRAINING (recipes/audioset_strong/detect_any_sound/passt/train.py:66-120 `DASMTrainer.train`,
 * recipes/audioset_strong/base/passt_cnn/train.py:103-140 `Trainer.train`): the backward of everything above, fp32.
 *
 * General fp32 GEMM on v_mfma_f32_32x32x2_f32:  C[z][m][n] (+)= drop(act(sum_k A(z,m,k) B(z,n,k) + bias[n])) (+ R[z][m][n]) with
 * A(m,k) = transA ? A[k lda + m] : A[m lda + k], B(n,k) = transB ? B[k ldb + n] : B[n ldb + k] -- the three products of a Linear under
 * autograd (y = x W^T; dx = dy W: transB; dW += dy^T x: transA + transB + accumulate, split over the tokens by `ksplit` with fp32
 * atomics) and the einsum('bqc,bct->bqt') of detect_any_sound.py:378 with its gradients (batched).  Any M, N, K and leading dimensions
 * (operands that are not float4-addressable are read element-wise).  pre (nullable): the value before the activation (what the GELU
 * backward needs).  drop_p > 0: inverted dropout after the activation, before the residual, with the counter-based bits
 * keep(drop_seed, drop_site, (z M + m) N + n) that sed_gelu_bwd_f32 / sed_dropout_f32 re-evaluate (nn.TransformerDecoderLayer's
 * dropout1/2/3 and FFN dropout in train mode, at_adapter.py:37-45).  accumulate != 0: C += (no epilogue terms allowed). */
int sed_gemm_f32(const float* A, const float* B, const float* bias, const float* R, float* C, float* pre, int M, int N, int K, int lda,
                 int ldb, int ldc, int transA, int transB, int batch, int64_t strideA, int64_t strideB, int64_t strideC, int act,
                 int accumulate, int ksplit, float drop_p, int64_t drop_seed, int drop_site, hipStream_t stream);
/* sed_xattn_f32_fwd for a pass whose backward follows: also writes lse [B, H, Nq] (log2 of sum_j 2^(log2(e) s_ij)) and applies
 * attention-probability dropout (torch.nn.MultiheadAttention(dropout=p).train(): softmax -> dropout -> . v) with the bits
 * keep(seed, site, ((b H + h) Nq + i) Nk + j). */
int sed_xattn_f32_fwd_train(const float* Q, const float* K, const float* V, float* O, const uint8_t* mask, float* lse, int B, int H, int Nq,
                            int Nk, int head_dim, int ldq, int ldk, int ldv, int ldo, int64_t q_batch_stride, float drop_p, int64_t drop_seed,
                            int drop_site, hipStream_t stream);
/* its backward: dQ [B, Nq, lddq], dK / dV [B, Nk, lddk / lddv] at column h * head_dim (overwritten, packed projections are addressed in
 * place through the leading dimensions); O, dO [B, Nq, ldo]; Dq [B, H, Nq] scratch (dO_i . O_i). */
int sed_xattn_f32_bwd(const float* Q, const float* K, const float* V, const float* O, const float* dO, const float* lse, float* Dq, float* dQ,
                      float* dK, float* dV, const uint8_t* mask, int B, int H, int Nq, int Nk, int head_dim, int ldq, int ldk, int ldv, int ldo,
                      int lddq, int lddk, int lddv, int64_t q_batch_stride, float drop_p, int64_t drop_seed, int drop_site,
                      hipStream_t stream);
/* backward of sed_dasm_head_fwd: d strong [B,Q,T], d weak [B,Q], d at_out [B,Q] (each nullable) -> d logits [B,T,Q], d at_logit [B,Q].
 * at_logit == dat_logit == NULL: the closed-set head strong = sigmoid(logit / temp) for any class count (passt_cnn.py:74-86 with the 407
 * AudioSet-Strong classes).  scratch: 3 B Q floats. */
int sed_dasm_head_bwd(const float* logits, const float* at_logit, const uint8_t* pad_mask, float temp, const float* strong,
                      const float* dstrong, const float* dweak, const float* dat_out, float* dlogits, float* dat_logit, float* scratch, int B,
                      int T, int Q, int clamp_strong, hipStream_t stream);
/* out = dy o keep / (1 - p) o gelu'(pre) over n elements (keep bits of element index i: see sed_gemm_f32) */
int sed_gelu_bwd_f32(const float* dy, const float* pre, float* out, int64_t n, float drop_p, int64_t drop_seed, int drop_site,
                     hipStream_t stream);
/* out = x o keep / (1 - p) (nullable) and / or the keep bits themselves as bytes (nullable; test aid for the CPU oracle) */
int sed_dropout_f32(const float* x, float* out, uint8_t* mask_u8, int64_t n, float drop_p, int64_t drop_seed, int drop_site,
                    hipStream_t stream);
/* out = drop(act(x)) (+ res) over n elements; act 1 = GELU; dropout bits of element index i (see sed_gemm_f32) */
int sed_act_drop_res_f32(const float* x, const float* res, float* out, int64_t n, int act, float drop_p, int64_t drop_seed, int drop_site,
                         hipStream_t stream);
/* out[c] += sum_r x[r ld + c] */
int sed_colsum_f32(const float* x, float* out, int rows, int cols, int64_t ld, hipStream_t stream);
/* supervised losses of src/functional/loss/__init__.py (loss_function_factory), mean over n elements, loss[0] += value (caller zeroes),
 * grad (nullable) = d loss / d pred.  kind 0: -[(1-p)^gamma_pos t max(log p, -100) + pm^gamma_neg (1-t) max(log(1-pm), -100)],
 * pm = max(p - margin, 0): BCELoss (0, 0, 0), AsymmetricalFocalLoss(gamma, zeta), AslLoss(rp, rn, margin); kind 1: MSELoss. */
int sed_sup_loss(const float* pred, const float* target, float* loss, float* grad, int64_t n, int kind, float gamma_pos, float gamma_neg,
                 float margin, hipStream_t stream);

#ifdef __cplusplus
}
#endif
