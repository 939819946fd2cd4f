"""Thin tensor-level wrappers over the C ABI (include/sed_hip.h).  torch is used for device memory and the
current HIP stream only; every op here launches hand-written gfx950 kernels and raises if it cannot."""
import math
import os

import torch

from ._lib import lib

BF16 = torch.bfloat16
F16 = torch.float16
F32 = torch.float32


def kind(t):
    """operand-kind code of the C ABI: 0 bf16, 1 f32, 2 f16."""
    return {BF16: 0, F32: 1, F16: 2}[t.dtype]


def is_f16(t):
    return 1 if t.dtype == F16 else 0

EPI_F32, EPI_F32_RESID, EPI_BF16, EPI_GELU, EPI_DGELU, EPI_ATOMIC, EPI_QKV, EPI_F32_BF16, EPI_GELU32 = range(9)


_PIN_RINGS = {}
_PIN_DEPTH = 16


def h2d(x, dtype, dev):
    """Small host array -> device tensor WITHOUT a host/stream synchronisation: staged in pinned memory and copied with
    non_blocking=True (torch.tensor(list, device=...) / .to(device) from pageable memory block the host until the stream has
    drained, which serialises the Python schedule against the GPU at every call).  The pinned staging buffers come from a small ring
    per (shape, dtype): allocating pinned memory per call (`Tensor.pin_memory()`) costs milliseconds -- 19 ms per step for the
    per-step mel filterbank alone; a ring slot (16 per key) is reused only after the copy that last read it has completed, which bounds how far the
    host can run ahead of the GPU to a few steps."""
    import numpy as np
    t = x if isinstance(x, torch.Tensor) else torch.as_tensor(np.asarray(x))
    t = t.to(dtype).contiguous()
    if t.device.type != "cpu":
        return t.to(dev)
    key = (tuple(t.shape), dtype)
    ring = _PIN_RINGS.get(key)
    if ring is None:
        # (one pinned allocation for the whole ring: a slot allocated on first use costs ~1 ms each for the first 16 calls of a shape)
        block = torch.empty((_PIN_DEPTH,) + tuple(t.shape), dtype=dtype).pin_memory()
        ring = _PIN_RINGS[key] = dict(bufs=[block[j] for j in range(_PIN_DEPTH)], evs=[None] * _PIN_DEPTH, i=0)
    i = ring["i"]
    if ring["evs"][i] is None:
        ring["evs"][i] = torch.cuda.Event()
    else:
        ring["evs"][i].synchronize()   # only waits when the host is a full ring (several steps) ahead of the GPU
    buf = ring["bufs"][i]
    buf.copy_(t)
    out = buf.to(dev, non_blocking=True)
    ring["evs"][i].record()
    ring["i"] = (i + 1) % _PIN_DEPTH
    return out


class UploadBlock:
    """Several small host arrays -> ONE pinned staging block -> ONE asynchronous H2D copy; the arrays come back as typed views of the device
    copy.  A train step used to issue ~25 separate small uploads (shift / permutation / mixing tables, warp and filter tables, the mel
    filterbank, window descriptors), each a pinned copy + a blit kernel; everything the step's preprocessing draws on the host now travels
    in one (trainer.MatSedTrainer.preprocess)."""

    def __init__(self, dev):
        self.dev, self.parts, self.size = dev, [], 0

    def add(self, arr, dtype):
        """Stage `arr` (array-like) as `dtype` (torch.int32 / torch.float32); returns a handle for `view` after `commit`."""
        import numpy as np
        t = arr if isinstance(arr, torch.Tensor) else torch.as_tensor(np.asarray(arr))
        t = t.to(dtype).contiguous()
        off = (self.size + 15) // 16 * 16
        self.parts.append((off, t))
        self.size = off + t.numel() * t.element_size()
        return len(self.parts) - 1

    def commit(self):
        cap = max(4096, 1 << (self.size - 1).bit_length())        # few distinct capacities -> few pinned rings
        host = torch.zeros(cap, dtype=torch.uint8)
        for off, t in self.parts:
            host[off:off + t.numel() * t.element_size()] = t.reshape(-1).view(torch.uint8)
        self.block = h2d(host, torch.uint8, self.dev)
        return self

    def view(self, handle):
        off, t = self.parts[handle]
        return self.block[off:off + t.numel() * t.element_size()].view(t.dtype).view(t.shape)


def _ptr(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("transformer4sed_amd ops need tensors on an MI355X (HIP) device; there is no CPU path")
    if not t.is_contiguous():
        raise RuntimeError("non-contiguous tensor passed to a HIP op")
    return t.data_ptr()


class KernelTimer:
    """Brackets selected C-ABI launches with HIP events on the launch stream (used by bench.py for the roofline line)."""

    def __init__(self, names):
        self.names = set(names)
        self.records = []  # (name, start_event, end_event, algorithmic flops, algorithmic bytes, issued flops, shape key)
        self.pool = []     # recycled events: creating ~600 HIP events inside a timed step costs tens of ms of host time

    def event(self):
        return self.pool.pop() if self.pool else torch.cuda.Event(enable_timing=True)

    def recycle(self):
        """Drop the records, keep their events for the next instrumented step."""
        for _, e0, e1, *_ in self.records:
            self.pool += [e0, e1]
        self.records = []

    def summarize(self):
        out = {}
        for name, e0, e1, fl, by, fi, _ in self.records:
            d = out.setdefault(name, dict(launches=0, ms=0.0, flops=0.0, bytes=0.0, flops_issued=0.0))
            d["launches"] += 1
            d["ms"] += e0.elapsed_time(e1)
            d["flops"] += fl
            d["flops_issued"] += fi
            d["bytes"] += by
        return out


TIMER = None
ALG_K_SCALE = 1.0     # set by `split_precision()` around a GEMM whose K is the 3x concatenated split-precision reduction dim


class split_precision:
    """Marks the GEMM launches inside the block as split-precision (operands [hi|lo|hi] x [hi|hi|lo]): the kernel timer credits them
    with their logical 2 M N K, a third of the MFMA FLOPs they issue."""

    def __enter__(self):
        global ALG_K_SCALE
        ALG_K_SCALE = 1.0 / 3.0

    def __exit__(self, *exc):
        global ALG_K_SCALE
        ALG_K_SCALE = 1.0


class plain_precision:
    """Inside a split_precision() block: the GEMM launches in THIS block run on plain operands again (logical K = issued K)."""

    def __enter__(self):
        global ALG_K_SCALE
        self.prev, ALG_K_SCALE = ALG_K_SCALE, 1.0

    def __exit__(self, *exc):
        global ALG_K_SCALE
        ALG_K_SCALE = self.prev


def _flops_of(name, args):
    if name in ("sed_gemm_nt", "sed_gemm_nt_gb", "sed_gemm_nt_gb_e4m3", "sed_gemm_nt_w2", "sed_gemm_nt_w2f8", "sed_gemm_nt_lnp", "sed_gemm_nt_lnp8", "sed_gemm_nt_lnc", "sed_gemm_nt_lnc8"):       # (w2: the logical 2 M N K, half of the MFMA FLOPs issued)
        return 2.0 * args[2] * args[3] * args[4]
    if name in ("sed_gemm_qkv_lnc", "sed_gemm_qkv_lnc8"):
        return 2.0 * args[5] * args[6] * (3 * args[7] * 64)
    if name in ("sed_gemm_qkv", "sed_gemm_qkv_gb", "sed_gemm_qkv_w2", "sed_gemm_qkv_w2f8", "sed_gemm_qkv_w2s"):
        return 2.0 * args[3] * args[4] * (3 * args[5] * 64)
    if name == "sed_gemm_dw_tn":
        return 2.0 * args[3] * args[4] * args[5]
    if name == "sed_gemm_f32_nt":                # (A, B, bias, R, C, M, N, K, lda, ldb, ldc, batch, ...): fp32-input MFMA, its own roofline
        return 2.0 * args[5] * args[6] * args[7] * args[11]
    if name == "sed_gemm_f32":                   # (A, B, bias, R, C, pre, M, N, K, lda, ldb, ldc, transA, transB, batch, ...)
        return 2.0 * args[6] * args[7] * args[8] * args[14]
    if name in ("sed_xattn_f32_fwd", "sed_xattn_f32_fwd_train", "sed_xattn_f32_bwd"):
        # (Q, K, V, O, mask, [lse,] B, H, Nq, Nk, head_dim, ...): 2 products forward, 5 backward (S, dP, dQ, dK, dV) -- the backward's two
        # launches recompute S and dP once more each: 7 issued, 5 algorithmic
        o = {"sed_xattn_f32_fwd": 5, "sed_xattn_f32_fwd_train": 6, "sed_xattn_f32_bwd": 11}[name]
        B, H, Nq, Nk, dh = args[o:o + 5]
        return (10.0 if name.endswith("bwd") else 4.0) * B * H * Nq * Nk * dh
    return 0.0


def _shape_of(name, args):
    """(M, N, K, epilogue / output count) of one GEMM launch, for per-shape tables (tools/gemm_shapes.py)."""
    if name in ("sed_gemm_nt", "sed_gemm_nt_gb"):
        return (args[2], args[3], args[4], "epi%d" % args[7] + ("gb" if name.endswith("_gb") else ""))
    if name in ("sed_gemm_qkv_gb", "sed_gemm_qkv_w2"):
        return (args[3], 3 * args[5] * 64, args[4], "qkv3" + name[-2:])
    if name == "sed_gemm_qkv_w2f8":
        return (args[3], 3 * args[5] * 64, args[4], "qkv3w2f8")
    if name in ("sed_gemm_nt_w2", "sed_gemm_nt_w2f8"):
        return (args[2], args[3], args[4], "epi%d" % args[7] + name[12:])
    if name in ("sed_gemm_nt_lnp", "sed_gemm_nt_lnp8", "sed_gemm_nt_lnc", "sed_gemm_nt_lnc8"):
        return (args[2], args[3], args[4], "epi3lnc" if "lnc" in name else "epi1lnp")
    if name in ("sed_gemm_qkv_lnc", "sed_gemm_qkv_lnc8"):
        return (args[5], 3 * args[7] * 64, args[6], "qkv3lnc")
    if name in ("sed_gemm_qkv", "sed_gemm_qkv_w2s"):
        return (args[3], 3 * args[5] * 64, args[4], "qkv%d" % sum(a is not None for a in args[8:16]) + ("w2s" if name.endswith("w2s") else ""))
    if name == "sed_gemm_dw_tn":
        return (args[4], args[5], args[3], "tn")
    return None


def _bytes_of(name, args):
    """Algorithmic HBM bytes of one GEMM launch: operands once + every output / side input once."""
    if name in ("sed_gemm_qkv_gb", "sed_gemm_qkv_w2", "sed_gemm_qkv_w2f8"):
        M, K, D = args[3], args[4], args[5] * 64
        if name.endswith("f8"):
            return 3.0 * K * (M + 3 * D) + 2.0 * M * D * 3
        return 2.0 * K * (M + 3 * D * (2 if name.endswith("w2") else 1)) + 2.0 * M * D * 3
    if name in ("sed_gemm_nt_lnp", "sed_gemm_nt_lnp8"):
        # operands + the residual read (fp32 4 B / f16 planes 4 B / f16 + byte planes 3 B) + fp32 and f16 image (6 B) or the two planes (4 B / 3 B)
        M, N, K = args[2], args[3], args[4]
        pl = 3.0 if name.endswith("8") else 4.0
        return 2.0 * K * (M + N) + ((pl if args[10] is not None else 4.0) + (pl if args[13] is not None else 6.0)) * M * N
    if name in ("sed_gemm_nt_lnc", "sed_gemm_nt_lnc8"):
        M, N, K = args[2], args[3], args[4]
        return 2.0 * K * (M + N) + 2.0 * M * N
    if name in ("sed_gemm_qkv_lnc", "sed_gemm_qkv_lnc8"):
        M, K, D = args[5], args[6], args[7] * 64
        return 2.0 * K * (M + 3 * D) + 2.0 * M * D * 3
    if name in ("sed_gemm_nt_w2", "sed_gemm_nt_w2f8"):
        M, N, K, epi = args[2], args[3], args[4], args[7]
        out = {1: 8, 3: 2 * ((args[11] is not None) + (args[12] is not None))}.get(epi, 4)
        return (3.0 * K * (M + N) if name.endswith("f8") else 2.0 * K * (M + 2 * N)) + float(out) * M * N
    if name in ("sed_gemm_nt", "sed_gemm_nt_gb"):
        M, N, K, epi = args[2], args[3], args[4], args[7]
        out = {0: 4, 1: 8, 2: 2, 3: 2 * ((args[11] is not None) + (args[12] is not None)), 4: 4, 5: 8, 7: 6, 8: 6}.get(epi, 4)
        return 2.0 * K * (M + N) + float(out) * M * N
    if name in ("sed_gemm_qkv", "sed_gemm_qkv_w2s"):
        M, K, D = args[3], args[4], args[5] * 64
        nout = sum(a is not None for a in args[8:16])
        return 2.0 * K * (M + 3 * D * (2 if name.endswith("w2s") else 1)) + 2.0 * M * D * nout
    if name == "sed_gemm_dw_tn":
        T, M, N = args[3], args[4], args[5]
        return 2.0 * T * (M + N) + 8.0 * M * N
    return 0.0


def _hbm_bytes_of(name, args):
    """Algorithmic HBM bytes of one launch of an HBM-bound kernel (operands once, outputs once)."""
    if name == "sed_logmel_fwd":                 # (wav, out, tmp, window, twiddle, melw, mel_range, B, L, T, do_log)
        B, L, T = args[7], args[8], args[9]
        return 4.0 * B * (L + 128 * T)
    if name == "sed_adamw_ema":                  # (p, g, m, v, ema, n, ..., do_adam)
        n, ema, adam = args[5], args[4] is not None, bool(args[13])
        return float(n) * ((28 + (8 if ema else 0)) if adam else 12)
    if name == "sed_layernorm_fwd":              # (x, gamma, beta, eps, scale, y16, y32, mean, rstd, M, D, f16)
        M, D = args[9], args[10]
        return float(M) * (D * (4 + (2 if args[5] is not None else 0) + (4 if args[6] is not None else 0)) + (8 if args[7] is not None else 0))
    if name == "sed_layernorm_bwd":              # (dy, x, mean, rstd, gamma, scale, dx, accumulate, dgamma, dbeta, M, D)
        M, D = args[10], args[11]
        return float(M) * (D * (12 + (4 if args[7] else 0)) + 8)
    if name == "sed_layernorm_bwd_x16":          # (..., dbeta, dx16, M, D): + the bf16 image
        M, D = args[11], args[12]
        return float(M) * (D * (14 + (4 if args[7] else 0)) + 8)
    return 0.0


HBM_KERNELS = ("sed_logmel_fwd", "sed_adamw_ema", "sed_layernorm_fwd", "sed_layernorm_bwd", "sed_layernorm_bwd_x16")

_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)   # the current stream's handle without building a Stream object


def _stream_of(dev_index):
    if _raw_stream is not None:
        return _raw_stream(dev_index if dev_index is not None else torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def call(name, *args):
    """Launch one C-ABI entry point on torch's current HIP stream of the tensors' device.  This runs ~1100 times per train step:
    argument conversion is a single pass and the stream handle comes from the raw query (the Stream-object path was 9 us a call)."""
    conv, dev = [], None
    for a in args:
        if isinstance(a, torch.Tensor):
            if not a.is_cuda:
                raise RuntimeError("transformer4sed_amd ops need tensors on an MI355X (HIP) device; there is no CPU path")
            if not a.is_contiguous():
                raise RuntimeError("non-contiguous tensor passed to a HIP op")
            conv.append(a.data_ptr())
            dev = a.device.index
        else:
            conv.append(a)
    if TIMER is not None and name in TIMER.names:
        e0, e1 = TIMER.event(), TIMER.event()
        e0.record()
        lib().call(name, *conv, _stream_of(dev))
        e1.record()
        fi = _flops_of(name, args)
        by = _hbm_bytes_of(name, args) if name in HBM_KERNELS else _bytes_of(name, args)
        two = name.endswith(("_w2", "_w2s", "_w2f8"))
        issued = fi * (2.0 if two else 1.0)      # two-term weights: the A panel is multiplied twice (w2f8: the second time as e4m3 on the fp8 path)
        # (a two-term launch is given its LOGICAL K: no split-precision scaling even inside a split_precision() block)
        TIMER.records.append((name, e0, e1, fi * (1.0 if two else ALG_K_SCALE), by, issued, _shape_of(name, args)))
        return
    lib().call(name, *conv, _stream_of(dev))


def pad64(n):
    return (n + 63) // 64 * 64


def gemm_nt(A, B, epi, M=None, bias=None, res=None, outF=None, outH=None, outH2=None, aux=None, alpha=1.0, ksplit=1,
            lda=None, ldb=None, ldc=None, gbias=None, gb_rows=0, two_term=False, K=None):
    """C[M,N] = A[M,K] . B[N,K]^T (bf16 operands) with fused epilogue; see include/sed_hip.h.  `gbias` [M / gb_rows, N]: row-group bias.
    `two_term`: B is the [N, 2K] image [f16(W) | f16(W - f16(W))] over an A of K columns."""
    M = A.shape[0] if M is None else M
    N, K = B.shape[0], (B.shape[1] if K is None else K)      # (`K` < B.shape[1] with ldb = B.shape[1]: the first K columns of B's rows)
    if two_term:
        if A.dtype != F16 or B.dtype != F16 or K != 2 * A.shape[1]:
            raise RuntimeError("gemm_nt: two-term weights take an f16 A [M, K] and an f16 B [N, 2K]")
        call("sed_gemm_nt_w2", A, B, M, N, K // 2, lda or A.shape[1], ldb or K, epi, bias, res, outF, outH, outH2, ldc or N, 1)
        return
    if gbias is not None:
        if A.dtype != B.dtype:
            raise RuntimeError("gemm_nt: operand types differ")
        call("sed_gemm_nt_gb", A, B, M, N, K, lda or A.shape[1], ldb or K, epi, bias, res, outF, outH, outH2, aux, ldc or N, float(alpha),
             is_f16(A), gbias, int(gb_rows))
        return
    # the saved pre-activation of a GELU GEMM (epi 3 / 8) is consumed only by the bf16 backward: it may be bf16 in an f16 GEMM
    pre_bf16 = epi in (EPI_GELU, EPI_GELU32) and outH is not None and outH.dtype == BF16 and A.dtype == F16
    if A.dtype != B.dtype or any(t is not None and t.dtype != A.dtype for t in ((None if pre_bf16 else outH), outH2, aux)):
        raise RuntimeError("gemm_nt: all 16-bit operands/outputs of one GEMM must share one type (f16 or bf16)")
    call("sed_gemm_nt", A, B, M, N, K, lda or A.shape[1], ldb or B.shape[1], epi, bias, res, outF, outH, outH2, aux, ldc or N,
         float(alpha), ksplit, 3 if pre_bf16 else is_f16(A))


def gemm_nt_cols(A, B, epi, ncols, bias=None, res=None, outF=None, outH=None, ldc=None):
    """gemm_nt whose result has only `ncols` (< B.shape[0], the padded operand width) columns in memory; see sed_gemm_nt_cols."""
    N, K = B.shape
    call("sed_gemm_nt_cols", A, B, A.shape[0], N, K, A.shape[1], K, epi, bias, res, outF, outH, None, None, ldc or ncols, 1.0,
         is_f16(A), ncols)


def dw_ksplit(n_out, k_in, mpad):
    tiles = ((n_out + 127) // 128) * (k_in // 128)
    ks = max(1, min((640 + tiles - 1) // tiles, mpad // 256))
    return ks


def gemm_dw(dYt, Xt, dW, n_out=None):
    """dW[n_out, k_in] += dY^T . X  given the transposed bf16 operands dYt [n_out, Mpad], Xt [k_in, Mpad]."""
    n_out = dYt.shape[0] if n_out is None else n_out
    k_in, mpad = Xt.shape
    gemm_nt(dYt, Xt, EPI_ATOMIC, M=n_out, outF=dW, ksplit=dw_ksplit(n_out, k_in, mpad), ldc=k_in)


def gemm_dw_tn(dY, X, dW, tokens=None, ldc=None, dbias=None, k_in=None):
    """dW[n_out, k_in] (fp32) += dY[tokens, n_out]^T . X[tokens, k_in]; both operands in their row-major [token][feature] layout
    (dY bf16, X bf16 or f16).  Needs both feature counts % 64 == 0 (see `dw_tn_ok`); any token count.  `k_in` < X.shape[1]: only the
    first k_in columns of X's rows are the operand (the `hi` third of a split-precision image [tokens, 3 k_in])."""
    T = dY.shape[0] if tokens is None else tokens
    ws = _dw_workspace(dY.device)
    k = X.shape[1] if k_in is None else k_in
    call("sed_gemm_dw_tn", dY, X, is_f16(X), T, dY.shape[1], k, dY.shape[1], X.shape[1], dW, ldc or k, dbias, ws, ws.numel() * 4)


_DW_WS = {}


def _dw_workspace(dev):
    """Split-K partial-sum workspace of the TN weight-gradient GEMM (96 MiB per device, allocated once)."""
    key = str(dev)
    if key not in _DW_WS:
        _DW_WS[key] = torch.empty(24 << 20, dtype=F32, device=dev)
    return _DW_WS[key]


def dw_tn_ok(tokens, n_out, k_in):
    return n_out % 64 == 0 and k_in % 64 == 0


def transpose_bf16(x, rows, cols, out_t, out_s=None, colsum=None, ld=None):
    """x [rows, cols] (f32 / bf16 / f16) -> out_t [cols, Rpad] (nullable); optional straight 16-bit copy / fp32 column
    sums (+=).  Output kinds follow the output tensors' dtypes."""
    call("sed_transpose_to_bf16", x, kind(x), rows, cols, ld or cols, out_t, out_t.shape[1] if out_t is not None else pad64(rows),
         kind(out_t) if out_t is not None else 0, out_s, kind(out_s) if out_s is not None else 0, colsum)


def to_bf16_(t):
    """In-place f16 -> bf16 conversion of a saved forward operand; returns the bf16 view (no-op for bf16 tensors)."""
    if t is None or t.dtype == BF16:
        return t
    call("sed_f16_to_bf16_inplace", t, t.numel())
    return t.view(BF16)


def split3(x32, rows, cols, weight=False):
    """fp32 [rows, cols] -> f16 [rows, 3 cols] split-precision operand image: [hi|lo|hi] (activations) or [hi|hi|lo]
    (weights); a GEMM over the concatenated K accumulates hi*hi + lo*hi + hi*lo in fp32."""
    out = torch.empty(rows, 3 * cols, dtype=F16, device=x32.device)
    call("sed_split3_f16", x32, out, rows, cols, 1 if weight else 0)
    return out


def two_term_weight(w32):
    """fp32 weight [N, K] -> f16 [N, 2K] = [f16(W) | f16(W - f16(W))] for gemm_nt(two_term=True) / sed_gemm_qkv_w2."""
    N, K = w32.shape
    out = torch.empty(N, 2 * K, dtype=F16, device=w32.device)
    call("sed_split3_f16", w32.contiguous(), out, N, K, 2)
    return out


def two_term_weight_f8(w32):
    """fp32 weight [N, K] -> (uint8 image [N, 3K] with rows [f16(W) | e4m3(2^s (W - f16(W)))], s) for gemm_nt_w2f8 / sed_gemm_qkv_w2f8.
    s (even) puts the largest rounding residual just below the e4m3 maximum of 448."""
    N, K = w32.shape
    w32 = w32.contiguous()
    m = float((w32 - w32.to(F16).float()).abs().max())
    s = 0 if m == 0.0 else int(math.floor(math.log2(448.0 / m)))
    s = max(-60, min(60, s)) & ~1
    out = torch.empty(N, 3 * K, dtype=torch.uint8, device=w32.device)
    call("sed_weight_two_term_f8", w32, out, N, K, s)
    return out, s


def fp8_rows(M, K, device, dtype=F16):
    """[M, 3K / 2] 16-bit buffer whose rows are [K f16 | K e4m3]; `[:, :K]` is the activation, the rest its e4m3 image (sed_fp8_tail)."""
    return torch.empty(M, K + K // 2, dtype=dtype, device=device)


def fp8_tail(x, K):
    """fills the e4m3 half of rows [K f16 | K e4m3] (x: [M, 3K / 2] f16) from their f16 half."""
    call("sed_fp8_tail", x, x.shape[0], K, x.shape[1])


def gemm_nt_w2f8(A, Bimg, s, epi, K, bias=None, res=None, outF=None, outH=None, outH2=None, ldc=None, out_e4m3=False):
    """gemm_nt(two_term=True) with the lo product on the fp8 matrix path: A [M, 3K / 2] f16 rows [K f16 | K e4m3] (fp8_rows / fp8_tail),
    (Bimg, s) = two_term_weight_f8(W).  out_e4m3 (epi GELU): outH2 is an fp8_rows buffer of N columns and gets both halves."""
    M, N = A.shape[0], Bimg.shape[0]
    if A.dtype != F16 or Bimg.dtype != torch.uint8 or Bimg.shape[1] != 3 * K or A.shape[1] < K + K // 2:
        raise RuntimeError("gemm_nt_w2f8: A is f16 [M, 3K / 2], B the uint8 image [N, 3K] of two_term_weight_f8")
    call("sed_gemm_nt_w2f8", A, Bimg, M, N, K, A.shape[1], 3 * K // 2, epi, bias, res, outF, outH, outH2, ldc or (N + N // 2 if out_e4m3 else N),
         1 if out_e4m3 else 0, s)


def o_kind(t):
    """storage code of an attention output for the backward pre-pass: 0 bf16, 1 f16, 2 f32."""
    return {BF16: 0, F16: 1, F32: 2}[t.dtype]
