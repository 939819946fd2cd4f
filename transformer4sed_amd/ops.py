"""Thin tensor-level wrappers over the C ABI (include/sed_hip.h).  torch is used for device memory and the
current HIP stream only; every op here launches hand-written gfx950 kernels and raises if it cannot."""
import torch

from ._lib import lib

BF16 = torch.bfloat16
F32 = torch.float32

EPI_F32, EPI_F32_RESID, EPI_BF16, EPI_GELU, EPI_DGELU, EPI_ATOMIC, EPI_QKV, EPI_F32_BF16 = range(8)


def _ptr(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("transformer4sed_amd ops need tensors on an MI355X (HIP) device; there is no CPU path")
    if not t.is_contiguous():
        raise RuntimeError("non-contiguous tensor passed to a HIP op")
    return t.data_ptr()


def call(name, *args):
    conv = [(_ptr(a) if (a is None or isinstance(a, torch.Tensor)) else a) for a in args]
    lib().call(name, *conv, torch.cuda.current_stream().cuda_stream)


def pad64(n):
    return (n + 63) // 64 * 64


def gemm_nt(A, B, epi, M=None, bias=None, res=None, outF=None, outH=None, outH2=None, aux=None, alpha=1.0, ksplit=1,
            lda=None, ldb=None, ldc=None):
    """C[M,N] = A[M,K] . B[N,K]^T (bf16 operands) with fused epilogue; see include/sed_hip.h."""
    M = A.shape[0] if M is None else M
    N, K = B.shape[0], B.shape[1]
    call("sed_gemm_nt", A, B, M, N, K, lda or A.shape[1], ldb or K, epi, bias, res, outF, outH, outH2, aux, ldc or N,
         float(alpha), ksplit)


def dw_ksplit(n_out, k_in, mpad):
    tiles = ((n_out + 127) // 128) * (k_in // 128)
    ks = max(1, min((640 + tiles - 1) // tiles, mpad // 256))
    return ks


def gemm_dw(dYt, Xt, dW, n_out=None):
    """dW[n_out, k_in] += dY^T . X  given the transposed bf16 operands dYt [n_out, Mpad], Xt [k_in, Mpad]."""
    n_out = dYt.shape[0] if n_out is None else n_out
    k_in, mpad = Xt.shape
    gemm_nt(dYt, Xt, EPI_ATOMIC, M=n_out, outF=dW, ksplit=dw_ksplit(n_out, k_in, mpad), ldc=k_in)


def transpose_bf16(x, rows, cols, out_t, out_s=None, colsum=None, ld=None):
    """x [rows, cols] (f32 or bf16) -> out_t bf16 [cols, Rpad]; optional straight bf16 copy / fp32 column sums (+=)."""
    call("sed_transpose_to_bf16", x, 1 if x.dtype == F32 else 0, rows, cols, ld or cols, out_t, out_t.shape[1], out_s,
         colsum)
