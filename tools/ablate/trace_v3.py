"""Phase timeline of the 256x256 GEMM kernel (developer tool; needs a GPU).

Builds gemm.hip with -DSED_GEMM_TRACE into tools/ablate/libgemm_trace.so, runs one launch per shape and prints,
per workgroup: prologue (launch -> first operand tile landed), main loop, epilogue issue, store drain; plus how the
workgroups of one CU follow each other."""
import ctypes, os, subprocess, sys
import numpy as np
import torch
here = os.path.dirname(os.path.abspath(__file__))
root = os.path.dirname(os.path.dirname(here))
so = os.path.join(here, "libgemm_trace.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-ffast-math",
                       "-fno-finite-math-only", "-DSED_GEMM_TRACE", "-I" + os.path.join(root, "include"),
                       os.path.join(root, "transformer4sed_amd", "csrc", "gemm.hip"), "-o", so])
lib = ctypes.CDLL(so)
V = ctypes.c_void_p
lib.sed_gemm_nt.argtypes = [V, V, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, V, V, V, V, V, V,
                            ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_int, V]
lib.sed_debug_set_gemm_trace.argtypes = [V]
lib.sed_debug_set_gemm_trace_wave.argtypes = [ctypes.c_int]
EPI_F32, EPI_F32_RESID, EPI_BF16, EPI_GELU = 0, 1, 2, 3
for name, M, N, K, epi in (("fc1 gelu", 38080, 3072, 768, EPI_GELU), ("plain f16", 38080, 3072, 768, EPI_BF16),
                           ("fc2 resid", 38080, 768, 3072, EPI_F32_RESID), ("square", 8192, 8192, 8192, EPI_BF16)):
    A = (torch.randn(M, K, device="cuda") * 0.5).half(); B = (torch.randn(N, K, device="cuda") * 0.05).half()
    bias = torch.randn(N, device="cuda")
    outF = torch.zeros(M, N, device="cuda") if epi in (EPI_F32, EPI_F32_RESID) else None
    outH = torch.empty(M, N, dtype=torch.half, device="cuda") if epi in (EPI_GELU, EPI_BF16) else None
    outH2 = torch.empty(M, N, dtype=torch.half, device="cuda") if epi == EPI_GELU else None
    nwg = ((M + 255) // 256) * (N // 256)
    tr = torch.zeros(nwg * 8, dtype=torch.int64, device="cuda")
    p = lambda t: None if t is None else t.data_ptr()
    st = torch.cuda.current_stream().cuda_stream
    def run():
        rc = lib.sed_gemm_nt(A.data_ptr(), B.data_ptr(), M, N, K, K, K, epi, bias.data_ptr(), p(outF) if epi == EPI_F32_RESID else None,
                             p(outF), p(outH), p(outH2), None, N, 1.0, 1, 1, st)
        assert rc == 0, rc
    lib.sed_debug_set_gemm_trace(None); run(); torch.cuda.synchronize()
    waits = []
    for w in range(8):   # one launch per reporting wave
        lib.sed_debug_set_gemm_trace_wave(w)
        lib.sed_debug_set_gemm_trace(tr.data_ptr()); run(); torch.cuda.synchronize()
        tt = tr.cpu().numpy().reshape(nwg, 8)
        waits.append((np.median(tt[:, 5]) / max(1, K // 64 - 1), np.median(tt[:, 6]) / max(1, K // 64 - 1)))
    lib.sed_debug_set_gemm_trace_wave(0)
    lib.sed_debug_set_gemm_trace(tr.data_ptr()); run(); torch.cuda.synchronize()
    t = tr.cpu().numpy().reshape(nwg, 8).astype(np.int64)
    t0 = t[:, 0].min()
    us = (t[:, :5] - t0) / 100.0  # 100 MHz
    pro, main, epi_t, drain = us[:, 1] - us[:, 0], us[:, 2] - us[:, 1], us[:, 3] - us[:, 2], us[:, 4] - us[:, 3]
    q = lambda x: "p10 %6.1f med %6.1f p90 %6.1f" % tuple(np.percentile(x, [10, 50, 90]))
    print(f"== {name} M={M} N={N} K={K} wgs={nwg} total {us[:, 4].max():.1f} us")
    print("  prologue ", q(pro)); print("  main loop", q(main), " per k-iter med %.2f" % (np.median(main) / (K // 64)))
    print("  epi issue", q(epi_t)); print("  drain    ", q(drain))
    nkk = K // 64 - 1
    cyc = (t[:, 2] - t[:, 1]) * 10.0   # ns of the main loop (100 MHz stamps)
    print("  per K tile, shader cycles (DMA wait / barrier wait) by wave: " + "  ".join("w%d %.0f/%.0f" % (w, a, b) for w, (a, b) in enumerate(waits))
          + "   (main loop %.0f ns per K tile)" % (np.median(cyc) / (K // 64)))
    hw = t[:, 7]
    cu = (hw >> 32) * 1000 + ((hw >> 8) & 0xF) + 16 * ((hw >> 13) & 0x7)   # xcc, cu_id, se_id
    gaps = []
    for c in np.unique(cu):
        idx = np.where(cu == c)[0]
        idx = idx[np.argsort(us[idx, 0])]
        gaps += list(us[idx[1:], 0] - us[idx[:-1], 4])
    print("  CUs seen", len(np.unique(cu)), " gap between a workgroup's end and its successor's start on the CU:", q(np.array(gaps)))
    first = us[:, 0] < 5
    print("  first-round wgs", first.sum(), " start spread p90 %.1f us" % np.percentile(us[first, 0], 90))
    # how many workgroups are inside their epilogue+drain at the same time
    ev = np.zeros(int(us[:, 4].max() * 2) + 2)
    for a, b in zip(us[:, 2], us[:, 4]):
        ev[int(a * 2):int(b * 2) + 1] += 1
    print("  workgroups in epilogue/drain: mean %.1f max %d (of 256 resident)" % (ev.mean(), ev.max()))
