import ctypes, os, subprocess, sys, torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "abl.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
                       os.path.join(here, "gemm_ablate.hip"), "-o", so])
lib = ctypes.CDLL(so)
lib.abl_launch.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 3 + [ctypes.c_int] * 3 + [ctypes.c_void_p]
names = {0: "full", 1: "no DMA", 3: "no DMA, no ds_read", 8: "no MFMA (DMA+ds_read)", 10: "DMA + barrier only", 14: "DMA only, no barrier"}
for (M, N, K) in ((8192, 8192, 8192), (38080, 3072, 768)):
    A = (torch.randn(M, K, device="cuda") * 0.5).half(); B = (torch.randn(N, K, device="cuda") * 0.05).half()
    C = torch.empty(M, N, dtype=torch.half, device="cuda")
    for abl, nm in names.items():
        st = torch.cuda.current_stream().cuda_stream
        lib.abl_launch(abl, A.data_ptr(), B.data_ptr(), C.data_ptr(), M, N, K, st); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            lib.abl_launch(abl, A.data_ptr(), B.data_ptr(), C.data_ptr(), M, N, K, st)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(f"M{M} N{N} K{K}  {nm:22s} {ms:7.3f} ms {2.0*M*N*K/ms/1e9:8.1f} TF/s", flush=True)
lib.abl2_launch.argtypes = [ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 3 + [ctypes.c_int] * 3 + [ctypes.c_void_p]
for (M, N, K) in ((8192, 8192, 8192), (38080, 3072, 768)):
    A = (torch.randn(M, K, device="cuda") * 0.5).half(); B = (torch.randn(N, K, device="cuda") * 0.05).half()
    C = torch.empty(M, N, dtype=torch.half, device="cuda")
    for abl, nst, nm in ((0, 2, "v2 full 2 stages"), (0, 3, "v2 full 3 stages"), (1, 3, "v2 no DMA"), (8, 3, "v2 no MFMA 3st"), (8, 2, "v2 no MFMA 2st")):
        st = torch.cuda.current_stream().cuda_stream
        lib.abl2_launch(abl, nst, A.data_ptr(), B.data_ptr(), C.data_ptr(), M, N, K, st); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            lib.abl2_launch(abl, nst, A.data_ptr(), B.data_ptr(), C.data_ptr(), M, N, K, st)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(f"M{M} N{N} K{K}  {nm:22s} {ms:7.3f} ms {2.0*M*N*K/ms/1e9:8.1f} TF/s", flush=True)
