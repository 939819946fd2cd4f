import torch, time
for (M,N,K) in ((38080,3072,768),(38080,768,3072),(38080,768,768),(211904,3072,768),(8192,8192,8192)):
    a=(torch.randn(M,K,device="cuda")*0.5).half(); w=(torch.randn(N,K,device="cuda")*0.05).half()
    for _ in range(3): c=torch.nn.functional.linear(a,w)
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): c=torch.nn.functional.linear(a,w)
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/10
    print(f"torch(hipBLASLt) f16 M={M} N={N} K={K} {ms:.3f} ms {2.0*M*N*K/ms/1e9:.1f} TFLOP/s")
