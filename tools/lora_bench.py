"""sed_lora_grad on the encoder's weight shapes (developer tool; needs a GPU)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformer4sed_amd.ops import call
dev = "cuda"
for n, k in ((768, 768), (2304, 768), (3072, 768), (768, 3072)):
    dW = torch.randn(n, k, device=dev); A = torch.randn(8, k, device=dev); B = torch.randn(n, 8, device=dev)
    dA = torch.zeros(8, k, device=dev); dB = torch.zeros(n, 8, device=dev)
    f = lambda: call("sed_lora_grad", dW, A, B, 2.0, dA, dB, n, k, 8)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    print(f"lora_grad [{n} x {k}]: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us")
