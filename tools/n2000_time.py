import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fft_amd import time_kernel, describe
dev = "cuda:0"
for n in (2000, 4096, 3000, 1024):
    B = 256 if n >= 3000 or n == 1024 else 384
    V = torch.randn(B, n, 768, device=dev).to(torch.bfloat16)
    g = torch.randn(B, 4, n // 2 + 1, dtype=torch.complex64, device=dev) * 0.3
    out = torch.empty_like(V)
    t = sorted(time_kernel(V, g, None, n, out=out, warmup=30 if i == 0 else 5, iters=10) for i in range(5))[2]
    print(f"n_fft {n} bf16 -> bf16 (B={B}): {t:.4f} ms  [{describe(V, g, None, n, out=out).split(' tiles')[0]}]", flush=True)
