"""Row-predicated 4096 kernel (SPECTRE_P64=0) with several tiles per workgroup (SPECTRE_TPW): does a persistent loop help the round-1 kernel?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fft_amd import describe, time_kernel
dev = "cuda:0"
B, N, D, G = 256, 4096, 768, 4
g = torch.randn(B, G, N // 2 + 1, dtype=torch.complex64, device=dev) * 0.3
for dt in (torch.float32, torch.bfloat16):
    V = torch.randn(B, N, D, device=dev).to(dt)
    out = torch.empty_like(V)
    ms = min(time_kernel(V, g, None, N, out=out, warmup=2, iters=8) for _ in range(3))
    print(f"TPW={os.environ.get('SPECTRE_TPW', '1')} {str(dt)[6:]}: {ms:.3f} ms [{describe(V, g, None, N)[:50]}]")
