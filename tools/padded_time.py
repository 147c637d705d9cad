"""Padded sequence (N_in < n_fft = 4096) and memory_fft timings next to the fast mode."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fft_amd import describe, time_kernel
dev = "cuda:0"
B, D, G, n_fft = 256, 768, 4, 4096
g = torch.randn(B, G, n_fft // 2 + 1, dtype=torch.complex64, device=dev) * 0.3
mem = torch.randn(n_fft // 2 + 1, D, dtype=torch.complex64, device=dev) * 0.2
for (N, m) in [(4096, None), (4000, None), (3000, None), (4096, mem), (4000, mem)]:
    V = torch.randn(B, N, D, device=dev)
    out = torch.empty(B, min(N, n_fft), D, device=dev)
    ms = min(time_kernel(V, g, m, n_fft, out=out, warmup=2, iters=8) for _ in range(3))
    print(f"(256,{N},768) n_fft=4096 mem={m is not None}: {ms:.3f} ms [{describe(V, g, m, n_fft)}]")
