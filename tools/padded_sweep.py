"""Padded sequences (N_in < n_fft) next to full ones for the power-of-two register-tile kernels (general modes use bounds-checked buffer
instructions since round 2)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fft_amd import describe, time_kernel
dev = "cuda:0"
B, D, G = 256, 768, 4
for n_fft in (4096, 2048, 1024, 512, 256):
    g = torch.randn(B, G, n_fft // 2 + 1, dtype=torch.complex64, device=dev) * 0.3
    mem = torch.randn(n_fft // 2 + 1, D, dtype=torch.complex64, device=dev) * 0.2
    for (N, m, dt) in [(n_fft, None, torch.float32), (n_fft - n_fft // 32, None, torch.float32), (n_fft, mem, torch.float32),
                       (n_fft, None, torch.bfloat16), (n_fft - n_fft // 32, None, torch.bfloat16)]:
        V = torch.randn(B, N, D, device=dev).to(dt)
        out = torch.empty(B, min(N, n_fft), D, device=dev, dtype=dt)
        ms = min(time_kernel(V, g, m, n_fft, out=out, warmup=2, iters=8) for _ in range(3))
        print(f"n_fft={n_fft} N_in={N} mem={m is not None} {str(dt)[6:]}: {ms:.3f} ms [{describe(V, g, m, n_fft)[:58]}]")
for n_fft in (3000, 2000, 1536, 768):
    g = torch.randn(B, G, n_fft // 2 + 1, dtype=torch.complex64, device=dev) * 0.3
    mem = torch.randn(n_fft // 2 + 1, D, dtype=torch.complex64, device=dev) * 0.2
    for (N, m) in [(n_fft, None), (n_fft - n_fft // 30, None), (n_fft, mem)]:
        V = torch.randn(B, N, D, device=dev)
        out = torch.empty(B, min(N, n_fft), D, device=dev)
        ms = min(time_kernel(V, g, m, n_fft, out=out, warmup=2, iters=8) for _ in range(3))
        print(f"n_fft={n_fft} N_in={N} mem={m is not None} float32: {ms:.3f} ms [{describe(V, g, m, n_fft)[:58]}]")
