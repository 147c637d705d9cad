"""The six persistent mixed-radix lengths at (B, n, 768), B scaled to ~590 M elements: forward fp32, full and 7 rows short.  A/B between
libraries through SPECTRE_HIP_LIB (see tools/mixed_engine_ab.py)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fft_amd import time_kernel
dev = "cuda:0"; D, G = 768, 4
for n in [2400, 2560, 3000, 3072, 3600, 3840]:
    B = (256 * 3000) // n
    torch.manual_seed(n)
    V = torch.randn(B, n, D, device=dev); g = torch.randn(B, G, n // 2 + 1, dtype=torch.complex64, device=dev) * 0.3
    out = torch.empty(B, n, D, device=dev)
    t = min(time_kernel(V, g, None, n, out=out, warmup=5, iters=10) for _ in range(4))
    Vs = V[:, : n - 7].contiguous(); outs = torch.empty(B, n - 7, D, device=dev)
    ts = min(time_kernel(Vs, g, None, n, out=outs, warmup=5, iters=10) for _ in range(4))
    by = 2 * B * n * D * 4 + B * G * (n // 2 + 1) * 8
    print(f"n={n}: {t:.4f} {ts:.4f}   frac {by / t / 1e6 / 8000:.3f}", flush=True)
