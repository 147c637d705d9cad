"""What does the measured tile order decide, and is it right?  (B, N, D) per argv, n_fft = N."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fft_amd import spectral_mix, time_kernel, describe
B, N, D = (int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "256,3000,768").split(","))
dev = "cuda:0"
V = torch.randn(B, N, D, device=dev); out = torch.empty_like(V)
gate = torch.randn(B, 4, N // 2 + 1, dtype=torch.complex64, device=dev) * 0.3
ms = time_kernel(V, gate, None, N, out=out, warmup=60, iters=20)
print(f"{ms:.4f} ms  {describe(V, gate, None, N, out=out)}")
