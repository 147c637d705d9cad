"""A few spectre_mix_bwd launches at one shape (profiling target):  python tools/run_bwd.py [B,N,D] [dv|dgate|both] [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fft_amd import spectral_mix_backward
B, N, D = (int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "256,4096,768").split(","))
what = sys.argv[2] if len(sys.argv) > 2 else "both"
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = "cuda:0"
torch.manual_seed(0)
V = torch.randn(B, N, D, device=dev)
g = torch.randn(B, 4, N // 2 + 1, dtype=torch.complex64, device=dev) * 0.3
do = torch.randn(B, N, D, device=dev)
for _ in range(iters):
    spectral_mix_backward(V, g, do, N, need_dv=what in ("dv", "both"), need_dgate=what in ("dgate", "both"))
torch.cuda.synchronize()
