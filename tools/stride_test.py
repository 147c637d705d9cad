import sys, torch
sys.path.insert(0, "/root/repo")
from fft_amd import time_kernel, describe
dev="cuda:0"; B,N,D,G=256,4096,768,4
g = torch.randn(B, G, N//2+1, dtype=torch.complex64, device=dev)*0.3
for pad in (0, 16, 32, 64, 128, 256):
    Vb = torch.randn(B, N, D+pad, device=dev); Ob = torch.empty(B, N, D+pad, device=dev)
    V = Vb[:, :, :D]; O = Ob[:, :, :D]
    ms = min(time_kernel(V, g, None, N, out=O, warmup=3, iters=10) for _ in range(3))
    print(f"row stride {D+pad} floats ({(D+pad)*4} B): {ms:.3f} ms [{describe(V, g, None, N)[:40]}]")
