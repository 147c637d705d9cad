"""Sweep of the C-ABI copy probes (fft_amd.copy_probe) at the headline shape: segment width x workgroups per CU x mode."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fft_amd import copy_probe, spectral_mix
dev = torch.device("cuda:0")
B, N, D = 256, 4096, 768
src = torch.randn(B, N, D, device=dev)
dst = torch.empty_like(src)
gate = torch.randn(B, 4, N // 2 + 1, dtype=torch.complex64, device=dev) * 0.3
for _ in range(60):
    spectral_mix(src, gate, None, N, out=dst)
torch.cuda.synchronize()
byt = 2.0 * src.numel() * 4
for seg in (0, 32, 64, 128, 256, 512):
    for mode in ("copy", "load", "store"):
        row = []
        for per_cu in (1, 2, 3, 4, 6, 8):
            ms = copy_probe(src, dst, seg, mode=mode, wgs_per_cu=per_cu, warmup=3, iters=10)
            row.append(f"{per_cu}/CU {ms:6.3f} ms {(byt if mode == 'copy' else byt / 2) / ms / 1e6:6.0f} GB/s")
        print(f"seg {seg:4d} {mode:5s}: " + " | ".join(row), flush=True)
