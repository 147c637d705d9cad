"""Twelve allocations of the tensor size in one process: dense / 128-byte-segment load-only and store-only rates per buffer (C ABI
spectre_probe_copy) and copies between them.  Allocations fall into two classes (profiles/r03_placement_classes.log)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fft_amd import copy_probe
dev = torch.device("cuda:0")
B, N, D, K = 256, 4096, 768, 12
bufs = [torch.randn(B, N, D, device=dev) for _ in range(K)]
for i, b in enumerate(bufs): print(f"buffer {i} at {b.data_ptr():#x}")
def t(i, o, seg, mode="copy"): return min(copy_probe(bufs[i], bufs[o], seg, mode=mode, wgs_per_cu=w, warmup=8, iters=16) for w in (2, 4))
print("dense load-only per buffer:", " ".join(f"{t(i, (i+1)%K, 0, 'load'):.4f}" for i in range(K)))
print("dense store-only per buffer:", " ".join(f"{t((i+1)%K, i, 0, 'store'):.4f}" for i in range(K)))
print("pattern(128B) load-only per buffer:", " ".join(f"{t(i, (i+1)%K, 128, 'load'):.4f}" for i in range(K)))
print("pattern(128B) store-only per buffer:", " ".join(f"{t((i+1)%K, i, 128, 'store'):.4f}" for i in range(K)))
print("dense copy src 0 -> dst k:", " ".join(f"{t(0, o, 0):.4f}" for o in range(1, K)))
print("dense copy src k -> dst 0:", " ".join(f"{t(i, 0, 0):.4f}" for i in range(1, K)))
print("dense copy k -> k+1:", " ".join(f"{t(i, i+1, 0):.4f}" for i in range(K-1)))
print("dense copy k -> k+2:", " ".join(f"{t(i, i+2, 0):.4f}" for i in range(K-2)))
