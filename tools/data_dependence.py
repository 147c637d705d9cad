"""Does the CONTENT of V change the launch time (power / clocks)?  Same (V, out) pair throughout, the tile order pinned."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fft_amd import spectral_mix, time_kernel, describe
B, N, D = 256, 4096, 768
dev = "cuda:0"
V = torch.empty(B, N, D, device=dev); out = torch.empty_like(V)
gate = torch.randn(B, 4, N // 2 + 1, dtype=torch.complex64, device=dev) * 0.3
fills = {"randn": lambda: V.normal_(), "zeros": lambda: V.zero_(), "uniform(-1,1)": lambda: V.uniform_(-1, 1), "randn * 1e-3": lambda: V.normal_().mul_(1e-3),
         "one row pattern repeated": lambda: V.copy_(torch.randn(1, 1, D, device=dev).expand(B, N, D))}
print(describe(V, gate, None, N, out=out))
for rep in range(2):
    for name, f in fills.items():
        f(); torch.cuda.synchronize()
        ms = time_kernel(V, gate, None, N, out=out, warmup=30, iters=30)
        print(f"rep {rep} {name:28s} {ms:.4f} ms")

# the same question for pure copies of the same tensors (C ABI spectre_probe_copy): flat float4 copy and the 128-byte pattern copy
from fft_amd import copy_probe
for name, f in fills.items():
    f(); torch.cuda.synchronize()
    flat = copy_probe(V, out, -1, mode="copy", warmup=5, iters=20)
    pat = min(copy_probe(V, out, 128, mode="copy", wgs_per_cu=w, warmup=5, iters=20) for w in (1, 2))
    st = min(copy_probe(V, out, 64, mode="store", wgs_per_cu=w, warmup=5, iters=20) for w in (1, 2))
    print(f"copies of {name:28s} flat {flat:.4f} ms   128-byte pattern {pat:.4f} ms   64-byte store-only (stores the probe's constant) {st:.4f} ms")
