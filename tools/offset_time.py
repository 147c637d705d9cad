"""Does the placement of `out` relative to `V` matter?  The headline kernel with out = views of one big buffer at different byte offsets
(the row stride stays 3072 B).  Steady state: 40 untimed + 60 timed launches per point."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fft_amd import spectral_mix
dev = torch.device("cuda:0")
B, N, D = 256, 4096, 768
n = B * N * D
V = torch.randn(B, N, D, device=dev)
gate = torch.randn(B, 4, N // 2 + 1, dtype=torch.complex64, device=dev) * 0.3
big = torch.empty(n + (96 << 20), device=dev)          # + 384 MiB of slack
print(f"V at {V.data_ptr():#x}, big at {big.data_ptr():#x}, distance {(big.data_ptr() - V.data_ptr()) / 2**20:.3f} MiB", flush=True)
def t(out, reps=60):
    for _ in range(40):
        spectral_mix(V, gate, None, N, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        spectral_mix(V, gate, None, N, out=out)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
base = t(big[:n].view(B, N, D))
print(f"offset 0: {base:.4f} ms", flush=True)
for off_bytes in (128, 256, 1024, 4096, 3072 * 4, 16384, 65536, 1 << 18, 1 << 20, 3 << 20, 1 << 22, 12 << 20, 1 << 24, 48 << 20, 1 << 26, 192 << 20, 1 << 28, 0):
    o = big[off_bytes // 4: off_bytes // 4 + n].view(B, N, D)
    print(f"offset {off_bytes:>10d} B ({off_bytes / 2**20:8.3f} MiB): {t(o):.4f} ms", flush=True)
