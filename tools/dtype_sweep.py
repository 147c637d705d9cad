"""Every register-tile length at (B, n, 768), B = 768000 / n: fp32 rows against bf16 rows, forward and gate gradient.  bf16 moves half
the bytes; a length where it is not faster than fp32 has a code-generation problem (round 3: hipcc had serialised the bf16 loads of the
RF = 48 / 64 mixed-radix kernels, one request in flight per wave)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fft_amd import time_kernel, describe, spectral_mix_backward
dev = "cuda:0"
LENGTHS = [64, 128, 196, 256, 384, 512, 640, 768, 960, 1000, 1024, 1200, 1280, 1536, 1920, 2000, 2048, 2400, 2560, 3000, 3072, 3600, 3840, 4096, 6144, 8192]
def dgate(V, g, n):
    do = torch.randn_like(V)
    try:
        for _ in range(3): spectral_mix_backward(V, g, do, n, need_dv=False, need_dgate=True)
    except Exception:
        return float("nan")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): spectral_mix_backward(V, g, do, n, need_dv=False, need_dgate=True)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 5
for n in LENGTHS:
    B = max(8, min(4096, (256 * 3000) // n))
    V = torch.randn(B, n, 768, device=dev); g = torch.randn(B, 4, n // 2 + 1, dtype=torch.complex64, device=dev) * 0.3
    r = {}
    for name, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
        Vd = V.to(dt); out = torch.empty_like(Vd)
        r[name] = min(time_kernel(Vd, g, None, n, out=out, warmup=20, iters=8) for _ in range(2))
        r["dgate_" + name] = dgate(Vd, g, n)
    flag = "  <-- bf16 not faster" if r["bf16"] > 0.98 * r["f32"] else ""
    flag += "  <-- bf16 dgate slower" if r["dgate_bf16"] > 1.05 * r["dgate_f32"] else ""
    print(f"n={n:5d}: fwd f32 {r['f32']:.4f}  bf16 {r['bf16']:.4f}   dgate f32 {r['dgate_f32']:.4f}  bf16 {r['dgate_bf16']:.4f}{flag}", flush=True)
