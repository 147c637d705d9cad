"""Every mixed-radix length of the register-tile family, one line each: the fast mode (full sequences), a general mode (7 rows short),
memory_fft, bf16 rows and the gate gradient at (B, n, 768) with B scaled to ~600 M input elements.  A/B between two builds:
SPECTRE_HIP_LIB=<path> python tools/mixed_engine_ab.py > a.log, again with the other library, then paste the two side by side
(tools/mixed_engine_ab.py --join a.log b.log)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

LENGTHS = [64, 128, 196, 384, 640, 768, 960, 1000, 1200, 1280, 1536, 1920, 2000, 2400, 2560, 3000, 3072, 3600, 3840]

def join(a, b):
    ra = {l.split(":")[0]: l.split(":")[1].split() for l in open(a) if l.startswith("n=")}
    rb = {l.split(":")[0]: l.split(":")[1].split() for l in open(b) if l.startswith("n=")}
    names = ["fast", "short", "mem", "bf16", "dgate"]
    print(f"{'':8s}" + "".join(f"{n + ' A':>9s}{n + ' B':>9s}{'B/A':>7s}" for n in names))
    for k in ra:
        if k not in rb: continue
        row = f"{k:8s}"
        for x, y in zip(ra[k], rb[k]):
            row += f"{x:>9s}{y:>9s}" + (f"{float(y) / float(x):7.3f}" if x != "-" and y != "-" else f"{'':7s}")
        print(row)

if len(sys.argv) == 4 and sys.argv[1] == "--join":
    join(sys.argv[2], sys.argv[3]); sys.exit(0)

import torch
from fft_amd import time_kernel, spectral_mix_backward
dev = "cuda:0"
D, G = 768, 4
def t_fwd(V, g, m, n):
    out = torch.empty(V.shape[0], min(V.shape[1], n), D, device=dev, dtype=V.dtype)
    return min(time_kernel(V, g, m, n, out=out, warmup=3, iters=8) for _ in range(3))
def t_dgate(V, g, n):
    do = torch.randn_like(V)
    try:
        for _ in range(2): spectral_mix_backward(V, g, do, n, need_dv=False, need_dgate=True)
    except Exception:
        return None
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): spectral_mix_backward(V, g, do, n, need_dv=False, need_dgate=True)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 5)
    return best
for n in LENGTHS:
    B = max(8, min(4096, (256 * 3000) // n))
    torch.manual_seed(n)
    V = torch.randn(B, n, D, device=dev); g = torch.randn(B, G, n // 2 + 1, dtype=torch.complex64, device=dev) * 0.3
    m = torch.randn(n // 2 + 1, D, dtype=torch.complex64, device=dev)
    r = [t_fwd(V, g, None, n), t_fwd(V[:, : n - 7].contiguous(), g, None, n), t_fwd(V, g, m, n), t_fwd(V.bfloat16(), g, None, n), t_dgate(V, g, n)]
    print(f"n={n}: " + " ".join("-" if x is None else f"{x:.4f}" for x in r), flush=True)
