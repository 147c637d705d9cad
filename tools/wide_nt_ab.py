import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys
sys.path.insert(0, %r)
import torch
from fft_amd import time_kernel
dev = "cuda:0"
res = []
for (B, N, D) in [(256, 1024, 768), (512, 512, 768)]:
    torch.manual_seed(0)
    V = torch.randn(B, N, D, device=dev); g = torch.randn(B, 4, N // 2 + 1, dtype=torch.complex64, device=dev) * 0.3; out = torch.empty_like(V)
    res.append("%%d: %%.4f" %% (N, min(time_kernel(V, g, None, N, out=out, warmup=30, iters=20) for _ in range(3))))
print("MS " + "  ".join(res))
''' % ROOT
for r in range(3):
    for nt in ("0", "1", "2", "3"):
        out = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, SPECTRE_TUNING="1", SPECTRE_WIDE_NT=nt), capture_output=True, text=True)
        print("nt=" + nt, [l for l in out.stdout.splitlines() if l.startswith("MS")], out.stderr[-300:] if out.returncode else "")
