import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys
sys.path.insert(0, %r)
import torch
from fft_amd import time_kernel, spectral_mix
dev = "cuda:0"
B, N, D = 256, 3000, 768
torch.manual_seed(0)
V = torch.randn(B, N, D, device=dev); g = torch.randn(B, 4, N // 2 + 1, dtype=torch.complex64, device=dev) * 0.3; out = torch.empty_like(V)
ms = [time_kernel(V, g, None, N, out=out, warmup=40, iters=20) for _ in range(3)]
ref = spectral_mix(V[:1], g[:1], None, N, algo="stockham")
print("MS %%.4f %%.4f %%.4f  maxdiff %%.1e" %% (ms[0], ms[1], ms[2], float((out[:1] - ref).abs().max())))
''' % ROOT
for r in range(3):
    for bar in ("0", "4", "8", "12"):
        out = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, SPECTRE_TUNING="1", SPECTRE_MIXEDP_BAR=bar), capture_output=True, text=True)
        print("bar=" + bar, [l for l in out.stdout.splitlines() if l.startswith("MS")], out.stderr[-400:] if out.returncode else "")
