"""Whole-line tiles: persistent + register double buffering (SPECTRE_WIDEP=1) against one tile per workgroup (SPECTRE_WIDEP=0)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys
sys.path.insert(0, %r)
import torch
from fft_amd import time_kernel, describe, spectral_mix
dev = "cuda:0"
res = []
for (B, N, D, dt) in [(256, 1024, 768, torch.float32), (512, 512, 768, torch.float32), (1024, 256, 768, torch.float32), (256, 1024, 768, torch.bfloat16)]:
    torch.manual_seed(0)
    V = torch.randn(B, N, D, device=dev).to(dt); g = torch.randn(B, 4, N // 2 + 1, dtype=torch.complex64, device=dev) * 0.3; out = torch.empty_like(V)
    ms = min(time_kernel(V, g, None, N, out=out, warmup=30, iters=20) for _ in range(3))
    ref = spectral_mix(V[:3], g[:3], None, N, algo="stockham").float()
    err = float((out[:3].float() - ref).abs().max())
    res.append("%%d%%s %%.4f (%%.0e)" %% (N, "b" if dt == torch.bfloat16 else "", ms, err))
print("MS " + "  ".join(res) + "  [" + describe(V, g, None, N)[:22] + "]")
''' % ROOT
for r in range(3):
    for sel in ("1", "0"):
        out = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, SPECTRE_TUNING="1", SPECTRE_WIDEP=sel), capture_output=True, text=True)
        print("widep=" + sel, [l for l in out.stdout.splitlines() if l.startswith("MS")], out.stderr[-400:] if out.returncode else "")
