"""n_fft <= 1024: whole-line (32-channel) tiles against the 16-channel tiles, same process, interleaved (SPECTRE_TUNING=1 SPECTRE_WIDE=0 is
read once per process, so the two forms run in two child processes, alternating)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys
sys.path.insert(0, %r)
import torch
from fft_amd import time_kernel, describe
dev = "cuda:0"
for (B, N, D, dt) in [(256, 1024, 768, torch.float32), (512, 512, 768, torch.float32), (1024, 256, 768, torch.float32), (256, 1024, 768, torch.bfloat16)]:
    torch.manual_seed(0)
    V = torch.randn(B, N, D, device=dev).to(dt); g = torch.randn(B, 4, N // 2 + 1, dtype=torch.complex64, device=dev) * 0.3; out = torch.empty_like(V)
    ms = min(time_kernel(V, g, None, N, out=out, warmup=30, iters=20) for _ in range(3))
    byt = 2 * B * N * D * V.element_size() + B * 4 * (N // 2 + 1) * 8
    print("%%s %%d %%s %%.4f ms %%.3f of 8 TB/s [%%s]" %% (str(dt)[6:], N, B, ms, byt / ms / 1e6 / 8000, describe(V, g, None, N)[:24]))
''' % ROOT
for r in range(3):
    for env in ({}, {"SPECTRE_TUNING": "1", "SPECTRE_WIDE": "0"}):
        out = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, **env), capture_output=True, text=True)
        print("wide" if not env else "16ch", "|", " | ".join(l for l in out.stdout.splitlines() if " ms " in l))
        if out.returncode: print(out.stderr[-600:])
