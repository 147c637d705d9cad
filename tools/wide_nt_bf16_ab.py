import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys
sys.path.insert(0, %r)
import torch
from fft_amd import time_kernel
dev = "cuda:0"
res = []
B, N, D = 256, 1024, 768
torch.manual_seed(0)
V = torch.randn(B, N, D, device=dev); g = torch.randn(B, 4, N // 2 + 1, dtype=torch.complex64, device=dev) * 0.3
for tin, tout in ((torch.float32, torch.float32), (torch.bfloat16, torch.float32), (torch.bfloat16, torch.bfloat16)):
    Vv = V.to(tin); out = torch.empty(B, N, D, dtype=tout, device=dev)
    res.append("%%s->%%s %%.4f" %% (str(tin)[6:10], str(tout)[6:10], min(time_kernel(Vv, g, None, N, out=out, warmup=30, iters=20) for _ in range(3))))
print("MS " + "  ".join(res))
''' % ROOT
for r in range(3):
    for nt in (None, "0", "3"):
        env = dict(os.environ)
        if nt is not None: env.update(SPECTRE_TUNING="1", SPECTRE_WIDE_NT=nt)
        out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        print("nt=" + str(nt), [l for l in out.stdout.splitlines() if l.startswith("MS")], out.stderr[-300:] if out.returncode else "")
