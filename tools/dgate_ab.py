"""A/B of the gate gradient at (256, 4096, 768) between two builds of the library, interleaved in ONE run: each build in its own process
(SPECTRE_HIP_LIB selects the .so), alternating, several rounds; prints per-round times.
    python tools/dgate_ab.py <lib_a.so> <lib_b.so> [rounds]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys
sys.path.insert(0, %r)
import torch
from fft_amd import spectral_mix_backward
dev = "cuda:0"
B, N, D = 256, 4096, 768
torch.manual_seed(0)
V = torch.randn(B, N, D, device=dev); g = torch.randn(B, 4, N // 2 + 1, dtype=torch.complex64, device=dev) * 0.3; do = torch.randn(B, N, D, device=dev)
for dt in (torch.float32, torch.bfloat16):
    Vv, dd = V.to(dt), do.to(dt)
    for _ in range(25):
        spectral_mix_backward(Vv, g, dd, N, need_dv=False, need_dgate=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        spectral_mix_backward(Vv, g, dd, N, need_dv=False, need_dgate=True)
    e1.record(); torch.cuda.synchronize()
    print(str(dt)[6:], "%%.4f" %% (e0.elapsed_time(e1) / 10))
''' % ROOT
libs = sys.argv[1:3]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
res = {l: [] for l in libs}
for r in range(rounds):
    for l in libs:
        out = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, SPECTRE_HIP_LIB=os.path.abspath(l)), capture_output=True, text=True)
        vals = [ln.split() for ln in out.stdout.splitlines() if ln.split() and ln.split()[0] in ("float32", "bfloat16")]
        res[l].append({k: float(v) for k, v in vals})
        if out.returncode:
            print(out.stderr[-800:])
for l in libs:
    print(l, " | ".join("f32 %.4f bf16 %.4f" % (d.get("float32", -1), d.get("bfloat16", -1)) for d in res[l]))
