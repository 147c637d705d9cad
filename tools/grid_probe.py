"""Does the 4096 kernel want all 256 CUs?  One subprocess per (tiles per workgroup, tile order): SPECTRE_P64_TPW forces the tiles a
workgroup walks through (48 = 256 workgroups at the headline shape, 52 = 238, 56 = 220, 64 = 192), the time is the median of 40 launches
behind 60 warm-up launches, all on ONE pair of tensors per subprocess (allocation classes differ between processes: read columns, not rows).

    python tools/grid_probe.py [f32|bf16|bf16out]
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, torch
sys.path.insert(0, %r)
from fft_amd import spectral_mix, describe
io = sys.argv[1]
B, N, D = 256, 4096, 768
dt = torch.float32 if io == "f32" else torch.bfloat16
torch.manual_seed(0)
V = torch.randn(B, N, D, device="cuda").to(dt)
gate = torch.randn(B, 4, N // 2 + 1, dtype=torch.complex64, device="cuda") * 0.3
out = torch.empty(B, N, D, device="cuda", dtype=torch.float32 if io == "bf16" else dt)
for _ in range(60): spectral_mix(V, gate, None, N, out=out)
torch.cuda.synchronize()
ts = []
for _ in range(40):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); spectral_mix(V, gate, None, N, out=out); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
ts.sort()
print("%%.4f %%.4f %%s" %% (ts[len(ts) // 2], ts[0], describe(V, gate, None, N, out=out)))
''' % ROOT
io = sys.argv[1] if len(sys.argv) > 1 else "f32"
for order in ("static", "tickets"):
    for tpw in (48, 50, 52, 56, 64, 96):
        env = dict(os.environ, SPECTRE_TUNING="1", SPECTRE_P64_TPW=str(tpw), SPECTRE_TILE_ORDER=order)
        r = subprocess.run([sys.executable, "-c", CHILD, io], env=env, capture_output=True, text=True)
        print(f"{io:8s} order={order:8s} tpw={tpw:3d}  median/min ms + kernel: {r.stdout.strip() or r.stderr.strip()[-300:]}", flush=True)
