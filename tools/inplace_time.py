"""In-place (out aliases v) against out-of-place at the headline shape, interleaved, tickets and static.   python tools/inplace_time.py [rounds]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fft_amd import set_tile_order, time_kernel

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = "cuda:0"
B, N, D, G = 256, 4096, 768, 4
torch.manual_seed(0)
gate = torch.randn(B, G, N // 2 + 1, dtype=torch.complex64, device=dev) * 0.3
for dt in (torch.float32, torch.bfloat16):
    V = torch.randn(B, N, D, device=dev).to(dt)
    out = torch.empty_like(V)
    for order in ("tickets", "static"):
        set_tile_order(N, order)
        res = {"out of place": [], "in place": []}
        time_kernel(V, gate, None, N, out=out, warmup=30, iters=5)
        for r in range(rounds):
            for name, o in ((("out of place", out), ("in place", V)) if r % 2 else (("in place", V), ("out of place", out))):
                res[name].append(time_kernel(V, gate, None, N, out=o, warmup=6, iters=12))
        med = {k: sorted(v)[len(v) // 2] for k, v in res.items()}
        print(f"{str(dt):16s} {order:8s} out of place {med['out of place']:.4f} ms   in place {med['in place']:.4f} ms  ({100 * (med['in place'] / med['out of place'] - 1):+.1f} %)", flush=True)
set_tile_order(N, "auto")
