"""Tile order A/B through the library's own switch (round 6: spectre_plan_set_tile_order): static map against tickets, interleaved, one process,
one set of tensors, for the headline shape with and without memory_fft and for the bf16 forms.   python tools/order_ab.py [rounds]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fft_amd import describe, set_tile_order, time_kernel

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = "cuda:0"
B, N, D, G = 256, 4096, 768, 4
F = N // 2 + 1
torch.manual_seed(0)
gate = torch.randn(B, G, F, dtype=torch.complex64, device=dev) * 0.3
mem = torch.randn(F, D, dtype=torch.complex64, device=dev) * 0.2
cases = [("fp32", torch.float32, torch.float32, None), ("fp32 + memory_fft", torch.float32, torch.float32, mem),
         ("bf16 -> fp32", torch.bfloat16, torch.float32, None), ("bf16 -> bf16", torch.bfloat16, torch.bfloat16, None)]
for name, din, dout, m in cases:
    V = torch.randn(B, N, D, device=dev).to(din)
    out = torch.empty(B, N, D, device=dev, dtype=dout)
    res = {"static": [], "tickets": []}
    for o in ("tickets", "static"):
        set_tile_order(N, o)
        time_kernel(V, gate, m, N, out=out, warmup=30, iters=5)
    for r in range(rounds):
        for o in (("static", "tickets") if r % 2 else ("tickets", "static")):
            set_tile_order(N, o)
            res[o].append(time_kernel(V, gate, m, N, out=out, warmup=6, iters=12))
    med = {o: sorted(v)[len(v) // 2] for o, v in res.items()}
    set_tile_order(N, "tickets")
    print(f"{name:20s} static {med['static']:.4f} ms  tickets {med['tickets']:.4f} ms  ({100 * (med['tickets'] / med['static'] - 1):+.1f} %)   [{describe(V, gate, m, N, out=out).split(' in=')[0]}]", flush=True)
    del V, out
set_tile_order(N, "auto")
