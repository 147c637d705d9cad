"""Row stride of `out` against the allocation classes: for each of `reps` rounds a NEW output buffer per stride (earlier ones stay allocated, so
every buffer sits on other physical memory), the headline launch under both tile orders on each; medians and spread per stride.
python tools/stride_ab2.py [reps] [which=out|both]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fft_amd import set_tile_order, time_kernel

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
which = sys.argv[2] if len(sys.argv) > 2 else "out"
dev = "cuda:0"
B, N, D, G = 256, 4096, 768, 4
F = N // 2 + 1
torch.manual_seed(0)
gate = torch.randn(B, G, F, dtype=torch.complex64, device=dev) * 0.3
Vc = torch.randn(B, N, D, device=dev)
pads = [768, 800, 864, 896, 928]
keep = []
res = {(Dp, o): [] for Dp in pads for o in ("tickets", "static")}
for rep in range(reps):
    for Dp in (pads if rep % 2 == 0 else pads[::-1]):
        outb = torch.empty(B, N, Dp, device=dev)
        keep.append(outb)
        out = outb[:, :, :D]
        V = Vc
        if which == "both":
            vb = torch.empty(B, N, Dp, device=dev)
            keep.append(vb)
            vb[:, :, :D] = Vc
            V = vb[:, :, :D]
        for o in ("tickets", "static"):
            set_tile_order(N, o)
            time_kernel(V, gate, None, N, out=out, warmup=25, iters=5)
            res[(Dp, o)].append(time_kernel(V, gate, None, N, out=out, warmup=4, iters=16))
    print(f"round {rep}: " + "  ".join(f"{Dp * 4}B t {res[(Dp, 'tickets')][-1]:.3f} s {res[(Dp, 'static')][-1]:.3f}" for Dp in pads), flush=True)
print(f"\nstride of {which}: median (min .. max) over {reps} buffers each")
for Dp in pads:
    row = []
    for o in ("tickets", "static"):
        t = sorted(res[(Dp, o)])
        row.append(f"{o} {t[len(t) // 2]:.4f} ({t[0]:.4f} .. {t[-1]:.4f})")
    best = sorted(min(a, b) for a, b in zip(res[(Dp, 'tickets')], res[(Dp, 'static')]))
    print(f"{Dp * 4:5d} B: " + "   ".join(row) + f"   better of the two: {best[len(best) // 2]:.4f}", flush=True)
set_tile_order(N, "auto")
