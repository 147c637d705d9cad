"""Gate gradient at (256, 4096, 768): the prefetch form (kernel_regtile_grad.h, PN row blocks of the next tile in registers of their own)
against the shipped form, SAME process, same tensors, alternating (SPECTRE_TUNING=1 SPECTRE_DGATE_PREFETCH=<PN> is read per call).
    python tools/dgate_pn.py [rounds] [PN list, default 0,16,24,32]
Prints per-round times per PN and dtype and checks every PN's result against PN = 0 (bit-equal: same arithmetic, same order)."""
import os, sys
os.environ["SPECTRE_TUNING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fft_amd import spectral_mix_backward

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
pns = [x for x in (sys.argv[2] if len(sys.argv) > 2 else "0,1").split(",")]      # "PN" or "PN:GRID" (grid of the persistent form)
shapes = [(256, 4096, 768, 4)] + ([(64, 4096, 512, 8)] if "--more" in sys.argv else [])
dev = "cuda:0"
for (B, N, D, G) in shapes:
    torch.manual_seed(0)
    V = torch.randn(B, N, D, device=dev); g = torch.randn(B, G, N // 2 + 1, dtype=torch.complex64, device=dev) * 0.3
    do = torch.randn(B, N, D, device=dev)
    for dt in (torch.float32, torch.bfloat16):
        Vv, dd = V.to(dt), do.to(dt)
        ref = None
        res = {pn: [] for pn in pns}
        for r in range(rounds + 1):
            for pn in pns:
                os.environ["SPECTRE_DGATE_PREFETCH"] = pn.split(":")[0]
                if ":" in pn:
                    os.environ["SPECTRE_DGATE_GRID"] = pn.split(":")[1]
                else:
                    os.environ.pop("SPECTRE_DGATE_GRID", None)
                for _ in range(25 if r == 0 else 4):
                    out = spectral_mix_backward(Vv, g, dd, N, need_dv=False, need_dgate=True)
                torch.cuda.synchronize()
                if r == 0:
                    dg = out[1] if isinstance(out, (tuple, list)) else out
                    if ref is None:
                        ref = dg.clone()
                    else:
                        same = torch.equal(torch.view_as_real(dg), torch.view_as_real(ref))
                        again = spectral_mix_backward(Vv, g, dd, N, need_dv=False, need_dgate=True)[1]
                        det = torch.equal(torch.view_as_real(dg), torch.view_as_real(again))
                        print(f"  ({B},{N},{D}) {str(dt)[6:]} PN={pn}: {'bit-equal to PN=0' if same else 'max |diff| / max |ref| = %.2e' % ((dg - ref).abs().max().item() / ref.abs().max().item())}"
                              f"{'' if det else '   NOT DETERMINISTIC (two launches differ)'}", flush=True)
                    continue
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    spectral_mix_backward(Vv, g, dd, N, need_dv=False, need_dgate=True)
                e1.record(); torch.cuda.synchronize()
                res[pn].append(e0.elapsed_time(e1) / 10)
        for pn in pns:
            med = sorted(res[pn])[len(res[pn]) // 2]
            print(f"({B},{N},{D}) {str(dt)[6:]:9s} PN={pn:>9s}: " + " ".join("%.4f" % x for x in res[pn]) + f"   median {med:.4f} ms  ({med / sorted(res[pns[0]])[len(res[pns[0]]) // 2] - 1:+.1%})", flush=True)
