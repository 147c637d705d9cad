"""What would feeding the mix kernel straight from the W_v GEMM cost?  (SURVEY §8(f) N3 "longer-term", declined in DESIGN §1(f) on an
arithmetic argument; VERDICT r02 asked for a measurement.)  A fused tile needs ALL 4096 rows of 16 output channels of one batch element:
  V[b][:, 16t:16t+16] = x[b] (4096 x 768) @ W_v[16t:16t+16, :].T          -- a skinny GEMM per (b, t), x[b] re-read 48 times
Timed here with hipBLASLt through torch: (1) the layer's real projection, one big GEMM; (2) the same FLOPs as 48 skinny GEMMs per batch
element, the access pattern a W_v epilogue inside the mix kernel would have (x[b] = 12.6 MB fp32 stays in the 256-MiB Infinity Cache between
the 48 passes at best); (3) the HBM round trip of V that the fusion would save, as a dense copy of the same bytes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fft_amd import copy_probe
dev = torch.device("cuda:0")
B, N, D, T = 256, 4096, 768, 16
def timeit(f, reps=5, warm=2):
    for _ in range(warm): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for dt in (torch.float32, torch.bfloat16):
    x = torch.randn(B, N, D, device=dev, dtype=dt)
    W = torch.randn(D, D, device=dev, dtype=dt) * 0.03
    V = torch.empty(B, N, D, device=dev, dtype=dt)
    big = timeit(lambda: torch.matmul(x, W.t(), out=V))
    flops = 2.0 * B * N * D * D
    # skinny: for every batch element, 48 GEMMs (4096 x 768) @ (768 x 16); batched over b to keep launch overhead out of the picture
    Wt = W.view(D // T, T, D)                                  # (48, 16, 768)
    Vt = torch.empty(D // T, B, N, T, device=dev, dtype=dt)
    def skinny():
        for t in range(D // T):
            torch.matmul(x, Wt[t].t(), out=Vt[t])              # (B, 4096, 768) @ (768, 16) -> (B, 4096, 16)
    sk = timeit(skinny, reps=2, warm=1)
    # one tile-column for all batch elements = the unit the mix kernel would interleave with its own work
    one = timeit(lambda: torch.matmul(x, Wt[0].t(), out=Vt[0]), reps=5)
    es = x.element_size()
    print(f"{str(dt)[6:]:9s} W_v as one GEMM {big:7.3f} ms ({flops / big / 1e9:6.1f} TFLOP/s) | as 48 skinny (4096x768)@(768x16) GEMMs per batch element {sk:8.2f} ms "
          f"({flops / sk / 1e9:5.1f} TFLOP/s; one column tile of all batch elements {one:6.3f} ms = {B * N * D * es / one / 1e6:6.0f} GB/s of x re-read)", flush=True)
    del Vt
src = torch.randn(B, N, D, device=dev); dst = torch.empty_like(src)
ms = min(copy_probe(src, dst, 0, wgs_per_cu=k, warmup=2, iters=5) for k in (1, 2, 4))
print(f"the HBM round trip of V the fusion would save (fp32: write 3.2 GB + read 3.2 GB, as a dense copy): {ms:.3f} ms")
