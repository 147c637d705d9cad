"""How long does the chip need to reach its steady state?  Per-launch duration of the headline kernel over a long back-to-back run that
starts from an idle GPU (and again after a 2 s pause, and after a burst of light kernels)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fft_amd import spectral_mix

dev = torch.device("cuda:0")
B, N, D, G = 256, 4096, 768, 4
V = torch.randn(B, N, D, device=dev)
gate = torch.randn(B, G, N // 2 + 1, dtype=torch.complex64, device=dev) * 0.3
out = torch.empty_like(V)
torch.cuda.synchronize()

def run(tag, n=500):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        spectral_mix(V, gate, None, N, out=out)
        ev[i + 1].record()
    torch.cuda.synchronize()
    ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
    def avg(a, b): return sum(ms[a:b]) / (b - a)
    print(f"{tag:38s} launches 0-4 {avg(0,5):.3f}  5-9 {avg(5,10):.3f}  10-24 {avg(10,25):.3f}  25-49 {avg(25,50):.3f}  50-99 {avg(50,100):.3f}  "
          f"100-199 {avg(100,200):.3f}  200-349 {avg(200,350):.3f}  350-499 {avg(350,500):.3f} ms", flush=True)

run("from idle (after setup)")
run("immediately again")
time.sleep(2.0)
run("after a 2 s pause")
x = torch.zeros(1 << 20, device=dev)
for _ in range(3000):
    x.add_(1.0)                       # ~10 us kernels: the GPU is busy but almost idle
run("after 3000 tiny kernels")
time.sleep(10.0)
run("after a 10 s pause")
