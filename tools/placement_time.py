"""Does it matter WHICH allocations the headline kernel reads and writes?  K buffers of the tensor's size, every ordered pair (in, out)
timed with the same kernel, same data: 40 untimed + 40 timed launches per pair, twice.  (bench.py runs showed the forward and the
dV launch of one process — the same kernel on different tensors — on different levels, 1.44 vs 1.56 ms, and the other way round on
the next box.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fft_amd import spectral_mix
dev = torch.device("cuda:0")
B, N, D, K = 256, 4096, 768, 6
gate = torch.randn(B, 4, N // 2 + 1, dtype=torch.complex64, device=dev) * 0.3
bufs = [torch.randn(B, N, D, device=dev) for _ in range(K)]
for i, b in enumerate(bufs):
    print(f"buffer {i} at {b.data_ptr():#x}", flush=True)
def t(i, o, reps=40):
    for _ in range(40):
        spectral_mix(bufs[i], gate, None, N, out=bufs[o])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        spectral_mix(bufs[i], gate, None, N, out=bufs[o])
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for rnd in range(2):
    print(f"round {rnd}:  rows = input buffer, columns = output buffer (ms)")
    for i in range(K):
        print(f"  in {i}: " + " ".join("  --  " if i == o else f"{t(i, o):.4f}" for o in range(K)), flush=True)

# the pure-copy probes (C ABI spectre_probe_copy) on the same buffers: is it the allocation (every access pattern) or the pattern?
from fft_amd import copy_probe
for name, seg in (("dense copy", 0), ("128-byte row segments", 128), ("64-byte row segments", 64)):
    print(f"{name}: rows = source buffer, columns = destination buffer (ms, best of 1/2/4 workgroups per CU)")
    for i in range(K):
        row = []
        for o in range(K):
            if i == o:
                row.append("  --  "); continue
            row.append(f"{min(copy_probe(bufs[i], bufs[o], seg, wgs_per_cu=w, warmup=10, iters=20) for w in (1, 2, 4)):.4f}")
        print(f"  src {i}: " + " ".join(row), flush=True)
