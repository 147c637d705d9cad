"""Latency of one decode step (SpectreHead.decode_step on a PrefixFFTCache) and of the prefill, at the headline width."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fft_amd import PrefixFFTCache, SpectreHead
dev = torch.device("cuda:0")
for (N, d, G) in [(4096, 768, 4), (4096, 64, 4), (1024, 768, 4)]:
    torch.manual_seed(0)
    head = SpectreHead(d, N, num_groups=G, pooling_type="mean").to(dev).eval()
    cache = PrefixFFTCache(N, d, device=dev)
    Q, V = torch.randn(N - 8, d, device=dev), torch.randn(N - 8, d, device=dev)
    cache.prefill(Q, V); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): cache.prefill(Q, V)
    torch.cuda.synchronize(); pre = (time.perf_counter() - t0) / 5 * 1e3
    q, v = torch.randn(64, d, device=dev), torch.randn(64, d, device=dev)
    for i in range(16): head.decode_step(q[i], v[i], cache)          # crosses the ring wrap
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(16, 64): head.decode_step(q[i], v[i], cache)
    torch.cuda.synchronize(); step = (time.perf_counter() - t0) / 48 * 1e3
    # kernel-only: the fused spectrum pass
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(48): cache.decode_step(q[i], v[i])
    e1.record(); torch.cuda.synchronize()
    print(f"n_fft={N} d={d}: prefill {pre:.3f} ms; decode step {step*1e3:.0f} us end to end (one C-ABI call, four launches); "
          f"state-only step {e0.elapsed_time(e1)/48*1e3:.0f} us; spectrum = {(N//2+1)*d*8/1e6:.1f} MB read + written per step")
