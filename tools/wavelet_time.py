#!/usr/bin/env python3
"""Time of the wavelet refinement launch at the headline shape (B, 4096, 768), in place, for a few on-rates; torch events on the current
stream.  Algorithmic bytes = the switched-on elements' rows read once and written once (+ the second read of v for the final add, which the
launch takes from the L2 / HBM again: counted as traffic, not as algorithmic bytes)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fft_amd

B, N, D = 256, 4096, 768
dev = "cuda:0"
v = torch.randn(B, N, D, device=dev)
gate = torch.rand(B, D, device=dev)
for rate in (0.0, 0.1, 0.5, 1.0):
    g = torch.Generator(device=dev).manual_seed(7)
    mask = torch.rand(B, device=dev, generator=g) < rate
    n_on = int(mask.sum())
    for _ in range(3):
        fft_amd.wavelet_refine(v, gate, mask, inplace=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fft_amd.wavelet_refine(v, gate, mask, inplace=True)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    alg = 2 * n_on * N * D * 4
    print(f"on_rate {rate:.1f}: {n_on:3d} of {B} elements on, {ms:.4f} ms, algorithmic {alg / 1e9:.3f} GB -> {alg / ms / 1e9 if ms else 0:.2f} TB/s", flush=True)
    v.copy_(torch.randn(B, N, D, device=dev))
