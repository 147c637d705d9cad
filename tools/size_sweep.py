"""Time the library on the BASELINE.json shapes (HIP events on the launch stream)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fft_amd import describe, time_kernel
dev = "cuda:0"
shapes = [(256, 4096, 768, "f32"), (256, 4096, 768, "bf16"), (256, 1024, 768, "f32"), (256, 256, 768, "f32"),
          (256, 2048, 768, "f32"), (256, 512, 768, "f32"), (256, 3000, 768, "f32"), (256, 3000, 768, "bf16"),
          (256, 768, 768, "f32"), (256, 1000, 768, "f32"), (256, 1280, 768, "f32"), (256, 1536, 768, "f32"), (256, 2000, 768, "f32"),
          (256, 2560, 768, "f32"), (256, 3072, 768, "f32"), (256, 3840, 768, "f32"),
          (4096, 64, 768, "f32"), (4096, 128, 768, "f32"), (2048, 196, 768, "f32"), (1024, 384, 768, "f32"), (1024, 640, 768, "f32"),
          (512, 960, 768, "f32"), (256, 1200, 768, "f32"), (256, 1920, 768, "f32"), (256, 2400, 768, "f32"), (128, 3600, 768, "f32"),
          (64, 8192, 768, "f32"), (64, 6144, 768, "f32"), (32, 16384, 768, "f32"), (32, 12288, 768, "f32"),
          (64, 6000, 768, "f32"), (64, 2039, 768, "f32")]
if len(sys.argv) > 1:
    shapes = [tuple(int(x) if x.isdigit() else x for x in s.split(",")) for s in sys.argv[1:]]
for (B, N, D, io) in shapes:
    dt = torch.float32 if io == "f32" else torch.bfloat16
    torch.manual_seed(0)
    V = torch.randn(B, N, D, device=dev).to(dt)
    g = torch.randn(B, 4, N // 2 + 1, dtype=torch.complex64, device=dev) * 0.3
    out = torch.empty_like(V)
    ms = min(time_kernel(V, g, None, N, out=out, warmup=2, iters=8) for _ in range(3))
    byt = 2 * B * N * D * V.element_size() + B * 4 * (N // 2 + 1) * 8
    print(f"({B},{N},{D}) {io:4s}: {ms:7.3f} ms  {B*N/ms/1e3:7.1f} Mtok/s  {byt/ms/1e6:6.0f} GB/s  {byt/ms/1e6/80:5.1f}% of 8TB/s  [{describe(V, g, None, N)[:60]}]")
    del V, g, out
