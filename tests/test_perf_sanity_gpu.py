"""Dispatch sanity on the GPU (round 6; VERDICT r05 item 8: no timing asserts).  Rounds 3-5 guarded against a code-generation pathology
— hipcc serialising the bf16 loads of five mixed-radix lengths and of the bf16 gate gradient: one request in flight per wave, 2 x slower,
invisible to every parity test — by asserting that a bf16 launch takes at most 1.4 x the fp32 launch's time: green, but a flake waiting
for a noisy box.  The pathology itself is now caught where it arises: fft_amd/isa_lint.py recounts `load ; s_waitcnt vmcnt(0)` runs in
the listing of EVERY translation unit at build time and the build fails on a new one (tests/test_isa_lint_cpu.py).  What is left for the
GPU tier is what only `spectre_mix_describe` can say: every BASELINE shape and every length of the old timing test is dispatched to the
register-tile family it is meant to take, for fp32 and for bf16 rows alike — a shape that silently fell back to the LDS Stockham
kernel (0.12 of the roofline) would pass every parity test too."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

WANT = {  # n_fft -> kernel family named by describe() at (B, n, 768), G = 4, fast mode
    256: "regtile-wide", 512: "regtile-wide", 1024: "regtile-wide", 2048: "regtile 64x32", 4096: "regtile-pipelined 64x64",
    3000: "regtile-mixed-pipelined 60x50", 2560: "regtile-mixed-pipelined", 2400: "regtile-mixed-pipelined", 3072: "regtile-mixed-pipelined",
    3600: "regtile-mixed-pipelined", 3840: "regtile-mixed-pipelined", 960: "regtile-mixed", 1536: "regtile-mixed", 1920: "regtile-mixed",
    2000: "regtile-mixed", 6144: "regtile-long", 8192: "regtile-long", 12288: "regtile-quad", 16384: "regtile-quad",
}


@pytest.mark.parametrize("n", sorted(WANT))
def test_every_length_takes_its_register_tile_kernel_for_fp32_and_bf16_rows(n):
    from fft_amd import describe
    B = 4
    g = torch.zeros(B, 4, n // 2 + 1, dtype=torch.complex64, device=DEV)
    for dt in (torch.float32, torch.bfloat16):
        V = torch.zeros(B, n, 768, device=DEV, dtype=dt)
        d = describe(V, g, None, n)
        if dt == torch.bfloat16 and n in (3000, 2560, 2400, 3072, 3600, 3840):
            assert d.startswith("regtile-mixed"), (n, dt, d)       # bf16 rows: the one-tile-per-workgroup mixed-radix form of the same length
        else:
            assert d.startswith(WANT[n]), (n, dt, d)
        assert "stockham" not in d


def test_headline_shapes_name_their_shipped_instantiations():
    from fft_amd import describe
    g = torch.zeros(2, 4, 2049, dtype=torch.complex64, device=DEV)
    V = torch.zeros(2, 4096, 768, device=DEV)
    assert "order=" in describe(V, g, None, 4096)                    # the persistent kernel with a tile order
    mem = torch.zeros(2049, 768, dtype=torch.complex64, device=DEV)
    assert "order=" in describe(V, g, mem, 4096)                     # memory_fft: the same persistent kernel family (round 6: on tickets too)
    assert describe(V[:, :3000], g, None, 4096).startswith("regtile-pipelined")     # a padded sequence costs what a full one costs
    assert describe(V.to(torch.bfloat16), g, None, 4096, out_dtype=torch.float32).startswith("regtile-pipelined")
