"""Parity of the HIP path with the oracle, through the C ABI, on a real MI355X (`-m gpu`).

  * every golden fixture (outputs of the reference itself), on both kernels
  * seeded random problems against the float64 oracle at sizes it finishes in seconds
  * BASELINE.json's full sizes through size-independent properties (unit gate = identity, linearity,
    circular-shift equivariance, Hermitian-edge rule, shard/concat equality)
Tolerance: |y - e| <= 1e-4 |e| + 1e-4 RMS(e) for fp32 output; bf16 output equals the
bf16-rounded fp32 result of the same kernel bit for bit, and the oracle's rounding within 1 ulp.
"""
import numpy as np
import pytest
import torch

from conftest import golden_files, golden_ids, load_golden
from oracle.spectral_mix_oracle import assert_close, bf16_round, spectral_mix_numpy

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _mix(*a, **k):
    from fft_amd import spectral_mix
    y = spectral_mix(*a, **k)
    torch.cuda.synchronize()
    return y


def _describe(*a, **k):
    from fft_amd import describe
    return describe(*a, **k)


def _t(x):
    return None if x is None else torch.from_numpy(np.ascontiguousarray(x)).to(DEV)


def _problem(seed, B, N, D, G, n_fft, dtype=torch.float32, mem=False, zero_frac=0.18):
    g = torch.Generator(device="cpu").manual_seed(seed)
    V = torch.randn(B, N, D, generator=g).to(dtype)
    F = n_fft // 2 + 1
    gate = torch.complex(torch.randn(B, G, F, generator=g), torch.randn(B, G, F, generator=g)) * 0.3
    gate = gate * (torch.rand(B, G, F, generator=g) >= zero_frac)          # modReLU leaves exact zeros
    m = torch.complex(torch.randn(F, D, generator=g), torch.randn(F, D, generator=g)) * 0.2 if mem else None
    return V, gate.to(torch.complex64), m


def _oracle(V, gate, mem, n_fft):
    return spectral_mix_numpy(V.float().numpy(), gate.numpy(), None if mem is None else mem.numpy(), n_fft)


def test_native_library_is_the_thing_that_runs():
    from fft_amd import _native
    lib = _native.load()
    assert lib.spectre_version() == _native.ABI_VERSION
    V, gate, _ = _problem(0, 2, 4096, 32, 2, 4096)
    assert _describe(V.to(DEV), gate.to(DEV)).startswith("regtile-pipelined 64x64")   # fast mode, fp32, 16-byte aligned rows
    assert _describe(V.to(DEV)[:, :4000], gate.to(DEV), None, 4096).startswith("regtile-pipelined 64x64 in=f32 out=f32 mode=3")   # padded: same kernel
    assert _describe(V.to(DEV).bfloat16(), gate.to(DEV), None, 4096).startswith("regtile-pipelined 64x64 in=bf16 out=bf16")   # bf16 rows in and out: same kernel
    Vb = torch.zeros(2, 4096, 34, device=DEV, dtype=torch.bfloat16)[:, :, 2:]     # a bf16 view that is only 4-byte aligned: the round-1 kernel
    assert _describe(Vb, gate.to(DEV), None, 4096).startswith("regtile 64x64 in=bf16 out=bf16")
    assert _describe(V.to(DEV), gate.to(DEV), algo="stockham").startswith("stockham")
    for n, tag in ((256, "16x16"), (512, "32x16"), (1024, "32x32"), (2048, "64x32")):
        Vn, gn, _ = _problem(0, 1, n, 16, 1, n)
        assert _describe(Vn.to(DEV), gn.to(DEV)).startswith("regtile " + tag)
    V3, g3, _ = _problem(0, 1, 3000, 16, 1, 3000)
    assert _describe(V3.to(DEV), g3.to(DEV)).startswith("regtile-mixed-pipelined 60x50 in=f32 out=f32 mode=0")   # fp32 rows: persistent kernel
    assert _describe(V3.to(DEV).bfloat16(), g3.to(DEV)).startswith("regtile-mixed 60x50 in=bf16 out=bf16")     # bf16 rows: one tile per workgroup
    for n, tag in ((2560, "64x40"), (2400, "60x40"), (3072, "64x48"), (3600, "60x60"), (3840, "64x60")):
        Vn, gn, _ = _problem(0, 1, n, 16, 1, n)
        assert _describe(Vn.to(DEV), gn.to(DEV)).startswith("regtile-mixed-pipelined " + tag)
    for n, tag in ((768, "32x24"), (1536, "48x32"), (1000, "40x25"), (2000, "50x40"),
                   (1280, "40x32"), (64, "8x8"), (128, "16x8"), (196, "14x14"), (384, "24x16"),
                   (640, "32x20"), (960, "32x30"), (1200, "40x30"), (1920, "48x40")):
        Vn, gn, _ = _problem(0, 1, n, 16, 1, n)
        assert _describe(Vn.to(DEV), gn.to(DEV)).startswith("regtile-mixed " + tag)
    Vn, gn, _ = _problem(0, 1, 4096, 24, 2, 4096)    # D % 16 != 0: ragged last tile, general mode of the same kernel
    assert _describe(Vn.to(DEV), gn.to(DEV)).startswith("regtile 64x64 in=f32 out=f32 mode=1 tiles=2")
    Vn, gn, _ = _problem(0, 1, 4000, 32, 2, 4096)    # padded sequence: row predicates, gate still staged in LDS
    assert "mode=3" in _describe(Vn.to(DEV), gn.to(DEV), None, 4096)
    Vn, gn, _ = _problem(0, 1, 8192, 8, 1, 8192)
    assert _describe(Vn.to(DEV), gn.to(DEV)).startswith("regtile-long 64x128 in=f32 out=f32 mode=0 tiles=1")
    for n, tag in ((16384, "64x256"), (12288, "48x256")):
        Vn, gn, _ = _problem(0, 1, n, 4, 1, n)
        assert _describe(Vn.to(DEV), gn.to(DEV)).startswith("regtile-quad " + tag)
    for n, tag in ((6144, "48x128"),):
        Vn, gn, _ = _problem(0, 1, n, 8, 1, n)
        assert _describe(Vn.to(DEV), gn.to(DEV)).startswith("regtile-long " + tag)
    Vn, gn, _ = _problem(0, 1, 768, 16, 1, 768)      # secondary lengths are built for equal storage dtypes only
    assert _describe(Vn.to(DEV).bfloat16(), gn.to(DEV), out_dtype=torch.float32).startswith("stockham")


@pytest.mark.parametrize("algo", ["auto", "stockham"])
@pytest.mark.parametrize("path", golden_files(), ids=golden_ids())
def test_golden_vectors(path, algo):
    d = load_golden(path)
    n = int(d["n_fft"])
    y = _mix(_t(d["V"]), _t(d["gate"]), _t(d.get("mem")), n, algo=algo)
    assert tuple(y.shape) == d["out"].shape
    err = assert_close(y.cpu().numpy(), d["out"], what=f"{path} [{algo}]")
    assert err < 2e-5           # both sides are fp32 results ~1e-6 of RMS away from the truth


SHAPES = [  # (B, N, D, G, n_fft)
    (3, 4096, 64, 4, 4096), (2, 1024, 48, 3, 1024), (3, 256, 32, 2, 256),          # register-tile sizes, square
    (2, 2048, 32, 2, 2048), (2, 512, 32, 4, 512), (3, 2048, 64, 2, 2048), (3, 512, 48, 3, 512),   # 2*RS*RS
    (2, 1500, 32, 2, 2048), (2, 300, 32, 2, 512), (2, 3000, 32, 2, 2048),          # pad / truncate on those
    (2, 3000, 64, 4, 3000), (2, 2000, 32, 2, 3000), (2, 3500, 32, 2, 3000), (2, 3000, 48, 2, 3000),   # mixed-radix register tile (60 x 50)
    (2, 768, 32, 2, 768), (2, 1536, 32, 2, 1536), (2, 3072, 32, 2, 3072), (2, 1000, 32, 2, 1000), (2, 2000, 32, 2, 2000),
    (2, 1280, 32, 2, 1280), (2, 2560, 32, 2, 2560), (2, 3840, 32, 2, 3840),         # the other mixed-radix register-tile lengths
    (2, 700, 32, 2, 768), (2, 1111, 48, 2, 1536), (2, 4000, 32, 2, 3072), (2, 999, 16, 2, 1000), (1, 1999, 48, 2, 2000),
    (2, 1279, 32, 4, 1280), (2, 2000, 48, 2, 2560), (2, 3000, 32, 2, 3840),         # ... with row predicates / narrow groups
    (3, 64, 32, 2, 64), (3, 128, 32, 2, 128), (3, 196, 32, 2, 196), (2, 384, 32, 2, 384), (2, 640, 32, 4, 640), (2, 960, 32, 2, 960),
    (2, 1200, 32, 2, 1200), (2, 1920, 32, 2, 1920), (2, 2400, 32, 2, 2400), (1, 3600, 32, 2, 3600),   # ... incl. radix 7 (196 = 14 x 14)
    (3, 50, 48, 2, 64), (3, 150, 24, 2, 196), (2, 500, 32, 2, 384), (2, 2000, 48, 2, 2400), (1, 4000, 32, 2, 3600),
    (2, 8192, 32, 2, 8192), (2, 5000, 24, 2, 8192), (1, 9000, 12, 2, 8192), (1, 8192, 768, 4, 8192),   # 8192: lane-pair kernel
    (2, 6144, 24, 2, 6144), (2, 6000, 16, 2, 6144), (1, 6144, 768, 4, 6144),   # ... RF = 48
    (1, 16384, 16, 2, 16384), (2, 12288, 8, 2, 12288), (1, 12000, 6, 1, 16384), (1, 12288, 12, 2, 12288),   # lane-quad kernels
    (2, 1500, 32, 4, 1500), (2, 2304, 32, 2, 2304),                                 # Stockham, smooth
    (2, 1000, 32, 2, 1024), (2, 5000, 32, 2, 4096), (1, 100, 16, 2, 128),          # pad / truncate
    (2, 97, 12, 2, 97), (2, 331, 8, 2, 331), (1, 2039, 8, 1, 2039),                # primes: Bluestein
    (2, 60, 6, 2, 60), (2, 64, 10, 2, 64), (2, 256, 24, 8, 256),                   # odd d_g (solo): Stockham
    (2, 4096, 24, 2, 4096), (2, 3000, 40, 2, 3000), (3, 1024, 8, 2, 1024), (2, 256, 100, 2, 256), (2, 200, 36, 6, 196),   # ragged last tile (D % 16 != 0)
    (1, 8192, 8, 2, 8192), (1, 6000, 8, 2, 6000), (1, 16384, 4, 1, 16384), (1, 10000, 4, 2, 10000), (1, 9216, 4, 2, 9216), (1, 5003, 4, 1, 5003), (2, 1, 4, 2, 1), (2, 2, 4, 2, 2), (2, 3, 4, 1, 3),
]


@pytest.mark.parametrize("mem", [False, True], ids=["nomem", "mem"])
@pytest.mark.parametrize("shape", SHAPES, ids=[f"B{s[0]}_N{s[1]}_D{s[2]}_G{s[3]}_fft{s[4]}" for s in SHAPES])
def test_random_vs_fp64_oracle(shape, mem):
    B, N, D, G, n_fft = shape
    V, gate, m = _problem(hash(shape) % 1000, B, N, D, G, n_fft, mem=mem)
    ref = _oracle(V, gate, m, n_fft)
    for algo in ("auto", "stockham"):
        y = _mix(V.to(DEV), gate.to(DEV), None if m is None else m.to(DEV), n_fft, algo=algo)
        err = assert_close(y.cpu().numpy(), ref, what=f"{shape} mem={mem} {algo}")
        assert err < 2e-5


@pytest.mark.parametrize("n_fft", [256, 512, 1024, 2048, 4096, 3000, 97, 1536, 2000, 8192])
@pytest.mark.parametrize("io", ["bf16->bf16", "bf16->f32", "f32->bf16"])
def test_bf16_io(n_fft, io):
    src, dst = io.split("->")
    tin = torch.bfloat16 if src == "bf16" else torch.float32
    tout = torch.bfloat16 if dst == "bf16" else torch.float32
    V, gate, _ = _problem(7, 2, n_fft, 32, 2, n_fft, dtype=tin)
    ref = _oracle(V, gate, None, n_fft)                       # oracle on the bf16-representable values
    y32 = _mix(V.to(DEV), gate.to(DEV), None, n_fft, out_dtype=torch.float32)
    assert_close(y32.cpu().numpy(), ref, what="fp32-output variant of the same kernel")
    y = _mix(V.to(DEV), gate.to(DEV), None, n_fft, out_dtype=tout)
    assert y.dtype == tout
    if tout == torch.bfloat16:
        same_kernel = _describe(V.to(DEV), gate.to(DEV), None, n_fft, out_dtype=torch.float32).split(" in=")[0] == \
            _describe(V.to(DEV), gate.to(DEV), None, n_fft, out_dtype=tout).split(" in=")[0]
        if same_kernel:                                        # (secondary lengths: mixed storage dtypes run on Stockham)
            assert torch.equal(y, y32.bfloat16())             # RNE of the kernel's own fp32 result, bit exact
        yb = y.float().cpu().numpy()
        rb = bf16_round(ref.astype(np.float32))
        ulp = np.maximum(np.abs(rb), 1e-30) * 2.0 ** -7       # 1 bf16 ulp (8-bit significand)
        assert np.all(np.abs(yb - rb) <= ulp + 1e-4 * np.sqrt(np.mean(ref ** 2)))


def test_strided_channel_chunk_views():
    """spectre.py:703 hands each head a channel chunk of a wider tensor: row stride != D."""
    H, d = 3, 32
    Vfull, gate, _ = _problem(11, 2, 1024, H * d, 2 * H, 1024)
    Vd = Vfull.to(DEV)
    outfull = torch.zeros(2, 1024, H * d, device=DEV)
    from fft_amd import spectral_mix
    for h in range(H):
        chunk = Vd[:, :, h * d:(h + 1) * d]
        assert not chunk.is_contiguous()
        g = gate[:, 2 * h:2 * h + 2].to(DEV)
        y = _mix(chunk, g, None, 1024)
        assert_close(y.cpu().numpy(), _oracle(Vfull[:, :, h * d:(h + 1) * d], gate[:, 2 * h:2 * h + 2], None, 1024))
        spectral_mix(chunk, g, None, 1024, out=outfull[:, :, h * d:(h + 1) * d])      # strided output too
    torch.cuda.synchronize()
    # one fused launch over all heads (G_tot = H*G) == per-head launches, bit for bit
    yall = _mix(Vd, gate.to(DEV), None, 1024)
    assert torch.equal(yall, outfull)


def test_unaligned_view_falls_back_to_stockham_and_stays_correct():
    V, gate, _ = _problem(12, 2, 256, 34, 2, 256)
    Vd = V.to(DEV)[:, :, 1:33]                                 # odd channel offset: not pair-aligned
    assert "stockham" in _describe(Vd, gate.to(DEV))
    y = _mix(Vd, gate.to(DEV), None, 256)
    assert_close(y.cpu().numpy(), _oracle(V[:, :, 1:33], gate, None, 256))
    with pytest.raises(ValueError, match="not applicable"):
        _mix(Vd, gate.to(DEV), None, 256, algo="regtile")


def test_error_behaviour():
    from fft_amd import spectral_mix
    V, gate, _ = _problem(13, 2, 64, 8, 2, 64)
    with pytest.raises(RuntimeError, match="HIP device only"):
        spectral_mix(V, gate, None, 64)
    with pytest.raises(ValueError):
        spectral_mix(V.to(DEV), gate[:, :, :-1].to(DEV), None, 64)          # wrong F
    with pytest.raises(ValueError):
        spectral_mix(V.to(DEV), torch.ones(2, 3, 33, dtype=torch.complex64, device=DEV), None, 64)   # D % G
    with pytest.raises(TypeError):
        spectral_mix(V.double().to(DEV), gate.to(DEV), None, 64)            # fp64 unsupported (as in the reference)
    with pytest.raises(NotImplementedError, match="LDS"):
        big = torch.zeros(1, 32768, 2, device=DEV)
        spectral_mix(big, torch.ones(1, 1, 16385, dtype=torch.complex64, device=DEV), None, 32768)
    y = spectral_mix(V[:0].to(DEV), gate[:0].to(DEV), None, 64)             # empty batch
    assert tuple(y.shape) == (0, 64, 8)


def test_runs_on_the_callers_stream():
    V, gate, _ = _problem(14, 2, 1024, 32, 2, 1024)
    ref = _oracle(V, gate, None, 1024)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        Vd, gd = V.to(DEV, non_blocking=True), gate.to(DEV, non_blocking=True)
        from fft_amd import spectral_mix
        y = spectral_mix(Vd, gd, None, 1024)
    s.synchronize()
    assert_close(y.cpu().numpy(), ref)


# ------------------------------------------------------------------------------------------------------
# full benchmark sizes: properties instead of the (too slow) oracle
# ------------------------------------------------------------------------------------------------------
FULL = [(256, 4096, 768, 4, torch.float32), (256, 1024, 768, 4, torch.float32), (256, 3000, 768, 4, torch.float32),
        (256, 4096, 768, 4, torch.bfloat16)]


@pytest.mark.parametrize("B,N,D,G,dt", FULL, ids=[f"B{c[0]}_N{c[1]}_D{c[2]}_{str(c[4])[6:]}" for c in FULL])
def test_full_size_properties(B, N, D, G, dt):
    torch.manual_seed(0)
    F = N // 2 + 1
    V = torch.randn(B, N, D, device=DEV).to(dt)
    rms = float(V.float().pow(2).mean().sqrt())
    tol = lambda a, b, k=1.0: float((a - b).abs().max()) <= k * 2e-4 * max(rms, 1e-6) * 4  # noqa: E731  (abs floor ~ 1e-4*RMS scale)
    # (1) unit gate -> identity (encode -> decode round trip), with junk in Im(DC)/Im(Nyquist) that must be ignored
    ones = torch.ones(B, G, F, dtype=torch.complex64, device=DEV)
    ones[..., 0] += 3j
    if N % 2 == 0:
        ones[..., -1] -= 5j
    y = _mix(V, ones, None, N, out_dtype=torch.float32)
    assert tol(y, V.float())
    del y
    # (2) linearity in V and (3) circular-shift equivariance, random gate with exact zeros
    gate = torch.randn(B, G, F, dtype=torch.complex64, device=DEV) * 0.3
    gate = gate * (torch.rand(B, G, F, device=DEV) >= 0.18)
    y1 = _mix(V, gate, None, N, out_dtype=torch.float32)
    y2 = _mix((V.float() * 2).to(dt), gate, None, N, out_dtype=torch.float32)      # exact scaling by 2 in any dtype
    assert float((y2 - 2 * y1).abs().max()) == 0.0
    ys = _mix(torch.roll(V, 5, dims=1), gate, None, N, out_dtype=torch.float32)
    yrms = float(y1.pow(2).mean().sqrt())
    assert float((ys - torch.roll(y1, 5, dims=1)).abs().max()) <= 1e-4 * 8 * yrms
    # (4) batch-shard / concat equality (the multi-GPU partition), bit for bit
    h = B // 2
    ya = _mix(V[:h], gate[:h], None, N, out_dtype=torch.float32)
    yb = _mix(V[h:], gate[h:], None, N, out_dtype=torch.float32)
    assert torch.equal(torch.cat([ya, yb]), y1)
    # (5) spot-check a few whole columns against the float64 oracle
    idx_b = [0, B // 3, B - 1]
    cols = [0, 1, D // 2 + 1, D - 2, D - 1]
    Vs = V[idx_b][:, :, cols].float().cpu()
    d_g = D // G
    for j, c in enumerate(cols):
        gsel = gate[idx_b][:, c // d_g:c // d_g + 1].cpu()
        ref = spectral_mix_numpy(Vs[:, :, j:j + 1].numpy(), gsel.numpy(), None, N)
        assert_close(y1[idx_b][:, :, c:c + 1].cpu().numpy(), ref, what=f"column {c}")


@pytest.mark.parametrize("n_fft", [4096, 3000, 1024, 300, 97, 8192, 16384])
def test_in_place_is_allowed(n_fft):
    """A workgroup reads every row of its channel tile before it writes any, and tiles are disjoint: out may alias V."""
    from fft_amd import spectral_mix
    V, gate, _ = _problem(21, 2, n_fft, 32, 2, n_fft)
    Vd = V.to(DEV)
    want = _mix(Vd, gate.to(DEV), None, n_fft)
    spectral_mix(Vd, gate.to(DEV), None, n_fft, out=Vd)
    torch.cuda.synchronize()
    assert torch.equal(Vd, want)


@pytest.mark.parametrize("B,N,D,G,n_fft,mem", [(2, 1000, 32, 2, 1024, False), (3, 200, 48, 3, 256, False), (2, 500, 32, 2, 512, True), (2, 2000, 64, 4, 2048, False),
                                               (2, 4000, 40, 5, 4096, False), (2, 4096, 24, 3, 4096, True), (2, 2900, 32, 2, 3000, False), (2, 3000, 40, 5, 3000, False)])
def test_bf16_in_f32_out_outside_the_fast_mode_runs_on_the_register_tile_kernels(B, N, D, G, n_fft, mem):
    """Activations under autocast (bf16 rows in, fp32 rows out) with a padded sequence, a ragged group width or memory_fft: round 2 built
    the differing storage dtypes for the fast mode only and dropped to the 4x slower LDS Stockham path otherwise."""
    V, gate, memory = _problem(31 + n_fft, B, N, D, G, n_fft, dtype=torch.bfloat16, mem=mem)
    md = None if memory is None else memory.to(DEV)
    d = _describe(V.to(DEV), gate.to(DEV), md, n_fft, out_dtype=torch.float32)
    assert d.startswith("regtile") and "in=bf16 out=f32" in d, d
    y = _mix(V.to(DEV), gate.to(DEV), md, n_fft, out_dtype=torch.float32)
    assert y.dtype == torch.float32
    assert_close(y.cpu().numpy(), _oracle(V, gate, memory, n_fft), what=f"bf16 -> f32, {d}")
    # the same arithmetic on the general path
    ys = _mix(V.to(DEV), gate.to(DEV), md, n_fft, out_dtype=torch.float32, algo="stockham")
    assert_close(y.cpu().numpy(), ys.cpu().numpy(), what="register tile vs stockham")


def test_f32_in_bf16_out_outside_the_fast_mode_still_takes_the_general_path():
    """The opposite pairing (fp32 rows in, bf16 rows out) is built for the fast mode only; a padded sequence must still work and say so."""
    V, gate, _ = _problem(33, 2, 1000, 32, 2, 1024)
    d = _describe(V.to(DEV), gate.to(DEV), None, 1024, out_dtype=torch.bfloat16)
    assert d.startswith("stockham") and "storage dtypes differ" in d
    y = _mix(V.to(DEV), gate.to(DEV), None, 1024, out_dtype=torch.bfloat16)
    assert y.dtype == torch.bfloat16
    ref = torch.from_numpy(_oracle(V, gate, None, 1024)).bfloat16().float().numpy()
    assert_close(y.float().cpu().numpy(), ref, rtol=1e-2, atol_rms=1e-2, what="f32 -> bf16, padded")
    with pytest.raises(NotImplementedError, match="storage dtypes differ"):
        _mix(V.to(DEV), gate.to(DEV), None, 1024, out_dtype=torch.bfloat16, algo="regtile")


def test_concurrent_streams():
    """Launches are asynchronous on the caller's stream and share nothing but the read-only plan."""
    from fft_amd import spectral_mix
    V1, g1, _ = _problem(41, 4, 4096, 64, 4, 4096)
    V2, g2, _ = _problem(42, 4, 3000, 64, 4, 3000)
    V1d, g1d, V2d, g2d = V1.to(DEV), g1.to(DEV), V2.to(DEV), g2.to(DEV)
    want1, want2 = _mix(V1d, g1d, None, 4096), _mix(V2d, g2d, None, 3000)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    outs = []
    for _ in range(5):
        with torch.cuda.stream(s1):
            a = spectral_mix(V1d, g1d, None, 4096)
        with torch.cuda.stream(s2):
            b = spectral_mix(V2d, g2d, None, 3000)
        outs.append((a, b))
    torch.cuda.synchronize()
    for a, b in outs:
        assert torch.equal(a, want1) and torch.equal(b, want2)
