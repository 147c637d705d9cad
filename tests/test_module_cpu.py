"""Host logic of the drop-in module, CPU only: it loads the reference's state_dict unchanged, its gate
producer reproduces the tensors the reference fed to the hot path (captured in the fixtures), and the
forward refuses to run without a HIP device."""
import inspect

import numpy as np
import pytest
import torch

from conftest import golden_files, golden_ids, load_golden
from fft_amd import SpectreHead, batch_shard

MODULE_CASES = [(p, i) for p, i in zip(golden_files(), golden_ids()) if any(k.startswith("sd/") for k in load_golden(p))]


def _build(d):
    sd = {k[3:]: torch.from_numpy(v) for k, v in d.items() if k.startswith("sd/")}
    dim = sd["W_v.weight"].shape[0]
    pooling = "attention" if any(k.startswith("pooling.") for k in sd) else "mean"
    kw = {"use_toeplitz": True, "toeplitz_bw": int(d["toeplitz_bw"])} if "toeplitz_kernel" in sd else {}      # fixtures g14_*
    head = SpectreHead(dim, int(d["n_fft"]), num_groups=int(d["G"]), pooling_type=pooling, **kw).eval()
    missing, unexpected = head.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    return head


def test_constructor_surface_matches_reference():
    sig = inspect.signature(SpectreHead.__init__)
    names = list(sig.parameters)
    assert names == ["self", "embed_dim", "fft_size", "num_groups", "num_buckets", "d_gate", "use_toeplitz",
                     "toeplitz_bw", "dropout_p", "pooling_type"]                      # spectre.py:404-416
    assert sig.parameters["num_groups"].kind is inspect.Parameter.KEYWORD_ONLY
    assert sig.parameters["pooling_type"].default == "dct" and sig.parameters["d_gate"].default == 256
    f = inspect.signature(SpectreHead.forward)
    assert list(f.parameters) == ["self", "x", "pos_phase", "return_q_pool", "memory_fft"]   # spectre.py:479-485
    h = SpectreHead(64, 256, num_groups=4, pooling_type="mean")
    assert (h.d, h.n_fft, h.G, h.d_g, h.F_half, h.B) == (64, 256, 4, 16, 129, 11)
    assert sorted(h.state_dict()) == sorted(["W_q.weight", "W_v.weight", "gate_mlp.0.weight", "gate_mlp.0.bias",
                                             "gate_mlp.2.weight", "gate_mlp.2.bias", "q_norm.weight", "q_norm.bias",
                                             "modrelu.bias", "modrelu.eps"])
    with pytest.raises(AssertionError):
        SpectreHead(10, 16, num_groups=4)                                             # spectre.py:422


@pytest.mark.parametrize("path,cid", MODULE_CASES, ids=[c[1] for c in MODULE_CASES])
def test_gate_producer_reproduces_reference_tensors(path, cid):
    d = load_golden(path)
    head = _build(d)
    x = torch.from_numpy(d["x"])
    pp = torch.from_numpy(d["pos_phase"]) if "pos_phase" in d else None
    with torch.no_grad():
        V, gate, q_pool = head.spectral_gate(x, pp)
    assert torch.equal(V, torch.from_numpy(d["V"]))                      # same GEMM, same bits
    g_ref = torch.from_numpy(d["gate"])
    assert gate.shape == g_ref.shape and gate.dtype == torch.complex64
    assert torch.allclose(torch.view_as_real(gate), torch.view_as_real(g_ref), rtol=1e-6, atol=1e-7)
    assert q_pool.shape == (x.shape[0], x.shape[2])


def test_forward_requires_hip_device():
    head = SpectreHead(8, 16, num_groups=2, pooling_type="mean").eval()
    with torch.no_grad(), pytest.raises(RuntimeError, match="HIP device only"):
        head(torch.randn(2, 16, 8))


def test_assert_on_wrong_width_and_toeplitz():
    head = SpectreHead(8, 16, num_groups=2, pooling_type="mean")
    with pytest.raises(AssertionError):
        head.spectral_gate(torch.randn(2, 16, 6))                         # spectre.py:499
    assert head.toeplitz_kernel is None and "toeplitz_kernel" not in head.state_dict()
    # the option the reference's constructor cannot build (KeyError at :457): here it constructs what :464-474 intends
    t = SpectreHead(8, 16, num_groups=2, pooling_type="mean", use_toeplitz=True, toeplitz_bw=3)
    assert isinstance(t.toeplitz_kernel, torch.nn.Parameter) and t.toeplitz_kernel.shape == (7,) and t.toeplitz_kernel.dtype == torch.complex64
    assert "toeplitz_kernel" in t.state_dict() and t.toeplitz_bw == 3


def test_complex_conv1d_is_the_circular_correlation():
    """fft_amd.complex_conv1d (one 2-in / 2-out real conv1d on padded planes) against the definition the reference's four conv1ds implement
    (spectre.py:334-395): out[l] = sum_t kernel[t] * x[(l + t - padding) mod L]; linear in both arguments, differentiable."""
    from fft_amd import complex_conv1d
    g = torch.Generator().manual_seed(0)
    for shape, bw in (((2, 3, 9), 4), ((3, 17), 2), ((2, 2, 5), 2), ((1, 1, 33), 1)):
        x = torch.complex(torch.randn(*shape, generator=g), torch.randn(*shape, generator=g))
        k = torch.complex(torch.randn(2 * bw + 1, generator=g), torch.randn(2 * bw + 1, generator=g))
        L = shape[-1]
        want = torch.zeros_like(x)
        for l in range(L):
            for t in range(2 * bw + 1):
                want[..., l] += k[t] * x[..., (l + t - bw) % L]
        got = complex_conv1d(x, k, bw)
        assert got.shape == x.shape and torch.allclose(torch.view_as_real(got), torch.view_as_real(want), rtol=1e-5, atol=1e-5)
    k = k.clone().requires_grad_(True)
    complex_conv1d(x, k, bw).abs().sum().backward()
    assert k.grad is not None and k.grad.abs().sum() > 0


def test_dct_pooling_falls_back_like_the_reference():
    head = SpectreHead(8, 16, num_groups=2)                               # default pooling_type="dct"
    x = torch.randn(2, 16, 8)
    with pytest.warns(UserWarning, match="DCT pooling unavailable"):
        _, _, qp = head.spectral_gate(x)
    assert qp.shape == (2, 8)


def test_batch_shard_partitions():
    for B in (0, 1, 7, 256, 2048):
        for w in (1, 2, 3, 8):
            spans = [batch_shard(B, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [e - s for s, e in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        batch_shard(4, 2, 2)


def test_reference_names_of_the_host_helpers():
    """`interp_complex_1d(x, size, mode="linear")` (spectre.py:26-30) is importable under the reference's name with its default mode; endpoints
    are aligned in the cubic mode (align_corners=True), so with one group the first and last anchors come back unchanged."""
    import inspect
    from fft_amd import interp_complex_1d, resample_complex
    assert list(inspect.signature(interp_complex_1d).parameters) == ["x", "size", "mode"]
    assert inspect.signature(interp_complex_1d).parameters["mode"].default == "linear"
    x = torch.randn(2, 3, 9, dtype=torch.complex64, generator=torch.Generator().manual_seed(0))
    for mode in ("cubic", "linear", "nearest"):
        assert torch.equal(interp_complex_1d(x, 33, mode), resample_complex(x, 33, mode))
    x1 = x[:, :1]                                                    # one group: with G > 1 the reference's stack(dim=1).reshape(B*G, 2, 1, K) deals the
    y = interp_complex_1d(x1, 33, "cubic")                           # real / imaginary planes across the groups (spectre.py:42) — reproduced, not corrected
    assert torch.allclose(y[..., 0], x1[..., 0], atol=1e-6) and torch.allclose(y[..., -1], x1[..., -1], atol=1e-6)
    assert torch.equal(interp_complex_1d(x, 33), resample_complex(x, 33, "linear"))
