"""The wavelet refinement's oracle is only trustworthy once it reproduces what the REFERENCE produced: fixtures g13_wavelet_* hold inputs, the
coin flips and the outputs / autograd results of the reference's own WaveletRefinement (tests/golden/make_golden.py `case_wavelet`).  CPU only."""
import glob
import inspect
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR, load_golden
from oracle.spectral_mix_oracle import assert_close
from oracle.wavelet_oracle import haar_round_trip, wavelet_gate_grad_numpy, wavelet_refinement_numpy

WV = sorted(glob.glob(os.path.join(GOLDEN_DIR, "g13_wavelet_*.npz")))
IDS = [os.path.basename(p)[:-4] for p in WV]


def test_inventory():
    assert len(WV) >= 5
    assert any(load_golden(p)["mask"].all() for p in WV) and any(not load_golden(p)["mask"].all() for p in WV)


@pytest.mark.parametrize("path", WV, ids=IDS)
def test_oracle_reproduces_the_reference_forward(path):
    d = load_golden(path)
    y = wavelet_refinement_numpy(d["v"], d["gate"], d["mask"])
    assert_close(d["out"], y, rtol=1e-5, atol_rms=1e-6, what="reference fp32 vs numpy fp64")
    off = ~d["mask"]
    assert np.array_equal(d["out"][off], d["v"][off])                 # switched-off elements pass through bit for bit (v + 0)


@pytest.mark.parametrize("path", WV, ids=IDS)
def test_oracle_reproduces_the_reference_gate_gradient(path):
    """d/d(gate) chained through the fixture's own gate_mlp (Linear, SiLU, Linear, Sigmoid — rebuilt from the state_dict with torch) must give
    the reference's d/dq_pool and parameter gradients; d/dv is the identity."""
    d = load_golden(path)
    assert np.array_equal(d["grad_v"], d["dout"])
    dgate = wavelet_gate_grad_numpy(d["v"], d["dout"], d["mask"])
    dim = d["v"].shape[2]
    mlp = torch.nn.Sequential(torch.nn.Linear(dim, dim), torch.nn.SiLU(), torch.nn.Linear(dim, dim), torch.nn.Sigmoid()).double()
    mlp.load_state_dict({k[len("sd/gate_mlp."):]: torch.from_numpy(v).double() for k, v in d.items() if k.startswith("sd/gate_mlp.")})
    q = torch.from_numpy(d["q_pool"]).double().requires_grad_(True)
    (mlp(q) * torch.from_numpy(dgate)).sum().backward()
    assert_close(d["grad_q_pool"], q.grad.numpy(), rtol=1e-4, atol_rms=1e-5, what="d/dq_pool")
    for name, prm in mlp.named_parameters():
        assert_close(d["grad/gate_mlp." + name], prm.grad.numpy(), rtol=1e-4, atol_rms=1e-5, what="d/d gate_mlp." + name)


def test_round_trip_properties():
    """What the kernel's in-place scheme relies on: one level is the shift-swap 0..7 -> 0,7,2,1,4,3,6,5 plus a term that only depends on the
    approximation band; lengths 1 and 2 are the identity; R is linear and preserves the mean; non-powers of two raise."""
    rng = np.random.default_rng(0)
    assert np.allclose(haar_round_trip(np.arange(8.0)), [0, 7, 6, 5, 4, 3, 2, 1])
    for n in (1, 2):
        x = rng.standard_normal((3, n))
        assert np.allclose(haar_round_trip(x), x)
    x, y = rng.standard_normal((2, 5, 64))
    assert np.allclose(haar_round_trip(2.0 * x - 3.0 * y), 2.0 * haar_round_trip(x) - 3.0 * haar_round_trip(y))
    assert np.allclose(haar_round_trip(x).mean(-1), x.mean(-1))
    assert not np.allclose(haar_round_trip(x), x)                      # NOT perfect reconstruction
    for n in (3, 6, 12, 3000):
        with pytest.raises(ValueError):
            haar_round_trip(np.zeros(n))


def test_module_surface_and_reference_state_dict():
    from fft_amd import SpectreBlock, SpectreMultiHead, WaveletRefinement
    assert list(inspect.signature(WaveletRefinement.__init__).parameters) == ["self", "embed_dim", "on_rate"]       # spectre.py:824
    assert list(inspect.signature(WaveletRefinement.forward).parameters)[:3] == ["self", "v", "q_pool"]
    assert inspect.signature(WaveletRefinement.__init__).parameters["on_rate"].default == 0.1
    assert inspect.signature(SpectreMultiHead.__init__).parameters["wavelet_on_rate"].default == 0.1                # spectre.py:675
    assert inspect.signature(SpectreBlock.__init__).parameters["wavelet_on_rate"].default == 0.1                    # spectre.py:921
    for p in WV:
        d = load_golden(p)
        wr = WaveletRefinement(d["v"].shape[2], on_rate=float(d["on_rate"]))
        missing, unexpected = wr.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in d.items() if k.startswith("sd/")}, strict=True)
        assert not missing and not unexpected
    wr = WaveletRefinement(8, on_rate=0.0)
    v = torch.randn(2, 12, 8)
    assert wr(v, torch.randn(2, 8)) is v                               # never on: the early exit, whatever the length (spectre.py:845-846)
    with pytest.raises(RuntimeError):                                  # on a CPU tensor the launch refuses (no CPU path)
        WaveletRefinement(8, on_rate=1.0)(torch.randn(2, 16, 8), torch.randn(2, 8))
    with pytest.raises(ValueError, match="power of two"):
        WaveletRefinement(8, on_rate=0.5)(v, torch.randn(2, 8))
