"""Backward of the hot path: the oracle's closed form and its autograd restatement
against the gradients the REFERENCE's autograd produced (fixtures g9_*).  CPU only."""
import glob
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR, load_golden
from oracle.spectral_mix_oracle import assert_close, spectral_mix_backward_numpy, spectral_mix_backward_torch

BWD = sorted(glob.glob(os.path.join(GOLDEN_DIR, "g9_*.npz")))
IDS = [os.path.basename(f)[:-4] for f in BWD]


def test_backward_fixture_inventory():
    assert len(BWD) >= 6 and any("pad" in i for i in IDS) and any("trunc" in i for i in IDS)


@pytest.mark.parametrize("path", BWD, ids=IDS)
def test_autograd_restatement_is_bit_exact(path):
    d = load_golden(path)
    dV, dg = spectral_mix_backward_torch(torch.from_numpy(d["V"]), torch.from_numpy(d["gate"]), torch.from_numpy(d["dout"]), int(d["n_fft"]))
    assert np.array_equal(dV.numpy(), d["dV"]) and np.array_equal(dg.numpy(), d["dgate"])


@pytest.mark.parametrize("path", BWD, ids=IDS)
def test_closed_form_matches_reference_autograd(path):
    d = load_golden(path)
    dV, dg = spectral_mix_backward_numpy(d["V"], d["gate"], d["dout"], int(d["n_fft"]))
    assert dV.shape == d["dV"].shape and dg.shape == d["dgate"].shape
    assert assert_close(d["dV"], dV, what="dV") < 5e-6
    assert assert_close(d["dgate"].view(np.float32), dg.astype(np.complex64).view(np.float32), what="dgate") < 5e-6
    N, n = d["V"].shape[1], int(d["n_fft"])
    if N > n:
        assert not d["dV"][:, n:].any()          # rows rfft truncated get exactly zero gradient
