"""Randomised parity of the persistent kernels (tools/stress_parity.py): their overlap of requests and arithmetic rests on hand-counted
s_waitcnt values and LDS-only barriers, so a mistake there would show up as a wrong column only now and then.  120 random
(B, N_in, D, G, n_fft, dtype, memory_fft) cases, four launches each (bit-identical), sampled columns against the fp64 oracle."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [11, 12])
def test_random_shapes_repeatable_and_equal_to_oracle(seed):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stress_parity.py"), str(seed), "60"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "STRESS OK" in r.stdout, r.stdout[-3000:]
    assert r.stdout.count("\nok  ") + r.stdout.startswith("ok  ") == 60


def test_plan_destroy_while_launches_are_in_flight_is_safe():
    """Round 2 documented spectre_plan_destroy racing a launch as a use-after-free by contract; plans are now retired, not freed: destroy
    right behind asynchronous launches (the kernels are still reading the twiddle table), keep launching, compare with an undisturbed run."""
    import ctypes
    from fft_amd import _native, spectral_mix
    lib = _native.load()
    dev = torch.device("cuda:0")
    V = torch.randn(48, 4096, 64, device=dev)
    gate = torch.randn(48, 4, 2049, dtype=torch.complex64, device=dev) * 0.3
    want = spectral_mix(V, gate, None, 4096).clone()
    torch.cuda.synchronize()
    outs = []
    for i in range(12):
        outs.append(spectral_mix(V, gate, None, 4096))                    # asynchronous
        rc = lib.spectre_plan_destroy(dev.index or 0, ctypes.c_int64(4096))
        assert rc == 0, lib.spectre_last_error()
        if i % 3 == 0:
            assert lib.spectre_plan_create(dev.index or 0, ctypes.c_int64(4096)) == 0
    torch.cuda.synchronize()
    for y in outs:
        assert torch.equal(y, want)
    assert lib.spectre_plan_destroy(dev.index or 0, ctypes.c_int64(4096)) != 0   # the last iteration left nothing in service: reported, not fatal
    assert lib.spectre_plan_create(dev.index or 0, ctypes.c_int64(4096)) == 0     # back in service (the retired tables, no upload)
    assert lib.spectre_plan_destroy(dev.index or 0, ctypes.c_int64(4096)) == 0
    assert torch.equal(spectral_mix(V, gate, None, 4096), want)                 # and the next call puts it back


def test_retired_plans_can_be_released_behind_a_synchronisation():
    """spectre_plan_destroy only retires (safe against launches in flight), so a process that cycles through many lengths
    grows without bound unless it can hand the tables back: spectre_plans_release_retired, called behind a device synchronisation."""
    import ctypes
    from fft_amd import _native, spectral_mix
    from oracle.spectral_mix_oracle import assert_close, spectral_mix_numpy
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    dev = torch.device("cuda:0")
    lib = _native.load()
    torch.cuda.synchronize()
    lib.spectre_plans_release_retired(-1)                                     # whatever earlier tests retired
    lengths = [4099, 3001, 2053, 1031, 521]                                   # primes: Bluestein plans (four tables each)
    g = torch.Generator().manual_seed(5)
    Vw = torch.randn(1, 4099, 16, generator=g).to(dev)                        # (torch's caching allocator takes its small-block segment now, not later)
    gw = (torch.complex(torch.randn(1, 2, 2050, generator=g), torch.randn(1, 2, 2050, generator=g)) * 0.3).to(dev)
    spectral_mix(Vw, gw, None, 4099)
    torch.cuda.synchronize()
    lib.spectre_plan_destroy(dev.index or 0, ctypes.c_int64(4099))
    assert lib.spectre_plans_release_retired(dev.index or 0) == 1
    free0 = torch.cuda.mem_get_info(dev)[0]
    for n in lengths:
        V = torch.randn(1, n, 16, generator=g)
        gate = torch.complex(torch.randn(1, 2, n // 2 + 1, generator=g), torch.randn(1, 2, n // 2 + 1, generator=g)) * 0.3
        y = spectral_mix(V.to(dev), gate.to(dev), None, n)
        torch.cuda.synchronize()
        assert_close(y.cpu().numpy(), spectral_mix_numpy(V.numpy(), gate.numpy(), None, n), what=f"bluestein {n}")
        assert lib.spectre_plan_destroy(dev.index or 0, ctypes.c_int64(n)) == 0
    assert lib.spectre_plans_release_retired(dev.index or 0) == len(lengths)  # all five handed back
    assert lib.spectre_plans_release_retired(dev.index or 0) == 0             # nothing left
    free1 = torch.cuda.mem_get_info(dev)[0]
    assert free1 >= free0 - (1 << 20), (free0, free1)                         # the tables are gone (torch's own cache aside)
    # and the lengths still work afterwards (plans are rebuilt on demand)
    V = torch.randn(1, 4099, 16, generator=g)
    gate = torch.complex(torch.randn(1, 2, 2050, generator=g), torch.randn(1, 2, 2050, generator=g)) * 0.3
    y = spectral_mix(V.to(dev), gate.to(dev), None, 4099)
    torch.cuda.synchronize()
    assert_close(y.cpu().numpy(), spectral_mix_numpy(V.numpy(), gate.numpy(), None, 4099), what="rebuilt plan")
