"""Randomised parity of the persistent kernels (tools/stress_parity.py): their overlap of requests and arithmetic rests on hand-counted
s_waitcnt values and LDS-only barriers, so a mistake there would show up as a wrong column only now and then.  120 random
(B, N_in, D, G, n_fft, dtype, memory_fft) cases, four launches each (bit-identical), sampled columns against the fp64 oracle."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [11, 12])
def test_random_shapes_repeatable_and_equal_to_oracle(seed):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stress_parity.py"), str(seed), "60"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "STRESS OK" in r.stdout, r.stdout[-3000:]
    assert r.stdout.count("\nok  ") + r.stdout.startswith("ok  ") == 60
