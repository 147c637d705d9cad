"""The experiment harness of the 4096 kernel (tools/p64v.h, a fork of fft_amd/csrc/kernel_regtile64p.h with a template switch per experiment)
must stay the product: its copy of the SHIPPED instantiation has to produce the library kernel's output bit for bit on the headline shape.
Otherwise an A/B result measured in the harness says nothing about the library.  The binary is built by
`__graft_entry__.build()` (in-tree, travels with the snapshot)."""
import os
import re
import subprocess

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_harness_copy_of_the_shipped_kernel_equals_the_library():
    exe = os.path.join(ROOT, "tools", "p64v_bench")
    if not os.path.exists(exe):
        import __graft_entry__ as g
        g.build_harness()
    out = subprocess.run([exe, "1", "must equal"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    m = re.search(r"check harness copy of the shipped kernel \(must equal the library\)\s+max \|diff\| vs library (\S+), "
                  r"elements off by > 1e-4: (\d+), elements with different bits: (\d+)", out.stdout)
    assert m, out.stdout[-2000:]
    assert float(m.group(1).rstrip(",")) == 0.0 and int(m.group(2)) == 0 and int(m.group(3)) == 0, m.group(0)
