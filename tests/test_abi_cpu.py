"""The C-ABI shared library builds for gfx950 without a GPU, loads, and exports exactly the entry points
include/spectre_hip.h declares.  No compute calls here (no GPU in this tier)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "spectre_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(spectre_[a-z_]+)\s*\(", text)))


def test_header_declares_expected_entry_points():
    names = _declared_functions()
    for n in ("spectre_mix_fwd", "spectre_plan_create", "spectre_plan_destroy", "spectre_last_error", "spectre_version",
              "spectre_mix_describe", "spectre_mix_time"):
        assert n in names


def test_library_exports_every_declared_symbol(built_library):
    import torch  # noqa: F401  (HIP runtime first)
    lib = ctypes.CDLL(built_library)
    for n in _declared_functions():
        assert hasattr(lib, n), f"{n} declared in include/spectre_hip.h but not exported"


def test_binding_matches_header(built_library):
    from fft_amd import _native
    assert sorted(_native.EXPORTS) == _declared_functions()
    lib = _native.load()
    assert lib.spectre_version() == _native.ABI_VERSION
    hdr = open(os.path.join(ROOT, "include", "spectre_hip.h")).read()
    assert f"#define SPECTRE_ABI_VERSION {_native.ABI_VERSION}" in hdr
    # struct layout: 4 pointers + 9 int64 + 4 int32 + 1 pointer
    assert ctypes.sizeof(_native.SpectreMixArgs) == 4 * 8 + 9 * 8 + 4 * 4 + 8


def test_every_abi_struct_has_the_layout_the_c_compiler_gives_the_header(tmp_path):
    """sizeof and every field offset of the ten argument structs: ctypes mirror (fft_amd/_native.py) vs gcc on include/spectre_hip.h."""
    import subprocess
    from fft_amd import _native
    structs = ["SpectreMixArgs", "SpectreMixBwdArgs", "SpectreGateArgs", "SpectreGateBwdArgs", "SpectreRfftArgs", "SpectreDecodeArgs", "SpectreDecodeHeadArgs",
               "SpectreProbeArgs", "SpectreWaveletArgs", "SpectreWaveletGradArgs"]
    lines = ["#include <stdio.h>", "#include <stddef.h>", '#include "spectre_hip.h"', "int main(void) {"]
    for sname in structs:
        ct = getattr(_native, sname)
        lines.append(f'  printf("{sname} %zu\\n", sizeof({sname}));')
        for fname, _ in ct._fields_:
            lines.append(f'  printf("{sname}.{fname} %zu\\n", offsetof({sname}, {fname}));')
    lines += ["  return 0;", "}"]
    src = tmp_path / "abi.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "abi"
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = dict(l.split() for l in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.splitlines())
    for sname in structs:
        ct = getattr(_native, sname)
        assert ctypes.sizeof(ct) == int(out[sname]), sname
        for fname, _ in ct._fields_:
            assert getattr(ct, fname).offset == int(out[f"{sname}.{fname}"]), f"{sname}.{fname}"


def test_invalid_arguments_fail_loudly_without_a_gpu(built_library):
    from fft_amd import _native
    lib = _native.load()
    assert lib.spectre_mix_fwd(None) == 1                        # SPECTRE_E_INVALID
    assert b"NULL" in lib.spectre_last_error()
    assert lib.spectre_plan_create(0, -4) == 1
    a = _native.SpectreMixArgs()                                 # all-zero args: never reaches a kernel
    assert lib.spectre_mix_fwd(ctypes.byref(a)) != 0
    assert len(lib.spectre_last_error()) > 0
    assert lib.spectre_wavelet_refine(None) == 1 and b"spectre_wavelet_refine" in lib.spectre_last_error()
    w = _native.SpectreWaveletArgs()
    w.B, w.N, w.D = 1, 48, 8                                     # not a power of two: refused before anything touches a device
    assert lib.spectre_wavelet_refine(ctypes.byref(w)) == 2 and b"power-of-two" in lib.spectre_last_error()
    w.N = 65536
    assert lib.spectre_wavelet_refine(ctypes.byref(w)) == 2 and b"too long" in lib.spectre_last_error()
    assert lib.spectre_wavelet_gate_grad(None) == 1


def test_no_cpu_fallback_in_product_path():
    """CPU tensors must raise; the product package must not import the oracle."""
    import torch
    from fft_amd import spectral_mix
    V = torch.randn(1, 16, 4)
    g = torch.ones(1, 2, 9, dtype=torch.complex64)
    with pytest.raises(RuntimeError, match="HIP device only"):
        spectral_mix(V, g, None, 16)
    import ast
    for root, _, files in os.walk(os.path.join(ROOT, "fft_amd")):
        for f in files:
            path = os.path.join(root, f)
            if f.endswith((".hip", ".h")):
                src = open(path).read()
                assert "hipfft" not in src.lower() and "rocfft" not in src.lower(), f
            if not f.endswith(".py"):
                continue
            tree = ast.parse(open(path).read())
            for node in ast.walk(tree):
                if isinstance(node, (ast.Import, ast.ImportFrom)):
                    names = [a.name for a in node.names] + [getattr(node, "module", None) or ""]
                    assert not any(n.split(".")[0] == "oracle" for n in names), f"{f} imports the oracle"
                if isinstance(node, ast.Attribute) and node.attr == "fft" and isinstance(node.value, ast.Name) and node.value.id == "torch":
                    raise AssertionError(f"{f} calls torch.fft (line {node.lineno}): the product path must be the HIP kernel")
