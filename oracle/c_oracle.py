"""ctypes binding of oracle/spectral_mix_ref.c — TEST INFRASTRUCTURE ONLY."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libspectral_mix_ref.so")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "spectral_mix_ref.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


def spectral_mix_c(V, gate, mem, n_fft: int):
    """Same contract as spectral_mix_oracle.spectral_mix_*; float32 in/out, float64 inside."""
    lib = ctypes.CDLL(build())
    i64, fp = ctypes.c_int64, ctypes.POINTER(ctypes.c_float)
    lib.spectral_mix_ref.argtypes = [fp, fp, fp, fp] + [i64] * 9
    lib.spectral_mix_ref.restype = ctypes.c_int
    V = np.ascontiguousarray(V, dtype=np.float32)
    gate = np.ascontiguousarray(gate, dtype=np.complex64)
    B, N, D = V.shape
    G = gate.shape[1]
    n_out = min(N, n_fft)
    out = np.empty((B, n_out, D), dtype=np.float32)
    memp = None
    if mem is not None:
        mem = np.ascontiguousarray(mem, dtype=np.complex64)
        memp = mem.view(np.float32).ctypes.data_as(fp)
    rc = lib.spectral_mix_ref(V.ctypes.data_as(fp), gate.view(np.float32).ctypes.data_as(fp), memp,
                              out.ctypes.data_as(fp), B, N, n_fft, D, G, N * D, D, n_out * D, D)
    if rc != 0:
        raise RuntimeError(f"spectral_mix_ref failed rc={rc}")
    return out
