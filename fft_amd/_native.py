"""ctypes binding of libspectre_hip.so (C ABI: include/spectre_hip.h).

There is no CPU or PyTorch fallback behind this module: if the shared object is missing or does not export
the ABI the header declares, importing `fft_amd.functional` works but every call raises.
"""
from __future__ import annotations

import ctypes
import os
import threading

import torch  # noqa: F401  — must be imported first: the library binds to the HIP runtime torch already loaded

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SPECTRE_HIP_LIB") or os.path.join(_HERE, "lib", "libspectre_hip.so")   # env: A/B builds

ABI_VERSION = 10
F32, BF16 = 0, 1
ALGO = {"auto": 0, "stockham": 1, "regtile": 2}

# every symbol include/spectre_hip.h declares
EXPORTS = ("spectre_version", "spectre_last_error", "spectre_mix_fwd", "spectre_mix_describe",
           "spectre_plan_create", "spectre_plan_destroy", "spectre_mix_time", "spectre_mix_bwd",
           "spectre_mix_bwd_workspace_bytes", "spectre_gate_fwd", "spectre_gate_bwd", "spectre_rfft_fwd", "spectre_decode_workspace_bytes",
           "spectre_decode_step", "spectre_decode_head_workspace_bytes", "spectre_decode_head_step", "spectre_probe_copy", "spectre_plans_release_retired",
           "spectre_plan_set_tile_order", "spectre_plan_get_tile_order", "spectre_wavelet_refine", "spectre_wavelet_gate_grad")


class SpectreMixArgs(ctypes.Structure):
    _fields_ = [
        ("v", ctypes.c_void_p), ("gate", ctypes.c_void_p), ("mem", ctypes.c_void_p), ("out", ctypes.c_void_p),
        ("B", ctypes.c_int64), ("N_in", ctypes.c_int64), ("n_fft", ctypes.c_int64), ("D", ctypes.c_int64),
        ("G_tot", ctypes.c_int64),
        ("v_sb", ctypes.c_int64), ("v_sn", ctypes.c_int64), ("out_sb", ctypes.c_int64), ("out_sn", ctypes.c_int64),
        ("in_dtype", ctypes.c_int32), ("out_dtype", ctypes.c_int32), ("algo", ctypes.c_int32), ("device", ctypes.c_int32),
        ("stream", ctypes.c_void_p),
    ]


class SpectreGateArgs(ctypes.Structure):
    _fields_ = [
        ("anchors", ctypes.c_void_p), ("bias", ctypes.c_void_p), ("phase", ctypes.c_void_p), ("gate", ctypes.c_void_p),
        ("B", ctypes.c_int64), ("G", ctypes.c_int64), ("K", ctypes.c_int64), ("F", ctypes.c_int64),
        ("phase_sb", ctypes.c_int64), ("eps", ctypes.c_float), ("device", ctypes.c_int32), ("stream", ctypes.c_void_p),
    ]


class SpectreGateBwdArgs(ctypes.Structure):
    _fields_ = [
        ("anchors", ctypes.c_void_p), ("bias", ctypes.c_void_p), ("phase", ctypes.c_void_p), ("dgate", ctypes.c_void_p),
        ("workspace", ctypes.c_void_p), ("danchors", ctypes.c_void_p), ("dbias", ctypes.c_void_p), ("dphase", ctypes.c_void_p),
        ("B", ctypes.c_int64), ("G", ctypes.c_int64), ("K", ctypes.c_int64), ("F", ctypes.c_int64),
        ("phase_sb", ctypes.c_int64), ("eps", ctypes.c_float), ("device", ctypes.c_int32), ("stream", ctypes.c_void_p),
    ]


class SpectreWaveletArgs(ctypes.Structure):
    _fields_ = [
        ("v", ctypes.c_void_p), ("out", ctypes.c_void_p), ("vref", ctypes.c_void_p), ("mask", ctypes.c_void_p), ("gate", ctypes.c_void_p),
        ("B", ctypes.c_int64), ("N", ctypes.c_int64), ("D", ctypes.c_int64),
        ("v_sb", ctypes.c_int64), ("v_sn", ctypes.c_int64), ("out_sb", ctypes.c_int64), ("out_sn", ctypes.c_int64),
        ("ref_sb", ctypes.c_int64), ("ref_sn", ctypes.c_int64),
        ("dtype", ctypes.c_int32), ("device", ctypes.c_int32), ("stream", ctypes.c_void_p),
    ]


class SpectreWaveletGradArgs(ctypes.Structure):
    _fields_ = [
        ("dout", ctypes.c_void_p), ("vref", ctypes.c_void_p), ("mask", ctypes.c_void_p), ("dgate", ctypes.c_void_p),
        ("B", ctypes.c_int64), ("N", ctypes.c_int64), ("D", ctypes.c_int64),
        ("d_sb", ctypes.c_int64), ("d_sn", ctypes.c_int64), ("ref_sb", ctypes.c_int64), ("ref_sn", ctypes.c_int64),
        ("dtype", ctypes.c_int32), ("device", ctypes.c_int32), ("stream", ctypes.c_void_p),
    ]


class SpectreRfftArgs(ctypes.Structure):
    _fields_ = [
        ("v", ctypes.c_void_p), ("spec", ctypes.c_void_p),
        ("B", ctypes.c_int64), ("N_in", ctypes.c_int64), ("n_fft", ctypes.c_int64), ("D", ctypes.c_int64),
        ("v_sb", ctypes.c_int64), ("v_sn", ctypes.c_int64),
        ("in_dtype", ctypes.c_int32), ("device", ctypes.c_int32), ("stream", ctypes.c_void_p),
    ]


class SpectreDecodeArgs(ctypes.Structure):
    _fields_ = [
        ("prefix", ctypes.c_void_p), ("v_old", ctypes.c_void_p), ("v_new", ctypes.c_void_p), ("gate", ctypes.c_void_p),
        ("out", ctypes.c_void_p), ("workspace", ctypes.c_void_p),
        ("n_fft", ctypes.c_int64), ("d", ctypes.c_int64), ("G", ctypes.c_int64), ("t", ctypes.c_int64),
        ("device", ctypes.c_int32), ("stream", ctypes.c_void_p),
    ]


class SpectreDecodeHeadArgs(ctypes.Structure):
    _fields_ = [
        ("prefix", ctypes.c_void_p), ("V_buf", ctypes.c_void_p), ("Q_buf", ctypes.c_void_p), ("sum_q", ctypes.c_void_p),
        ("q_t", ctypes.c_void_p), ("v_t", ctypes.c_void_p), ("out", ctypes.c_void_p), ("workspace", ctypes.c_void_p),
        ("ln_w", ctypes.c_void_p), ("ln_b", ctypes.c_void_p), ("w1", ctypes.c_void_p), ("b1", ctypes.c_void_p),
        ("w2", ctypes.c_void_p), ("b2", ctypes.c_void_p), ("modrelu_bias", ctypes.c_void_p),
        ("ln_eps", ctypes.c_float), ("modrelu_eps", ctypes.c_float),
        ("n_fft", ctypes.c_int64), ("d", ctypes.c_int64), ("G", ctypes.c_int64), ("K", ctypes.c_int64), ("h1", ctypes.c_int64),
        ("t", ctypes.c_int64), ("device", ctypes.c_int32), ("stream", ctypes.c_void_p),
    ]


class SpectreMixBwdArgs(ctypes.Structure):
    _fields_ = [
        ("v", ctypes.c_void_p), ("gate", ctypes.c_void_p), ("dout", ctypes.c_void_p), ("dv", ctypes.c_void_p),
        ("dgate", ctypes.c_void_p), ("workspace", ctypes.c_void_p),
        ("B", ctypes.c_int64), ("N_in", ctypes.c_int64), ("n_fft", ctypes.c_int64), ("D", ctypes.c_int64),
        ("G_tot", ctypes.c_int64),
        ("v_sb", ctypes.c_int64), ("v_sn", ctypes.c_int64), ("dout_sb", ctypes.c_int64), ("dout_sn", ctypes.c_int64),
        ("dv_sb", ctypes.c_int64), ("dv_sn", ctypes.c_int64), ("workspace_bytes", ctypes.c_int64),
        ("io_dtype", ctypes.c_int32), ("device", ctypes.c_int32), ("stream", ctypes.c_void_p),
    ]


class SpectreProbeArgs(ctypes.Structure):
    _fields_ = [
        ("src", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("rows", ctypes.c_int64), ("row_bytes", ctypes.c_int64),
        ("seg_bytes", ctypes.c_int32), ("tile_rows", ctypes.c_int32), ("mode", ctypes.c_int32), ("wgs_per_cu", ctypes.c_int32),
        ("device", ctypes.c_int32), ("stream", ctypes.c_void_p),
    ]


_lib = None
_lock = threading.Lock()


class NativeLibraryError(RuntimeError):
    pass


def load():
    """Load (once) and return the shared library; raises NativeLibraryError if it cannot be used."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise NativeLibraryError(
                f"{LIB_PATH} not found — build it with `python -m fft_amd.build` (hipcc, gfx950). "
                "There is no fallback implementation.")
        lib = ctypes.CDLL(LIB_PATH)
        missing = [s for s in EXPORTS if not hasattr(lib, s)]
        if missing:
            raise NativeLibraryError(f"{LIB_PATH} lacks symbols {missing}")
        lib.spectre_version.restype = ctypes.c_int
        lib.spectre_last_error.restype = ctypes.c_char_p
        lib.spectre_mix_fwd.argtypes = [ctypes.POINTER(SpectreMixArgs)]
        lib.spectre_mix_fwd.restype = ctypes.c_int
        lib.spectre_mix_describe.argtypes = [ctypes.POINTER(SpectreMixArgs), ctypes.c_char_p, ctypes.c_size_t]
        lib.spectre_mix_describe.restype = ctypes.c_int
        lib.spectre_plan_create.argtypes = [ctypes.c_int, ctypes.c_int64]
        lib.spectre_plan_create.restype = ctypes.c_int
        lib.spectre_plan_destroy.argtypes = [ctypes.c_int, ctypes.c_int64]
        lib.spectre_plan_destroy.restype = ctypes.c_int
        lib.spectre_plans_release_retired.argtypes = [ctypes.c_int]
        lib.spectre_plans_release_retired.restype = ctypes.c_int
        lib.spectre_plan_set_tile_order.argtypes = [ctypes.c_int, ctypes.c_int64, ctypes.c_int]
        lib.spectre_plan_set_tile_order.restype = ctypes.c_int
        lib.spectre_plan_get_tile_order.argtypes = [ctypes.c_int, ctypes.c_int64, ctypes.POINTER(ctypes.c_int)]
        lib.spectre_plan_get_tile_order.restype = ctypes.c_int
        lib.spectre_mix_time.argtypes = [ctypes.POINTER(SpectreMixArgs), ctypes.c_int, ctypes.c_int,
                                         ctypes.POINTER(ctypes.c_float)]
        lib.spectre_mix_time.restype = ctypes.c_int
        lib.spectre_mix_bwd.argtypes = [ctypes.POINTER(SpectreMixBwdArgs)]
        lib.spectre_mix_bwd.restype = ctypes.c_int
        lib.spectre_mix_bwd_workspace_bytes.argtypes = [ctypes.c_int64] * 4
        lib.spectre_mix_bwd_workspace_bytes.restype = ctypes.c_int64
        lib.spectre_gate_fwd.argtypes = [ctypes.POINTER(SpectreGateArgs)]
        lib.spectre_gate_fwd.restype = ctypes.c_int
        lib.spectre_gate_bwd.argtypes = [ctypes.POINTER(SpectreGateBwdArgs)]
        lib.spectre_gate_bwd.restype = ctypes.c_int
        lib.spectre_rfft_fwd.argtypes = [ctypes.POINTER(SpectreRfftArgs)]
        lib.spectre_rfft_fwd.restype = ctypes.c_int
        lib.spectre_decode_workspace_bytes.argtypes = [ctypes.c_int64] * 2
        lib.spectre_decode_workspace_bytes.restype = ctypes.c_int64
        lib.spectre_decode_step.argtypes = [ctypes.POINTER(SpectreDecodeArgs)]
        lib.spectre_decode_step.restype = ctypes.c_int
        lib.spectre_decode_head_workspace_bytes.argtypes = [ctypes.c_int64] * 4
        lib.spectre_decode_head_workspace_bytes.restype = ctypes.c_int64
        lib.spectre_decode_head_step.argtypes = [ctypes.POINTER(SpectreDecodeHeadArgs)]
        lib.spectre_decode_head_step.restype = ctypes.c_int
        lib.spectre_probe_copy.argtypes = [ctypes.POINTER(SpectreProbeArgs), ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_float)]
        lib.spectre_probe_copy.restype = ctypes.c_int
        lib.spectre_wavelet_refine.argtypes = [ctypes.POINTER(SpectreWaveletArgs)]
        lib.spectre_wavelet_refine.restype = ctypes.c_int
        lib.spectre_wavelet_gate_grad.argtypes = [ctypes.POINTER(SpectreWaveletGradArgs)]
        lib.spectre_wavelet_gate_grad.restype = ctypes.c_int
        ver = lib.spectre_version()
        if ver != ABI_VERSION:
            raise NativeLibraryError(f"ABI mismatch: library {ver}, binding {ABI_VERSION}")
        _lib = lib
    return _lib


_ERR = {1: ValueError, 2: NotImplementedError, 3: RuntimeError, 4: ValueError}


def check(rc: int, what: str):
    if rc != 0:
        msg = load().spectre_last_error().decode("utf-8", "replace")
        raise _ERR.get(rc, RuntimeError)(f"{what}: {msg} (code {rc})")
