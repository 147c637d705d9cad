"""fft_amd — MI355X-native SPECTRE spectral token mixing (fused rfft -> gate -> irfft, gfx950 HIP).

    from fft_amd import SpectreHead, spectral_mix

`spectral_mix` is the fused kernel behind the C ABI in include/spectre_hip.h; `SpectreHead` is the drop-in
nn.Module for the reference layer (/root/reference/spectre.py:400-557).  HIP devices only — no CPU path.
"""
from .functional import (copy_probe, describe, empty_on_fast_allocation, get_tile_order, set_tile_order, spectral_gate_fused, spectral_mix,
                         spectral_mix_backward, time_kernel, wavelet_refine)
from .decode import PrefixFFTCache, rfft_prefill
from .shard import batch_shard
from .spectre import (AttentionPooling, ComplexModReLU, DCTPooling, MeanPool, SpectreBlock, SpectreHead, SpectreMultiHead,
                      WaveletRefinement, complex_conv1d, interp_complex_1d, resample_complex)

__all__ = ["spectral_mix", "spectral_mix_backward", "spectral_gate_fused", "describe", "time_kernel", "set_tile_order", "get_tile_order", "copy_probe", "empty_on_fast_allocation", "SpectreHead", "SpectreMultiHead", "SpectreBlock", "WaveletRefinement", "wavelet_refine", "ComplexModReLU", "DCTPooling",
           "AttentionPooling", "MeanPool", "resample_complex", "interp_complex_1d", "complex_conv1d", "batch_shard", "PrefixFFTCache", "rfft_prefill"]
__version__ = "0.1.0"
