"""gnuradio4_amd -- MI355X-native execution of the GNU Radio 4 filter / fourier / math hot path.

Host-side mirror (Python) of the reference's block interface for that path, on top of the C-ABI kernel library
libgr4hip.so (include/gr4hip.h).  torch is used only for device memory, streams and torch.distributed.
"""
from . import capi  # noqa: F401
from .blocks import (FFT, BasicDecimatingFilter, BasicFilter, Chain, Decimator, FirBatched, Merged, Rotator, fir_filter, fir_interpolator, iir_filter,  # noqa: F401
                     math_const, math_nary, synth_c32, synth_draws, synth_f32)

__all__ = ["capi", "fir_filter", "fir_interpolator", "iir_filter", "BasicFilter", "BasicDecimatingFilter", "Decimator", "FirBatched", "FFT", "Chain", "Merged", "Rotator",
           "math_const", "math_nary", "synth_c32", "synth_draws", "synth_f32"]
