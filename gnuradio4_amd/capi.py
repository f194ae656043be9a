"""ctypes binding of libgr4hip.so (include/gr4hip.h).  The HIP kernel library is mandatory: there is no CPU fallback
in this package -- a missing or unloadable library raises immediately (tests on a GPU box must exercise native code)."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgr4hip.so")

OK, DONE, INSUFFICIENT_INPUT, INSUFFICIENT_OUTPUT = 0, -1, -2, -3
ERROR, INVALID_ARGUMENT, RUNTIME_ERROR, UNSUPPORTED, NO_DEVICE = -100, -101, -102, -103, -104
U8, U16, U32, U64, I8, I16, I32, I64, F32, F64, C32, C64, UF32, UF64 = range(14)  # UF32 / UF64: gr::UncertainValue<float | double>, {value, uncertainty} pairs
ADD, SUB, MUL, DIV = range(4)
DF_I, DF_II, DF_I_TRANSPOSED, DF_II_TRANSPOSED = range(4)
WINDOWS = ["None", "Rectangular", "Hamming", "Hann", "HannExp", "Blackman", "Nuttall", "BlackmanHarris",
           "BlackmanNuttall", "FlatTop", "Exponential", "Kaiser"]
FFT_OUTPUT_IN_DB, FFT_OUTPUT_IN_DEG, FFT_UNWRAP_PHASE = 1, 2, 4
CHAIN_AUTO, CHAIN_UNFUSED, CHAIN_FUSED_TD, CHAIN_FUSED_FD, CHAIN_TIME_DOMAIN = range(5)
GUARD_STRICT, GUARD_DEFERRED, GUARD_OFF = range(3)
FIR_AUTO, FIR_TIME_DOMAIN, FIR_EXACT_F32, FIR_TIME_DOMAIN_F32, FIR_TIME_DOMAIN_BF16X3 = range(5)
ROTATOR_CLOSED_FORM, ROTATOR_RECURRENCE = range(2)
IIR_AUTO, IIR_PARALLEL, IIR_SEQUENTIAL_F32 = range(3)
FIR_IIR_AUTO, FIR_IIR_ONE_LAUNCH, FIR_IIR_TWO_LAUNCHES = range(3)
SYNTH_MIX = 0xd1b54a32d192ed03  # group g of a synth stream: Xoshiro256pp(seed ^ SYNTH_MIX * (g + 1)) (include/gr4hip.h)


class Gr4HipError(RuntimeError):
    def __init__(self, status: int, where: str, detail: str):
        self.status = status
        super().__init__(f"{where}: status {status} ({detail})")


# every symbol include/gr4hip.h declares: (name, restype, argtypes)
_vp, _sz, _i, _f, _d = C.c_void_p, C.c_size_t, C.c_int, C.c_float, C.c_double
_pvp, _psz, _pi, _pf = C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_int), C.POINTER(C.c_float)
SIGNATURES = {
    "gr4hip_abi_version": (_i, []),
    "gr4hip_last_error": (C.c_char_p, []),
    "gr4hip_status_string": (C.c_char_p, [_i]),
    "gr4hip_device_count": (_i, [_pi]),
    "gr4hip_set_device": (_i, [_i]),
    "gr4hip_get_device": (_i, [_pi]),
    "gr4hip_device_name": (_i, [_i, C.c_char_p, _sz]),
    "gr4hip_malloc": (_i, [_pvp, _sz]),
    "gr4hip_free": (_i, [_vp]),
    "gr4hip_malloc_host": (_i, [_pvp, _sz]),
    "gr4hip_free_host": (_i, [_vp]),
    "gr4hip_host_ring_create": (_i, [_pvp, _sz]),
    "gr4hip_host_ring_destroy": (_i, [_vp, _sz]),
    "gr4hip_memcpy_h2d": (_i, [_vp, _vp, _sz, _vp]),
    "gr4hip_memcpy_d2h": (_i, [_vp, _vp, _sz, _vp]),
    "gr4hip_memcpy_d2d": (_i, [_vp, _vp, _sz, _vp]),
    "gr4hip_memset": (_i, [_vp, _i, _sz, _vp]),
    "gr4hip_stream_create": (_i, [_pvp]),
    "gr4hip_stream_destroy": (_i, [_vp]),
    "gr4hip_stream_synchronize": (_i, [_vp]),
    "gr4hip_stream_query": (_i, [_vp, C.POINTER(C.c_int)]),
    "gr4hip_event_create": (_i, [_pvp]),
    "gr4hip_event_destroy": (_i, [_vp]),
    "gr4hip_event_record": (_i, [_vp, _vp]),
    "gr4hip_event_synchronize": (_i, [_vp]),
    "gr4hip_event_query": (_i, [_vp, _pi]),
    "gr4hip_stream_wait_event": (_i, [_vp, _vp]),
    "gr4hip_event_elapsed_ms": (_i, [_vp, _vp, _pf]),
    "gr4hip_ring_create": (_i, [_pvp, _sz]),
    "gr4hip_ring_destroy": (_i, [_vp]),
    "gr4hip_ring_base": (_i, [_vp, _pvp]),
    "gr4hip_ring_size": (_i, [_vp, _psz]),
    "gr4hip_fir64_create": (_i, [_pvp, _vp, _sz, _sz]),
    "gr4hip_fir64_set_taps": (_i, [_vp, _vp, _sz]),
    "gr4hip_fir64_reset": (_i, [_vp]),
    "gr4hip_fir64_process": (_i, [_vp, _vp, _sz, _vp, _psz, _vp]),
    "gr4hip_fir64_destroy": (_i, [_vp]),
    "gr4hip_iir64_create": (_i, [_pvp, _i, _sz, _vp, _sz, _vp, _sz]),
    "gr4hip_iir64_reset": (_i, [_vp]),
    "gr4hip_iir64_process": (_i, [_vp, _vp, _sz, _vp, _vp]),
    "gr4hip_iir64_destroy": (_i, [_vp]),
    "gr4hip_fft64_create": (_i, [_pvp, _sz, _i, _i]),
    "gr4hip_fft64_process": (_i, [_vp, _vp, _sz, _vp, _vp, _vp, _vp, _vp]),
    "gr4hip_fft64_destroy": (_i, [_vp]),
    "gr4hip_rotator64_create": (_i, [_pvp, _d, _d]),
    "gr4hip_rotator64_reset": (_i, [_vp, _d]),
    "gr4hip_rotator64_process": (_i, [_vp, _vp, _vp, _sz, _vp]),
    "gr4hip_rotator64_phase": (_i, [_vp, C.POINTER(C.c_double)]),
    "gr4hip_rotator64_destroy": (_i, [_vp]),
    "gr4hip_fir_create": (_i, [_pvp, _i, _vp, _sz, _sz]),
    "gr4hip_fir_set_taps": (_i, [_vp, _vp, _sz]),
    "gr4hip_fir_set_algo": (_i, [_vp, _i]),
    "gr4hip_fir_reset": (_i, [_vp]),
    "gr4hip_fir_process": (_i, [_vp, _vp, _sz, _vp, _psz, _vp]),
    "gr4hip_fir_destroy": (_i, [_vp]),
    "gr4hip_fir_interp_create": (_i, [_pvp, _i, _vp, _sz, _sz]),
    "gr4hip_fir_interp_set_taps": (_i, [_vp, _vp, _sz]),
    "gr4hip_fir_interp_reset": (_i, [_vp]),
    "gr4hip_fir_interp_process": (_i, [_vp, _vp, _sz, _vp, _psz, _vp]),
    "gr4hip_fir_interp_destroy": (_i, [_vp]),
    "gr4hip_decimate": (_i, [_i, _vp, _sz, _sz, _vp, _psz, _vp]),
    "gr4hip_iir_create": (_i, [_pvp, _i, _sz, _vp, _sz, _vp, _sz]),
    "gr4hip_iir_reset": (_i, [_vp]),
    "gr4hip_iir_process": (_i, [_vp, _vp, _sz, _vp, _vp]),
    "gr4hip_iir_status": (_i, [_vp, _vp]),
    "gr4hip_iir_set_algo": (_i, [_vp, _i]),
    "gr4hip_iir_get_algo": (_i, [_vp, _pi, _pf, _pf]),
    "gr4hip_iir_destroy": (_i, [_vp]),
    "gr4hip_fir_iir_process": (_i, [_vp, _vp, _vp, _sz, _vp, _psz, _i, _vp]),
    "gr4hip_filter_params_default": (_i, [_vp]),
    "gr4hip_fir_design": (_i, [_i, _vp, _i, _vp, _sz, _psz]),
    "gr4hip_iir_design": (_i, [_i, _vp, _i, _vp, _vp, _sz, _psz]),
    "gr4hip_fft_create": (_i, [_pvp, _i, _sz, _i, _i]),
    "gr4hip_fft_process": (_i, [_vp, _vp, _sz, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gr4hip_fft_spectrum": (_i, [_vp, _vp, _sz, _vp, _vp]),
    "gr4hip_fft_mag2": (_i, [_vp, _vp, _sz, _vp, _vp]),
    "gr4hip_fft_set_epilogue": (_i, [_vp, _vp]),
    "gr4hip_fft_destroy": (_i, [_vp]),
    "gr4hip_fft_plan": (_i, [_sz, _pi, _vp, _pi]),
    "gr4hip_window_create": (_i, [_i, _vp, _sz, _f]),
    "gr4hip_window_create_f64": (_i, [_i, _vp, _sz, _d]),
    "gr4hip_chain_create": (_i, [_pvp, _vp, _sz, _sz, _i, _i]),
    "gr4hip_chain_reset": (_i, [_vp]),
    "gr4hip_chain_process": (_i, [_vp, _vp, _sz, _vp, _psz, _vp]),
    "gr4hip_chain_get_algo": (_i, [_vp, _pi]),
    "gr4hip_chain_last_power_ratio": (_i, [_vp, _pf, _pi, _vp]),
    "gr4hip_chain_last_guard_fractions": (_i, [_vp, _pf, _pf, _vp]),
    "gr4hip_chain_set_max_workgroups": (_i, [_vp, C.c_uint]),
    "gr4hip_chain_set_guard_mode": (_i, [_vp, _i]),
    "gr4hip_fir_set_guard_mode": (_i, [_vp, _i]),
    "gr4hip_developer_switch": (_i, [C.c_char_p, _i]),
    "gr4hip_chain_process_multi": (_i, [_vp, _sz, _vp, _sz, _vp, _vp, _psz, _vp]),
    "gr4hip_chain_destroy": (_i, [_vp]),
    "gr4hip_fanin_unique_id": (_i, [_vp]),
    "gr4hip_fanin_create": (_i, [_pvp, _vp, _i, _i]),
    "gr4hip_fanin_rank": (_i, [_vp, _pi, _pi]),
    "gr4hip_fanin_reduce_scatter_sum_f32": (_i, [_vp, _vp, _vp, _sz, _vp]),
    "gr4hip_fanin_all_to_all_sum_f32": (_i, [_vp, _vp, _vp, _vp, _sz, _vp]),
    "gr4hip_fanin_all_reduce_sum_f32": (_i, [_vp, _vp, _vp, _sz, _vp]),
    "gr4hip_fanin_destroy": (_i, [_vp]),
    "gr4hip_math_const": (_i, [_i, _i, _vp, _vp, _sz, _vp, _vp]),
    "gr4hip_math_nary": (_i, [_i, _i, _vp, _sz, _vp, _sz, _vp]),
    "gr4hip_ewise_create": (_i, [_pvp, _i]),
    "gr4hip_ewise_append_const": (_i, [_vp, _i, _vp]),
    "gr4hip_ewise_append_rotator": (_i, [_vp, _f, _f]),
    "gr4hip_ewise_length": (_i, [_vp, _psz]),
    "gr4hip_ewise_reset": (_i, [_vp]),
    "gr4hip_ewise_position": (_i, [_vp, C.POINTER(C.c_uint64)]),
    "gr4hip_ewise_process": (_i, [_vp, _vp, _vp, _sz, _vp]),
    "gr4hip_ewise_decimate": (_i, [_vp, _vp, _sz, _sz, _vp, _psz, _vp]),
    "gr4hip_ewise_destroy": (_i, [_vp]),
    "gr4hip_fir_set_prologue": (_i, [_vp, _vp]),
    "gr4hip_fir_set_epilogue": (_i, [_vp, _vp]),
    "gr4hip_rotator_create": (_i, [_pvp, _f, _f]),
    "gr4hip_rotator_set_algo": (_i, [_vp, _i]),
    "gr4hip_rotator_reset": (_i, [_vp, _f]),
    "gr4hip_rotator_process": (_i, [_vp, _vp, _vp, _sz, _vp]),
    "gr4hip_rotator_phase": (_i, [_vp, _pf, _vp]),
    "gr4hip_rotator_destroy": (_i, [_vp]),
    "gr4hip_fir_batched_create": (_i, [_pvp, _sz, _vp, _sz]),
    "gr4hip_fir_batched_reset": (_i, [_vp]),
    "gr4hip_fir_batched_process": (_i, [_vp, _vp, _sz, _sz, _vp, _sz, _vp]),
    "gr4hip_fir_batched_destroy": (_i, [_vp]),
    "gr4hip_synth_draws": (_i, [_vp, _sz, C.c_uint64, C.c_uint64, _vp]),
    "gr4hip_synth_c32": (_i, [_vp, _sz, C.c_uint64, _d, _f, _f, _vp]),
    "gr4hip_synth_f32": (_i, [_vp, _sz, C.c_uint64, _d, _f, _f, _vp]),
}

class FilterParams(C.Structure):
    _fields_ = [("order", C.c_size_t), ("f_low", _d), ("f_high", _d), ("gain", _d), ("ripple_db", _d), ("attenuation_db", _d),
                ("beta", _d), ("fs", _d)]


LOWPASS, HIGHPASS, BANDPASS, BANDSTOP = range(4)
BUTTERWORTH, BESSEL, CHEBYSHEV1, CHEBYSHEV2 = range(4)

_lib = None


def lib():
    """Load libgr4hip.so (once).  torch is imported first so that its bundled HIP runtime (same SONAME
    libamdhip64.so.7) is the one instance both share -- device pointers and streams are then interchangeable."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(the HIP kernel library is required; there is no CPU fallback)")
        import torch  # noqa: F401  (must precede the CDLL load, see docstring)
        L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError here == header/library mismatch
            fn.restype = res
            fn.argtypes = args
        if L.gr4hip_abi_version() != 1:
            raise ImportError("libgr4hip.so ABI version mismatch")
        _lib = L
    return _lib


def check(rc: int, where: str) -> int:
    if rc < 0 and rc not in (DONE, INSUFFICIENT_INPUT, INSUFFICIENT_OUTPUT):
        raise Gr4HipError(rc, where, lib().gr4hip_last_error().decode(errors="replace"))
    return rc


def device_count() -> int:
    n = C.c_int(0)
    check(lib().gr4hip_device_count(C.byref(n)), "device_count")
    return n.value


def developer_switch(name: str, value: int = 1) -> None:
    """kernel A/B switches of the library (include/gr4hip.h: gr4hip_developer_switch); the tests use it where they compare two kernels"""
    check(lib().gr4hip_developer_switch(name.encode(), int(value)), "developer_switch")
