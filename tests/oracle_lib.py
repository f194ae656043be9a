"""ctypes loader for the CPU oracle (oracle/liboracle.so) -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Nothing here (and nothing under /root/reference) is reachable from the product package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

U8, U16, U32, U64, I8, I16, I32, I64, F32, F64, C32, C64, UF32, UF64 = range(14)
NP_DTYPES = [np.uint8, np.uint16, np.uint32, np.uint64, np.int8, np.int16, np.int32, np.int64,
             np.float32, np.float64, np.complex64, np.complex128,
             np.float32, np.float64]  # UF32 / UF64 (gr::UncertainValue<float | double>): arrays of shape [n, 2] = {value, uncertainty} rows
ADD, SUB, MUL, DIV = range(4)
DF_I, DF_II, DF_I_T, DF_II_T = range(4)
LOWPASS, HIGHPASS, BANDPASS, BANDSTOP = range(4)
BUTTERWORTH, BESSEL, CHEBYSHEV1, CHEBYSHEV2 = range(4)
WINDOWS = ["None", "Rectangular", "Hamming", "Hann", "HannExp", "Blackman", "Nuttall", "BlackmanHarris",
           "BlackmanNuttall", "FlatTop", "Exponential", "Kaiser"]
MAX_ORDER = 16


class Section(C.Structure):
    _fields_ = [("nb", C.c_int), ("na", C.c_int),
                ("b", C.c_double * (MAX_ORDER + 1)), ("a", C.c_double * (MAX_ORDER + 1)),
                ("xh", C.c_double * (MAX_ORDER + 1)), ("yh", C.c_double * (MAX_ORDER + 1))]


class FilterParams(C.Structure):
    _fields_ = [("order", C.c_size_t), ("fLow", C.c_double), ("fHigh", C.c_double), ("gain", C.c_double),
                ("rippleDb", C.c_double), ("attenuationDb", C.c_double), ("beta", C.c_double), ("fs", C.c_double)]


def build(force: bool = False) -> None:
    """Compile the oracle (and oracle/_ref when /root/reference is present). Building the checker is not using it."""
    so = os.path.join(ORACLE_DIR, "liboracle.so")
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("gr4_oracle.c", "gr4_oracle_tmpl.inc", "gr4_oracle.h", "Makefile")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s", "all"], stdout=subprocess.DEVNULL)


_lib = None


def _cpu_signature() -> str:
    """model name + ISA flags of the host this process runs on (what -march=native resolves against)"""
    try:
        txt = open("/proc/cpuinfo").read()
        model = next((l.split(":", 1)[1].strip() for l in txt.splitlines() if l.startswith("model name")), "?")
        flags = next((l.split(":", 1)[1].strip() for l in txt.splitlines() if l.startswith("flags")), "")
        import hashlib
        return model + " / " + hashlib.sha1(flags.encode()).hexdigest()[:12]
    except Exception:
        return "unknown"


def build_fast_for_this_host() -> str:
    """liboracle_fast.so is compiled with -march=native (the reference benchmarks' flags): a binary built in one container and shipped to another box carries the
    BUILD host's ISA.  The signature of the host it was built on is kept beside it; on a different host it is rebuilt here (gcc is part of the image) before the
    cpu_baseline is timed.  Returns what happened, for the bench line."""
    so, tag = os.path.join(ORACLE_DIR, "liboracle_fast.so"), os.path.join(ORACLE_DIR, "liboracle_fast.so.host")
    sig = _cpu_signature()
    have = open(tag).read().strip() if os.path.exists(tag) else None
    if os.path.exists(so) and have == sig:
        return "built on this host (" + sig.split(" / ")[0] + ")"
    try:
        if os.path.exists(so):
            os.remove(so)
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s", "liboracle_fast.so"], stdout=subprocess.DEVNULL)
        open(tag, "w").write(sig + "\n")
        return "rebuilt -march=native on this host (" + sig.split(" / ")[0] + ")"
    except Exception as e:  # no compiler here: the shipped binary is what there is
        return "shipped binary (built elsewhere: " + str(have) + "); rebuild failed: " + str(e)[:80]


def lib(fast: bool = False):
    global _lib
    build()
    if fast:
        return _declare(C.CDLL(os.path.join(ORACLE_DIR, "liboracle_fast.so")))
    if _lib is None:
        _lib = _declare(C.CDLL(os.path.join(ORACLE_DIR, "liboracle.so")))
    return _lib


def ref_lib():
    """oracle/_ref/libgr4ref.so: compiled from the reference's own rng headers. None if not built."""
    p = os.path.join(ORACLE_DIR, "_ref", "libgr4ref.so")
    if not os.path.exists(p):
        return None
    L = C.CDLL(p)
    L.gr4ref_xoshiro_draws.argtypes = [C.c_uint64, C.c_void_p, C.c_size_t]
    L.gr4ref_gauss_fill_f32.argtypes = [C.c_uint64, C.c_void_p, C.c_size_t, C.c_float, C.c_float]
    L.gr4ref_gauss_fill_c32.argtypes = [C.c_uint64, C.c_void_p, C.c_size_t, C.c_float, C.c_float]
    return L


def _declare(L):
    vp, sz, f32, f64, i = C.c_void_p, C.c_size_t, C.c_float, C.c_double, C.c_int
    L.gr4o_xoshiro_seed.argtypes = [vp, C.c_uint64]
    L.gr4o_xoshiro_next.argtypes = [vp]
    L.gr4o_xoshiro_next.restype = C.c_uint64
    L.gr4o_gauss_fill_f32.argtypes = [vp, vp, sz, f32, f32]
    L.gr4o_gauss_fill_c32.argtypes = [vp, vp, sz, f32, f32]
    L.gr4o_signal_c32.argtypes = [C.c_uint64, vp, sz, f64, f64, f32]
    L.gr4o_signal_f32.argtypes = [C.c_uint64, vp, sz, f64, f64, f32]
    for n in ("gr4o_fir_f32", "gr4o_fir_f32_acc64", "gr4o_fir_c32", "gr4o_fir_c32_acc64"):
        getattr(L, n).argtypes = [vp, sz, vp, vp, vp, sz]
    L.gr4o_fir_decim_f32_acc64.argtypes = [vp, sz, vp, vp, vp, sz, sz]
    L.gr4o_fir_interp_f32_acc64.argtypes = [vp, sz, sz, vp, vp, vp, sz]
    L.gr4o_fir_interp_c32_acc64.argtypes = [vp, sz, sz, vp, vp, vp, sz]
    L.gr4o_decimate_bytes.argtypes = [vp, vp, sz, sz, sz]
    L.gr4o_decimate_bytes.restype = sz
    L.gr4o_section_init.argtypes = [vp, vp, i, vp, i]
    L.gr4o_section_step.argtypes = [vp, f64, i, i]
    L.gr4o_section_step.restype = f64
    L.gr4o_iir_cascade_f32.argtypes = [vp, i, i, vp, vp, sz]
    L.gr4o_iir_cascade_f64.argtypes = [vp, i, i, vp, vp, sz]
    L.gr4o_filter_params_default.argtypes = [vp]
    L.gr4o_fir_design.argtypes = [i, vp, i, i, vp, i]
    L.gr4o_iir_design.argtypes = [i, vp, i, i, vp, i]
    L.gr4o_section_response.argtypes = [vp, f64]
    L.gr4o_analog_response.argtypes = [i, vp, i, f64]
    L.gr4o_analog_response.restype = f64
    L.gr4o_section_response.restype = f64
    L.gr4o_window_f32.argtypes = [i, vp, sz, f32]
    L.gr4o_window_f64.argtypes = [i, vp, sz, f64]
    L.gr4o_dft_c64.argtypes = [vp, vp, sz]
    L.gr4o_fft_c32.argtypes = [vp, vp, sz]
    L.gr4o_magnitude_f32.argtypes = [vp, sz, vp, i, i, i]
    L.gr4o_magnitude_f64.argtypes = [vp, sz, vp, i, i, i]
    L.gr4o_unwrap_f64.argtypes = [vp, sz]
    L.gr4o_unwrap_f32.argtypes = [vp, sz]
    L.gr4o_phase_f32.argtypes = [vp, sz, vp, i, i, i, i]
    L.gr4o_phase_f64.argtypes = [vp, sz, vp, i, i, i, i]
    L.gr4o_fft_block_c32.argtypes = [vp, sz, i, i, i, i, vp, vp, vp, vp]
    L.gr4o_fft_block_c32_truth.argtypes = [vp, sz, i, i, i, i, vp, vp, vp, vp]
    L.gr4o_fft_block_f32_truth.argtypes = [vp, sz, i, i, i, i, vp, vp, vp, vp]
    L.gr4o_chain_c32.argtypes = [vp, sz, vp, sz, i, vp, vp, sz]
    L.gr4o_chain_c32_truth.argtypes = [vp, sz, vp, sz, i, vp, vp, sz]
    L.gr4o_math_const.argtypes = [i, i, vp, vp, sz, vp]
    L.gr4o_math_nary.argtypes = [i, i, vp, sz, vp, sz]
    L.gr4o_rotator_c32.argtypes = [vp, f32, vp, vp, sz]
    L.gr4o_rotator_c64.argtypes = [vp, f64, vp, vp, sz]
    return L


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


# ---------------------------------------------------------------- convenience wrappers (numpy in / numpy out)
def xoshiro_draws(seed: int, n: int) -> np.ndarray:
    st = (C.c_uint64 * 4)()
    lib().gr4o_xoshiro_seed(st, seed)
    return np.array([lib().gr4o_xoshiro_next(st) for _ in range(n)], dtype=np.uint64)


def gauss_f32(seed: int, n: int, amp=1.0, off=0.0) -> np.ndarray:
    st = (C.c_uint64 * 4)()
    lib().gr4o_xoshiro_seed(st, seed)
    out = np.empty(n, np.float32)
    lib().gr4o_gauss_fill_f32(st, _p(out), n, amp, off)
    return out


def gauss_c32(seed: int, n: int, amp=1.0, off=0.0) -> np.ndarray:
    st = (C.c_uint64 * 4)()
    lib().gr4o_xoshiro_seed(st, seed)
    out = np.empty(n, np.complex64)
    lib().gr4o_gauss_fill_c32(st, _p(out), n, amp, off)
    return out


def signal_c32(seed: int, n: int, tone_frel=0.1, tone_amp=1.0, noise_amp=1.0) -> np.ndarray:
    out = np.empty(n, np.complex64)
    lib().gr4o_signal_c32(seed, _p(out), n, tone_frel, tone_amp, noise_amp)
    return out


def signal_f32(seed: int, n: int, tone_frel=0.1, tone_amp=1.0, noise_amp=1.0) -> np.ndarray:
    out = np.empty(n, np.float32)
    lib().gr4o_signal_f32(seed, _p(out), n, tone_frel, tone_amp, noise_amp)
    return out


def fir(b, x, hist=None, acc64=True, L=None):
    """fir_filter::processOne over x. returns (y, hist). x float32 or complex64; b float32."""
    L = L or lib()
    b = np.ascontiguousarray(b, np.float32)
    x = np.ascontiguousarray(x)
    cplx = np.iscomplexobj(x)
    x = x.astype(np.complex64 if cplx else np.float32, copy=False)
    H = len(b) - 1
    if hist is None:
        hist = np.zeros(max(H, 1), x.dtype)
    hist = np.ascontiguousarray(hist).copy()
    ydt = (np.complex128 if cplx else np.float64) if acc64 else x.dtype
    y = np.empty(len(x), ydt)
    fn = {(False, False): L.gr4o_fir_f32, (False, True): L.gr4o_fir_f32_acc64,
          (True, False): L.gr4o_fir_c32, (True, True): L.gr4o_fir_c32_acc64}[(cplx, acc64)]
    fn(_p(b), len(b), _p(hist), _p(x), _p(y), len(x))
    return y, hist


def fir_decim(b, x, decim, hist=None):
    b = np.ascontiguousarray(b, np.float32)
    x = np.ascontiguousarray(x, np.float32)
    H = len(b) - 1
    hist = np.zeros(max(H, 1), np.float32) if hist is None else np.ascontiguousarray(hist, np.float32).copy()
    y = np.empty(len(x) // decim, np.float64)
    lib().gr4o_fir_decim_f32_acc64(_p(b), len(b), _p(hist), _p(x), _p(y), len(x), decim)
    return y, hist


def fir_interp(b, x, interp, hist_up=None):
    """interpolating FIR (parity unpinned by the reference): zero-stuff by `interp`, a1 sum at the output rate in float64, gain `interp`.
    returns (y float64 / complex128, hist_up)."""
    b = np.ascontiguousarray(b, np.float32)
    x = np.ascontiguousarray(x)
    cplx = np.iscomplexobj(x)
    x = x.astype(np.complex64 if cplx else np.float32, copy=False)
    H = max(len(b) - 1, 1)
    hist_up = np.zeros(H, x.dtype) if hist_up is None else np.ascontiguousarray(hist_up, x.dtype).copy()
    y = np.empty(len(x) * interp, np.complex128 if cplx else np.float64)
    (lib().gr4o_fir_interp_c32_acc64 if cplx else lib().gr4o_fir_interp_f32_acc64)(_p(b), len(b), interp, _p(hist_up), _p(x), _p(y), len(x))
    return y, hist_up


def make_sections(coeffs):
    """coeffs: list of (b, a) -> ctypes array of Section."""
    arr = (Section * len(coeffs))()
    for s, (b, a) in zip(arr, coeffs):
        b = np.ascontiguousarray(b, np.float64)
        a = np.ascontiguousarray(a, np.float64)
        lib().gr4o_section_init(C.byref(s), _p(b), len(b), _p(a), len(a))
    return arr


def iir_cascade(sections, x, form=DF_II, f64=True):
    x = np.ascontiguousarray(x, np.float32)
    y = np.empty(len(x), np.float64 if f64 else np.float32)
    (lib().gr4o_iir_cascade_f64 if f64 else lib().gr4o_iir_cascade_f32)(sections, len(sections), form, _p(x), _p(y), len(x))
    return y


def filter_params(**kw) -> FilterParams:
    p = FilterParams()
    lib().gr4o_filter_params_default(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def fir_design(response, params, window, is_float=True) -> np.ndarray:
    taps = np.empty(65536, np.float64)
    n = lib().gr4o_fir_design(response, C.byref(params), window, int(is_float), _p(taps), len(taps))
    if n < 0:
        raise ValueError("fir design failed")
    return taps[:n].copy()


def iir_design(response, params, design, is_float=True):
    arr = (Section * 32)()
    n = lib().gr4o_iir_design(response, C.byref(params), design, int(is_float), arr, 32)
    if n < 0:
        raise ValueError("iir design failed")
    return [(np.array(s.b[:s.nb]), np.array(s.a[:s.na])) for s in arr[:n]]


def window(type_id: int, n: int, dtype=np.float32, beta=1.6) -> np.ndarray:
    w = np.empty(n, dtype)
    rc = (lib().gr4o_window_f32 if dtype == np.float32 else lib().gr4o_window_f64)(type_id, _p(w), n, beta)
    if rc:
        raise ValueError("window")
    return w


def dft64(x) -> np.ndarray:
    x = np.ascontiguousarray(x, np.complex128)
    out = np.empty_like(x)
    lib().gr4o_dft_c64(_p(x), _p(out), len(x))
    return out


def fft32(x) -> np.ndarray:
    x = np.ascontiguousarray(x, np.complex64)
    out = np.empty_like(x)
    lib().gr4o_fft_c32(_p(x), _p(out), len(x))
    return out


def magnitude(spec, half=False, in_db=False, shift=False):
    spec = np.ascontiguousarray(spec)
    f32 = spec.dtype == np.complex64
    out = np.empty(len(spec) // 2 if half else len(spec), np.float32 if f32 else np.float64)
    (lib().gr4o_magnitude_f32 if f32 else lib().gr4o_magnitude_f64)(_p(spec), len(spec), _p(out), int(half), int(in_db), int(shift))
    return out


def phase(spec, half=False, in_deg=False, unwrap=False, shift=False):
    spec = np.ascontiguousarray(spec)
    f32 = spec.dtype == np.complex64
    out = np.empty(len(spec) // 2 if half else len(spec), np.float32 if f32 else np.float64)
    (lib().gr4o_phase_f32 if f32 else lib().gr4o_phase_f64)(_p(spec), len(spec), _p(out), int(half), int(in_deg), int(unwrap), int(shift))
    return out


def fft_block_truth(frame, window_id=3, in_db=False, in_deg=False, unwrap=False):
    frame = np.ascontiguousarray(frame)
    N = len(frame)
    if np.iscomplexobj(frame):
        frame = frame.astype(np.complex64, copy=False)
        outs = [np.empty(N, np.float64) for _ in range(4)]
        lib().gr4o_fft_block_c32_truth(_p(frame), N, window_id, int(in_db), int(in_deg), int(unwrap), *[_p(o) for o in outs])
    else:
        frame = frame.astype(np.float32, copy=False)
        outs = [np.empty(N // 2, np.float64) for _ in range(4)]
        lib().gr4o_fft_block_f32_truth(_p(frame), N, window_id, int(in_db), int(in_deg), int(unwrap), *[_p(o) for o in outs])
    return outs  # mag, phase, re, im


def chain(b, x, N, window_id=0, truth=True, hist=None, L=None):
    """cf32 FIR -> N-pt FFT (window) -> mag2 natural order. returns (mag2[frames*N], hist)."""
    L = L or lib()
    b = np.ascontiguousarray(b, np.float32)
    x = np.ascontiguousarray(x, np.complex64)
    H = len(b) - 1
    hist = np.zeros(max(H, 1), np.complex64) if hist is None else np.ascontiguousarray(hist, np.complex64).copy()
    frames = len(x) // N
    out = np.empty(frames * N, np.float64 if truth else np.float32)
    (L.gr4o_chain_c32_truth if truth else L.gr4o_chain_c32)(_p(b), len(b), _p(hist), N, window_id, _p(x), _p(out), len(x))
    return out, hist


def math_const(op, dtype_id, x, value):
    x = np.ascontiguousarray(x, NP_DTYPES[dtype_id])
    v = np.array([value], NP_DTYPES[dtype_id])
    out = np.empty_like(x)
    assert lib().gr4o_math_const(op, dtype_id, _p(x), _p(out), len(x), _p(v)) == 0
    return out


def math_nary(op, dtype_id, inputs):
    ins = [np.ascontiguousarray(a, NP_DTYPES[dtype_id]) for a in inputs]
    ptrs = (C.c_void_p * len(ins))(*[a.ctypes.data for a in ins])
    out = np.empty_like(ins[0])
    assert lib().gr4o_math_nary(op, dtype_id, ptrs, len(ins), _p(out), len(out)) == 0
    return out


def rotator(x, inc, phase0=0.0):
    x = np.ascontiguousarray(x)
    if x.dtype == np.complex64:
        st = C.c_float(phase0)
        y = np.empty_like(x)
        lib().gr4o_rotator_c32(C.byref(st), inc, _p(x), _p(y), len(x))
    else:
        x = x.astype(np.complex128)
        st = C.c_double(phase0)
        y = np.empty_like(x)
        lib().gr4o_rotator_c64(C.byref(st), inc, _p(x), _p(y), len(x))
    return y, st.value


def design_taps_hamming_lowpass(ntaps: int, fc: float) -> np.ndarray:
    """Bench/test tap recipe of SURVEY.md 8(d): fir::generateCoefficients-style Hamming windowed-sinc, DC gain 1.
    (float64 computed, rounded to float32).  Uses the oracle's window restatement."""
    w = window(2, ntaps, np.float64)
    M = (ntaps - 1) / 2.0
    i = np.arange(ntaps, dtype=np.float64)
    c = w * 2 * fc * np.sinc(2 * fc * (i - M))
    c = c / c.sum()
    return c.astype(np.float32)
