"""Host-side (Python) mirror of the reference's block interface for the hot path, on top of include/gr4hip.h.

Names, settings and error behaviour follow the reference blocks (citations relative to /root/reference):
  fir_filter / iir_filter / BasicFilter / BasicDecimatingFilter / Decimator  blocks/filter/.../time_domain_filter.hpp
  FFT                                                                         blocks/fourier/.../fft.hpp
  math_const / math_nary (AddConst..Divide)                                   blocks/math/.../Math.hpp
  Rotator                                                                     blocks/math/.../Rotator.hpp
  Chain                                                                       the runtime fusion of fir_filter -> FFT -> mag2
                                                                              (Merge<> analogue, core/.../BlockMerging.hpp:136-320)
Blocks consume and produce torch tensors that live on the GPU (`process_bulk(x) -> y`); torch only provides the
device memory and the current HIP stream.  Every block is device-only: without libgr4hip.so or without a GPU the
constructor / call raises (Gr4HipError / ImportError) -- there is no CPU fallback in this package.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np
import torch

from . import capi
from .capi import check, lib

_TORCH_DTYPE = {
    capi.U8: torch.uint8, capi.U16: torch.uint16, capi.U32: torch.uint32, capi.U64: torch.uint64,
    capi.I8: torch.int8, capi.I16: torch.int16, capi.I32: torch.int32, capi.I64: torch.int64,
    capi.F32: torch.float32, capi.F64: torch.float64, capi.C32: torch.complex64, capi.C64: torch.complex128,
}
_DTYPE_ID = {v: k for k, v in _TORCH_DTYPE.items()}
_NP_DTYPE = [np.uint8, np.uint16, np.uint32, np.uint64, np.int8, np.int16, np.int32, np.int64,
             np.float32, np.float64, np.complex64, np.complex128]
_OPS = {"Add": capi.ADD, "Subtract": capi.SUB, "Multiply": capi.MUL, "Divide": capi.DIV,
        "+": capi.ADD, "-": capi.SUB, "*": capi.MUL, "/": capi.DIV}


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _dev(x: torch.Tensor, what: str) -> torch.Tensor:
    if not isinstance(x, torch.Tensor) or not x.is_cuda:
        raise capi.Gr4HipError(capi.INVALID_ARGUMENT, what, "input must be a CUDA/HIP torch tensor (device-only path)")
    return x.resolve_conj().contiguous()  # materialise lazy conj views: the kernels read raw memory


def _out(out: Optional[torch.Tensor], numel: int, dtype, x: torch.Tensor, what: str, shape=None) -> torch.Tensor:
    """the tensor a kernel writes: allocated here, or the caller's -- which must be a contiguous device tensor of the result's dtype with room for it (the kernels
    write raw memory: an undersized or host tensor would be an out-of-bounds device write)"""
    if out is None:
        return torch.empty(shape if shape is not None else numel, dtype=dtype, device=x.device)
    if not isinstance(out, torch.Tensor) or not out.is_cuda or out.device != x.device or not out.is_contiguous() or out.dtype != dtype or out.numel() < numel:
        raise capi.Gr4HipError(capi.INVALID_ARGUMENT, what, f"out must be a contiguous tensor on the input's device, dtype {dtype}, at least {numel} elements")
    return out


def _window_id(window) -> int:
    if isinstance(window, str):
        names = [w.lower() for w in capi.WINDOWS]
        if window.lower() not in names:
            raise ValueError(f"unknown window '{window}'")
        return names.index(window.lower())
    return int(window)


class _Handle:
    _destroy = None

    def __init__(self):
        self._h = C.c_void_p()

    def close(self):
        if getattr(self, "_h", None) and self._h.value and self._destroy:
            getattr(lib(), self._destroy)(self._h)
            self._h = C.c_void_p()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


class fir_filter(_Handle):
    """gr::filter::fir_filter<T> (time_domain_filter.hpp:22-48): settings `b`; T in {float32, complex64} and the second registered type float64
    (double taps, the plain FP64 kernel of csrc/f64.hip)."""
    _destroy = "gr4hip_fir_destroy"

    def __init__(self, b: Sequence[float], dtype=torch.float32, decimate: int = 1):
        super().__init__()
        self.dtype = dtype
        self.decimate = int(decimate)
        self._f64 = dtype == torch.float64
        if self._f64:
            self._destroy = "gr4hip_fir64_destroy"
            self.b = np.ascontiguousarray(b, np.float64)
            check(lib().gr4hip_fir64_create(C.byref(self._h), self.b.ctypes.data, len(self.b), self.decimate), "fir_filter<float64>")
            return
        self.b = np.ascontiguousarray(b, np.float32)
        check(lib().gr4hip_fir_create(C.byref(self._h), _DTYPE_ID[dtype], self.b.ctypes.data, len(self.b), self.decimate), "fir_filter")

    def settings_changed(self, b: Sequence[float]):  # settingsChanged (:38-42)
        if self._f64:
            self.b = np.ascontiguousarray(b, np.float64)
            check(lib().gr4hip_fir64_set_taps(self._h, self.b.ctypes.data, len(self.b)), "fir_filter.set_taps")
            return
        self.b = np.ascontiguousarray(b, np.float32)
        check(lib().gr4hip_fir_set_taps(self._h, self.b.ctypes.data, len(self.b)), "fir_filter.set_taps")

    def reset(self):
        check((lib().gr4hip_fir64_reset if self._f64 else lib().gr4hip_fir_reset)(self._h), "fir_filter.reset")

    def set_prologue(self, prog: Optional["Merged"]):
        """per-sample blocks in FRONT of the filter, executed in its launch (gr4hip_fir_set_prologue): gains fold into the taps, anything else is a load hook"""
        check(lib().gr4hip_fir_set_prologue(self._h, prog._h if prog is not None else None), "fir_filter.set_prologue")

    def set_epilogue(self, prog: Optional["Merged"]):
        """per-sample blocks BEHIND the filter, executed in its launch (gr4hip_fir_set_epilogue)"""
        check(lib().gr4hip_fir_set_epilogue(self._h, prog._h if prog is not None else None), "fir_filter.set_epilogue")

    def set_algo(self, algo: int):
        """capi.FIR_AUTO / capi.FIR_TIME_DOMAIN (direct form also for long complex spans: error relative to the output) / capi.FIR_EXACT_F32 (IEEE float32
        multiply-add only: the reference's Inf / NaN behaviour) -- include/gr4hip.h"""
        check(lib().gr4hip_fir_set_algo(self._h, int(algo)), "fir_filter.set_algo")

    def set_guard_mode(self, mode: int):
        """capi.GUARD_STRICT (default) / GUARD_DEFERRED / GUARD_OFF: the dynamic-range guard of the frequency-domain kernels, and (OFF) of the f16 direct
        form's per-segment verdict -- gr4hip_fir_set_guard_mode, include/gr4hip.h"""
        check(lib().gr4hip_fir_set_guard_mode(self._h, int(mode)), "fir_filter.set_guard_mode")

    def process_bulk(self, x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        x = _dev(x, "fir_filter")
        if x.dtype != self.dtype:
            raise capi.Gr4HipError(capi.INVALID_ARGUMENT, "fir_filter", f"expected {self.dtype}, got {x.dtype}")
        n_out = x.numel() // self.decimate
        out = _out(out, n_out, self.dtype, x, "fir_filter.process")
        fn = lib().gr4hip_fir64_process if self._f64 else lib().gr4hip_fir_process
        check(fn(self._h, x.data_ptr(), x.numel(), out.data_ptr(), None, _stream()), "fir_filter.process")
        return out


class fir_interpolator(_Handle):
    """Interpolating FIR (north_star "decimating / interpolating FIR"; the reference only has the rate declaration Resampling<1, L>,
    annotated.hpp:121-128): zero-stuff by `interpolate`, fir_filter's sum at the output rate, gain `interpolate`,
    evaluated as a polyphase bank.  settings `b`, `interpolate`; T in {float32, complex64}; n_out = n_in * interpolate."""
    _destroy = "gr4hip_fir_interp_destroy"

    def __init__(self, b: Sequence[float], interpolate: int, dtype=torch.float32):
        super().__init__()
        self.b = np.ascontiguousarray(b, np.float32)
        self.dtype = dtype
        self.interpolate = int(interpolate)
        check(lib().gr4hip_fir_interp_create(C.byref(self._h), _DTYPE_ID[dtype], self.b.ctypes.data, len(self.b), self.interpolate), "fir_interpolator")

    def settings_changed(self, b: Sequence[float]):
        self.b = np.ascontiguousarray(b, np.float32)
        check(lib().gr4hip_fir_interp_set_taps(self._h, self.b.ctypes.data, len(self.b)), "fir_interpolator.set_taps")

    def reset(self):
        check(lib().gr4hip_fir_interp_reset(self._h), "fir_interpolator.reset")

    def process_bulk(self, x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        x = _dev(x, "fir_interpolator")
        if x.dtype != self.dtype:
            raise capi.Gr4HipError(capi.INVALID_ARGUMENT, "fir_interpolator", f"expected {self.dtype}, got {x.dtype}")
        out = _out(out, x.numel() * self.interpolate, self.dtype, x, "fir_interpolator.process")
        check(lib().gr4hip_fir_interp_process(self._h, x.data_ptr(), x.numel(), out.data_ptr(), None, _stream()), "fir_interpolator.process")
        return out


class iir_filter(_Handle):
    """gr::filter::iir_filter<float, form> (time_domain_filter.hpp:62-122) or a cascade of sections
    (gr::filter::Filter<float>, FilterTool.hpp:223-247).  b, a: [nsections][n] or 1-D for a single section."""
    _destroy = "gr4hip_iir_destroy"

    def __init__(self, b, a, form: int = capi.DF_II, dtype=torch.float32):
        super().__init__()
        self.dtype = dtype
        self._f64 = dtype == torch.float64  # iir_filter<double, form> (time_domain_filter.hpp:57-60): csrc/f64.hip
        npt = np.float64 if self._f64 else np.float32
        b = np.atleast_2d(np.asarray(b, npt))
        a = np.atleast_2d(np.asarray(a, npt))
        if b.shape[0] != a.shape[0]:
            raise ValueError("b and a need the same number of sections")
        self.b, self.a, self.form = np.ascontiguousarray(b), np.ascontiguousarray(a), form
        if self._f64:
            self._destroy = "gr4hip_iir64_destroy"
            check(lib().gr4hip_iir64_create(C.byref(self._h), form, b.shape[0], self.b.ctypes.data, b.shape[1], self.a.ctypes.data, a.shape[1]), "iir_filter<float64>")
            return
        check(lib().gr4hip_iir_create(C.byref(self._h), form, b.shape[0], self.b.ctypes.data, b.shape[1], self.a.ctypes.data, a.shape[1]), "iir_filter")

    def reset(self):
        check((lib().gr4hip_iir64_reset if self._f64 else lib().gr4hip_iir_reset)(self._h), "iir_filter.reset")

    def set_algo(self, algo: int):
        """capi.IIR_AUTO (default: the parallel-in-time kernels unless the create-time self-test finds that float32 cannot carry this cascade's state through them),
        capi.IIR_PARALLEL or capi.IIR_SEQUENTIAL_F32 (the reference's arithmetic in the requested form, one lane, sample by sample) -- include/gr4hip.h"""
        if self._f64:
            raise capi.Gr4HipError(capi.UNSUPPORTED, "iir_filter", "the float64 cascade has one evaluation")
        check(lib().gr4hip_iir_set_algo(self._h, int(algo)), "iir_filter.set_algo")

    @property
    def algo_in_use(self):
        """(evaluation in use, create-time error of the parallel kernels, of the sequential float32 form) -- errors as max |error| / output rms, < 0: not measured"""
        a, ep, es = C.c_int(0), C.c_float(0), C.c_float(0)
        check(lib().gr4hip_iir_get_algo(self._h, C.byref(a), C.byref(ep), C.byref(es)), "iir_filter.get_algo")
        return a.value, ep.value, es.value

    def status(self):
        """synchronise the current stream and raise if an earlier launch of this handle reported a look-back time-out (include/gr4hip.h)"""
        if not self._f64:
            check(lib().gr4hip_iir_status(self._h, _stream()), "iir_filter.status")

    def process_bulk(self, x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        x = _dev(x, "iir_filter")
        if x.dtype != self.dtype:
            raise capi.Gr4HipError(capi.INVALID_ARGUMENT, "iir_filter", f"expected {self.dtype}, got {x.dtype}")
        out = _out(out, x.numel(), x.dtype, x, "iir_filter.process")
        fn = lib().gr4hip_iir64_process if self._f64 else lib().gr4hip_iir_process
        check(fn(self._h, x.data_ptr(), x.numel(), out.data_ptr(), _stream()), "iir_filter.process")
        return out


def fir_iir_process(fir: "fir_filter", iir: "iir_filter", x: torch.Tensor, out: Optional[torch.Tensor] = None, mode: int = 0) -> torch.Tensor:
    """BasicDecimatingFilter (FIR) -> IIR cascade in one call (gr4hip_fir_iir_process, BASELINE configs[2]).  mode capi.FIR_IIR_ONE_LAUNCH: the cascade as the
    frequency-domain decimator's store epilogue where both qualify -- one launch, no decimated stream in HBM; FIR_IIR_TWO_LAUNCHES; FIR_IIR_AUTO (0): the faster one
    as measured (two launches).  Both handles keep their own state"""
    x = _dev(x, "fir_iir_process")
    if x.dtype != torch.float32 or fir.dtype != torch.float32 or iir.dtype != torch.float32:
        raise capi.Gr4HipError(capi.INVALID_ARGUMENT, "fir_iir_process", "float32 stream, filter and cascade")
    n_out = x.numel() // fir.decimate
    out = _out(out, n_out, torch.float32, x, "fir_iir_process")
    check(lib().gr4hip_fir_iir_process(fir._h, iir._h, x.data_ptr(), x.numel(), out.data_ptr(), None, int(mode), _stream()), "fir_iir_process")
    return out


def design_fir(filter_response: int, order: int, f_low: float, f_high: float, sample_rate: float, window="Kaiser", gain=1.0,
               attenuation_db=40.0, beta=1.6) -> np.ndarray:
    """fir::designFilter<float> (FilterTool.hpp:1006-1071) through the library's host-side restatement."""
    p = capi.FilterParams()
    lib().gr4hip_filter_params_default(C.byref(p))
    p.order, p.f_low, p.f_high, p.fs, p.gain, p.attenuation_db, p.beta = order, f_low, f_high, sample_rate, gain, attenuation_db, beta
    n = C.c_size_t(0)
    check(lib().gr4hip_fir_design(filter_response, C.byref(p), _window_id(window), None, 0, C.byref(n)), "fir_design")
    taps = np.empty(n.value, np.float32)
    check(lib().gr4hip_fir_design(filter_response, C.byref(p), _window_id(window), taps.ctypes.data, len(taps), C.byref(n)), "fir_design")
    return taps


def design_iir(filter_response: int, order: int, f_low: float, f_high: float, sample_rate: float, design: int = capi.BUTTERWORTH,
               gain=1.0, ripple_db=0.1, attenuation_db=40.0):
    """iir::designFilter<float> (FilterTool.hpp:848-917): biquad sections (b[ns][3], a[ns][3])."""
    p = capi.FilterParams()
    lib().gr4hip_filter_params_default(C.byref(p))
    p.order, p.f_low, p.f_high, p.fs, p.gain, p.ripple_db, p.attenuation_db = order, f_low, f_high, sample_rate, gain, ripple_db, attenuation_db
    b = np.zeros((32, 3), np.float32)
    a = np.zeros((32, 3), np.float32)
    ns = C.c_size_t(0)
    check(lib().gr4hip_iir_design(filter_response, C.byref(p), design, b.ctypes.data, a.ctypes.data, 32, C.byref(ns)), "iir_design")
    return b[:ns.value].copy(), a[:ns.value].copy()


class BasicFilter:
    """gr::filter::BasicFilterProto<float, Args...> (time_domain_filter.hpp:129-205), defaults :148-157."""

    def __init__(self, filter_type="IIR", filter_response=capi.LOWPASS, filter_order=3, f_low=0.1, f_high=0.2, sample_rate=1.0,
                 decimate=1, iir_design_method=capi.BUTTERWORTH, fir_design_method="Kaiser"):
        self.filter_type, self.filter_response, self.filter_order = filter_type, filter_response, filter_order
        self.f_low, self.f_high, self.sample_rate, self.decimate = f_low, f_high, sample_rate, int(decimate)
        self.iir_design_method, self.fir_design_method = iir_design_method, fir_design_method
        self.input_chunk_size = 1
        self._fir = self._iir = None
        self.design_filter()

    def design_filter(self):  # designFilter() :163-182
        self.input_chunk_size = self.decimate
        if self.filter_type == "FIR":
            self.taps = design_fir(self.filter_response, self.filter_order, self.f_low, self.f_high, self.sample_rate, self.fir_design_method)
            self._fir, self._iir = fir_filter(self.taps, torch.float32, self.decimate), None
        elif self.filter_type == "IIR":
            self.sections = design_iir(self.filter_response, self.filter_order, self.f_low, self.f_high, self.sample_rate, self.iir_design_method)
            self._iir, self._fir = iir_filter(*self.sections), None
        else:
            raise ValueError("filter_type must be 'FIR' or 'IIR'")

    def process_bulk(self, x: torch.Tensor) -> torch.Tensor:  # :184-204
        if self._fir is not None:
            return self._fir.process_bulk(x)
        y = self._iir.process_bulk(x)
        return Decimator(self.decimate).process_bulk(y) if self.decimate > 1 else y


BasicDecimatingFilter = BasicFilter  # BasicFilterProto<T, Resampling<1,1,false>> (:210-211): same class, decimate > 1


class Decimator:
    """gr::filter::Decimator<T> (time_domain_filter.hpp:215-245): keep every decim-th sample."""

    def __init__(self, decim: int = 1):
        self.decim = int(decim)
        self.input_chunk_size = self.decim

    def process_bulk(self, x: torch.Tensor) -> torch.Tensor:
        x = _dev(x, "Decimator")
        n_out = (x.numel() + self.decim - 1) // self.decim
        out = torch.empty(n_out, dtype=x.dtype, device=x.device)
        check(lib().gr4hip_decimate(_DTYPE_ID[x.dtype], x.data_ptr(), x.numel(), self.decim, out.data_ptr(), None, _stream()), "Decimator")
        return out


class FFT(_Handle):
    """gr::blocks::fft::FFT<T> (blocks/fourier/.../fft.hpp:31-251): settings fftSize, window, outputInDb, outputInDeg,
    unwrapPhase.  process_bulk returns a dict per call with the DataSet signals for all frames:
    magnitude / phase / re / im tensors [frames, n] plus `ranges` [frames, 4, 2] (fft.hpp:173-250)."""
    _destroy = "gr4hip_fft_destroy"

    def __init__(self, fftSize: int = 1024, window="Hann", outputInDb=False, outputInDeg=False, unwrapPhase=False, dtype=torch.complex64,
                 sample_rate: float = 1.0):
        super().__init__()
        self.fftSize, self.window, self.dtype, self.sample_rate = int(fftSize), window, dtype, sample_rate
        self.outputInDb, self.outputInDeg, self.unwrapPhase = outputInDb, outputInDeg, unwrapPhase
        flags = (capi.FFT_OUTPUT_IN_DB if outputInDb else 0) | (capi.FFT_OUTPUT_IN_DEG if outputInDeg else 0) | (capi.FFT_UNWRAP_PHASE if unwrapPhase else 0)
        self._f64 = dtype == torch.float64  # FFT<double> (fourier/fft.hpp:29): real double frames, outputs in double (csrc/f64.hip)
        self.input_chunk_size = self.fftSize  # fft.hpp:131-134
        self.n_out = self.fftSize if dtype == torch.complex64 else self.fftSize // 2
        if self._f64:
            self._destroy = "gr4hip_fft64_destroy"
            check(lib().gr4hip_fft64_create(C.byref(self._h), self.fftSize, _window_id(window), flags), "FFT<float64>")
            return
        check(lib().gr4hip_fft_create(C.byref(self._h), _DTYPE_ID[dtype], self.fftSize, _window_id(window), flags), "FFT")

    def _frames(self, x):
        x = _dev(x, "FFT")
        if x.dtype != self.dtype:
            raise capi.Gr4HipError(capi.INVALID_ARGUMENT, "FFT", f"expected {self.dtype}, got {x.dtype}")
        return x, x.numel() // self.fftSize

    def process_bulk(self, x: torch.Tensor, ranges: bool = True) -> dict:
        x, frames = self._frames(x)
        if self._f64:
            mk64 = lambda: torch.empty((frames, self.n_out), dtype=torch.float64, device=x.device)
            out = {"magnitude": mk64(), "phase": mk64(), "re": mk64(), "im": mk64()}
            check(lib().gr4hip_fft64_process(self._h, x.data_ptr(), frames, out["magnitude"].data_ptr(), out["phase"].data_ptr(), out["re"].data_ptr(), out["im"].data_ptr(),
                                             _stream()), "FFT.process")
            if ranges:  # fft.hpp:229-232: min / max of the four signals per frame (host-side glue for the float64 corner)
                out["ranges"] = torch.stack([torch.stack([out[k].amin(dim=1), out[k].amax(dim=1)], dim=1) for k in ("magnitude", "phase", "re", "im")], dim=1)
            return out
        mk = lambda: torch.empty((frames, self.n_out), dtype=torch.float32, device=x.device)
        out = {"magnitude": mk(), "phase": mk(), "re": mk(), "im": mk()}
        rg = torch.empty((frames, 4, 2), dtype=torch.float32, device=x.device) if ranges else None
        check(lib().gr4hip_fft_process(self._h, x.data_ptr(), frames, out["magnitude"].data_ptr(), out["phase"].data_ptr(), out["re"].data_ptr(),
                                       out["im"].data_ptr(), rg.data_ptr() if ranges else None, _stream()), "FFT.process")
        if ranges:
            out["ranges"] = rg
        n = self.n_out
        fw = self.sample_rate / self.fftSize  # frequency axis (fft.hpp:187-193)
        out["frequency"] = (np.arange(n) * fw - (n // 2) * fw) if self.dtype == torch.complex64 else np.arange(n) * fw
        return out

    def spectrum(self, x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        x, frames = self._frames(x)
        out = _out(out, frames * self.fftSize, torch.complex64, x, "FFT.spectrum", (frames, self.fftSize))
        check(lib().gr4hip_fft_spectrum(self._h, x.data_ptr(), frames, out.data_ptr(), _stream()), "FFT.spectrum")
        return out

    def set_epilogue(self, prog: Optional["Merged"]):
        """float blocks BEHIND the power spectrum (normalisation, an offset) in the transform's launch: applied to every |X|^2 of mag2() (gr4hip_fft_set_epilogue)"""
        check(lib().gr4hip_fft_set_epilogue(self._h, prog._h if prog is not None else None), "FFT.set_epilogue")

    def mag2(self, x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        x, frames = self._frames(x)
        out = _out(out, frames * self.fftSize, torch.float32, x, "FFT.mag2", (frames, self.fftSize))
        check(lib().gr4hip_fft_mag2(self._h, x.data_ptr(), frames, out.data_ptr(), _stream()), "FFT.mag2")
        return out


class Chain(_Handle):
    """complex<float> fir_filter -> FFT frames -> |X|^2, one fused device pipeline (BASELINE.json configs[1])."""
    _destroy = "gr4hip_chain_destroy"

    def __init__(self, b: Sequence[float], fftSize: int = 8192, window="None", algo: int = capi.CHAIN_AUTO):
        super().__init__()
        self.b = np.ascontiguousarray(b, np.float32)
        self.fftSize = int(fftSize)
        check(lib().gr4hip_chain_create(C.byref(self._h), self.b.ctypes.data, len(self.b), self.fftSize, _window_id(window), algo), "Chain")
        a = C.c_int(0)
        check(lib().gr4hip_chain_get_algo(self._h, C.byref(a)), "Chain.algo")
        self.algo = a.value

    def set_max_workgroups(self, n: int):
        """cap the persistent grid of the fused kernels (0 = every CU); leaves CUs to kernels on other streams, e.g. an RCCL fan-in"""
        check(lib().gr4hip_chain_set_max_workgroups(self._h, int(n)), "Chain.set_max_workgroups")

    def reset(self):
        check(lib().gr4hip_chain_reset(self._h), "Chain.reset")

    def last_power_ratio(self):
        """(output / input power of the last measured fused launch or -1, chain now in the time domain?) -- the dynamic-range guard of CHAIN_AUTO (include/gr4hip.h)"""
        r, td = C.c_float(0), C.c_int(0)
        check(lib().gr4hip_chain_last_power_ratio(self._h, C.byref(r), C.byref(td), _stream()), "Chain.last_power_ratio")
        return r.value, bool(td.value)

    def last_guard_fractions(self):
        """(fraction of the last measured launch's frames the fused kernel marked or -1, fraction of all frames since create / reset that ended in float64)"""
        m, f = C.c_float(0), C.c_float(0)
        check(lib().gr4hip_chain_last_guard_fractions(self._h, C.byref(m), C.byref(f), _stream()), "Chain.last_guard_fractions")
        return m.value, f.value

    def process_bulk(self, x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        x = _dev(x, "Chain")
        if x.dtype != torch.complex64:
            raise capi.Gr4HipError(capi.INVALID_ARGUMENT, "Chain", "input must be complex64")
        frames = x.numel() // self.fftSize
        out = _out(out, frames * self.fftSize, torch.float32, x, "Chain.process", (frames, self.fftSize))
        nf = C.c_size_t(0)
        check(lib().gr4hip_chain_process(self._h, x.data_ptr(), x.numel(), out.data_ptr(), C.byref(nf), _stream()), "Chain.process")
        return out

    def set_guard_mode(self, mode: int):
        """capi.GUARD_STRICT (default: the frames a launch marks are evaluated again in the time domain by a launch enqueued behind it; asynchronous),
        GUARD_DEFERRED (no second evaluation: the call that measures the drop is published from the fused kernel, the switch lags one call) or GUARD_OFF (include/gr4hip.h)"""
        check(lib().gr4hip_chain_set_guard_mode(self._h, int(mode)), "Chain.set_guard_mode")


def chain_process_multi(chains: Sequence[Chain], xs: Sequence[torch.Tensor], outs: Optional[Sequence[torch.Tensor]] = None,
                        sum_out: Optional[torch.Tensor] = None, want_outs: bool = True):
    """n chains fed the same number of samples in ONE call (gr4hip_chain_process_multi): the parallel SDR channels of a flowgraph that share
    this device.  Returns (outs or None, sum_out or None).  want_outs=False with sum_out given: only the combiner output sum_i |FFT(fir(x_i))|^2
    (math::Add over the channels) is produced -- one launch with the fold in registers when all chains have the same taps."""
    n = len(chains)
    assert n >= 1 and len(xs) == n
    xs = [_dev(x, "chain_process_multi") for x in xs]
    N = chains[0].fftSize
    frames = xs[0].numel() // N
    for x in xs:
        if x.dtype != torch.complex64 or x.numel() != xs[0].numel():
            raise capi.Gr4HipError(capi.INVALID_ARGUMENT, "chain_process_multi", "inputs must be complex64 spans of one length")
    if want_outs and outs is None:
        outs = [torch.empty((frames, N), dtype=torch.float32, device=xs[0].device) for _ in range(n)]
    if not want_outs:
        outs = None
        if sum_out is None:
            sum_out = torch.empty((frames, N), dtype=torch.float32, device=xs[0].device)
    hs = (C.c_void_p * n)(*[c._h for c in chains])
    ins = (C.c_void_p * n)(*[x.data_ptr() for x in xs])
    os_ = (C.c_void_p * n)(*[o.data_ptr() for o in outs]) if outs is not None else None
    nf = C.c_size_t(0)
    check(lib().gr4hip_chain_process_multi(hs, n, ins, xs[0].numel(), os_, sum_out.data_ptr() if sum_out is not None else None, C.byref(nf), _stream()),
          "chain_process_multi")
    return outs, sum_out


class FirBatched(_Handle):
    """nchannels independent fir_filter<float> instances with per-channel taps b[c][k] on channel-major samples x[c][n]
    (BASELINE.json configs[3]); evaluated as a block-Toeplitz contraction on the f32 MFMA units."""
    _destroy = "gr4hip_fir_batched_destroy"

    def __init__(self, b):
        super().__init__()
        self.b = np.ascontiguousarray(b, np.float32)
        if self.b.ndim != 2:
            raise ValueError("taps must be [nchannels][ntaps]")
        check(lib().gr4hip_fir_batched_create(C.byref(self._h), self.b.shape[0], self.b.ctypes.data, self.b.shape[1]), "FirBatched")

    def reset(self):
        check(lib().gr4hip_fir_batched_reset(self._h), "FirBatched.reset")

    def process_bulk(self, x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        x = _dev(x, "FirBatched")
        if x.dtype != torch.float32 or x.dim() != 2 or x.shape[0] != self.b.shape[0]:
            raise capi.Gr4HipError(capi.INVALID_ARGUMENT, "FirBatched", "input must be float32 [nchannels][n]")
        if out is None:
            out = torch.empty((x.shape[0], (x.shape[1] + 3) // 4 * 4), dtype=torch.float32, device=x.device)
        check(lib().gr4hip_fir_batched_process(self._h, x.data_ptr(), x.stride(0), x.shape[1], out.data_ptr(), out.stride(0), _stream()), "FirBatched.process")
        return out[:, :x.shape[1]]


def _uncertain_id(t: torch.Tensor, who: str) -> int:
    """gr::UncertainValue<float | double> streams are float tensors of shape [n, 2]: rows of {value, uncertainty} (meta/.../UncertainValue.hpp:34-40)"""
    if t.dim() != 2 or t.shape[1] != 2 or t.dtype not in (torch.float32, torch.float64):
        raise capi.Gr4HipError(capi.INVALID_ARGUMENT, who, "UncertainValue streams are float32 / float64 tensors of shape [n, 2]")
    return capi.UF32 if t.dtype == torch.float32 else capi.UF64


def math_const(op, x: torch.Tensor, value, uncertain: bool = False) -> torch.Tensor:
    """MathOpImpl<T,op>::processOne (Math.hpp:38-56): AddConst / SubtractConst / MultiplyConst / DivideConst.
    uncertain: T = UncertainValue<float | double> -- x of shape [n, 2], value = (value, uncertainty)."""
    x = _dev(x, "math_const")
    if uncertain:
        did = _uncertain_id(x, "math_const")
        v = np.asarray(value, np.float32 if did == capi.UF32 else np.float64).reshape(2)
        n = x.shape[0]
    else:
        did = _DTYPE_ID[x.dtype]
        v = np.array([value]).astype(_NP_DTYPE[did])
        n = x.numel()
    out = torch.empty_like(x)
    check(lib().gr4hip_math_const(_OPS.get(op, op), did, x.data_ptr(), out.data_ptr(), n, v.ctypes.data, _stream()), "math_const")
    return out


def math_nary(op, inputs: Sequence[torch.Tensor], out: Optional[torch.Tensor] = None, uncertain: bool = False) -> torch.Tensor:
    """MathOpMultiPortImpl<T,op>::processBulk (Math.hpp:100-107): Add / Subtract / Multiply / Divide over n_inputs.
    `out` (optional): a contiguous device tensor of the inputs' dtype and length that receives the result.
    uncertain: T = UncertainValue<float | double>, every stream of shape [n, 2]."""
    ins = [_dev(t, "math_nary") for t in inputs]
    if not 1 <= len(ins) <= 32:
        raise capi.Gr4HipError(capi.INVALID_ARGUMENT, "math_nary", "n_inputs must be in [1, 32] (Math.hpp:90)")
    if any(t.dtype != ins[0].dtype or t.numel() != ins[0].numel() for t in ins):
        raise capi.Gr4HipError(capi.INVALID_ARGUMENT, "math_nary", "all inputs need the same dtype and length")
    ptrs = (C.c_void_p * len(ins))(*[t.data_ptr() for t in ins])
    if out is None:
        out = torch.empty_like(ins[0])
    elif not out.is_cuda or not out.is_contiguous() or out.dtype != ins[0].dtype or out.numel() != ins[0].numel():
        raise capi.Gr4HipError(capi.INVALID_ARGUMENT, "math_nary", "out must be a contiguous device tensor of the inputs' dtype and length")
    did = _uncertain_id(ins[0], "math_nary") if uncertain else _DTYPE_ID[ins[0].dtype]
    check(lib().gr4hip_math_nary(_OPS.get(op, op), did, ptrs, len(ins), out.data_ptr(), ins[0].shape[0] if uncertain else out.numel(), _stream()), "math_nary")
    return out


class Merged(_Handle):
    """A run of per-sample blocks as ONE launch: the run-time counterpart of Merge<A, "out", B, "in"> (core/.../BlockMerging.hpp:126-240) for chains of
    MathOpImpl<T, op> (Math.hpp:38-56) and Rotator<complex<float>> (Rotator.hpp:51-61; closed-form phase).  `ops`: a sequence of
    ("Add" | "Subtract" | "Multiply" | "Divide", value) and ("Rotator", phase_increment[, initial_phase]) in stream order.  2 sizeof(T) bytes of HBM traffic per
    sample whatever the length; integer types bit-exact, float ops single IEEE operations in program order (include/gr4hip.h, gr4hip_ewise_*).
    The same object is what fir_filter.set_prologue / set_epilogue take."""
    _destroy = "gr4hip_ewise_destroy"

    def __init__(self, dtype, ops=()):
        super().__init__()
        self.dtype = dtype
        self._did = _DTYPE_ID[dtype]
        check(lib().gr4hip_ewise_create(C.byref(self._h), self._did), "Merged")
        self.ops = []
        for op in ops:
            self.append(*op)

    def append(self, op, *args):
        if op == "Rotator":
            inc = float(np.float32(args[0]))
            ph0 = float(np.float32(args[1])) if len(args) > 1 else 0.0
            check(lib().gr4hip_ewise_append_rotator(self._h, inc, ph0), "Merged.append_rotator")
        else:
            v = np.array([args[0]]).astype(_NP_DTYPE[self._did])
            check(lib().gr4hip_ewise_append_const(self._h, _OPS[op], v.ctypes.data), "Merged.append_const")
        self.ops.append((op,) + tuple(args))
        return self

    def reset(self):
        check(lib().gr4hip_ewise_reset(self._h), "Merged.reset")

    @property
    def position(self) -> int:
        v = C.c_uint64(0)
        check(lib().gr4hip_ewise_position(self._h, C.byref(v)), "Merged.position")
        return v.value

    def decimate(self, x: torch.Tensor, decim: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Decimator<T> with this program's blocks behind it in ONE launch: out[m] = program(x[m * decim]) (gr4hip_ewise_decimate)"""
        x = _dev(x, "Merged.decimate")
        if x.dtype != self.dtype:
            raise capi.Gr4HipError(capi.INVALID_ARGUMENT, "Merged", f"expected {self.dtype}, got {x.dtype}")
        n_out = -(-x.numel() // int(decim))
        out = _out(out, n_out, self.dtype, x, "Merged.decimate")
        check(lib().gr4hip_ewise_decimate(self._h, x.data_ptr(), x.numel(), int(decim), out.data_ptr(), None, _stream()), "Merged.decimate")
        return out

    def process_bulk(self, x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        x = _dev(x, "Merged")
        if x.dtype != self.dtype:
            raise capi.Gr4HipError(capi.INVALID_ARGUMENT, "Merged", f"expected {self.dtype}, got {x.dtype}")
        out = _out(out, x.numel(), x.dtype, x, "Merged.process")
        check(lib().gr4hip_ewise_process(self._h, x.data_ptr(), out.data_ptr(), x.numel(), _stream()), "Merged.process")
        return out


class Rotator(_Handle):
    """gr::blocks::math::Rotator<complex<float>> (Rotator.hpp:16-63): XOR settings frequency_shift / phase_increment.
    algo: "closed_form" (default: float64 phase of sample i = carried + (i+1) inc, HBM-bound) or "recurrence" (the reference's float
    accumulator bit for bit, sequential); the carried phase is the same float member either way (include/gr4hip.h)."""
    _destroy = "gr4hip_rotator_destroy"

    def __init__(self, phase_increment: Optional[float] = None, frequency_shift: Optional[float] = None, sample_rate: float = 1.0,
                 initial_phase: float = 0.0, algo: str = "closed_form", dtype=torch.complex64):
        super().__init__()
        if phase_increment is not None and frequency_shift is not None:  # Rotator.hpp:45-46 throws
            raise ValueError("cannot set both 'frequency_shift' and 'phase_increment' in new setting (XOR)")
        self.dtype = dtype
        self._f64 = dtype == torch.complex128  # Rotator<complex<double>> (Rotator.hpp:15): closed form in float64 (csrc/f64.hip)
        if self._f64:
            self._destroy = "gr4hip_rotator64_destroy"
            self.sample_rate = float(sample_rate)
            if frequency_shift is not None:
                self.frequency_shift = float(frequency_shift)
                self.phase_increment = 2.0 * np.pi * self.frequency_shift / self.sample_rate
            else:
                self.phase_increment = float(phase_increment or 0.0)
                self.frequency_shift = self.phase_increment / (2.0 * np.pi) * self.sample_rate
            self.initial_phase, self.algo = float(initial_phase), "closed_form"
            check(lib().gr4hip_rotator64_create(C.byref(self._h), self.phase_increment, self.initial_phase), "Rotator<complex128>")
            return
        self.sample_rate = np.float32(sample_rate)
        if frequency_shift is not None:  # :41-42 (float arithmetic)
            self.frequency_shift = np.float32(frequency_shift)
            self.phase_increment = np.float32(2) * np.float32(np.float32(np.pi) * self.frequency_shift / self.sample_rate)
        else:
            self.phase_increment = np.float32(phase_increment or 0.0)
            self.frequency_shift = np.float32(self.phase_increment / (np.float32(2) * np.float32(np.pi))) * self.sample_rate
        self.initial_phase = np.float32(initial_phase)
        check(lib().gr4hip_rotator_create(C.byref(self._h), float(self.phase_increment), float(self.initial_phase)), "Rotator")
        self.set_algo(algo)

    def set_algo(self, algo: str):
        ids = {"closed_form": capi.ROTATOR_CLOSED_FORM, "recurrence": capi.ROTATOR_RECURRENCE}
        if algo not in ids:
            raise ValueError(f"unknown rotator algo '{algo}'")
        if self._f64:
            if algo != "closed_form":
                raise capi.Gr4HipError(capi.UNSUPPORTED, "Rotator", "the float64 rotator has the closed form only")
            return
        check(lib().gr4hip_rotator_set_algo(self._h, ids[algo]), "Rotator.set_algo")
        self.algo = algo

    def settings_changed(self, initial_phase: float = 0.0):
        """settingsChanged (Rotator.hpp:40-49): _accumulated_phase = initial_phase.  Host-side note; the device word (recurrence) is stored on the next call's stream."""
        if self._f64:
            check(lib().gr4hip_rotator64_reset(self._h, float(initial_phase)), "Rotator.reset")
        else:
            check(lib().gr4hip_rotator_reset(self._h, float(np.float32(initial_phase))), "Rotator.reset")
        self.initial_phase = initial_phase

    def process_bulk(self, x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        x = _dev(x, "Rotator")
        if x.dtype != self.dtype:
            raise capi.Gr4HipError(capi.INVALID_ARGUMENT, "Rotator", f"expected {self.dtype}, got {x.dtype}")
        out = _out(out, x.numel(), x.dtype, x, "Rotator.process")
        fn = lib().gr4hip_rotator64_process if self._f64 else lib().gr4hip_rotator_process
        check(fn(self._h, x.data_ptr(), out.data_ptr(), x.numel(), _stream()), "Rotator.process")
        return out

    @property
    def accumulated_phase(self) -> float:
        if self._f64:
            d = C.c_double(0)
            check(lib().gr4hip_rotator64_phase(self._h, C.byref(d)), "Rotator.phase")
            return d.value
        v = C.c_float(0)
        check(lib().gr4hip_rotator_phase(self._h, C.byref(v), _stream()), "Rotator.phase")
        return v.value


def synth_c32(n: int, seed: int = 42, tone_frel: float = 0.1, tone_amp: float = 1.0, noise_amp: float = 1.0, device="cuda") -> torch.Tensor:
    out = torch.empty(n, dtype=torch.complex64, device=device)
    check(lib().gr4hip_synth_c32(out.data_ptr(), n, seed, tone_frel, tone_amp, noise_amp, _stream()), "synth_c32")
    return out


def synth_draws(n_draws: int, seed: int, group: int) -> torch.Tensor:
    """raw 64-bit draws of the generator behind 8-sample group `group` of a synth stream (int64 tensor holding the uint64 bit patterns)"""
    out = torch.empty(n_draws, dtype=torch.int64, device="cuda")
    check(lib().gr4hip_synth_draws(out.data_ptr(), n_draws, seed & (2 ** 64 - 1), group, _stream()), "synth_draws")
    return out


def synth_f32(n: int, seed: int = 42, tone_frel: float = 0.1, tone_amp: float = 1.0, noise_amp: float = 1.0, device="cuda") -> torch.Tensor:
    out = torch.empty(n, dtype=torch.float32, device=device)
    check(lib().gr4hip_synth_f32(out.data_ptr(), n, seed, tone_frel, tone_amp, noise_amp, _stream()), "synth_f32")
    return out
