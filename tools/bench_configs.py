#!/usr/bin/env python
"""Secondary configurations of BASELINE.json (parity-test cases, not the bench line): timings on one MI355X.
  configs[2]: polyphase decimating FIR (decim 8, 1024 taps) + IIR biquad x4 cascade      (float stream)
  configs[3]: batched 64-channel x 256-tap FIR on the matrix pipe (two-term f16 splits under a block exponent since round 4; the three-term bf16 kernel beside it)
  plus the stand-alone blocks (math, FFT block, fir_filter) for the roofline table in DESIGN.md
usage: bench_configs.py [--json out.json]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _timing import steady
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gnuradio4_amd as G  # noqa: E402
from gnuradio4_amd import capi  # noqa: E402


def timeit(fn, reps=None, warm=None):
    return steady(fn)  # back to back at settled clocks (tools/_timing.py)


def lowpass(ntaps, fc):
    w = np.empty(ntaps, np.float32)
    capi.check(capi.lib().gr4hip_window_create(2, w.ctypes.data, ntaps, 1.6), "window")
    k = np.arange(ntaps, dtype=np.float64)
    t = w.astype(np.float64) * 2 * fc * np.sinc(2 * fc * (k - (ntaps - 1) / 2.0))
    return (t / t.sum()).astype(np.float32)


res = {}
# ---- configs[2]
n = 1 << 27
x = G.synth_f32(n, seed=42)
fir = G.fir_filter(lowpass(1024, 0.05), torch.float32, decimate=8)
b, a = G.blocks.design_iir(capi.LOWPASS, 8, 0.05, float("nan"), 1.0, capi.BUTTERWORTH)  # Butterworth order 8 -> 4 biquads, fc = 0.05 fs (decimated)
iir = G.iir_filter(b, a)
yd = torch.empty(n // 8, dtype=torch.float32, device="cuda")
yo = torch.empty_like(yd)
t_fir = timeit(lambda: fir.process_bulk(x, yd))
t_iir = timeit(lambda: iir.process_bulk(yd, yo))
res["configs[2] decim-8 1024-tap FIR + 4 biquads"] = {
    "Msamples/s (input rate, both kernels)": round(n / (t_fir + t_iir) / 1e6, 1), "fir_ms": round(t_fir * 1e3, 3), "iir_ms": round(t_iir * 1e3, 3),
    "fir_Msamples/s (input)": round(n / t_fir / 1e6, 1), "alg_GB/s": round(n * 5.5 / (t_fir + t_iir) / 1e9, 1), "hbm_frac": round(n * 5.5 / (t_fir + t_iir) / 8e12, 3),
    "fir_direct_form_equivalent_TFLOP/s": round(n / 8 * 2048 / t_fir / 1e12, 1),
    "note": "5.5 B/input sample: 4 in + 0.5 decimated stream written + 0.5 read + 0.5 out; since late round 4 the FIR runs in band form on the f16 matrix pipe "
            "(csrc/fir_decim_f16.hip); the frequency-domain kernel of rounds 1-3 (GR4HIP_FIR_NO_DECIM_F16=1) ran this at 530"}
capi.developer_switch("GR4HIP_FIR_NO_DECIM_FD", 1)
fir_p = G.fir_filter(lowpass(1024, 0.05), torch.float32, decimate=8)
t_firp = timeit(lambda: fir_p.process_bulk(x, yd))
capi.developer_switch("GR4HIP_FIR_NO_DECIM_FD", 0)
res["configs[2] with the polyphase MFMA decimator (round-1 path)"] = {"Msamples/s (input rate, both kernels)": round(n / (t_firp + t_iir) / 1e6, 1), "fir_ms": round(t_firp * 1e3, 3),
                                                                      "fir_TFLOP/s": round(n / 8 * 2048 / t_firp / 1e12, 1)}
del x, yd, yo
# ---- configs[3]
nch, ntaps, n = 64, 256, 1 << 22
xb = torch.stack([G.synth_f32(n, seed=42 + c) for c in range(nch)])
rng = np.random.default_rng(0)
fb = G.FirBatched(np.stack([lowpass(ntaps, 0.05 + 0.005 * c) for c in range(nch)]))
yb = torch.empty_like(xb)
t = timeit(lambda: fb.process_bulk(xb, yb))
res["configs[3] 64 ch x 256-tap FIR (f16 matrix pipe, two-term splits under a block exponent)"] = {"Msamples/s (all channels)": round(nch * n / t / 1e6, 1), "ms": round(t * 1e3, 3), "alg_GB/s": round(nch * n * 8 / t / 1e9, 1),
                                               "hbm_frac": round(nch * n * 8 / t / 8e12, 3), "float32_equivalent_TFLOP/s": round(nch * n * 512 / t / 1e12, 1),
                                               "executed_f16_TFLOP/s": round(nch * n * 3 * 2 * 288 / t / 1e12, 1), "f16_mfma_peak_frac": round(nch * n * 3 * 2 * 288 / t / 2.5e15, 3),
                                               "note": "three f16 MFMAs per 32-sample K-step of the 288-sample window (fir_f16.hip, round 4), every segment judged in the kernel; the three-term bf16 kernel (six MFMAs, GR4HIP_FIR_NO_F16X2=1) is the row below, the f32 MFMA kernel (GR4HIP_FIR_NO_BF16X3=1) ran this at 251 Gsamples/s"}
capi.developer_switch("GR4HIP_FIR_NO_F16X2", 1)
fb3 = G.FirBatched(np.stack([lowpass(ntaps, 0.05 + 0.005 * c) for c in range(nch)]))
t = timeit(lambda: fb3.process_bulk(xb, yb))
capi.developer_switch("GR4HIP_FIR_NO_F16X2", 0)
res["configs[3] on the three-term bf16 kernel (rounds 2-3)"] = {"Msamples/s (all channels)": round(nch * n / t / 1e6, 1), "ms": round(t * 1e3, 3), "hbm_frac": round(nch * n * 8 / t / 8e12, 3),
                                                               "executed_bf16_TFLOP/s": round(nch * n * 6 * 2 * 288 / t / 1e12, 1)}
del fb3
# the same work on the VALU kernel (one fir_filter at a time)
f1 = G.fir_filter(lowpass(ntaps, 0.05), torch.float32)
x1 = xb.reshape(-1)  # one long real stream
y1 = torch.empty_like(x1)
t1 = timeit(lambda: f1.process_bulk(x1, y1))
res["fir_filter<float> 256 taps (single stream, MFMA kernel)"] = {"Msamples/s": round(x1.numel() / t1 / 1e6, 1), "useful_TFLOP/s": round(x1.numel() * 512 / t1 / 1e12, 1)}
f1 = G.fir_filter(lowpass(64, 0.05), torch.float32)
t1 = timeit(lambda: f1.process_bulk(x1, y1))
res["fir_filter<float> 64 taps (single stream, MFMA kernel)"] = {"Msamples/s": round(x1.numel() / t1 / 1e6, 1), "alg_GB/s": round(x1.numel() * 8 / t1 / 1e9, 1), "hbm_frac": round(x1.numel() * 8 / t1 / 8e12, 3)}
f1 = G.fir_filter(lowpass(32, 0.05), torch.float32)
t1 = timeit(lambda: f1.process_bulk(x1, y1))
res["fir_filter<float> 32 taps (VALU register-window kernel)"] = {"Msamples/s": round(x1.numel() / t1 / 1e6, 1), "alg_GB/s": round(x1.numel() * 8 / t1 / 1e9, 1), "hbm_frac": round(x1.numel() * 8 / t1 / 8e12, 3)}
del x1, y1
# complex<float> fir_filter, 256 taps, long input: frequency-domain path (2 transforms per 8192-sample frame), 16 B/sample
nc = 1 << 27
xcf = G.synth_c32(nc)
ycf = torch.empty(nc, dtype=torch.complex64, device="cuda")
fc = G.fir_filter(lowpass(256, 0.05), torch.complex64)
tc = timeit(lambda: fc.process_bulk(xcf, ycf))
res["fir_filter<complex<float>> 256 taps (fast convolution)"] = {"Msamples/s": round(nc / tc / 1e6, 1), "alg_GB/s": round(nc * 16 / tc / 1e9, 1), "hbm_frac": round(nc * 16 / tc / 8e12, 3),
                                                                 "direct_form_equivalent_TFLOP/s": round(nc * 1024 / tc / 1e12, 1)}
del xcf, ycf
del xb, yb
# ---- stand-alone blocks
n = 1 << 28
xi = torch.randint(-1000, 1000, (n,), dtype=torch.int32, device="cuda")
t = timeit(lambda: G.math_const("Multiply", xi, 3))
res["MultiplyConst<int32>"] = {"Msamples/s": round(n / t / 1e6, 1), "alg_GB/s": round(n * 8 / t / 1e9, 1), "hbm_frac": round(n * 8 / t / 8e12, 3)}
ins = [xi, xi + 1, xi + 2, xi + 3]
t = timeit(lambda: G.math_nary("Add", ins))
res["Add<int32> n_inputs=4"] = {"Msamples/s": round(n / t / 1e6, 1), "alg_GB/s": round(n * 20 / t / 1e9, 1), "hbm_frac": round(n * 20 / t / 8e12, 3)}
del xi, ins
n = 1 << 27
xc = G.synth_c32(n)
m2 = torch.empty(n, dtype=torch.float32, device="cuda")
sp = torch.empty(n, dtype=torch.complex64, device="cuda")
for N in (256, 1024, 4096, 8192):
    F = G.FFT(N, "Hann")
    t = timeit(lambda: F.mag2(xc, m2.view(n // N, N)))
    res[f"FFT block {N} (Hann) -> mag2"] = {"Msamples/s": round(n / t / 1e6, 1), "alg_GB/s": round(n * 12 / t / 1e9, 1), "hbm_frac": round(n * 12 / t / 8e12, 3)}
    t = timeit(lambda: F.spectrum(xc, sp.view(n // N, N)))
    res[f"FFT block {N} (Hann) -> complex spectrum"] = {"Msamples/s": round(n / t / 1e6, 1), "alg_GB/s": round(n * 16 / t / 1e9, 1), "hbm_frac": round(n * 16 / t / 8e12, 3)}
F = G.FFT(1024, "Hann")
t = timeit(lambda: F.process_bulk(xc))
res["FFT block 1024 (Hann) -> DataSet (mag, phase, re, im, ranges)"] = {"Msamples/s": round(n / t / 1e6, 1), "alg_GB/s": round(n * 24 / t / 1e9, 1), "hbm_frac": round(n * 24 / t / 8e12, 3),
                                                                         "note": "24 B/sample: 8 in + 4 x 4 out; per-frame min/max ranges reduced inside the kernel; includes torch output allocation"}
# chain at the FFT block's default size (<= 64 taps: fused time-domain kernel -- direct-form filter + one transform per frame in one launch), Decimator, Rotator
ch = G.Chain(lowpass(64, 0.1), 1024, "Hann")
m2 = torch.empty((n // 1024, 1024), dtype=torch.float32, device="cuda")
t = timeit(lambda: ch.process_bulk(xc, m2))
res["chain complex 64-tap FIR -> 1024-pt FFT (Hann) -> mag2 (fused time domain, AUTO)"] = {"Msamples/s": round(n / t / 1e6, 1), "alg_GB/s": round(n * 12 / t / 1e9, 1), "hbm_frac": round(n * 12 / t / 8e12, 3),
                                                                               "note": "one launch, traffic = algorithmic (the unfused pair moved 28 B/sample: y written and re-read)"}
chf = G.Chain(lowpass(64, 0.1), 1024, "Hann", G.capi.CHAIN_FUSED_FD)
t = timeit(lambda: chf.process_bulk(xc, m2))
res["chain complex 64-tap FIR -> 1024-pt FFT (Hann) -> mag2 (fused fast convolution, fftSize < 8192)"] = {"Msamples/s": round(n / t / 1e6, 1), "alg_GB/s": round(n * 12 / t / 1e9, 1), "hbm_frac": round(n * 12 / t / 8e12, 3)}
# interpolating FIR: x8 with 256 taps (block-Toeplitz on the f32 matrix pipe), outputs per second
itp = G.fir_interpolator(lowpass(256, 0.05), 8)
xi8 = xc.view(torch.float32)[: 1 << 23]
yi8 = torch.empty(xi8.numel() * 8, dtype=torch.float32, device="cuda")
t = timeit(lambda: itp.process_bulk(xi8, yi8))
res["fir_interpolator<float> x8, 256 taps"] = {"Moutputs/s": round(yi8.numel() / t / 1e6, 1), "alg_GB/s": round(yi8.numel() * 4.5 / t / 1e9, 1), "hbm_frac": round(yi8.numel() * 4.5 / t / 8e12, 3)}
del yi8
cf64 = G.fir_filter(lowpass(64, 0.1), torch.complex64)
t = timeit(lambda: cf64.process_bulk(xc, sp))
res["fir_filter<complex<float>> 64 taps (direct form on the matrix pipe)"] = {"Msamples/s": round(n / t / 1e6, 1), "alg_GB/s": round(n * 16 / t / 1e9, 1), "hbm_frac": round(n * 16 / t / 8e12, 3)}
xr = xc.view(torch.float32)
dec = G.Decimator(10)
t = timeit(lambda: dec.process_bulk(xr))
res["Decimator<float> decim 10"] = {"Msamples/s (input)": round(xr.numel() / t / 1e6, 1), "note": "strided 4-byte reads: every input cache line is touched, 1/10 of it used"}
# the one filter the reference publishes a number for: 1-pole IIR low-pass y[n] = (1 - a) x[n] + a y[n-1] (docs/USER_API_Connecting_Blocks.md:209-222:
# 994 k/s through a feedback edge, 113 M/s merged, 656 M/s as a constexpr loop, unstated CPU)
pole = G.iir_filter([[0.05]], [[1.0, -0.95]])
yp = torch.empty_like(xr)
t = timeit(lambda: pole.process_bulk(xr, yp))
res["1-pole IIR low-pass (reference FeedbackMerge benchmark shape)"] = {"Msamples/s": round(xr.numel() / t / 1e6, 1), "alg_GB/s": round(xr.numel() * 8 / t / 1e9, 1),
                                                                       "reference_published": "113 M/s merged, 656 M/s constexpr (unstated CPU)"}
rot = G.Rotator(0.6283)
yr = torch.empty_like(xc)
t = timeit(lambda: rot.process_bulk(xc))
res["Rotator<complex<float>> (closed-form float64 phase, default)"] = {"Msamples/s": round(n / t / 1e6, 1), "alg_GB/s": round(n * 16 / t / 1e9, 1), "hbm_frac": round(n * 16 / t / 8e12, 3),
                                                                       "note": "includes torch output allocation"}
rot = G.Rotator(0.6283, algo="recurrence")
nr = 1 << 24
t = timeit(lambda: rot.process_bulk(xc[:nr]), reps=2)
res["Rotator<complex<float>> (bit-exact float phase recurrence)"] = {"Msamples/s": round(nr / t / 1e6, 1), "note": "bounded by the single sequential phase chain of Rotator.hpp:51-61, reproduced exactly"}
print(json.dumps(res, indent=1))
if "--json" in sys.argv:
    json.dump(res, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)
