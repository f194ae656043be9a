"""developer tool: CHAIN_AUTO at extreme stream levels (the guard's sums of fourth powers leave float32's range): error against float64, marked / float64 fractions"""
import sys
sys.path.insert(0, "tests")
import numpy as np, torch
from scipy.signal import lfilter
import gnuradio4_amd as G
from gnuradio4_amd import capi
rng = np.random.default_rng(3)
def lowpass(nt, fc):
    k = np.arange(nt); t = np.hamming(nt) * 2 * fc * np.sinc(2 * fc * (k - (nt - 1) / 2)); return (t / t.sum()).astype(np.float32)
for N, win in ((8192, "None"), (8192, "Hann"), (1024, "Hann")):
    frames = 24 * 8192 // N; n = frames * N
    w32 = np.empty(N, np.float32); capi.check(capi.lib().gr4hip_window_create({"None": 0, "Hann": 3}[win], w32.ctypes.data, N, 1.6), "window")
    w = w32.astype(np.float64) if win != "None" else np.ones(N)
    for fc, amp in ((0.1, 0.0), (0.005, 0.0), (0.02, 30.0)):
        taps = lowpass(256, fc)
        base = (rng.standard_normal(n) + 1j * rng.standard_normal(n)) + amp * np.exp(2j * np.pi * 0.31 * np.arange(n))
        for lvl in (1e-17, 1e-12, 1e-6, 1.0, 1e6, 1e9, 1e12, 1e17):
            x = (base * lvl).astype(np.complex64)
            y = lfilter(taps.astype(np.float64), [1.0], x.astype(np.complex128)).reshape(frames, N)
            T = np.abs(np.fft.fft(y * w, axis=1)) ** 2
            ch = G.Chain(taps, N, win)
            got = ch.process_bulk(torch.from_numpy(x).cuda()).cpu().numpy().astype(np.float64).reshape(frames, N)
            rms = np.sqrt(np.mean(T ** 2, axis=1, keepdims=True))
            ok = np.isfinite(T).all() and T.max() < 3e38 and rms.min() > 1e-37
            e = float(np.max(np.abs(got - T) / np.maximum(T, rms))) if ok else float("nan")
            m, f = ch.last_guard_fractions()
            print(f"N={N} {win:5s} fc {fc} tone {amp} level {lvl:g}: err {e:.3g} marked {m:.2f} float64 {f:.2f}" + ("" if ok else "   (|Y|^2 outside float32's normal range: not judged)"), flush=True)
