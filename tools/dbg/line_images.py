"""developer tool: err / sqrt(T) of the fused fast convolution (guard OFF) on a line the filter only dents, over many line frequencies / levels / windows / sizes:
where kGuardPeakMax can sit.  T = wg^2 |X_peak|^2 / rms_k(|Y_k|^2)."""
import sys
sys.path.insert(0, "tests")
import numpy as np, torch
from scipy.signal import lfilter
import gnuradio4_amd as G
from gnuradio4_amd import capi
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
res = {}
for N in (8192, 4096, 1024, 256):
    frames = 24 * 8192 // N
    for win, wid in (("None", 0), ("Hann", 3), ("BlackmanHarris", 7)):
        w32 = np.empty(N, np.float32); capi.check(capi.lib().gr4hip_window_create(wid, w32.ctypes.data, N, 1.6), "window")
        w = w32.astype(np.float64) if wid else np.ones(N); wg = float(np.mean(w))
        kw = 0.0; ew = 0.0; tw = 0.0
        for trial in range(40 if N == 8192 else 16):
            kind = trial % 3
            n = frames * N
            x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
            if kind == 0:   # 2-tap average, line near its null at fs / 2
                taps = np.array([0.5, 0.5], np.float32); f0 = 0.5 + float(rng.uniform(0.01, 0.06)) * (1 if rng.random() < 0.5 else -1)
            elif kind == 1:  # 65-tap low-pass, line in its transition band
                nt, fc = 65, 0.05
                k = np.arange(nt); t = np.hamming(nt) * 2 * fc * np.sinc(2 * fc * (k - (nt - 1) / 2)); taps = (t / t.sum()).astype(np.float32); f0 = float(rng.uniform(0.06, 0.085))
            else:            # 17-tap low-pass: a wide transition band
                nt, fc = 17, 0.1
                k = np.arange(nt); t = np.hamming(nt) * 2 * fc * np.sinc(2 * fc * (k - (nt - 1) / 2)); taps = (t / t.sum()).astype(np.float32); f0 = float(rng.uniform(0.15, 0.3))
            amp = float(10 ** rng.uniform(0.0, 1.5))
            x += (amp * np.exp(2j * np.pi * (f0 * np.arange(n) + rng.random()))).astype(np.complex64)
            y = lfilter(taps.astype(np.float64), [1.0], x.astype(np.complex128)).reshape(frames, N)
            T = np.abs(np.fft.fft(y * w, axis=1)) ** 2
            rms = np.sqrt(np.mean(T ** 2, axis=1))
            got = G.Chain(taps, N, win, capi.CHAIN_FUSED_FD).process_bulk(torch.from_numpy(x).cuda()).cpu().numpy().astype(np.float64).reshape(frames, N)
            e = np.max(np.abs(got - T) / np.maximum(T, rms[:, None]), axis=1)[1:]
            X = np.fft.fft(x.reshape(frames, N), axis=1); tt = (wg * np.abs(X).max(axis=1)) ** 2 / rms
            tt = tt[1:]
            m = tt > 200  # (below that the spread error carries the frame)
            if m.any():
                kk = e[m] / np.sqrt(tt[m]); i = int(np.argmax(kk))
                if kk[i] > kw: kw, ew, tw = float(kk[i]), float(e[m][i]), float(tt[m][i])
        print(f"N={N} {win:14s}: worst err / sqrt(T) {kw:.3g} (err {ew:.3g} at T {tw:.4g})", flush=True)
