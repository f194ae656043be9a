"""developer tool: windowed 8192-point chains with and without a strong passed / rejected tone: fused fast convolution, time-domain pair and unfused kernels against
float64 and a float32 CPU chain (where the worst bin sits and how large the truth is there)"""
import sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from scipy.signal import lfilter
import gnuradio4_amd as G
from gnuradio4_amd import capi
rng = np.random.default_rng(2)
N, frames = 8192, 40
n = frames * N
k = np.arange(N) / (N - 1)
wins = {"BlackmanHarris": 0.35875 - 0.48829 * np.cos(2 * np.pi * k) + 0.14128 * np.cos(4 * np.pi * k) - 0.01168 * np.cos(6 * np.pi * k), "Hann": 0.5 - 0.5 * np.cos(2 * np.pi * k)}
for nt in (2, 17):
    t = np.hamming(nt); t = (t / t.sum()).astype(np.float32)
    for win in ("BlackmanHarris", "Hann"):
        for amp in (0.0, 100.0):
            x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
            x += (amp * np.exp(2j * np.pi * 0.41 * np.arange(n))).astype(np.complex64)
            y = lfilter(t.astype(np.float64), [1.0], x.astype(np.complex128)).reshape(frames, N)
            truth = np.abs(np.fft.fft(y * wins[win], axis=1)) ** 2
            rms = np.sqrt(np.mean(truth ** 2, axis=1, keepdims=True))
            # float32 everything on the CPU: filter, window, transform
            y32 = lfilter(t, np.float32([1.0]), x).astype(np.complex64).reshape(frames, N) * wins[win].astype(np.float32)
            t32 = np.abs(np.fft.fft(y32.astype(np.complex64), axis=1).astype(np.complex64)) ** 2
            e32 = np.abs(t32 - truth) / np.maximum(truth, rms)
            line = f"taps={nt} {win} amp={amp}: cpu f32 (filter+window in f32) {e32.max():.2e}"
            for algo, name in ((capi.CHAIN_FUSED_FD, "fd"), (capi.CHAIN_TIME_DOMAIN, "td"), (capi.CHAIN_UNFUSED, "unfused")):
                got = G.Chain(t, N, win, algo).process_bulk(torch.from_numpy(x).cuda()).cpu().numpy().reshape(frames, N)
                e = np.abs(got - truth) / np.maximum(truth, rms)
                fr, b = np.unravel_index(e.argmax(), e.shape)
                line += f" | {name} {e.max():.2e} (frame {fr} bin {b}, truth/rms {truth[fr, b] / rms[fr, 0]:.1e})"
            print(line, flush=True)
