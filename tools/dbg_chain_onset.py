"""developer tool: the frame in which a strong tone sets in late (where the window is small) through the fused chain: error of AUTO / forced FD against float64, guard verdict"""
import sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from scipy.signal import lfilter
import gnuradio4_amd as G
from gnuradio4_amd import capi
rng = np.random.default_rng(7)
N, frames = 8192, 40
n = frames * N
k = np.arange(N) / (N - 1)
w = 0.35875 - 0.48829 * np.cos(2 * np.pi * k) + 0.14128 * np.cos(4 * np.pi * k) - 0.01168 * np.cos(6 * np.pi * k)
t = np.float32([0.5, 0.5])
for ampdb in (30, 40, 50):
    for off in (4096, 6600, 7400, 7900, 8150):
        start = 33 * N + off; amp = 10 ** (ampdb / 20)
        x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
        x[start:start + (N - off)] += (amp * np.exp(2j * np.pi * 0.41 * np.arange(N - off))).astype(np.complex64)  # the tone lives in the rest of frame 33 only
        y = lfilter(t.astype(np.float64), [1.0], x.astype(np.complex128)).reshape(frames, N)
        truth = np.abs(np.fft.fft(y * w, axis=1)) ** 2
        rms = np.sqrt(np.mean(truth ** 2, axis=1, keepdims=True))
        fr_in = np.mean(np.abs(x.reshape(frames, N)[33]) ** 2); fr_out = truth[33].sum() / (N * np.mean(w ** 2) * N)
        line = f"amp {ampdb} dB from sample {off} of frame 33: frame ratio {fr_out / fr_in:.3g}"
        for algo, name in ((capi.CHAIN_AUTO, "auto"), (capi.CHAIN_FUSED_FD, "fd")):
            ch = G.Chain(t, N, "BlackmanHarris", algo)
            got = ch.process_bulk(torch.from_numpy(x).cuda()).cpu().numpy().reshape(frames, N)
            e = np.abs(got - truth) / np.maximum(truth, rms)
            fr, b = np.unravel_index(e.argmax(), e.shape)
            line += f" | {name} {e.max():.2e} (frame {fr})"
            if algo == capi.CHAIN_AUTO: line += f" ratio, td = {ch.last_power_ratio()}"
        print(line, flush=True)
