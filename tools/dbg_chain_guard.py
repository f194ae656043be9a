"""developer tool: an 8192-point Hann chain under a rejected interferer (amplitude, onset frame): worst |Y|^2 error against float64 of AUTO (strict guard), the
time-domain pair and the forced fused fast convolution, with the guard's reported ratio"""
import sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from scipy.signal import lfilter
import gnuradio4_amd as G
from gnuradio4_amd import capi
def lowpass(nt, fc):
    k = np.arange(nt); t = np.hamming(nt) * 2 * fc * np.sinc(2 * fc * (k - (nt - 1) / 2)); return (t / t.sum()).astype(np.float32)
rng = np.random.default_rng(11)
N, nt, frames, win = 8192, 100, 93, "Hann"
n = frames * N
for fc in (0.02, 0.05, 0.2):
  for ampdb in (20, 35, 50):
    for start in (0, 40 * N + 1234, 90 * N + 77):
        taps = lowpass(nt, fc)
        x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
        amp = 10 ** (ampdb / 20)
        x[start:] += (amp * np.exp(2j * np.pi * 0.41 * np.arange(n - start))).astype(np.complex64)
        y = lfilter(taps.astype(np.float64), [1.0], x.astype(np.complex128)).reshape(frames, N)
        k = np.arange(N) / (N - 1); w = 0.5 - 0.5 * np.cos(2 * np.pi * k)
        truth = np.abs(np.fft.fft(y * w, axis=1)) ** 2
        rms = np.sqrt(np.mean(truth ** 2, axis=1, keepdims=True))
        y32 = lfilter(taps, np.float32([1.0]), x).astype(np.complex64).reshape(frames, N)
        t32 = np.abs(np.fft.fft(y32.astype(np.complex128) * w, axis=1)) ** 2
        e32 = np.max(np.abs(t32 - truth) / np.maximum(truth, rms), axis=1)
        res = []
        for algo, name in ((capi.CHAIN_AUTO, "auto"), (capi.CHAIN_TIME_DOMAIN, "td"), (capi.CHAIN_FUSED_FD, "fd")):
            ch = G.Chain(taps, N, win, algo)
            got = ch.process_bulk(torch.from_numpy(x).cuda()).cpu().numpy().reshape(frames, N)
            e = np.max(np.abs(got - truth) / np.maximum(truth, rms), axis=1)
            extra = ""
            if algo == capi.CHAIN_AUTO:
                r, td = ch.last_power_ratio(); extra = f" ratio={r:.3g} td={td}"
            res.append(f"{name}: max {e.max():.2e} at frame {int(e.argmax())}{extra}")
        print(f"fc={fc} amp={ampdb}dB start_frame={start // N}: f32 cpu {e32.max():.2e} | " + " | ".join(res), flush=True)
