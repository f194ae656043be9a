"""developer tool: SEVERAL lines the filter only dents, each just below the guard's line limit (T' ~ 0.5 .. 0.95 of 2000): their images add in power, the kernel's statistic sees
the strongest alone.  CHAIN_AUTO, 8192 points, a 2-tap average with the lines near fs / 2 / a 65-tap low-pass with the lines in its transition band."""
import sys
sys.path.insert(0, "tests")
import numpy as np, torch
from scipy.signal import lfilter
import gnuradio4_amd as G
from gnuradio4_amd import capi
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 3)
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 20
N, frames = 8192, 16; n = frames * N
def lowpass(nt, fc):
    k = np.arange(nt); t = np.hamming(nt) * 2 * fc * np.sinc(2 * fc * (k - (nt - 1) / 2)); return (t / t.sum()).astype(np.float32)
for win, wid in (("None", 0), ("Hann", 3)):
    w32 = np.empty(N, np.float32); capi.check(capi.lib().gr4hip_window_create(wid, w32.ctypes.data, N, 1.6), "window")
    w = w32.astype(np.float64) if wid else np.ones(N); wg = float(np.mean(w))
    for L in (1, 2, 4, 8, 16):
        worst = 0.0; marked = 0.0; bad = 0
        for trial in range(trials):
            kind = trial % 2
            if kind == 0: taps = np.array([0.5, 0.5], np.float32); fs = 0.5 + rng.uniform(0.02, 0.06, L) * rng.choice([-1, 1], L)
            else: taps = lowpass(65, 0.05); fs = rng.uniform(0.062, 0.08, L)
            noise = (rng.standard_normal(n) + 1j * rng.standard_normal(n))
            tones = sum(np.exp(2j * np.pi * (f * np.arange(n) + rng.random())) for f in fs)
            # scale the lines so that the strongest one's T' is `aim` x 2000 (bisection on the amplitude: T' depends on the output's rms)
            aim = float(rng.uniform(0.5, 0.95)); lo, hi = 0.01, 100.0
            for _ in range(30):
                amp = np.sqrt(lo * hi)
                x = (noise + amp * tones).astype(np.complex64)
                y = lfilter(taps.astype(np.float64), [1.0], x[:2 * N].astype(np.complex128)).reshape(2, N)[1]
                T1 = np.abs(np.fft.fft(y * w)) ** 2; rms1 = np.sqrt(np.mean(T1 ** 2))
                X1 = np.fft.fft(x[N:2 * N]); tp = 2 * wg * wg * max(np.abs(X1.real).max(), np.abs(X1.imag).max()) ** 2 / rms1
                if tp > aim * 2000: hi = amp
                else: lo = amp
            y = lfilter(taps.astype(np.float64), [1.0], x.astype(np.complex128)).reshape(frames, N)
            T = np.abs(np.fft.fft(y * w, axis=1)) ** 2
            rms = np.sqrt(np.mean(T ** 2, axis=1, keepdims=True))
            ch = G.Chain(taps, N, win)
            got = ch.process_bulk(torch.from_numpy(x).cuda()).cpu().numpy().astype(np.float64).reshape(frames, N)
            e = float(np.max((np.abs(got - T) / np.maximum(T, rms))[1:]))
            worst = max(worst, e); marked += ch.last_guard_fractions()[0]; bad += e > 1e-5
        print(f"{win:5s} {L:2d} lines: {trials} streams, {bad} above 1e-5, worst {worst:.3g}, marked fraction {marked / trials:.2f}", flush=True)
