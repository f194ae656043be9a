"""developer tool: the fused fast convolution's error (guard OFF: explicit CHAIN_FUSED_FD) per frame against float64, binned by the frame's fourth-moment statistic R4 --
where kGuardR4Max can sit.  Narrow filters over white noise, with and without weak rejected tones (the strong-line cases belong to the second statistic and are left out: T' < 2000)."""
import sys
sys.path.insert(0, "tests")
import numpy as np, torch
from scipy.signal import lfilter
import gnuradio4_amd as G
from gnuradio4_amd import capi
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
N, frames = 8192, 600
edges = [0, 4, 8, 12, 16, 20, 24, 32, 48, 64, 1e9]
worst = np.zeros(len(edges) - 1); count = np.zeros(len(edges) - 1, int); kworst = np.zeros(len(edges) - 1)
def lowpass(nt, fc):
    k = np.arange(nt); t = np.hamming(nt) * 2 * fc * np.sinc(2 * fc * (k - (nt - 1) / 2)); return (t / t.sum()).astype(np.float32)
for win, wid in (("None", 0), ("Hann", 3), ("BlackmanHarris", 7)):
    w32 = np.empty(N, np.float32); capi.check(capi.lib().gr4hip_window_create(wid, w32.ctypes.data, N, 1.6), "window")
    w = w32.astype(np.float64) if wid else np.ones(N); w2 = float(np.mean(w * w)); wg = float(np.mean(w))
    for nt in (256, 129, 64):
        for fc in (0.0025, 0.005, 0.0075, 0.01, 0.015):
            for amp in (0.0, 1.0, 2.0, 4.0, 8.0, 16.0):
                taps = lowpass(nt, fc)
                n = frames * N
                x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
                if amp:  # a wide-band neighbour: noise of `amp` x the stream's rms in a band of 5 .. 25 % of fs somewhere outside the pass band, on and off in bursts half of the time
                    bw = float(rng.uniform(0.05, 0.25)); f0 = float(rng.uniform(0.05 + bw / 2, 0.5 - bw / 2)) * (1 if rng.random() < 0.5 else -1)
                    v = (rng.standard_normal(n) + 1j * rng.standard_normal(n))
                    V = np.fft.fft(v); fr = np.fft.fftfreq(n); V[np.abs(fr - f0) > bw / 2] = 0; v = np.fft.ifft(V); v *= amp * np.sqrt(2.0) / np.sqrt(np.mean(np.abs(v) ** 2))
                    if rng.random() < 0.5: v *= (np.sin(2 * np.pi * np.arange(n) / float(rng.uniform(3, 40) * N)) > 0)
                    x += v.astype(np.complex64)
                if rng.random() < 0.3: x *= np.float32(10 ** float(rng.uniform(-3, 3)))
                y = lfilter(taps.astype(np.float64), [1.0], x.astype(np.complex128)).reshape(frames, N)
                T = np.abs(np.fft.fft(y * w, axis=1)) ** 2
                rms = np.sqrt(np.mean(T ** 2, axis=1))
                got = G.Chain(taps, N, win, capi.CHAIN_FUSED_FD).process_bulk(torch.from_numpy(x).cuda()).cpu().numpy().astype(np.float64).reshape(frames, N)
                e = np.max(np.abs(got - T) / np.maximum(T, rms[:, None]), axis=1)
                xf = x.reshape(frames, N)
                pin = np.mean(np.abs(xf) ** 2, axis=1)
                r4 = w2 * N * pin / rms
                X = np.fft.fft(xf, axis=1); pk = np.maximum(np.abs(X.real), np.abs(X.imag)).max(axis=1)
                tp = 2 * wg * wg * pk ** 2 / rms
                ok = tp < 2000
                for i in range(len(edges) - 1):
                    m = ok & (r4 >= edges[i]) & (r4 < edges[i + 1])
                    if m.any():
                        count[i] += int(m.sum()); worst[i] = max(worst[i], float(e[m].max())); kworst[i] = max(kworst[i], float((e[m] / np.sqrt(r4[m])).max()))
    print(win, "done", flush=True)
for i in range(len(edges) - 1):
    print(f"R4 in [{edges[i]}, {edges[i + 1]}): {count[i]} frames, worst err {worst[i]:.3g}, worst K = err / sqrt(R4) {kworst[i]:.3g}")
