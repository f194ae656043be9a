import sys
sys.path.insert(0, "tests")
import numpy as np, torch
from scipy.signal import lfilter
import gnuradio4_amd as G
from gnuradio4_amd import capi
d = np.load(sys.argv[1]); x, taps, N, win, frames = d["x"], d["taps"], int(d["N"]), str(d["win"]), int(d["frames"])
print("taps", taps[:4], "N", N, win)
y = lfilter(taps.astype(np.float64), [1.0], x.astype(np.complex128)).reshape(frames, N)
w32 = np.empty(N, np.float32); capi.check(capi.lib().gr4hip_window_create({"Hann": 3, "Hamming": 2, "BlackmanHarris": 7, "None": 0}[win], w32.ctypes.data, N, 1.6), "window"); w = w32.astype(np.float64) if win != "None" else np.ones(N)
truth = np.abs(np.fft.fft(y * w, axis=1)) ** 2
rms = np.sqrt(np.mean(truth ** 2, axis=1, keepdims=True))
xd = torch.from_numpy(x).cuda()
gf = G.Chain(taps, N, win, capi.CHAIN_FUSED_FD).process_bulk(xd).cpu().numpy().reshape(frames, N).astype(np.float64)
e = np.abs(gf - truth) / np.maximum(truth, rms)
ef = e.max(axis=1)
print("per-frame err percentiles:", np.percentile(ef, [50, 90, 99, 100]))
X = np.fft.fft(x.reshape(frames, N), axis=1)
for fr in list(np.argsort(-ef)[:4]) + [10]:
    k = int(np.argmax(e[fr])); top = np.argsort(-truth[fr])[:3]
    print(f"frame {fr}: err {ef[fr]:.3g} at bin {k}: truth {truth[fr][k]:.4g} rms {rms[fr][0]:.4g} got {gf[fr][k]:.6g}; top bins {top.tolist()} truth {truth[fr][top[0]]:.4g}; |X| at worst bin {abs(X[fr][k]):.4g}, max|X| {np.abs(X[fr]).max():.4g} at {int(np.argmax(np.abs(X[fr])))}, median|X| {np.median(np.abs(X[fr])):.4g}")
    # signed relative error around the worst bin
    km = int(np.argmax(np.abs(X[fr]))); wg = float(np.mean(w)); w2 = float(np.mean(w * w))
    pin = float(np.mean(np.abs(x.reshape(frames, N)[fr]) ** 2))
    print(f"   offset from the peak bin {(k - km) % N} (N/{N / max(1, (k - km) % N):.3g}); R4 {w2 * N * pin / rms[fr][0]:.3g}; peak statistic wg^2 Xmax^2/rms {wg * wg * np.abs(X[fr]).max() ** 2 / rms[fr][0]:.4g}; err/sqrt(that) {ef[fr] / np.sqrt(wg * wg * np.abs(X[fr]).max() ** 2 / rms[fr][0]):.3g}")
    print("   rel err around:", " ".join(f"{(gf[fr][j]-truth[fr][j])/max(truth[fr][j], rms[fr][0]):+.2e}" for j in range(k - 3, k + 4)))
