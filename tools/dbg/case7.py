import sys
sys.path.insert(0, "tests")
import numpy as np, torch
from scipy.signal import lfilter
import gnuradio4_amd as G
from gnuradio4_amd import capi
d = np.load(sys.argv[1]); x, taps, N, win, frames = d["x"], d["taps"], int(d["N"]), str(d["win"]), int(d["frames"])
y = lfilter(taps.astype(np.float64), [1.0], x.astype(np.complex128)).reshape(frames, N)
w32 = np.empty(N, np.float32); capi.check(capi.lib().gr4hip_window_create({"Hann": 3, "Hamming": 2, "BlackmanHarris": 7, "None": 0}[win], w32.ctypes.data, N, 1.6), "window"); w = w32.astype(np.float64)
truth = np.abs(np.fft.fft(y * w, axis=1)) ** 2
rms = np.sqrt(np.mean(truth ** 2, axis=1, keepdims=True))
xd = torch.from_numpy(x).cuda()
def err(got): return np.max(np.abs(got.reshape(frames, N) - truth) / np.maximum(truth, rms), axis=1)
a = G.Chain(taps, N, win, capi.CHAIN_AUTO); ga = a.process_bulk(xd).cpu().numpy(); print("auto ratio", a.last_power_ratio())
f = G.Chain(taps, N, win, capi.CHAIN_FUSED_FD); gf = f.process_bulk(xd).cpu().numpy()
u = G.Chain(taps, N, win, capi.CHAIN_UNFUSED); gu = u.process_bulk(xd).cpu().numpy()
ea, ef, eu = err(ga), err(gf), err(gu)
same = np.all(ga.reshape(frames, N) == gf.reshape(frames, N), axis=1)
print("frames where AUTO == plain FD bit for bit:", int(same.sum()), "of", frames, "; first differing:", np.nonzero(~same)[0][:8])
for fr in (90, 92, 93, 100, 134, 200):
    k = int(np.argmax(np.abs(ga.reshape(frames, N)[fr] - truth[fr]) / np.maximum(truth[fr], rms[fr])))
    print(f"frame {fr}: err auto {ea[fr]:.3g} fd {ef[fr]:.3g} pair {eu[fr]:.3g}  same-as-fd {bool(same[fr])}  worst bin {k} truth {truth[fr][k]:.3g} rms {rms[fr][0]:.3g} auto {ga.reshape(frames,N)[fr][k]:.6g} pair {gu.reshape(frames,N)[fr][k]:.6g}")
