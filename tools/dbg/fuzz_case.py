"""one dumped case of tools/fuzz_chain.py through every way the library has of computing it"""
import sys
sys.path.insert(0, "tests")
import numpy as np, torch
from scipy.signal import lfilter
import oracle_lib as O
import gnuradio4_amd as G
from gnuradio4_amd import capi
d = np.load(sys.argv[1]); x, taps, N, win, cuts, frames = d["x"], d["taps"], int(d["N"]), str(d["win"]), list(d["cuts"]), int(d["frames"])
y = lfilter(taps.astype(np.float64), [1.0], x.astype(np.complex128)).reshape(frames, N)
w = 1.0
if win != "None":
    w32 = np.empty(N, np.float32); capi.check(capi.lib().gr4hip_window_create({"Hann": 3, "Hamming": 2, "BlackmanHarris": 7}[win], w32.ctypes.data, N, 1.6), "window"); w = w32.astype(np.float64)
truth = np.abs(np.fft.fft(y * w, axis=1)) ** 2
rms = np.sqrt(np.mean(truth ** 2, axis=1, keepdims=True)) + 1e-300
pin = np.mean(np.abs(x.reshape(frames, N)) ** 2, axis=1)
print(f"N={N} taps={len(taps)} win={win} frames={frames} cuts={cuts}  level rms {np.sqrt(pin.mean()):.3g}")
def run(algo, guard=None, whole=False):
    ch = G.Chain(taps, N, win, algo)
    if guard is not None: ch.set_guard_mode(guard)
    cc = [0, frames] if whole else cuts
    got = np.concatenate([ch.process_bulk(torch.from_numpy(x[a * N:b * N]).cuda()).cpu().numpy() for a, b in zip(cc[:-1], cc[1:]) if b > a]).reshape(frames, N)
    e = np.max(np.abs(got - truth) / np.maximum(truth, rms), axis=1)
    return e, ch
for name, algo, guard, whole in (("AUTO strict, the fuzzer's cuts", capi.CHAIN_AUTO, None, False), ("AUTO strict, one call", capi.CHAIN_AUTO, None, True), ("AUTO guard off", capi.CHAIN_AUTO, capi.GUARD_OFF, True),
                                 ("FUSED_FD (no guard)", capi.CHAIN_FUSED_FD, None, True), ("UNFUSED (kernel pair)", capi.CHAIN_UNFUSED, None, True), ("TIME_DOMAIN (f32 products)", capi.CHAIN_TIME_DOMAIN, None, True)):
    try:
        e, ch = run(algo, guard, whole)
        fr = int(np.argmax(e)); T = truth[fr]
        r4 = (np.mean(w * w) if win != "None" else 1.0) * N * pin[fr] / np.sqrt(np.mean(T ** 2))
        print(f"{name:34s}: worst {e.max():.3g} at frame {fr} (power ratio {T.sum() / (N * N * pin[fr] * (np.mean(w*w) if win != 'None' else 1.0)):.3g}, R4 {r4:.3g}); frames above 1e-5: {np.nonzero(e > 1e-5)[0][:10].tolist()}  algo {ch.algo}")
    except Exception as ex:
        print(name, "->", str(ex)[:100])
