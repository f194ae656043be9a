"""developer tool: the one case of tools/fuzz_chain.py 120 41 above the bar (case 828: N = 8192, 65 taps fc 0.02, BlackmanHarris, 319 frames, a +20 dB tone from frame 286.78 on,
calls cut at frames 66 / 209): per-frame error of the third call under AUTO (strict), forced fused FD, the kernel pair and the direct form"""
import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from scipy.signal import lfilter
import gnuradio4_amd as G
from gnuradio4_amd import capi
rng = np.random.default_rng(41)
def lowpass(nt, fc):
    k = np.arange(nt); t = np.hamming(nt) * 2 * fc * np.sinc(2 * fc * (k - (nt - 1) / 2)); return (t / t.sum()).astype(np.float32)
while True:  # replay the fuzzer's draws
    N = int(2 ** rng.integers(8, 14)); nt = int(rng.choice([2, 17, 64, 65, 100, 200, 256]))
    win = str(rng.choice(["None", "Hann", "Hamming", "BlackmanHarris"]))
    frames = int(rng.integers(20, 400)) if N >= 2048 else int(rng.integers(100, 3000))
    n = frames * N
    taps = lowpass(nt, float(rng.choice([0.02, 0.05, 0.2])))
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    if rng.random() < 0.6:
        start = int(rng.integers(0, n)); amp = 10 ** (float(rng.uniform(20, 50)) / 20)
        x[start:] += (amp * np.exp(2j * np.pi * 0.41 * np.arange(n - start))).astype(np.complex64)
    cuts = sorted(set([0, frames] + [int(c) for c in rng.integers(0, frames, size=int(rng.integers(0, 4)))]))
    if N == 8192 and nt == 65 and win == "BlackmanHarris" and frames == 319: break
w32 = np.empty(N, np.float32)
capi.check(capi.lib().gr4hip_window_create(7, w32.ctypes.data, N, 1.6), "window")
y = lfilter(taps.astype(np.float64), [1.0], x.astype(np.complex128)).reshape(frames, N) * w32.astype(np.float64)
truth = np.abs(np.fft.fft(y, axis=1)) ** 2
rms = np.sqrt(np.mean(truth ** 2, axis=1, keepdims=True))
def run(algo, cuts, guard=None):
    ch = G.Chain(taps, N, win, algo)
    if guard is not None: ch.set_guard_mode(guard)
    parts = [ch.process_bulk(torch.from_numpy(x[a * N:b * N]).cuda()).cpu().numpy() for a, b in zip(cuts[:-1], cuts[1:]) if b > a]
    e = np.abs(np.concatenate(parts).reshape(frames, N) - truth) / np.maximum(truth, rms)
    return e.max(axis=1), ch.last_power_ratio()
for name, algo, cc, gd in (("auto strict, fuzz cuts", capi.CHAIN_AUTO, cuts, None), ("auto strict, one call", capi.CHAIN_AUTO, [0, frames], None), ("auto guard off", capi.CHAIN_AUTO, cuts, capi.GUARD_OFF),
                           ("fused fd", capi.CHAIN_FUSED_FD, cuts, None), ("unfused pair", capi.CHAIN_UNFUSED, cuts, None), ("time domain", capi.CHAIN_TIME_DOMAIN, cuts, None)):
    e, pr = run(algo, cc, gd)
    print(f"{name:24s}: worst {e.max():.3e} in frame {e.argmax()}; frames 284..292 " + " ".join(f"{v:.1e}" for v in e[284:293]) + "; 300..306 " + " ".join(f"{v:.1e}" for v in e[300:307]) + f"  ratio/td {pr}", flush=True)
