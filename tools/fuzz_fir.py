#!/usr/bin/env python
"""developer tool: random fir_filter<float> / batched FIR / FFT<double> configurations against float64 numpy (spans of ragged length, several calls per stream).
usage: fuzz_fir.py [seconds = 120] [seed = 0]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from scipy.signal import lfilter
import gnuradio4_amd as G

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t0 = time.time(); cases = 0; worst = 0.0
def rel(a, b):
    rms = np.sqrt(np.mean(np.abs(b) ** 2)) + 1e-30
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), rms)))
while time.time() - t0 < secs:
    kind = rng.integers(0, 3)
    if kind == 0:  # single stream, several calls of ragged length
        nt = int(rng.choice([33, 64, 100, 129, 200, 209, 210, 224, 241, 242, 255, 256, 384, 500, 512, 777, 1024]))
        taps = (rng.standard_normal(nt) * np.hamming(nt)).astype(np.float32)
        n = int(rng.integers(1 << 17, 1 << 21)) + int(rng.integers(0, 5000))
        x = rng.standard_normal(n).astype(np.float32)
        f = G.fir_filter(taps, torch.float32)
        cuts = sorted(set([0, n] + [int(c) // 4 * 4 for c in rng.integers(0, n, size=int(rng.integers(0, 3)))]))
        y = np.concatenate([f.process_bulk(torch.from_numpy(x[a:b]).cuda()).cpu().numpy() for a, b in zip(cuts[:-1], cuts[1:]) if b > a])
        r = rel(y, lfilter(taps.astype(np.float64), [1.0], x.astype(np.float64)))
        tag = f"fir taps={nt} n={n} cuts={cuts}"
    elif kind == 1:  # batched
        nch = int(rng.integers(1, 9)); nt = int(rng.choice([40, 64, 200, 230, 256]))
        n = int(rng.integers(1 << 15, 1 << 18)) // 4 * 4
        taps = (rng.standard_normal((nch, nt)) * np.hamming(nt)).astype(np.float32)
        x = rng.standard_normal((nch, n)).astype(np.float32)
        fb = G.FirBatched(taps)
        h = n // 2 // 4 * 4
        y = np.concatenate([fb.process_bulk(torch.from_numpy(np.ascontiguousarray(x[:, :h])).cuda()).cpu().numpy(), fb.process_bulk(torch.from_numpy(np.ascontiguousarray(x[:, h:])).cuda()).cpu().numpy()], axis=1)
        truth = np.stack([lfilter(taps[c].astype(np.float64), [1.0], x[c].astype(np.float64)) for c in range(nch)])
        r = rel(y, truth); tag = f"batched nch={nch} taps={nt} n={n}"
    else:  # FFT<double>
        N = int(2 ** rng.integers(1, 14)); frames = int(rng.integers(1, 2000 if N <= 256 else 9))
        x = rng.standard_normal(frames * N)
        out = G.FFT(N, "None", dtype=torch.float64).process_bulk(torch.from_numpy(x).cuda())
        X = np.fft.fft(x.reshape(frames, N), axis=1)
        r = max(rel(out["magnitude"].cpu().numpy(), np.abs(X[:, :N // 2]) * 2 / N) * 1e7, rel(out["re"].cpu().numpy(), X[:, N // 2:].real) * 1e7)  # (scaled: 1e-12 -> 1e-5)
        tag = f"fft64 N={N} frames={frames}"
    cases += 1; worst = max(worst, r)
    if r > 1e-5: print("FAIL", tag, r, flush=True)
print(f"{cases} cases in {time.time() - t0:.0f} s, worst relative error {worst:.3g} (bar 1e-5; FFT<double> errors scaled by 1e7)")
