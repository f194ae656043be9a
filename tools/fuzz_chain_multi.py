#!/usr/bin/env python
"""developer tool: gr4hip_chain_process_multi (several channels in one launch, per-channel spectra or the fold in registers) under the wide chain fuzzer's streams:
2 .. 6 channels at 8192 points, shared or per-channel taps, an interferer / second tone / burst on a random subset of the channels, 1 .. 3 calls per stream,
against float64 numpy.  usage: fuzz_chain_multi.py [seconds = 120] [seed = 0]"""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
from scipy.signal import lfilter
import gnuradio4_amd as G
from gnuradio4_amd.blocks import chain_process_multi
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
N = 8192
def lowpass(nt, fc):
    k = np.arange(nt); t = np.hamming(nt) * 2 * fc * np.sinc(2 * fc * (k - (nt - 1) / 2)); return (t / t.sum()).astype(np.float32)
def stream(n, fc, nt):
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    if rng.random() < 0.3: x *= np.float32(10 ** float(rng.uniform(-3, 3)))
    if rng.random() < 0.6:
        start = int(rng.integers(0, n)); amp = 10 ** (float(rng.uniform(-5, 50)) / 20) * float(np.sqrt(np.mean(np.abs(x[:4096]) ** 2) / 2))
        f = float(rng.uniform(min(0.45, fc + 2.0 / nt + 0.01), 0.49)) * (1 if rng.random() < 0.5 else -1)
        stop = n if rng.random() < 0.7 else min(n, start + int(rng.integers(100, 4 * N)))
        x[start:stop] += (amp * np.exp(2j * np.pi * f * np.arange(stop - start))).astype(np.complex64)
    return x
t0 = time.time(); cases = fails = 0; worst = 0.0
while time.time() - t0 < secs:
    nch = int(rng.integers(2, 7)); frames = int(rng.integers(8, 120)); n = frames * N
    shared = rng.random() < 0.6; fold = shared and rng.random() < 0.5
    nt0 = int(rng.choice([2, 17, 64, 100, 256])); fc0 = float(rng.choice([0.005, 0.02, 0.05, 0.2]))
    taps = [lowpass(nt0, fc0)] * nch if shared else [lowpass(int(rng.choice([17, 64, 100, 256])), float(rng.choice([0.005, 0.02, 0.05, 0.2]))) for _ in range(nch)]
    xs = [stream(n, fc0, nt0) for _ in range(nch)]
    chains = [G.Chain(taps[0] if shared else taps[c], N, "None") for c in range(nch)]
    if shared:  # (the one launch with shared taps wants the SAME handle's taps: same array)
        pass
    cuts = sorted(set([0, frames] + [int(c) for c in rng.integers(0, frames, size=int(rng.integers(0, 3)))]))
    sums, outs = [], [[] for _ in range(nch)]
    for a, b in zip(cuts[:-1], cuts[1:]):
        o, s = chain_process_multi(chains, [torch.from_numpy(x[a * N:b * N]).cuda() for x in xs], want_outs=not fold, sum_out=torch.empty((b - a, N), dtype=torch.float32, device="cuda"))
        sums.append(s.cpu().numpy())
        if not fold:
            for c in range(nch): outs[c].append(o[c].cpu().numpy())
    truth = [np.abs(np.fft.fft(lfilter(taps[c].astype(np.float64), [1.0], xs[c].astype(np.complex128)).reshape(frames, N), axis=1)) ** 2 for c in range(nch)]
    def rel(g, t):
        rms = np.sqrt(np.mean(t ** 2, axis=1, keepdims=True)) + 1e-300
        return float(np.max(np.abs(g - t) / np.maximum(t, rms)))
    r = rel(np.concatenate(sums), sum(truth))
    if not fold: r = max(r, max(rel(np.concatenate(outs[c]), truth[c]) for c in range(nch)))
    cases += 1; worst = max(worst, r)
    if r > 1e-5:
        fails += 1
        if fails <= 8: print("FAIL", f"channels={nch} shared={shared} fold={fold} taps={[len(t) for t in taps]} frames={frames} cuts={cuts}", r, flush=True)
print(f"{cases} cases in {time.time() - t0:.0f} s, {fails} above 1e-5, worst relative error {worst:.3g}")
