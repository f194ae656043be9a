"""developer tool: the chain's kernel pair with MORE than 256 taps (complex filter on the bf16 / f32 kernels + fir_judge_kernel's verdicts): rate on white noise and per-frame error
against float64 with and without a tone 22 dB above the noise far outside the pass band.  usage: pair_long_taps.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, "tests")
import numpy as np, torch
from scipy.signal import lfilter
import gnuradio4_amd as G
from gnuradio4_amd import capi
from _timing import steady
def lowpass(nt, fc):
    k = np.arange(nt); t = np.hamming(nt) * 2 * fc * np.sinc(2 * fc * (k - (nt - 1) / 2)); return (t / t.sum()).astype(np.float32)
N = 8192
rng = np.random.default_rng(5)
xs = G.synth_c32(1 << 26, seed=3); m = torch.empty(((1 << 26) // N, N), dtype=torch.float32, device="cuda")
for nt, fc in ((512, 0.004), (512, 0.02), (1024, 0.002)):
    b = lowpass(nt, fc)
    ch = G.Chain(b, N, "None")
    for _ in range(2): ch.process_bulk(xs, m)
    line = f"{nt} taps fc {fc}: algo {ch.algo}  {xs.numel() / steady(lambda: ch.process_bulk(xs, m)) / 1e9:.0f} Gsamples/s"
    frames = 24
    noise = (rng.standard_normal(frames * N) + 1j * rng.standard_normal(frames * N)).astype(np.complex64)
    for db in (None, 18.0, 22.0, 30.0):
        x = noise.copy()
        if db is not None: x[6 * N:] += (10 ** (db / 20) * np.exp(2j * np.pi * 0.41 * np.arange((frames - 6) * N))).astype(np.complex64)
        truth = np.abs(np.fft.fft(lfilter(b.astype(np.float64), [1.0], x.astype(np.complex128)).reshape(frames, N), axis=1)) ** 2
        rms = np.sqrt(np.mean(truth ** 2, axis=1, keepdims=True))
        got = G.Chain(b, N, "None").process_bulk(torch.from_numpy(x).cuda()).cpu().numpy().reshape(frames, N)
        line += f"  tone {db}: {float(np.max(np.abs(got - truth) / np.maximum(truth, rms))):.2e}"
    print(line, flush=True)
