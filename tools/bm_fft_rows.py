#!/usr/bin/env python
"""developer tool: the reference's own FFT benchmark shapes (algorithm/benchmarks/bm_fft.cpp:44-58: N = 512, 1024, 8192, 65536 and the prime 1009, complex<float>, forward
transform of a whole batch) on this GPU: gr4hip_fft_spectrum (complex spectrum out: 16 B of HBM traffic per point) and gr4hip_fft_mag2 (12 B), 2^27 points per launch,
back-to-back launches between two events.  One row per N: Gsamples/s, fraction of 8 TB/s at the row's algorithmic bytes, transforms per second, N log N "ops" per second
(bm_fft's own scaling), and the oracle's error on one frame.  usage: bm_fft_rows.py [> profiles/r05_bm_fft.txt]"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, "tests")
import numpy as np, torch
import gnuradio4_amd as G
import oracle_lib as O

total = 1 << 27
print("# tools/bm_fft_rows.py: the shapes of algorithm/benchmarks/bm_fft.cpp:44-58 (complex<float>, forward), 2^27 points per launch, window None")
print(f"{'N':>6s} {'output':>9s} {'Gsamples/s':>11s} {'frac of 8 TB/s':>15s} {'transforms/s':>13s} {'N ln N /s':>10s} {'max rel err vs float64 oracle':>30s}")
for N in (512, 1024, 8192, 65536, 1009):
    frames = total // N
    x = G.synth_c32(frames * N, seed=5)
    F = G.FFT(N, "None")
    for kind, fn, out, bytes_pt in (("spectrum", F.spectrum, torch.empty((frames, N), dtype=torch.complex64, device="cuda"), 16), ("mag2", F.mag2, torch.empty((frames, N), dtype=torch.float32, device="cuda"), 12)):
        for _ in range(3):
            fn(x, out)
        torch.cuda.synchronize()
        ms = []
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10):
                fn(x, out)
            b.record(); b.synchronize()
            ms.append(a.elapsed_time(b) / 10)
        t = sorted(ms)[2] * 1e-3
        xf = x[3 * N:4 * N].cpu().numpy()
        truth = np.fft.fft(xf.astype(np.complex128))
        got = out[3].cpu().numpy()
        ref = truth if kind == "spectrum" else np.abs(truth) ** 2
        rms = np.sqrt(np.mean(np.abs(ref) ** 2))
        err = float(np.max(np.abs(got - ref) / np.maximum(np.abs(ref), rms)))
        rate = frames * N / t
        print(f"{N:6d} {kind:>9s} {rate / 1e9:11.1f} {rate * bytes_pt / 8e12:15.3f} {frames / t:13.3e} {frames * N * math.log(N) / t:10.3e} {err:30.2e}")
    del x, F
