"""developer tool: where the guard sends a stream whose filter passes little of it -- the kernel pair (direct-form complex FIR -> y in HBM -> FFT kernel) -- and its two kernels alone,
256 taps / 8192-point frames, 2^27 samples: Gsamples/s (steady state)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
import gnuradio4_amd as G
from gnuradio4_amd import capi
from _timing import steady

def lowpass(nt, fc):
    k = np.arange(nt); t = np.hamming(nt) * 2 * fc * np.sinc(2 * fc * (k - (nt - 1) / 2)); return (t / t.sum()).astype(np.float32)
n, N = 1 << 27, 8192
x = G.synth_c32(n, seed=3)
y = torch.empty_like(x)
m = torch.empty((n // N, N), dtype=torch.float32, device="cuda")
for fc, what in ((0.05, "passes 10 % of the power"), (0.005, "passes 1 %")):
    b = lowpass(256, fc)
    row = [f"cut-off {fc} fs ({what}):"]
    for name, algo in (("AUTO (settled)", capi.CHAIN_AUTO), ("UNFUSED pair", capi.CHAIN_UNFUSED), ("TIME_DOMAIN pair (f32 products)", capi.CHAIN_TIME_DOMAIN)):
        ch = G.Chain(b, N, "None", algo)
        for _ in range(3): ch.process_bulk(x, m)   # (AUTO: the guard has moved the stream by now, if it is going to)
        row.append(f"{name} {n / steady(lambda: ch.process_bulk(x, m)) / 1e9:.0f} (ratio, td = {ch.last_power_ratio() if algo == capi.CHAIN_AUTO else '-'})")
    for name, algo in (("fir AUTO", None), ("fir TIME_DOMAIN", capi.FIR_TIME_DOMAIN), ("fir TIME_DOMAIN_F32", capi.FIR_TIME_DOMAIN_F32)):
        f = G.fir_filter(b, torch.complex64)
        if algo is not None: f.set_algo(algo)
        row.append(f"{name} {n / steady(lambda: f.process_bulk(x, y)) / 1e9:.0f}")
    ff = G.FFT(N, "None")
    row.append(f"fft mag2 {n / steady(lambda: ff.mag2(y, m)) / 1e9:.0f}")
    print("  ".join(row), flush=True)
