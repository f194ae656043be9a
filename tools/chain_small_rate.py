import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import gnuradio4_amd as G
n = 1 << 27
xc = G.synth_c32(n)
k = np.arange(64); w = np.hamming(64); t = w * 0.2 * np.sinc(0.2 * (k - 31.5)); taps = (t / t.sum()).astype(np.float32)
m2 = torch.empty((n // 1024, 1024), dtype=torch.float32, device="cuda")
for algo in (3, 0, 3, 0):
    ch = G.Chain(taps, 1024, "Hann", algo)
    for _ in range(3): ch.process_bulk(xc, m2)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): ch.process_bulk(xc, m2)
    b.record(); b.synchronize()
    print("algo", algo, "%.1f Gs/s" % (n * 10 / a.elapsed_time(b) / 1e6))
ch = G.Chain(taps, 1024, "Hann", 0)
ch.process_bulk(xc, m2)
print("guard state after one call: ratio %.4f time_domain %s" % ch.last_power_ratio())
