"""20 calls of the merged (mult -> div -> add)^10 program and 20 of the channeliser front end: the kernel trace must show ONE dispatch per call for the former and
two (filter with its load hook, transform) for the latter (rocprofv3 --kernel-trace --stats)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import gnuradio4_amd as G

n = 1 << 26
x = G.synth_f32(n, seed=1)
y = torch.empty_like(x)
m = G.Merged(torch.float32, [("Multiply", 2.0), ("Divide", 2.0), ("Add", -1.0)] * 10)
for _ in range(20):
    m.process_bulk(x, y)
xc = G.synth_c32(n // 2, seed=2)
k = np.arange(64, dtype=np.float64)
b = np.hamming(64) * 0.1 * np.sinc(0.1 * (k - 31.5))
fir = G.fir_filter((b / b.sum()).astype(np.float32), torch.complex64, decimate=8)
fir.set_prologue(G.Merged(torch.complex64, [("Rotator", 0.3, 0.25)]))
fft = G.FFT(1024, "Hann")
yd = torch.empty(n // 16, dtype=torch.complex64, device="cuda")
sp = torch.empty((n // 16 // 1024, 1024), dtype=torch.float32, device="cuda")
for _ in range(20):
    fir.process_bulk(xc, yd)
    fft.mag2(yd, sp)
torch.cuda.synchronize()
