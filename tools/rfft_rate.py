import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import numpy as np, torch
import gnuradio4_amd as G
from _timing import steady
n = 1 << 28
x = G.synth_f32(n, seed=1)
for N in (256, 1024, 4096, 8192):
    f = G.FFT(N, "Hann", dtype=torch.float32)
    out = f.process_bulk(x)  # warm + allocate
    t = steady(lambda: f.process_bulk(x))
    tot = sum(v.numel() * v.element_size() for v in out.values() if torch.is_tensor(v))
    print(f"FFT<float> real input N={N}: {n / t / 1e9:.0f} G real samples/s, {(n * 4 + tot) / t / 1e12:.2f} TB/s moved (all DataSet signals)")
