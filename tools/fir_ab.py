import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import numpy as np, torch
import gnuradio4_amd as G
from _timing import steady
from gnuradio4_amd import capi
def lowpass(ntaps, fc):
    k = np.arange(ntaps, dtype=np.float64)
    t = np.hamming(ntaps) * 2 * fc * np.sinc(2 * fc * (k - (ntaps - 1) / 2.0))
    return (t / t.sum()).astype(np.float32)
n = 1 << 28
x = G.synth_f32(n, seed=42)
y = torch.empty_like(x)
out = []
for nt in (48, 64, 100, 128, 200, 256, 512, 1024):
    f = G.fir_filter(lowpass(nt, 0.05), torch.float32)
    t = steady(lambda: f.process_bulk(x, y))
    out.append(f"{nt}: {n / t / 1e9:.0f}")
print("float FIR Gsamples/s ", "  ".join(out))
nch, n2 = 64, 1 << 22
xb = x[: nch * n2].view(nch, n2)
fb = G.FirBatched(np.stack([lowpass(256, 0.05 + 0.005 * c) for c in range(nch)]))
yb = torch.empty_like(xb)
t = steady(lambda: fb.process_bulk(xb, yb))
print(f"configs[3] {nch * n2 / t / 1e9:.0f} G")
nc = 1 << 27
xc = G.synth_c32(nc)
yc = torch.empty(nc, dtype=torch.complex64, device="cuda")
out = []
for nt in (128, 200, 224, 256):
    f = G.fir_filter(lowpass(nt, 0.05), torch.complex64)
    f.set_algo(capi.FIR_TIME_DOMAIN)
    t = steady(lambda: f.process_bulk(xc, yc))
    out.append(f"{nt}: {nc / t / 1e9:.0f}")
print("complex direct-form FIR Gsamples/s ", "  ".join(out))
ch = G.Chain(lowpass(256, 0.05), 8192, "None", capi.CHAIN_TIME_DOMAIN)
m2 = torch.empty((nc // 8192, 8192), dtype=torch.float32, device="cuda")
t = steady(lambda: ch.process_bulk(xc, m2))
print(f"chain 256 taps -> 8192 -> mag2, time-domain pair (where the guard sends the headline stream): {nc / t / 1e9:.0f} G")
