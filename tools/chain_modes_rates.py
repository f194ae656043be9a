"""developer tool: rates of the fused frame-pipeline kernel's other outputs (Hann-windowed |X|^2 at 8192, complex fast-convolution FIR) on 2^27 samples"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import gnuradio4_amd as G
n = 1 << 27
x = G.synth_c32(n, seed=5)
k = np.arange(256); taps = np.hamming(256) * 0.2 * np.sinc(0.2 * (k - 127.5)); taps = (taps / taps.sum()).astype(np.float32)  # bench.py's taps (a bare Hamming window as taps
# removes > 14 dB of a white input: the dynamic-range guard would move the chain to the time-domain kernels)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _timing import steady
def rate(fn):
    return n / steady(fn) / 1e6  # back to back at settled clocks (tools/_timing.py)
ch = G.Chain(taps, 8192, "Hann"); m2 = torch.empty((n // 8192, 8192), dtype=torch.float32, device="cuda")
print("chain 256 taps -> 8192 Hann -> mag2: %.1f Msamples/s" % rate(lambda: ch.process_bulk(x, m2)))
ch0 = G.Chain(taps, 8192, "None")
print("chain 256 taps -> 8192 None -> mag2: %.1f Msamples/s" % rate(lambda: ch0.process_bulk(x, m2)))
f = G.fir_filter(taps, dtype=torch.complex64)
y = torch.empty_like(x)
print("fir_filter<complex<float>> 256 taps: %.1f Msamples/s" % rate(lambda: f.process_bulk(x, y)))
F = G.FFT(8192, "Hann"); sp = torch.empty((n // 8192, 8192), dtype=torch.complex64, device="cuda")
print("FFT block 8192 Hann -> complex spectrum: %.1f Msamples/s" % rate(lambda: F.spectrum(x, sp)))
os.environ["GR4HIP_FFT_NO_PIPELINE"] = "1"
print("   block kernel (GR4HIP_FFT_NO_PIPELINE): %.1f Msamples/s" % rate(lambda: F.spectrum(x, sp)))
