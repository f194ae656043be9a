import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import numpy as np, torch
import gnuradio4_amd as G
from _timing import steady
from gnuradio4_amd import capi
def lowpass(ntaps, fc):
    k = np.arange(ntaps, dtype=np.float64)
    t = np.hamming(ntaps) * 2 * fc * np.sinc(2 * fc * (k - (ntaps - 1) / 2.0))
    return (t / t.sum()).astype(np.float32)
n = 1 << 27
x = G.synth_f32(n, seed=42)
yd = torch.empty(n // 8, dtype=torch.float32, device="cuda")
for rep in range(2):
    fir = G.fir_filter(lowpass(1024, 0.05), torch.float32, decimate=8)
    t_fd = steady(lambda: fir.process_bulk(x, yd))
    capi.developer_switch("GR4HIP_FIR_NO_DECIM_FD", 1)
    fir_p = G.fir_filter(lowpass(1024, 0.05), torch.float32, decimate=8)
    t_p = steady(lambda: fir_p.process_bulk(x, yd))
    capi.developer_switch("GR4HIP_FIR_NO_DECIM_FD", 0)
    print(f"decim-8 1024 taps: FD {n / t_fd / 1e9:.0f} G  polyphase {n / t_p / 1e9:.0f} G input samples/s")
