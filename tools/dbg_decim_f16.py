"""developer tool: the f16 decimate-by-8 kernel under a rejected tone 50 dB above the noise -- error against float64 per 1024-output block"""
import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import gnuradio4_amd as G, oracle_lib as O
from gnuradio4_amd import capi
nt = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n = 8 * 40_000
b = O.design_taps_hamming_lowpass(nt, 0.05)
x = (O.signal_f32(7, n) * 0.05 + 316.0 * np.cos(2 * np.pi * 0.31 * np.arange(n))).astype(np.float32)
truth, _ = O.fir_decim(b, x, 8)
rms = float(np.sqrt(np.mean(truth[nt:] ** 2)))
t = torch.empty(n + 4, dtype=torch.float32, device="cuda")[4:]; t.copy_(torch.from_numpy(x))
for name, guard, sw in (("default", None, 0), ("guard off", capi.GUARD_OFF, 0), ("fd/polyphase", None, 1)):
    capi.developer_switch("GR4HIP_FIR_NO_DECIM_F16", sw)
    f = G.fir_filter(b, torch.float32, decimate=8)
    if guard is not None: f.set_guard_mode(guard)
    y = f.process_bulk(t).cpu().numpy()
    e = np.abs(y - truth) / np.maximum(np.abs(truth), rms)
    blk = e[: (len(e) // 1024) * 1024].reshape(-1, 1024).max(axis=1)
    print(f"{name:12s} max {e.max():.2e} first blocks {blk[:6]} median {np.median(blk):.2e} tail {e[(len(e)//1024)*1024:].max() if len(e) % 1024 else 0:.2e} bad blocks {np.flatnonzero(blk > 6e-5)[:20]}")
