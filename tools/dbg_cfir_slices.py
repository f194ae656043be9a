import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import gnuradio4_amd as G
import oracle_lib as O
def dev16c(x):
    t = torch.empty(x.size + 2, dtype=torch.complex64, device="cuda")[2:]; t.copy_(torch.from_numpy(x)); return t
def rel(got, truth):
    rms = float(np.sqrt(np.mean(np.abs(truth) ** 2))); e = np.abs(got - truth) / np.maximum(np.abs(truth), rms); return float(e.max()), int(e.argmax())
n = 400_000
x = O.signal_c32(91, n, tone_frel=0.31, tone_amp=30.0)
for ntaps in (480, 496, 512, 300):
    b = O.design_taps_hamming_lowpass(ntaps, 0.02)
    for scale in (1.0, 1e-30, 1e30):
        xs_ = (x.astype(np.complex128) * scale).astype(np.complex64)
        ts, _ = O.fir(b, xs_)
        y1 = G.fir_filter(b, torch.complex64).process_bulk(dev16c(xs_)).cpu().numpy()
        f = G.fir_filter(b, torch.complex64); f.set_guard_mode(G.capi.GUARD_OFF)
        y2 = f.process_bulk(dev16c(xs_)).cpu().numpy()
        t32 = O.fir(b, xs_, acc64=False)[0]
        print(ntaps, scale, "one call", rel(y1, ts), "guard off", rel(y2, ts), "reference float32", rel(t32, ts), flush=True)
