"""developer tool: the f16 FIR kernels under a rejected tone 50 dB above the noise -- error against float64 per 2048-sample block, default / guard off / float32 kernels"""
import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import gnuradio4_amd as G, oracle_lib as O
from gnuradio4_amd import capi
n, ntaps = 200_000, int(sys.argv[1]) if len(sys.argv) > 1 else 200
cplx = len(sys.argv) < 3 or sys.argv[2] == "c"
fc = float(sys.argv[3]) if len(sys.argv) > 3 else 0.1
amp = float(sys.argv[4]) if len(sys.argv) > 4 else 316.0
ft = float(sys.argv[5]) if len(sys.argv) > 5 else 0.31
kk = np.arange(ntaps); bw = (np.hamming(ntaps) * 2 * fc * np.sinc(2 * fc * (kk - (ntaps - 1) / 2.0))).astype(np.float32)
if cplx:
    xi = (O.signal_c32(7, n, tone_amp=0.0) * 0.05).astype(np.complex64)
    xi += (amp * np.exp(2j * np.pi * ft * np.arange(n))).astype(np.complex64)
else:
    xi = (O.signal_f32(7, n, tone_amp=0.0) * 0.05 + amp * np.cos(2 * np.pi * ft * np.arange(n))).astype(np.float32)
ti, _ = O.fir(bw, xi)
rms = float(np.sqrt(np.mean(np.abs(ti) ** 2)))
dt = torch.complex64 if cplx else torch.float32
pad = 2 if cplx else 4
def dev(x):
    t = torch.empty(x.size + pad, dtype=dt, device="cuda")[pad:]
    t.copy_(torch.from_numpy(x)); return t
yr = O.fir(bw, xi, acc64=False)[0]
er = np.abs(yr - ti) / np.maximum(np.abs(ti), rms)
print(f"reference float32 sequential sum: max {er[ntaps:].max():.2e} rms {np.sqrt(np.mean(er[ntaps:]**2)):.2e}")
for name, algo, guard in (("default", capi.FIR_TIME_DOMAIN if cplx else capi.FIR_AUTO, None), ("guard off", capi.FIR_TIME_DOMAIN if cplx else capi.FIR_AUTO, capi.GUARD_OFF), ("f32 mfma", capi.FIR_TIME_DOMAIN_F32, None), ("exact f32", capi.FIR_EXACT_F32, None), ("bf16x3", capi.FIR_TIME_DOMAIN_BF16X3, None)):
    f = G.fir_filter(bw, dt); f.set_algo(algo)
    if guard is not None: f.set_guard_mode(guard)
    y = f.process_bulk(dev(xi)).cpu().numpy()
    e = np.abs(y - ti) / np.maximum(np.abs(ti), rms)
    blk = e[: (n // 2048) * 2048].reshape(-1, 2048).max(axis=1)
    print("   blocks above 1e-4:", np.flatnonzero(blk > 1e-4)[:40])
    print(f"{name:10s} max {e[ntaps:].max():.2e}  rms {np.sqrt(np.mean(e[ntaps:]**2)):.2e}  per-block max: first {blk[:4]}, median {np.median(blk):.2e}, worst block {int(blk.argmax())}")
