"""developer tool: fir_filter's f16 kernels (22-bit products) under a tone the filter removes, at levels just BELOW the guard's marking threshold (the segment's output power
1/128 .. 1/8 of what white noise of the input's power would pass): the products' error is coherent on the tone -- how large is it against the filtered samples' rms?
float / complex, 65 / 256 taps, decimate by 1 / 8 (1024 taps) / 16"""
import sys
sys.path.insert(0, "tests")
import numpy as np, torch
from scipy.signal import lfilter
import gnuradio4_amd as G
import oracle_lib as O
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 60
def lowpass(nt, fc):
    k = np.arange(nt); t = np.hamming(nt) * 2 * fc * np.sinc(2 * fc * (k - (nt - 1) / 2)); return (t / t.sum()).astype(np.float32)
n = 1 << 18
CASES = ((True, 65, 0.05, 1), (True, 256, 0.02, 1), (False, 256, 0.02, 1), (False, 65, 0.05, 1), (False, 1024, 0.02, 8), (False, 512, 0.01, 16))
for cplx, nt, fc, D in (CASES[int(sys.argv[3]):int(sys.argv[3]) + 1] if len(sys.argv) > 3 else CASES):
    taps = lowpass(nt, fc); tp = float(np.sum(taps.astype(np.float64) ** 2))
    worst = 0.0; w32 = 0.0; bad = 0; top = []
    for trial in range(trials):
        f0 = float(rng.uniform(fc + 2.0 / nt, 0.49))
        sig2 = 2.0 if cplx else 1.0
        thr = float(10 ** rng.uniform(np.log10(1 / 120.0), np.log10(1 / 8.0)))   # P_y / (tap power x P_x) aimed at: just above the guard's 1 / 128
        a2 = sig2 * (1.0 / thr - 1.0); amp = np.sqrt(a2 if cplx else 2 * a2)
        ph = rng.random()
        if cplx: x = (rng.standard_normal(n) + 1j * rng.standard_normal(n) + amp * np.exp(2j * np.pi * (f0 * np.arange(n) + ph))).astype(np.complex64)
        else: x = (rng.standard_normal(n) + amp * np.cos(2 * np.pi * (f0 * np.arange(n) + ph))).astype(np.float32)
        t = lfilter(taps.astype(np.float64), [1.0], x.astype(np.complex128 if cplx else np.float64))[::D] if D > 1 else lfilter(taps.astype(np.float64), [1.0], x.astype(np.complex128 if cplx else np.float64))
        if D > 1: t = lfilter(taps.astype(np.float64), [1.0], x.astype(np.float64))[::D]
        f = G.fir_filter(taps, torch.complex64 if cplx else torch.float32, decimate=D)
        got = f.process_bulk(torch.from_numpy(x).cuda()).cpu().numpy()
        m = min(len(got), len(t)); got, t = got[:m], t[:m]
        rms = np.sqrt(np.mean(np.abs(t) ** 2))
        e = float(np.max(np.abs(got - t) / np.maximum(np.abs(t), rms)))
        worst = max(worst, e); top.append((e, thr, amp, f0))
        if e > 1e-5:
            y32 = O.fir(taps, x, acc64=False)[0][::D][:m] if D == 1 or True else None
            e32 = float(np.max(np.abs(y32 - t) / np.maximum(np.abs(t), rms)))
            if e > max(1e-5, e32):
                bad += 1
                if bad <= 4: print(f"FAIL {'complex' if cplx else 'float'} {nt} taps D={D}: tone {amp:.3g} at {f0:.4f}, aimed ratio {thr:.4f}: err {e:.3g}, reference float32 {e32:.3g}", flush=True)
    print(f"{'complex' if cplx else 'float'} {nt} taps cut-off {fc} decimate {D}: {trials} streams, {bad} above the bar, worst {worst:.3g}; the five worst (err, aimed ratio, tone, f): " + " ".join(f"({a:.2g}, 1/{1 / b:.0f}, {c:.3g}, {d:.3f})" for a, b, c, d in sorted(top, reverse=True)[:5]), flush=True)
