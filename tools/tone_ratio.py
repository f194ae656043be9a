#!/usr/bin/env python
"""developer tool: the rejected-signal regime of every matrix-pipe FIR / decimator family against the REFERENCE's float32 arithmetic -- the oracle's sequential
float32 sum (oracle/gr4_oracle.c gr4o_fir_f32 / gr4o_fir_c32: time_domain_filter.hpp:44-47 evaluated in order), not another device kernel.
Per case: e = error of the device against float64, e32 = error of the reference float32 sum against float64 (both with the parity contract's formula), ratio e / e32.
Prints the distribution of the ratio per family over the cases where e > 1e-5 (the cases the contract's second clause is about).
usage: tone_ratio.py [seconds = 120] [seed = 0]"""
import sys, time
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import numpy as np, torch
import gnuradio4_amd as G
import oracle_lib as O
from gnuradio4_amd import capi

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t0 = time.time()
fam = {}


def run(f, x, cplx):
    pad = 2 if cplx else 4
    t = torch.empty(len(x) + pad, dtype=torch.complex64 if cplx else torch.float32, device="cuda")[pad:]
    t.copy_(torch.from_numpy(x))
    return f.process_bulk(t).cpu().numpy()


def rel(y, t, sl):
    y, t = y[sl], t[sl]
    rms = float(np.sqrt(np.mean(np.abs(t) ** 2)))
    return float(np.max(np.abs(y - t) / np.maximum(np.abs(t), rms)))


while time.time() - t0 < secs:
    cplx = bool(rng.integers(0, 2))
    D = int(rng.choice([1, 1, 8, 16, 32, 2, 4, 5, 10]))
    if D == 1:
        nt = int(rng.choice([33, 64, 100, 129, 200, 256] + ([] if cplx else [384, 512, 777, 1024])))
    elif D in (8, 16, 32):
        nt = int(rng.choice({8: [97, 200, 257, 513], 16: [33, 64, 130, 300, 449], 32: [33, 64, 130, 321]}[D])) if cplx else int(rng.choice({8: [97, 200, 400, 513, 1024], 16: [33, 129, 300, 641, 897], 32: [33, 129, 385, 641]}[D]))
    else:
        nt = int(rng.choice([64, 100, 256, 300]))
    n = (int(rng.integers(1 << 18, 1 << 19)) // (4 * D)) * 4 * D
    k = np.arange(nt); fc = 0.4 / D if D > 1 else 0.05
    taps = (np.hamming(nt) * 2 * fc * np.sinc(2 * fc * (k - (nt - 1) / 2.0))).astype(np.float32)
    amp = 10.0 ** rng.uniform(1, 3)
    ph = 2 * np.pi * rng.uniform(0.2, 0.45) * np.arange(n)
    x = rng.standard_normal(n) + (1j * rng.standard_normal(n) if cplx else 0)
    x = (x * 0.05 + amp * (np.exp(1j * ph) if cplx else np.cos(ph))).astype(np.complex64 if cplx else np.float32)
    truth = O.fir(taps, x, acc64=True)[0][::D]
    ref32 = O.fir(taps, x, acc64=False)[0][::D]
    f = G.fir_filter(taps, torch.complex64 if cplx else torch.float32, decimate=D)
    if cplx and D == 1:
        f.set_algo(capi.FIR_TIME_DOMAIN)
    y = run(f, x, cplx)
    sl = slice(nt // D + 1, None)
    e, e32 = rel(y, truth, sl), rel(ref32, truth, sl)
    key = f"{'c' if cplx else 'f'} D={D}"
    fam.setdefault(key, []).append((e, e32, nt, amp))

print(f"# tools/tone_ratio.py {secs:.0f} s: rejected tone 26 .. 66 dB above the noise that passes; e = device vs float64, e32 = reference sequential float32 sum vs float64")
print(f"{'family':10s} {'cases':>5s} {'e>1e-5':>6s} {'max e':>9s} {'ratio e/e32 over e>1e-5: median':>32s} {'p90':>6s} {'max':>6s}   worst (taps, amp, e, e32)")
for key in sorted(fam):
    v = fam[key]
    hot = [(e / max(e32, 1e-30), nt, amp, e, e32) for e, e32, nt, amp in v if e > 1e-5]
    if hot:
        r = np.array([h[0] for h in hot]); w = max(hot)
        print(f"{key:10s} {len(v):5d} {len(hot):6d} {max(e for e, *_ in v):9.2e} {np.median(r):32.2f} {np.quantile(r, 0.9):6.2f} {r.max():6.2f}   ({w[1]}, {w[2]:.0f}, {w[3]:.2e}, {w[4]:.2e})")
    else:
        print(f"{key:10s} {len(v):5d} {0:6d} {max(e for e, *_ in v):9.2e}")
