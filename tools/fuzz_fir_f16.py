#!/usr/bin/env python
"""developer tool: the f16 FIR kernels (csrc/fir_f16.hip, and for a third of the cases the decimate-by-8 / 16 / 32 kernel of fir_decim_f16.hip, float and complex: block exponent per segment,
per-segment guard, float32 paths) against float64 -- random tap counts, float / complex,
ragged and unaligned calls, stream levels 1e-30 .. 1e30, levels that jump by up to 1e12 from stretch to stretch, zero and denormal stretches, sparse outliers up to 1e30,
Inf / NaN samples, rejected tones up to 60 dB above the noise.  The error is judged block by block (4096 outputs) against the level the LOCAL input gives the products:
|y - truth| <= 1e-5 max(|truth|, sqrt(sum b^2) rms(x over the block, the taps in front of it and a segment either side)); under a rejected tone against the error of
the reference's own float32 arithmetic (the oracle's sequential float32 sum), factor ONE at every shape -- the parity contract of include/gr4hip.h.
usage: fuzz_fir_f16.py [seconds = 120] [seed = 0]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from scipy.signal import lfilter
sys.path.insert(0, "tests")
import gnuradio4_amd as G
import oracle_lib as O
from gnuradio4_amd import capi

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t0 = time.time(); cases = 0; worst = 0.0; fails = 0; kinds = {}; kfail = {}


def run(f, x, cuts, cplx, misalign):
    parts = []
    pad = 2 if cplx else 4
    for a, b in zip(cuts[:-1], cuts[1:]):
        if b <= a:
            continue
        t = torch.empty(b - a + pad + 1, dtype=torch.complex64 if cplx else torch.float32, device="cuda")[pad + (1 if misalign else 0):][: b - a]
        t.copy_(torch.from_numpy(x[a:b]))
        parts.append(f.process_bulk(t).cpu().numpy())
    return np.concatenate(parts)


def local_err(y, truth, x, taps, D=1):
    """max over blocks of 4096 outputs of |y - truth| / max(|truth|, local product level): the level of sqrt(sum b^2) x over the block, the taps in front of it and one
    kernel segment either side (the kernel's segments start at the CALL's first sample, not on the block grid); float32's own resolution (denormals) as a floor"""
    nt = len(taps); g = float(np.sqrt(np.sum(taps.astype(np.float64) ** 2)))
    x2 = np.abs(x.astype(np.complex128 if np.iscomplexobj(x) else np.float64)) ** 2
    worst, where = 0.0, -1
    B = 4096 // D if D == 1 else 1024  # (the decimator's segments are 1024 outputs)
    for s in range(0, len(y), B):
        e = min(s + B, len(y)); lo = max(0, D * (s - B) - nt - 32); hi = min(len(x), D * (e + B))
        lvl = g * np.sqrt(float(np.sum(x2[lo:hi])) / max(hi - lo, 1))  # (summed per window: a running sum loses the quiet stretches behind a loud one)
        den = np.maximum(np.maximum(np.abs(truth[s:e]), lvl), 1e-38)
        q = np.abs(y[s:e] - truth[s:e]) / den
        if float(q.max()) > worst:
            worst, where = float(q.max()), s + int(q.argmax())
    return worst, where


def fir64(taps, x):
    """the reference's arithmetic in float64: real taps on each component (a complex product would mix an Inf of one component into the other as NaN)"""
    with np.errstate(all="ignore"):
        b = taps.astype(np.float64)
        if np.iscomplexobj(x):
            out = np.empty(len(x), np.complex128)  # (not re + 1j * im: 0 x Inf)
            out.real, out.imag = lfilter(b, [1.0], x.real.astype(np.float64)), lfilter(b, [1.0], x.imag.astype(np.float64))
            return out
        return lfilter(b, [1.0], x.astype(np.float64))


while time.time() - t0 < secs:
    cplx = bool(rng.integers(0, 2))
    dec8 = rng.integers(0, 3) == 0  # BasicDecimatingFilter<float> / <complex<float>>, decimate by 8 / 16 / 32: fir_decim_f16.hip
    DD = int(rng.choice([8, 16, 32])) if dec8 else 1  # the decimation of the dec8 cases: 8, 16 or 32
    nt = int(rng.choice({8: [97, 100, 129, 130, 200, 257, 258, 385, 386, 513], 16: [33, 64, 65, 66, 129, 130, 193, 194, 300, 321, 322, 449], 32: [33, 64, 65, 129, 130, 193, 194, 257, 258, 321]}[DD])) if dec8 and cplx else int(rng.choice({8: [97, 100, 168, 200, 257, 258, 400, 513, 514, 700, 769, 770, 1000, 1024, 1025], 16: [33, 64, 129, 130, 300, 385, 386, 641, 642, 897], 32: [33, 64, 129, 130, 385, 386, 641]}[DD])) if dec8 else int(rng.choice([33, 40, 64, 65, 81, 82, 100, 128, 129, 200, 224, 255, 256] + ([] if cplx else [384, 512, 777, 1024])))
    n = int(rng.integers(1 << 16, 1 << 19)) + int(rng.integers(0, 5000))
    if dec8:
        n = (int(rng.integers(1 << 18, 1 << 20)) + int(rng.integers(0, 5000))) // DD * DD
    kind = str(rng.choice(["plain", "level", "jumps", "holes", "outliers", "nonfinite", "tone"]))
    kinds[kind] = kinds.get(kind, 0) + 1
    taps = (rng.standard_normal(nt) * np.hamming(nt) * 10.0 ** rng.uniform(-3, 1)).astype(np.float32)
    x = rng.standard_normal(n) + (1j * rng.standard_normal(n) if cplx else 0)
    if kind == "level":
        x = x * 10.0 ** rng.uniform(-30, 30)
    elif kind == "jumps":
        pos = np.sort(rng.integers(0, n, size=int(rng.integers(2, 12))))
        lev = np.ones(n)
        for a in pos:
            lev[a:] = 10.0 ** rng.uniform(-6, 6)
        x = x * lev
    elif kind == "holes":
        for _ in range(int(rng.integers(1, 6))):
            a = int(rng.integers(0, n)); b = min(n, a + int(rng.integers(1, 20000)))
            x[a:b] = 0 if rng.integers(0, 2) else x[a:b] * 1e-42
    elif kind == "outliers":
        for a in rng.integers(0, n, size=int(rng.integers(1, 6))):
            x[a] = 10.0 ** rng.uniform(3, 30) * (1 if rng.integers(0, 2) else -1)
    elif kind == "tone":
        taps = None
    dt = np.complex64 if cplx else np.float32
    x = x.astype(dt)
    if kind == "tone":
        k = np.arange(nt); fc = 0.05
        taps = (np.hamming(nt) * 2 * fc * np.sinc(2 * fc * (k - (nt - 1) / 2.0))).astype(np.float32)
        amp = 10.0 ** rng.uniform(1, 3)
        ph = 2 * np.pi * rng.uniform(0.2, 0.45) * np.arange(n)
        x = (x * 0.05 + amp * (np.exp(1j * ph) if cplx else np.cos(ph))).astype(dt)
    if kind == "nonfinite":
        for a in rng.integers(0, n, size=int(rng.integers(1, 4))):
            v = [np.inf, -np.inf, np.nan][int(rng.integers(0, 3))]
            x[a] = v if not cplx else (complex(v, x[a].imag) if rng.integers(0, 2) else complex(x[a].real, v))  # (one component: real taps keep it out of the other)
    ncut = int(rng.integers(0, 3)); al = 4 * DD if dec8 else (2 if cplx else 4)
    cuts = sorted(set([0, n] + [int(c) // al * al for c in rng.integers(0, n, size=ncut)]))
    misalign = bool(rng.integers(0, 8) == 0)
    truth = fir64(taps, x)
    D = DD
    truth = truth[::D]
    f = G.fir_filter(taps, torch.complex64 if cplx else torch.float32, decimate=D)
    if cplx and not dec8:
        f.set_algo(capi.FIR_TIME_DOMAIN)
    y = run(f, x, cuts, cplx, misalign)
    tag = f"{kind} cplx={cplx} decim={DD} taps={nt} n={n} cuts={cuts} misalign={misalign}"
    if kind == "nonfinite":
        with np.errstate(all="ignore"):
            t32 = truth.astype(dt)
        bad = ~np.isfinite(t32)
        ok = True
        if misalign or (nt > 256 and not dec8) or min(b - a for a, b in zip(cuts[:-1], cuts[1:])) < ((1 << 17) if dec8 else (1 << 16)):  # other kernels (short calls: the register-window kernel's padded taps): a superset of the reference's non-finite outputs
            ok = not np.any(bad & np.isfinite(y))
        else:
            for part in ((np.real, np.imag) if cplx else (np.asarray,)):
                ok &= np.array_equal(np.isnan(part(y)), np.isnan(part(t32))) and np.array_equal(np.isposinf(part(y)), np.isposinf(part(t32))) and np.array_equal(np.isneginf(part(y)), np.isneginf(part(t32)))
        good = np.isfinite(t32) & np.isfinite(y)
        xz = np.where(np.isfinite(x), x, 0).astype(dt)
        r, w = local_err(np.where(good, y, 0), np.where(good, truth, 0), xz, taps, D) if ok else (1.0, -1)
        tag += f" classes_ok={ok} nonfinite_at={np.flatnonzero(~np.isfinite(x))[:4]} worst_at={w}"
    elif kind == "tone":
        ye = O.fir(taps, x, acc64=False)[0][::D]  # the reference's own float32 arithmetic: the sequential sum of transform_reduce, every D-th output kept (oracle restatement, test infrastructure)
        rms = float(np.sqrt(np.mean(np.abs(truth[nt:]) ** 2)))
        e = float(np.max(np.abs(y[nt:] - truth[nt:]) / np.maximum(np.abs(truth[nt:]), rms))); e32 = float(np.max(np.abs(ye[nt:] - truth[nt:]) / np.maximum(np.abs(truth[nt:]), rms)))
        # the contract (include/gr4hip.h): 1e-5, or the reference's float32 error where that is larger -- factor ONE, every shape
        r = 0.0 if e <= max(1e-5, e32) else e
        tag += f" amp={amp:.0f} err={e:.2e} reference_f32={e32:.2e}"
    else:
        r, w = local_err(y, truth, x, taps, D)
        tag += f" worst_at={w} y={y[w]:.6g} truth={truth[w]:.6g} |x| around: {np.abs(x[max(0, D * w - 300):D * w + 1]).max():.3g} / segment max {np.abs(x[max(0, D * w - 4400 * D):D * w + 4400 * D]).max():.3g}"
    cases += 1; worst = max(worst, r)
    if r > 1e-5:
        fails += 1; kfail[kind] = kfail.get(kind, 0) + 1
        if kfail[kind] <= 6: print("FAIL", tag, r, flush=True)
print(f"failures per kind: {kfail}")
print(f"{cases} cases in {time.time() - t0:.0f} s ({kinds}), {fails} above the bar, worst {worst:.3g}")
