#!/usr/bin/env python
"""developer tool: random IIR cascades, decimating / interpolating FIRs, complex FIRs and FFT-block frames (any size) against float64 scipy / numpy, streams cut
into several calls of ragged length.  usage: fuzz_blocks.py [seconds = 120] [seed = 0]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from scipy.signal import lfilter, sosfilt, butter, cheby1, upfirdn
sys.path.insert(0, "tests")
import gnuradio4_amd as G
import oracle_lib as O
from gnuradio4_amd import capi

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
def rel(a, b):
    rms = np.sqrt(np.mean(np.abs(b) ** 2)) + 1e-30
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), rms)))
def cuts_of(n, align):
    c = sorted(set([0, n] + [int(v) // align * align for v in rng.integers(0, n, size=int(rng.integers(0, 3)))]))
    return [(a, b) for a, b in zip(c[:-1], c[1:]) if b > a]
t0 = time.time(); cases = 0; worst = {}
while time.time() - t0 < secs:
    kind = int(rng.integers(0, 5))
    if kind == 0:  # IIR cascade of biquads
        order = int(rng.choice([2, 4, 6, 8, 10, 12, 16])); fc = float(rng.uniform(0.01, 0.45))
        sos = butter(order, 2 * fc, output="sos") if rng.random() < 0.5 else cheby1(order, 1.0, 2 * fc, output="sos")
        n = int(rng.integers(1 << 12, 1 << 20)) + int(rng.integers(0, 999))
        x = rng.standard_normal(n).astype(np.float32)
        f = G.iir_filter(sos[:, :3], sos[:, 3:])
        y = np.concatenate([f.process_bulk(torch.from_numpy(x[a:b]).cuda()).cpu().numpy() for a, b in cuts_of(n, 1)])
        # truth: the SAME float32 coefficients the block holds, float64 arithmetic (the oracle's cascade); the contract's bound: the reference's own float32 cascade -- its
        # default form section by section (oracle: gr4o_iir_cascade_f32) -- factor ONE
        secs_ = O.make_sections([(bb, aa) for bb, aa in zip(sos[:, :3].astype(np.float32), sos[:, 3:].astype(np.float32))])
        truth = O.iir_cascade(secs_, x, 3, f64=True)
        r = rel(y, truth); r32 = min(rel(O.iir_cascade(secs_, x, form, f64=False), truth) for form in (O.DF_I, O.DF_II))
        tag = f"iir order={order} fc={fc:.3f} n={n} (reference float32 cascade {r32:.1e})"; bar = max(1e-5, r32)
    elif kind == 1:  # decimating FIR
        D = int(rng.integers(2, 17)); nt = int(rng.choice([16, 64, 100, 256, 320, 500, 1024]))
        taps = (rng.standard_normal(nt) * np.hamming(nt) / np.sqrt(nt)).astype(np.float32)
        n = (int(rng.integers(1 << 15, 1 << 21)) // D) * D
        x = rng.standard_normal(n).astype(np.float32)
        f = G.fir_filter(taps, torch.float32, decimate=D)
        y = np.concatenate([f.process_bulk(torch.from_numpy(x[a:b]).cuda()).cpu().numpy() for a, b in cuts_of(n, 4 * D)])
        r = rel(y, lfilter(taps.astype(np.float64), [1.0], x.astype(np.float64))[D - 1::D] if False else lfilter(taps.astype(np.float64), [1.0], x.astype(np.float64))[::D]); tag = f"decim D={D} taps={nt} n={n}"; bar = 1e-5
    elif kind == 2:  # interpolating FIR
        L = int(rng.integers(2, 17)); nt = int(rng.choice([16, 64, 128, 256, 512]))
        taps = (rng.standard_normal(nt) * np.hamming(nt) / np.sqrt(nt)).astype(np.float32)
        n = int(rng.integers(1 << 13, 1 << 18)) // 4 * 4
        cplx = rng.random() < 0.4
        x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64) if cplx else rng.standard_normal(n).astype(np.float32)
        f = G.fir_interpolator(taps, L, torch.complex64 if cplx else torch.float32)
        y = np.concatenate([f.process_bulk(torch.from_numpy(x[a:b]).cuda()).cpu().numpy() for a, b in cuts_of(n, 4)])
        truth = L * upfirdn(taps.astype(np.float64), x.astype(np.complex128 if cplx else np.float64), up=L)[: n * L]
        r = rel(y, truth); tag = f"interp L={L} taps={nt} n={n} complex={cplx}"; bar = 1e-5
    elif kind == 3:  # complex FIR, any algorithm the library picks
        nt = int(rng.choice([3, 32, 33, 64, 96, 97, 128, 200, 256, 300])); n = int(rng.integers(1 << 14, 1 << 21)) // 2 * 2
        taps = (rng.standard_normal(nt) * np.hamming(nt) / np.sqrt(nt)).astype(np.float32)
        x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
        f = G.fir_filter(taps, torch.complex64)
        y = np.concatenate([f.process_bulk(torch.from_numpy(x[a:b]).cuda()).cpu().numpy() for a, b in cuts_of(n, 2)])
        r = rel(y, lfilter(taps.astype(np.float64), [1.0], x.astype(np.complex128))); tag = f"cfir taps={nt} n={n}"; bar = 1e-5
    else:  # FFT block, any size
        N = int(rng.choice([int(2 ** rng.integers(1, 14)), int(rng.integers(2, 9000)), int(rng.choice([1000, 1536, 3000, 6000, 7776, 6561]))]))
        frames = int(rng.integers(1, 40))
        x = (rng.standard_normal(frames * N) + 1j * rng.standard_normal(frames * N)).astype(np.complex64)
        out = G.FFT(N, "None").process_bulk(torch.from_numpy(x).cuda())
        X = np.fft.fft(x.reshape(frames, N).astype(np.complex128), axis=1)
        r = max(rel(out["re"].cpu().numpy(), X.real), rel(out["im"].cpu().numpy(), X.imag)); tag = f"fft N={N} frames={frames}"; bar = 1e-5
    cases += 1; worst[tag.split()[0]] = max(worst.get(tag.split()[0], 0.0), r)
    if not (r <= bar): print("FAIL", tag, r, flush=True)
print(f"{cases} cases in {time.time() - t0:.0f} s; worst relative errors: " + ", ".join(f"{k} {v:.2e}" for k, v in sorted(worst.items())))
