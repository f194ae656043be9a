#!/usr/bin/env python
"""developer tool: random fusion configurations against float64 numpy -- element-wise programs (float / complex, rotators), FIR filters with random prologues /
epilogues over random (taps, decimation, type, span cuts: every decimator kernel and its hook sites), decimating FIR + IIR through gr4hip_fir_iir_process in every mode.
usage: fuzz_fusion.py [seconds = 120] [seed = 0]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from scipy import signal
sys.path.insert(0, "tests")
import gnuradio4_amd as G
import oracle_lib as O
import gnuradio4_amd.blocks as B
from gnuradio4_amd import capi

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
t0 = time.time(); cases = 0; worst = {}; bad = 0
def rel(a, b):
    rms = np.sqrt(np.mean(np.abs(b) ** 2)) + 1e-30
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), rms)))
def rand_prog(cplx, maxlen=4):
    ops = []
    for _ in range(int(rng.integers(0, maxlen + 1))):
        k = rng.integers(0, 5 if cplx else 4)
        if k == 4:
            ops.append(("Rotator", float(np.float32(rng.uniform(-3, 3))), float(np.float32(rng.uniform(-3, 3)))))
        else:
            v = complex(np.complex64(rng.uniform(0.3, 2) * np.exp(1j * rng.uniform(0, 6.28)))) if cplx and rng.random() < 0.7 else float(np.float32(rng.uniform(0.3, 2) * rng.choice([-1, 1])))
            ops.append((["Add", "Subtract", "Multiply", "Divide"][k], v))
    return ops
def run_prog(x, ops, pos0=0):
    y = x.astype(np.complex128 if np.iscomplexobj(x) else np.float64)
    for op in ops:
        if op[0] == "Rotator":
            k = np.arange(1, len(y) + 1, dtype=np.float64) + pos0
            y = y * np.exp(1j * (float(np.float32(op[2])) + k * float(np.float32(op[1]))))
        else:
            v = np.complex128(np.complex64(op[1])) if np.iscomplexobj(y) else np.float64(np.float32(op[1]))
            y = {"Add": y + v, "Subtract": y - v, "Multiply": y * v, "Divide": y / v}[op[0]]
    return y
def dev(a): return torch.from_numpy(np.ascontiguousarray(a)).cuda()
while time.time() - t0 < secs:
    kind = int(rng.integers(0, 4))
    if kind == 0:  # a program on its own
        cplx = bool(rng.integers(0, 2)); n = int(rng.integers(1, 1 << 18))
        x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64) if cplx else rng.standard_normal(n).astype(np.float32)
        ops = rand_prog(cplx, 8)
        if not ops: continue
        m = G.Merged(torch.complex64 if cplx else torch.float32, ops)
        cuts = sorted(set([0, n] + [int(c) for c in rng.integers(0, n + 1, size=int(rng.integers(0, 3)))]))
        y = np.concatenate([m.process_bulk(dev(x[a:b])).cpu().numpy() for a, b in zip(cuts[:-1], cuts[1:]) if b > a])
        r = rel(y, run_prog(x, ops)); tag = f"program cplx={cplx} n={n} ops={ops} cuts={cuts}"; key = "program"
    elif kind in (1, 2):  # FIR with neighbours
        cplx = bool(rng.integers(0, 2))
        D = int(rng.choice([1, 1, 2, 3, 4, 5, 8, 8, 10, 12, 16, 20]))
        nt = int(rng.choice([1, 7, 16, 31, 33, 64, 65, 87, 88, 100, 128, 200, 256, 300, 520, 1024]))
        n_out = int(rng.choice([rng.integers(1, 2000), rng.integers(1 << 14, 1 << 16), rng.integers(1 << 16, 1 << 18)]))
        n = n_out * D
        taps = (rng.standard_normal(nt) / np.sqrt(nt)).astype(np.float32)
        x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64) if cplx else rng.standard_normal(n).astype(np.float32)
        pre, post = rand_prog(cplx, 3), rand_prog(cplx, 3)
        dt = torch.complex64 if cplx else torch.float32
        f = G.fir_filter(taps, dt, decimate=D)
        if pre: f.set_prologue(G.Merged(dt, pre))
        if post: f.set_epilogue(G.Merged(dt, post))
        cuts = sorted(set([0, n] + [int(c) // (4 * D) * (4 * D) for c in rng.integers(0, n + 1, size=int(rng.integers(0, 3)))]))
        xd = dev(x)
        y = np.concatenate([f.process_bulk(xd[a:b]).cpu().numpy() for a, b in zip(cuts[:-1], cuts[1:]) if b > a])
        mid = np.convolve(run_prog(x, pre), taps.astype(np.float64))[:n][::D]
        truth = run_prog(mid, post)
        r = rel(y, truth); tag = f"fir cplx={cplx} D={D} taps={nt} n={n} pre={pre} post={post} cuts={cuts}"; key = f"fir{'_c' if cplx else '_f'}{'_hooked' if pre or post else ''}"
    else:  # decimating FIR + IIR cascade
        order = int(rng.choice([2, 4, 6, 8])); fc = float(rng.uniform(0.02, 0.2)); nblocks = int(rng.integers(1, 400))
        n = nblocks * 7168 + 8 * int(rng.integers(0, 900))
        nt = int(rng.choice([64, 500, 1000, 1024]))
        k = np.arange(nt); t = np.hamming(nt) * np.sinc(0.1 * (k - (nt - 1) / 2)); taps = (t / t.sum()).astype(np.float32)
        b, a = B.design_iir(capi.LOWPASS, order, fc, float("nan"), 1.0, capi.BUTTERWORTH)
        x = rng.standard_normal(n).astype(np.float32)
        mode = int(rng.integers(0, 3))
        fir, iir = G.fir_filter(taps, torch.float32, decimate=8), G.iir_filter(b, a)
        cut = int(rng.integers(0, n // 8 + 1)) * 8
        xd = dev(x)
        y = np.concatenate([B.fir_iir_process(fir, iir, xd[:cut], mode=mode).cpu().numpy(), B.fir_iir_process(fir, iir, xd[cut:], mode=mode).cpu().numpy()])
        mid = np.convolve(x.astype(np.float64), taps.astype(np.float64))[:n][::8].astype(np.float32).astype(np.float64)
        sos = np.array([[bb[0], bb[1], bb[2], aa[0], aa[1], aa[2]] for bb, aa in zip(np.asarray(b, np.float64).reshape(-1, 3), np.asarray(a, np.float64).reshape(-1, 3))])
        truth = signal.sosfilt(sos, mid)
        r = rel(y, truth); tag = f"fir_iir mode={mode} order={order} fc={fc:.3f} n={n} cut={cut} taps={nt}"; key = f"fir_iir_mode{mode}"
        if r > 1e-5:  # the contract's second clause: the reference's own float32 cascade on the same decimated stream (oracle restatement, test infrastructure), factor ONE
            sections = O.make_sections([(bb, aa) for bb, aa in zip(np.asarray(b, np.float32).reshape(-1, 3), np.asarray(a, np.float32).reshape(-1, 3))])
            t64 = O.iir_cascade(sections, mid.astype(np.float32), 3, f64=True)
            e_ref = min(rel(O.iir_cascade(sections, mid.astype(np.float32), form, f64=False), t64) for form in (O.DF_I, O.DF_II))
            tag += f" reference_f32={e_ref:.2e}"
            if r <= e_ref: r = 0.0
    cases += 1
    worst[key] = max(worst.get(key, 0.0), r)
    if not (r <= 1e-5):
        bad += 1
        print("FAIL %.3e %s" % (r, tag), flush=True)
print("cases", cases, "failures", bad, "worst per kind", {k: "%.2e" % v for k, v in sorted(worst.items())})
