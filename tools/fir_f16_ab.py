"""developer tool: the two-term f16 FIR kernels (fir_f16.hip) against the three-term bf16 ones (fir_bf16.hip) on one box -- error against the float64 oracle
and rate, float FIR over tap counts and BASELINE configs[3]."""
import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import gnuradio4_amd as G
import oracle_lib as O
from _timing import steady
from gnuradio4_amd import capi


def rel(got, truth):
    rms = float(np.sqrt(np.mean(np.abs(truth) ** 2)))
    return float(np.max(np.abs(got - truth) / np.maximum(np.abs(truth), rms)))


MODES = [("bf16x3", {"GR4HIP_FIR_NO_F16X2": 1}), ("f16x2", {"GR4HIP_FIR_NO_F16X2": 0}), ("f16x2 unguarded", {"GR4HIP_FIR_NO_F16X2": 0})]


def mode(m):
    for k, v in m.items():
        capi.developer_switch(k, v)


# ---- accuracy: ordinary input, and a rejected tone 50 dB above the passed noise
na = 1 << 18
xa = O.signal_f32(7, na)
t = np.arange(na)
xi = (xa * 0.05 + 316.0 * np.cos(2 * np.pi * 0.31 * t)).astype(np.float32)  # interferer far in the stop band
for nt in (64, 100, 200, 256, 512):
    b = O.design_taps_hamming_lowpass(nt, 0.1)
    row = []
    for name, m in MODES:
        mode(m)
        for tag, x in (("noise", xa), ("+50dB", xi)):
            truth, _ = O.fir(b, x)
            f = G.fir_filter(b, torch.float32)
            if "unguarded" in name:
                f.set_guard_mode(capi.GUARD_OFF)
            y = f.process_bulk(torch.from_numpy(x).cuda()).cpu().numpy()
            row.append(f"{name} {tag} {rel(y, truth):.2e}")
    print(f"taps {nt}: " + "  ".join(row), flush=True)

# ---- the guard's price where it acts: the whole stream under the interferer (every segment judged, rejected, and evaluated again as float32 sums)
ni = 1 << 26
xid = torch.from_numpy(np.tile(xi, ni // na)).cuda()
yid = torch.empty_like(xid)
mode(MODES[1][1])
for nt in (64, 256):
    f = G.fir_filter(O.design_taps_hamming_lowpass(nt, 0.1), torch.float32)
    tt = steady(lambda: f.process_bulk(xid, yid))
    f32 = G.fir_filter(O.design_taps_hamming_lowpass(nt, 0.1), torch.float32)
    f32.set_algo(capi.FIR_TIME_DOMAIN_F32)
    t32 = steady(lambda: f32.process_bulk(xid, yid))
    print(f"taps {nt}, every segment rejected: {ni / tt / 1e9:.1f} Gsamples/s (GR4HIP_FIR_TIME_DOMAIN_F32 on the same stream: {ni / t32 / 1e9:.1f})", flush=True)
# ---- rates
n = 1 << 28
x = G.synth_f32(n, seed=42)
y = torch.empty_like(x)
for name, m in MODES:
    mode(m)
    out = []
    for nt in (48, 64, 100, 128, 200, 256, 512, 1024):
        f = G.fir_filter(O.design_taps_hamming_lowpass(nt, 0.05), torch.float32)
        if "unguarded" in name:
            f.set_guard_mode(capi.GUARD_OFF)
        tt = steady(lambda: f.process_bulk(x, y))
        out.append(f"{nt}: {n / tt / 1e9:.0f}")
    nch, n2 = 64, 1 << 22
    xb = x[: nch * n2].view(nch, n2)
    fb = G.FirBatched(np.stack([O.design_taps_hamming_lowpass(256, 0.05 + 0.005 * c) for c in range(nch)]))
    yb = torch.empty_like(xb)
    tt = steady(lambda: fb.process_bulk(xb, yb))
    print(f"{name}: float FIR Gsamples/s " + "  ".join(out) + f"   configs[3] {nch * n2 / tt / 1e9:.0f}", flush=True)
