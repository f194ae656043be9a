"""developer tool: decimate by 8 / 16 / 32 on the f16 band-form kernel (csrc/fir_decim_f16.hip) against the kernels it replaces (GR4HIP_FIR_NO_DECIM_F16=1): error on
noise against the float64 oracle and rate, per (D, taps)"""
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


n = 1 << 27
x = G.synth_f32(n, seed=42)
na = 32 * 12_000
xa = O.signal_f32(7, na)
ta = torch.empty(na + 4, dtype=torch.float32, device="cuda")[4:]; ta.copy_(torch.from_numpy(xa))
for D, Ks in ((8, (100, 256, 1024)), (16, (33, 64, 128, 256, 512, 897)), (32, (64, 128, 256, 512, 641))):
    y = torch.empty(n // D, dtype=torch.float32, device="cuda")
    for K in Ks:
        b = (np.hamming(K) / K).astype(np.float32)
        truth, _ = O.fir_decim(b, xa, D)
        row = []
        for name, sw in (("old", 1), ("f16", 0)):
            capi.developer_switch("GR4HIP_FIR_NO_DECIM_F16", sw)
            f = G.fir_filter(b, torch.float32, decimate=D)
            cut = D * 5000
            e = rel(np.concatenate([f.process_bulk(ta[:cut]).cpu().numpy(), f.process_bulk(ta[cut:]).cpu().numpy()]), truth)
            f2 = G.fir_filter(b, torch.float32, decimate=D)
            t = steady(lambda: f2.process_bulk(x, y))
            row.append(f"{name} {n / t / 1e9:6.0f} G (err {e:.1e})")
        print(f"D={D:2d} K={K:4d}: " + "   ".join(row), flush=True)
capi.developer_switch("GR4HIP_FIR_NO_DECIM_F16", 0)
