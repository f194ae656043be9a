"""developer tool: the f16 band-form decimate-by-8 kernel (csrc/fir_decim_f16.hip) against the frequency-domain kernel (GR4HIP_FIR_NO_DECIM_F16=1) on one box:
error against the float64 oracle (ordinary input; a rejected tone 50 dB above the noise), rate over the tap count, BASELINE configs[2] (FIR + 4 biquads)"""
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


def dev16(x):
    t = torch.empty(x.size + 4, dtype=torch.float32, device="cuda")[4:]
    t.copy_(torch.from_numpy(x)); return t


MODES = [("fd", 1), ("f16", 0)]
na = 8 * 40_000
xa = O.signal_f32(7, na)
xi = (xa * 0.05 + 316.0 * np.cos(2 * np.pi * 0.31 * np.arange(na))).astype(np.float32)
for nt in (200, 256, 300, 520, 777, 1024, 1025):
    b = O.design_taps_hamming_lowpass(nt, 0.05)
    row = []
    for name, sw in MODES:
        capi.developer_switch("GR4HIP_FIR_NO_DECIM_F16", sw)
        for tag, x in (("noise", xa), ("+50dB", xi)):
            truth, _ = O.fir_decim(b, x, 8)
            f = G.fir_filter(b, torch.float32, decimate=8)
            cut = 8 * 17_001
            y = np.concatenate([f.process_bulk(dev16(x[:cut])).cpu().numpy(), f.process_bulk(dev16(x[cut:])).cpu().numpy()])
            row.append(f"{name} {tag} {rel(y, truth):.2e}")
    print(f"taps {nt}: " + "  ".join(row), flush=True)
n = 1 << 27
x = G.synth_f32(n, seed=42)
y = torch.empty(n // 8, dtype=torch.float32, device="cuda")
for name, sw in MODES:
    capi.developer_switch("GR4HIP_FIR_NO_DECIM_F16", sw)
    out = []
    for nt in (200, 256, 520, 777, 1024):
        f = G.fir_filter(O.design_taps_hamming_lowpass(nt, 0.05), torch.float32, decimate=8)
        tt = steady(lambda: f.process_bulk(x, y))
        out.append(f"{nt}: {n / tt / 1e9:.0f}")
    fir = G.fir_filter(O.design_taps_hamming_lowpass(1024, 0.05), torch.float32, decimate=8)
    bi, ai = G.blocks.design_iir(capi.LOWPASS, 8, 0.05, float("nan"), 1.0, capi.BUTTERWORTH)
    iir = G.iir_filter(bi, ai)
    yo = torch.empty_like(y)
    def both():
        fir.process_bulk(x, y); iir.process_bulk(y, yo)
    tt = steady(both)
    print(f"{name}: decimate-by-8 G input samples/s " + "  ".join(out) + f"   configs[2] (FIR + 4 biquads) {n / tt / 1e9:.0f}", flush=True)
