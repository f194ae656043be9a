"""developer tool: single-pass vs three-pass IIR on 2^26 samples -- rates and agreement.  usage: iir_onepass_cmp.py one; GR4HIP_IIR_THREE_PASS=1 iir_onepass_cmp.py three"""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import gnuradio4_amd as G
from gnuradio4_amd import capi
mode = sys.argv[1]
n = 1 << 26
x = G.synth_f32(n, seed=3)
for name, (b, a) in {"biquad4": G.blocks.design_iir(capi.LOWPASS, 8, 0.05, float("nan"), 1.0, capi.BUTTERWORTH),
                     "pole1": (np.array([[0.3]], np.float32), np.array([[1.0, -0.7]], np.float32))}.items():
    f = G.iir_filter(b, a)
    y = torch.empty_like(x)
    f.process_bulk(x[: n // 3], y[: n // 3]); f.process_bulk(x[n // 3:], y[n // 3:])
    torch.cuda.synchronize()
    np.save(f"/tmp/iir_{name}_{mode}.npy", y.cpu().numpy())
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    for _ in range(3): f.process_bulk(x, y)
    t0.record()
    for _ in range(10): f.process_bulk(x, y)
    t1.record(); t1.synchronize()
    print(mode, name, "%.1f Gsamples/s" % (n * 10 / t0.elapsed_time(t1) / 1e6))
if mode == "three":
    for name in ("biquad4", "pole1"):
        a_, b_ = np.load(f"/tmp/iir_{name}_one.npy"), np.load(f"/tmp/iir_{name}_three.npy")
        rms = np.sqrt(np.mean(b_.astype(np.float64) ** 2))
        print(name, "one-pass vs three-pass: max |diff| / rms = %.3g" % (np.max(np.abs(a_.astype(np.float64) - b_)) / rms))
