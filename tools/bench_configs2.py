"""BASELINE configs[2]: decimate-by-8 1024-tap FIR -> 4 biquads, 2^27 input samples resident in HBM: gr4hip_fir_iir_process with the cascade as the decimator's store
epilogue (ONE launch) against the same call as two launches (mode GR4HIP_FIR_IIR_TWO_LAUNCHES) and against the decimator alone; per guard mode."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import gnuradio4_amd as G
import gnuradio4_amd.blocks as B
from gnuradio4_amd import capi
from bench_merged import timed


def lowpass(ntaps, fc):
    k = np.arange(ntaps, dtype=np.float64)
    t = np.hamming(ntaps) * 2 * fc * np.sinc(2 * fc * (k - (ntaps - 1) / 2.0))
    return (t / t.sum()).astype(np.float32)


n = (1 << 27) // 7168 * 7168
x = G.synth_f32(n, seed=42)
b, a = B.design_iir(capi.LOWPASS, 8, 0.05, float("nan"), 1.0, capi.BUTTERWORTH)
y = torch.empty(n // 8, dtype=torch.float32, device="cuda")
for guard, gname in ((capi.GUARD_STRICT, "strict"), (capi.GUARD_OFF, "off")):
    row = {"guard": gname, "n": n}
    for mode in ("one_launch", "two_launches"):
        m = capi.FIR_IIR_TWO_LAUNCHES if mode == "two_launches" else capi.FIR_IIR_ONE_LAUNCH
        fir = G.fir_filter(lowpass(1024, 0.05), torch.float32, decimate=8)
        capi.check(capi.lib().gr4hip_fir_set_guard_mode(fir._h, guard), "guard")
        iir = G.iir_filter(b, a)
        t = timed(lambda: B.fir_iir_process(fir, iir, x, y, mode=m), reps=20, warm=5)
        row[mode + "_G_input_samples_s"] = round(n / t / 1e6, 1)
        row[mode + "_ms"] = round(t, 4)
    fir = G.fir_filter(lowpass(1024, 0.05), torch.float32, decimate=8)
    capi.check(capi.lib().gr4hip_fir_set_guard_mode(fir._h, guard), "guard")
    t = timed(lambda: fir.process_bulk(x, y), reps=20, warm=5)
    row["decimator_alone_G_input_samples_s"] = round(n / t / 1e6, 1)
    row["hbm_frac_one_launch_at_4.5_B"] = round(row["one_launch_G_input_samples_s"] * 1e9 * 4.5 / 8e12, 3)
    print(json.dumps(row), flush=True)
