"""Neighbours in the filter's launch (gr4hip_fir_set_prologue / _epilogue) against the same blocks launched one after the other; 2^27-sample streams resident in HBM,
median launch times (HIP events).  Rows: gain -> FIR (folded into the taps), add -> FIR (load hook), the channeliser front end rotator -> decimate-by-8 complex FIR
(load hook; the only intermediate is the decimated stream) -> 1024-point power spectrum."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import gnuradio4_amd as G
from bench_merged import timed

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))


def lowpass(ntaps, fc):
    k = np.arange(ntaps, dtype=np.float64)
    t = np.hamming(ntaps) * 2 * fc * np.sinc(2 * fc * (k - (ntaps - 1) / 2.0))
    return (t / t.sum()).astype(np.float32)


def main():
    n = 1 << 27
    x = G.synth_f32(n, seed=1)
    y = torch.empty_like(x)
    tmp = torch.empty_like(x)
    for ntaps in (64, 256):
        b = lowpass(ntaps, 0.1)
        for name, pre in (("gain -> fir (folded into the taps)", [("Multiply", 2.0)]), ("add -> fir (load hook, register-window kernel)", [("Add", 0.5)])):
            plain = G.fir_filter(b, torch.float32)
            fused = G.fir_filter(b, torch.float32)
            fused.set_prologue(G.Merged(torch.float32, pre))
            m = G.Merged(torch.float32, pre)
            t_f = timed(lambda: fused.process_bulk(x, y), reps=10)
            t_s = timed(lambda: (m.process_bulk(x, tmp), plain.process_bulk(tmp, y)), reps=10)
            print(json.dumps({"chain": f"{name}, float, {ntaps} taps", "n": n, "one_launch_Gsamples_s": round(n / t_f / 1e6, 1), "two_launches_Gsamples_s": round(n / t_s / 1e6, 1)}), flush=True)
    # the channeliser: rotator -> decimate-by-8 complex FIR (64 taps) -> 1024-point |X|^2
    nc = 1 << 26
    xc = G.synth_c32(nc, seed=2)
    D, N = 8, 1024
    b = lowpass(64, 0.05)
    yd = torch.empty(nc // D, dtype=torch.complex64, device="cuda")
    xr = torch.empty_like(xc)
    spec = torch.empty((nc // D // N, N), dtype=torch.float32, device="cuda")
    fft = G.FFT(N, "Hann")
    fir_h = G.fir_filter(b, torch.complex64, decimate=D)
    fir_h.set_prologue(G.Merged(torch.complex64, [("Rotator", 0.3, 0.25)]))
    fir_p = G.fir_filter(b, torch.complex64, decimate=D)
    rot = G.Rotator(phase_increment=0.3, initial_phase=0.25)
    t_f = timed(lambda: (fir_h.process_bulk(xc, yd), fft.mag2(yd, spec)), reps=10)
    t_s = timed(lambda: (rot.process_bulk(xc, xr), fir_p.process_bulk(xr, yd), fft.mag2(yd, spec)), reps=10)
    print(json.dumps({"chain": "rotator -> decimate-by-8 complex FIR (64 taps) -> 1024-pt Hann |X|^2", "n": nc, "fused_two_launches_Gsamples_s": round(nc / t_f / 1e6, 1),
                      "three_launches_Gsamples_s": round(nc / t_s / 1e6, 1), "hbm_bytes_per_input_sample_fused": 8 + 2 * 8 / D + 4 / D, "unfused": 8 + 16 + 2 * 8 / D + 4 / D}), flush=True)


if __name__ == "__main__":
    main()
