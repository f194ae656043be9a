import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
import gnuradio4_amd as G
import oracle_lib as O
n = 96 * 7168
for ntaps, fc in ((1024, 0.05), (1024, 0.02), (256, 0.05), (1025, 0.01)):
    b = O.design_taps_hamming_lowpass(ntaps, fc)
    x = O.signal_f32(41, n, tone_amp=0.0)
    truth, _ = O.fir_decim(b, x, 8)
    f = G.fir_filter(b, torch.float32, decimate=8)
    G.capi.check(G.capi.lib().gr4hip_fir_set_guard_mode(f._h, G.capi.GUARD_OFF), "g")
    y = f.process_bulk(torch.from_numpy(x).cuda()).cpu().numpy().astype(np.float64)
    in_rms = np.sqrt(np.mean(x.astype(np.float64) ** 2)); out_rms = np.sqrt(np.mean(truth ** 2))
    e = y - truth
    print(f"taps {ntaps} fc {fc}: err max {np.max(np.abs(e))/in_rms:.2e} rms {np.sqrt(np.mean(e**2))/in_rms:.2e} of the input rms; power ratio out/in {(out_rms/in_rms)**2:.4f}; max err / out rms {np.max(np.abs(e))/out_rms:.2e}")
    f2 = G.fir_filter(b, torch.float32, decimate=8)
    y2 = f2.process_bulk(torch.from_numpy(x).cuda()).cpu().numpy().astype(np.float64)
    print("   strict guard: max err / out rms %.2e" % (np.max(np.abs(y2 - truth)) / out_rms))
