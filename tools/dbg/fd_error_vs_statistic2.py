"""K = err / sqrt(R4) of the fused fast convolution with the guard OFF (explicit CHAIN_FUSED_FD): rectangular / windowed, 8192 / small frames; R4 = w2 nf P_in / rms_k(|Y_k|^2)"""
import sys
sys.path.insert(0, "tests")
import numpy as np, torch
import oracle_lib as O
import gnuradio4_amd as G
def dev(a): return torch.from_numpy(np.ascontiguousarray(a)).cuda()
worst = {}
for N, window, wid in ((8192, "None", 0), (8192, "Hann", 3), (8192, "BlackmanHarris", 7), (1024, "Hann", 3), (256, "None", 0), (4096, "Kaiser", 11)):
    w = O.window(wid, N).astype(np.float64) if wid else np.ones(N)
    w2 = float(np.mean(w * w))
    frames_blk = 8192 // N
    n = 10 * 8192
    for ntaps, fc, amp, f0 in ((256, 0.2, 0, 0), (256, 0.02, 0, 0), (256, 0.005, 0, 0), (256, 0.0025, 0, 0), (100, 0.01, 0, 0), (256, 0.05, 3, 0.3), (256, 0.05, 10, 0.3), (256, 0.05, 100, 0.3),
                                (256, 0.01, 30, 0.3), (256, 0.02, 5, 0.06), (256, 0.02, 20, 0.033), (33, 0.1, 10, 0.4), (256, 0.05, 1000, 0.3)):
        b = O.design_taps_hamming_lowpass(ntaps, fc)
        x = O.signal_c32(7, n, tone_frel=f0, tone_amp=float(amp))
        truth, _ = O.chain(b, x, N, wid, truth=True)
        got = G.Chain(b, N, window, G.capi.CHAIN_FUSED_FD).process_bulk(dev(x)).cpu().numpy().astype(np.float64).ravel()
        T = truth.reshape(-1, N); Gt = got.reshape(-1, N)
        rms = np.sqrt(np.mean(T ** 2, axis=1, keepdims=True))
        e = np.max(np.abs(Gt - T) / np.maximum(np.abs(T), rms), axis=1)          # per spectrum
        eb = e.reshape(-1, frames_blk).max(axis=1)                               # per 8192-sample block (what the kernel judges)
        pin = np.mean(np.abs(x.reshape(-1, 8192)) ** 2, axis=1)                   # per block
        s4 = np.mean(T.reshape(-1, 8192) ** 2, axis=1)
        r4 = w2 * N * pin / np.sqrt(s4)
        k = eb / np.sqrt(r4)
        i = int(np.argmax(k))
        key = (N, window)
        worst[key] = max(worst.get(key, 0), k.max())
        print(f"N={N} {window:14s} taps {ntaps} fc {fc} amp {amp}@{f0}: err {eb.max():.3g}  R4 {r4[int(np.argmax(eb))]:.3g}  K = err / sqrt(R4) worst {k.max():.3g}")
print({f"{k[0]} {k[1]}": f"{v:.3g}" for k, v in worst.items()})
