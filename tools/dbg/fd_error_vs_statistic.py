"""round 6 experiment: where does the fused fast convolution (guard OFF) really leave the 1e-5 bar?  Its error is c * eps * rms(x) per output bin; the parity metric normalises by
max(|truth_k|, rms_k(truth)) with truth = |Y_k|^2 -- so what matters is  R4 = mean|x|^2 N / sqrt(mean_k |Y_k|^4)  (input power per bin over the rms of the OUTPUT spectrum's mag2),
not the power ratio the guard uses today (P_out / P_in < 0.08)."""
import sys
sys.path.insert(0, "tests")
import numpy as np, torch
import oracle_lib as O
import gnuradio4_amd as G
def rel_frames(got, truth, N):
    got = np.asarray(got, np.float64).reshape(-1, N); truth = np.asarray(truth).reshape(-1, N)
    rms = np.sqrt(np.mean(truth ** 2, axis=1, keepdims=True))
    return np.max(np.abs(got - truth) / np.maximum(np.abs(truth), rms), axis=1)
def dev(a): return torch.from_numpy(np.ascontiguousarray(a)).cuda()
N, frames = 8192, 12
n = frames * N
rng = np.random.default_rng(1)
print("case: worst frame error | power ratio (guard: < 0.08 marks) | R4 = P_in N / rms_k(|Y_k|^2)")
for ntaps, fc, amp, f0, kind in ((256, 0.2, 0, 0, "noise"), (256, 0.05, 0, 0, "noise"), (256, 0.02, 0, 0, "noise"), (256, 0.01, 0, 0, "noise"), (256, 0.005, 0, 0, "noise"), (256, 0.0025, 0, 0, "noise"),
                                   (256, 0.05, 3, 0.3, "tone"), (256, 0.05, 10, 0.3, "tone"), (256, 0.05, 30, 0.3, "tone"), (256, 0.05, 100, 0.3, "tone"), (256, 0.01, 10, 0.3, "tone"), (256, 0.01, 30, 0.3, "tone"),
                                   (256, 0.05, 30, 0.02, "tone in band"), (256, 0.005, 30, 0.001, "tone in band"), (64, 0.01, 0, 0, "noise"), (256, 0.05, 1000, 0.3, "tone")):
    b = O.design_taps_hamming_lowpass(ntaps, fc)
    x = O.signal_c32(7, n, tone_frel=f0, tone_amp=float(amp))
    truth, _ = O.chain(b, x, N, 0, truth=True)
    ch = G.Chain(b, N, "None", G.capi.CHAIN_FUSED_FD)  # explicit fused algorithm: no guard, never switches
    got = ch.process_bulk(dev(x)).cpu().numpy().ravel()
    e = rel_frames(got, truth, N)
    T = truth.reshape(-1, N)
    pin = np.mean(np.abs(x.reshape(-1, N)) ** 2, axis=1) * N   # = mean_k |X_k|^2 (Parseval)
    ratio = np.sum(T, axis=1) / (pin * N)
    r4 = pin / np.sqrt(np.mean(T ** 2, axis=1))
    w = int(np.argmax(e))
    print(f"taps {ntaps} fc {fc} {kind} amp {amp}@{f0}: err {e.max():.3g} (frame {w}) | ratio {ratio[w]:.3g} | R4 {r4[w]:.3g}   (err / R4 = {e.max() / r4[w]:.3g})")
