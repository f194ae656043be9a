"""chain_td16_kernel (round 6): accuracy against the float64 oracle and rates, in-stream (every frame marked by the fused launch) and settled (the stream moved to the time domain)"""
import sys, time
sys.path.insert(0, "tests")
import numpy as np, torch
import oracle_lib as O
import gnuradio4_amd as G
def rel(got, truth):
    got = np.asarray(got, np.float64).ravel(); truth = np.asarray(truth).ravel()
    rms = np.sqrt(np.mean(truth ** 2)); return float(np.max(np.abs(got - truth) / np.maximum(np.abs(truth), rms)))
def dev(a): return torch.from_numpy(np.ascontiguousarray(a)).cuda()
for N, ntaps, window, wid, fc, amp, f0 in ((8192, 256, "None", 0, 0.005, 1.0, 0.1), (8192, 256, "Hann", 3, 0.005, 1.0, 0.1), (8192, 200, "None", 0, 0.01, 0.0, 0.1), (1024, 200, "Hamming", 2, 0.005, 1.0, 0.1),
                                            (8192, 256, "None", 0, 0.02, 300.0, 0.31), (8192, 256, "Hann", 3, 0.02, 30.0, 0.31), (256, 100, "BlackmanHarris", 7, 0.004, 3.0, 0.2)):
    b = O.design_taps_hamming_lowpass(ntaps, fc)
    n = 24 * 8192
    x = O.signal_c32(5, n, tone_frel=f0, tone_amp=amp)
    truth, _ = O.chain(b, x, N, wid, truth=True)
    ch = G.Chain(b, N, window)
    d = dev(x)
    y1 = ch.process_bulk(d).cpu().numpy()
    r1 = ch.last_power_ratio()
    y2 = ch.process_bulk(d[: 8 * 8192]).cpu().numpy()          # (moved?  then this call is the settled path; its history is the end of x)
    t2, _ = O.chain(b, np.concatenate([x[-(ntaps - 1):], x[: 8 * 8192]]), N, wid, truth=True) if False else (None, None)
    r2 = ch.last_power_ratio()
    # settled from a fresh start: force the move with a first call, reset the oracle's view by comparing a second fresh chain's settled output on the same x
    ch2 = G.Chain(b, N, window); ch2.process_bulk(d); ch2.last_power_ratio(); ch2.process_bulk(d[:8192]); moved = ch2.last_power_ratio()[1]
    y3 = None
    if moved:
        xs = np.concatenate([x, x[:8192], x])
        t3, _ = O.chain(b, xs, N, wid, truth=True)
        y3 = ch2.process_bulk(d).cpu().numpy()
        e3 = rel(y3, t3[-y3.size:])
    print(f"N={N} taps={ntaps} {window} fc={fc} tone {amp}@{f0}: in-stream err {rel(y1, truth):.3g}  ratio {r1[0]:.3g} moved {r2[1]}  settled err {e3 if moved else float('nan'):.3g}")
# rates
n = 1 << 27
x = G.synth_c32(n, seed=42)
out = torch.empty((n // 8192, 8192), dtype=torch.float32, device="cuda")
def lowpass(nt, fc):
    k = np.arange(nt); t = np.hamming(nt) * 2 * fc * np.sinc(2 * fc * (k - (nt - 1) / 2)); return (t / t.sum()).astype(np.float32)
for window in ("None", "Hann"):
    for fc in (0.005, 0.025, 0.05):
        ch = G.Chain(lowpass(256, fc), 8192, window)
        def timed(reset, reps=5):
            ts = []
            for _ in range(reps):
                if reset: ch.reset()
                torch.cuda.synchronize(); a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); ch.process_bulk(x, out); b_.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b_))
            return sorted(ts)[len(ts) // 2]
        t_in = timed(True); ch.last_power_ratio(); ch.process_bulk(x, out); moved = ch.last_power_ratio()[1]; t_set = timed(False)
        print(f"{window} fc={fc}: in-stream {n / t_in / 1e6:.1f} Gsamples/s   moved {moved}   settled {n / t_set / 1e6:.1f} Gsamples/s")
