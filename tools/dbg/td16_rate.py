import sys
import numpy as np, torch
import gnuradio4_amd as G
n = 1 << 27
x = G.synth_c32(n, seed=42)
out = torch.empty((n // 8192, 8192), dtype=torch.float32, device="cuda")
def lowpass(nt, fc):
    k = np.arange(nt); t = np.hamming(nt) * 2 * fc * np.sinc(2 * fc * (k - (nt - 1) / 2)); return (t / t.sum()).astype(np.float32)
res = []
for window in ("None", "Hann"):
    ch = G.Chain(lowpass(256, 0.005), 8192, window)
    def timed(reset, reps=5):
        ts = []
        for _ in range(reps):
            if reset: ch.reset()
            torch.cuda.synchronize(); a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); ch.process_bulk(x, out); b_.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b_))
        return sorted(ts)[len(ts) // 2]
    t_in = timed(True); ch.last_power_ratio(); ch.process_bulk(x, out); ch.last_power_ratio(); t_set = timed(False)
    res.append(f"{window}: in-stream {n / t_in / 1e6:.1f}  settled {n / t_set / 1e6:.1f}")
print((sys.argv[1] if len(sys.argv) > 1 else "") + "  " + "   ".join(res))
