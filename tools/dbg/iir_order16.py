import sys
sys.path.insert(0, "tests")
import numpy as np, torch
from scipy.signal import butter, cheby1
import gnuradio4_amd as G
import oracle_lib as O
rng = np.random.default_rng(1)
def rel(a, t):
    rms = np.sqrt(np.mean(np.abs(t) ** 2)); return float(np.max(np.abs(a - t) / np.maximum(np.abs(t), rms)))
for order, fc in ((16, 0.017), (16, 0.03), (12, 0.017), (10, 0.017), (16, 0.1)):
    for seed in range(3):
        n = 878215
        x = np.random.default_rng(seed).standard_normal(n).astype(np.float32)
        sos = cheby1(order, 1.0, 2 * fc, output="sos")
        f = G.iir_filter(sos[:, :3], sos[:, 3:])
        y = f.process_bulk(torch.from_numpy(x).cuda()).cpu().numpy()
        secs = O.make_sections([(bb, aa) for bb, aa in zip(sos[:, :3].astype(np.float32), sos[:, 3:].astype(np.float32))])
        truth = O.iir_cascade(secs, x, 3, f64=True)
        r1, r2 = rel(O.iir_cascade(secs, x, O.DF_I, f64=False), truth), rel(O.iir_cascade(secs, x, O.DF_II, f64=False), truth)
        algo = f.algo if hasattr(f, "algo") else None
        print(f"order {order} fc {fc} seed {seed}: device {rel(y, truth):.3g} (algo {algo}); reference float32 cascade DF_I {r1:.3g}, DF_II {r2:.3g}", flush=True)
