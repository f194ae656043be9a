"""developer tool: where a block of the single-pass IIR kernel spends its time.  Build csrc/iir.hip with -DGR4_IIR_TIMING (the host then dumps the s_memrealtime stamps of the last launch to /tmp/iir_stamps.txt), run this on the GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import gnuradio4_amd as G
from gnuradio4_amd import capi
n = 1 << 26
x = G.synth_f32(n, seed=3)
b, a = G.blocks.design_iir(capi.LOWPASS, 8, 0.05, float("nan"), 1.0, capi.BUTTERWORTH)
f = G.iir_filter(b, a); y = torch.empty_like(x)
for _ in range(3): f.process_bulk(x, y)
torch.cuda.synchronize()
S = np.loadtxt("/tmp/iir_stamps.txt", dtype=np.float64) / 100.0  # s_memrealtime: 100 MHz -> us
t0 = S[:, 0].min()
print("blocks", len(S), "span us", S[:, 7].max() - t0)
names = ["stage", "run1", "scan", "lookback", "phase5", "run2", "store"]
for k in range(7):
    d = S[:, k + 1] - S[:, k]
    print("%-9s mean %.2f  p10 %.2f  p50 %.2f  p90 %.2f  max %.2f us" % (names[k], d.mean(), *np.percentile(d, [10, 50, 90]), d.max()))
d = S[:, 7] - S[:, 0]
print("residence mean %.2f p50 %.2f" % (d.mean(), np.median(d)))
# start order vs ticket
print("start time monotone fraction", np.mean(np.diff(S[:, 0]) >= 0))
zt = S[:, 3]  # Z published
# how long after own Z is the slowest of the 64 predecessors' Z
w = [max(zt[max(0, i - 64):i].max() - zt[i], 0) for i in range(1, len(S))]
print("wait for slowest of 64 predecessors' Z: mean %.2f p50 %.2f p90 %.2f" % (np.mean(w), np.median(w), np.percentile(w, 90)))
print("Z-publish time (us from t0) by ticket, stride 64:")
print(np.round(S[2048:2048 + 1600:64, 3] - t0, 1))
print("start time by ticket, stride 64:")
print(np.round(S[2048:2048 + 1600:64, 0] - t0, 1))
lb = S[:, 4] - S[:, 3]
print("lookback by ticket stride 64:", np.round(lb[2048:2048 + 1600:64], 1))
print("T known (us) stride 64:", np.round(S[2048:2048 + 1600:64, 4] - t0, 1))
m = S[:, 10] > 0
print("first poll round returns after Z: mean %.2f p50 %.2f" % ((S[m, 8] - S[m, 3]).mean(), np.median(S[m, 8] - S[m, 3])))
print("window 1 complete after Z: mean %.2f p50 %.2f" % ((S[m, 9] - S[m, 3]).mean(), np.median(S[m, 9] - S[m, 3])))
print("windows: mean %.2f  hist" % (S[m, 10] * 100).mean(), np.bincount((S[m, 10] * 100).round().astype(int))[:8])
print("poll rounds: mean %.2f  hist" % (S[m, 11] * 100).mean(), np.bincount((S[m, 11] * 100).round().astype(int))[:12])
