"""configs[2] pair (decimate-by-8 1024-tap FIR -> 4 biquads): does the cascade of chunk c run beside the decimator of chunk c + 1?  (developer probe)
   python tools/pair_overlap_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import gnuradio4_amd as G
from gnuradio4_amd import capi

n = 1 << 27
x = G.synth_f32(n, seed=42)
k = np.arange(1024, dtype=np.float64)
t = np.hamming(1024) * 0.1 * np.sinc(0.1 * (k - 1023 / 2.0))
b = (t / t.sum()).astype(np.float32)
bi, ai = G.blocks.design_iir(capi.LOWPASS, 8, 0.05, float("nan"), 1.0, capi.BUTTERWORTH)
yd = torch.empty(n // 8, dtype=torch.float32, device="cuda")
yo = torch.empty_like(yd)


def rate(fn, reps=10, rounds=5):
    fn(); torch.cuda.synchronize()
    ms = []
    for _ in range(rounds):
        a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a_.record()
        for _ in range(reps):
            fn()
        b_.record(); b_.synchronize()
        ms.append(a_.elapsed_time(b_) / reps)
    return sorted(ms)[len(ms) // 2]


fir, iir = G.fir_filter(b, torch.float32, decimate=8), G.iir_filter(bi, ai)
def plain():
    fir.process_bulk(x, yd); iir.process_bulk(yd, yo)
ms = rate(plain)
print(f"two launches back to back: {n / ms / 1e6:.1f} G input samples/s")
ms = rate(lambda: fir.process_bulk(x, yd)); print(f"  decimator alone {n / ms / 1e6:.1f}")
ms = rate(lambda: iir.process_bulk(yd, yo)); print(f"  cascade alone {n / ms / 1e6:.1f} (input-rate equivalent)")
ref = None
plain(); torch.cuda.synchronize(); ref = yo.clone()
s2 = torch.cuda.Stream()
for lg in (22, 23, 24, 25, 26):
    c = 1 << lg
    fir2, iir2 = G.fir_filter(b, torch.float32, decimate=8), G.iir_filter(bi, ai)
    evs = [torch.cuda.Event() for _ in range(n // c)]
    def chunks():
        s1 = torch.cuda.current_stream()
        s2.wait_stream(s1)
        for i in range(n // c):
            fir2.process_bulk(x[i * c:(i + 1) * c], yd[i * c // 8:(i + 1) * c // 8])
            evs[i].record(s1)
            with torch.cuda.stream(s2):
                s2.wait_event(evs[i])
                iir2.process_bulk(yd[i * c // 8:(i + 1) * c // 8], yo[i * c // 8:(i + 1) * c // 8])
        s1.wait_stream(s2)
    ms = rate(chunks)
    fir2.reset() if hasattr(fir2, "reset") else None
    print(f"chunks of 2^{lg} on two streams: {n / ms / 1e6:.1f}")
    def serial():
        for i in range(n // c):
            fir2.process_bulk(x[i * c:(i + 1) * c], yd[i * c // 8:(i + 1) * c // 8])
            iir2.process_bulk(yd[i * c // 8:(i + 1) * c // 8], yo[i * c // 8:(i + 1) * c // 8])
    ms = rate(serial)
    print(f"chunks of 2^{lg} on one stream : {n / ms / 1e6:.1f}")
