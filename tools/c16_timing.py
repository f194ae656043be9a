#!/usr/bin/env python
"""developer tool: per-phase cycle breakdown of the 16-wave chain kernel (needs libgr4hip_c16t.so built with -DGR4_C16_TIMING:
tools/build_variant.sh c16t chain16.hip -DGR4_C16_TIMING)"""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["GR4HIP_CHAIN16"] = "1"
import gnuradio4_amd.capi as capi
capi.LIB_PATH = os.path.join(ROOT, "gnuradio4_amd", "libgr4hip_c16t.so")
import gnuradio4_amd as G
L = capi.lib()
N, frames = 8192, 8192
x = G.synth_c32(frames * N)
taps = np.ones(256, np.float32) / 256
ch = G.Chain(taps, N, "None", 3)
for _ in range(3):
    ch.process_bulk(x)
torch.cuda.synchronize()
n_cu = torch.cuda.get_device_properties(0).multi_processor_count
iters = frames // n_cu
buf = np.zeros(frames * 16 * 16, np.uint64)
fn = C.CDLL(capi.LIB_PATH).gr4hip_dbg_c16_timing
fn.argtypes = [C.c_void_p, C.c_size_t]
assert fn(buf.ctypes.data, frames) == 0
st = buf.reshape(iters, n_cu, 16, 16).astype(np.int64)[4:-2]  # [iteration][workgroup][wave][stamp], steady state
names = ["top barrier (frame landed)", "pick-up + cross reads", "stores, W16 twiddles, fft8, twX", "split wait A", "W writes", "barrier #1", "DMA + MFMA block (w0-7)",
         "s1: read, fft8, tw1", "private tail X", "H, e-sum (w8-15), DMA", "split wait C", "E s1 (pruned) + tw1", "private tail E", "combine + |Y|^2 writes"]
t0 = st[:, :, :, 0].min(axis=2)[:, :, None, None]
rel = st - t0
tot = (st[:, :, :, 13].max(axis=2) - st[:, :, :, 0].min(axis=2)).mean()
print(f"frame loop body: {tot:.0f} cycles (earliest wave start -> latest wave end)")
print("stamp: mean arrival per wave (cycles since frame start), waves 0..15")
for i in range(14):
    print(f"  s{i:2d} " + " ".join(f"{rel[:, :, w, i].mean():6.0f}" for w in range(16)))
print("phase durations: wave-min / mean / wave-max, share of the frame")
for i in range(13):
    d = st[:, :, :, i + 1] - st[:, :, :, i]
    print(f"  {names[i]:34s} {d.min(axis=2).mean():7.0f} {d.mean():7.0f} {d.max(axis=2).mean():7.0f}   {100*d.mean()/tot:5.1f}%")
