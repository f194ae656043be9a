#!/usr/bin/env python
"""developer tool: per-phase cycle breakdown of the fused chain kernel (needs libgr4hip_timing.so built with -DGR4_FD_TIMING)"""
import ctypes as C, os, sys, shutil
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
shutil.copy(os.path.join(ROOT, "gnuradio4_amd/libgr4hip.so"), "/tmp/orig.so")
shutil.copy(os.path.join(ROOT, "gnuradio4_amd/libgr4hip_timing.so"), os.path.join(ROOT, "gnuradio4_amd/libgr4hip.so"))
try:
    import gnuradio4_amd as G
    L = C.CDLL(G.capi.LIB_PATH)
    N, frames = 8192, 8192
    x = G.synth_c32(frames * N)
    taps = np.ones(256, np.float32) / 256
    ch = G.Chain(taps, N, "None", 3)
    for _ in range(3):
        ch.process_bulk(x)
    torch.cuda.synchronize()
    buf = np.zeros(frames * 8 * 16, np.uint64)
    L.gr4hip_dbg_fd_timing.argtypes = [C.c_void_p, C.c_size_t]
    assert L.gr4hip_dbg_fd_timing(buf.ctypes.data, frames) == 0
    st = buf.reshape(frames, 8, 16).astype(np.int64)
    st = st[512:-512]  # steady state
    names = ["top barrier (DMA landed)", "DMA issue + pass A", "barrier #1", "e-FIR + pass-B gathers", "barrier #2", "e-sum + X pass B", "barrier #3",
             "X pass C + H", "barrier #4", "E pass B", "barrier #5", "E pass C", "combine + stores", "-"]
    t0 = st[:, :, 0].min(axis=1)[:, None, None]  # frame start = earliest wave
    rel = st - t0
    tot = (st[:, :, 14].max(axis=1) - st[:, :, 0].min(axis=1)).mean()
    print(f"frame loop body: {tot:.0f} cycles (earliest wave start -> latest wave end)")
    print("stamp: mean arrival per wave (cycles since frame start)")
    for i in range(15):
        print(f"  s{i:2d} " + " ".join(f"{rel[:, w, i].mean():7.0f}" for w in range(8)))
    print("phase durations, mean over frames: wave-min / wave-mean / wave-max")
    for i in range(14):
        d = st[:, :, i + 1] - st[:, :, i]
        print(f"  {names[i]:28s} {d.min(axis=1).mean():7.0f} {d.mean():7.0f} {d.max(axis=1).mean():7.0f}   {100*d.mean()/tot:5.1f}%")
finally:
    shutil.copy("/tmp/orig.so", os.path.join(ROOT, "gnuradio4_amd/libgr4hip.so"))
