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
    buf = np.zeros(frames * 16, np.uint64)
    L.gr4hip_dbg_fd_timing.argtypes = [C.c_void_p, C.c_size_t]
    assert L.gr4hip_dbg_fd_timing(buf.ctypes.data, frames) == 0
    st = buf.reshape(frames, 16).astype(np.int64)
    st = st[512:-512]  # steady state
    names = ["T wait(DMA)+barrier", "passA", "barrier#1", "eFIR+passB load", "barrier#2", "passB compute+store", "barrier#3", "passC X + H", "barrier#4",
             "E passB", "barrier#5", "E passC load+tw", "barrier#6", "dma issue", "fft16+combine+store"]
    tot = (st[:, 14] - st[:, 0]).mean()
    print(f"frame loop body (wave 0): {tot:.0f} cycles")
    for i in range(14):
        d = (st[:, i + 1] - st[:, i]).mean()
        print(f"  {names[i]:28s} {d:8.0f}  {100*d/tot:5.1f}%")
finally:
    shutil.copy("/tmp/orig.so", os.path.join(ROOT, "gnuradio4_amd/libgr4hip.so"))
