"""per-phase cycle breakdown of chain_td16_kernel (libgr4hip_timing.so built with -DGR4_FD_TIMING; s_memtime = 100 MHz reference ticks)"""
import ctypes as C, os, sys, shutil
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
shutil.copy(os.path.join(ROOT, "gnuradio4_amd/libgr4hip.so"), "/tmp/orig.so")
shutil.copy(os.path.join(ROOT, "gnuradio4_amd/libgr4hip_timing.so"), os.path.join(ROOT, "gnuradio4_amd/libgr4hip.so"))
try:
    import gnuradio4_amd as G
    L = C.CDLL(G.capi.LIB_PATH)
    N, frames = 8192, 8192
    x = G.synth_c32(frames * N)
    k = np.arange(256); t = np.hamming(256) * 0.01 * np.sinc(0.01 * (k - 127.5)); taps = (t / t.sum()).astype(np.float32)
    ch = G.Chain(taps, N, sys.argv[1] if len(sys.argv) > 1 else "None")
    ch.process_bulk(x); ch.last_power_ratio(); ch.process_bulk(x); print("moved:", ch.last_power_ratio())
    for _ in range(3): ch.process_bulk(x)
    torch.cuda.synchronize()
    buf = np.zeros(frames * 8 * 16, np.uint64)
    L.gr4hip_dbg_fd_timing.argtypes = [C.c_void_p, C.c_size_t]
    assert L.gr4hip_dbg_fd_timing(buf.ctypes.data, frames) == 0
    st = buf.reshape(frames, 8, 16).astype(np.int64)[512:-512]
    names = ["statistics + barrier", "split -> planes + barrier", "products (f16 matrix pipe) + window -> S", "power sum + prefetch + barrier", "pass A + barrier", "pass B (+ exchange)", "pass C + |.|^2 + stores"]
    tot = (st[:, :, 7].max(axis=1) - st[:, :, 0].min(axis=1)).mean()
    print(f"frame body: {tot:.0f} ticks of s_memtime (earliest wave start -> latest wave end)")
    for i in range(7):
        d = st[:, :, i + 1] - st[:, :, i]
        print(f"  {names[i]:44s} wave-min {d.min(axis=1).mean():7.0f}  mean {d.mean():7.0f}  wave-max {d.max(axis=1).mean():7.0f}   {100*d.mean()/tot:5.1f}%")
    gap = (st[1:, :, 0].min(axis=1) - st[:-1, :, 7].max(axis=1))
    print("  (stamps are per frame index; consecutive frames belong to different workgroups)")
finally:
    shutil.copy("/tmp/orig.so", os.path.join(ROOT, "gnuradio4_amd/libgr4hip.so"))
