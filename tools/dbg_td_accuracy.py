"""developer tool: the time-domain kernel pair under a rejected +50 dB interferer: bf16 three-term products against the f32 MFMA kernels (GR4HIP_FIR_NO_BF16X3) and the
float32 CPU direct form, all against float64"""
import sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from scipy.signal import lfilter
import gnuradio4_amd as G
from gnuradio4_amd import capi
def lowpass(nt, fc):
    k = np.arange(nt); t = np.hamming(nt) * 2 * fc * np.sinc(2 * fc * (k - (nt - 1) / 2)); return (t / t.sum()).astype(np.float32)
rng = np.random.default_rng(3)
N, frames = 8192, 64
n = frames * N
for nt in (64, 100, 256):
    for ampdb in (30, 50):
        taps = lowpass(nt, 0.05)
        x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
        x += (10 ** (ampdb / 20) * np.exp(2j * np.pi * 0.41 * np.arange(n))).astype(np.complex64)
        y = lfilter(taps.astype(np.float64), [1.0], x.astype(np.complex128)).reshape(frames, N)
        truth = np.abs(np.fft.fft(y, axis=1)) ** 2
        rms = np.sqrt(np.mean(truth ** 2, axis=1, keepdims=True))
        y32 = lfilter(taps, np.float32([1.0]), x).astype(np.complex64).reshape(frames, N)
        e32 = np.max(np.abs(np.abs(np.fft.fft(y32.astype(np.complex128), axis=1)) ** 2 - truth) / np.maximum(truth, rms))
        line = f"taps {nt} interferer +{ampdb} dB: float32 cpu {e32:.2e}"
        for sw, name in ((0, "bf16 three-term"), (1, "f32 MFMA")):
            capi.developer_switch("GR4HIP_FIR_NO_BF16X3", sw)
            got = G.Chain(taps, N, "None", capi.CHAIN_TIME_DOMAIN).process_bulk(torch.from_numpy(x).cuda()).cpu().numpy().reshape(frames, N)
            line += f" | {name} {np.max(np.abs(got - truth) / np.maximum(truth, rms)):.2e}"
        capi.developer_switch("GR4HIP_FIR_NO_BF16X3", 0)
        print(line, flush=True)
