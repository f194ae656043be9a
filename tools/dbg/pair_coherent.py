"""developer tool: the kernel pair CHAIN_AUTO serves fft sizes > 8192 with (f16 FIR under its two verdicts + the FFT kernel) under a tone the filter removes by 40 .. 60 dB whose
residue reaches the output spectrum's rms: the 22-bit products' coherent error in the one bin where the metric looks"""
import sys
sys.path.insert(0, "tests")
import numpy as np, torch
from scipy.signal import lfilter
import gnuradio4_amd as G
import oracle_lib as O
from gnuradio4_amd import capi
rng = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 5)
def lowpass(nt, fc):
    k = np.arange(nt); t = np.hamming(nt) * 2 * fc * np.sinc(2 * fc * (k - (nt - 1) / 2)); return (t / t.sum()).astype(np.float32)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
algo = int(sys.argv[2]) if len(sys.argv) > 2 else capi.CHAIN_AUTO
frames = 12; n = frames * N
worst = 0.0; bad = 0; cases = 0
for nt, fc in ((17, 0.2), (17, 0.1), (33, 0.1), (65, 0.05)):
    taps = lowpass(nt, fc)
    H = np.abs(np.fft.fft(taps.astype(np.float64), 65536))
    for trial in range(int(sys.argv[4]) if len(sys.argv) > 4 else 40):
        f0 = float(rng.uniform(fc + 1.5 / nt, 0.49)); h = H[int(round(f0 * 65536))]
        amp = float(10 ** rng.uniform(-0.5, 1.0)) * np.sqrt(2.0 * 0.9 * np.sqrt(2 * fc) / N) / max(h, 1e-9)  # residue at 0.3 .. 10 x the rms of the output spectrum
        x = (rng.standard_normal(n) + 1j * rng.standard_normal(n) + amp * np.exp(2j * np.pi * (f0 * np.arange(n) + rng.random()))).astype(np.complex64)
        y = lfilter(taps.astype(np.float64), [1.0], x.astype(np.complex128)).reshape(frames, N)
        T = np.abs(np.fft.fft(y, axis=1)) ** 2
        rms = np.sqrt(np.mean(T ** 2, axis=1, keepdims=True))
        ch = G.Chain(taps, N, "None", algo)
        got = ch.process_bulk(torch.from_numpy(x).cuda()).cpu().numpy().astype(np.float64).reshape(frames, N)
        e = float(np.max((np.abs(got - T) / np.maximum(T, rms))[1:]))
        y32 = O.fir(taps, x, acc64=False)[0].reshape(frames, N)
        t32 = np.abs(np.fft.fft(y32.astype(np.complex128), axis=1)) ** 2
        e32 = float(np.max((np.abs(t32 - T) / np.maximum(T, rms))[1:]))
        cases += 1; worst = max(worst, e)
        if e > max(1e-5, e32):
            bad += 1
            if bad <= 10: print(f"FAIL taps {nt} fc {fc} tone {amp:.3g} at {f0:.4f} (|H| {20 * np.log10(h):.1f} dB): err {e:.3g}, reference float32 {e32:.3g}, algo {ch.algo}", flush=True)
print(f"N={N} algo {algo}: {cases} cases, {bad} above the bar, worst {worst:.3g}")
