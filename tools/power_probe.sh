#!/bin/bash
# developer tool: package power and shader clock while ONE kernel configuration runs back to back for ~6 s (rocm-smi sampled every 0.3 s, the four highest-power samples)
#   tools/power_probe.sh fir256 cfir256 fft8192 headline pair       (see the table in the python part)
cd $GRAFT_REPO_ROOT
for cfg in "$@"; do
python - $cfg <<'PY' &
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tools")
import numpy as np, torch
import gnuradio4_amd as G
from gnuradio4_amd import capi
cfg = sys.argv[1]
def lowpass(nt, fc=0.05):
    k = np.arange(nt); t = np.hamming(nt) * 2 * fc * np.sinc(2 * fc * (k - (nt - 1) / 2)); return (t / t.sum()).astype(np.float32)
n = 1 << 27
if cfg.startswith("fir"):
    x = G.synth_f32(2 * n, seed=42); y = torch.empty_like(x); f = G.fir_filter(lowpass(int(cfg[3:])), torch.float32); run = lambda: f.process_bulk(x, y); units = 2 * n
elif cfg.startswith("cfir"):
    x = G.synth_c32(n); y = torch.empty_like(x); f = G.fir_filter(lowpass(int(cfg[4:])), torch.complex64); f.set_algo(capi.FIR_TIME_DOMAIN); run = lambda: f.process_bulk(x, y); units = n
elif cfg.startswith("fft"):
    N = int(cfg[3:]); x = G.synth_c32(n); y = torch.empty((n // N, N), dtype=torch.float32, device="cuda"); f = G.FFT(N, "None"); run = lambda: f.mag2(x, y); units = n
elif cfg == "iir4":
    b, a = G.blocks.design_iir(capi.LOWPASS, 8, 0.05, float("nan"), 1.0, capi.BUTTERWORTH)
    x = G.synth_f32(2 * n, seed=42); y = torch.empty_like(x); f = G.iir_filter(b, a); run = lambda: f.process_bulk(x, y); units = 2 * n
elif cfg == "decim8":
    x = G.synth_f32(2 * n, seed=42); y = torch.empty(2 * n // 8, dtype=torch.float32, device="cuda"); f = G.fir_filter(lowpass(1024), torch.float32, decimate=8); run = lambda: f.process_bulk(x, y); units = 2 * n
elif cfg == "decim8off":  # (the guard off: for the timing-only builds of tools/ab_dh.sh, whose outputs are garbage)
    x = G.synth_f32(2 * n, seed=42); y = torch.empty(2 * n // 8, dtype=torch.float32, device="cuda"); f = G.fir_filter(lowpass(1024), torch.float32, decimate=8); f.set_guard_mode(capi.GUARD_OFF); run = lambda: f.process_bulk(x, y); units = 2 * n
elif cfg == "rotator":
    x = G.synth_c32(n); y = torch.empty_like(x); f = G.Rotator(phase_increment=0.37); run = lambda: f.process_bulk(x, out=y); units = n
elif cfg.startswith("chain"):  # chain<taps>_<fftSize>_<window>[_fd]: GR4HIP_CHAIN_AUTO (or the fused fast convolution with _fd)
    p = cfg[5:].split("_"); N = int(p[1])
    x = G.synth_c32(n); y = torch.empty((n // N, N), dtype=torch.float32, device="cuda"); f = G.Chain(lowpass(int(p[0])), N, p[2], capi.CHAIN_FUSED_FD if len(p) > 3 else capi.CHAIN_AUTO); run = lambda: f.process_bulk(x, y); units = n
else:
    algo = {"headline": capi.CHAIN_AUTO, "pair": capi.CHAIN_TIME_DOMAIN}[cfg]
    x = G.synth_c32(n); y = torch.empty((n // 8192, 8192), dtype=torch.float32, device="cuda"); f = G.Chain(lowpass(256), 8192, "None", algo); run = lambda: f.process_bulk(x, y); units = n
for _ in range(10): run()
torch.cuda.synchronize(); t0 = time.perf_counter(); it = 0
while time.perf_counter() - t0 < 6.0:
    for _ in range(20): run()
    torch.cuda.synchronize(); it += 20
print(f"{cfg}: {units * it / (time.perf_counter() - t0) / 1e9:.0f} Gsamples/s", flush=True)
PY
  pid=$!
  : > /tmp/smi.txt
  while kill -0 $pid 2>/dev/null; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power \(W\)|sclk" | tr -s ' ' | tr '\n' ';' >> /tmp/smi.txt; echo >> /tmp/smi.txt; sleep 0.3; done
  wait $pid
  grep -v "^$" /tmp/smi.txt | awk -F'Power \\(W\\): ' '{print $2+0, $0}' | sort -n | tail -4 | cut -d' ' -f2- | sed -e 's/GPU\[0\]\t*//g' | cut -c1-120
done
