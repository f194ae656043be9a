#!/bin/bash
# developer tool: package power and shader clock while one FIR configuration runs back to back for a few seconds
#   tools/fir_power_probe.sh [ntaps ...]          (float fir_filter, 2^28 samples per call)
cd $GRAFT_REPO_ROOT
for nt in "$@"; do
python - $nt <<'PY' &
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import gnuradio4_amd as G
nt = int(sys.argv[1]); n = 1 << 28
x = G.synth_f32(n, seed=42); y = torch.empty_like(x)
k = np.arange(nt); t = np.hamming(nt) * 0.1 * np.sinc(0.1 * (k - (nt - 1) / 2)); t = (t / t.sum()).astype(np.float32)
f = G.fir_filter(t, torch.float32)
for _ in range(20): f.process_bulk(x, y)
torch.cuda.synchronize(); t0 = time.perf_counter(); it = 0
while time.perf_counter() - t0 < 6.0:
    for _ in range(50): f.process_bulk(x, y)
    torch.cuda.synchronize(); it += 50
dt = time.perf_counter() - t0
print(f"taps {nt}: {n * it / dt / 1e9:.0f} Gsamples/s", flush=True)
PY
  pid=$!
  : > /tmp/smi.txt
  while kill -0 $pid 2>/dev/null; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power \(W\)|sclk" | tr -s ' ' | tr '\n' ';' >> /tmp/smi.txt; echo >> /tmp/smi.txt; sleep 0.3; done
  wait $pid
  sort -t: -k6 -n /tmp/smi.txt | grep -v "^$" | awk -F'Power \\(W\\): ' '{print $2+0, $0}' | sort -n | tail -4 | cut -d' ' -f2- | cut -c1-150
done
