#!/bin/bash
# developer tool: Hann-windowed chain (8192, and the small-frame variant at 1024) with e on the bf16 pipe (default build) and on the f32 MFMA (libgr4hip_ewf32.so), alternating
cp gnuradio4_amd/libgr4hip.so /tmp/orig.so
for rep in 1 2; do
for tag in base ewf32; do
  if [ $tag = base ]; then cp /tmp/orig.so gnuradio4_amd/libgr4hip.so; else cp gnuradio4_amd/libgr4hip_$tag.so gnuradio4_amd/libgr4hip.so; fi
  echo "== $tag"; timeout 300 python tools/chain_modes_rates.py 2>&1 | grep -i "hann -> mag2"
  timeout 300 python - <<'P' 2>&1 | tail -1
import sys, os; sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch, numpy as np, gnuradio4_amd as G
from _timing import steady
n = 1 << 27; x = G.synth_c32(n, seed=5)
k = np.arange(100); t = np.hamming(100) * 0.6 * np.sinc(0.6 * (k - 49.5)); t = (t / t.sum()).astype(np.float32)
ch = G.Chain(t, 1024, "Hann"); m2 = torch.empty((n // 1024, 1024), dtype=torch.float32, device="cuda")
print("chain 100 taps -> 1024 Hann -> mag2 (%s): %.1f Msamples/s" % (ch.algo_name() if hasattr(ch, "algo_name") else "", n / steady(lambda: ch.process_bulk(x, m2)) / 1e6))
P
done
done
cp /tmp/orig.so gnuradio4_amd/libgr4hip.so
