#!/bin/bash
# developer tool: build/host/test_host_device N times on generated inputs; prints every FAILED line (flakiness hunt)
cd $GRAFT_REPO_ROOT
python - <<'PY'
import sys, numpy as np
sys.path.insert(0, "tests")
import oracle_lib as O
N, frames, ntaps = 1024, 5, 64
x = O.signal_c32(3, N * frames); b = O.design_taps_hamming_lowpass(ntaps, 0.1)
x.tofile("/tmp/hd_in.bin"); b.tofile("/tmp/hd_taps.bin")
PY
fails=0
for i in $(seq 1 ${1:-30}); do
  ./build/host/test_host_device /tmp/hd_in.bin /tmp/hd_taps.bin 1024 /tmp/hd_o > /tmp/hd_out.txt 2>&1 || { fails=$((fails+1)); grep "FAILED\|first differing" /tmp/hd_out.txt | head -4; }
done
echo "runs ${1:-30} failures $fails"
