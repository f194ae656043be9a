#!/bin/bash
# developer tool: the 16-wave chain kernel (GR4HIP_CHAIN16=1) against the parity suite and the 8-wave kernel
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c16
export GR4HIP_CHAIN16=1
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "chain or fft_mag2 or fft_linearity or many_frames" > gpurun_out/c16/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/c16/pytest.log
tail -15 gpurun_out/c16/pytest.log
for v in 0 1; do
  echo "== GR4HIP_CHAIN16=$v"
  GR4HIP_CHAIN16=$v timeout 300 python bench.py --steps 10 --warmup 5 --no-cpu-baseline 2>&1 | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('Gs/s %.1f launch_ms %.4f frac %.4f verify %s' % (d['value']/1e3, d['roofline']['avg_launch_ms'], d['roofline']['frac'], d.get('verify')))
    else: print(l.rstrip())
"
done
