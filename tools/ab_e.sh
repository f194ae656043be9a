#!/bin/bash
# developer tool: headline bench with the correction e on the bf16 matrix pipe (default build) and on the f32 MFMA (libgr4hip_ef32.so), alternating
cp gnuradio4_amd/libgr4hip.so /tmp/orig.so
for rep in 1 2; do
for tag in base ef32; do
  if [ $tag = base ]; then cp /tmp/orig.so gnuradio4_amd/libgr4hip.so; else cp gnuradio4_amd/libgr4hip_$tag.so gnuradio4_amd/libgr4hip.so; fi
  python bench.py --no-graph8 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$tag', d['value'], d['roofline']['frac'], d['verify']['max_rel_err'])"
done
done
cp /tmp/orig.so gnuradio4_amd/libgr4hip.so
