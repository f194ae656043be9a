#!/bin/bash
# developer tool (round 4, VERDICT #3 i): headline bench with scalar complex multiplies (default build) and with hand-packed ones (libgr4hip_cmulpk.so:
# tools/build_variant.sh cmulpk chain_fused.hip -DGR4_CMUL_PK=1), alternating on one box
cp gnuradio4_amd/libgr4hip.so /tmp/orig.so
for rep in 1 2 3; do
for tag in base cmulpk; do
  if [ $tag = base ]; then cp /tmp/orig.so gnuradio4_amd/libgr4hip.so; else cp gnuradio4_amd/libgr4hip_$tag.so gnuradio4_amd/libgr4hip.so; fi
  timeout 300 python bench.py --no-graph8 --no-cpu-baseline --no-hann-row 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$tag', d['value'], d['value_at_median_step'], d['roofline']['frac'], d['verify']['max_rel_err'])"
done
done
cp /tmp/orig.so gnuradio4_amd/libgr4hip.so
