#!/bin/bash
# developer tool (round 5): hooked decimators and the channeliser with the default build and with libgr4hip_<tag>.so builds (swapped in place), alternating
cp gnuradio4_amd/libgr4hip.so /tmp/orig.so
for tag in base "$@" base; do
  if [ $tag = base ]; then cp /tmp/orig.so gnuradio4_amd/libgr4hip.so; else cp gnuradio4_amd/libgr4hip_$tag.so gnuradio4_amd/libgr4hip.so; fi
  echo "== $tag"
  python tools/decim_d4_rates.py 2>&1 | grep "rotator as"
  python tools/bench_fusion.py 2>&1 | grep -i "channeli\|rotator -> dec" | head -3
done
cp /tmp/orig.so gnuradio4_amd/libgr4hip.so
