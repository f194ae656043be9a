#!/bin/bash
# developer tool: tools/chain_td_rates.py for each alternative build gnuradio4_amd/libgr4hip_<tag>.so (swapped in place on the box)
cp gnuradio4_amd/libgr4hip.so /tmp/orig.so
for tag in base "$@"; do
  if [ $tag = base ]; then cp /tmp/orig.so gnuradio4_amd/libgr4hip.so; else cp gnuradio4_amd/libgr4hip_$tag.so gnuradio4_amd/libgr4hip.so; fi
  echo "== $tag"; python tools/chain_td_rates.py 2>&1 | cut -c1-150
done
cp /tmp/orig.so gnuradio4_amd/libgr4hip.so
