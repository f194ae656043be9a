#!/bin/bash
# developer tool: configs[2] rates for alternative builds gnuradio4_amd/libgr4hip_<tag>.so of the frequency-domain decimator (swapped in place; timing-only builds give wrong results)
cd $GRAFT_REPO_ROOT
cp gnuradio4_amd/libgr4hip.so /tmp/orig.so
for tag in base "$@" base; do
  if [ $tag = base ]; then cp /tmp/orig.so gnuradio4_amd/libgr4hip.so; else cp gnuradio4_amd/libgr4hip_$tag.so gnuradio4_amd/libgr4hip.so; fi
  echo -n "$tag: "; python tools/c2_rate.py | head -1
done
cp /tmp/orig.so gnuradio4_amd/libgr4hip.so
