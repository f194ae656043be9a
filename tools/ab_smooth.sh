#!/bin/bash
# developer tool: FFT rates at smooth sizes for alternative builds gnuradio4_amd/libgr4hip_<tag>.so (tools/build_variant.sh <tag> fft.hip -D...)
cd $GRAFT_REPO_ROOT
cp gnuradio4_amd/libgr4hip.so /tmp/orig.so
for tag in base "$@" base; do
  if [ $tag = base ]; then cp /tmp/orig.so gnuradio4_amd/libgr4hip.so; else cp gnuradio4_amd/libgr4hip_$tag.so gnuradio4_amd/libgr4hip.so; fi
  echo -n "$tag: "; python tools/fft_smooth_rates.py 2>&1 | tail -1
done
cp /tmp/orig.so gnuradio4_amd/libgr4hip.so
