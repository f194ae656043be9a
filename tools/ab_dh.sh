#!/bin/bash
# developer tool: tools/dh_timing.py for each timing-only build gnuradio4_amd/libgr4hip_dh_<tag>.so (tools/build_variant.sh dh_<tag> fir_decim_f16.hip -DGR4_T_DH_<tag>), swapped in place
cd $GRAFT_REPO_ROOT
cp gnuradio4_amd/libgr4hip.so /tmp/orig.so
for tag in base "$@" base; do
  if [ $tag = base ]; then cp /tmp/orig.so gnuradio4_amd/libgr4hip.so; python tools/dh_timing.py full 2>&1 | tail -1 | sed "s/^/$tag: /"
  else cp gnuradio4_amd/libgr4hip_dh_$tag.so gnuradio4_amd/libgr4hip.so; python tools/dh_timing.py 2>&1 | tail -1 | sed "s/^/$tag: /"; fi
done
cp /tmp/orig.so gnuradio4_amd/libgr4hip.so
