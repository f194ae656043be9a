#!/bin/bash
# developer tool: one python script against alternative builds gnuradio4_amd/libgr4hip_<tag>.so (tools/build_variant.sh <tag> <file>.hip -D...)
#   tools/ab_run.sh tools/fir_ab.py tagA tagB      (runs base, tagA, tagB, base)
cd $GRAFT_REPO_ROOT
S=$1; shift
cp gnuradio4_amd/libgr4hip.so /tmp/orig.so
for tag in base "$@" base; do
  if [ $tag = base ]; then cp /tmp/orig.so gnuradio4_amd/libgr4hip.so; else cp gnuradio4_amd/libgr4hip_$tag.so gnuradio4_amd/libgr4hip.so; fi
  echo "== $tag"; python $S 2>&1 | tail -4
done
cp /tmp/orig.so gnuradio4_amd/libgr4hip.so
