#!/bin/bash
# developer tool: run a command with the default library and with gnuradio4_amd/libgr4hip_<tag>.so in its place, alternating:  tools/ab_generic.sh <tag> <reps> <command...>
TAG=$1; REPS=$2; shift 2
cp gnuradio4_amd/libgr4hip.so /tmp/orig.so
for rep in $(seq 1 $REPS); do
for tag in base $TAG; do
  if [ $tag = base ]; then cp /tmp/orig.so gnuradio4_amd/libgr4hip.so; else cp gnuradio4_amd/libgr4hip_$tag.so gnuradio4_amd/libgr4hip.so; fi
  echo "== $tag"; timeout 300 "$@" 2>/dev/null | grep "^{"
done
done
cp /tmp/orig.so gnuradio4_amd/libgr4hip.so
