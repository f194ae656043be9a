#!/bin/bash
cd $GRAFT_REPO_ROOT; export PYTHONPATH=$PWD
cp gnuradio4_amd/libgr4hip.so /tmp/orig.so
for tag in base ${TAGS:-NOMFMA NOFFT NOFIR} base; do
  if [ $tag = base ]; then cp /tmp/orig.so gnuradio4_amd/libgr4hip.so; else cp gnuradio4_amd/libgr4hip_t16_$tag.so gnuradio4_amd/libgr4hip.so; fi
  python tools/dbg/td16_rate.py $tag 2>&1 | tail -1
done
cp /tmp/orig.so gnuradio4_amd/libgr4hip.so
