#!/bin/bash
# developer tool (profiles/r06_decim_floor.txt): rates of the decimator's timing-only builds (tools/ab_dh.sh) and package power / clock of the ones that bound the two levers
cd $GRAFT_REPO_ROOT
bash tools/ab_dh.sh MFMA32 NOMFMA NOSTATS NOSPLIT MFMA32_NOSTATS
cp gnuradio4_amd/libgr4hip.so /tmp/orig2.so
for tag in base MFMA32 MFMA32_NOSTATS NOSTATS base; do
  if [ $tag = base ]; then cp /tmp/orig2.so gnuradio4_amd/libgr4hip.so; else cp gnuradio4_amd/libgr4hip_dh_$tag.so gnuradio4_amd/libgr4hip.so; fi
  echo "== power probe: $tag"; bash tools/power_probe.sh decim8off
done
cp /tmp/orig2.so gnuradio4_amd/libgr4hip.so
