#!/bin/bash
# developer tool: tools/ubench/pk_power.hip per mode with rocm-smi sampled while it runs (package power, shader clock)
cd $GRAFT_REPO_ROOT
hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize tools/ubench/pk_power.hip -o /tmp/pk_power 2>/dev/null || exit 1
for m in 2 3 5 4; do
  /tmp/pk_power $m 4 > /tmp/pk_$m.log &
  pid=$!
  sleep 2
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "sclk|Package Power" | sed 's/^.*: //' | tr '\n' ' '
  wait $pid
  cat /tmp/pk_$m.log
done
