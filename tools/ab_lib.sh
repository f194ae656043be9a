#!/bin/bash
# developer tool: A/B two builds of the library on the headline bench (swap the .so in place); extra args = env settings
cd $GRAFT_REPO_ROOT
for v in a b a b; do
  if [ $v = b ]; then cp gnuradio4_amd/libgr4hip.so /tmp/orig.so; cp gnuradio4_amd/libgr4hip_w2.so gnuradio4_amd/libgr4hip.so; fi
  echo -n "variant $v: "
  env "$@" python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('Gs/s %.1f  launch_ms %.4f' % (d['value']/1e3, d['roofline']['avg_launch_ms']))"
  if [ $v = b ]; then cp /tmp/orig.so gnuradio4_amd/libgr4hip.so; fi
done
