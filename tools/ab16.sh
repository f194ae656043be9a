#!/bin/bash
# developer tool: headline bench with the 16-wave kernel for each alternative build gnuradio4_amd/libgr4hip_<tag>.so (swapped in place); "old" = 8-wave kernel
cd $GRAFT_REPO_ROOT
cp gnuradio4_amd/libgr4hip.so /tmp/orig.so
run() { python bench.py --algo 3 --steps 8 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('Gs/s %.1f  launch_ms %.4f  err %s' % (d['value']/1e3, d['roofline']['avg_launch_ms'], d.get('verify',{}).get('max_rel_err')))"; }
echo -n "old: "; GR4HIP_CHAIN16=0 run
export GR4HIP_CHAIN16=1
for tag in base "$@" base; do
  if [ $tag = base ]; then cp /tmp/orig.so gnuradio4_amd/libgr4hip.so; else cp gnuradio4_amd/libgr4hip_$tag.so gnuradio4_amd/libgr4hip.so; fi
  echo -n "$tag: "; run
done
cp /tmp/orig.so gnuradio4_amd/libgr4hip.so
