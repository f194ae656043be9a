#!/bin/bash
# developer tool: headline bench for each alternative build gnuradio4_amd/libgr4hip_<tag>.so given as argument (swapped in place)
cd $GRAFT_REPO_ROOT
cp gnuradio4_amd/libgr4hip.so /tmp/orig.so
for tag in base "$@" base; do
  if [ $tag = base ]; then cp /tmp/orig.so gnuradio4_amd/libgr4hip.so; else cp gnuradio4_amd/libgr4hip_$tag.so gnuradio4_amd/libgr4hip.so; fi
  echo -n "$tag: "
  python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('Gs/s %.1f  launch_ms %.4f' % (d['value']/1e3, d['roofline']['avg_launch_ms']))"
done
cp /tmp/orig.so gnuradio4_amd/libgr4hip.so
