#!/bin/bash
# developer tool: headline bench + secondary configs for each alternative build gnuradio4_amd/libgr4hip_<tag>.so (swapped in place)
cd $GRAFT_REPO_ROOT
cp gnuradio4_amd/libgr4hip.so /tmp/orig.so
for tag in base "$@"; do
  if [ $tag = base ]; then cp /tmp/orig.so gnuradio4_amd/libgr4hip.so; else cp gnuradio4_amd/libgr4hip_$tag.so gnuradio4_amd/libgr4hip.so; fi
  echo "==== $tag"
  python bench.py --steps 5 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline Gs/s %.1f  launch_ms %.4f' % (d['value']/1e3, d['roofline']['avg_launch_ms']))"
  python tools/bench_configs.py 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
for k,v in d.items():
    r=[x for x in v if 'samples/s' in x]
    print('  %-90s %s' % (k[:90], ' '.join('%s=%s'%(x,v[x]) for x in r)))
"
done
cp /tmp/orig.so gnuradio4_amd/libgr4hip.so
