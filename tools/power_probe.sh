#!/bin/bash
# developer tool: package power and shader clock while the headline bench runs (DVFS: is the kernel power-limited?)
cd $GRAFT_REPO_ROOT
for c in 0 1; do
  echo "== GR4HIP_CHAIN16=$c"
  GR4HIP_CHAIN16=$c python bench.py --steps 1500 --warmup 10 --no-cpu-baseline --no-verify > /tmp/b_$c.json 2>/dev/null &
  pid=$!
  sleep 3.5
  for i in 1 2 3 4; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power \(W\)|sclk|mclk|fclk" | tr -s ' ' | tr '\n' ';'; echo; sleep 0.4; done
  wait $pid
  python -c "import json; d=json.loads(open('/tmp/b_$c.json').read().strip().splitlines()[-1]); print('Gs/s %.1f launch_ms %.4f' % (d['value']/1e3, d['roofline']['avg_launch_ms']))"
done
rocm-smi --showmaxpower 2>/dev/null | grep -i "max"
