#!/bin/bash
# developer tool: headline bench under a list of "VAR=value" environment settings, e.g. tools/ab_env.sh GR4HIP_FD_TOUCH=0 GR4HIP_FD_TOUCH=7
cd $GRAFT_REPO_ROOT
for kv in "$@"; do
  echo -n "$kv: "
  env $kv python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('Gs/s %.1f  launch_ms %.4f' % (d['value']/1e3, d['roofline']['avg_launch_ms']))"
done
