#!/bin/bash
# developer tool: time the fused FD chain kernel with parts switched off (GR4HIP_FD_ABLATE bit mask)
for ab in 0 16 17 18 19 20 23 24; do
  echo -n "ablate=$ab  "
  GR4HIP_FD_ABLATE=$ab python bench.py --steps 3 --warmup 1 --log2-samples 28 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('Gs/s %.1f  launch_ms %.4f' % (d['value']/1e3, d['roofline']['avg_launch_ms']))"
done
