#!/bin/bash
# developer tool (round 5): the bounds of the three headline experiments VERDICT r04 asks for, as timing-only builds of chain_fused.hip (their results are wrong on purpose):
#   nobar = in-frame barriers without s_barrier (bound of hiding every barrier wait: two phase-offset workgroups per CU)   nob5 = E's exchange without its barrier (a one-exchange E)
#   noe = no E transform at all   noeb = E's pass B without arithmetic   guardoff = the default build with the guard's power sums off (bound of moving them to the matrix pipe)
cd $GRAFT_REPO_ROOT
cp gnuradio4_amd/libgr4hip.so /tmp/orig.so
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-graph8 --no-hann-row --no-secondary --no-live-traffic --no-verify"
one() { echo -n "$1: "; $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('Gs/s %.1f  launch_ms %.4f  frac %.4f' % (d['value']/1e3, d['roofline']['avg_launch_ms'], d['roofline']['frac']))"; }
for tag in ${TAGS:-base nobar nob5 base noe noeb base}; do
  if [ $tag = base ]; then cp /tmp/orig.so gnuradio4_amd/libgr4hip.so; else cp gnuradio4_amd/libgr4hip_$tag.so gnuradio4_amd/libgr4hip.so; fi
  one $tag "$B"
done
cp /tmp/orig.so gnuradio4_amd/libgr4hip.so
one guardoff "$B --guard-mode 2"
one base "$B"
