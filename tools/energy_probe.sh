#!/bin/bash
# developer tool: package power, shader clock and energy per sample of the headline kernel and of its timing-only variants
#   tools/energy_probe.sh [tag ...]      tags = gnuradio4_amd/libgr4hip_<tag>.so built by tools/build_variant.sh (wrong results, timing only)
# base and zero (= base on all-zero input: same instructions and traffic, datapaths not toggling) always run.
cd $GRAFT_REPO_ROOT
cp gnuradio4_amd/libgr4hip.so /tmp/orig.so
rocm-smi --showmaxpower 2>/dev/null | grep -i "max"
for tag in base zero "$@" base; do
  if [ $tag = base ] || [ $tag = zero ]; then cp /tmp/orig.so gnuradio4_amd/libgr4hip.so; else cp gnuradio4_amd/libgr4hip_$tag.so gnuradio4_amd/libgr4hip.so; fi
  Z=0; [ $tag = zero ] && Z=1
  GR4HIP_BENCH_ZERO_INPUT=$Z python bench.py --steps 1500 --warmup 10 --no-cpu-baseline --no-verify --no-graph8 > /tmp/b_$tag.json 2>/dev/null &
  pid=$!
  sleep 3.6
  : > /tmp/smi_$tag.txt
  for i in 1 2 3 4; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power \(W\)|sclk" | tr -s ' ' | tr '\n' ';' >> /tmp/smi_$tag.txt; echo >> /tmp/smi_$tag.txt; sleep 0.35; done
  wait $pid
  python - "$tag" <<'PY'
import json, re, sys
tag = sys.argv[1]
d = json.loads(open(f"/tmp/b_{tag}.json").read().strip().splitlines()[-1])
txt = open(f"/tmp/smi_{tag}.txt").read()
pw = [float(x) for x in re.findall(r"Power \(W\): ([0-9.]+)", txt)]
ck = [float(x) for x in re.findall(r"sclk[^(]*\(([0-9.]+)Mhz\)", txt, flags=re.I)]
rate = d["value"] * 1e6
P = sum(pw) / max(1, len(pw)); F = sum(ck) / max(1, len(ck))
print(f"{tag:8s} {rate/1e9:7.1f} Gsamples/s  launch {d['roofline']['avg_launch_ms']:.4f} ms  frac {d['roofline']['frac']:.3f}  package {P:6.0f} W  sclk {F:5.0f} MHz  {P/rate*1e9:5.2f} nJ/sample  "
      f"{rate/1e9/(F/1e3):6.1f} Gsamples/s per GHz")
PY
done
cp /tmp/orig.so gnuradio4_amd/libgr4hip.so
