#!/bin/bash
# developer tool (round 5): SQ counters of chain_fd_kernel<0,13> (one 2^28-sample launch) for the default build and the timing-only variants of tools/ab_headline_r05.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cp $R/gnuradio4_amd/libgr4hip.so /tmp/orig.so
for tag in base "$@"; do
  if [ $tag = base ]; then cp /tmp/orig.so $R/gnuradio4_amd/libgr4hip.so; else cp $R/gnuradio4_amd/libgr4hip_$tag.so $R/gnuradio4_amd/libgr4hip.so; fi
  rm -rf /tmp/pm_$tag
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS -d /tmp/pm_$tag -o p --output-format csv -- \
    python $R/bench.py --steps 1 --warmup 1 --log2-samples 28 --no-cpu-baseline --no-graph8 --no-hann-row --no-secondary --no-live-traffic --no-verify > /tmp/pm_$tag.log 2>&1
  python - $tag <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob(f"/tmp/pm_{tag}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "chain_fd_kernel<0" in r["Kernel_Name"] and int(r["Grid_Size"]) >= 256 * 512:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(tag, {k: "%.3e" % (sum(v) / len(v)) for k, v in sorted(acc.items())}, "dispatches", max((len(v) for v in acc.values()), default=0))
PY
done
cp /tmp/orig.so $R/gnuradio4_amd/libgr4hip.so
