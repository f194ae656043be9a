#!/bin/bash
cd "$(dirname "$0")/../.."
for i in 1 2 3 4 5 6; do
  GR4HIP_DBG_STALE=1 python -m pytest "tests/test_host_cpp.py::test_device_graphs_match_oracle" -q -m gpu -x -s > /tmp/hc.txt 2>&1
  echo "run $i: $(tail -1 /tmp/hc.txt)"; grep -h "stale\]" /tmp/hc.txt | sort | uniq -c | head -5
done
