#!/bin/bash
# builds the C++ host-layer test programs and the plugin (g++ -std=c++20); the four targets compile in parallel and only when stale
set -e
cd "$(dirname "$0")"
OUT=../../build/host
mkdir -p $OUT
HDRS="include/gr4/core.hpp include/gr4/blocks.hpp include/gr4/merge.hpp include/gr4/hip.hpp include/gr4/plugin.hpp include/gr4/grc.hpp ../../include/gr4hip.h"
CXX="g++ -std=c++20 -Wall -Wextra -Iinclude"
LINK="-L.. -lgr4hip -Wl,-rpath,\$ORIGIN/../../gnuradio4_amd -Wl,-rpath,/opt/rocm/lib"
stale() { # target sources...
  local t=$1; shift
  [ ! -e "$t" ] && return 0
  for s in "$@" $HDRS ../libgr4hip.so; do [ "$s" -nt "$t" ] && return 0; done
  return 1
}
pids=()
# the CPU program links the library too: the filter-design and window functions behind BasicFilter / FFT are host code in libgr4hip.so
if stale $OUT/test_host_cpu tests/test_host_cpu.cpp; then $CXX -O2 tests/test_host_cpu.cpp -o $OUT/test_host_cpu $LINK & pids+=($!); fi
if stale $OUT/test_host_fanin tests/test_host_fanin.cpp; then $CXX -O2 tests/test_host_fanin.cpp -o $OUT/test_host_fanin $LINK & pids+=($!); fi
if stale $OUT/test_host_device tests/test_host_device.cpp; then $CXX -O2 tests/test_host_device.cpp -o $OUT/test_host_device $LINK & pids+=($!); fi
# the plugin (gr_plugin_make / gr_plugin_free) next to libgr4hip.so, and a loader test that links neither
if stale ../libgr4hip_blocks.so plugin/gr4hip_blocks.cpp; then
  $CXX -O1 -fPIC -shared -fvisibility=hidden plugin/gr4hip_blocks.cpp -o ../libgr4hip_blocks.so -L.. -lgr4hip -Wl,-rpath,'$ORIGIN' -Wl,-rpath,/opt/rocm/lib & pids+=($!)
fi
if stale $OUT/bench_host_feed tests/bench_host_feed.cpp; then $CXX -O2 tests/bench_host_feed.cpp -o $OUT/bench_host_feed $LINK & pids+=($!); fi
if stale $OUT/bench_host_fanin tests/bench_host_fanin.cpp; then $CXX -O2 tests/bench_host_fanin.cpp -o $OUT/bench_host_fanin $LINK & pids+=($!); fi
if stale $OUT/dump_signal_generator tests/dump_signal_generator.cpp; then $CXX -O2 tests/dump_signal_generator.cpp -o $OUT/dump_signal_generator $LINK & pids+=($!); fi
if stale $OUT/test_host_plugin tests/test_host_plugin.cpp; then $CXX -O2 tests/test_host_plugin.cpp -o $OUT/test_host_plugin -ldl & pids+=($!); fi
for p in "${pids[@]}"; do wait $p; done
echo "built $(realpath $OUT)"
