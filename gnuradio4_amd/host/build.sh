#!/bin/bash
# builds the C++ host-layer test programs (g++ -std=c++20; the device tests link libgr4hip.so)
set -e
cd "$(dirname "$0")"
OUT=../../build/host
mkdir -p $OUT
# the CPU program links the library too: the filter-design and window functions behind BasicFilter / FFT are host code in libgr4hip.so
g++ -std=c++20 -O2 -Wall -Wextra -Iinclude tests/test_host_cpu.cpp -o $OUT/test_host_cpu -L.. -lgr4hip -Wl,-rpath,'$ORIGIN/../../gnuradio4_amd' -Wl,-rpath,/opt/rocm/lib
if [ -f tests/test_host_device.cpp ]; then
  g++ -std=c++20 -O2 -Wall -Wextra -Iinclude tests/test_host_device.cpp -o $OUT/test_host_device -L.. -lgr4hip -Wl,-rpath,'$ORIGIN/../../gnuradio4_amd' -Wl,-rpath,/opt/rocm/lib
fi
echo "built $(realpath $OUT)"
