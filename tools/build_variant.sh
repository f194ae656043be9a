#!/bin/bash
# developer tool: gnuradio4_amd/libgr4hip_<tag>.so = the current objects with ONE source recompiled under extra flags
#   tools/build_variant.sh timing chain_fused.hip -DGR4_FD_TIMING
set -e
TAG=$1; SRC=$2; shift 2
cd "$(dirname "$0")/../gnuradio4_amd/csrc"
O=../../build/obj
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -Wall -Wno-unused-function "$@" -c $SRC -o $O/${SRC%.hip}_$TAG.o
OBJS=""
for s in runtime fir fir_interp fir_decim_fd fft fft_fast_pk math ewise iir chain chain_fused chain_td chain16 fir_batched fir_bf16 fir_f16 fir_decim_f16 fir_exact design f64 fanin; do
  if [ "$s.hip" = "$SRC" ]; then OBJS="$OBJS $O/${s}_$TAG.o"; else OBJS="$OBJS $O/$s.o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -o ../libgr4hip_$TAG.so $OBJS -ldl
echo "built libgr4hip_$TAG.so"
