#!/bin/bash
# builds gnuradio4_amd/libgr4hip.so for gfx950 (hipcc cross-compiles without a GPU)
set -e
cd "$(dirname "$0")"
OUT=../libgr4hip.so
SRCS="runtime.hip fir.hip fir_interp.hip fir_decim_fd.hip fft.hip fft_fast_pk.hip math.hip ewise.hip iir.hip chain.hip chain_fused.hip chain_td.hip chain16.hip fir_batched.hip fir_bf16.hip fir_f16.hip fir_decim_f16.hip fir_exact.hip design.hip f64.hip fanin.hip"
mkdir -p ../../build/obj
OBJS=""
pids=()
for s in $SRCS; do
  o=../../build/obj/${s%.hip}.o
  OBJS="$OBJS $o"
  if [ ! -f "$o" ] || [ "$s" -nt "$o" ] || [ common.hpp -nt "$o" ] || [ fir_window.hpp -nt "$o" ] || [ fft_radix.hpp -nt "$o" ] || [ buffer_ops.hpp -nt "$o" ] || [ fft_kernels.hpp -nt "$o" ] || [ fft_smooth.hpp -nt "$o" ] || [ fft_smooth_sizes.inc -nt "$o" ] || [ wave16_common.hpp -nt "$o" ] || [ ewise.hpp -nt "$o" ] || [ fir_band_hooks.hpp -nt "$o" ] || [ fir_exact.hpp -nt "$o" ] || [ fir_f16_common.hpp -nt "$o" ] || [ build.sh -nt "$o" ] || [ ../../include/gr4hip.h -nt "$o" ]; then
    # hipcc's SLP vectoriser turns float math into v_pk_*_f32 on register pairs it assembles with v_mov: measured slower on every kernel
    # of this library (fused chain -12 %, register-window FIR -32 %) except the two FFT sizes in fft_fast_pk.hip
    SLP=-fno-slp-vectorize
    [ "$s" = fft_fast_pk.hip ] && SLP=
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $SLP -Wall -Wno-unused-function -c "$s" -o "$o" &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT $OBJS -ldl
echo "built $(realpath $OUT)"
