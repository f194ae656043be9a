cd $GRAFT_REPO_ROOT
cp gnuradio4_amd/libgr4hip.so /tmp/orig.so
for tag in base slex sbal; do
  if [ $tag = base ]; then cp /tmp/orig.so gnuradio4_amd/libgr4hip.so; else cp gnuradio4_amd/libgr4hip_$tag.so gnuradio4_amd/libgr4hip.so; fi
  python tools/fft_smooth_sweep.py 2>/dev/null | tail -1 > gpurun_out/smooth_sweep_$tag.json
done
cp /tmp/orig.so gnuradio4_amd/libgr4hip.so
wc -c gpurun_out/smooth_sweep_*.json
