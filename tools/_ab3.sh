cp gnuradio4_amd/libgr4hip.so /tmp/orig.so
for tag in nocasc pipe1 pipe2; do cp gnuradio4_amd/libgr4hip_$tag.so gnuradio4_amd/libgr4hip.so; echo "== $tag"; timeout 300 python tools/bench_configs2.py 2>/dev/null | grep "^{" | cut -c1-200; done
cp /tmp/orig.so gnuradio4_amd/libgr4hip.so
