#!/bin/bash
# round-2 first GPU call: baseline parity + bench + per-phase stamps of the headline kernel
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2a
python -m pytest tests -m gpu -x -q > gpurun_out/r2a/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2a/pytest.log
python bench.py > gpurun_out/r2a/bench.json 2> gpurun_out/r2a/bench.err
python bench.py --channels 8 --steps 5 --warmup 2 > gpurun_out/r2a/bench_8ch_1gpu.json 2> gpurun_out/r2a/bench_8ch.err
python tools/fd_timing.py > gpurun_out/r2a/fd_timing.txt 2>&1
tail -3 gpurun_out/r2a/pytest.log; cat gpurun_out/r2a/bench.json gpurun_out/r2a/bench_8ch_1gpu.json; cat gpurun_out/r2a/fd_timing.txt
