#!/usr/bin/env python
"""bench.py -- headline metric of BASELINE.json on MI355X.

  python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)

A "step" is one pass of the hot path over one batch: a 2^30-sample complex<float> stream already resident in HBM
-> 256-tap FIR -> 8192-point FFT frames -> |X|^2 (BASELINE.json configs[1]; rectangular window).
N > 1 (configs[4] shape): every rank owns one SDR channel of the same size (weak scaling) and the per-frame spectra
are summed over channels by an RCCL reduce_scatter that overlaps the next chunk's compute.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

NTAPS, NFFT = 256, 8192
ALGO_BYTES_PER_SAMPLE = 12.0  # 8 B complex<float> in + 4 B float mag2 out (fused lower bound)
HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def _usable_cores() -> int:
    """hardware threads this process may actually use: affinity mask and cgroup CPU quota, not just os.cpu_count()"""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_baseline(target_seconds: float = 12.0):
    """Reference-faithful CPU path (oracle, -O3 -march=native like core/benchmarks/CMakeLists.txt:19-23) on a bounded
    sample of the same workload.  Single chain == one thread (GR4 never splits one block chain across threads)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np

    import oracle_lib as O
    L = O.lib(fast=True)
    b = O.design_taps_hamming_lowpass(NTAPS, 0.1)
    x = O.signal_c32(42, 8 * NFFT)
    t0 = time.perf_counter()
    O.chain(b, x, NFFT, 0, truth=False, L=L)
    dt = time.perf_counter() - t0
    frames = int(max(8, min(4096, target_seconds / (dt / 8))))
    x = O.signal_c32(42, frames * NFFT)
    t0 = time.perf_counter()
    O.chain(b, x, NFFT, 0, truth=False, L=L)
    dt = time.perf_counter() - t0
    res = {"value": round(frames * NFFT / dt / 1e6, 3), "unit": "Msamples/s", "cores": 1, "kind": "port",
           "sample": f"{frames} frames x {NFFT} samples (one chain, float32 oracle restatement of fir_filter+FFT+mag2, 1 thread of {os.cpu_count()})"}
    # GR4's multiThreaded policy never splits one chain across threads, so the host's best case is one independent
    # chain per core.  Same oracle, one chain per hardware thread (ctypes releases the GIL), a few seconds.
    try:
        from concurrent.futures import ThreadPoolExecutor
        ncore = _usable_cores()
        per = max(8, min(frames, 384, int(3.0 / (dt / frames))))  # <= ~3 s and <= 12 MB of output per thread
        xs = [O.signal_c32(42 + c, per * NFFT) for c in range(min(ncore, 8))]
        t0 = time.perf_counter()
        with ThreadPoolExecutor(ncore) as ex:
            list(ex.map(lambda c: O.chain(b, xs[c % len(xs)], NFFT, 0, truth=False, L=L), range(ncore)))
        dta = time.perf_counter() - t0
        res["all_cores"] = {"value": round(ncore * per * NFFT / dta / 1e6, 3), "unit": "Msamples/s", "cores": ncore,
                            "sample": f"{ncore} independent chains x {per} frames, one per hardware thread"}
    except Exception as e:  # the single-core number above is the contract; this one is context
        res["all_cores"] = {"error": str(e)}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--log2-samples", type=int, default=30, help="stream length per step and rank (default 2^30 = configs[1])")
    ap.add_argument("--log2-chunk", type=int, default=28, help="samples per launch")
    ap.add_argument("--algo", type=int, default=0, help="0 auto, 1 unfused, 2 fused time-domain, 3 fused frequency-domain")
    ap.add_argument("--fanin-cus", type=int, default=32, help="N > 1: CUs left to the RCCL fan-in kernels (the fused kernel is persistent and fills every CU it gets)")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, the measured configuration) or gloo (functional check of the N > 1 path on a box with fewer GPUs than ranks)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import gnuradio4_amd as G
    from gnuradio4_amd import capi, fanin

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    local = local % torch.cuda.device_count() if args.dist_backend != "nccl" else local
    torch.cuda.set_device(local)
    if world > 1:
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))  # nccl == RCCL on ROCm
        else:
            dist.init_process_group(args.dist_backend)

    n = 1 << args.log2_samples
    chunk = min(1 << args.log2_chunk, n)
    nchunks = n // chunk
    frames_per_chunk = chunk // NFFT

    # synthetic input, generated on the device (noise seed 42 + channel index, tone at 0.1 fs)
    x = G.synth_c32(n, seed=42 + rank)
    out = torch.empty((n // NFFT, NFFT), dtype=torch.float32, device="cuda")
    # fan-in result: this rank's shard of the channel-summed spectra, one slab per launch
    rs_out = torch.empty((nchunks, frames_per_chunk // world, NFFT), dtype=torch.float32, device="cuda") if world > 1 else None
    import numpy as np
    k = np.arange(NTAPS, dtype=np.float64)
    w = np.empty(NTAPS, np.float32)
    capi.check(capi.lib().gr4hip_window_create(2, w.ctypes.data, NTAPS, 1.6), "window")
    taps = (w.astype(np.float64) * 0.2 * np.sinc(0.2 * (k - (NTAPS - 1) / 2.0)))
    taps = (taps / taps.sum()).astype(np.float32)  # Hamming windowed-sinc, fc = 0.1, DC gain 1
    chain = G.Chain(taps, NFFT, "None", args.algo)
    if world > 1 and args.fanin_cus > 0:  # the collective of chunk c runs beside the transform of chunk c+1 instead of behind it
        n_cu = torch.cuda.get_device_properties(local).multi_processor_count
        chain.set_max_workgroups(max(1, n_cu - args.fanin_cus))

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(nchunks)]
    kernel_ms = []

    def step(record: bool):
        chain.reset()
        works = []
        for c in range(nchunks):
            xs = x[c * chunk:(c + 1) * chunk]
            os_ = out[c * frames_per_chunk:(c + 1) * frames_per_chunk]
            if record:
                ev[c][0].record()
            chain.process_bulk(xs, os_)
            if record:
                ev[c][1].record()
            if world > 1:  # fan-in combiner (math::Add over channels) as reduce_scatter, async on RCCL's stream
                works.append(fanin.fan_in_sum(os_, rs_out[c], async_op=True)[1])
        for wk in works:
            if wk is not None:
                wk.wait()

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
        # event times are read after the timed region
    fence()
    dt = time.perf_counter() - t0
    for a, b_ in ev:  # last step's launches (all steps are identical work)
        kernel_ms.append(a.elapsed_time(b_))
    tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    if rank == 0:
        total_samples = float(n) * world * args.steps
        value = total_samples / dt / 1e6
        launch_ms = sum(kernel_ms) / len(kernel_ms)
        achieved = chunk * ALGO_BYTES_PER_SAMPLE / (launch_ms * 1e-3) / 1e9
        algo_names = {1: "fir_poly_kernel + fft_block_kernel (unfused)", 2: "chain_fused_td_kernel", 3: "chain_fused_fd_kernel"}
        traffic = None  # HBM bytes per launch from the committed PMC profile of the same kernel + launch size (profiles/), else null
        try:
            prof = json.load(open(os.path.join(ROOT, "profiles", "latest_pmc.json")))
            if prof.get("kernel") == algo_names.get(chain.algo) and prof.get("samples_per_launch") == chunk:
                traffic = prof["hbm_bytes_per_launch"]
        except Exception:
            pass
        res = {
            "metric": "Msamples/s through 256-tap cplx FIR->8192-pt FFT chain; %HBM roofline @1/2/4/8 GPU",
            "value": round(value, 3), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"complex<float> {NTAPS}-tap FIR -> {NFFT}-pt FFT -> mag2, 2^{args.log2_samples}-sample stream per GPU "
                                   f"(BASELINE.json configs[1]), rectangular window, {nchunks} launches of 2^{args.log2_chunk} samples"
                                   + (f"; {world} channels, {'RCCL reduce_scatter' if args.dist_backend == 'nccl' else args.dist_backend + ' all_reduce (functional check only)'} fan-in sum (configs[4] shape)" if world > 1 else ""),
                       "chain_algo": algo_names.get(chain.algo, str(chain.algo)), "parallelism": f"{world} independent channel(s), 1 per GPU"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": traffic, "kernel": algo_names.get(chain.algo, str(chain.algo)),
                         "algorithmic_bytes_per_launch": chunk * ALGO_BYTES_PER_SAMPLE, "avg_launch_ms": round(launch_ms, 4),
                         "frac_of_measured_copy_rate": round(achieved / 6290.0, 4)},  # 6.29 TB/s: what a float4 copy reaches on this part (MI355X_MICROARCH.md)
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline()
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
