#!/usr/bin/env python
"""bench.py -- headline metric of BASELINE.json on MI355X.

  python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)

A "step" is one pass of the hot path over one batch of synthetic input that is already resident in HBM.

  N = 1 (BASELINE.json configs[1]): ONE 2^30-sample complex<float> stream -> 256-tap FIR -> 8192-point FFT frames -> |X|^2
        (rectangular window), 4 launches of 2^28 samples, FIR history carried between launches.
  N > 1 (configs[4]): the 8-channel graph.  8 independent SDR channels of 2^30 samples each, channel c on
        rank c mod N (8/N channels per GPU, each with its own chain handle and HIP stream), the combiner math::Add<float> with
        n_inputs = 8 (blocks/math/.../Math.hpp:73-108) as a local n-ary fold on every GPU followed by ONE exchange step per launch:
        an RCCL reduce_scatter(sum) of the per-frame spectra, asynchronous, overlapping the next chunk's transforms.
        Total work is fixed (8 channels) whatever N: strong scaling.  `--channels 8` runs the same graph on one GPU.

After the timed region a self-check pulls sampled output frames back and compares them with the float64 oracle on the same
device-generated input (checker only; exit code 3 above 1e-5).  Prints ONE JSON line on rank 0.
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
XGMI_LINK_GBS = 153.0         # per link and direction (7 links per GPU), nominal
PARITY_TOL = 1e-5             # BASELINE.json north_star: <= 1e-5 rel for float32 FIR/FFT
KERNEL_SYMBOLS = {1: "gr4::fir_poly_kernel + gr4::fft_fast_kernel (unfused)", 3: "gr4::chain_fd_kernel<0, 13>",
                  4: "gr4::fir_poly_kernel + gr4::fft_fast_kernel (time domain)"}


class Watchdog:
    """A multi-rank run nobody watches must not hang its lease: every phase that waits for another rank (rendezvous, communicator, collectives, the timed
    loop) runs under a wall-clock deadline.  When one passes, this rank says where it is (rank-tagged, stderr), rank 0 prints the bench's JSON line with an
    "error" key, and the process exits with code 4 (torch.distributed.run then takes the other ranks down).  A hang inside the library's own communicator
    (--fanin-impl capi, never exercised between two GPUs before its first driver run) gets ONE retry: every rank re-executes itself with
    --fanin-impl torch --fanin-cus 0, the most conservative configuration."""

    def __init__(self, rank: int, world: int, argv, timeout: float):
        import threading
        self.rank, self.world, self.argv, self.timeout = rank, world, list(argv), timeout
        self.phase, self.deadline, self.retryable, self.t0 = "start", None, False, time.time()
        self.lock = threading.Lock()
        if world > 1 and timeout > 0:
            threading.Thread(target=self._run, daemon=True).start()

    def enter(self, phase: str, seconds: float = None, retryable: bool = False):
        with self.lock:
            self.phase, self.retryable = phase, retryable
            self.deadline = time.time() + (seconds if seconds is not None else self.timeout)

    def leave(self):
        with self.lock:
            self.deadline = None

    def _run(self):
        while True:
            time.sleep(0.5)
            with self.lock:
                late = self.deadline is not None and time.time() > self.deadline
                phase, retryable = self.phase, self.retryable
            if not late:
                continue
            msg = f"rank {self.rank} of {self.world}: no progress in phase '{phase}' within its deadline ({self.timeout:.0f} s budget per waiting phase); {time.time() - self.t0:.0f} s since start"
            print(f"[bench][watchdog] {msg}", file=sys.stderr, flush=True)
            if retryable and os.environ.get("GR4HIP_BENCH_RETRY") != "1":
                print(f"[bench][watchdog] rank {self.rank}: retrying once with --fanin-impl torch --fanin-cus 0", file=sys.stderr, flush=True)
                os.environ["GR4HIP_BENCH_RETRY"] = "1"
                args = [a for a in self.argv]
                os.execv(sys.executable, [sys.executable] + args + ["--fanin-impl", "torch", "--fanin-cus", "0"])
            if self.rank == 0:
                print(json.dumps({"metric": "Msamples/s through 256-tap cplx FIR->8192-pt FFT chain; %HBM roofline @1/2/4/8 GPU", "value": None, "n_gpus": self.world,
                                  "error": msg, "phase": phase, "retried": os.environ.get("GR4HIP_BENCH_RETRY") == "1"}), flush=True)
            os._exit(4)


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


def _oracle():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    return O


def cpu_baseline(target_seconds: float = 12.0):
    """Reference-faithful CPU path (oracle, -O3 -march=native like core/benchmarks/CMakeLists.txt:19-23) on a bounded
    sample of the same workload.  Single chain == one thread (GR4 never splits one block chain across threads)."""
    O = _oracle()
    provenance = O.build_fast_for_this_host()  # -march=native must mean THIS box's cores, not the container the binary was first built in
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
           "sample": f"{frames} frames x {NFFT} samples (one chain, float32 oracle restatement of fir_filter+FFT+mag2, 1 thread of {os.cpu_count()})",
           "binary": "oracle/liboracle_fast.so, gcc -O3 -march=native: " + provenance}
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
                            "sample": f"{ncore} independent chains x {per} frames, one per usable hardware thread: {ncore} of the host's {os.cpu_count()} threads (this process's affinity mask); "
                                      f"the whole host's one-chain-per-thread ceiling is about {os.cpu_count() / max(ncore, 1):.0f} x this value"}
    except Exception as e:  # the single-core number above is the contract; this one is context
        res["all_cores"] = {"error": str(e)}
    return res


def _rel_err(got, truth):
    """the parity metric of tests/test_gpu_parity.py::_rel: relative above the rms level, rms-normalised below it"""
    import numpy as np
    truth = np.asarray(truth, np.float64)
    scale = np.maximum(np.abs(truth), np.sqrt(np.mean(truth ** 2)))
    return float(np.max(np.abs(np.asarray(got, np.float64) - truth) / scale))


def oracle_frame(O, taps, x_dev, f):
    """float64 oracle |FFT(fir(x))|^2 of frame f of one channel: the frame before it and frame f go through the oracle chain (255 samples
    of history are all a 256-tap FIR needs).  The bench streams the same buffer step after step without resetting the chain, so the
    frame in front of frame 0 is the LAST frame of the buffer (the previous step's tail is the carried history)."""
    import numpy as np
    nfr = x_dev.numel() // NFFT
    prev = x_dev[((f - 1) % nfr) * NFFT:((f - 1) % nfr + 1) * NFFT].cpu().numpy()
    cur = x_dev[f * NFFT:(f + 1) * NFFT].cpu().numpy()
    out, _ = O.chain(taps, np.concatenate([prev, cur]), NFFT, 0, truth=True)
    return out.reshape(-1, NFFT)[1]


def live_traffic(kernel_symbol, log2_samples, log2_chunk):
    """HBM bytes per launch of the headline kernel from the PMC counters, as MI355X_MICROARCH.md (HBM section) prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE
    counter-only rocprofv3 passes (they do not fit one pass; no trace domains beside them), FETCH_SIZE doubled (gfx950 tallies the 128-byte requests of 16-byte-per-lane
    streaming reads -- this kernel's LDS-DMA -- at 64 bytes), per full-size dispatch.  Each pass is a short child run of this script at the same launch size.
    None when rocprofv3 is not there or a pass fails: the line then quotes the committed profile's figure only."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 not on PATH"
    if "rocprofiler" in os.environ.get("LD_PRELOAD", "") or any(k.startswith(("ROCPROF", "ROCP_TOOL")) for k in os.environ):
        return None, "this run is itself being profiled: no counter passes nested inside a trace"
    match = kernel_symbol.split("gr4::")[-1].split("<")[0] + "<" + kernel_symbol.split("<")[-1].split(",")[0]  # chain_fd_kernel<0 (the Hann row's instantiation is <1, ...>)
    kib = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        out = tempfile.mkdtemp(prefix="gr4pmc_", dir="/tmp")
        cmd = ["rocprofv3", "--pmc", ctr, "-d", out, "-o", "p", "--output-format", "csv", "--", sys.executable, os.path.abspath(__file__), "--steps", "2", "--warmup", "1",
               "--log2-samples", str(log2_samples), "--log2-chunk", str(log2_chunk), "--no-cpu-baseline", "--no-graph8", "--no-hann-row", "--no-secondary", "--no-verify", "--no-live-traffic"]
        try:
            subprocess.run(cmd, capture_output=True, text=True, timeout=240, cwd="/tmp", env={**os.environ, "TMPDIR": "/tmp", "GR4HIP_BENCH_CHILD": "1"})
            rows = []
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                rows += [r for r in csv.DictReader(open(f)) if match in r["Kernel_Name"] and r["Counter_Name"] == ctr]
            full = max((int(r["Grid_Size"]) for r in rows), default=0)  # (the guard's probe of a stream's first call is a dispatch of the same kernel on eight frames)
            vals = [float(r["Counter_Value"]) for r in rows if int(r["Grid_Size"]) == full]
            if not vals:
                return None, f"no {ctr} rows for {match}"
            kib[ctr] = sum(vals) / len(vals)
        except Exception as e:
            return None, f"{ctr} pass failed: {str(e)[:120]}"
        finally:
            shutil.rmtree(out, ignore_errors=True)
    return int((2.0 * kib["FETCH_SIZE"] + kib["WRITE_SIZE"]) * 1024), (f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate counter-only passes over short runs of this script at the same "
                                                                      f"launch size: FETCH_SIZE {kib['FETCH_SIZE']:.0f} KiB x 2 (gfx950 16-byte/lane correction) + WRITE_SIZE {kib['WRITE_SIZE']:.0f} KiB per dispatch")


def secondary_configs(G, verify):
    """BASELINE.json configs[2] and configs[3] on this GPU: steady-state rates (back-to-back launches between two events, medians over rounds), each output checked against
    the oracle on a sampled stretch.  Not the headline metric: rows beside it, so that the driver's record carries them."""
    import numpy as np
    import torch
    from gnuradio4_amd import capi

    def lowpass(ntaps, fc):
        k = np.arange(ntaps, dtype=np.float64)
        t = np.hamming(ntaps) * 2 * fc * np.sinc(2 * fc * (k - (ntaps - 1) / 2.0))
        return (t / t.sum()).astype(np.float32)

    def rate(fn, reps, rounds=5):
        """median ms per call over `rounds` groups of back-to-back calls (>= 30 ms each) between events, with NO host synchronisation between the groups and >= 60 ms of
        back-to-back calls in front: the first ~20 ms after a host sync run at a sagged clock (profiles/r02_clock_ramp.txt; the headline's own pre-warm exists for the same
        reason).  Until round 5 this synchronised after every group of 20 calls (4 - 10 ms), which under-reported these rows by ~10 % against tools/_timing.py's steady()."""
        import time
        fn()
        torch.cuda.synchronize()
        t0, k = time.perf_counter(), 0
        while True:
            fn()
            k += 1
            if k % 4 == 0:
                torch.cuda.synchronize()
                if time.perf_counter() - t0 >= 0.06:
                    break
        per = (time.perf_counter() - t0) / k
        n = max(reps, int(0.03 / max(per, 1e-6)))
        for _ in range(max(4, int(0.02 / max(per, 1e-6)))):  # the syncs above let the clock sag again: run up, no sync from here on
            fn()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(rounds + 1)]
        ev[0].record()
        for r in range(rounds):
            for _ in range(n):
                fn()
            ev[r + 1].record()
        ev[-1].synchronize()
        ms = [ev[r].elapsed_time(ev[r + 1]) / n for r in range(rounds)]
        return sorted(ms)[len(ms) // 2]

    TIMING_NOTE = "steady state: >= 60 ms of back-to-back launches in front, median of 5 groups of >= 30 ms between events, no host sync between the groups (until round 5: a sync per 20 launches, ~10 % lower)"
    out = {}
    O = _oracle() if verify else None
    # configs[3]: 64 channels x 256 taps (csrc/fir_f16.hip: two-term f16 splits under a block exponent on the f16 matrix pipe, every segment judged)
    nch, ntaps, n = 64, 256, 1 << 22
    taps = np.stack([lowpass(ntaps, 0.05 + 0.005 * c) for c in range(nch)])
    xb = torch.stack([G.synth_f32(n, seed=42 + c) for c in range(nch)])
    yb = torch.empty_like(xb)
    fb = G.FirBatched(taps)
    ms = rate(lambda: fb.process_bulk(xb, yb), 20)
    row = {"workload": "64 channels x 256-tap float FIR x 2^22 samples (BASELINE.json configs[3]), csrc/fir_f16.hip", "value": round(nch * n / (ms * 1e-3) / 1e6, 1), "unit": "Msamples/s",
           "ms_per_launch": round(ms, 4), "bytes_per_sample": 8, "hbm_frac": round(nch * n * 8 / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
           "float32_equivalent_TFLOP/s": round(nch * n * 2 * ntaps / (ms * 1e-3) / 1e12, 1), "executed_f16_TFLOP/s": round(nch * n * 3 * 2 * 288 / (ms * 1e-3) / 1e12, 1)}
    row["arithmetic"] = "f16x2 (two-term f16 splits under a per-segment block exponent: 22-bit products, float32 accumulation; segments that reject > 21 dB again in float64)"
    row["timing"] = TIMING_NOTE
    try:  # the 24-bit form beside it: three-term bf16 products (csrc/fir_bf16.hip) on the same launch
        capi.developer_switch("GR4HIP_FIR_NO_F16X2", 1)
        fb3 = G.FirBatched(taps)
        ms3 = rate(lambda: fb3.process_bulk(xb, yb), 20)
        row["bf16x3_24bit_products_msamples"] = round(nch * n / (ms3 * 1e-3) / 1e6, 1)
        del fb3
    except Exception as e:
        row["bf16x3_24bit_products_msamples"] = str(e)[:120]
    finally:
        capi.developer_switch("GR4HIP_FIR_NO_F16X2", 0)
    if verify:
        fb2 = G.FirBatched(taps)  # (a fresh history: the timed handle has seen the span many times)
        fb2.process_bulk(xb, yb)
        errs = []
        for c in (0, 37, 63):
            m = 3 * 4096 + 100
            truth, _ = O.fir(taps[c], xb[c, :m].cpu().numpy())
            errs.append(_rel_err(yb[c, :m].cpu().numpy(), truth))
        row["verify_max_rel_err"] = float(f"{max(errs):.3e}")
    out["configs[3]"] = row
    del xb, yb, fb
    # configs[2]: decimate-by-8 1024-tap FIR + Butterworth order 8 (4 biquads) at the decimated rate, gr4hip_fir_process -> gr4hip_iir_process
    n2 = 1 << 27
    x = G.synth_f32(n2, seed=42)
    b1024 = lowpass(1024, 0.05)
    fir = G.fir_filter(b1024, torch.float32, decimate=8)
    bi, ai = G.blocks.design_iir(capi.LOWPASS, 8, 0.05, float("nan"), 1.0, capi.BUTTERWORTH)
    iir = G.iir_filter(bi, ai)
    yd = torch.empty(n2 // 8, dtype=torch.float32, device="cuda")
    yo = torch.empty_like(yd)

    def both():
        fir.process_bulk(x, yd)
        iir.process_bulk(yd, yo)
    ms = rate(both, 20)
    row = {"workload": "decimate-by-8 1024-tap float FIR (csrc/fir_decim_f16.hip) + 4 biquads x 2^27 input samples (BASELINE.json configs[2]), two launches", "value": round(n2 / (ms * 1e-3) / 1e6, 1),
           "unit": "Msamples/s (input rate)", "ms_per_pass": round(ms, 4), "bytes_per_input_sample": 5.5, "hbm_frac": round(n2 * 5.5 / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    row["arithmetic"] = "decimator: f16x2 (22-bit products, float32 accumulation; segments that reject > 21 dB again in float64); cascade: float32, exact parallel-in-time scan"
    row["timing"] = TIMING_NOTE
    try:  # the 24-bit forms beside it: the frequency-domain decimator (float32 transforms) -- what decimate-by-8 at 1024 taps takes without the f16 kernel
        capi.developer_switch("GR4HIP_FIR_NO_DECIM_F16", 1)
        fir3, iir3 = G.fir_filter(b1024, torch.float32, decimate=8), G.iir_filter(bi, ai)

        def both3():
            fir3.process_bulk(x, yd)
            iir3.process_bulk(yd, yo)
        ms3 = rate(both3, 20)
        row["f32_frequency_domain_decimator_msamples"] = round(n2 / (ms3 * 1e-3) / 1e6, 1)
        del fir3, iir3
    except Exception as e:
        row["f32_frequency_domain_decimator_msamples"] = str(e)[:120]
    finally:
        capi.developer_switch("GR4HIP_FIR_NO_DECIM_F16", 0)
    if verify:
        fir2, iir2 = G.fir_filter(b1024, torch.float32, decimate=8), G.iir_filter(bi, ai)
        fir2.process_bulk(x, yd)
        iir2.process_bulk(yd, yo)
        m = 8 * 20000
        td, _ = O.fir_decim(b1024, x[:m].cpu().numpy(), 8)
        ti = O.iir_cascade(O.make_sections([(bb, aa) for bb, aa in zip(bi, ai)]), td.astype(np.float32), O.DF_II, f64=True)
        row["verify_max_rel_err"] = float(f"{max(_rel_err(yd[: m // 8].cpu().numpy(), td), _rel_err(yo[: m // 8].cpu().numpy(), ti)):.3e}")
    out["configs[2]"] = row
    return out


def _compact_line(res: dict, detail_file: str) -> dict:
    """THE line rank 0 prints (the driver keeps 2 000 characters of tail and only the key NAMES of nested rows): every contract key, the secondary rows as
    top-level NUMBERS right behind `value`, short strings, < 2 000 characters.  Everything else -- the prose, per-row timing notes, per-rank logs -- goes to `detail_file`."""
    def g(d, *ks, default=None):
        for k in ks:
            if not isinstance(d, dict) or k not in d:
                return default
            d = d[k]
        return d
    line = {k: res[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step") if k in res}
    if "error" in res:
        line["error"] = res["error"]
    sc = res.get("secondary_configs") or {}
    rows = {  # Msamples/s and fraction of 8 TB/s (algorithmic bytes), each checked against the float64 oracle in this run (max relative error beside it)
        "hann_msamples": g(res, "hann_second_row", "value"), "hann_frac": g(res, "hann_second_row", "frac"), "hann_err": g(res, "hann_second_row", "verify_max_rel_err"),
        "guard_in_stream_msamples": g(res, "guard_tripped_row", "in_stream_msamples"), "guard_settled_msamples": g(res, "guard_tripped_row", "settled_msamples"),
        "guard_err": g(res, "guard_tripped_row", "in_stream_verify_max_rel_err"), "narrow_msamples": g(res, "guard_tripped_row", "narrow_msamples"), "narrow_err": g(res, "guard_tripped_row", "narrow_verify_max_rel_err"),
        "configs2_msamples": g(sc, "configs[2]", "value"), "configs2_frac": g(sc, "configs[2]", "hbm_frac"), "configs2_err": g(sc, "configs[2]", "verify_max_rel_err"),
        "configs3_msamples": g(sc, "configs[3]", "value"), "configs3_frac": g(sc, "configs[3]", "hbm_frac"), "configs3_err": g(sc, "configs[3]", "verify_max_rel_err"),
        "graph8_msamples": g(res, "eight_channel_graph_on_one_gpu", "value"), "graph8_err": g(res, "eight_channel_graph_on_one_gpu", "verify", "max_rel_err"),
        "host_feed_msamples": g(res, "host_feed_row", "value"),
    }
    line.update({k: v for k, v in rows.items() if v is not None})
    for k in ("median_ms_per_step", "value_at_median_step", "prewarm_ms", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        if k in res:
            line[k] = res[k]
    cfg = res.get("config") or {}
    line["config"] = {"workload": f"configs[{4 if cfg.get('channels', 1) > 1 else 1}]: complex<float> {NTAPS}-tap FIR -> {NFFT}-pt FFT -> mag2, rectangular window, "
                                  + cfg.get("workload", "").split("rectangular window, ")[-1].split(";")[0],
                      "chain_algo": cfg.get("chain_algo"), "channels": cfg.get("channels"), "parallelism": cfg.get("parallelism"),
                      "guard": (cfg.get("dynamic_range_guard") or "").split(":")[0]}
    rf = res.get("roofline") or {}
    line["roofline"] = {k: rf[k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_over_algorithmic", "kernel", "algorithmic_bytes_per_launch",
                                          "avg_launch_ms", "median_launch_ms", "timed_launches") if k in rf}
    if "verify" in res:
        line["verify"] = {k: res["verify"][k] for k in ("verified_frames", "max_rel_err", "tolerance") if k in res["verify"]}
    cb = res.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {"value": cb.get("value"), "unit": cb.get("unit"), "cores": cb.get("cores"), "kind": cb.get("kind"),
                                "sample": (cb.get("sample") or "").split(" (")[0], "all_cores_value": g(cb, "all_cores", "value"), "all_cores": g(cb, "all_cores", "cores")}
    if "fanin" in res:
        line["fanin"] = {k: res["fanin"][k] for k in ("collective", "xgmi_ceiling_msamples", "probe_seconds_per_launch", "implementation") if k in res["fanin"]}
    if detail_file:
        try:
            os.makedirs(os.path.dirname(os.path.abspath(detail_file)), exist_ok=True)
            with open(detail_file, "w") as f:
                json.dump(res, f, indent=1)
            line["detail_file"] = os.path.relpath(detail_file, ROOT) if os.path.abspath(detail_file).startswith(ROOT) else detail_file
        except OSError:
            pass
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--channels", type=int, default=0, help="SDR channels in the graph: 0 = 1 at --gpus 1 (configs[1]) and 8 at --gpus N > 1 (configs[4])")
    ap.add_argument("--log2-samples", type=int, default=30, help="stream length per step and channel (default 2^30 = configs[1])")
    ap.add_argument("--log2-chunk", type=int, default=0, help="samples per launch (default: 2^30 = the whole 1G-sample stream in one call at one channel on one GPU -- a strict-guard "
                    "chain call returns when its launch has finished, so launches are as long as the stream allows; 2^28 when the graph has a fan-in to overlap with the next launch)")
    ap.add_argument("--algo", type=int, default=0, help="0 auto, 1 unfused, 3 fused frequency-domain, 4 time domain")
    ap.add_argument("--fanin-cus", type=int, default=32, help="N > 1: CUs left to the RCCL fan-in kernels (the fused kernel is persistent and fills every CU it gets)")
    ap.add_argument("--fanin-algo", default="auto", choices=["auto", "reduce_scatter", "all_to_all"],
                    help="N > 1: how the partial sums cross xGMI (gnuradio4_amd/fanin.py); auto = both are timed before the warm-up, the faster one runs")
    ap.add_argument("--fanin-impl", default="capi", choices=["capi", "torch"], help="N > 1 over RCCL: the collectives through the library's C entry points (gr4hip_fanin_*: the "
                    "communicator the C++ engine uses; torch.distributed only ships its id) or through torch.distributed's own communicator")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, the measured configuration) or gloo (functional check of the N > 1 path on a box with fewer GPUs than ranks)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--detail-file", default=None, help="where rank 0 writes the long form of the result (every row with its workload text, timing notes, per-rank logs); "
                                                        "default gpurun_out/bench_detail.json under the repo root; '' = nowhere.  The printed line stays under 2 000 characters")
    ap.add_argument("--no-verify", action="store_true", help="skip the oracle self-check of sampled output frames")
    ap.add_argument("--no-multi", action="store_true", help="several channels per GPU: one kernel per channel on its own stream + a separate math::Add fold (round 2's shape) "
                    "instead of ONE launch with the fold in registers (gr4hip_chain_process_multi)")
    ap.add_argument("--guard-mode", type=int, default=0, help="dynamic-range guard of the AUTO chain: 0 strict (default of the library), 1 deferred, 2 off")
    ap.add_argument("--no-graph8", action="store_true", help="N = 1: skip the extra measurement of the 8-channel graph on this one GPU (the 1-GPU point of the strong-scaling curve)")
    ap.add_argument("--prewarm-ms", type=float, default=40.0, help="untimed launches for at least this long BEFORE the counted --warmup steps: the shader clock needs ~20 ms under load to "
                    "settle (profiles/r02_clock_ramp.txt), and a short --steps run would otherwise be timed inside that ramp; reported as prewarm_ms / prewarm_steps")
    ap.add_argument("--no-live-traffic", action="store_true", help="N = 1: do not collect roofline.traffic (two short child runs of this script under rocprofv3 --pmc FETCH_SIZE / "
                                                                   "--pmc WRITE_SIZE, counters only, after the headline); traffic stays null and the committed profile's figure is quoted")
    ap.add_argument("--no-secondary", action="store_true", help="N = 1: skip the rows of BASELINE.json configs[2] and configs[3] (a second or two each, after the headline)")
    ap.add_argument("--no-hann-row", action="store_true", help="N = 1: skip the second row with the FFT block's default Hann window (SURVEY.md 8(d))")
    ap.add_argument("--fanin-timeout", type=float, default=90.0, help="N > 1: seconds any phase that waits for another rank may take before the run gives up with a rank-tagged diagnostic "
                    "(exit code 4, an \"error\" key in the JSON line) instead of hanging; 0 = no watchdog")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    import gnuradio4_amd as G
    from gnuradio4_amd import capi, fanin
    from gnuradio4_amd.blocks import chain_process_multi

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    local = local % torch.cuda.device_count() if args.dist_backend != "nccl" else local
    torch.cuda.set_device(local)
    wd = Watchdog(rank, world, sys.argv, args.fanin_timeout)
    rank_log = {}  # per-rank timings of the phases that involve other ranks (printed by every rank on stderr, gathered into the JSON line)
    if world > 1:
        wd.enter("rendezvous (init_process_group)")
        t_ = time.perf_counter()
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))  # nccl == RCCL on ROCm
        else:
            dist.init_process_group(args.dist_backend)
        dist.barrier()
        rank_log["rendezvous_s"] = round(time.perf_counter() - t_, 3)
        wd.leave()

    n_channels = args.channels or (1 if world == 1 else 8)
    plan = fanin.channel_plan(n_channels, world)
    mine = plan[rank]  # channels this rank owns
    combine = n_channels > 1
    n = 1 << args.log2_samples
    log2_chunk = args.log2_chunk or (28 if (combine or world > 1) else 30)
    chunk = min(1 << log2_chunk, n)
    nchunks = n // chunk
    frames_per_chunk = chunk // NFFT
    assert frames_per_chunk % world == 0, "frames per launch must split evenly over the ranks (reduce_scatter shards)"

    # synthetic input, generated on the device (noise seed 42 + channel index, tone at 0.1 fs)
    xs = [G.synth_c32(n, seed=42 + c) for c in mine]
    if os.environ.get("GR4HIP_BENCH_ZERO_INPUT") == "1":  # developer experiment (DVFS: how far the clocks rise when the datapaths stop toggling); never a reported number
        for x_ in xs:
            x_.zero_()
    multi = len(mine) > 1 and not args.no_multi  # this rank's channels in ONE launch, their math::Add fold kept in registers: only the sum reaches HBM
    outs = [torch.empty((n // NFFT, NFFT), dtype=torch.float32, device="cuda") for _ in mine] if not multi else []
    # combiner output of this rank: the local fold of its channels per launch (double-buffered: the collective of launch c reads
    # slab c & 1 while the fold of launch c + 1 writes the other) and its shard of the all-channel sum of every launch
    acc = torch.empty((2, frames_per_chunk, NFFT), dtype=torch.float32, device="cuda") if combine and len(mine) > 1 else None
    rs_out = torch.empty((nchunks, frames_per_chunk // world, NFFT), dtype=torch.float32, device="cuda") if world > 1 else None
    sum_out = torch.empty((n // NFFT, NFFT), dtype=torch.float32, device="cuda") if combine and world == 1 else None

    w = np.empty(NTAPS, np.float32)
    capi.check(capi.lib().gr4hip_window_create(2, w.ctypes.data, NTAPS, 1.6), "window")
    k = np.arange(NTAPS, dtype=np.float64)
    taps = (w.astype(np.float64) * 0.2 * np.sinc(0.2 * (k - (NTAPS - 1) / 2.0)))
    taps = (taps / taps.sum()).astype(np.float32)  # Hamming windowed-sinc, fc = 0.1, DC gain 1
    chains = [G.Chain(taps, NFFT, "None", args.algo) for _ in mine]
    if args.guard_mode:
        for ch in chains:
            ch.set_guard_mode(args.guard_mode)
    streams = [torch.cuda.Stream() for _ in mine] if len(mine) > 1 and not multi else [torch.cuda.current_stream()]
    n_cu = torch.cuda.get_device_properties(local).multi_processor_count
    if world > 1 and args.fanin_cus > 0:  # the collective of chunk c runs beside the transforms of chunk c+1 instead of behind them
        for ch in chains:
            ch.set_max_workgroups(max(1, n_cu - args.fanin_cus))

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(nchunks)]
    ev_all = []   # (start, end) of every timed launch of this rank's first channel, all steps
    step_ev = []  # (start, end) of every timed step on the main stream
    done = [[torch.cuda.Event() for _ in mine] for _ in range(2)]
    main_stream = torch.cuda.current_stream()
    # the fan-in of launch c runs on its own stream beside the chains of launch c + 1; slab c & 1 is free again when its fan-in has finished
    fan_stream = torch.cuda.Stream() if world > 1 else None
    fan_done = [torch.cuda.Event() for _ in range(2)] if world > 1 else None
    recv = torch.empty((frames_per_chunk, NFFT), dtype=torch.float32, device="cuda") if world > 1 else None  # all_to_all landing area
    comm = None
    if world > 1 and args.dist_backend == "nccl" and args.fanin_impl == "capi":
        wd.enter("gr4hip_fanin communicator (ncclCommInitRank through the library)", retryable=True)
        t_ = time.perf_counter()
        try:
            comm = fanin.Communicator()
            ok = 1
        except Exception as e:  # e.g. no librccl the library can open: every rank must take the same path
            print(f"[bench] rank {rank}: gr4hip_fanin communicator unavailable ({e}); falling back to torch.distributed", file=sys.stderr, flush=True)
            ok = 0
        agree = torch.tensor([ok], dtype=torch.int32, device="cuda")
        dist.all_reduce(agree, op=dist.ReduceOp.MIN)
        if int(agree.item()) == 0:
            comm = None
        rank_log["communicator_s"] = round(time.perf_counter() - t_, 3)
        wd.leave()
    if world > 1 and os.environ.get("GR4HIP_BENCH_STALL_RANK") == str(rank):  # test hook: this rank never reaches its first collective
        print(f"[bench] rank {rank}: stalling on purpose (GR4HIP_BENCH_STALL_RANK)", file=sys.stderr, flush=True)
        time.sleep(10 ** 6)
    fanin_algo = args.fanin_algo if args.fanin_algo != "auto" else "reduce_scatter"
    fanin_probe = None
    if world > 1 and args.fanin_algo == "auto":
        # which collective the node's RCCL moves faster is measured, not assumed: three fan-ins of a launch-sized slab each, max over ranks
        probe_src = torch.zeros((frames_per_chunk, NFFT), dtype=torch.float32, device="cuda")
        fanin_probe = {}
        for algo in ("reduce_scatter", "all_to_all"):
            wd.enter(f"fan-in probe: {algo} ({'gr4hip_fanin_*' if comm is not None else 'torch.distributed'})", retryable=comm is not None)
            try:
                fanin.fan_in_sum(probe_src, rs_out[0], algo=algo, recv=recv, comm=comm)
                torch.cuda.synchronize()
                dist.barrier()
                t0 = time.perf_counter()
                for _ in range(3):
                    fanin.fan_in_sum(probe_src, rs_out[0], algo=algo, recv=recv, comm=comm)
                torch.cuda.synchronize()
                t = torch.tensor([(time.perf_counter() - t0) / 3], dtype=torch.float64, device="cuda")
            except RuntimeError as e:  # a collective this RCCL build refuses: the other one runs (an error on one rank is an error on all: same outcome everywhere)
                print(f"[bench] fan-in probe: {algo} failed on rank {rank}: {e}", file=sys.stderr, flush=True)
                t = torch.tensor([float("inf")], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            fanin_probe[algo] = float(t.item())
            wd.leave()
        fanin_algo = min(fanin_probe, key=fanin_probe.get)  # the same numbers on every rank: the same choice
        rank_log["probe_choice"] = fanin_algo
        del probe_src
    elif world > 1:
        wd.enter(f"first fan-in: {fanin_algo}", retryable=comm is not None)
        fanin.fan_in_sum(torch.zeros((frames_per_chunk, NFFT), dtype=torch.float32, device="cuda"), rs_out[0], algo=fanin_algo, recv=recv, comm=comm)
        torch.cuda.synchronize()
        wd.leave()

    def step(record: bool):
        # (no reset between steps: the stream simply continues, the FIR history of a step's first frame is the previous step's tail)
        if record:
            for c in range(nchunks):
                ev[c] = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev_all.extend(ev)
            step_ev.append((torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)))
            step_ev[-1][0].record()
        if len(mine) > 1 and not multi:  # a channel stream must not overwrite a slice the previous step's fold still reads
            for s in streams:
                s.wait_stream(main_stream)
        for c in range(nchunks):
            fr = slice(c * frames_per_chunk, (c + 1) * frames_per_chunk)
            if multi:
                if world > 1 and c >= 2:
                    main_stream.wait_event(fan_done[c & 1])  # the fan-in that read this slab two launches ago
                dst = sum_out[fr] if world == 1 else acc[c & 1]
                if record:
                    ev[c][0].record()
                chain_process_multi(chains, [x_[c * chunk:(c + 1) * chunk] for x_ in xs], want_outs=False, sum_out=dst)
                if record:
                    ev[c][1].record()
                if world > 1:
                    fan_stream.wait_stream(main_stream)
                    with torch.cuda.stream(fan_stream):
                        fanin.fan_in_sum(dst, rs_out[c], algo=fanin_algo, recv=recv, comm=comm)
                        fan_done[c & 1].record()
                continue
            for i, ch in enumerate(chains):  # every channel on its own stream
                with torch.cuda.stream(streams[i]):
                    if record and i == 0:
                        ev[c][0].record()  # on the stream the kernel is launched on
                    ch.process_bulk(xs[i][c * chunk:(c + 1) * chunk], outs[i][fr])
                    if record and i == 0:
                        ev[c][1].record()
                    if len(mine) > 1:
                        done[c & 1][i].record()
            if not combine:
                continue
            if len(mine) > 1:  # local part of math::Add over the channels: ONE n-ary fold (left fold order, Math.hpp:100-107)
                for e in done[c & 1]:
                    main_stream.wait_event(e)
                if world > 1 and c >= 2:
                    main_stream.wait_event(fan_done[c & 1])  # the fan-in that read this slab two launches ago
                dst = sum_out[fr] if world == 1 else acc[c & 1]
                G.math_nary("Add", [o[fr] for o in outs], out=dst)
            else:
                dst = outs[0][fr]
            if world > 1:  # the one exchange step: this launch's partial sums cross xGMI while the next launch computes
                fan_stream.wait_stream(main_stream)
                with torch.cuda.stream(fan_stream):
                    fanin.fan_in_sum(dst, rs_out[c], algo=fanin_algo, recv=recv, comm=comm)
                    fan_done[c & 1].record()
        if world > 1:
            main_stream.wait_stream(fan_stream)
        if len(mine) > 1 and not multi:
            for s in streams:
                main_stream.wait_stream(s)
        if record:
            step_ev[-1][1].record()

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # untimed, uncounted launches until the clocks have settled under this load (then the counted warm-up, then the timed steps)
    wd.enter("pre-warm + warm-up steps", seconds=max(args.fanin_timeout, 1.0) * 2)
    prewarm_steps, t_pre = 0, time.perf_counter()
    while (time.perf_counter() - t_pre) * 1e3 < args.prewarm_ms:
        step(False)
        torch.cuda.synchronize()
        prewarm_steps += 1
        if world > 1:  # every rank runs the same number of steps: the slowest rank's clock decides
            more = torch.tensor([1 if (time.perf_counter() - t_pre) * 1e3 < args.prewarm_ms else 0], dtype=torch.int32, device="cuda")
            dist.all_reduce(more, op=dist.ReduceOp.MAX)
            if int(more.item()) == 0:
                break
    prewarm_ms = (time.perf_counter() - t_pre) * 1e3
    for _ in range(args.warmup):
        step(False)
    fence()
    wd.enter("timed steps", seconds=max(args.fanin_timeout, 1.0) * 2 + 0.05 * args.steps)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
        # event times are read after the timed region
    fence()
    dt = time.perf_counter() - t0
    wd.leave()
    kernel_ms = [a.elapsed_time(b_) for a, b_ in ev_all]  # every timed launch of this rank's first channel
    step_ms = sorted(a.elapsed_time(b_) for a, b_ in step_ev)
    median_step_ms = step_ms[len(step_ms) // 2] if len(step_ms) % 2 else 0.5 * (step_ms[len(step_ms) // 2 - 1] + step_ms[len(step_ms) // 2])
    if world > 1:  # per-rank view of the timed region, for the record of a first run nobody watches
        rank_log.update({"launch_ms_mean": round(sum(kernel_ms) / len(kernel_ms), 4), "step_ms_median": round(median_step_ms, 4), "wall_s": round(dt, 4)})
        print(f"[bench] rank {rank}: {json.dumps(rank_log)}", file=sys.stderr, flush=True)
        gathered = [None] * world
        dist.all_gather_object(gathered, rank_log)
    else:
        gathered = None
    tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    # ---- self-check (outside the timed region): sampled frames of the last step against the float64 oracle on the same input
    verify = None
    if not args.no_verify:
        O = _oracle()
        errs = []
        lastf = n // NFFT - 1
        cand = sorted({0, 1, 255, 256, frames_per_chunk - 1, frames_per_chunk, min(lastf, 3 * frames_per_chunk + 4097), lastf} & set(range(lastf + 1)))
        if multi:  # only the sum exists: more of its frames
            cand = sorted(set(cand) | ({2, 3, 127, 257, 511, 512, lastf // 2, lastf - 1} & set(range(lastf + 1))))
        if not combine or world == 1:  # per-channel spectra of this rank (and, for the one-GPU graph, their sum)
            for i in range(len(outs)):
                for f in (cand if i == 0 else cand[:2]):
                    errs.append(_rel_err(outs[i][f].cpu().numpy(), oracle_frame(O, taps, xs[i], f)))
            if combine:
                for f in (cand if multi else cand[:3]):
                    truth = sum(oracle_frame(O, taps, xs[i], f) for i in range(len(mine)))
                    errs.append(_rel_err(sum_out[f].cpu().numpy(), truth))
        elif rank == 0:  # the reduced shard of rank 0: regenerate every channel's stream (deterministic by seed), sum the oracle spectra
            per = frames_per_chunk // world
            picks = sorted({(0, 0), (0, 1), (0, per - 1), (nchunks - 1, per // 2)})
            truth = {p: 0.0 for p in picks}
            for c in range(n_channels):
                xc = xs[mine.index(c)] if c in mine else G.synth_c32(n, seed=42 + c)
                for (ck, j) in picks:
                    truth[(ck, j)] = truth[(ck, j)] + oracle_frame(O, taps, xc, ck * frames_per_chunk + j)  # rank 0 owns frames [0, per) of every launch
                del xc
            for (ck, j) in picks:
                errs.append(_rel_err(rs_out[ck, j].cpu().numpy(), truth[(ck, j)]))
        if errs:
            verify = {"verified_frames": len(errs), "max_rel_err": float(f"{max(errs):.3e}"), "tolerance": PARITY_TOL,
                      "metric": "max |gpu - f64 oracle| / max(|oracle|, rms(oracle)) per frame (tests/test_gpu_parity.py::_rel)"}

    rc = 0
    if rank == 0:
        total_samples = float(n) * n_channels * args.steps
        value = total_samples / dt / 1e6
        launch_ms = sum(kernel_ms) / len(kernel_ms)
        achieved = chunk * ALGO_BYTES_PER_SAMPLE / (launch_ms * 1e-3) / 1e9
        algo = chains[0].algo
        committed = None  # HBM bytes per launch from the committed PMC profile of the same kernel + launch size (profiles/); this run does not measure it
        try:
            prof = json.load(open(os.path.join(ROOT, "profiles", "latest_pmc.json")))
            if prof.get("kernel") == KERNEL_SYMBOLS.get(algo) and prof.get("samples_per_launch"):
                # the PMC passes profile one 2^28-sample launch; traffic is proportional to the frames a launch covers (every frame is read and written once)
                committed = int(round(prof["hbm_bytes_per_launch"] * (chunk / prof["samples_per_launch"])))
        except Exception:
            pass
        per_gpu = len(mine)
        graph = (f"; {n_channels}-channel graph: {per_gpu} channel(s) per GPU " + ("in ONE launch, math::Add fold in its registers" if multi else "on own streams, math::Add fold on device")
                 + (f", {('RCCL ' + fanin_algo + (' through gr4hip_fanin_*' if comm is not None else ' through torch.distributed')) if args.dist_backend == 'nccl' else args.dist_backend + ' all_reduce (functional check only)'} fan-in per launch (configs[4])" if world > 1 else " (one GPU, no collective)")) if combine else ""
        res = {
            "metric": "Msamples/s through 256-tap cplx FIR->8192-pt FFT chain; %HBM roofline @1/2/4/8 GPU",
            "value": round(value, 3), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "median_ms_per_step": round(median_step_ms, 4),
            "value_at_median_step": round(float(n) * n_channels / (median_step_ms * 1e-3) / 1e6, 3),  # the median over the timed steps (HIP events on the launch stream)
            "prewarm_ms": round(prewarm_ms, 1), "prewarm_steps": prewarm_steps, "higher_is_better": True, "scaling": "strong" if combine else "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "arithmetic": "f32 (float32 butterflies and twiddles; the 255-sample correction FIR on three-term bf16 products, 24 bits; marked frames again in float64)",
            "config": {"workload": f"complex<float> {NTAPS}-tap FIR -> {NFFT}-pt FFT -> mag2, 2^{args.log2_samples}-sample stream per channel "
                                   f"(BASELINE.json configs[{4 if combine else 1}]), rectangular window, {nchunks} launch(es) of 2^{log2_chunk} samples per channel" + graph,
                       "chain_algo": KERNEL_SYMBOLS.get(algo, str(algo)), "channels": n_channels,
                       "parallelism": f"{n_channels} independent channel(s), {per_gpu} per GPU",
                       "dynamic_range_guard": ["strict: every frame measured and judged inside the kernel, the marked frames evaluated again in the time domain by a launch enqueued behind it; no host wait (library default)",
                                               "deferred", "off"][args.guard_mode]},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": None, "traffic_from_committed_profile": committed, "kernel": KERNEL_SYMBOLS.get(algo, str(algo)),
                         "algorithmic_bytes_per_launch": chunk * ALGO_BYTES_PER_SAMPLE, "avg_launch_ms": round(launch_ms, 4), "timed_launches": len(kernel_ms),
                         "median_launch_ms": round(sorted(kernel_ms)[len(kernel_ms) // 2], 4),
                         "frac_of_measured_copy_rate": round(achieved / 6290.0, 4),  # 6.29 TB/s: what a float4 copy reaches on this part (MI355X_MICROARCH.md)
                         "whole_job_frac_per_gpu": round(value * 1e6 * ALGO_BYTES_PER_SAMPLE / world / 1e9 / HBM_PEAK_GBS, 4)},
        }
        if world > 1:  # SURVEY.md 8(e) / DESIGN.md 5: what the fan-in allows at the nominal link rate
            # after the local fold a GPU holds ONE partial-sum stream for its per_gpu channels; the reduce_scatter sends (N-1)/N of it
            # out, 1/N to each peer over that peer's link: 4 B x (N-1)/N per frame bin = per per_gpu input samples
            egress = 4.0 * (world - 1) / world / per_gpu
            res["fanin"] = {"collective": (fanin_algo + "(f32)" if args.dist_backend == "nccl" else f"{fanin_algo} on {args.dist_backend}, staged through the host (functional check)"),
                            "probe_seconds_per_launch": ({k: (None if v == float("inf") else round(v, 6)) for k, v in fanin_probe.items()} if fanin_probe else None),
                            "xgmi_egress_bytes_per_input_sample": round(egress, 4),
                            "xgmi_ceiling_msamples": round(world * (world - 1) * XGMI_LINK_GBS * 1e9 / egress / 1e6, 1),
                            "note": "ceiling = N GPUs x (N-1) links x 153 GB/s nominal per direction / egress bytes per input sample; the collective of launch c overlaps the transforms of launch c+1",
                            "implementation": "gr4hip_fanin_* (the library's own RCCL communicator)" if comm is not None else ("torch.distributed (" + args.dist_backend + ")"),
                            "retried_after_watchdog": os.environ.get("GR4HIP_BENCH_RETRY") == "1", "per_rank": gathered}
        if verify:
            res["verify"] = verify
            if not (verify["max_rel_err"] <= PARITY_TOL):
                rc = 3
        if world == 1 and not combine and not args.no_hann_row:
            # "rectangular for the headline run and Hann as a second row" -- the FFT block's default window (blocks/fourier/.../fft.hpp:99); same
            # stream, same taps, same launch size; three transforms per frame instead of two (DESIGN.md 3.1 "Any window")
            try:
                hann = G.Chain(taps, NFFT, "Hann", 0)
                if args.guard_mode:
                    hann.set_guard_mode(args.guard_mode)
                hout = outs[0]
                hsteps = max(3, min(args.steps, 12))
                for _ in range(2):
                    for c in range(nchunks):
                        hann.process_bulk(xs[0][c * chunk:(c + 1) * chunk], hout[c * frames_per_chunk:(c + 1) * frames_per_chunk])
                torch.cuda.synchronize()
                hev = []
                for _ in range(hsteps):
                    a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a_.record()
                    for c in range(nchunks):
                        hann.process_bulk(xs[0][c * chunk:(c + 1) * chunk], hout[c * frames_per_chunk:(c + 1) * frames_per_chunk])
                    b_.record()
                    hev.append((a_, b_))
                torch.cuda.synchronize()
                hms = sorted(a_.elapsed_time(b_) for a_, b_ in hev)
                hmed = hms[len(hms) // 2]
                row = {"window": "Hann", "value": round(n / (hmed * 1e-3) / 1e6, 3), "unit": "Msamples/s", "median_ms_per_step": round(hmed, 4), "steps": hsteps,
                       "frac": round(n * ALGO_BYTES_PER_SAMPLE / (hmed * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
                if not args.no_verify:
                    O = _oracle()
                    herr = []
                    for f in (1, 255, n // NFFT - 1):
                        prev = xs[0][(f - 1) * NFFT:f * NFFT].cpu().numpy()
                        cur = xs[0][f * NFFT:(f + 1) * NFFT].cpu().numpy()
                        truth = O.chain(taps, np.concatenate([prev, cur]), NFFT, 3, truth=True)[0].reshape(-1, NFFT)[1]
                        herr.append(_rel_err(hout[f].cpu().numpy(), truth))
                    row["verify_max_rel_err"] = float(f"{max(herr):.3e}")
                    if not (max(herr) <= PARITY_TOL):
                        rc = 3
                res["hann_second_row"] = row
                del hann
            except Exception as e:  # never at the price of the headline line
                res["hann_second_row"] = {"error": str(e)[:200]}
        if world == 1 and not combine and not args.no_hann_row and not args.guard_mode:
            # the chain under a narrow filter: the same samples through the same Hamming windowed-sinc at cut-off 0.005 fs (256 taps, passes 1 % of the stream's power --
            # VERDICT r05's "ordinary channel filter": every frame marked until round 6, 31 Gsamples/s then).
            # "narrow": white noise alone (a second synthetic stream, no tone) -- round 6's guard judges the frames on what the error depends on (R4 = 6 .. 13 of 20: not
            # marked): the fused launch alone.
            # "in_stream": the bench's own stream, whose tone at 0.1 fs at the noise's level the filter removes -- every frame marked (the line statistic), chain_td16_kernel,
            # enqueued behind the fused launch, evaluates them again in the time domain (22-bit products on the f16 matrix pipe, stored where it agrees with the fused result;
            # float64 for the frames that leaves); "settled": the same call once the host has seen the measurement -- a stream only moves to the time-domain kernel pair when
            # more than a tenth of its frames end in float64, which this one's do not, so it stays (moved_to_time_domain false).  None waits for the host.
            try:
                ng = min(n, 1 << 27)
                gt = w.astype(np.float64) * 0.01 * np.sinc(0.01 * (k - (NTAPS - 1) / 2.0))
                gt = (gt / gt.sum()).astype(np.float32)
                gx, go = xs[0][:ng], outs[0][:ng // NFFT]
                gch = G.Chain(gt, NFFT, "None", 0)

                def _timed(reset, xin):
                    ts_ = []
                    for _ in range(5):
                        if reset:
                            gch.reset()
                        torch.cuda.synchronize()
                        a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        a_.record(); gch.process_bulk(xin, go); b_.record()
                        torch.cuda.synchronize()
                        ts_.append(a_.elapsed_time(b_))
                    return sorted(ts_)[len(ts_) // 2]

                def _check(xin, key):
                    nonlocal rc
                    f = 77
                    truth = _oracle().chain(gt, xin[(f - 1) * NFFT:(f + 1) * NFFT].cpu().numpy(), NFFT, 0, truth=True)[0].reshape(-1, NFFT)[1]
                    row[key] = float(f"{_rel_err(go[f].cpu().numpy(), truth):.3e}")
                    if not (row[key] <= PARITY_TOL):
                        rc = 3
                row = {"taps": "256-tap Hamming low-pass, cut-off 0.005 fs (passes 1 % of the stream's power)", "samples": ng}
                gx0 = G.synth_c32(ng, seed=43, tone_amp=0.0)
                t_nb = _timed(True, gx0)
                row["narrow_msamples"] = round(ng / (t_nb * 1e-3) / 1e6, 1)
                row["narrow_marked_fraction"] = round(float(gch.last_guard_fractions()[0]), 5)
                if not args.no_verify:
                    _check(gx0, "narrow_verify_max_rel_err")
                del gx0
                gx2 = gx
                t_in = _timed(True, gx2)
                row["in_stream_msamples"] = round(ng / (t_in * 1e-3) / 1e6, 1)
                row["in_stream_note"] = "the bench stream's tone (0.1 fs, at the noise's level) removed by the filter: fused launch + the second evaluation of every frame behind it (chain_td16_kernel on the f16 matrix pipe, float64 for what that leaves), one stream, no host wait"
                if not args.no_verify:
                    _check(gx2, "in_stream_verify_max_rel_err")
                gch.last_power_ratio()          # (the measurement has arrived: a stream that is to move moves with the next call)
                gch.process_bulk(gx2, go)
                row["ratio"], row["moved_to_time_domain"] = [round(float(gch.last_power_ratio()[0]), 5), bool(gch.last_power_ratio()[1])]
                row["marked_fraction"], row["float64_fraction"] = [round(float(v), 5) for v in gch.last_guard_fractions()]
                t_td = _timed(False, gx2)
                row["settled_msamples"] = round(ng / (t_td * 1e-3) / 1e6, 1)
                res["guard_tripped_row"] = row
                del gch, gx2
            except Exception as e:  # never at the price of the headline line
                res["guard_tripped_row"] = {"error": str(e)[:200]}
        if world == 1 and not combine and not args.no_live_traffic and os.environ.get("GR4HIP_BENCH_CHILD") != "1":
            try:
                tb, how = live_traffic(KERNEL_SYMBOLS.get(algo, ""), args.log2_samples, log2_chunk)
                res["roofline"]["traffic"] = tb
                res["roofline"]["traffic_how"] = how
                if tb:
                    res["roofline"]["traffic_over_algorithmic"] = round(tb / (chunk * ALGO_BYTES_PER_SAMPLE), 4)
            except Exception as e:  # never at the price of the headline line
                res["roofline"]["traffic_how"] = f"not collected: {str(e)[:160]}"
        if world == 1 and not combine and not args.no_secondary and os.environ.get("GR4HIP_BENCH_CHILD") != "1":
            try:
                res["secondary_configs"] = secondary_configs(G, not args.no_verify)
                if any(isinstance(v, dict) and v.get("verify_max_rel_err", 0.0) > PARITY_TOL for v in res["secondary_configs"].values()):
                    rc = 3
            except Exception as e:  # never at the price of the headline line
                res["secondary_configs"] = {"error": str(e)[:200]}
        if world == 1 and not combine and not args.no_secondary and os.environ.get("GR4HIP_BENCH_CHILD") != "1":
            # the PCIe-inclusive rate (never `value`): the same chain host-fed through the C++ engine (gnuradio4_amd/host: Graph::connect, hip::plan, scheduler::Simple on one host thread)
            # between page-locked edges of the REFERENCE'S DEFAULT SIZE, 65 536 items (Graph.hpp:102) -- VERDICT r05 item 5 -- and of 2^24 items (what the link gives)
            try:
                import re as _re
                import subprocess
                exe = os.path.join(ROOT, "build", "host", "bench_host_feed")
                row = {"workload": "DmaSource -> fir_filter<complex<float>> 256 taps -> PowerSpectrum 8192 -> NullSink, 2^29 samples, page-locked edges, one host thread"}
                for key, edge in (("value", "16"), ("edges_2^24_msamples", "24")):
                    out = subprocess.run([exe, "29", str(NFFT), str(NTAPS), "dma", edge], capture_output=True, text=True, timeout=120).stdout
                    m = _re.search(r"= ([0-9.]+) Msamples/s", out)
                    row[key] = float(m.group(1)) if m else None
                row["unit"], row["edge_items"] = "Msamples/s", 65536
                res["host_feed_row"] = row
            except Exception as e:  # never at the price of the headline line
                res["host_feed_row"] = {"error": str(e)[:200]}
        if world == 1 and not combine and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline()
        if world == 1 and not combine and not args.no_graph8 and os.environ.get("GR4HIP_BENCH_CHILD") != "1":
            # the N > 1 runs shard the 8-channel graph of configs[4] (total work fixed); its 1-GPU point -- all eight channels on this GPU -- is measured here,
            # after the headline (its own process, shorter streams: the rate is a steady-state one), so that the scaling curve has its origin in the same file
            del xs, outs
            torch.cuda.empty_cache()
            import subprocess
            try:
                child = subprocess.run([sys.executable, os.path.abspath(__file__), "--channels", "8", "--log2-samples", "28", "--steps", "12", "--warmup", "4", "--no-cpu-baseline"],
                                       capture_output=True, text=True, timeout=600, env={**os.environ, "GR4HIP_BENCH_CHILD": "1"})
                g8 = json.loads(child.stdout.strip().splitlines()[-1])
                res["eight_channel_graph_on_one_gpu"] = {"value": g8["value"], "unit": g8["unit"], "ms_per_step": g8["ms_per_step"], "samples_per_channel_and_step": 1 << 28,
                                                         "steps": g8["steps"], "verify": g8.get("verify"), "workload": g8["config"]["workload"]}
            except Exception as e:  # never at the price of the headline line
                res["eight_channel_graph_on_one_gpu"] = {"error": str(e)[:200]}
        detail = args.detail_file
        if detail is None:
            detail = "" if os.environ.get("GR4HIP_BENCH_CHILD") == "1" else os.path.join(ROOT, "gpurun_out", "bench_detail.json")
        print(json.dumps(_compact_line(res, detail)), flush=True)
    if world > 1:
        dist.destroy_process_group()
    sys.exit(rc)


if __name__ == "__main__":
    main()
