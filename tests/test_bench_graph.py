"""bench.py end to end on the GPU box: the N = 1 headline line with its oracle self-check, the 8-channel graph (BASELINE.json
configs[4]) on one GPU, and the N > 1 code path executed for real -- two ranks (4 channels each) sharing the one
GPU of the box, gloo carrying the fan-in -- with rank 0's shard of the channel sum checked against the float64 oracle inside
bench.py (exit code 3 if any sampled frame is off by more than 1e-5)."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(cmd, timeout=900):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, f"{' '.join(cmd)}\nrc={p.returncode}\n{p.stdout[-3000:]}\n{p.stderr[-3000:]}"
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    assert len(lines[0]) < 2000, len(lines[0])  # the driver's record keeps 2 000 characters of tail: the whole line fits
    r = json.loads(lines[0])
    if "detail_file" in r:  # the long form (rows with their workload text and timing notes): the tests below read both through one dict, the printed line's keys on top
        d = json.load(open(os.path.join(ROOT, r["detail_file"])))
        for k, v in d.items():
            if isinstance(v, dict) and isinstance(r.get(k), dict):
                r[k] = {**v, **r[k]}
            else:
                r.setdefault(k, v)
    return r


def test_bench_single_channel_self_check():
    r = _run([sys.executable, "bench.py", "--steps", "2", "--warmup", "1", "--log2-samples", "26", "--log2-chunk", "24", "--no-cpu-baseline", "--no-graph8", "--no-live-traffic", "--no-secondary"])
    assert r["n_gpus"] == 1 and r["config"]["channels"] == 1 and r["scaling"] == "weak"
    assert r["roofline"]["kernel"] == "gr4::chain_fd_kernel<0, 13>"
    v = r["verify"]
    assert v["verified_frames"] >= 6 and v["max_rel_err"] <= 1e-5


def test_bench_single_channel_line_carries_the_one_gpu_point_of_the_graph():
    """the N = 1 line also reports the 8-channel graph on this one GPU (own process, after the headline): the origin of the strong-scaling curve"""
    r = _run([sys.executable, "bench.py", "--steps", "2", "--warmup", "1", "--log2-samples", "26", "--log2-chunk", "24", "--no-cpu-baseline", "--no-live-traffic", "--no-secondary"])
    g8 = r["eight_channel_graph_on_one_gpu"]
    assert "error" not in g8 and g8["value"] > 0 and g8["verify"]["max_rel_err"] <= 1e-5 and r["config"]["channels"] == 1


def test_bench_eight_channel_graph_one_gpu():
    r = _run([sys.executable, "bench.py", "--channels", "8", "--steps", "2", "--warmup", "1", "--log2-samples", "24"])
    assert r["config"]["channels"] == 8 and r["scaling"] == "strong" and "cpu_baseline" not in r
    assert r["verify"]["verified_frames"] >= 8 and r["verify"]["max_rel_err"] <= 1e-5


@pytest.mark.parametrize("algo", ["auto", "reduce_scatter", "all_to_all"])
def test_bench_two_ranks_share_the_gpu_gloo_fanin(algo):
    """auto: both fan-in algorithms are timed before the warm-up and the faster one runs (the choice is the same on every rank)"""
    r = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
              "--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--dist-backend", "gloo", "--log2-samples", "24",
              "--steps", "2", "--warmup", "1", "--fanin-algo", algo])
    assert r["n_gpus"] == 2 and r["config"]["channels"] == 8 and r["scaling"] == "strong"
    assert "4 per GPU" in r["config"]["parallelism"]
    assert r["fanin"]["xgmi_ceiling_msamples"] > 0
    if algo == "auto":
        assert set(r["fanin"]["probe_seconds_per_launch"]) == {"reduce_scatter", "all_to_all"} and r["fanin"]["collective"].split(" ")[0] in ("reduce_scatter", "all_to_all")
    else:
        assert r["fanin"]["collective"].startswith(algo)
    v = r["verify"]  # rank 0's shard of the 8-channel sum vs the oracle's sum of the eight channels' spectra
    assert v["verified_frames"] >= 3 and v["max_rel_err"] <= 1e-5


def test_bench_stalled_rank_gives_a_diagnostic_not_a_hang():
    """a rank that never reaches its first collective (injected: GR4HIP_BENCH_STALL_RANK) must not hang the run: every waiting phase has a wall-clock deadline,
    the ranks say where they are, rank 0 prints the JSON line with an "error" key, and the launcher exits non-zero -- well inside a minute"""
    import time
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", GR4HIP_BENCH_STALL_RANK="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           "bench.py", "--gpus", "2", "--dist-backend", "gloo", "--log2-samples", "24", "--steps", "2", "--warmup", "1", "--fanin-timeout", "12"]
    t0 = time.time()
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=240)
    took = time.time() - t0
    assert p.returncode != 0, p.stdout[-2000:]
    assert took < 60, took
    assert "[bench][watchdog] rank 0 of 2: no progress in phase" in p.stderr, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["error"].startswith("rank 0 of 2: no progress in phase"), p.stdout[-2000:]


def test_bench_line_has_the_median_the_prewarm_the_hann_row_and_the_secondary_configs():
    r = _run([sys.executable, "bench.py", "--steps", "5", "--warmup", "1", "--log2-samples", "26", "--log2-chunk", "24", "--no-cpu-baseline", "--no-graph8"])
    assert r["prewarm_ms"] >= 40 and r["prewarm_steps"] >= 1 and r["median_ms_per_step"] > 0 and r["value_at_median_step"] > 0
    assert r["roofline"]["timed_launches"] == 5 * 4
    assert r["hann_msamples"] > 0 and r["configs2_msamples"] > 0 and r["configs3_msamples"] > 0 and r["guard_settled_msamples"] > 0  # the rows as top-level numbers of the printed line
    assert r["host_feed_msamples"] > 0  # the chain host-fed through the C++ engine at the reference's default 65 536-item edges (PCIe-inclusive: never `value`)
    h = r["hann_second_row"]
    assert h["window"] == "Hann" and h["value"] > 0 and 0 < h["frac"] < 1 and h["verify_max_rel_err"] <= 1e-5
    sc = r["secondary_configs"]  # BASELINE.json configs[2] and configs[3] beside the headline, each checked against the oracle
    for k in ("configs[2]", "configs[3]"):
        assert sc[k]["value"] > 0 and 0 < sc[k]["hbm_frac"] < 1 and sc[k]["verify_max_rel_err"] <= 1e-5, sc[k]
    # roofline.traffic: measured live where rocprofv3 is there (two counter-only child passes), null otherwise.  At this test's small launches (2^24 samples) the per-workgroup
    # tables (H, twiddles, taps: ~25 MB over 256 workgroups) are 11 % on top of the 201 MB of samples; at the headline's 2^30-sample launch the ratio is 1.02
    rf = r["roofline"]
    import shutil
    if shutil.which("rocprofv3"):
        assert rf["traffic"] and 0.95 <= rf["traffic_over_algorithmic"] <= 1.20, (rf.get("traffic"), rf.get("traffic_how"))
    else:
        assert rf["traffic"] is None
