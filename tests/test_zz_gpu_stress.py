"""Concurrency, timing and hostile-ordering tests of the C-ABI (run with -m gpu on an MI355X).  This file sorts LAST on purpose: every §8 row's oracle / golden
comparison lives in the files collected before it, so a timing-dependent failure here can never hide one of them again.

The stream rule (include/gr4hip.h "LIFECYCLE CALLS ARE STREAM-ORDERED WITH THE DATA", csrc/common.hpp): *_reset / *_set_taps / set_prologue never touch device memory -- the
next *_process call applies them on ITS stream.  Every test below forces the timing that round 5's NULL-stream hipMemset lost: a single-lane spin kernel holds the caller's
(non-blocking) stream for >= 100 ms, the host queues process -> lifecycle call -> process behind it and is back long before any of it has run, and BOTH spans are then
compared with the CPU oracle -- not with another device run."""
import time

import numpy as np
import pytest

import oracle_lib as O

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

TOL = 1e-5
BUSY_MS = 100.0


def _rel(got, truth):
    """the parity contract's formula (include/gr4hip.h; tests/test_gpu_parity.py::_rel)"""
    got = np.asarray(got).astype(np.complex128 if np.iscomplexobj(got) else np.float64).ravel()
    truth = np.asarray(truth).ravel()
    rms = np.sqrt(np.mean(np.abs(truth) ** 2))
    return float(np.max(np.abs(got - truth) / np.maximum(np.abs(truth), rms if rms > 0 else 1.0)))


@pytest.fixture(scope="module")
def G():
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    import gnuradio4_amd as G
    G.capi.lib()
    return G


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


_cycles_per_ms = None


def _busy(ms=BUSY_MS):
    """hold the CURRENT stream for >= ms with one spinning workgroup (torch.cuda._sleep, calibrated once with events): the rest of the GPU stays free, the host runs ahead"""
    global _cycles_per_ms
    if _cycles_per_ms is None:
        torch.cuda.synchronize()
        rate = 0.0
        for cycles in (2_000_000, 20_000_000):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); torch.cuda._sleep(cycles); e1.record(); e1.synchronize()
            rate = cycles / max(e0.elapsed_time(e1), 1e-3)
        _cycles_per_ms = rate
    torch.cuda._sleep(int(_cycles_per_ms * ms))


class _Hostile:
    """with _Hostile() as h: ... queue work ...; h.queued() right after the last call asserts that the stream is still held (the calls did not wait)"""

    def __enter__(self):
        torch.cuda.synchronize()
        self.st = torch.cuda.Stream()
        self.ctx = torch.cuda.stream(self.st)
        self.ctx.__enter__()
        self.t0 = time.perf_counter()
        _busy()
        return self

    def queued(self, must_not_have_waited=True):
        ev = torch.cuda.Event()
        ev.record()
        if must_not_have_waited:
            assert not ev.query(), "the calls waited for the stream (%.1f ms on the host)" % (1e3 * (time.perf_counter() - self.t0))

    def __exit__(self, *a):
        self.ctx.__exit__(*a)
        self.st.synchronize()
        torch.cuda.synchronize()
        return False


def _np(t):
    return t.cpu().numpy()


# ---------------------------------------------------------------------------------------------------------------- reset behind a held stream, per handle type
@pytest.mark.parametrize("ntaps,n,cplx,decim", [(31, 8192, False, 1), (200, 1 << 17, False, 1), (1000, 1 << 17, False, 1), (256, 1 << 17, True, 1), (64, 1 << 16, True, 1),
                                                (1024, 1 << 18, False, 8), (80, 1 << 18, True, 4), (48, 131070, False, 5)])
def test_fir_reset_is_ordered_behind_the_launch_in_flight(G, ntaps, n, cplx, decim):
    """fir_filter (register-window, f16 matrix-pipe, sliced, complex, decimating kernels): process(x1) | reset | process(x2) queued behind a held stream == two fresh filters"""
    b = O.design_taps_hamming_lowpass(ntaps, 0.1 / decim)
    x1 = (O.signal_c32 if cplx else O.signal_f32)(11, n)
    x2 = (O.signal_c32 if cplx else O.signal_f32)(12, n // 2)
    t1, t2 = O.fir(b, x1)[0][::decim], O.fir(b, x2)[0][::decim]
    f = G.fir_filter(b, torch.complex64 if cplx else torch.float32, decimate=decim)
    d1, d2 = dev(x1), dev(x2)
    with _Hostile() as h:
        y1 = f.process_bulk(d1)
        f.reset()
        y2 = f.process_bulk(d2)
        h.queued()
    assert _rel(_np(y1), t1) <= TOL and _rel(_np(y2), t2) <= TOL


def test_fir_set_taps_is_ordered_behind_the_launch_in_flight(G):
    """settingsChanged (time_domain_filter.hpp:38-42) in mid-stream: the launch in flight keeps ITS taps, the next one sees the new ones and the carried history; every
    kernel family's tables (register window, f16 fragments, slices).  The first call after the change may wait for the stream (pageable upload, stream-ordered)"""
    for ntaps, n in ((31, 8192), (200, 1 << 17), (600, 1 << 17)):
        b1, b2 = O.design_taps_hamming_lowpass(ntaps, 0.1), O.design_taps_hamming_lowpass(ntaps, 0.23)
        x = O.signal_f32(21, 2 * n)
        t1, hist = O.fir(b1, x[:n])
        t2, _ = O.fir(b2, x[n:], hist=hist)
        f = G.fir_filter(b1, torch.float32)
        d = dev(x)
        with _Hostile() as h:
            y1 = f.process_bulk(d[:n])
            f.settings_changed(b2)
            y2 = f.process_bulk(d[n:])
            h.queued(must_not_have_waited=False)
        assert _rel(_np(y1), t1) <= TOL and _rel(_np(y2), t2) <= TOL, ntaps


def test_fir_prologue_change_rescales_the_history_on_the_stream(G):
    """a gain prologue that moves into the taps in mid-stream: the stored history is brought to the new convention by a kernel ON THE CALL'S STREAM (until round 5: two
    device-wide synchronisations and a NULL-stream launch).  Truth: the filter over [x1, 2 x2] -- its memory keeps what the outgoing prologue made of x1"""
    ntaps, n = 200, 1 << 17
    b = O.design_taps_hamming_lowpass(ntaps, 0.1)
    x = O.signal_f32(23, 2 * n)
    xs = x.copy(); xs[n:] *= np.float32(2.0)
    truth = O.fir(b, xs)[0]
    f = G.fir_filter(b, torch.float32)
    d = dev(x)
    with _Hostile() as h:
        y1 = f.process_bulk(d[:n])
        f.set_prologue(G.Merged(torch.float32, [("Multiply", 2.0)]))
        y2 = f.process_bulk(d[n:])
        h.queued(must_not_have_waited=False)
    assert _rel(np.concatenate([_np(y1), _np(y2)]), truth) <= TOL


@pytest.mark.parametrize("interp,ntaps,cplx", [(4, 64, False), (3, 50, True), (8, 256, False)])
def test_fir_interpolator_reset_is_ordered(G, interp, ntaps, cplx):
    b = O.design_taps_hamming_lowpass(ntaps, 0.4 / interp)
    n = 1 << 15
    x1, x2 = (O.signal_c32 if cplx else O.signal_f32)(31, n), (O.signal_c32 if cplx else O.signal_f32)(32, n)
    t1, t2 = O.fir_interp(b, x1, interp)[0], O.fir_interp(b, x2, interp)[0]
    f = G.fir_interpolator(b, interp, torch.complex64 if cplx else torch.float32)
    d1, d2 = dev(x1), dev(x2)
    with _Hostile() as h:
        y1 = f.process_bulk(d1)
        f.reset()
        y2 = f.process_bulk(d2)
        h.queued()
    assert _rel(_np(y1), t1) <= TOL and _rel(_np(y2), t2) <= TOL


@pytest.mark.parametrize("order,algo", [(4, "auto"), (8, "auto"), (12, "auto"), (8, "sequential")])
def test_iir_reset_is_ordered_behind_the_launch_in_flight(G, order, algo):
    """iir cascades (2 / 4 / 8 biquads: the one-pass scan, the split cascade) and GR4HIP_IIR_SEQUENTIAL_F32: the state pair is zeroed on the call's stream"""
    bi, ai = G.blocks.design_iir(G.capi.LOWPASS, order, 0.1 if order > 8 else 0.05, float("nan"), 1.0, G.capi.BUTTERWORTH)  # (12: six biquads = the split cascade)
    sec = lambda: O.make_sections([(bb, aa) for bb, aa in zip(bi, ai)])  # (the oracle's sections carry their state: a fresh cascade per span)
    n = (1 << 20) + 77 if algo == "auto" else 50_000
    x1, x2 = O.signal_f32(41, n), O.signal_f32(42, n // 2)
    f = G.iir_filter(bi, ai)
    if algo == "sequential":
        f.set_algo(G.capi.IIR_SEQUENTIAL_F32)
    t1, t2 = O.iir_cascade(sec(), x1, O.DF_II, f64=True), O.iir_cascade(sec(), x2, O.DF_II, f64=True)
    d1, d2 = dev(x1), dev(x2)
    with _Hostile() as h:
        y1 = f.process_bulk(d1)
        f.reset()
        y2 = f.process_bulk(d2)
        h.queued()
    if algo == "sequential":  # the reference's float32 arithmetic in its order: bit for bit
        assert np.array_equal(_np(y1), O.iir_cascade(sec(), x1, O.DF_II, f64=False)) and np.array_equal(_np(y2), O.iir_cascade(sec(), x2, O.DF_II, f64=False))
    else:
        assert _rel(_np(y1), t1) <= TOL and _rel(_np(y2), t2) <= TOL
    f.status()


def test_float64_blocks_reset_is_ordered(G):
    """fir_filter<double>, iir_filter<double> (csrc/f64.hip)"""
    b = O.design_taps_hamming_lowpass(100, 0.1).astype(np.float64)
    n = 1 << 16
    x1, x2 = O.signal_f32(51, n).astype(np.float64), O.signal_f32(52, n).astype(np.float64)
    t1, t2 = np.convolve(x1, b)[:n], np.convolve(x2, b)[:n]
    f = G.fir_filter(b, torch.float64)
    bi, ai = G.blocks.design_iir(G.capi.LOWPASS, 4, 0.1, float("nan"), 1.0, G.capi.BUTTERWORTH)
    from scipy.signal import sosfilt
    sos = np.concatenate([np.asarray(bi, np.float64), np.asarray(ai, np.float64)], axis=1)
    u1, u2 = sosfilt(sos, x1), sosfilt(sos, x2)
    q = G.iir_filter(bi, ai, dtype=torch.float64)
    d1, d2 = dev(x1), dev(x2)
    with _Hostile() as h:
        y1 = f.process_bulk(d1); z1 = q.process_bulk(d1)
        f.reset(); q.reset()
        y2 = f.process_bulk(d2); z2 = q.process_bulk(d2)
        h.queued()
    assert _rel(_np(y1), t1) <= 1e-12 and _rel(_np(y2), t2) <= 1e-12
    assert _rel(_np(z1), u1) <= 1e-10 and _rel(_np(z2), u2) <= 1e-10


@pytest.mark.parametrize("ntaps,N,window,wid,algo", [(256, 8192, "None", 0, "auto"), (256, 8192, "Hann", 3, "auto"), (200, 1024, "Hamming", 2, "auto"), (64, 1024, "Hann", 3, "auto"),
                                                      (256, 8192, "None", 0, "unfused"), (256, 8192, "None", 0, "time_domain")])
def test_chain_reset_is_ordered_behind_the_launch_in_flight(G, ntaps, N, window, wid, algo):
    """the merged fir_filter -> FFT -> |.|^2 chain: the fused frequency-domain kernel (round 5's red test: its carried history), the fused time-domain kernel, the kernel pair"""
    b = O.design_taps_hamming_lowpass(ntaps, 0.1)
    frames = 40 if N == 8192 else 320
    x1, x2 = O.signal_c32(61, frames * N), O.signal_c32(62, frames * N // 2)
    t1, t2 = O.chain(b, x1, N, wid, truth=True)[0], O.chain(b, x2, N, wid, truth=True)[0]
    a = {"auto": G.capi.CHAIN_AUTO, "unfused": G.capi.CHAIN_UNFUSED, "time_domain": G.capi.CHAIN_TIME_DOMAIN}[algo]
    ch = G.Chain(b, N, window, a)
    d1, d2 = dev(x1), dev(x2)
    with _Hostile() as h:
        y1 = ch.process_bulk(d1)
        ch.reset()
        y2 = ch.process_bulk(d2)
        ch.reset()
        y3 = ch.process_bulk(d1)
        h.queued()
    assert _rel(_np(y1).ravel(), t1) <= TOL and _rel(_np(y2).ravel(), t2) <= TOL
    assert torch.equal(y3, y1)


def test_fir_batched_reset_is_ordered(G):
    nch, ntaps, n = 8, 64, 1 << 15
    rng = np.random.default_rng(7)
    b = np.stack([O.design_taps_hamming_lowpass(ntaps, 0.05 + 0.04 * c) for c in range(nch)])
    x1 = rng.standard_normal((nch, n)).astype(np.float32)
    x2 = rng.standard_normal((nch, n)).astype(np.float32)
    f = G.FirBatched(b)
    d1, d2 = dev(x1), dev(x2)
    with _Hostile() as h:
        y1 = f.process_bulk(d1)
        f.reset()
        y2 = f.process_bulk(d2)
        h.queued()
    for c in range(nch):
        assert _rel(_np(y1[c]), O.fir(b[c], x1[c])[0]) <= TOL and _rel(_np(y2[c]), O.fir(b[c], x2[c])[0]) <= TOL, c


def test_rotator_recurrence_reset_is_ordered(G):
    """Rotator under the reference's float recurrence keeps _accumulated_phase in a device word: settingsChanged stores the new phase on the call's stream"""
    n = 100_000
    x = O.signal_c32(71, n)
    w1, _ = O.rotator(x, 0.3, 0.25)
    w2, ph2 = O.rotator(x[: n // 2], 0.3, -1.0)
    r = G.Rotator(phase_increment=0.3, initial_phase=0.25, algo="recurrence")
    d = dev(x)
    with _Hostile() as h:
        y1 = r.process_bulk(d)
        r.settings_changed(-1.0)
        y2 = r.process_bulk(d[: n // 2])
        h.queued()
    assert np.max(np.abs(_np(y1) - w1)) <= 1e-5 * np.max(np.abs(w1)) and np.max(np.abs(_np(y2) - w2)) <= 1e-5 * np.max(np.abs(w2))
    assert r.accumulated_phase == np.float32(ph2)


def test_merged_program_change_is_ordered(G):
    """an element-wise program that grows in mid-stream: the launch in flight runs the program it was queued with"""
    n = 1 << 20
    x = O.signal_f32(81, n)
    m = G.Merged(torch.float32, [("Add", 1.5)])
    d = dev(x)
    with _Hostile() as h:
        y1 = m.process_bulk(d)
        m.append("Multiply", 3.0)
        y2 = m.process_bulk(d)
        h.queued(must_not_have_waited=False)
    a = O.math_const(0, 8, x, 1.5)
    assert np.array_equal(_np(y1), a) and np.array_equal(_np(y2), O.math_const(2, 8, a, 3.0))


def test_decimator_and_cascade_in_one_launch_reset_is_ordered(G):
    """BASELINE configs[2]: decimate-by-8 FIR + 4 biquads (gr4hip_fir_iir_process): both handles reset behind a held stream"""
    ntaps, n = 1024, 7168 * 80
    b = O.design_taps_hamming_lowpass(ntaps, 0.05)
    bi, ai = G.blocks.design_iir(G.capi.LOWPASS, 8, 0.05, float("nan"), 1.0, G.capi.BUTTERWORTH)
    x1, x2 = O.signal_f32(91, n), O.signal_f32(92, n)
    truth = [O.iir_cascade(O.make_sections([(bb, aa) for bb, aa in zip(bi, ai)]), O.fir_decim(b, x, 8)[0].astype(np.float32), O.DF_II, f64=True) for x in (x1, x2)]
    f, q = G.fir_filter(b, torch.float32, decimate=8), G.iir_filter(bi, ai)
    f.set_guard_mode(G.capi.GUARD_DEFERRED)  # (strict waits for its own measurement on this path: a different contract, tested in test_gpu_parity.py)
    d1, d2 = dev(x1), dev(x2)
    with _Hostile() as h:
        y1 = G.blocks.fir_iir_process(f, q, d1, mode=G.capi.FIR_IIR_ONE_LAUNCH)
        f.reset(); q.reset()
        y2 = G.blocks.fir_iir_process(f, q, d2, mode=G.capi.FIR_IIR_ONE_LAUNCH)
        h.queued()
    assert _rel(_np(y1), truth[0]) <= TOL and _rel(_np(y2), truth[1]) <= TOL


# ---------------------------------------------------------------------------------------------------------------- handles on concurrent streams
def test_independent_handles_on_concurrent_streams(G):
    """one HIP stream per fused chain (SURVEY 8b "Threading"): handles driven from different (non-blocking) streams at the same time do not share mutable state, and a reset
    queued while the handle's previous launch is still in flight lands behind it.  Round 5's driver run failed here (outs[2] row 0: filtered with the previous stream's tail)."""
    N, frames = 8192, 300
    b = O.design_taps_hamming_lowpass(256, 0.1)
    xs = [G.synth_c32(frames * N, seed=60 + i) for i in range(3)]
    bi, ai = G.blocks.design_iir(G.capi.LOWPASS, 8, 0.05, float("nan"), 1.0, G.capi.BUTTERWORTH)
    xr = G.synth_f32(1 << 22, seed=70)
    ref = [G.Chain(b, N, w).process_bulk(x) for x, w in zip(xs, ("None", "Hann", "None"))]
    ref_iir = G.iir_filter(bi, ai).process_bulk(xr)
    torch.cuda.synchronize()
    # the references themselves against the oracle (first frames: the part a stale history would corrupt)
    for i, wid in enumerate((0, 3, 0)):
        t, _ = O.chain(b, _np(xs[i][: 4 * N]), N, wid, truth=True)
        assert _rel(_np(ref[i][:4]).ravel(), t) <= TOL
    streams = [torch.cuda.Stream() for _ in range(4)]
    chains = [G.Chain(b, N, w) for w in ("None", "Hann", "None")]
    iir = G.iir_filter(bi, ai)
    outs = [None] * 4
    for rnd in range(20):
        for rep in range(3):  # interleaved submission, nothing synchronised in between
            for i, st in enumerate(streams):
                with torch.cuda.stream(st):
                    if rnd % 2 and rep == 0:
                        _busy(5.0)
                    if i < 3:
                        chains[i].reset()
                        outs[i] = chains[i].process_bulk(xs[i])
                    else:
                        iir.reset()
                        outs[3] = iir.process_bulk(xr)
        torch.cuda.synchronize()
        for i in range(3):
            assert torch.equal(outs[i], ref[i]), (rnd, i)
        assert torch.equal(outs[3], ref_iir), rnd


def test_chain_strict_guard_does_not_wait_for_its_launch(G):
    """GR4HIP_GUARD_STRICT without the host: gr4hip_chain_process returns while its launch is still running (an event recorded behind the call has not completed
    when the call is back), two guarded chains on two streams overlap.  docs/USER_API_advanced_work.md: user code must not block in work().  (The parity half of this
    test -- every frame marked -- stays in test_gpu_parity.py::test_chain_every_frame_marked_meets_the_bar.)"""
    N, ntaps = 8192, 256
    b = O.design_taps_hamming_lowpass(ntaps, 0.05)
    frames = 1 << 14                                        # 2^27 samples: ~0.4 ms of kernel
    x = G.synth_c32(frames * N, seed=3)
    out = torch.empty(frames * N, dtype=torch.float32, device="cuda")
    ch = G.Chain(b, N, "None")
    assert ch.algo == G.capi.CHAIN_FUSED_FD
    ch.process_bulk(x, out); torch.cuda.synchronize()       # (warm: tables, first-launch costs)
    pending = 0
    for _ in range(5):
        ev = torch.cuda.Event()
        ch.process_bulk(x, out)
        ev.record()
        pending += 0 if ev.query() else 1
        torch.cuda.synchronize()
    assert pending >= 4, pending                            # the call came back before its kernels were through
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    c1, c2 = G.Chain(b, N, "None"), G.Chain(b, N, "None")
    o1, o2 = torch.empty_like(out), torch.empty_like(out)
    torch.cuda.synchronize()
    with torch.cuda.stream(s1):
        c1.process_bulk(x, o1); e1 = torch.cuda.Event(); e1.record()
    with torch.cuda.stream(s2):
        c2.process_bulk(x, o2); e2 = torch.cuda.Event(); e2.record()
    assert not e1.query() or not e2.query()
    torch.cuda.synchronize()
    assert torch.equal(o1, o2)
