"""GPU parity tests of kernel-level fusion (run with -m gpu on an MI355X): runs of per-sample blocks as ONE launch (gr4hip_ewise_*: the run-time counterpart of
the reference's Merge<A, "out", B, "in">, core/include/gnuradio-4.0/BlockMerging.hpp:126-240) and the same programs as load / store hooks of a FIR filter
(gr4hip_fir_set_prologue / _epilogue).  The checker applies the blocks one after the other on the CPU (oracle: gr4o_math_const per block, the float64 rotator,
the float64 FIR): a fused launch must give what the chain of separate blocks gives -- bit for bit for integer types and for real float ops (single IEEE
operations in program order on both sides), within 1e-5 of the float64 truth where a rotator or a filter is involved."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

_OPS = {"Add": O.ADD, "Subtract": O.SUB, "Multiply": O.MUL, "Divide": O.DIV}
_TORCH = {0: torch.uint8, 1: torch.uint16, 2: torch.uint32, 3: torch.uint64, 4: torch.int8, 5: torch.int16, 6: torch.int32, 7: torch.int64,
          8: torch.float32, 9: torch.float64, 10: torch.complex64, 11: torch.complex128}


@pytest.fixture(scope="module")
def G():
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    import gnuradio4_amd as G
    G.capi.lib()
    return G


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.fixture
def devsw(G):
    """developer switches of the library (gr4hip_developer_switch: which of two kernels serves a call), restored when the test ends"""
    used = set()

    def set_(name, value=1):
        used.add(name)
        G.capi.developer_switch(name, value)
    yield set_
    for name in used:
        G.capi.developer_switch(name, 0)


def _rel(got, truth):
    got = np.asarray(got).astype(np.complex128 if np.iscomplexobj(got) else np.float64).ravel()
    truth = np.asarray(truth).ravel()
    rms = np.sqrt(np.mean(np.abs(truth) ** 2))
    return float(np.max(np.abs(got - truth) / np.maximum(np.abs(truth), rms if rms > 0 else 1.0)))


def _rand(dtype_id, n, rng):
    dt = O.NP_DTYPES[dtype_id]
    if np.issubdtype(dt, np.integer):
        info = np.iinfo(dt)
        return rng.integers(info.min, info.max, size=n, dtype=dt, endpoint=True)
    if np.issubdtype(dt, np.complexfloating):
        return (rng.uniform(0.5, 4, n) * np.exp(2j * np.pi * rng.uniform(0, 1, n))).astype(dt)
    return (rng.uniform(0.5, 4, n) * rng.choice([-1, 1], n)).astype(dt)


def _rand_value(dtype_id, rng, op):
    dt = O.NP_DTYPES[dtype_id]
    if np.issubdtype(dt, np.integer):
        info = np.iinfo(dt)
        v = int(rng.integers(max(info.min, -1000), min(info.max, 1000), endpoint=True))
        if op == "Divide" and v in (0, -1):  # x / 0 and INT_MIN / -1 are UB in the reference
            v = 3
        return dt(v)
    if np.issubdtype(dt, np.complexfloating):
        return dt(rng.uniform(0.5, 2) * np.exp(2j * np.pi * rng.uniform(0, 1)))
    return dt(rng.choice([2.0, 0.5, 4.0, 3.0, 0.7, 1.25, -1.5, 10.0, 0.1]) if op in ("Multiply", "Divide") else rng.uniform(-3, 3))


def _chain_on_cpu(dtype_id, x, ops):
    """the blocks one after the other, each the oracle's MathOpImpl"""
    y = x
    for name, v in ops:
        y = O.math_const(_OPS[name], dtype_id, y, v)
    return y


@pytest.mark.parametrize("dtype_id", range(12))
def test_merged_math_chain_is_the_chain_of_blocks(G, dtype_id):
    """random programs of 1 .. 40 const ops, every sample type: vector body, scalar head / tail, spans of any alignment, several calls"""
    rng = np.random.default_rng(100 + dtype_id)
    n = 70_001
    x = _rand(dtype_id, n, rng)
    exact = dtype_id < 10  # integers and real floats: the same IEEE / modular operations in the same order
    for trial in range(6):
        length = int(rng.integers(1, 41)) if trial else 30
        names = list(rng.choice(list(_OPS), length, p=[0.3, 0.2, 0.3, 0.2]))
        if trial == 0:  # the reference's benchmark chain (bm_MergeApi.cpp:174), ten times over
            names = ["Multiply", "Divide", "Add"] * 10
        ops = [(nm, _rand_value(dtype_id, rng, nm)) for nm in names]
        if dtype_id >= 8:  # keep float magnitudes in range over long chains: every divide undoes a multiply of similar size
            ops = [(nm, v if nm in ("Add", "Subtract") else (v if abs(v) < 4 else v / abs(v) * 2)) for nm, v in ops]
        want = _chain_on_cpu(dtype_id, x, ops)
        m = G.Merged(_TORCH[dtype_id], ops)
        got = m.process_bulk(dev(x)).cpu().numpy()
        if exact:
            assert np.array_equal(got.view(np.uint8), want.view(np.uint8)), (dtype_id, trial, ops[:5])
        else:  # complex: the same formulas, but a long chain of complex products, quotients and cancelling sums is judged against its float64 / longdouble evaluation
            wide = np.complex128 if dtype_id == 10 else np.clongdouble
            truth = x.astype(wide)
            for nm, v in ops:
                truth = {"Add": truth + wide(v), "Subtract": truth - wide(v), "Multiply": truth * wide(v), "Divide": truth / wide(v)}[nm]
            scale = np.sqrt(np.mean(np.abs(truth) ** 2))
            e_dev, e_cpu = np.max(np.abs(got - truth)) / scale, np.max(np.abs(want - truth)) / scale
            assert e_dev <= max(3 * e_cpu, 1e-6 if dtype_id == 10 else 1e-14), (dtype_id, trial, float(e_dev), float(e_cpu))
        # ragged calls on spans that start at any element: the same bits
        buf = torch.empty(n + 9, dtype=_TORCH[dtype_id], device="cuda")
        out = torch.empty(n + 9, dtype=_TORCH[dtype_id], device="cuda")
        for off_in, off_out in ((1, 1), (3, 0)):
            buf[off_in:off_in + n].copy_(dev(x))
            m2 = G.Merged(_TORCH[dtype_id], ops)
            m2.process_bulk(buf[off_in:off_in + 777], out[off_out:off_out + 777])
            m2.process_bulk(buf[off_in + 777:off_in + n], out[off_out + 777:off_out + n])
            assert m2.position == n
            assert torch.equal(out[off_out:off_out + n].view(torch.uint8), dev(got).view(torch.uint8)), (off_in, off_out)
    assert G.Merged(_TORCH[dtype_id], []).process_bulk(dev(x)).cpu().numpy().tobytes() == x.tobytes()  # the empty program is the copy block
    with pytest.raises(G.capi.Gr4HipError):
        G.Merged(_TORCH[dtype_id], [("Rotator", 0.1)] if dtype_id != 10 else [("Add", 1)]).process_bulk(dev(_rand((dtype_id + 1) % 12, 8, rng)))


def test_merged_integer_runs_collapse_to_one_item(G):
    """+, -, * modulo 2^w are a ring: 1000 integer ops are still one multiply-add per sample (and still bit-exact); divisions keep their places"""
    rng = np.random.default_rng(5)
    x = _rand(6, 50_000, rng)
    ops = [(nm, _rand_value(6, rng, nm)) for nm in rng.choice(["Add", "Subtract", "Multiply"], 1000)]
    ops.insert(400, ("Divide", np.int32(7)))
    ops.insert(900, ("Divide", np.int32(-3)))
    want = _chain_on_cpu(6, x, ops)
    got = G.Merged(torch.int32, ops).process_bulk(dev(x)).cpu().numpy()
    assert np.array_equal(got, want)


def test_merged_rotator_and_gains(G):
    """Rotator<complex<float>> as an op of a program: MultiplyConst -> Rotator -> AddConst in one launch, the phase a function of the absolute sample index
    (ragged calls, an 8-byte-aligned span), against the float64 oracle rotator"""
    n = 300_000 + 5
    x = O.signal_c32(21, n)
    for inc, ph0 in ((0.6283185, 0.25), (-0.01, 1.0), (7.5, -2.0), (0.0, 0.5)):
        g, a = np.complex64(0.5 - 0.25j), np.complex64(0.125 + 2j)
        rot, _ = O.rotator((x.astype(np.complex128) * np.complex128(g)), float(np.float32(inc)), float(np.float32(ph0)))
        want = rot + np.complex128(a)
        m = G.Merged(torch.complex64, [("Multiply", g), ("Rotator", inc, ph0), ("Add", a)])
        xd = dev(x)
        buf = torch.empty(n + 1, dtype=torch.complex64, device="cuda")
        buf[1:].copy_(xd)
        got = np.concatenate([m.process_bulk(xd[:1001]).cpu().numpy(), m.process_bulk(buf[1:][1001:100_000]).cpu().numpy(), m.process_bulk(xd[100_000:]).cpu().numpy()])
        assert np.max(np.abs(got - want)) <= 1e-5 * np.max(np.abs(want)), (inc, ph0)
        m.reset()
        again = m.process_bulk(xd[:1001]).cpu().numpy()
        assert np.array_equal(again, got[:1001])


def _program_on_cpu64(x, ops):
    """a float / complex program in float64 (truth for filters with hooks): const ops and the oracle rotator"""
    y = x.astype(np.complex128 if np.iscomplexobj(x) else np.float64)
    for op in ops:
        if op[0] == "Rotator":
            y, _ = O.rotator(y, float(np.float32(op[1])), float(np.float32(op[2] if len(op) > 2 else 0.0)))
        else:
            v = np.complex128(op[1]) if np.iscomplexobj(y) else np.float64(np.float32(op[1]))
            y = {"Add": y + v, "Subtract": y - v, "Multiply": y * v, "Divide": y / v}[op[0]]
    return y


def _fir64(b, x, decim=1):
    y = np.convolve(x, np.asarray(b, np.float64))[:len(x)]
    return y[::decim]


@pytest.mark.parametrize("cplx", [False, True])
@pytest.mark.parametrize("ntaps,decim", [(45, 1), (200, 1), (64, 4), (1024, 8), (31, 3), (200, 16), (100, 8), (520, 8), (80, 10), (64, 2), (300, 5), (40, 100)])
def test_fir_takes_its_neighbours_into_its_launch(G, cplx, ntaps, decim):
    """per-sample blocks in front of and behind a FIR filter, executed by the filter's kernel: gains folded into the taps, adds / complex gains / a rotator as
    load and store hooks -- of the register-window kernel, or of the band-form matrix-pipe decimators (decimation 2 .. 12 float, 3 .. 16 complex, long spans) where the
    samples are split into bf16 planes; streamed in ragged calls (the carried history is the prologue's output); against float64 of the chain of separate blocks"""
    n = 12 * 8192 * decim if not cplx else 6 * 8192 * decim
    x = O.signal_c32(4, n) if cplx else O.signal_f32(4, n)
    b = O.design_taps_hamming_lowpass(ntaps, 0.1)
    dt = torch.complex64 if cplx else torch.float32
    if cplx:
        cases = [([("Multiply", 2.0), ("Divide", 0.8)], []),                                      # gains: folded
                 ([("Rotator", 0.3, 0.25)], []),                                                   # the channeliser's front end
                 ([("Multiply", 0.5 + 0.5j), ("Add", 0.25 - 1j)], [("Rotator", -0.2), ("Subtract", 1j)]),
                 ([], [("Multiply", 3.0)])]
    else:
        cases = [([("Multiply", 2.0)], []),
                 ([("Multiply", 1.5), ("Add", 0.75)], []),                                        # an add in front: the first ntaps - 1 outputs see the zero history
                 ([("Subtract", 0.5)], [("Multiply", 4.0), ("Add", -1.0)]),
                 ([("Divide", 3.0)], [("Divide", 0.5)])]                                          # gains on both sides: one product in the taps
    for pre, post in cases:
        truth = _program_on_cpu64(_fir64(b, _program_on_cpu64(x, pre), decim), post)
        f = G.fir_filter(b, dt, decimate=decim)
        if pre:
            f.set_prologue(G.Merged(dt, pre))
        if post:
            f.set_epilogue(G.Merged(dt, post))
        xd = dev(x)
        cuts = [0, 7 * decim, 1000 * decim, (n // decim // 2) * decim, n]
        got = np.concatenate([f.process_bulk(xd[a:b_]).cpu().numpy() for a, b_ in zip(cuts[:-1], cuts[1:])])
        assert len(got) == n // decim
        assert _rel(got, truth) <= 1e-5, (pre, post)


@pytest.mark.parametrize("cplx,ntaps,decim", [(False, 64, 4), (True, 64, 4), (True, 200, 10), (False, 300, 5), (False, 48, 3), (True, 40, 2)])
def test_hooked_fir_answers_to_the_guard(G, cplx, ntaps, decim):
    """a filter that carries its neighbours as load / store programs under a rejected tone 76 dB above what passes: the band-form decimators judge the prologue's OUTPUT
    against the filter's, the second evaluation (fir_exact_kernel<true>) runs the same programs; the register-window kernel redoes a rejected workgroup from the samples
    it staged.  The programs here are exact in float32 (x 1j; + 0.75 on samples of a 2^-10 grid), so what is measured is the filter's arithmetic: within the float64 bar,
    or within the error of the reference's float32 sum (oracle, reference order) on the prologue's output where float32 cannot reach it"""
    rng = np.random.default_rng(ntaps)
    n = decim * 4 * 30_000
    b = O.design_taps_hamming_lowpass(ntaps, 0.4 / decim)
    ph = 2 * np.pi * 0.31 * np.arange(n)
    x = 0.05 * (rng.standard_normal(n) + (1j * rng.standard_normal(n) if cplx else 0)) + 316.0 * (np.exp(1j * ph) if cplx else np.cos(ph))
    x = (np.round(x * 1024) / 1024).astype(np.complex64 if cplx else np.float32)
    dt = torch.complex64 if cplx else torch.float32
    pre, post = ([("Multiply", 1j)], [("Multiply", -1j)]) if cplx else ([("Add", 0.75)], [("Subtract", 0.25)])
    mid = _program_on_cpu64(x, pre)
    assert np.array_equal(mid, mid.astype(x.dtype))  # (exact in float32)
    truth = _program_on_cpu64(_fir64(b, mid, decim), post)
    ref32 = _program_on_cpu64(O.fir(b, mid.astype(x.dtype), acc64=False)[0][::decim], post)
    f = G.fir_filter(b, dt, decimate=decim)
    f.set_prologue(G.Merged(dt, pre))
    f.set_epilogue(G.Merged(dt, post))
    xd = dev(x)
    cuts = [0, (n // decim // 3) * decim, n]
    got = np.concatenate([f.process_bulk(xd[a:b_]).cpu().numpy() for a, b_ in zip(cuts[:-1], cuts[1:])])
    sl = slice(ntaps // decim + 1, None)
    e, e_ref = _rel(got[sl], truth[sl]), _rel(ref32[sl], truth[sl])
    assert e <= max(1e-5, e_ref), (e, e_ref)



@pytest.mark.parametrize("ntaps,decim", [(64, 8), (128, 4), (520, 4), (64, 16), (100, 32), (80, 10), (300, 5), (700, 8)])
def test_down_converter_steps_the_rotor_in_integers(G, ntaps, decim):
    """rotator -> decimating complex FIR (a down-converter: the channeliser's front end): the band-form decimators recognise a load program that is ONE rotator and step
    its phase -- a 64-bit fraction of a turn -- by integer additions instead of walking the program per sample (fir_band_hooks.hpp: BdRotor).  The phase is exact modulo
    2^64 either way, so the launch must be BIT-identical to the same filter with the program walked (a second op, + 0, keeps it off the special path), in ragged calls
    whose history is the rotor's output; and it answers to the float64 oracle"""
    n = 8 * 8192 * decim
    x = O.signal_c32(7, n)
    b = O.design_taps_hamming_lowpass(ntaps, 0.4 / decim)
    xd = dev(x)
    for inc, ph0 in ((0.3, 0.25), (-2.9, 1.0), (1e-4, -0.5)):
        truth = _fir64(b, _program_on_cpu64(x, [("Rotator", inc, ph0)]), decim)
        outs = []
        for prog in ([("Rotator", inc, ph0)], [("Rotator", inc, ph0), ("Add", 0.0)]):
            f = G.fir_filter(b, torch.complex64, decimate=decim)
            f.set_prologue(G.Merged(torch.complex64, prog))
            cuts = [0, 2 * decim, 1001 * decim, (n // decim // 2) * decim, n]
            outs.append(torch.cat([f.process_bulk(xd[a:b_]) for a, b_ in zip(cuts[:-1], cuts[1:])]))
        assert bool((outs[0] == outs[1]).all()), (inc, ph0)
        assert _rel(outs[0].cpu().numpy(), truth) <= 1e-5, (inc, ph0)


def test_fir_prologue_replaced_in_mid_stream(G):
    """a gain step in front of a filter (settings-by-tag on the MultiplyConst of MultiplyConst -> fir_filter): the samples already in the filter's history keep
    the OLD gain, exactly as when the two blocks run one after the other; the same for a hook (AddConst) replaced by another"""
    n, cut = 200_000, 77_777
    x = O.signal_f32(9, n)
    b = O.design_taps_hamming_lowpass(129, 0.1)
    for old, new in (([("Multiply", 0.5)], [("Multiply", 3.0)]), ([("Add", 1.0)], [("Add", -2.0), ("Multiply", 2.0)]), ([("Multiply", 2.0)], [])):
        staged = np.concatenate([_program_on_cpu64(x[:cut], old), _program_on_cpu64(x[cut:], new)])
        truth = _fir64(b, staged)
        f = G.fir_filter(b, torch.float32)
        f.set_prologue(G.Merged(torch.float32, old))
        xd = dev(x)
        y0 = f.process_bulk(xd[:cut]).cpu().numpy()
        f.set_prologue(G.Merged(torch.float32, new) if new else None)
        y1 = f.process_bulk(xd[cut:cut + 50]).cpu().numpy()  # a short call right behind the change: the history still holds samples of the old prologue
        y2 = f.process_bulk(xd[cut + 50:]).cpu().numpy()
        assert _rel(np.concatenate([y0, y1, y2]), truth) <= 1e-5, (old, new)


def test_fir_hook_dtype_mismatch_is_refused(G):
    f = G.fir_filter(np.ones(8, np.float32) / 8, torch.float32)
    with pytest.raises(G.capi.Gr4HipError) as e:
        f.set_prologue(G.Merged(torch.complex64, [("Rotator", 0.1)]))
    assert e.value.status == G.capi.UNSUPPORTED


def test_decimator_takes_the_blocks_behind_it(G):
    """Decimator<T> + per-sample blocks in one launch (only the kept samples are read): bit-exact for an integer type, a rotator behind the decimator counts OUTPUT samples"""
    rng = np.random.default_rng(3)
    x = _rand(5, 100_001, rng)
    ops = [("Multiply", np.int16(7)), ("Add", np.int16(-300)), ("Divide", np.int16(3))]
    for decim in (1, 3, 8, 100):
        want = _chain_on_cpu(5, np.ascontiguousarray(x[::decim]), ops)
        got = G.Merged(torch.int16, ops).decimate(dev(x), decim).cpu().numpy()
        assert np.array_equal(got, want), decim
    xc = O.signal_c32(3, 80_000)
    m = G.Merged(torch.complex64, [("Rotator", 0.25, 0.5), ("Multiply", 2.0)])
    got = np.concatenate([m.decimate(dev(xc[:40_000]), 8).cpu().numpy(), m.decimate(dev(xc[40_000:]), 8).cpu().numpy()])
    rot, _ = O.rotator(xc[::8].astype(np.complex128), float(np.float32(0.25)), float(np.float32(0.5)))
    assert np.max(np.abs(got - 2.0 * rot)) <= 1e-5 * np.max(np.abs(rot))


@pytest.mark.parametrize("N,frames", [(1024, 40), (256, 33), (1000, 12), (6, 50), (8192, 300), (16384, 3), (1009, 4)])
def test_power_spectrum_takes_the_blocks_behind_it(G, N, frames):
    """float blocks behind a power spectrum -- |X|^2 / N^2 + offset -- ride in the transform's launch (gr4hip_fft_set_epilogue): every kernel family of the FFT block
    (fast power-of-two, mixed-radix, small block kernel, the 8192-point frame pipeline, four-step, chirp), against the separate blocks on the same output"""
    x = O.signal_c32(N, N * frames)
    ops = [("Divide", float(N) * N), ("Multiply", 3.0), ("Add", 0.125)]
    plain = G.FFT(N, "Hann").mag2(dev(x))
    want = G.Merged(torch.float32, ops).process_bulk(plain.reshape(-1)).cpu().numpy()
    f = G.FFT(N, "Hann")
    f.set_epilogue(G.Merged(torch.float32, ops))
    got = f.mag2(dev(x)).cpu().numpy().ravel()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))  # the same float operations on the same |X|^2
    f.set_epilogue(None)
    assert torch.equal(f.mag2(dev(x)), plain)
    with pytest.raises(G.capi.Gr4HipError):
        f.set_epilogue(G.Merged(torch.complex64, [("Add", 1)]))


def test_division_by_a_constant_is_the_ieee_quotient_for_every_float(G):
    """DivideConst<float> inside a merged program multiplies by the correctly rounded reciprocal and corrects with two fused multiply-adds (three operations instead of
    the general quotient's eleven).  That this IS the IEEE quotient is not argued but checked: ALL 2^32 float bit patterns (zeros, subnormals, infinities, NaNs included)
    through both forms, for constants across the admitted range, compared bit for bit on the device"""
    consts = [3.0, 0.7, 10.0, 0.1, -1.5, 1.25, 7.0e-9, 9.87654e11, 1.0000001, 0.99999994, 3.4028235e-12, 12345.678, float(np.float32(2.0) ** -40), float(np.float32(2.0) ** 40)]
    chunk = 1 << 28
    base = torch.arange(chunk, dtype=torch.int64, device="cuda")
    for c in consts:
        fast = G.Merged(torch.float32, [("Divide", c)])
        G.capi.developer_switch("GR4HIP_EWISE_NO_DIV_RCP", 1)
        try:
            slow = G.Merged(torch.float32, [("Divide", c)])
            slow.process_bulk(torch.zeros(8, device="cuda"))  # (the program is compiled at its first launch: with the switch set)
        finally:
            G.capi.developer_switch("GR4HIP_EWISE_NO_DIV_RCP", 0)
        fast.process_bulk(torch.zeros(8, device="cuda"))
        for k in range(16):
            bits = (base + k * chunk).to(torch.int32) if k < 8 else (base + k * chunk - (1 << 32)).to(torch.int32)
            x = bits.view(torch.float32)
            a, b = fast.process_bulk(x), slow.process_bulk(x)
            same = a.view(torch.int32) == b.view(torch.int32)
            if not bool(same.all()):
                bad = (~same).nonzero()[:4].ravel().tolist()
                raise AssertionError((c, k, [(float(x[i]), float(a[i]), float(b[i])) for i in bad]))
        want = (np.float32(1.2345) / np.float32(c))
        assert fast.process_bulk(torch.full((4,), 1.2345, device="cuda"))[0].item() == want


def _lowpass(ntaps, fc):
    k = np.arange(ntaps, dtype=np.float64)
    t = np.hamming(ntaps) * 2 * fc * np.sinc(2 * fc * (k - (ntaps - 1) / 2.0))
    return (t / t.sum()).astype(np.float32)


@pytest.mark.parametrize("order,nblocks,extra", [(8, 200, 0), (8, 1500, 4321 * 8), (4, 777, 8), (2, 64, 7160), (6, 3000, 0)])
def test_decimating_fir_and_iir_cascade_in_one_launch(G, order, nblocks, extra):
    """BASELINE configs[2] (decimate-by-8 1024-tap FIR -> biquad cascade) through gr4hip_fir_iir_process: the cascade as the frequency-domain decimator's store epilogue
    (one launch: contiguous block runs per workgroup, the state carried in one wave, warm-up blocks in front of every run) against the float64 oracle AND against the
    same call with the decimated stream in HBM; the handles' states are interchangeable with the separate calls (a second span continues the stream either way)"""
    import gnuradio4_amd.blocks as B
    n = nblocks * 7168 + extra
    x = O.signal_f32(order + nblocks, n)
    taps = _lowpass(1024, 0.05)
    b, a = B.design_iir(G.capi.LOWPASS, order, 0.05, float("nan"), 1.0, G.capi.BUTTERWORTH)
    secs = O.make_sections([(bb, aa) for bb, aa in zip(b, a)])
    yd, _ = O.fir_decim(taps, x, 8)
    truth = O.iir_cascade(secs, yd.astype(np.float32), O.DF_II, f64=True)
    # (the oracle's cascade takes float32 input: the decimated stream as the filter block hands it on -- rounded to float32 -- as on the device)
    cut = (n // 2) // 8 * 8
    res = {}
    for mode, m in (("fused", G.capi.FIR_IIR_ONE_LAUNCH), ("two", G.capi.FIR_IIR_TWO_LAUNCHES), ("auto", G.capi.FIR_IIR_AUTO)):
        fir, iir = G.fir_filter(taps, torch.float32, decimate=8), G.iir_filter(b, a)
        xd = dev(x)
        y = torch.cat([B.fir_iir_process(fir, iir, xd[:cut], mode=m), B.fir_iir_process(fir, iir, xd[cut:], mode=m)])
        res[mode] = y.cpu().numpy()
        assert len(res[mode]) == n // 8
        assert _rel(res[mode], truth) <= 1e-5, (mode, order, nblocks)
    assert _rel(res["fused"], res["two"].astype(np.float64)) <= 1e-5  # (each is within 1e-5 of float64: different float32 roundings of the same cascade)
    # mixed with the separate calls on the same handles: first half fused, second half as gr4hip_fir_process + gr4hip_iir_process
    fir, iir = G.fir_filter(taps, torch.float32, decimate=8), G.iir_filter(b, a)
    xd = dev(x)
    y1 = B.fir_iir_process(fir, iir, xd[:cut], mode=G.capi.FIR_IIR_ONE_LAUNCH)
    y2 = iir.process_bulk(fir.process_bulk(xd[cut:]))
    assert _rel(torch.cat([y1, y2]).cpu().numpy(), truth) <= 1e-5
    with pytest.raises(G.capi.Gr4HipError):
        B.fir_iir_process(fir, iir, xd[:cut], mode=7)
