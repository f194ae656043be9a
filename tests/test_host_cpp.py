"""The C++ host layer (gnuradio4_amd/host: gr::Block<>/Port<>/Graph::connect surface + compute_domain seam).
CPU: its self-test program, BASELINE configs[0] plumbing against the oracle, and the loud failure of device blocks without a GPU.
GPU: device graphs (seam offload and fused device run) through the C++ API against the oracle."""
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O

ROOT = O.ROOT
BIN = os.path.join(ROOT, "build", "host")


@pytest.fixture(scope="module")
def host_bins():
    subprocess.check_call(["bash", os.path.join(ROOT, "gnuradio4_amd", "host", "build.sh")], stdout=subprocess.DEVNULL)
    return BIN


def test_host_selftest_and_config0_plumbing(host_bins, tmp_path):
    dump = tmp_path / "c1.bin"
    r = subprocess.run([os.path.join(host_bins, "test_host_cpu"), str(dump)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all checks passed" in r.stdout
    raw = open(dump, "rb").read()
    K, n = np.frombuffer(raw[:16], np.uint64)
    taps = np.frombuffer(raw[16:16 + 4 * int(K)], np.float32)
    y = np.frombuffer(raw[16 + 4 * int(K):], np.float32)
    assert K == 64 and n == 1000448 and len(y) == n  # BASELINE configs[0]: sample count in == out
    x = np.sin(2 * np.pi * 50.0 * (np.arange(n, dtype=np.float64) / 1000.0)).astype(np.float32)
    truth, _ = O.fir(taps, x[:65536])
    assert np.max(np.abs(y[:65536] - truth)) <= 1e-5 * np.sqrt(np.mean(truth ** 2))  # first outputs equal the oracle
    # steady state: 50 Hz at fs = 1 kHz sits in the passband of the fc = 0.1 low-pass
    assert abs(np.max(np.abs(y[-2000:])) - 1.0) < 2e-2


REFERENCE = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="container-only: needs the reference tree where it lies (nothing of it travels)")
def test_reference_block_headers_compile_unmodified(tmp_path):
    """SURVEY.md 8(b) row 1, "existing block source must compile unchanged against the new headers": the reference's own
    blocks/math/include/gnuradio-4.0/math/{Math,Rotator}.hpp, included from /root/reference as they are, compiled against this host layer's
    gnuradio-4.0/ forwarding headers, instantiated in a Graph and run on the host path against the vectors of qa_Math.cpp:59-149 and
    qa_Rotator.cpp:69-92 (gnuradio4_amd/host/tests/test_reference_dropin.cpp).  The only include paths: this layer, and the reference's blocks/math."""
    exe = tmp_path / "test_reference_dropin"
    cmd = ["g++", "-std=c++20", "-Wall", "-Wextra", "-O1", "-I" + os.path.join(ROOT, "gnuradio4_amd", "host", "include"),
           "-I" + os.path.join(REFERENCE, "blocks", "math", "include"), os.path.join(ROOT, "gnuradio4_amd", "host", "tests", "test_reference_dropin.cpp"), "-o", str(exe)]
    c = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert c.returncode == 0, c.stderr[-4000:]
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "reference drop-in: all checks passed" in r.stdout, r.stdout + r.stderr
    assert "(Math.hpp unmodified)" in r.stdout and "(MathOpMultiPortImpl unmodified)" in r.stdout and "(Rotator.hpp unmodified)" in r.stdout
    # nothing of the reference's text is kept in the repository: the test program only #includes it
    src = open(os.path.join(ROOT, "gnuradio4_amd", "host", "tests", "test_reference_dropin.cpp")).read()
    assert "#include <gnuradio-4.0/math/Math.hpp>" in src and "struct MathOpImpl" not in src


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="container-only: needs the reference tree where it lies (nothing of it travels)")
def test_reference_algorithm_headers_compile_unmodified(tmp_path, golden):
    """the reference's algorithm/fourier/window.hpp and fft_common.hpp, included from /root/reference as they are, compile against this host layer (gr4/compat.hpp
    carries gr::meta::fixed_string and gr::meta::array_or_vector_type for them) and reproduce what the reference's own test holds for them
    (qa_algorithm_fourier.cpp:145-180: the N = 8 array of every window, the numpy.unwrap vector) -- and they agree with the oracle's restatement and with the
    library's host-side window::create (gr4hip_window_create), i.e. the restatements are checked against the reference's own code, not only against its test vectors"""
    import ctypes as C
    import numpy as np
    import oracle_lib as O
    exe = tmp_path / "test_reference_algorithm_dropin"
    cmd = ["g++", "-std=c++20", "-w", "-O1", "-I" + os.path.join(ROOT, "gnuradio4_amd", "host", "include"), "-I" + os.path.join(REFERENCE, "algorithm", "include"),
           "-I" + os.path.join(REFERENCE, "third_party", "magic_enum"), os.path.join(ROOT, "gnuradio4_amd", "host", "tests", "test_reference_algorithm_dropin.cpp"), "-o", str(exe)]
    c = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert c.returncode == 0, c.stderr[-4000:]
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "reference algorithm drop-in: done (window.hpp, fft_common.hpp unmodified)" in r.stdout and "FAILED" not in r.stdout, r.stdout + r.stderr
    rows = {ln.split()[0] + (" " + ln.split()[1] if ln.startswith("window ") else ""): ln for ln in r.stdout.splitlines()}
    vals = lambda key, skip: np.array([float(v) for v in rows[key].split()[skip:]])
    from gnuradio4_amd import capi
    for wid, name in enumerate(O.WINDOWS):
        got = vals(f"window {wid}", 2)
        np.testing.assert_allclose(got, golden["window_n8"][name], atol=2e-6, err_msg=name)           # the vectors the reference's test holds
        np.testing.assert_allclose(got, O.window(wid, 8, np.float32), rtol=0, atol=1e-7, err_msg=name)  # the oracle's restatement against the reference's code
        w = np.empty(8, np.float32)
        assert capi.lib().gr4hip_window_create(wid, w.ctypes.data, 8, C.c_float(1.6)) == 0
        np.testing.assert_allclose(got, w, rtol=0, atol=1e-7, err_msg=name)                               # the library's host-side window::create
    assert rows["typenames"].split(" ", 1)[1] == "[" + ", ".join(O.WINDOWS) + "]"
    np.testing.assert_allclose(vals("unwrap", 1), golden["unwrap"]["expected"], atol=1e-7)
    spec = np.array([1, -2j, -3 + 3j, 0.5 + 0.25j, 0, 2 - 1j, -1 - 1j, 4], np.complex64)
    np.testing.assert_allclose(vals("magnitude_shifted", 1), O.magnitude(spec, shift=True), rtol=1e-6)
    np.testing.assert_allclose(vals("magnitude_half_db", 1), O.magnitude(spec, half=True, in_db=True), rtol=1e-6)
    np.testing.assert_allclose(vals("phase_deg_unwrapped_shifted", 1), O.phase(spec, in_deg=True, unwrap=True, shift=True), rtol=1e-5, atol=1e-4)
    src = open(os.path.join(ROOT, "gnuradio4_amd", "host", "tests", "test_reference_algorithm_dropin.cpp")).read()
    assert "#include <gnuradio-4.0/algorithm/fourier/window.hpp>" in src and "bessel_i0" not in src  # only #included, never copied


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="container-only: needs the reference tree where it lies (nothing of it travels)")
def test_reference_uncertain_value_header_compiles_unmodified_and_pins_the_fixture():
    """the reference's meta/UncertainValue.hpp, included from /root/reference as it is, compiles against this host layer (its <gnuradio-4.0/meta/utils.hpp> is the
    layer's forwarding header) -- its operators on 2304 operand pairs per type ARE the committed fixture tests/golden/uncertain_value_ops.npz (which the oracle and
    the device are compared with, tests/test_oracle_golden.py / test_gpu_parity.py), and the layer's own gr::UncertainValue gives the same numbers bit for bit"""
    import importlib.util
    import numpy as np
    spec = importlib.util.spec_from_file_location("make_uncertain_fixture", os.path.join(ROOT, "tests", "golden", "make_uncertain_fixture.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    ref, own = mod.table(True), mod.table(False)
    fx = np.load(os.path.join(ROOT, "tests", "golden", "uncertain_value_ops.npz"))
    for k in ("f32", "f64"):
        assert ref[k].shape == (2304, 12)
        assert np.array_equal(ref[k], fx[k].astype(np.float64)), k   # the fixture is what the reference's code computes
        assert np.array_equal(own[k], ref[k]), k                      # the layer's type: the same arithmetic
    src = open(os.path.join(ROOT, "gnuradio4_amd", "host", "tests", "test_reference_uncertain_value.cpp")).read()
    assert "#include REF_UNCERTAIN_HPP" in src and "hypot" not in src  # only #included, never copied


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="container-only: needs the reference tree where it lies (nothing of it travels)")
def test_hbm_ring_models_the_reference_bufferlike_concept():
    """the reference's own BufferLike / BufferReaderLike / BufferWriterLike concepts (core/include/gnuradio-4.0/Buffer.hpp:78-102, included
    unmodified from /root/reference; std-only header) hold for gr::hip::CircularBuffer<T> (static_asserts in host/tests/test_reference_bufferlike.cpp)"""
    cmd = ["g++", "-std=c++20", "-fsyntax-only", "-w", "-I" + os.path.join(ROOT, "gnuradio4_amd", "host", "include"), "-I" + os.path.join(REFERENCE, "core", "include"),
           os.path.join(ROOT, "gnuradio4_amd", "host", "tests", "test_reference_bufferlike.cpp")]
    c = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert c.returncode == 0, c.stderr[-4000:]
    # and the C-ABI's status codes against the reference's own WorkStatus.hpp (include/gr4hip.h: "status codes reuse gr::work::Status values")
    c = subprocess.run(["g++", "-std=c++20", "-fsyntax-only", "-w", "-I" + os.path.join(REFERENCE, "core", "include"),
                        os.path.join(ROOT, "gnuradio4_amd", "host", "tests", "test_reference_workstatus.cpp")], capture_output=True, text=True, timeout=600)
    assert c.returncode == 0, c.stderr[-4000:]


def _inputs(tmp_path, N, frames, ntaps):
    x = O.signal_c32(42, frames * N)
    b = O.design_taps_hamming_lowpass(ntaps, 0.1)
    x.tofile(tmp_path / "in.bin")
    b.tofile(tmp_path / "taps.bin")
    return x, b


def test_device_blocks_fail_loudly_without_gpu(host_bins, tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    _inputs(tmp_path, 1024, 2, 33)
    r = subprocess.run([os.path.join(host_bins, "test_host_device"), str(tmp_path / "in.bin"), str(tmp_path / "taps.bin"), "1024", str(tmp_path / "o")],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 3, (r.returncode, r.stdout, r.stderr)  # work::Status::ERROR from the seam, no host fallback
    assert "NO_DEVICE" in r.stderr and not os.path.exists(tmp_path / "o_fir.bin")
    assert "port domains: CPU -> GPU refused, converter required" in r.stdout  # checked before anything touches a device


PLUGIN = os.path.join(ROOT, "gnuradio4_amd", "libgr4hip_blocks.so")


def test_plugin_entry_host_domain(host_bins):
    """gr_plugin_make / gr_plugin_free (Plugin.hpp:82-85): a loader that links nothing finds the blocks under the reference's registry names,
    builds a graph from property_maps and runs it (host bodies); a library without the entry points is refused with a reason"""
    r = subprocess.run([os.path.join(host_bins, "test_host_plugin"), PLUGIN, "host", os.path.join(ROOT, "gnuradio4_amd", "libgr4hip.so")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all checks passed (compute_domain host)" in r.stdout and "refused:" in r.stdout and "gr_plugin_make / gr_plugin_free missing" in r.stdout
    exported = subprocess.run(["nm", "-D", "--defined-only", PLUGIN], capture_output=True, text=True).stdout
    assert " T gr_plugin_make" in exported and " T gr_plugin_free" in exported


def test_plugin_device_domain_fails_loudly_without_gpu(host_bins):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    r = subprocess.run([os.path.join(host_bins, "test_host_plugin"), PLUGIN, "gpu:hip:0"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 3, (r.returncode, r.stdout, r.stderr)


@pytest.mark.gpu
def test_plugin_entry_device_domain(host_bins):
    """the same registry graph with compute_domain: gpu:hip:0 on every block: the seam is taken (bit-exact int32 wrap, FIR + Decimator, tag rescaled)"""
    r = subprocess.run([os.path.join(host_bins, "test_host_plugin"), PLUGIN, "gpu:hip:0"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all checks passed (compute_domain gpu:hip:0)" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("N,ntaps", [(8192, 256), (1024, 64)])
def test_device_graphs_match_oracle(host_bins, tmp_path, N, ntaps):
    frames = 5
    x, b = _inputs(tmp_path, N, frames, ntaps)
    r = subprocess.run([os.path.join(host_bins, "test_host_device"), str(tmp_path / "in.bin"), str(tmp_path / "taps.bin"), str(N), str(tmp_path / "o")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "chain_fir_fft_mag2  (1 stage)" in r.stdout
    assert "wrapping spans contiguous" in r.stdout  # gr::hip::CircularBuffer<T>: BufferLike ring in HBM, double-mapped

    def rel(got, truth):
        rms = np.sqrt(np.mean(np.abs(truth) ** 2))
        return float(np.max(np.abs(got - truth) / np.maximum(np.abs(truth), rms)))
    fir = np.fromfile(tmp_path / "o_fir.bin", np.complex64)
    truth, _ = O.fir(b, x)
    assert len(fir) == len(x) and rel(fir, truth) <= 1e-5
    for name, wid in (("o_chain.bin", 0), ("o_chain_hann.bin", 3)):
        got = np.fromfile(tmp_path / name, np.float32)
        t, _ = O.chain(b, x, N, wid, truth=True)
        assert len(got) == frames * N and rel(got, t) <= 1e-5, name
    # the planner's runs: fused chain with a Blackman-Harris window, and MultiplyConst -> fir_filter<float> | host AddConst | MultiplyConst
    assert "planner: 2 runs: [chain_fir_fft_mag2] [fir_f32[pre: mul]]" in r.stdout  # the gain in front of the filter rides in its launch (folded into the taps)
    got = np.fromfile(tmp_path / "o_chain_planned_bh.bin", np.float32)
    t, _ = O.chain(b, x, N, [w.lower() for w in O.WINDOWS].index("blackmanharris"), truth=True)
    assert len(got) == frames * N and rel(got, t) <= 1e-5
    pf = np.fromfile(tmp_path / "o_planned_float.bin", np.float32)
    xin = np.resize(np.array([1.0, -2.0, 3.0, 0.5, 0.25], np.float64), 200000) * 2.0
    want = (np.convolve(xin, [0.5, 0.25, 0.25])[:200000] + 1.0) * 3.0
    assert len(pf) == 200000 and np.max(np.abs(pf - want)) <= 1e-5
    # the page-locked host edge is a double-mapped RING (CircularBuffer.hpp:75-172 on the host side of the link): 20 x its capacity through it, nothing moved, nothing staged
    assert "host ring edge: page-locked double-mapped rings" in r.stdout and "every sample arrived, 0 bytes staged" in r.stdout
    # a tee on a GPU-domain edge: two read cursors on ONE ring in HBM (CircularBuffer.hpp:880-946: one writer -> N readers), no copy; both streams match the oracle
    assert "gpu-domain tee: 2 readers on one ring in HBM" in r.stdout
    gt = np.fromfile(tmp_path / "o_gpu_tee_fir.bin", np.complex64)
    assert len(gt) == len(x) and rel(gt, truth) <= 1e-5
    gs = np.fromfile(tmp_path / "o_gpu_tee_spec.bin", np.float32)
    t0, _ = O.chain(b, x, N, 0, truth=True)
    assert len(gs) == frames * N and rel(gs, t0) <= 1e-5
    # a tee behind a device block: nothing fused across it, both readers complete, the filtered stream equals the oracle's
    assert "tee'd device edge: 0 fused runs" in r.stdout
    tee = np.fromfile(tmp_path / "o_tee_fir.bin", np.complex64)
    assert len(tee) == len(x) and rel(tee, truth) <= 1e-5
    # GPU-domain ports with explicit converter blocks: the same spectra as the fused device run, the input edge pinned by the "hip" provider
    assert "port domains: CPU -> GPU refused, converter required" in r.stdout
    assert "input edge pinned by the hip provider" in r.stdout and "(0 staged)" in r.stdout
    got = np.fromfile(tmp_path / "o_gpu_ports_hann.bin", np.float32)
    t, _ = O.chain(b, x, N, 3, truth=True)
    assert len(got) == frames * N and rel(got, t) <= 1e-5
    # merge API: the reference benchmark's FeedbackMerge IIR collapses into one first-order section of the scan kernel
    assert "merge IIR low-pass (FeedbackMerge) on the device: stage 'iir_f32'" in r.stdout
    assert "merge MultiplyConst -> fir_filter on the device: stage 'fir_f32[pre: mul]'" in r.stdout
    # kernel-level fusion (the run-time Merge<>): the reference's merged benchmark chains are ONE program each, bit-identical to the merged block on the host ...
    for value in (2, 3):
        assert f"merge mult->div->add (value {value}) on the device: stage 'ewise[mul,div,add]', bit-identical to the host block; (mult->div->add)^10: stage 'ewise[30 ops]', bit-identical" in r.stdout
    assert "merge mult->div->add <int32> on the device: bit-identical to the host block" in r.stdout
    # ... three graph blocks are one stage, one launch per chunk; fir_filter -> Decimator is the polyphase decimating FIR with its neighbours absorbed
    assert "planner (math chain): 1 run: ewise[mul,div,add]  (1 stage, 1 launch), output bit-identical to the host graph" in r.stdout
    assert "planner (gain -> fir -> Decimator -> add): 1 run: fir_f32/5[pre: mul; post: add]" in r.stdout
    assert "planner (blocks around a Decimator, behind a PowerSpectrum): [decimator[post: add,mul]] bit-identical to the host graph; [power_spectrum_c32[post: div,add]] max rel err" in r.stdout
    # ... and the channeliser Rotator -> BasicDecimatingFilter<complex<float>> -> PowerSpectrum is two launches per chunk with only the decimated stream in HBM:
    # the rotator is the filter's load hook.  Against the oracle: float64 rotator, float64 FIR on the rotator's float32 output, every 8th sample, 256-point Hann frames
    assert "planner (channeliser): 1 run: basic_fir_decim[pre: rot] -> power_spectrum_c32  (2 stages, 2 launches" in r.stdout
    ctaps = np.fromfile(tmp_path / "o_channeliser_taps.bin", np.float32)
    nin = 8 * 256 * 24
    xin = np.resize(x, nin)
    rot, _ = O.rotator(xin.astype(np.complex128), float(np.float32(0.3)), float(np.float32(0.25)))
    yd = O.fir(ctaps, rot.astype(np.complex64))[0][::8]
    want, _ = O.chain(np.ones(1, np.float32), yd.astype(np.complex64), 256, 3, truth=True)
    got = np.fromfile(tmp_path / "o_channeliser.bin", np.float32)
    assert len(got) == 256 * 24 and rel(got, want) <= 1e-5
    assert "merge IIR low-pass (SplitMergeCombine feedback) on the device: stage 'iir_f32'" in r.stdout
    assert "SplitMergeCombine on the device: stage 'split(fir_f32 | math_const)'" in r.stdout
    assert "pipelined run:" in r.stdout  # (round 6: default-size edges are gathered in the device ring -- this short stream is ONE launch; the overlap of copies and kernels is bench_host_feed's subject)
    # tags through a fused device run
    assert "tags through the device run: 2 forwarded, 1 stage rebuilt" in r.stdout
    assert "tags: device run forwarded {0: 250 Hz, 10000: 250 Hz + gr:value}, host graph the same" in r.stdout
    # every other hot-path block behind the seam (device vs the host body of the same block, printed by the program) ...
    for what in ("iir_filter<float, DF_II>", "Decimator<int32> decim 7", "Rotator<complex<float>>", "BasicDecimatingFilter<float> FIR /5", "BasicFilter<float> IIR",
                 "Add<int32> n_inputs = 3", "FFT<complex<float>> 256 Hann", "FFT<complex<float>> 1000 B-Harris", "FFT<float> 512 Hamming dB", "planned run with two rate changes",
                 "fir_interpolator<float> x2", "fir_interpolator<float> x3", "fir_interpolator<float> x8", "fir_interpolator<float> x7", "Rotator<complex<float>> vs host body (first 64)"):
        assert f"seam {what}" in r.stdout, what
    assert "FAILED" not in r.stdout
    assert "settings-by-tag on a lone device block (MultiplyConst value)" in r.stdout and "settings-by-tag on a lone device block (fir_filter taps, history kept)" in r.stdout
    assert "planner (resampling): 1 run: basic_fir_decim[pre: mul] -> decimator -> iir_f32" in r.stdout
    # ... and BasicDecimatingFilter<float> (designed Hamming FIR / Chebyshev-1 IIR low-pass, order 4, 100 Hz at 1 kHz, decimate 5) against the oracle
    xin = np.fromfile(tmp_path / "o_basic_in.bin", np.float32)
    par = O.filter_params(order=4, fLow=100.0, fs=1000.0)
    want, _ = O.fir_decim(O.fir_design(0, par, [w.lower() for w in O.WINDOWS].index("hamming")).astype(np.float32), xin, 5)
    got = np.fromfile(tmp_path / "o_basic_fir5.bin", np.float32)
    assert len(got) == len(xin) // 5 and rel(got, want) <= 1e-5
    secs = O.iir_design(0, par, 2)  # CHEBYSHEV1
    want = O.iir_cascade(O.make_sections([(np.float32(b), np.float32(a)) for b, a in secs]), xin)[::5]
    got = np.fromfile(tmp_path / "o_basic_iir5.bin", np.float32)
    assert len(got) == len(xin) // 5 and rel(got, want) <= 1e-5
    m = np.fromfile(tmp_path / "o_math.bin", np.int32)
    src = np.resize(np.array([2147483647, -5, 7, 123456789], np.int32), 100000)
    assert np.array_equal(m, (src.astype(np.int64) * 3).astype(np.int32))  # wrap-around like the C++ int32 product


def test_signal_generator_is_the_reference_core(host_bins, tmp_path):
    """gr::basic::SignalGenerator<T> of the host mirror (all eleven signal types; float / double / int16 / complex<float>) against the reference's OWN
    SignalGeneratorCore<T> -- oracle/_ref/libgr4ref.so is compiled from algorithm/.../signal/{Tone,Noise,SignalGeneratorCore}.hpp and the rng headers where they
    lie under /root/reference (oracle/Makefile; no stand-ins).  200 000 samples: past the phasor renormalisation at 65 536.  Same arithmetic, same compiler:
    bit for bit, except where libm's sin / cos are evaluated in a different context (a few ulp allowed there)."""
    import ctypes as C
    ref_path = os.path.join(ROOT, "oracle", "_ref", "libgr4ref.so")
    if not os.path.exists(ref_path):
        pytest.skip("oracle/_ref not built (no reference tree here)")
    R = C.CDLL(ref_path)
    if not hasattr(R, "gr4ref_signal_f32"):
        pytest.skip("oracle/_ref predates the signal-generator harness")
    n = 200_000
    r = subprocess.run([os.path.join(host_bins, "dump_signal_generator"), str(tmp_path / "sg"), str(n)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    cases = {"f32": (np.float32, R.gr4ref_signal_f32, 1.5, 0.25), "f64": (np.float64, R.gr4ref_signal_f64, 1.5, 0.25), "i16": (np.int16, R.gr4ref_signal_i16, 30000.0, 9000.0),
             "c32": (np.complex64, R.gr4ref_signal_c32, 1.5, 0.25)}
    for tname, (dt, fn, amp, off) in cases.items():
        fn.argtypes = [C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_uint64, C.c_void_p, C.c_size_t]
        fn.restype = None
        for t in range(11):
            want = np.empty(n, dt)
            fn(t, 37.5, 1000.0, 0.3, amp, off, 12345, want.ctypes.data, n)
            got = np.fromfile(tmp_path / f"sg_{tname}_{t}.bin", dt)
            assert got.shape == want.shape, (tname, t)
            if np.array_equal(got, want):
                continue
            err = np.abs(got.astype(np.complex128) - want.astype(np.complex128)).max()
            assert t not in (0, 3, 8, 9) and err <= (1 if dt == np.int16 else 4e-6 if dt in (np.float32, np.complex64) else 1e-12), (tname, t, err)


@pytest.mark.gpu
def test_sharded_graph_one_rank_communicator(host_bins, tmp_path):
    """the 8-channel graph of BASELINE configs[4] written with compute_domain: gpu:hip:{c mod N} per branch and planned by gr::hip::plan_sharded for one rank:
    all channels in ONE launch per exchange (gr4hip_chain_process_multi, the local part of math::Add as its store epilogue), the cross-device edge as an RCCL
    collective on a one-rank communicator (ncclCommInitRank / ncclAllReduce / ncclReduceScatter execute; the transport does not).  Against the oracle sum."""
    N, frames, ntaps, C = 8192, 10, 256, 8
    x, b = _inputs(tmp_path, N, frames, ntaps)
    r = subprocess.run([os.path.join(host_bins, "test_host_fanin"), str(tmp_path / "in.bin"), str(tmp_path / "taps.bin"), str(N), str(tmp_path / "o"), str(C)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "communicator: rank 0 of 1" in r.stdout and "all sharded-graph checks passed" in r.stdout
    assert "plan for rank 0 of 2: ok (4 local channels, 6 blocks)" in r.stdout

    def rel(got, truth):
        rms = np.sqrt(np.mean(np.abs(truth) ** 2))
        return float(np.max(np.abs(got - truth) / np.maximum(np.abs(truth), rms)))
    for variant, gains in ((0, [1.0] * C), (1, [1.0 + 0.125 * c for c in range(C)]), (2, [1.0] * C)):
        want = 0.0
        for c in range(C):
            t, _ = O.chain((b.astype(np.float64) * gains[c]).astype(np.float32), np.roll(x, -977 * c), N, 0, truth=True)
            want = want + t
        got = np.fromfile(tmp_path / f"o_fanin{variant}.bin", np.float32)
        assert len(got) == frames * N and rel(got, want) <= 1e-5, variant


@pytest.mark.gpu
def test_fanin_entry_points_one_rank():
    """gr4hip_fanin_* straight through the C-ABI on a one-rank communicator: reduce_scatter, all_to_all + rank-order fold and all_reduce all return the input"""
    import ctypes as C
    import torch
    import gnuradio4_amd as G
    L = G.capi.lib()
    ident = (C.c_char * 128)()
    G.capi.check(L.gr4hip_fanin_unique_id(ident), "unique_id")
    h = C.c_void_p()
    G.capi.check(L.gr4hip_fanin_create(C.byref(h), ident, 0, 1), "create")
    rk, nr = C.c_int(-1), C.c_int(-1)
    G.capi.check(L.gr4hip_fanin_rank(h, C.byref(rk), C.byref(nr)), "rank")
    assert (rk.value, nr.value) == (0, 1)
    x = torch.randn(4 * 8192, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for name in ("reduce_scatter", "all_to_all", "all_reduce"):
        out = torch.zeros_like(x)
        if name == "reduce_scatter":
            G.capi.check(L.gr4hip_fanin_reduce_scatter_sum_f32(h, x.data_ptr(), out.data_ptr(), x.numel(), st), name)
        elif name == "all_to_all":
            scratch = torch.empty_like(x)
            G.capi.check(L.gr4hip_fanin_all_to_all_sum_f32(h, x.data_ptr(), scratch.data_ptr(), out.data_ptr(), x.numel(), st), name)
        else:
            G.capi.check(L.gr4hip_fanin_all_reduce_sum_f32(h, x.data_ptr(), out.data_ptr(), x.numel(), st), name)
        torch.cuda.synchronize()
        assert torch.equal(out, x), name
    G.capi.check(L.gr4hip_fanin_destroy(h), "destroy")
