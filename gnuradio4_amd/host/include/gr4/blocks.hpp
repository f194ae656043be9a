// gr4/blocks.hpp -- block definitions on the hot path, written against gr4/core.hpp in the reference's block style.
//
// Each block states the reference block it mirrors (settings names, port names, semantics, error behaviour).  The processOne /
// processBulk bodies are the host ("compute_domain = host") path used by BASELINE configs[0] (CPU scheduler plumbing); with
// compute_domain = "gpu:hip[:i]" the work loop dispatches to gr::hip::Kernel<Block> (gr4/hip.hpp) instead.
#pragma once
#include <cmath>
#include <numbers>

#include "../../../../include/gr4hip.h" // host-side design functions only (gr4hip_fir_design / gr4hip_iir_design / gr4hip_window_create: no device needed)
#include "core.hpp"

namespace gr::testing {

// settable-values source with a sample budget: the role of TagSource<T> (blocks/testing/.../TagMonitors.hpp:134-293) without tags
template <typename T>
struct VectorSource : Block<VectorSource<T>> {
    PortOut<T>     out;
    std::vector<T> values{};      // repeated cyclically when shorter than n_samples_max
    Size_t         n_samples_max = 0; // 0: exactly values.size() samples
    std::size_t    _produced     = 0;
    std::vector<Tag> _tags{};     // tags to emit, ascending index (TagSource::_tags)
    GR_MAKE_REFLECTABLE(VectorSource, out, values, n_samples_max);

    work::Result customWork(std::size_t requested) {
        const std::size_t total = n_samples_max ? n_samples_max : values.size();
        if (_produced >= total) return {requested, 0, work::Status::DONE};
        if (!out.connected()) return {requested, 0, work::Status::ERROR};
        const std::size_t n = std::min({total - _produced, out.buffer->free_space(), requested});
        if (n == 0) return {requested, 0, work::Status::INSUFFICIENT_OUTPUT_ITEMS};
        auto span = out.buffer->write_span(n);
        if (values.empty()) std::fill(span.begin(), span.end(), T{});
        else
            for (std::size_t i = 0; i < n;) { // whole runs of the cyclic pattern at a time
                const std::size_t at = (_produced + i) % values.size(), run = std::min(n - i, values.size() - at);
                std::copy_n(values.begin() + static_cast<std::ptrdiff_t>(at), run, span.begin() + static_cast<std::ptrdiff_t>(i));
                i += run;
            }
        for (const Tag& t : _tags)
            if (t.index >= _produced && t.index < _produced + n) out.buffer->publishTag(t.map, t.index - _produced);
        out.buffer->publish(n);
        _produced += n;
        return {requested, n, work::Status::OK};
    }
};

// records everything it receives: TagSink<T> (TagMonitors.hpp:391-479) without tags
template <typename T>
struct VectorSink : Block<VectorSink<T>> {
    PortIn<T>      in;
    Size_t         n_samples_expected = 0;
    std::vector<T> _samples;
    std::vector<Tag> _tags; // received tags with their absolute sample index (TagSink::_tags)
    GR_MAKE_REFLECTABLE(VectorSink, in, n_samples_expected);

    work::Result customWork(std::size_t requested) {
        if (!in.connected()) return {requested, 0, work::Status::ERROR};
        const std::size_t n = std::min(in.buffer->available(), requested);
        if (n == 0) return {requested, 0, in.buffer->done() ? work::Status::DONE : work::Status::INSUFFICIENT_INPUT_ITEMS};
        auto span = in.buffer->read_span(n);
        for (const Tag& t : in.buffer->tags)
            if (t.index < in.buffer->read_pos + n) _tags.push_back(t);
        _samples.insert(_samples.end(), span.begin(), span.end());
        in.buffer->consume(n);
        return {requested, n, work::Status::OK};
    }
};

// NullSink<T> / CountingSink<T> (blocks/testing/.../NullSources.hpp): consumes and counts
template <typename T>
struct NullSink : Block<NullSink<T>> {
    PortIn<T>   in;
    std::size_t _count = 0;
    GR_MAKE_REFLECTABLE(NullSink, in);
    work::Result customWork(std::size_t requested) {
        if (!in.connected()) return {requested, 0, work::Status::ERROR};
        const std::size_t n = std::min(in.buffer->available(), requested);
        if (n == 0) return {requested, 0, in.buffer->done() ? work::Status::DONE : work::Status::INSUFFICIENT_INPUT_ITEMS};
        in.buffer->consume(n);
        _count += n;
        return {requested, n, work::Status::OK};
    }
};
} // namespace gr::testing

namespace gr::basic {
namespace signal_generator {
enum class Type : int { Const, Sin, Cos, Square, Saw, Triangle, FastSin, FastCos, UniformNoise, TriangularNoise, GaussianNoise }; // SignalGeneratorCore.hpp:16
inline bool gr_enum_parse(Type& d, std::string_view s) {
    return gr::detail::enum_from_names(d, s, std::array<std::string_view, 11>{"Const", "Sin", "Cos", "Square", "Saw", "Triangle", "FastSin", "FastCos", "UniformNoise", "TriangularNoise", "GaussianNoise"});
}

// xoshiro256++ with the reference's float conversions (algorithm/.../rng/Xoshiro256pp.hpp:22-96: splitmix64 seeding :33-39, next :41-52, 24 / 53 mantissa
// bits -> [0, 1) :55-61) and its Marsaglia-polar Gaussian with the cached second variate (GaussianNoise.hpp:33-55)
template <typename F>
struct Noise {
    std::uint64_t s[4]{};
    F             spare{};
    bool          has_spare = false;
    void          seed(std::uint64_t v) {
        for (auto& w : s) { // splitmix64
            v += 0x9e3779b97f4a7c15ULL;
            std::uint64_t z = v;
            z               = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
            z               = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
            w               = z ^ (z >> 31);
        }
        has_spare = false;
    }
    static constexpr std::uint64_t rotl(std::uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    std::uint64_t next() {
        const std::uint64_t r = rotl(s[0] + s[3], 23) + s[0], t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
        s[2] ^= t;
        s[3] = rotl(s[3], 45);
        return r;
    }
    F u01() {
        if constexpr (std::is_same_v<F, float>) return static_cast<F>(next() >> 40) * F(0x1.0p-24);
        else return static_cast<F>(next() >> 11) * F(0x1.0p-53);
    }
    F um11() { return F(2) * u01() - F(1); }
    F triangular() { const F a = u01(), b = u01(); return a + b - F(1); }
    F gauss() {
        if (has_spare) { has_spare = false; return spare; }
        F u, v, q;
        do { u = um11(); v = um11(); q = u * u + v * v; } while (q >= F(1) || q == F(0));
        const F f = std::sqrt(F(-2) * std::log(q) / q);
        spare     = v * f;
        has_spare = true;
        return u * f;
    }
};
} // namespace signal_generator

// gr::basic::SignalGenerator<T> (blocks/basic/.../SignalGenerator.hpp:25-87) over SignalGeneratorCore<T> (algorithm/.../signal/SignalGeneratorCore.hpp:72-147),
// free-running with a sample budget instead of the wall-clock pacing.  All eleven signal types, sample by sample as the block does (generateSample):
//   tones (ToneGenerator.hpp): time accumulates by 1 / sample_rate in the compute type F (double for scalar T, the base type for complex T); Sin / Cos by
//   std::sin / std::cos of 2 pi f t + phase, FastSin / FastCos by a phasor rotated once per sample and renormalised every 65536 samples, Square / Saw /
//   Triangle from the cycle count f t + phase / 2 pi; frequency <= 0 turns every tone into Const; complex output = the analytic signal for the four
//   sinusoids (offset on the real part), {value, 0} otherwise
//   noise (NoiseGenerator.hpp): xoshiro256++ seeded with `seed`; Uniform [-1, 1), Triangular u1 + u2 - 1, Gaussian by Marsaglia's polar method; complex:
//   independent components (offset on the real part), Gaussian components scaled by 1 / sqrt 2
//   integer T: truncation, clamped to the type's range (clampToInt, SignalGeneratorCore.hpp:47-58)
template <typename T>
struct SignalGenerator : Block<SignalGenerator<T>> {
    using F = std::conditional_t<std::is_same_v<T, std::complex<float>>, float, double>;
    PortOut<T>             out;
    signal_generator::Type signal_type = signal_generator::Type::Sin;
    float                  sample_rate = 1000.f, frequency = 1.f, amplitude = 1.f, offset = 0.f, phase = 0.f;
    std::uint64_t          seed          = 0;
    Size_t                 n_samples_max = 0;
    std::size_t            _n            = 0;
    GR_MAKE_REFLECTABLE(SignalGenerator, out, signal_type, sample_rate, frequency, amplitude, offset, phase, seed, n_samples_max);

    // ---- the core's state
    signal_generator::Type     _type = signal_generator::Type::Sin;
    F                          _t = 0, _tick = 0, _omega = 0, _cycles0 = 0, _f = 1, _a = 1, _o = 0, _ph = 0;
    std::complex<F>            _phasor{1, 0}, _rot{1, 0};
    std::size_t                _count = 0;
    signal_generator::Noise<F> _noise;
    bool                       _configured = false;

    void configure() {
        using signal_generator::Type;
        constexpr F pi2 = F(2) * std::numbers::pi_v<F>;
        _f = static_cast<F>(frequency); _a = static_cast<F>(amplitude); _o = static_cast<F>(offset); _ph = static_cast<F>(phase);
        _tick    = F(1) / static_cast<F>(sample_rate);
        _type    = signal_type;
        if (static_cast<int>(_type) <= static_cast<int>(Type::FastCos) && _f <= F(0)) _type = Type::Const;
        _omega   = pi2 * _f;
        _cycles0 = _ph / pi2;
        _rot     = {std::cos(pi2 * _f * _tick), std::sin(pi2 * _f * _tick)};
        _phasor  = {std::cos(_ph), std::sin(_ph)};
        _count   = 0;
        _noise.seed(seed);
        _configured = true;
    }
    void settingsChanged(const property_map&, const property_map&) { configure(); } // (the time base keeps running across a settings change, as upstream)
    void reset() { _t = 0; configure(); }

    [[nodiscard]] F tone() const {
        using signal_generator::Type;
        const F theta = _omega * _t + _ph, cycle = _f * _t + _cycles0;
        switch (_type) {
        case Type::Sin: return _a * std::sin(theta) + _o;
        case Type::Cos: return _a * std::cos(theta) + _o;
        case Type::FastSin: return _a * _phasor.imag() + _o;
        case Type::FastCos: return _a * _phasor.real() + _o;
        case Type::Square: return (cycle - std::floor(cycle) < F(0.5)) ? _a + _o : -_a + _o;
        case Type::Saw: return _a * (F(2) * (cycle - std::floor(cycle + F(0.5)))) + _o;
        case Type::Triangle: return _a * (F(4) * std::abs(cycle - std::floor(cycle + F(0.75)) + F(0.25)) - F(1)) + _o;
        default: return _a + _o;
        }
    }
    void advance() {
        using signal_generator::Type;
        _t += _tick;
        if (_type == Type::FastSin || _type == Type::FastCos) {
            _phasor *= _rot;
            if ((++_count & 0xFFFF) == 0) { const F inv = F(1) / std::abs(_phasor); _phasor = {_phasor.real() * inv, _phasor.imag() * inv}; }
        }
    }
    [[nodiscard]] F noise() {
        using signal_generator::Type;
        return _type == Type::UniformNoise ? _noise.um11() : _type == Type::TriangularNoise ? _noise.triangular() : _noise.gauss();
    }
    [[nodiscard]] T generateSample() {
        using signal_generator::Type;
        if (!_configured) configure();
        const bool is_tone = static_cast<int>(_type) <= static_cast<int>(Type::FastCos);
        if constexpr (gr::detail::is_complex<T>::value) {
            T r;
            if (is_tone) {
                const F theta = _omega * _t + _ph;
                switch (_type) {
                case Type::Sin: r = T(_a * std::sin(theta) + _o, -_a * std::cos(theta)); break;
                case Type::Cos: r = T(_a * std::cos(theta) + _o, _a * std::sin(theta)); break;
                case Type::FastSin: r = T(_a * _phasor.imag() + _o, -_a * _phasor.real()); break;
                case Type::FastCos: r = T(_a * _phasor.real() + _o, _a * _phasor.imag()); break;
                default: r = T(tone(), F(0)); break;
                }
                advance();
            } else if (_type == Type::GaussianNoise) {
                constexpr F scale = F(1) / std::numbers::sqrt2_v<F>;
                const F     g1 = _noise.gauss() * scale, g2 = _noise.gauss() * scale;
                r = T(_a * g1 + _o, _a * g2);
            } else {
                const F n1 = noise(), n2 = noise();
                r = T(_a * n1 + _o, _a * n2);
            }
            return r;
        } else {
            F raw;
            if (is_tone) { raw = tone(); advance(); }
            else raw = _a * noise() + _o;
            if constexpr (std::is_integral_v<T>) {
                if (raw >= static_cast<F>(std::numeric_limits<T>::max())) return std::numeric_limits<T>::max();
                if (raw <= static_cast<F>(std::numeric_limits<T>::min())) return std::numeric_limits<T>::min();
                return static_cast<T>(raw);
            } else {
                return static_cast<T>(raw);
            }
        }
    }

    work::Result customWork(std::size_t requested) {
        if (n_samples_max && _n >= n_samples_max) return {requested, 0, work::Status::DONE};
        if (!out.connected()) return {requested, 0, work::Status::ERROR};
        std::size_t n = std::min(out.buffer->free_space(), requested);
        if (n_samples_max) n = std::min<std::size_t>(n, n_samples_max - _n);
        if (n == 0) return {requested, 0, work::Status::INSUFFICIENT_OUTPUT_ITEMS};
        auto span = out.buffer->write_span(n);
        for (std::size_t i = 0; i < n; ++i) span[i] = generateSample();
        out.buffer->publish(n);
        _n += n;
        return {requested, n, work::Status::OK};
    }
};
} // namespace gr::basic

namespace gr::filter {

// gr::filter::fir_filter<T> (blocks/filter/.../time_domain_filter.hpp:22-48): setting `b`, y[n] = sum_k b[k] x[n-k], zero initial
// history.  T = float / double as registered upstream, plus std::complex<float> (complex data, real taps).
template <typename T>
struct fir_filter : Block<fir_filter<T>> {
    using tap_type = std::conditional_t<detail::is_complex<T>::value, float, T>;
    PortIn<T>             in;
    PortOut<T>            out;
    Tensor<tap_type>      b{tap_type(1)}; // feedforward coefficients: the reference's settings type (time_domain_filter.hpp:32); a std::vector in a property_map is accepted
    std::vector<T>        _history = std::vector<T>(32, T{}); // newest first, like HistoryBuffer{32}
    GR_MAKE_REFLECTABLE(fir_filter, in, out, b);

    void settingsChanged(const property_map& /*oldSettings*/, const property_map& newSettings) {
        if (newSettings.contains("b") && b.size() > _history.size()) { // the reference replaces the HistoryBuffer only when it must grow (:38-42)
            std::size_t cap = 1;
            while (cap < b.size()) cap <<= 1;
            _history.assign(cap, T{});
        }
    }
    [[nodiscard]] T processOne(T input) noexcept {
        std::move_backward(_history.begin(), _history.end() - 1, _history.end());
        _history[0] = input;
        T acc{};
        for (std::size_t k = 0; k < b.size(); ++k) acc += b[k] * _history[k];
        return acc;
    }
};

enum class IIRForm { DF_I, DF_II, DF_I_TRANSPOSED, DF_II_TRANSPOSED }; // time_domain_filter.hpp:50-55

// gr::filter::iir_filter<T, form> (time_domain_filter.hpp:62-122); a[0] is assumed to be 1
template <typename T, IIRForm form = IIRForm::DF_II>
struct iir_filter : Block<iir_filter<T, form>> {
    PortIn<T>      in;
    PortOut<T>     out;
    Tensor<T>      b{T(1)}, a{T(1)}; // time_domain_filter.hpp:73-74
    std::vector<T> _x = std::vector<T>(32, T{}), _y = std::vector<T>(32, T{});
    GR_MAKE_REFLECTABLE(iir_filter, in, out, b, a);

    static void push(std::vector<T>& h, T v) {
        std::move_backward(h.begin(), h.end() - 1, h.end());
        h[0] = v;
    }
    static T dot(const Tensor<T>& c, std::size_t first, const std::vector<T>& h) {
        T acc{};
        for (std::size_t k = first; k < c.size(); ++k) acc += c[k] * h[k - first];
        return acc;
    }
    void settingsChanged(const property_map&, const property_map&) {
        const std::size_t n = std::max(a.size(), b.size());
        if (n >= _x.size()) { _x.assign(2 * n, T{}); _y.assign(2 * n, T{}); }
    }
    [[nodiscard]] T processOne(T input) noexcept {
        if constexpr (form == IIRForm::DF_I) {
            push(_x, input);
            const T o = dot(b, 0, _x) - dot(a, 1, _y);
            push(_y, o);
            return o;
        } else if constexpr (form == IIRForm::DF_II) {
            const T w = input - dot(a, 1, _x);
            push(_x, w);
            return dot(b, 0, _x);
        } else if constexpr (form == IIRForm::DF_I_TRANSPOSED) {
            const T v0 = input - dot(a, 1, _y);
            push(_y, v0);
            return dot(b, 0, _y);
        } else {
            const T o = b[0] * input + dot(b, 1, _x) - dot(a, 1, _y);
            push(_x, input);
            push(_y, o);
            return o;
        }
    }
};

// gr::filter::Decimator<T> (time_domain_filter.hpp:215-245): keep every decim-th sample
template <typename T>
struct Decimator : Block<Decimator<T>, Resampling<1, 1, false>> {
    PortIn<T>  in;
    PortOut<T> out;
    Size_t     decim = 1;
    GR_MAKE_REFLECTABLE(Decimator, in, out, decim);
    void settingsChanged(const property_map&, const property_map&) { this->input_chunk_size = decim; }
    [[nodiscard]] work::Status processBulk(std::span<const T> input, std::span<T> output) noexcept {
        std::size_t o = 0;
        for (std::size_t i = 0; i < input.size(); ++i)
            if (i % decim == 0) output[o++] = input[i];
        return work::Status::OK;
    }
};

// ---- BasicFilterProto<T, Args...> (time_domain_filter.hpp:126-210): FIR / IIR by specification, optionally decimating
enum class FilterType { FIR, IIR };                                   // time_domain_filter.hpp:127
enum class Type { LOWPASS, HIGHPASS, BANDPASS, BANDSTOP };            // filter::Type (FilterTool.hpp:64) == gr4hip_filter_response
namespace iir { enum class Design { BUTTERWORTH, BESSEL, CHEBYSHEV1, CHEBYSHEV2 }; // FilterTool.hpp:425-430 == gr4hip_iir_design_t
inline bool gr_enum_parse(Design& d, std::string_view s) { return gr::detail::enum_from_names(d, s, std::array<std::string_view, 4>{"BUTTERWORTH", "BESSEL", "CHEBYSHEV1", "CHEBYSHEV2"}); }
}
inline bool gr_enum_parse(FilterType& d, std::string_view s) { return gr::detail::enum_from_names(d, s, std::array<std::string_view, 2>{"FIR", "IIR"}); }
inline bool gr_enum_parse(Type& d, std::string_view s) { return gr::detail::enum_from_names(d, s, std::array<std::string_view, 4>{"LOWPASS", "HIGHPASS", "BANDPASS", "BANDSTOP"}); }
} // namespace gr::filter
namespace gr::algorithm::window {
enum class Type : int { None, Rectangular, Hamming, Hann, HannExp, Blackman, Nuttall, BlackmanHarris, BlackmanNuttall, FlatTop, Exponential, Kaiser }; // window.hpp:35 == GR4HIP_WIN_*
inline constexpr std::array<std::string_view, 12> TypeList{"None", "Rectangular", "Hamming", "Hann", "HannExp", "Blackman", "Nuttall", "BlackmanHarris", "BlackmanNuttall", "FlatTop", "Exponential", "Kaiser"};
inline bool gr_enum_parse(Type& d, std::string_view s) { return gr::detail::enum_from_names(d, s, TypeList); }
// window::create (window.hpp:69-183) through the library's host-side restatement
template <typename T>
std::vector<T> create(Type type, std::size_t n, float beta = 1.6f) {
    if constexpr (std::is_same_v<T, double>) { // create<double>: evaluated in double
        std::vector<double> w(n);
        if (n && gr4hip_window_create_f64(static_cast<int>(type), w.data(), n, static_cast<double>(beta)) != GR4HIP_OK) throw std::invalid_argument(std::string("window::create: ") + gr4hip_last_error());
        return w;
    } else {
        std::vector<float> w(n);
        if (n && gr4hip_window_create(static_cast<int>(type), w.data(), n, beta) != GR4HIP_OK) throw std::invalid_argument(std::string("window::create: ") + gr4hip_last_error());
        return std::vector<T>(w.begin(), w.end());
    }
}
} // namespace gr::algorithm::window
namespace gr::filter {

// designed coefficients: FIR = one section with a = {1}; IIR = biquad (or first-order) sections, applied as a cascade of DF_II sections
// (Filter<T>::processOne folds over the sections, FilterTool.hpp:223-247; DF_II is the float default :220)
struct DesignedFilter {
    bool                            fir = true;
    std::vector<float>              taps;     // FIR
    std::vector<std::array<float, 3>> b, a;   // IIR sections (zero padded to order 2)
};
inline DesignedFilter designFilter(FilterType ft, Type response, std::size_t order, double fLow, double fHigh, double fs, iir::Design iirDesign, algorithm::window::Type firWindow) {
    gr4hip_filter_params p;
    gr4hip_filter_params_default(&p);
    p.order = order; p.f_low = fLow; p.f_high = fHigh; p.fs = fs;
    DesignedFilter d;
    d.fir = ft == FilterType::FIR;
    const auto fail = [](const char* what) { throw std::invalid_argument(std::string(what) + ": " + gr4hip_last_error()); };
    if (d.fir) {
        std::size_t n = 0;
        if (gr4hip_fir_design(static_cast<int>(response), &p, static_cast<int>(firWindow), nullptr, 0, &n) != GR4HIP_OK) fail("fir::designFilter");
        d.taps.resize(n);
        if (gr4hip_fir_design(static_cast<int>(response), &p, static_cast<int>(firWindow), d.taps.data(), n, &n) != GR4HIP_OK) fail("fir::designFilter");
    } else {
        std::size_t        cap = 2 * order + 2, n = 0;
        std::vector<float> fb(3 * cap), fa(3 * cap);
        if (gr4hip_iir_design(static_cast<int>(response), &p, static_cast<int>(iirDesign), fb.data(), fa.data(), cap, &n) != GR4HIP_OK) fail("iir::designFilter");
        for (std::size_t s = 0; s < n; ++s) {
            d.b.push_back({fb[3 * s], fb[3 * s + 1], fb[3 * s + 2]});
            d.a.push_back({fa[3 * s], fa[3 * s + 1], fa[3 * s + 2]});
        }
    }
    return d;
}

template <typename T, typename... Args>
struct BasicFilterProto : Block<BasicFilterProto<T, Args...>, Args...> {
    using TParent = Block<BasicFilterProto<T, Args...>, Args...>;
    PortIn<T>  in;
    PortOut<T> out;
    Annotated<FilterType, "filter_type">                  filter_type     = FilterType::IIR;
    Annotated<Type, "filter_response">                    filter_response = Type::LOWPASS;
    Annotated<Size_t, "filter_order">                     filter_order{3};
    Annotated<float, "f_low">                             f_low{0.1f};
    Annotated<float, "f_high">                            f_high{0.2f};
    Annotated<float, "sample rate">                       sample_rate{1.0f};
    Annotated<Size_t, "decimation factor">                decimate{1U};
    Annotated<iir::Design, "iir_design_method">           iir_design_method = iir::Design::BUTTERWORTH;
    Annotated<algorithm::window::Type, "fir_design_method"> fir_design_method = algorithm::window::Type::Kaiser;
    GR_MAKE_REFLECTABLE(BasicFilterProto, in, out, filter_type, filter_response, filter_order, f_low, f_high, sample_rate, decimate, iir_design_method, fir_design_method);

    DesignedFilter              _design;
    std::vector<T>              _fir_hist;             // newest first
    std::vector<std::array<T, 2>> _w;                  // DF_II state per section
    bool                        _designed = false;

    void settingsChanged(const property_map&, const property_map&) { designFilter(); }
    void designFilter() { // time_domain_filter.hpp:163-182
        if constexpr (!TParent::ResamplingControl::kIsConst) this->input_chunk_size = decimate;
        _design = gr::filter::designFilter(filter_type, filter_response, filter_order, static_cast<double>(f_low.value), static_cast<double>(f_high.value), static_cast<double>(sample_rate.value), iir_design_method, fir_design_method);
        _fir_hist.assign(_design.taps.size(), T{});
        _w.assign(_design.b.size(), std::array<T, 2>{});
        _designed = true;
    }
    [[nodiscard]] T filterOne(T x) noexcept {
        if (_design.fir) {
            if (_fir_hist.empty()) return T{};
            std::move_backward(_fir_hist.begin(), _fir_hist.end() - 1, _fir_hist.end());
            _fir_hist[0] = x;
            T acc{};
            for (std::size_t k = 0; k < _fir_hist.size(); ++k) acc += static_cast<T>(_design.taps[k]) * _fir_hist[k];
            return acc;
        }
        for (std::size_t s = 0; s < _w.size(); ++s) { // DF_II section (FilterTool.hpp:127-135)
            const auto& b = _design.b[s];
            const auto& a = _design.a[s];
            const T w0 = x - static_cast<T>(a[1]) * _w[s][0] - static_cast<T>(a[2]) * _w[s][1];
            x          = static_cast<T>(b[0]) * w0 + static_cast<T>(b[1]) * _w[s][0] + static_cast<T>(b[2]) * _w[s][1];
            _w[s][1]   = _w[s][0];
            _w[s][0]   = w0;
        }
        return x;
    }
    [[nodiscard]] T processOne(T input) noexcept
    requires(TParent::ResamplingControl::kIsConst)
    {
        if (!_designed) designFilter();
        return filterOne(input);
    }
    [[nodiscard]] work::Status processBulk(std::span<const T> input, std::span<T> output) noexcept
    requires(!TParent::ResamplingControl::kIsConst)
    { // full-rate filter, keep the samples with i % decimate == 0 (time_domain_filter.hpp:190-204)
        if (!_designed) designFilter();
        std::size_t o = 0;
        for (std::size_t i = 0; i < input.size(); ++i) {
            const T y = filterOne(input[i]);
            if (i % decimate == 0) output[o++] = y;
        }
        return work::Status::OK;
    }
};
template <typename T> using BasicFilter           = BasicFilterProto<T>;
template <typename T> using BasicDecimatingFilter = BasicFilterProto<T, Resampling<1, 1, false>>;

// ---- interpolating FIR (BASELINE.json north_star).  The reference has no such block, only the rate declaration it would carry --
// Resampling<1, L> (annotated.hpp:121-128; chunk bookkeeping Block.hpp:1576-1636) -- so this follows fir_filter's shape (settings `b`, plus `interpolate`)
// and SURVEY.md Appendix A's definition: zero-stuff by L, fir_filter's sum at the output rate, gain L.  Host body: the polyphase form
// y[m L + p] = L sum_q b[q L + p] x[m - q]; output_chunk_size = interpolate, so gr:sample_rate is forwarded times L (Block.hpp:1088-1099).
template <typename T>
struct fir_interpolator : Block<fir_interpolator<T>, Resampling<1, 1, false>> {
    PortIn<T>          in;
    PortOut<T>         out;
    std::vector<float> b{1.f};
    Size_t             interpolate = 1;
    GR_MAKE_REFLECTABLE(fir_interpolator, in, out, b, interpolate);
    std::vector<T> _hist; // newest first, ceil(K / L) input samples

    void settingsChanged(const property_map&, const property_map&) {
        if (interpolate == 0) throw std::invalid_argument("fir_interpolator: interpolate must be >= 1");
        this->output_chunk_size = interpolate;
        const std::size_t kp = (b.size() + interpolate - 1) / interpolate;
        if (kp > _hist.size()) _hist.assign(kp, T{}); // like fir_filter: the history is replaced only when it must grow
    }
    [[nodiscard]] work::Status processBulk(std::span<const T> input, std::span<T> output) noexcept {
        const std::size_t L = interpolate, K = b.size(), kp = (K + L - 1) / L;
        if (_hist.size() < kp) _hist.assign(kp, T{});
        for (std::size_t m = 0; m < input.size(); ++m) {
            std::move_backward(_hist.begin(), _hist.end() - 1, _hist.end());
            _hist[0] = input[m];
            for (std::size_t p = 0; p < L; ++p) {
                T acc{};
                for (std::size_t q = 0; q * L + p < K; ++q) acc += static_cast<T>(b[q * L + p]) * _hist[q];
                output[m * L + p] = static_cast<T>(static_cast<float>(L)) * acc;
            }
        }
        return work::Status::OK;
    }
};
} // namespace gr::filter

namespace gr::blocks::math {
// MathOpImpl<T, op> (blocks/math/.../Math.hpp:30-57): out = in (op) value, default value 1
template <typename T, typename op>
struct MathOpImpl : Block<MathOpImpl<T, op>> {
    PortIn<T>  in;
    PortOut<T> out;
    T          value = T(1);
    GR_MAKE_REFLECTABLE(MathOpImpl, in, out, value);
    [[nodiscard]] constexpr T processOne(const T& a) const noexcept { return static_cast<T>(op()(a, value)); }
};
template <typename T> using AddConst      = MathOpImpl<T, std::plus<T>>;
template <typename T> using SubtractConst = MathOpImpl<T, std::minus<T>>;
template <typename T> using MultiplyConst = MathOpImpl<T, std::multiplies<T>>;
template <typename T> using DivideConst   = MathOpImpl<T, std::divides<T>>;

// MathOpMultiPortImpl<T, op> (Math.hpp:73-108): left fold over n_inputs (1..32) streams
template <typename T, typename op>
struct MathOpMultiPortImpl : Block<MathOpMultiPortImpl<T, op>> {
    std::vector<PortIn<T>> in;
    PortOut<T>             out;
    Size_t                 n_inputs = 0;
    GR_MAKE_REFLECTABLE(MathOpMultiPortImpl, in, out, n_inputs);
    void settingsChanged(const property_map&, const property_map& newSettings) {
        if (newSettings.contains("n_inputs")) {
            if (n_inputs < 1 || n_inputs > 32) throw std::invalid_argument("n_inputs must be in [1, 32]"); // Limits<1U, 32U> (Math.hpp:90)
            in.resize(n_inputs);
        }
    }
    work::Status processBulk(std::span<const std::span<const T>> ins, std::span<T> sout) const {
        std::copy(ins[0].begin(), ins[0].end(), sout.begin());
        for (std::size_t n = 1; n < ins.size(); ++n)
            std::transform(sout.begin(), sout.end(), ins[n].begin(), sout.begin(), [](T x, T y) { return static_cast<T>(op{}(x, y)); });
        return work::Status::OK;
    }
};
template <typename T> using Add      = MathOpMultiPortImpl<T, std::plus<T>>;
template <typename T> using Subtract = MathOpMultiPortImpl<T, std::minus<T>>;
template <typename T> using Multiply = MathOpMultiPortImpl<T, std::multiplies<T>>;
template <typename T> using Divide   = MathOpMultiPortImpl<T, std::divides<T>>;

// Rotator<std::complex<T>> (blocks/math/.../Rotator.hpp:16-63): XOR settings frequency_shift / phase_increment
template <typename T>
struct Rotator : Block<Rotator<T>> {
    using value_type = typename T::value_type;
    PortIn<T>  in;
    PortOut<T> out;
    float      sample_rate = 1.f, frequency_shift = 0.f;
    value_type phase_increment{0}, initial_phase{0};
    value_type _accumulated_phase{0};
    GR_MAKE_REFLECTABLE(Rotator, in, out, sample_rate, frequency_shift, initial_phase, phase_increment);
    void settingsChanged(const property_map&, const property_map& n) {
        const bool f = n.contains("frequency_shift"), p = n.contains("phase_increment");
        if (f && p) throw std::invalid_argument("cannot set both 'frequency_shift' and 'phase_increment' in new setting (XOR)");
        if (f) phase_increment = value_type(2) * static_cast<value_type>(std::numbers::pi_v<float> * frequency_shift / sample_rate);
        else if (p) frequency_shift = static_cast<float>(phase_increment / (value_type(2) * std::numbers::pi_v<value_type>)) * sample_rate;
        _accumulated_phase = initial_phase;
    }
    [[nodiscard]] T processOne(const T& x) noexcept {
        _accumulated_phase += phase_increment;
        if (_accumulated_phase > value_type(2) * std::numbers::pi_v<value_type>) _accumulated_phase -= value_type(2) * std::numbers::pi_v<value_type>;
        else if (_accumulated_phase < value_type(0)) _accumulated_phase += value_type(2) * std::numbers::pi_v<value_type>;
        return x * T(std::cos(_accumulated_phase), std::sin(_accumulated_phase));
    }
};
} // namespace gr::blocks::math

namespace gr::blocks::fft {
// Streaming |FFT|^2: one frame of fftSize complex samples in, fftSize floats (natural bin order) out.  The streaming counterpart of
// blocks::fft::FFT<T> (blocks/fourier/.../fft.hpp:31-251), which emits a DataSet per frame: mag2[(k + N/2) mod N] equals
// (DataSet magnitude[k] * N/2)^2.  Settings follow the FFT block: fftSize, window ("None" or "Hann").
template <typename T>
struct PowerSpectrum : Block<PowerSpectrum<T>, Resampling<1024, 1024, false>> {
    using value_type = typename T::value_type;
    PortIn<T>           in;
    PortOut<value_type> out;
    Size_t              fftSize = 1024;
    std::string         window  = "None";
    std::vector<value_type> _window;
    GR_MAKE_REFLECTABLE(PowerSpectrum, in, out, fftSize, window);

    void settingsChanged(const property_map&, const property_map&) {
        if (fftSize < 2 || (fftSize & (fftSize - 1))) throw std::invalid_argument("fftSize must be a power of two");
        this->input_chunk_size = this->output_chunk_size = fftSize; // fft.hpp:131-134: exactly whole frames
        _window.assign(fftSize, value_type(1));
        if (window == "Hann")
            for (std::size_t i = 0; i < fftSize; ++i) _window[i] = value_type(.5) - value_type(.5) * std::cos(value_type(2) * std::numbers::pi_v<value_type> / value_type(fftSize - 1) * value_type(i));
        else if (window != "None" && window != "Rectangular") _window.clear(); // the other ten windows exist on the device path only (gr4hip_window_create)
    }
    // host path: iterative radix-2 in double (plumbing only)
    work::Status processBulk(std::span<const T> input, std::span<value_type> output) {
        const std::size_t N = fftSize;
        if (_window.size() != N) settingsChanged({}, {});
        if (_window.size() != N) throw std::invalid_argument("PowerSpectrum: the host path implements None, Rectangular and Hann; window '" + window + "' needs compute_domain gpu:hip");
        std::vector<std::complex<double>> v(N);
        for (std::size_t f = 0; f + N <= input.size(); f += N) {
            for (std::size_t i = 0, j = 0; i < N; ++i) {
                v[j] = std::complex<double>(input[f + i]) * static_cast<double>(_window[i]);
                std::size_t m = N >> 1;
                while (m >= 1 && (j & m)) { j ^= m; m >>= 1; }
                j |= m;
            }
            for (std::size_t len = 2; len <= N; len <<= 1)
                for (std::size_t k = 0; k < N; k += len)
                    for (std::size_t n = 0; n < len / 2; ++n) {
                        const auto w = std::polar(1.0, -2.0 * std::numbers::pi * static_cast<double>(n) / static_cast<double>(len));
                        const auto t = v[k + n + len / 2] * w;
                        v[k + n + len / 2] = v[k + n] - t;
                        v[k + n] += t;
                    }
            for (std::size_t i = 0; i < N; ++i) output[f + i] = static_cast<value_type>(std::norm(v[i]));
        }
        return work::Status::OK;
    }
};

// ---- gr::blocks::fft::FFT<T> (blocks/fourier/.../fft.hpp:31-251): one DataSet per frame of fftSize samples
// window (default Hann) -> unnormalised forward DFT -> magnitude (hypot 2/N [dB], fft-shifted) + phase (atan2 [unwrap][deg], shifted) + Re + Im,
// frequency axis, per-signal min/max.  T = std::complex<float> (N bins) or float (N/2 bins: fft.hpp:140-143, 221-227).
// The host body is a float64 DFT (radix-2 for powers of two, the defining sum otherwise) -- plumbing; the device path is gr4hip_fft_process.
template <typename T, typename U = DataSet<float>>
struct FFT : Block<FFT<T, U>, Resampling<1024, 1, false>> {
    using value_type                         = typename U::value_type;
    static constexpr bool computeFullSpectrum = gr::detail::is_complex<T>::value;
    PortIn<T>                         in;
    PortOut<U, RequiredSamples<1, 1>> out;
    Annotated<Size_t, "FFT size">           fftSize{1024U};
    Annotated<std::string, "window type">   window = std::string("Hann");
    Annotated<bool, "output in dB">         outputInDb{false};
    Annotated<bool, "output in deg">        outputInDeg{false};
    Annotated<bool, "unwrap phase">         unwrapPhase{false};
    Annotated<float, "sample rate">         sample_rate = 1.f;
    Annotated<std::string, "signal name">   signal_name = std::string("unknown signal");
    Annotated<std::string, "signal unit">   signal_unit = std::string("a.u.");
    Annotated<float, "signal min">          signal_min  = -std::numeric_limits<float>::max();
    Annotated<float, "signal max">          signal_max  = +std::numeric_limits<float>::max();
    GR_MAKE_REFLECTABLE(FFT, in, out, fftSize, window, outputInDb, outputInDeg, unwrapPhase, sample_rate, signal_name, signal_unit, signal_min, signal_max);

    gr::algorithm::window::Type       _windowType = gr::algorithm::window::Type::Hann;
    std::vector<value_type>           _window;
    std::vector<std::complex<value_type>> _outData;
    std::vector<value_type>           _magnitudeSpectrum, _phaseSpectrum;

    void settingsChanged(const property_map&, const property_map& newSettings) {
        if (!newSettings.contains("fftSize") && !newSettings.contains("window") && !_window.empty()) return; // fft.hpp:126-129
        in.max_samples = in.min_samples = fftSize;
        this->input_chunk_size          = fftSize;
        gr::algorithm::window::Type t   = _windowType;
        if (gr_enum_parse(t, std::string_view(window.value))) _windowType = t; // enum_cast(...).value_or(_windowType) (:138)
        _window = gr::algorithm::window::create<value_type>(_windowType, fftSize);
    }
    [[nodiscard]] std::size_t nBins() const { return computeFullSpectrum ? fftSize.value : fftSize.value / 2; }

    [[nodiscard]] work::Status processBulk(std::span<const T> input, std::span<U> output) {
        const std::size_t N = fftSize;
        if (_window.size() != N) settingsChanged({}, {{"fftSize", std::int64_t(N)}});
        for (std::size_t f = 0; f < output.size(); ++f) {
            std::vector<std::complex<double>> v(N);
            for (std::size_t i = 0; i < N; ++i) v[i] = std::complex<double>(input[f * N + i]) * static_cast<double>(_window[i]);
            dft(v);
            _outData.assign(N, {});
            for (std::size_t i = 0; i < N; ++i) _outData[i] = std::complex<value_type>(static_cast<value_type>(v[i].real()), static_cast<value_type>(v[i].imag()));
            const std::size_t M = nBins();
            _magnitudeSpectrum.assign(M, 0);
            _phaseSpectrum.assign(M, 0);
            for (std::size_t k = 0; k < M; ++k) { // fft_common.hpp:20-56, 91-123
                const value_type mag = std::hypot(_outData[k].real(), _outData[k].imag()) * value_type(2) / static_cast<value_type>(N);
                _magnitudeSpectrum[k] = !outputInDb ? mag : mag > value_type(0) ? value_type(20) * std::log10(mag) : std::numeric_limits<value_type>::lowest();
                _phaseSpectrum[k]     = std::atan2(_outData[k].imag(), _outData[k].real());
            }
            if (unwrapPhase) { // fft_common.hpp:71-89
                const value_type pi = std::numbers::pi_v<value_type>;
                value_type prev = _phaseSpectrum.front();
                for (std::size_t k = 1; k < M; ++k) {
                    value_type& cur = _phaseSpectrum[k];
                    while (cur - prev > pi) cur -= 2 * pi;
                    while (cur - prev < -pi) cur += 2 * pi;
                    prev = cur;
                }
            }
            if (outputInDeg) for (auto& ph : _phaseSpectrum) ph = ph * value_type(180) * std::numbers::inv_pi_v<value_type>;
            if (computeFullSpectrum) {
                std::rotate(_magnitudeSpectrum.begin(), _magnitudeSpectrum.begin() + static_cast<std::ptrdiff_t>(M / 2), _magnitudeSpectrum.end());
                std::rotate(_phaseSpectrum.begin(), _phaseSpectrum.begin() + static_cast<std::ptrdiff_t>(M / 2), _phaseSpectrum.end());
            }
            output[f] = createDataset();
        }
        return work::Status::OK;
    }

    // the descriptive part of the DataSet (everything but signal_values / signal_ranges), fft.hpp:173-250
    [[nodiscard]] U datasetSkeleton() const {
        U ds{};
        const std::size_t N = nBins();
        ds.extents    = {static_cast<std::int32_t>(N)};
        ds.axis_names = {"Frequency"};
        ds.axis_units = {"Hz"};
        ds.axis_values.assign(1, std::vector<value_type>(N));
        const value_type freqWidth = static_cast<value_type>(sample_rate.value) / static_cast<value_type>(fftSize.value);
        const value_type freqOffset = computeFullSpectrum ? static_cast<value_type>(N / 2) * freqWidth : value_type(0);
        for (std::size_t i = 0; i < N; ++i) ds.axis_values[0][i] = static_cast<value_type>(i) * freqWidth - freqOffset;
        const std::string& n = signal_name.value;
        ds.signal_names      = {"Magnitude(" + n + ")", "Phase(" + n + ")", "Re(FFT(" + n + "))", "Im(FFT(" + n + "))"};
        ds.signal_quantities = {"Magnitude(FFT)", "Phase(FFT)", "Re(FFT)", "Im(FFT)"};
        ds.signal_units      = {signal_unit.value + "/\u221aHz", "rad", "Re" + signal_unit.value, "Im" + signal_unit.value};
        ds.signal_values.assign(4 * N, 0);
        ds.signal_ranges.assign(4, {});
        typename decltype(ds.meta_information)::value_type meta{{"sample_rate", sample_rate.value}, {"window", window.value}, {"output_in_db", outputInDb.value},
            {"output_in_deg", outputInDeg.value}, {"unwrap_phase", unwrapPhase.value}, {"input_chunk_size", std::uint64_t(this->input_chunk_size)},
            {"output_chunk_size", std::uint64_t(this->output_chunk_size)}};
        ds.meta_information.assign(4, meta);
        return ds;
    }
    [[nodiscard]] U createDataset() const {
        U                 ds = datasetSkeleton();
        const std::size_t N  = nBins();
        std::copy_n(_magnitudeSpectrum.begin(), N, ds.signalValues(0).begin());
        std::copy_n(_phaseSpectrum.begin(), N, ds.signalValues(1).begin());
        const auto spec = std::span<const std::complex<value_type>>(_outData).last(N); // real input: the upper half (fft.hpp:221-227)
        for (std::size_t i = 0; i < N; ++i) {
            ds.signalValues(2)[i] = spec[i].real();
            ds.signalValues(3)[i] = spec[i].imag();
        }
        for (std::size_t i = 0; i < 4; ++i) {
            const auto sv       = ds.signalValues(i);
            const auto mm       = std::minmax_element(sv.begin(), sv.end());
            ds.signal_ranges[i] = {*mm.first, *mm.second};
        }
        return ds;
    }

private:
    static void dft(std::vector<std::complex<double>>& v) {
        const std::size_t N = v.size();
        if (N && !(N & (N - 1))) {
            for (std::size_t i = 1, j = 0; i < N; ++i) {
                std::size_t bit = N >> 1;
                for (; j & bit; bit >>= 1) j ^= bit;
                j ^= bit;
                if (i < j) std::swap(v[i], v[j]);
            }
            for (std::size_t len = 2; len <= N; len <<= 1)
                for (std::size_t k = 0; k < N; k += len)
                    for (std::size_t n = 0; n < len / 2; ++n) {
                        const auto w = std::polar(1.0, -2.0 * std::numbers::pi * static_cast<double>(n) / static_cast<double>(len));
                        const auto t = v[k + n + len / 2] * w;
                        v[k + n + len / 2] = v[k + n] - t;
                        v[k + n] += t;
                    }
            return;
        }
        std::vector<std::complex<double>> o(N);
        for (std::size_t k = 0; k < N; ++k)
            for (std::size_t n = 0; n < N; ++n) o[k] += v[n] * std::polar(1.0, -2.0 * std::numbers::pi * static_cast<double>((k * n) % N) / static_cast<double>(N));
        v = std::move(o);
    }
};
} // namespace gr::blocks::fft
