// gr4/core.hpp -- the kept GNU Radio 4 surface: gr::Block<>, PortIn<>/PortOut<>, Graph::emplaceBlock/connect, scheduler::Simple.
//
// A from-scratch, C++20, header-only host layer that accepts block definitions written against the reference's API
// (core/include/gnuradio-4.0/Block.hpp:541-663 "struct X : gr::Block<X>", PortIn/PortOut members, GR_MAKE_REFLECTABLE,
// settingsChanged(old,new), processOne / processBulk(std::span<const T>, std::span<U>), Resampling<>), wires them with
// Graph::connect<"out","in">(a, b) (Graph.hpp:595-690; errors are returned, not thrown) and runs them with a single-threaded
// scheduler::Simple (Scheduler.hpp:1916-1952) following the work() chunking rules of Block.hpp:1950-2026.  It is NOT a port of the
// reference runtime: no messages / settings staging, no thread pool, no lock-free rings (one worker drives every block, so an
// edge is a plain compacting FIFO).  The compute_domain seam (Block.hpp:713, 1855-1862) is live: see gr4/hip.hpp.
#pragma once

#include <algorithm>
#include <array>
#include <cctype>
#include <complex>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <functional>
#include <limits>
#include <map>
#include <memory>
#include <memory_resource>
#include <mutex>
#include <optional>
#include <span>
#include <stdexcept>
#include <string>
#include <string_view>
#include <tuple>
#include <type_traits>
#include <typeindex>
#include <unordered_map>
#include <variant>
#include <vector>

namespace gr {

using Size_t = std::uint32_t;

// ---------------------------------------------------------------------------------------------- work status (WorkStatus.hpp:12-41)
namespace work {
enum class Status : int { ERROR = -100, INSUFFICIENT_OUTPUT_ITEMS = -3, INSUFFICIENT_INPUT_ITEMS = -2, DONE = -1, OK = 0 };
struct Result {
    std::size_t requested_work = 0, performed_work = 0;
    Status      status = Status::OK;
};
} // namespace work

// ---------------------------------------------------------------------------------------------- errors / expected (C++20 stand-in for std::expected)
struct Error {
    std::string message;
};
struct unexpected_t {
    Error error;
};
inline unexpected_t unexpected(std::string msg) { return {Error{std::move(msg)}}; }
template <typename T>
class expected {
    std::variant<T, Error> _v;

public:
    expected(T v) : _v(std::move(v)) {}
    expected(unexpected_t u) : _v(std::move(u.error)) {}
    [[nodiscard]] bool has_value() const noexcept { return _v.index() == 0; }
    explicit           operator bool() const noexcept { return has_value(); }
    [[nodiscard]] T&           value() { return std::get<0>(_v); }
    [[nodiscard]] const T&     value() const { return std::get<0>(_v); }
    [[nodiscard]] const Error& error() const { return std::get<1>(_v); }
};
template <>
class expected<void> {
    std::optional<Error> _e;

public:
    expected() = default;
    expected(unexpected_t u) : _e(std::move(u.error)) {}
    [[nodiscard]] bool has_value() const noexcept { return !_e.has_value(); }
    explicit           operator bool() const noexcept { return has_value(); }
    [[nodiscard]] const Error& error() const { return *_e; }
};

// ---------------------------------------------------------------------------------------------- gr::Tensor<T> (core/include/gnuradio-4.0/Tensor.hpp)
// The settings type of the reference's filter blocks (fir_filter::b, iir_filter::b / a: Tensor<T>, blocks/filter/.../time_domain_filter.hpp:32, 73-74).  The
// dynamic-extents managed form, row-major, as far as block code uses it: construction from values / a vector / extents, size / rank / extents, element and
// iterator access over the flat storage, comparison.  (Views, static extents and the mdspan bridge of the reference type are not part of the hot path.)
template <typename T>
struct Tensor {
    using value_type = T;
    std::vector<T>           _data;
    std::vector<std::size_t> _extents{0};

    Tensor() = default;
    Tensor(std::initializer_list<T> values) : _data(values), _extents{values.size()} {}                 // a rank-1 tensor of these values: Tensor<T> b{T{1}}
    Tensor(const std::vector<T>& values) : _data(values), _extents{values.size()} {}                    // NOLINT: a vector is a rank-1 tensor
    Tensor(std::vector<T>&& values) : _data(std::move(values)), _extents{_data.size()} {}               // NOLINT
    template <std::input_iterator It>
    Tensor(It first, It last) : _data(first, last), _extents{_data.size()} {}
    static Tensor with_extents(std::vector<std::size_t> extents, T fill = T{}) {
        Tensor t;
        std::size_t n = 1;
        for (auto e : extents) n *= e;
        t._data.assign(n, fill);
        t._extents = std::move(extents);
        return t;
    }
    Tensor& operator=(const std::vector<T>& values) { _data = values; _extents = {values.size()}; return *this; }
    Tensor& operator=(std::initializer_list<T> values) { _data = values; _extents = {values.size()}; return *this; }
    operator const std::vector<T>&() const noexcept { return _data; } // NOLINT: flat storage, for code that takes the std::vector spelling of the setting

    [[nodiscard]] std::size_t size() const noexcept { return _data.size(); }
    [[nodiscard]] bool        empty() const noexcept { return _data.empty(); }
    [[nodiscard]] std::size_t rank() const noexcept { return _extents.size(); }
    [[nodiscard]] std::size_t extent(std::size_t d) const { return _extents.at(d); }
    [[nodiscard]] std::span<const std::size_t> extents() const noexcept { return _extents; }
    [[nodiscard]] T*          data() noexcept { return _data.data(); }
    [[nodiscard]] const T*    data() const noexcept { return _data.data(); }
    [[nodiscard]] T&          operator[](std::size_t i) noexcept { return _data[i]; }
    [[nodiscard]] const T&    operator[](std::size_t i) const noexcept { return _data[i]; }
    [[nodiscard]] auto begin() noexcept { return _data.begin(); }
    [[nodiscard]] auto end() noexcept { return _data.end(); }
    [[nodiscard]] auto begin() const noexcept { return _data.begin(); }
    [[nodiscard]] auto end() const noexcept { return _data.end(); }
    [[nodiscard]] auto cbegin() const noexcept { return _data.cbegin(); }
    [[nodiscard]] auto cend() const noexcept { return _data.cend(); }
    void resize(std::size_t n, T fill = T{}) { _data.resize(n, fill); _extents = {n}; }
    void assign(std::size_t n, T fill) { _data.assign(n, fill); _extents = {n}; }
    [[nodiscard]] friend bool operator==(const Tensor& a, const Tensor& b) { return a._extents == b._extents && a._data == b._data; }
};

// ---------------------------------------------------------------------------------------------- UncertainValue<T> (meta/include/gnuradio-4.0/meta/UncertainValue.hpp)
// A value with its standard uncertainty; the sample type of two of the fourteen math-block registrations (Math.hpp:25-28, 68-71).  Real value types here
// (float, double).  Arithmetic between two uncertain values propagates UNCORRELATED errors (UncertainValue.hpp:121-250): sums and differences combine the
// uncertainties in quadrature, products and quotients the partial derivatives times the operands' uncertainties; a plain number as one operand has none.
// (GR4_COMPAT_NO_UNCERTAIN_VALUE: a translation unit that includes the REFERENCE's own UncertainValue.hpp against this layer defines it first.)
#ifndef GR4_COMPAT_NO_UNCERTAIN_VALUE
template <typename T>
struct UncertainValue {
    using value_type = T;
    T value{};
    T uncertainty{};
    constexpr UncertainValue() = default;
    constexpr UncertainValue(T v, T u = T{}) noexcept : value(v), uncertainty(u) {}
    friend constexpr bool operator==(const UncertainValue&, const UncertainValue&) = default;
};
namespace detail {
template <typename T>
struct is_uncertain : std::false_type {};
template <typename T>
struct is_uncertain<UncertainValue<T>> : std::true_type {};
} // namespace detail
template <typename T>
concept UncertainValueLike = detail::is_uncertain<std::remove_cvref_t<T>>::value;

template <std::floating_point T>
[[nodiscard]] inline UncertainValue<T> operator+(const UncertainValue<T>& a, const UncertainValue<T>& b) noexcept { return {a.value + b.value, std::hypot(a.uncertainty, b.uncertainty)}; }
template <std::floating_point T>
[[nodiscard]] inline UncertainValue<T> operator-(const UncertainValue<T>& a, const UncertainValue<T>& b) noexcept { return {a.value - b.value, std::hypot(a.uncertainty, b.uncertainty)}; }
template <std::floating_point T>
[[nodiscard]] inline UncertainValue<T> operator*(const UncertainValue<T>& a, const UncertainValue<T>& b) noexcept { return {a.value * b.value, std::hypot(a.value * b.uncertainty, b.value * a.uncertainty)}; }
template <std::floating_point T>
[[nodiscard]] inline UncertainValue<T> operator/(const UncertainValue<T>& a, const UncertainValue<T>& b) noexcept {
    return {a.value / b.value, std::hypot(a.uncertainty / b.value, b.uncertainty * a.value / (b.value * b.value))};
}
// one operand exact
template <std::floating_point T> [[nodiscard]] constexpr UncertainValue<T> operator+(const UncertainValue<T>& a, T b) noexcept { return {a.value + b, a.uncertainty}; }
template <std::floating_point T> [[nodiscard]] constexpr UncertainValue<T> operator+(T a, const UncertainValue<T>& b) noexcept { return {a + b.value, b.uncertainty}; }
template <std::floating_point T> [[nodiscard]] constexpr UncertainValue<T> operator-(const UncertainValue<T>& a, T b) noexcept { return {a.value - b, a.uncertainty}; }
template <std::floating_point T> [[nodiscard]] constexpr UncertainValue<T> operator-(T a, const UncertainValue<T>& b) noexcept { return {a - b.value, b.uncertainty}; }
template <std::floating_point T> [[nodiscard]] constexpr UncertainValue<T> operator*(const UncertainValue<T>& a, T b) noexcept { return {a.value * b, a.uncertainty * b}; }
template <std::floating_point T> [[nodiscard]] constexpr UncertainValue<T> operator*(T a, const UncertainValue<T>& b) noexcept { return {a * b.value, a * b.uncertainty}; }
template <std::floating_point T> [[nodiscard]] inline UncertainValue<T> operator/(const UncertainValue<T>& a, T b) noexcept { return {a.value / b, a.uncertainty / std::abs(b)}; }
template <std::floating_point T> [[nodiscard]] inline UncertainValue<T> operator/(T a, const UncertainValue<T>& b) noexcept { return {a / b.value, b.uncertainty * std::abs(a) / (b.value * b.value)}; }
template <std::floating_point T> [[nodiscard]] constexpr UncertainValue<T> operator-(const UncertainValue<T>& a) noexcept { return {-a.value, a.uncertainty}; }
// gr::value / gr::uncertainty (UncertainValue.hpp:79-97): the parts of an uncertain value; a plain number is its own value and has no uncertainty
template <typename T> [[nodiscard]] constexpr auto value(const T& v) noexcept { if constexpr (UncertainValueLike<T>) return v.value; else return v; }
template <typename T> [[nodiscard]] constexpr auto uncertainty(const T& v) noexcept { if constexpr (UncertainValueLike<T>) return v.uncertainty; else return T{}; }
#endif

// ---------------------------------------------------------------------------------------------- property_map (settings payload)
struct property_map;
using pmt_base = std::variant<bool, std::int64_t, std::uint64_t, double, float, std::string, std::complex<float>, std::complex<double>, std::vector<float>,
                              std::vector<double>, std::vector<std::int64_t>, std::shared_ptr<const property_map>>;
// a property value; the last alternative is a nested map ({"leftBlock", property_map{{"value", 2.0}}}: the settings of a merged block's parts,
// BlockMerging.hpp:103-110 forwardNestedSettings)
struct pmt : pmt_base {
    using pmt_base::pmt_base;
    pmt() = default;
    pmt(const property_map& nested);
    pmt(property_map&& nested);
    // a Tensor<T> setting travels as its flat values (the reference's property_map holds the tensor itself; the std::vector spelling stays accepted wherever a
    // Tensor<T> member is set)
    pmt(const Tensor<float>& t) : pmt_base(std::vector<float>(t.begin(), t.end())) {}
    pmt(const Tensor<double>& t) : pmt_base(std::vector<double>(t.begin(), t.end())) {}
    [[nodiscard]] const property_map* get_if_map() const noexcept {
        const auto* p = std::get_if<std::shared_ptr<const property_map>>(static_cast<const pmt_base*>(this));
        return p ? p->get() : nullptr;
    }
    [[nodiscard]] friend bool operator==(const pmt& a, const pmt& b);
};
// (the reference's property_map is pmt::Value::Map; block code uses the std::map surface plus find_value(), ValueMap.hpp:1811-1830)
struct property_map : std::map<std::string, pmt, std::less<>> {
    using base_t = std::map<std::string, pmt, std::less<>>;
    using base_t::base_t;
    property_map() = default;
    [[nodiscard]] std::optional<pmt> find_value(std::string_view key) const {
        const auto it = this->find(key);
        return it == this->end() ? std::nullopt : std::optional<pmt>(it->second);
    }
};
inline pmt::pmt(const property_map& nested) : pmt_base(std::make_shared<const property_map>(nested)) {}
inline pmt::pmt(property_map&& nested) : pmt_base(std::make_shared<const property_map>(std::move(nested))) {}
[[nodiscard]] inline bool operator==(const pmt& a, const pmt& b) { // nested maps compare by content
    const property_map *ma = a.get_if_map(), *mb = b.get_if_map();
    if (ma || mb) return ma && mb && static_cast<const property_map::base_t&>(*ma) == static_cast<const property_map::base_t&>(*mb);
    return static_cast<const pmt_base&>(a) == static_cast<const pmt_base&>(b);
}

namespace detail {
template <typename T>
struct is_vector : std::false_type {};
template <typename T, typename A>
struct is_vector<std::vector<T, A>> : std::true_type {};
template <typename T>
struct is_complex : std::false_type {};
template <typename T>
struct is_complex<std::complex<T>> : std::true_type {};
template <typename T>
struct is_tensor : std::false_type {};
template <typename T>
struct is_tensor<Tensor<T>> : std::true_type {};

// conversion of a property value to the member's type (arithmetic <-> arithmetic, vector<float|double> <-> vector<T>, string, enum by integer)
template <typename T>
bool assign_from(T& dst, const pmt& v) {
    return std::visit(
        [&dst](const auto& x) -> bool {
            using X = std::decay_t<decltype(x)>;
            if constexpr (std::is_same_v<T, X>) {
                dst = x;
                return true;
            } else if constexpr (std::is_enum_v<T> && std::is_arithmetic_v<X>) {
                dst = static_cast<T>(static_cast<std::underlying_type_t<T>>(x));
                return true;
            } else if constexpr (std::is_enum_v<T> && std::is_same_v<X, std::string>) {
                // enum settings by (case-insensitive) name, the role of magic_enum::enum_cast in the reference; the enum's header provides
                // bool gr_enum_parse(T&, std::string_view) next to its definition (found by ADL)
                if constexpr (requires(T& d, std::string_view sv) { { gr_enum_parse(d, sv) } -> std::same_as<bool>; }) return gr_enum_parse(dst, std::string_view(x));
                else return false;
            } else if constexpr (std::is_arithmetic_v<T> && std::is_arithmetic_v<X>) {
                dst = static_cast<T>(x);
                return true;
#ifndef GR4_COMPAT_NO_UNCERTAIN_VALUE
            } else if constexpr (is_uncertain<T>::value && std::is_arithmetic_v<X>) { // a plain number: no uncertainty
                dst = T(static_cast<typename T::value_type>(x));
                return true;
            } else if constexpr (is_uncertain<T>::value && is_vector<X>::value) { // {value, uncertainty}
                if constexpr (std::is_arithmetic_v<typename X::value_type>) {
                    if (x.size() != 2) return false;
                    dst = T(static_cast<typename T::value_type>(x[0]), static_cast<typename T::value_type>(x[1]));
                    return true;
                } else {
                    return false;
                }
#endif
            } else if constexpr (is_complex<T>::value && std::is_arithmetic_v<X>) {
                dst = T(static_cast<typename T::value_type>(x), 0);
                return true;
            } else if constexpr (is_complex<T>::value && is_complex<X>::value) {
                dst = T(static_cast<typename T::value_type>(x.real()), static_cast<typename T::value_type>(x.imag()));
                return true;
            } else if constexpr (is_tensor<T>::value && is_vector<X>::value) { // Tensor<T> settings take the vector<float | double | int64> spellings
                if constexpr (std::is_arithmetic_v<typename T::value_type> && std::is_arithmetic_v<typename X::value_type>) {
                    dst.assign(x.size(), {});
                    for (std::size_t i = 0; i < x.size(); ++i) dst[i] = static_cast<typename T::value_type>(x[i]);
                    return true;
                } else {
                    return false;
                }
            } else if constexpr (is_vector<T>::value && is_vector<X>::value) {
                if constexpr (std::is_arithmetic_v<typename T::value_type> && std::is_arithmetic_v<typename X::value_type>) {
                    dst.assign(x.size(), {});
                    for (std::size_t i = 0; i < x.size(); ++i) dst[i] = static_cast<typename T::value_type>(x[i]);
                    return true;
                } else {
                    return false;
                }
            } else {
                return false;
            }
        },
        static_cast<const pmt_base&>(v));
}
} // namespace detail

// case-insensitive lookup of `name` in a list of enumerator names (helper for the gr_enum_parse overloads)
namespace detail {
template <typename E, std::size_t N>
bool enum_from_names(E& dst, std::string_view name, const std::array<std::string_view, N>& names) {
    for (std::size_t i = 0; i < N; ++i) {
        if (names[i].size() != name.size()) continue;
        bool eq = true;
        for (std::size_t c = 0; c < name.size() && eq; ++c) eq = std::tolower(static_cast<unsigned char>(names[i][c])) == std::tolower(static_cast<unsigned char>(name[c]));
        if (eq) { dst = static_cast<E>(i); return true; }
    }
    return false;
}
} // namespace detail

// ---------------------------------------------------------------------------------------------- DataSet<T> (core/include/gnuradio-4.0/DataSet.hpp), the fields
// the FFT block fills (blocks/fourier/.../fft.hpp:173-250): one dependent axis, nSignals x N values stored signal after signal
template <typename T>
struct Range {
    T min{}, max{};
    bool operator==(const Range&) const = default;
};
template <typename T>
struct DataSet {
    using value_type = T;
    std::int64_t                  timestamp = 0;
    std::vector<std::string>      axis_names, axis_units;
    std::vector<std::vector<T>>   axis_values;
    std::vector<std::int32_t>     extents;
    std::vector<std::string>      signal_names, signal_quantities, signal_units;
    std::vector<T>                signal_values;
    std::vector<Range<T>>         signal_ranges;
    std::vector<std::map<std::string, std::variant<bool, std::int64_t, std::uint64_t, double, float, std::string>, std::less<>>> meta_information;
    [[nodiscard]] std::size_t nDimensions() const noexcept { return extents.size(); }
    [[nodiscard]] std::size_t size() const noexcept { return signal_names.size(); } // number of signals
    [[nodiscard]] std::span<T>       axisValues(std::size_t d) { return axis_values.at(d); }
    [[nodiscard]] std::span<const T> axisValues(std::size_t d) const { return axis_values.at(d); }
    [[nodiscard]] std::span<T> signalValues(std::size_t i) {
        const std::size_t n = size() ? signal_values.size() / size() : 0;
        return std::span<T>(signal_values).subspan(i * n, n);
    }
    [[nodiscard]] std::span<const T> signalValues(std::size_t i) const {
        const std::size_t n = size() ? signal_values.size() / size() : 0;
        return std::span<const T>(signal_values).subspan(i * n, n);
    }
};

// ---------------------------------------------------------------------------------------------- compile-time strings
template <std::size_t N>
struct fixed_string {
    char data[N]{};
    constexpr fixed_string(const char (&s)[N]) { std::copy_n(s, N, data); }
    [[nodiscard]] constexpr std::string_view view() const { return {data, N - 1}; }
};

// ---------------------------------------------------------------------------------------------- annotations (annotated.hpp subset)
template <fixed_string>
struct Doc {};
template <fixed_string>
struct Unit {};
struct Visible {};
template <auto, auto>
struct Limits {};
template <typename T, fixed_string Name, typename... Meta>
struct Annotated {
    using value_type = T;
    T value{};
    constexpr Annotated() = default;
    constexpr Annotated(T v) : value(std::move(v)) {}
    constexpr operator const T&() const noexcept { return value; }
    constexpr operator T&() noexcept { return value; }
    constexpr Annotated& operator=(T v) {
        value = std::move(v);
        return *this;
    }
    static constexpr std::string_view description() { return Name.view(); }
};
namespace detail {
template <typename T>
struct is_annotated : std::false_type {};
template <typename T, fixed_string N, typename... M>
struct is_annotated<Annotated<T, N, M...>> : std::true_type {};
} // namespace detail

// block arguments (annotated.hpp:121-128, Block.hpp:676-683)
template <std::size_t In = 1, std::size_t Out = 1, bool IsConst = false>
struct Resampling {
    static constexpr std::size_t kIn = In, kOut = Out;
    static constexpr bool        kIsConst = IsConst, kEnabled = true;
};
struct NoResampling {
    static constexpr std::size_t kIn = 1, kOut = 1;
    static constexpr bool        kIsConst = true, kEnabled = false;
};

namespace detail {
template <typename A>
inline constexpr bool has_resampling_v = requires { A::kEnabled; };
template <typename... Args>
struct find_resampling {
    using type = NoResampling;
};
template <typename A, typename... R>
struct find_resampling<A, R...> {
    using type = std::conditional_t<has_resampling_v<A>, A, typename find_resampling<R...>::type>;
};
} // namespace detail

// ---------------------------------------------------------------------------------------------- edges: a compacting FIFO (one worker thread)
// ---------------------------------------------------------------------------------------------- tags (Tag.hpp:70-110)
// (index, map) pairs travelling next to the samples.  Keys of the reference's default tags carry the "gr:" prefix on the wire
// (GR_TAG_PREFIX, Tag.hpp:112); index is the absolute position in the edge's stream.
inline constexpr std::string_view GR_TAG_PREFIX = "gr:";
struct Tag {
    std::size_t  index = 0;
    property_map map;
    bool         operator==(const Tag&) const = default;
};
namespace tag {
inline constexpr std::string_view SAMPLE_RATE = "sample_rate";
[[nodiscard]] inline std::string_view settingsKey(std::string_view wireKey) { return wireKey.starts_with(GR_TAG_PREFIX) ? wireKey.substr(GR_TAG_PREFIX.size()) : wireKey; } // "gr:x" -> "x"
[[nodiscard]] inline std::string      wireKey(std::string_view bareKey) { return std::string(GR_TAG_PREFIX) + std::string(bareKey); }
} // namespace tag

struct EdgeBufferBase {
    virtual ~EdgeBufferBase() = default;
    bool producer_done = false; // upstream returned DONE: remaining samples are the last ones
    // fan-out (one output port wired to several inputs): every further reader gets its own buffer, a mirror the writer's publish() copies into
    // -- one worker thread drives every block, so a tee is a copy, not a multi-reader ring.  `upstream` is the writer's own (first) buffer.
    const EdgeBufferBase*                        upstream = nullptr;
    std::vector<std::shared_ptr<EdgeBufferBase>> mirror_bases;
    [[nodiscard]] bool done() const noexcept { return producer_done || (upstream && upstream->producer_done); }
    // tag side channel: host-side for every edge type (a device ring moves samples, the few tags stay with the cursors that order them)
    std::size_t      read_pos = 0, write_pos = 0; // absolute sample counts consumed / published
    std::vector<Tag> tags;                        // ascending index, index >= read_pos
    void publishTag(const property_map& map, std::size_t offset = 0) { // at write_pos + offset, i.e. relative to the span being written (Port.hpp publishTag)
        if (map.empty()) return;
        for (auto& m : mirror_bases) m->publishTag(map, offset); // the mirrors are written in step with this buffer
        const std::size_t idx = write_pos + offset;
        auto it = std::find_if(tags.begin(), tags.end(), [idx](const Tag& t) { return t.index >= idx; });
        if (it != tags.end() && it->index == idx) for (const auto& kv : map) it->map.insert_or_assign(kv.first, kv.second);
        else tags.insert(it, Tag{idx, map});
    }
    [[nodiscard]] const Tag* tagAtReadPosition() const { return !tags.empty() && tags.front().index == read_pos ? &tags.front() : nullptr; }
    // every tag on the next n samples merged into one map, later tags overriding earlier ones key by key: what a chunk that had to span tags (a whole
    // frame, a decimation group) sees and forwards at its first output sample (mergedInputTag, Block.hpp:1511-1530)
    [[nodiscard]] property_map mergedTags(std::size_t n) const {
        property_map merged;
        for (const Tag& t : tags) {
            if (t.index >= read_pos + n) break;
            for (const auto& kv : t.map) merged.insert_or_assign(kv.first, kv.second);
        }
        return merged;
    }
    // samples until the first tag AFTER the read position (nSamplesUntilNextTag(port, 1), Block.hpp:1525): the chunk limit that keeps tags at chunk starts
    [[nodiscard]] std::size_t samplesUntilNextTag() const {
        for (const Tag& t : tags)
            if (t.index > read_pos) return t.index - read_pos;
        return std::numeric_limits<std::size_t>::max();
    }
    void advanceRead(std::size_t n) {
        read_pos += n;
        tags.erase(tags.begin(), std::find_if(tags.begin(), tags.end(), [this](const Tag& t) { return t.index >= read_pos; }));
    }
    void advanceWrite(std::size_t n) { write_pos += n; }
    // type-erased element IO (used by device runs that replace typed blocks at both ends of an edge)
    [[nodiscard]] virtual std::size_t elem_bytes() const noexcept           = 0;
    [[nodiscard]] virtual std::size_t available_items() const noexcept      = 0;
    [[nodiscard]] virtual std::size_t free_items() const noexcept           = 0;
    virtual void                      read_items(void* dst, std::size_t n)  = 0; // copy + consume
    // zero-copy variants for a neighbour that moves the samples itself (a copy engine when memory() is page-locked, or a pool of copy threads):
    //   lend_items(n): pointer to the next n items that are not yet lent, or nullptr; they stay in the buffer -- and available_items() stops counting
    //                  them -- until consume_items() releases them in order.  Several lent spans may be outstanding (chunks in flight).
    //   reserve_items(n) / publish_reserved(n): the same on the writing side: storage for n items behind everything already reserved; readers see
    //                  them once published, in order.
    // While anything is lent or reserved the storage does not move (no compaction): free_items() shrinks to the contiguous room that is left.
    [[nodiscard]] virtual const void*                lend_items(std::size_t /*n*/) { return nullptr; }
    virtual void                                     consume_items(std::size_t /*n*/) {}
    [[nodiscard]] virtual void*                      reserve_items(std::size_t /*n*/) { return nullptr; }
    virtual void                                     publish_reserved(std::size_t /*n*/) {}
    // hand the most recently lent / reserved n items back untouched (a launch that failed before anything was queued for them)
    virtual void                                     unlend_items(std::size_t /*n*/) {}
    virtual void                                     unreserve_items(std::size_t /*n*/) {}
    [[nodiscard]] virtual std::pmr::memory_resource* memory() const { return nullptr; }
    // grow an EMPTY edge to at least n items (same memory resource); false if it cannot (data in it, spans out, fan-out mirrors).  Used by the device planner:
    // an edge that feeds a device run wants chunks far larger than the reference's 65536-item default
    virtual bool ensure_capacity(std::size_t /*n*/) { return false; }
    [[nodiscard]] virtual std::size_t capacity_items() const noexcept { return 0; }
    virtual void                      write_items(const void* src, std::size_t n) = 0; // copy + publish
};
// A memory resource that can also hand out RINGS: `bytes` of storage mapped twice back to back (base[i] and base[i + bytes] are the same byte) -- the reference's
// double-mapped CircularBuffer (CircularBuffer.hpp:75-172) as an allocation mode.  The "hip" provider's page-locked resource is one (hip.hpp): an edge made of such a
// ring never moves samples to its front, and the copy engines read spans that wrap its end in place.
struct RingResource {
    virtual ~RingResource() = default;
    [[nodiscard]] virtual void* ring_allocate(std::size_t bytes) = 0; // nullptr: no ring of this size (the caller takes ordinary storage)
    virtual void                ring_deallocate(void* base, std::size_t bytes) = 0;
};
template <typename T>
struct EdgeBuffer final : EdgeBufferBase {
    std::pmr::vector<T> data; // storage from the edge's memory resource (Graph.hpp:738-775): default heap, or e.g. the "hip" provider's pinned pages
    // ring mode (round 5): the resource is a RingResource and capacity * sizeof(T) is a whole number of pages -- `capacity` items mapped twice; head / tail count on for ever,
    // item i lives at ring[i % capacity] and every span of up to `capacity` items is contiguous.  Linear mode (everything else): 2 x capacity items, compacted when drained.
    T*                         ring = nullptr;
    std::pmr::memory_resource* mr_  = nullptr;
    std::size_t         head = 0, tail = 0, capacity;
    std::size_t         lent = 0, reserved = 0; // items handed out for in-place reading (from head) / writing (from tail) that are still in flight
    explicit EdgeBuffer(std::size_t cap = 65536, std::pmr::memory_resource* mr = std::pmr::get_default_resource()) : data(mr), mr_(mr), capacity(cap) { allocate(cap); } // default edge size: Graph.hpp:102
    EdgeBuffer(const EdgeBuffer&)            = delete;
    EdgeBuffer& operator=(const EdgeBuffer&) = delete;
    ~EdgeBuffer() override { release_ring(); }
    void release_ring() {
        if (ring) dynamic_cast<RingResource*>(mr_)->ring_deallocate(ring, capacity * sizeof(T));
        ring = nullptr;
    }
    void allocate(std::size_t cap) {
        release_ring();
        capacity = cap;
        if constexpr (std::is_trivially_copyable_v<T>) {
            if (auto* rr = dynamic_cast<RingResource*>(mr_)) ring = static_cast<T*>(rr->ring_allocate(cap * sizeof(T)));
        }
        if (ring) { data.clear(); data.shrink_to_fit(); std::memset(static_cast<void*>(ring), 0, cap * sizeof(T)); }
        else data.assign(2 * cap, T{});
    }
    [[nodiscard]] bool        is_ring() const noexcept { return ring != nullptr; }
    [[nodiscard]] T*          base() noexcept { return ring ? ring : data.data(); }
    [[nodiscard]] const T*    base() const noexcept { return ring ? ring : data.data(); }
    [[nodiscard]] std::size_t at(std::size_t pos) const noexcept { return ring ? pos % capacity : pos; } // storage index of stream position `pos`
    [[nodiscard]] std::pmr::memory_resource* resource() const { return mr_; }
    std::vector<std::shared_ptr<EdgeBuffer<T>>> mirrors; // see EdgeBufferBase::upstream
    std::shared_ptr<EdgeBuffer<T>> add_mirror(std::size_t cap, std::pmr::memory_resource* mr) {
        auto m      = std::make_shared<EdgeBuffer<T>>(cap, mr);
        m->upstream = this;
        mirrors.push_back(m);
        mirror_bases.push_back(m);
        return m;
    }
    [[nodiscard]] std::size_t available() const noexcept { return tail - head; }
    [[nodiscard]] std::size_t free_space() const noexcept {
        std::size_t f = capacity - std::min(capacity, available() + reserved);
        if (!ring && (lent || reserved)) f = std::min(f, data.size() - tail - reserved); // the storage stays where it is while a neighbour reads or writes it in place (a ring never moves)
        for (const auto& m : mirrors) f = std::min(f, m->free_space());
        return f;
    }
    std::span<const T>        read_span(std::size_t n) const { return {base() + at(head), n}; }
    void compact() { // (linear mode) move the unread part to the front (amortised O(1) per sample)
        std::move(data.begin() + static_cast<std::ptrdiff_t>(head), data.begin() + static_cast<std::ptrdiff_t>(tail), data.begin());
        tail -= head;
        head = 0;
    }
    std::span<T>              write_span(std::size_t n) {
        if (ring) return {ring + at(tail + reserved), n}; // (n <= free_space() <= capacity: contiguous through the second mapping)
        if (tail + reserved + n > data.size()) {
            if (lent || reserved) throw std::logic_error("EdgeBuffer::write_span: beyond free_space() while spans are lent or reserved");
            compact();
        }
        return {data.data() + tail + reserved, n};
    }
    void publish(std::size_t n) noexcept {
        for (auto& m : mirrors) { // the tee: every further reader gets its copy
            auto dst = m->write_span(n);
            std::copy_n(base() + at(tail), n, dst.data());
            m->publish(n);
        }
        tail += n;
        advanceWrite(n);
    }
    void consume(std::size_t n) noexcept {
        head += n;
        advanceRead(n);
        if (!ring && head == tail && !lent && !reserved) head = tail = 0; // (linear mode) drained: the next span starts at the front again, nothing to move
    }
    [[nodiscard]] std::size_t elem_bytes() const noexcept override { return sizeof(T); }
    [[nodiscard]] std::size_t available_items() const noexcept override { return available() - lent; }
    [[nodiscard]] std::size_t free_items() const noexcept override { return free_space(); }
    void read_items(void* dst, std::size_t n) override {
        if constexpr (std::is_trivially_copyable_v<T>) std::memcpy(dst, read_span(n).data(), n * sizeof(T));
        else throw std::logic_error("type-erased element IO needs a trivially copyable sample type");
        consume(n);
    }
    [[nodiscard]] const void* lend_items(std::size_t n) override {
        if (lent + n > available()) return nullptr;
        const T* p = base() + at(head + lent);
        lent += n;
        return p;
    }
    void consume_items(std::size_t n) override {
        lent -= std::min(lent, n);
        consume(n);
    }
    [[nodiscard]] void* reserve_items(std::size_t n) override {
        if (n > free_space()) return nullptr;
        T* p = write_span(n).data(); // compacts only when nothing is lent or reserved (free_space() has bounded n otherwise)
        reserved += n;
        return p;
    }
    void publish_reserved(std::size_t n) override {
        reserved -= std::min(reserved, n);
        publish(n);
    }
    void unlend_items(std::size_t n) override { lent -= std::min(lent, n); }
    void unreserve_items(std::size_t n) override { reserved -= std::min(reserved, n); }
    [[nodiscard]] std::pmr::memory_resource* memory() const override { return resource(); }
    [[nodiscard]] std::size_t capacity_items() const noexcept override { return capacity; }
    bool ensure_capacity(std::size_t n) override {
        if (n <= capacity) return true;
        if (available() || lent || reserved || !mirrors.empty() || upstream) return false;
        allocate(n);
        head = tail = 0;
        return true;
    }
    void write_items(const void* src, std::size_t n) override {
        if constexpr (std::is_trivially_copyable_v<T>) std::memcpy(write_span(n).data(), src, n * sizeof(T));
        else throw std::logic_error("type-erased element IO needs a trivially copyable sample type");
        publish(n);
    }
};

// ---------------------------------------------------------------------------------------------- ports (Port.hpp)
enum class PortDirection { INPUT, OUTPUT };
struct GPU {}; // PortDomain tag (Port.hpp:183-189)
template <std::size_t Min, std::size_t Max>
struct RequiredSamples {
    static constexpr std::size_t kMin = Min, kMax = Max;
};

namespace hip {
template <typename T>
struct DeviceEdgeBuffer; // the edge of a GPU-domain port: a double-mapped ring in HBM behind the EdgeBuffer interface (gr4/hip.hpp)
}
template <typename T, PortDirection Dir, typename... Attr>
struct Port {
    using value_type                         = T;
    static constexpr PortDirection direction = Dir;
    // a port belongs to ONE computing domain (core/README.md:87-95); GPU ports carry device spans, and crossing the domains takes an explicit
    // converter block (gr::hip::H2D / D2H)
    static constexpr bool kGpu = (std::is_same_v<Attr, GPU> || ...);
    using buffer_type          = std::conditional_t<kGpu, hip::DeviceEdgeBuffer<T>, EdgeBuffer<T>>;
    static constexpr std::string_view domain() { return kGpu ? "GPU" : "CPU"; }
    std::size_t                  min_samples = 1, max_samples = std::numeric_limits<std::size_t>::max();
    std::shared_ptr<buffer_type> buffer; // shared between the connected output and input port
    [[nodiscard]] bool           connected() const noexcept { return static_cast<bool>(buffer); }
};
template <typename T, typename... Attr>
using PortIn = Port<T, PortDirection::INPUT, Attr...>;
template <typename T, typename... Attr>
using PortOut = Port<T, PortDirection::OUTPUT, Attr...>;

namespace detail {
template <typename T>
struct is_port : std::false_type {};
template <typename T, PortDirection D, typename... A>
struct is_port<Port<T, D, A...>> : std::true_type {};
template <typename T>
struct is_port_vector : std::false_type {};
template <typename T, PortDirection D, typename... A>
struct is_port_vector<std::vector<Port<T, D, A...>>> : std::true_type {};
template <typename T>
inline constexpr bool is_input_v = false;
template <typename T, typename... A>
inline constexpr bool is_input_v<Port<T, PortDirection::INPUT, A...>> = true;
template <typename T, typename... A>
inline constexpr bool is_input_v<std::vector<Port<T, PortDirection::INPUT, A...>>> = true;
} // namespace detail

// ---------------------------------------------------------------------------------------------- reflection: GR_MAKE_REFLECTABLE(Type, members...)
#define GR4_STR_1(x) #x,
#define GR4_PTR_1(x) &gr_self_t::x,
#define GR4_FE_1(M, a) M(a)
#define GR4_FE_2(M, a, ...) M(a) GR4_FE_1(M, __VA_ARGS__)
#define GR4_FE_3(M, a, ...) M(a) GR4_FE_2(M, __VA_ARGS__)
#define GR4_FE_4(M, a, ...) M(a) GR4_FE_3(M, __VA_ARGS__)
#define GR4_FE_5(M, a, ...) M(a) GR4_FE_4(M, __VA_ARGS__)
#define GR4_FE_6(M, a, ...) M(a) GR4_FE_5(M, __VA_ARGS__)
#define GR4_FE_7(M, a, ...) M(a) GR4_FE_6(M, __VA_ARGS__)
#define GR4_FE_8(M, a, ...) M(a) GR4_FE_7(M, __VA_ARGS__)
#define GR4_FE_9(M, a, ...) M(a) GR4_FE_8(M, __VA_ARGS__)
#define GR4_FE_10(M, a, ...) M(a) GR4_FE_9(M, __VA_ARGS__)
#define GR4_FE_11(M, a, ...) M(a) GR4_FE_10(M, __VA_ARGS__)
#define GR4_FE_12(M, a, ...) M(a) GR4_FE_11(M, __VA_ARGS__)
#define GR4_FE_13(M, a, ...) M(a) GR4_FE_12(M, __VA_ARGS__)
#define GR4_FE_14(M, a, ...) M(a) GR4_FE_13(M, __VA_ARGS__)
#define GR4_FE_15(M, a, ...) M(a) GR4_FE_14(M, __VA_ARGS__)
#define GR4_FE_16(M, a, ...) M(a) GR4_FE_15(M, __VA_ARGS__)
#define GR4_GET(_1, _2, _3, _4, _5, _6, _7, _8, _9, _10, _11, _12, _13, _14, _15, _16, N, ...) N
#define GR4_FOR_EACH(M, ...)                                                                                                                            \
    GR4_GET(__VA_ARGS__, GR4_FE_16, GR4_FE_15, GR4_FE_14, GR4_FE_13, GR4_FE_12, GR4_FE_11, GR4_FE_10, GR4_FE_9, GR4_FE_8, GR4_FE_7, GR4_FE_6, GR4_FE_5, \
            GR4_FE_4, GR4_FE_3, GR4_FE_2, GR4_FE_1)                                                                                                     \
    (M, __VA_ARGS__)
#define GR_MAKE_REFLECTABLE(Type, ...)                                                                    \
    using gr_self_t = Type;                                                                               \
    static constexpr auto gr_member_names() { return std::array{GR4_FOR_EACH(GR4_STR_1, __VA_ARGS__)}; } \
    static constexpr auto gr_member_ptrs() { return std::tuple{GR4_FOR_EACH(GR4_PTR_1, __VA_ARGS__)}; }

#define GR_REGISTER_BLOCK(...) // registration marker lines are consumed by the reference's blocklib generator; a no-op here

namespace detail {
template <typename Block, typename F>
void for_each_member(Block& b, F&& f) {
    constexpr auto names = Block::gr_member_names();
    std::apply([&](auto... ptr) { std::size_t i = 0; (f(std::string_view(names[i++]), b.*ptr), ...); }, Block::gr_member_ptrs());
}
} // namespace detail

// ---------------------------------------------------------------------------------------------- compute domain (ComputeDomain.hpp:47-100)
struct ComputeDomain {
    std::string kind = "host", backend;
    int         index = 0;
    static ComputeDomain parse(std::string_view s) { // "kind[:backend[:index]]"
        ComputeDomain d;
        if (s.empty()) return d;
        const auto p1 = s.find(':');
        d.kind        = std::string(s.substr(0, p1));
        if (p1 == std::string_view::npos) return d;
        const auto rest = s.substr(p1 + 1);
        const auto p2   = rest.find(':');
        d.backend       = std::string(rest.substr(0, p2));
        if (p2 != std::string_view::npos) d.index = std::stoi(std::string(rest.substr(p2 + 1)));
        return d;
    }
    [[nodiscard]] bool is_device() const { return kind == "gpu"; }
};

// ---------------------------------------------------------------------------------------------- memory seam (ComputeDomain.hpp:105-173)
// Given a domain (+ optional backend context) a provider returns the memory resource edge buffers of that domain are allocated from.
using ProviderFn = std::pmr::memory_resource* (*)(const ComputeDomain& dom, void* ctx);
class ComputeRegistry {
    mutable std::mutex                          _mtx;
    std::unordered_map<std::string, ProviderFn> _providers;

public:
    static ComputeRegistry& instance() {
        static ComputeRegistry r;
        return r;
    }
    void register_provider(std::string_view backend, ProviderFn fn) {
        std::scoped_lock lk(_mtx);
        _providers[std::string(backend)] = fn; // replace-or-insert
    }
    [[nodiscard]] expected<std::pmr::memory_resource*> resolve(const ComputeDomain& dom, void* ctx = nullptr) const {
        if (dom.kind == "host" || dom.backend == "none") return std::pmr::new_delete_resource();
        std::scoped_lock lk(_mtx);
        const auto       it = _providers.find(dom.backend);
        if (it == _providers.end()) return unexpected("no provider for backend '" + dom.backend + "'");
        if (auto* mr = it->second(dom, ctx)) return mr;
        return unexpected("provider returned null resource for backend '" + dom.backend + "'");
    }
    [[nodiscard]] std::pmr::memory_resource* tryResolve(const ComputeDomain& dom, void* ctx = nullptr) const noexcept {
        try {
            const auto r = resolve(dom, ctx);
            return r ? r.value() : nullptr;
        } catch (...) { return nullptr; }
    }
};

// ---------------------------------------------------------------------------------------------- type-erased block (BlockModel.hpp:493, 668-769)
struct BlockModel {
    virtual ~BlockModel()                                               = default;
    virtual work::Result               work(std::size_t requested)      = 0;
    virtual std::string_view           name() const                     = 0;
    virtual std::string_view           type_name() const                = 0;
    virtual const ComputeDomain&       compute_domain() const           = 0;
    virtual void*                      raw()                            = 0;
    virtual std::type_index            port_type(std::string_view port) = 0; // typeid(void) if unknown
    virtual std::string_view           port_domain(std::string_view) { return "CPU"; }
    // the index-th reflected input / output port member ("" when there is none) and whether a port member is a vector of ports (GRC wires by index)
    virtual std::string                port_name(bool /*output*/, std::size_t /*index*/) { return {}; }
    virtual bool                       port_is_vector(std::string_view) { return false; }
    virtual std::shared_ptr<EdgeBufferBase> make_edge(std::string_view out_port, std::size_t min_size, std::pmr::memory_resource* mr = nullptr) = 0;
    virtual bool                            attach_input(std::string_view in_port, std::shared_ptr<EdgeBufferBase> edge) = 0;
    virtual std::vector<std::shared_ptr<EdgeBufferBase>> input_edges()                                                   = 0;
    virtual std::vector<std::shared_ptr<EdgeBufferBase>> output_edges()                                                  = 0;
    // a gr::hip::Stage for this block's current settings, or null when the block type has no device kernel (gr4/hip.hpp; type-erased here
    // so that this header stays free of the device layer)
    virtual std::shared_ptr<void> make_device_stage() { return nullptr; }
    // settings-by-tag for a block that is driven from outside its own work loop (a member of a fused device run)
    virtual bool apply_tag_settings(const property_map&) { return false; }
};

// ---------------------------------------------------------------------------------------------- Block<Derived, Args...> (Block.hpp)
namespace hip {
template <typename Derived>
struct Kernel; // device implementation of a block type, specialised in gr4/hip.hpp; primary template is intentionally undefined
template <typename Derived>
concept HasKernel = requires { sizeof(Kernel<Derived>); };
} // namespace hip

template <typename Derived, typename... Args>
struct Block {
    using ResamplingControl = typename detail::find_resampling<Args...>::type;

    std::string   name = "block";
    std::string   compute_domain = "host"; // Block.hpp:713
    Size_t        input_chunk_size = static_cast<Size_t>(ResamplingControl::kIn), output_chunk_size = static_cast<Size_t>(ResamplingControl::kOut);
    ComputeDomain _domain{};
    bool          _warned_device_fallback = false;
    std::function<void(std::string_view)> _log = [](std::string_view) {};
    std::shared_ptr<void> _device_state;  // hip::Offload of this block (type-erased with its deleter: released with the block, i.e. with the graph)
    std::size_t   _settings_generation = 0; // bumped by every applySettings(): a device stage built from older settings is stale

    Derived&       self() { return *static_cast<Derived*>(this); }
    const Derived& self() const { return *static_cast<const Derived*>(this); }

    // does `key` name one of this block's settings?  (merged blocks hand a flat key to every part that has it, BlockMerging.hpp:206-210)
    [[nodiscard]] bool hasSetting(std::string_view key) {
        if constexpr (requires(Derived& d) { d.hasSettingOverride(key); }) {
            if (self().hasSettingOverride(key)) return true;
        }
        bool found = false;
        detail::for_each_member(self(), [&](std::string_view mname, auto& member) {
            using M = std::decay_t<decltype(member)>;
            if constexpr (!detail::is_port<M>::value && !detail::is_port_vector<M>::value) {
                bool match = (mname == key);
                if constexpr (detail::is_annotated<M>::value) match = match || (M::description() == key);
                found = found || match;
            }
        });
        return found;
    }
    // apply a property_map to the reflected members and call settingsChanged(old, new) like Block::init / applyChangedSettings
    void applySettings(const property_map& newSettings) {
        property_map applied;
        for (const auto& [key, value] : newSettings) {
            if (key == "name") { detail::assign_from(name, value); continue; }
            if (key == "compute_domain") { detail::assign_from(compute_domain, value); _domain = ComputeDomain::parse(compute_domain); continue; }
            bool found = false;
            detail::for_each_member(self(), [&](std::string_view mname, auto& member) {
                using M = std::decay_t<decltype(member)>;
                if constexpr (detail::is_port<M>::value || detail::is_port_vector<M>::value) {
                    (void)member;
                } else {
                    bool match = (mname == key);
                    if constexpr (detail::is_annotated<M>::value) match = match || (M::description() == key);
                    if (match) {
                        found = true;
                        bool ok;
                        if constexpr (detail::is_annotated<M>::value) ok = detail::assign_from(member.value, value);
                        else ok = detail::assign_from(member, value);
                        if (!ok) throw std::invalid_argument("setting '" + key + "': incompatible value type");
                        applied.emplace(key, value);
                    }
                }
            });
            if (!found) throw std::invalid_argument("unknown setting '" + key + "' for block " + name);
        }
        if (!applied.empty()) ++_settings_generation;
        if constexpr (requires(Derived& d, const property_map& m) { d.settingsChanged(m, m); }) self().settingsChanged(property_map{}, applied);
    }

    // ---- the work loop (Block.hpp:2028-2173 reduced to stream samples; tags/messages/lifecycle are out of scope)
    work::Result work(std::size_t requested) {
        if constexpr (requires(Derived& d) { d.customWork(requested); }) {
            return self().customWork(requested); // sources / sinks with their own pacing
        } else {
            return defaultWork(requested);
        }
    }

    // forwarded subset of a tag map: wire keys with the "gr:" prefix; gr:sample_rate (float) is scaled by output_chunk_size / input_chunk_size
    // on resampling blocks (insertOutputValue, Block.hpp:1088-1099; pinned by qa_filter.cpp:288-292)
    [[nodiscard]] property_map toOutputTags(const property_map& in) const {
        property_map out;
        for (const auto& [key, value] : in) {
            if (!std::string_view(key).starts_with(GR_TAG_PREFIX)) continue;
            const bool convert = ResamplingControl::kEnabled && input_chunk_size != output_chunk_size && input_chunk_size != 0 && tag::settingsKey(key) == tag::SAMPLE_RATE;
            if (const float* rate = convert ? std::get_if<float>(&value) : nullptr) out.insert_or_assign(key, static_cast<float>(output_chunk_size) / static_cast<float>(input_chunk_size) * *rate);
            else out.insert_or_assign(key, value);
        }
        return out;
    }
    // settings-by-tag (Settings.hpp:433 autoUpdate): keys that name a reflected setting of this block ("gr:value" or "value") are applied; returns
    // whether anything was applied
    bool applyTagSettings(const property_map& tagMap) {
        property_map matching;
        for (const auto& [key, value] : tagMap) {
            const std::string_view field = tag::settingsKey(key);
            detail::for_each_member(self(), [&](std::string_view mname, auto& member) {
                using M = std::decay_t<decltype(member)>;
                if constexpr (!detail::is_port<M>::value && !detail::is_port_vector<M>::value) {
                    bool match = mname == field;
                    if constexpr (detail::is_annotated<M>::value) match = match || M::description() == field;
                    if (match) { // only values that differ from the active setting count as a change
                        if constexpr (detail::is_annotated<M>::value) {
                            auto tmp = member.value;
                            if (detail::assign_from(tmp, value) && !(tmp == member.value)) matching.insert_or_assign(std::string(mname), value);
                        } else {
                            auto tmp = member;
                            if (detail::assign_from(tmp, value) && !(tmp == member)) matching.insert_or_assign(std::string(mname), value);
                        }
                    }
                }
            });
        }
        if (matching.empty()) return false;
        try {
            applySettings(matching);
        } catch (const std::exception& e) { // a tag with an unusable value does not stop the stream
            _log(std::string("settings-by-tag ignored: ") + e.what());
            return false;
        }
        ++_settings_by_tag;
        return true;
    }
    std::size_t _settings_by_tag = 0;

private:
    template <typename F>
    void each_in(F&& f) {
        detail::for_each_member(self(), [&](std::string_view, auto& m) {
            using M = std::decay_t<decltype(m)>;
            if constexpr (detail::is_port<M>::value && detail::is_input_v<M>) f(m);
            else if constexpr (detail::is_port_vector<M>::value && detail::is_input_v<M>) for (auto& p : m) f(p);
        });
    }
    template <typename F>
    void each_out(F&& f) {
        detail::for_each_member(self(), [&](std::string_view, auto& m) {
            using M = std::decay_t<decltype(m)>;
            if constexpr (detail::is_port<M>::value && !detail::is_input_v<M>) f(m);
            else if constexpr (detail::is_port_vector<M>::value && !detail::is_input_v<M>) for (auto& p : m) f(p);
        });
    }

    work::Result defaultWork(std::size_t requested) {
        // computeSampleLimits (Block.hpp:1950-1977): min over sync inputs / outputs, then resampling chunks (:1576-1636)
        std::size_t avail = std::numeric_limits<std::size_t>::max(), space = avail, maxIn = avail;
        bool        upstream_done = true, any_in = false;
        each_in([&](auto& p) {
            any_in = true;
            if (!p.connected()) { avail = 0; return; }
            avail         = std::min(avail, p.buffer->available());
            maxIn         = std::min(maxIn, p.max_samples);
            upstream_done = upstream_done && p.buffer->done();
        });
        each_out([&](auto& p) { if (p.connected()) space = std::min(space, p.buffer->free_space()); });
        if (!any_in) return {requested, 0, work::Status::ERROR};
        const std::size_t ic = std::max<std::size_t>(1, input_chunk_size), oc = std::max<std::size_t>(1, output_chunk_size);
        // a chunk ends where the next tag starts, so that every tag sits on the first sample of a chunk (Block.hpp:1511-1530, 1961-1971);
        // never below one input chunk (ensureMinimalDecimation): tags inside such a forced chunk are merged into the chunk's own tag below
        std::size_t nextTag = std::numeric_limits<std::size_t>::max();
        each_in([&](auto& p) { if (p.connected()) nextTag = std::min(nextTag, p.buffer->samplesUntilNextTag()); });
        avail = std::min(avail, std::max(nextTag, ic));
        std::size_t       k  = std::min({avail / ic, space / oc, std::min(maxIn, requested) / ic});
        if (k == 0) {
            if (avail / ic == 0) {
                if (upstream_done) { // trailing partial chunk is dropped (IncompleteFinalUpdateEnum::DROP, Block.hpp:677)
                    each_in([&](auto& p) { if (p.connected()) p.buffer->consume(p.buffer->available()); });
                    each_out([&](auto& p) { if (p.connected()) p.buffer->producer_done = true; });
                    return {requested, 0, work::Status::DONE};
                }
                return {requested, 0, work::Status::INSUFFICIENT_INPUT_ITEMS};
            }
            return {requested, 0, work::Status::INSUFFICIENT_OUTPUT_ITEMS};
        }
        const std::size_t nIn = k * ic, nOut = k * oc;
        // the chunk's tag: normally the one on its first sample; every tag of the chunk when it had to span some.  Settings-by-tag first
        // (Block.hpp:1979-1985), then the chunk is processed with the new settings
        property_map chunkTags;
        each_in([&](auto& p) {
            if (!p.connected()) return;
            for (auto& kv : p.buffer->mergedTags(nIn)) chunkTags.insert_or_assign(kv.first, std::move(kv.second)); // identical tags on several inputs collapse
        });
        if (!chunkTags.empty()) applyTagSettings(chunkTags);
        work::Status      st = dispatch(nIn, nOut);
        if (st == work::Status::ERROR) return {requested, 0, st};
        if (!chunkTags.empty()) { // default forwarding (Block.hpp:1113-1263): "gr:" keys only, at the first output sample of the chunk
            const property_map fwd = toOutputTags(chunkTags);
            each_out([&](auto& p) { if (p.connected()) p.buffer->publishTag(fwd, 0); });
        }
        // finaliseIO (Block.hpp:1989-2026): publish outputs, then consume inputs
        each_out([&](auto& p) { if (p.connected()) p.buffer->publish(nOut); });
        each_in([&](auto& p) { p.buffer->consume(nIn); });
        return {requested, nIn, work::Status::OK};
    }

    // dispatchProcessing (Block.hpp:1848-1917) for the 1-in/1-out and N-in/1-out shapes on the hot path
    work::Status dispatch(std::size_t nIn, std::size_t nOut) {
        if (_domain.is_device()) { // the device seam (Block.hpp:1855-1862)
            if constexpr (hip::HasKernel<Derived>) {
                return hip::Kernel<Derived>::work(self(), nIn, nOut);
            } else if (!_warned_device_fallback) {
                _warned_device_fallback = true; // reference behaviour pinned by qa_Block.cpp:1315-1343: warn once, run on the CPU
                _log("compute_domain '" + compute_domain + "' requested but block has no device implementation: running on host");
            }
        }
        return dispatchHost(nIn, nOut);
    }

public:
    work::Status dispatchHost(std::size_t nIn, std::size_t nOut) {
        Derived& d = self();
        if constexpr (requires { d.in.buffer; d.out.buffer; }) {
            using TIn  = typename std::decay_t<decltype(d.in)>::value_type;
            using TOut = typename std::decay_t<decltype(d.out)>::value_type;
            std::span<const TIn> is = d.in.buffer->read_span(nIn);
            std::span<TOut>      os = d.out.connected() ? d.out.buffer->write_span(nOut) : std::span<TOut>{};
            std::vector<TOut>    scratch;
            if (!d.out.connected()) { scratch.resize(nOut); os = scratch; }
            if constexpr (requires { d.processBulk(is, os); }) {
                return d.processBulk(is, os);
            } else {
                for (std::size_t i = 0; i < nIn; ++i) os[i] = d.processOne(is[i]); // invokeProcessOneNonConst (Block.hpp:1723-1761)
                return work::Status::OK;
            }
        } else if constexpr (requires { d.in.size(); d.out.buffer; }) { // std::vector<PortIn<T>> in; PortOut<T> out (Math.hpp:84-86)
            using T = typename std::decay_t<decltype(d.out)>::value_type;
            std::vector<std::span<const T>> ins;
            for (auto& p : d.in) ins.push_back(p.buffer->read_span(nIn));
            std::span<T> os = d.out.buffer->write_span(nOut);
            return d.processBulk(std::span<const std::span<const T>>(ins), os);
        } else {
            static_assert(sizeof(Derived) == 0, "unsupported port shape for the default work loop");
        }
    }
};

template <typename T>
struct BlockWrapper final : BlockModel {
    T                block{};
    std::string      _type;
    explicit BlockWrapper(std::string type) : _type(std::move(type)) {}
    work::Result         work(std::size_t requested) override { return block.work(requested); }
    std::string_view     name() const override { return block.name; }
    std::string_view     type_name() const override { return _type; }
    const ComputeDomain& compute_domain() const override { return block._domain; }
    void*                raw() override { return &block; }

    template <typename F>
    bool with_port(std::string_view port, F&& f) { // "out", "in", "in#2"
        std::string_view base = port;
        std::size_t      idx  = 0;
        if (const auto h = port.find('#'); h != std::string_view::npos) {
            base = port.substr(0, h);
            idx  = static_cast<std::size_t>(std::stoul(std::string(port.substr(h + 1))));
        }
        if constexpr (requires { T::port_alias(base); }) base = T::port_alias(base); // e.g. a merged block's ports under the names its parts gave them
        bool done = false;
        detail::for_each_member(block, [&](std::string_view mname, auto& m) {
            using M = std::decay_t<decltype(m)>;
            if (done || mname != base) return;
            if constexpr (detail::is_port<M>::value) { f(m); done = true; }
            else if constexpr (detail::is_port_vector<M>::value) { if (idx < m.size()) { f(m[idx]); done = true; } }
        });
        return done;
    }
    std::type_index port_type(std::string_view port) override {
        std::type_index t = typeid(void);
        with_port(port, [&](auto& p) { t = typeid(typename std::decay_t<decltype(p)>::value_type); });
        return t;
    }
    std::string port_name(bool output, std::size_t index) override {
        std::string found;
        std::size_t k = 0;
        detail::for_each_member(block, [&](std::string_view mname, auto& m) {
            using M = std::decay_t<decltype(m)>;
            if constexpr (detail::is_port<M>::value || detail::is_port_vector<M>::value) {
                if (detail::is_input_v<M> != output) {
                    if (k == index) found = std::string(mname);
                    ++k;
                }
            }
        });
        return found;
    }
    bool port_is_vector(std::string_view port) override {
        bool v = false;
        detail::for_each_member(block, [&](std::string_view mname, auto& m) {
            if (mname == port) v = detail::is_port_vector<std::decay_t<decltype(m)>>::value;
        });
        return v;
    }
    std::string_view port_domain(std::string_view port) override {
        std::string_view d = "CPU";
        with_port(port, [&](auto& p) { d = std::decay_t<decltype(p)>::domain(); });
        return d;
    }
    std::shared_ptr<EdgeBufferBase> make_edge(std::string_view out_port, std::size_t min_size, std::pmr::memory_resource* mr) override {
        std::shared_ptr<EdgeBufferBase> e;
        with_port(out_port, [&](auto& p) {
            using P = std::decay_t<decltype(p)>;
            if constexpr (P::direction == PortDirection::OUTPUT) {
                if (!p.buffer) {
                    if constexpr (P::kGpu) p.buffer = std::make_shared<typename P::buffer_type>(std::max<std::size_t>(min_size, 65536)); // HBM ring: its own allocator
                    else p.buffer = std::make_shared<typename P::buffer_type>(std::max<std::size_t>(min_size, 65536), mr ? mr : std::pmr::get_default_resource());
                    e = p.buffer;
                } else {
                    // the port already feeds a reader: this connection is a fan-out -- a mirror buffer of its own for the new reader
                    if constexpr (P::kGpu) e = p.buffer->add_reader(); // GPU-domain edges: another read cursor on the SAME ring in HBM (hip.hpp DeviceEdgeBuffer: no copy)
                    else e = p.buffer->add_mirror(std::max<std::size_t>(min_size, 65536), mr ? mr : std::pmr::get_default_resource());
                }
            }
        });
        return e;
    }
    bool attach_input(std::string_view in_port, std::shared_ptr<EdgeBufferBase> edge) override {
        bool ok = false;
        with_port(in_port, [&](auto& p) {
            using P = std::decay_t<decltype(p)>;
            if constexpr (P::direction == PortDirection::INPUT) {
                if (auto typed = std::dynamic_pointer_cast<typename P::buffer_type>(edge)) { p.buffer = typed; ok = true; }
            }
        });
        return ok;
    }
    bool apply_tag_settings(const property_map& m) override { return block.applyTagSettings(m); }
    std::shared_ptr<void> make_device_stage() override {
        if constexpr (requires { hip::Kernel<T>::make_stage(block); }) return std::shared_ptr<void>(hip::Kernel<T>::make_stage(block)); // deleter captured here
        else return nullptr;
    }
    std::vector<std::shared_ptr<EdgeBufferBase>> input_edges() override {
        std::vector<std::shared_ptr<EdgeBufferBase>> v;
        detail::for_each_member(block, [&](std::string_view, auto& m) {
            using M = std::decay_t<decltype(m)>;
            if constexpr (detail::is_port<M>::value && detail::is_input_v<M>) v.push_back(m.buffer);
            else if constexpr (detail::is_port_vector<M>::value && detail::is_input_v<M>) for (auto& p : m) v.push_back(p.buffer);
        });
        return v;
    }
    std::vector<std::shared_ptr<EdgeBufferBase>> output_edges() override {
        std::vector<std::shared_ptr<EdgeBufferBase>> v;
        detail::for_each_member(block, [&](std::string_view, auto& m) {
            using M = std::decay_t<decltype(m)>;
            if constexpr (detail::is_port<M>::value && !detail::is_input_v<M>) v.push_back(m.buffer);
            else if constexpr (detail::is_port_vector<M>::value && !detail::is_input_v<M>) for (auto& p : m) v.push_back(p.buffer);
        });
        return v;
    }
};

// ---------------------------------------------------------------------------------------------- Graph (Graph.hpp:361-786)
struct EdgeParameters { // BlockModel.hpp:64-72
    std::size_t minBufferSize = 65536;
    std::int32_t weight = 0;
    std::string name = "unnamed edge", domain; // domain "gpu:hip[:i]": the edge's storage comes from that backend's provider (ComputeRegistry)
    std::pmr::memory_resource* dataResource = nullptr; // explicit resource: most specific wins (Graph.hpp:742-765)
};
struct Edge {
    BlockModel* src;
    std::string src_port;
    BlockModel* dst;
    std::string dst_port;
    EdgeParameters params;
};

class Graph {
    std::vector<std::unique_ptr<BlockModel>> _blocks, _retired; // _retired: blocks absorbed by a fused device run (kept alive)
    std::vector<Edge>                        _edges;

    BlockModel* find(const void* raw) {
        for (auto& b : _blocks)
            if (b->raw() == raw) return b.get();
        return nullptr;
    }

public:
    template <typename T>
    T& emplaceBlock(const property_map& initial = {}) { // Graph.hpp:425-443
        auto w = std::make_unique<BlockWrapper<T>>(typeid(T).name());
        w->block.applySettings(initial);
        T& ref = w->block;
        _blocks.push_back(std::move(w));
        return ref;
    }

    // a block created elsewhere (plugin registry, PluginLoader::instantiate): Graph::addBlock (Graph.hpp:445-452)
    BlockModel& addBlock(std::unique_ptr<BlockModel> block) {
        _blocks.push_back(std::move(block));
        return *_blocks.back();
    }

    // runtime variant (Graph.hpp:564-593): connect(src, "out", dst, "in#0"); src / dst are typed blocks or BlockModels of this graph
    template <typename S, typename D>
    expected<void> connect(S& src, std::string_view srcPort, D& dst, std::string_view dstPort, EdgeParameters params = {}) {
        const auto model = [this](auto& b) -> BlockModel* {
            if constexpr (std::is_base_of_v<BlockModel, std::decay_t<decltype(b)>>) return find(b.raw());
            else return find(&b);
        };
        BlockModel *s = model(src), *d = model(dst);
        if (!s || !d) return unexpected("connect: block is not part of this graph");
        const auto ts = s->port_type(srcPort), td = d->port_type(dstPort);
        if (ts == typeid(void)) return unexpected("connect: source port '" + std::string(srcPort) + "' not found");
        if (td == typeid(void)) return unexpected("connect: destination port '" + std::string(dstPort) + "' not found");
        if (ts != td) return unexpected("connect: port value types differ");
        if (s->port_domain(srcPort) != d->port_domain(dstPort))
            return unexpected("connect: ports belong to different computing domains (" + std::string(s->port_domain(srcPort)) + " -> " + std::string(d->port_domain(dstPort)) +
                              "): insert an explicit converter block (gr::hip::H2D / gr::hip::D2H)");
        // resource precedence (Graph.hpp:742-765): EdgeParameters resource > non-host domain provider > default
        std::pmr::memory_resource* mr = params.dataResource;
        if (!mr && !params.domain.empty())
            if (const auto dom = ComputeDomain::parse(params.domain); dom.kind != "host") mr = ComputeRegistry::instance().tryResolve(dom);
        auto edge = s->make_edge(srcPort, params.minBufferSize, mr);
        if (!edge) return unexpected("connect: '" + std::string(srcPort) + "' is not an output port (or a second reader on a GPU-domain edge)");
        if (!d->attach_input(dstPort, edge)) return unexpected("connect: '" + std::string(dstPort) + "' is not an input port of the same type");
        _edges.push_back({s, std::string(srcPort), d, std::string(dstPort), std::move(params)});
        return {};
    }
    // compile-time variant (Graph.hpp:595-690): connect<"out","in">(src, dst)
    template <fixed_string SrcPort, fixed_string DstPort, typename S, typename D>
    expected<void> connect(S& src, D& dst, EdgeParameters params = {}) {
        return connect(src, SrcPort.view(), dst, DstPort.view(), std::move(params));
    }

    [[nodiscard]] std::span<const Edge>      edges() const { return _edges; }
    std::vector<std::unique_ptr<BlockModel>>& blocks() { return _blocks; }
    std::vector<std::unique_ptr<BlockModel>>& retired() { return _retired; }
};

// ---------------------------------------------------------------------------------------------- scheduler::Simple (Scheduler.hpp:1916-1952)
namespace scheduler {
class Simple {
    Graph       _graph;
    std::size_t _max_work_items = std::numeric_limits<std::size_t>::max();

public:
    expected<void> exchange(Graph&& g) {
        _graph = std::move(g);
        return {};
    }
    Graph& graph() { return _graph; }
    // runAndWait (Scheduler.hpp:581-621): traverseBlockListOnce (:718-736) until every block is DONE or nothing progresses
    expected<void> runAndWait() {
        auto&             blocks = _graph.blocks();
        std::vector<bool> done(blocks.size(), false);
        for (;;) {
            bool progress = false, all_done = true;
            for (std::size_t i = 0; i < blocks.size(); ++i) {
                if (done[i]) continue;
                const work::Result r = blocks[i]->work(_max_work_items);
                if (r.status == work::Status::ERROR) return unexpected("block '" + std::string(blocks[i]->name()) + "' returned ERROR");
                if (r.status == work::Status::DONE) {
                    done[i]  = true;
                    progress = true;
                    for (auto& e : blocks[i]->output_edges())
                        if (e) e->producer_done = true;
                    continue;
                }
                all_done = false;
                progress = progress || r.performed_work > 0;
            }
            if (all_done) return {};
            if (!progress) return unexpected("scheduler stalled: no block can make progress");
        }
    }
};
} // namespace scheduler

} // namespace gr
