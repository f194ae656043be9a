// gr4/compat.hpp -- what a block header written against the reference's include tree names beyond gr4/core.hpp, so that the files under
// blocks/*/include/gnuradio-4.0 of the reference compile UNCHANGED against this host layer (used through the forwarding
// headers in ../gnuradio-4.0/).  Everything here is API surface only: concept and trait names (meta/utils.hpp, meta/UncertainValue.hpp), the span
// concepts of processBulk (Port.hpp:421-452), gr::exception, the registration macro (a no-op: registration is explicit here, gr4/plugin.hpp), and --
// because this image's libstdc++ 11 has no <format> -- a minimal std::format for the diagnostics block headers build (replaced by the real one
// wherever <format> exists).
#pragma once
#include "core.hpp"

#include <sstream>

#if __has_include(<format>)
#include <format>
#else
namespace gr::compat_detail {
template <typename T>
void put(std::ostringstream& os, const T& v) {
    if constexpr (requires(std::ostringstream& o, const T& x) { o << x; }) os << v;
    else if constexpr (requires(const T& x) { x.size(); x.begin()->first; }) { // a map of settings: the keys
        os << "{";
        bool first = true;
        for (const auto& kv : v) { os << (first ? "" : ", ") << kv.first; first = false; }
        os << "}";
    } else os << "<" << sizeof(T) << "-byte object>";
}
inline void fmt_rest(std::ostringstream& os, std::string_view f) { os << f; }
template <typename A, typename... R>
void fmt_rest(std::ostringstream& os, std::string_view f, const A& a, const R&... r) {
    const auto open = f.find('{');
    if (open == std::string_view::npos) { os << f; return; }
    const auto close = f.find('}', open);
    os << f.substr(0, open);
    put(os, a);
    fmt_rest(os, close == std::string_view::npos ? std::string_view{} : f.substr(close + 1), r...);
}
} // namespace gr::compat_detail
namespace std { // NOLINT: stand-in for the C++20 library function this toolchain lacks; "{}" / "{:...}" fields are filled left to right
template <typename... Args>
string format(string_view f, const Args&... args) {
    ostringstream os;
    gr::compat_detail::fmt_rest(os, f, args...);
    return os.str();
}
} // namespace std
#endif

namespace gr {
namespace meta {
using gr::fixed_string; // gr::meta::fixed_string (meta/utils.hpp:133): a structural compile-time string, deducible from a literal
template <typename T>
struct is_std_array_type : std::false_type {};
template <typename T, std::size_t N>
struct is_std_array_type<std::array<T, N>> : std::true_type {};
template <typename T>
concept vector_type = gr::detail::is_vector<std::remove_cv_t<T>>::value;
template <typename T>
concept array_type = is_std_array_type<std::remove_cv_t<T>>::value;
template <typename T, typename V = void>
concept array_or_vector_type = (vector_type<T> || array_type<T>) && (std::same_as<V, void> || std::same_as<typename T::value_type, V>); // meta/utils.hpp:700
template <typename T>
inline constexpr bool always_false = false;
// is_instantiation_of<T, Template> (meta/utils.hpp): T is Template<...> of type arguments
namespace compat_inst {
template <typename T, template <typename...> class Template>
struct test : std::false_type {};
template <template <typename...> class Template, typename... Args>
struct test<Template<Args...>, Template> : std::true_type {};
} // namespace compat_inst
template <typename T, template <typename...> class Template>
concept is_instantiation_of = compat_inst::test<std::remove_cvref_t<T>, Template>::value;
template <typename T>
concept complex_like = detail::is_complex<std::remove_cvref_t<T>>::value;
// the scalar behind a sample type (meta/utils.hpp): complex<T> -> T
template <typename T>
struct fundamental_base_value_type { using type = T; };
template <typename T>
    requires requires { typename T::value_type; }
struct fundamental_base_value_type<T> { using type = typename fundamental_base_value_type<typename T::value_type>::type; };
template <typename T>
using fundamental_base_value_type_t = typename fundamental_base_value_type<T>::type;
// no SIMD evaluation on this host layer (the device path is the wide one): V is always the sample type itself
template <typename V, typename T>
concept t_or_simd = std::same_as<V, T>;
template <typename V, typename... T>
concept any_simd = false;
} // namespace meta
#ifndef GR4_COMPAT_NO_UNCERTAIN_VALUE // (gr::UncertainValue<T> itself: gr4/core.hpp; a unit that includes the reference's own UncertainValue.hpp gets both names from there)
template <typename T>
concept arithmetic_or_complex_like = std::is_arithmetic_v<T> || meta::complex_like<T>;
#endif

struct exception : std::runtime_error { // gr::exception (reporting.hpp): message + source location upstream
    using std::runtime_error::runtime_error;
};

// processBulk(InputSpanLike..., OutputSpanLike&) (Port.hpp:421-452): the default work loop hands std::span views of the edge
template <typename T>
concept InputSpanLike = requires(const T& s) { s.begin(); s.end(); s.size(); };
template <typename T>
concept OutputSpanLike = requires(T& s) { s.begin(); s.end(); s.size(); };
} // namespace gr

// GR_REGISTER_BLOCK(...) lines are input of the reference's build-time registry generator (blocklib_generator/tools/parse_registrations.cpp); blocks are
// registered explicitly on this layer (gr4/plugin.hpp: BlockRegistry::insert), so the marker expands to nothing
#define GR_REGISTER_BLOCK(...)
