// gr4/merge.hpp -- the merge API of the reference (core/include/gnuradio-4.0/BlockMerging.hpp:136-320 Merge, :600-800 FeedbackMerge;
// docs/USER_API_Connecting_Blocks.md "Merging blocks") for the 1-in / 1-out shapes of the hot path.
//
//     using IIRChain = gr::Merge<MultiplyConst<float>, "out", gr::FeedbackMerge<Adder<>, "out", MultiplyConst<float>, "out", "in2">, "in1">;
//
// is the reference benchmark's IIR low-pass y[n] = a x[n] + (1 - a) y[n-1] (core/benchmarks/bm_MergeApi.cpp:59-60) and compiles here
// unchanged.  On the host a merged block is what it is upstream: one processOne() that calls its parts, no buffer in between, the
// feedback path closed over a one-sample state.  With compute_domain = "gpu:hip" the parts become device stages of ONE block, and
// affine feedback loops (adder + constant gain) collapse into the parallel-in-time IIR kernel (gr4/hip.hpp): the run-time fusion the
// reference does at compile time.
// Deviation: a merged block always names its remaining ports `in` and `out` (upstream keeps the forward block's names, e.g. `in1`).
// Sub-block settings use dotted keys ("leftBlock.value", "feedback.value") instead of nested property_maps.
#pragma once
#include "blocks.hpp"

namespace gr {

// 2-input adder with processOne (bm_MergeApi.cpp:46-56; Math.hpp's Add<T> has dynamic ports and processBulk)
template <typename T = float>
struct Adder : Block<Adder<T>> {
    PortIn<T>  in1, in2;
    PortOut<T> out;
    GR_MAKE_REFLECTABLE(Adder, in1, in2, out);
    [[nodiscard]] constexpr T processOne(T a, T b) const noexcept { return a + b; }
};

namespace detail {
template <typename B>
using in_type_t = typename std::decay_t<decltype(std::declval<B&>().in)>::value_type;
template <typename B>
using out_type_t = typename std::decay_t<decltype(std::declval<B&>().out)>::value_type;

// split {"leftBlock.value": v, ...} into per-part maps; unknown prefixes throw like an unknown setting
inline void route_settings(const property_map& all, std::initializer_list<std::pair<std::string_view, property_map*>> parts, property_map& own) {
    for (const auto& [key, value] : all) {
        const auto dot = key.find('.');
        if (dot == std::string::npos) { own.emplace(key, value); continue; }
        bool routed = false;
        for (auto& [prefix, dst] : parts)
            if (std::string_view(key).substr(0, dot) == prefix) { dst->emplace(key.substr(dot + 1), value); routed = true; }
        if (!routed) throw std::invalid_argument("unknown sub-block '" + key.substr(0, dot) + "' in setting '" + key + "'");
    }
}
} // namespace detail

// Merge<A, "out", B, "in">: A.out -> B.in, exposed ports: A's input, B's output (BlockMerging.hpp:136-320)
template <typename A, fixed_string OutA, typename B, fixed_string InB>
struct Merge : Block<Merge<A, OutA, B, InB>> {
    using TIn  = detail::in_type_t<A>;
    using TOut = detail::out_type_t<B>;
    static_assert(std::is_same_v<detail::out_type_t<A>, detail::in_type_t<B>>, "Merge: port value types differ");
    PortIn<TIn>   in;
    PortOut<TOut> out;
    GR_MAKE_REFLECTABLE(Merge, in, out);
    A leftBlock{};
    B rightBlock{};

    void applySettings(const property_map& settings) { // keys "leftBlock.<setting>" / "rightBlock.<setting>" (Settings forwarding, USER_API_Connecting_Blocks.md)
        property_map l, r, own;
        detail::route_settings(settings, {{"leftBlock", &l}, {"rightBlock", &r}}, own);
        if (!l.empty()) leftBlock.applySettings(l);
        if (!r.empty()) rightBlock.applySettings(r);
        Block<Merge>::applySettings(own); // name, compute_domain
    }
    [[nodiscard]] TOut processOne(TIn x) { return rightBlock.processOne(leftBlock.processOne(x)); }
};

// FeedbackMerge<Forward, "out", Feedback, "out", "in2">: Forward.out -> Feedback.in, Feedback.out -> Forward.<in2> delayed by one sample
// (BlockMerging.hpp:600-800).  Exposed: Forward's other input, Forward's output.
template <typename Forward, fixed_string ForwardOut, typename Feedback, fixed_string FeedbackOut, fixed_string ForwardFeedbackIn>
struct FeedbackMerge : Block<FeedbackMerge<Forward, ForwardOut, Feedback, FeedbackOut, ForwardFeedbackIn>> {
    using T = detail::out_type_t<Forward>;
    static_assert(ForwardFeedbackIn.view() == "in1" || ForwardFeedbackIn.view() == "in2", "FeedbackMerge: the forward block's feedback input is in1 or in2");
    static constexpr bool kFeedbackIntoSecond = ForwardFeedbackIn.view() == "in2";
    PortIn<T>  in;
    PortOut<T> out;
    GR_MAKE_REFLECTABLE(FeedbackMerge, in, out);
    Forward  forward{};
    Feedback feedback{};
    T        _state{}; // what the feedback path delivered for the previous sample

    void applySettings(const property_map& settings) { // keys "forward.<setting>" / "feedback.<setting>"
        property_map f, b, own;
        detail::route_settings(settings, {{"forward", &f}, {"feedback", &b}}, own);
        if (!f.empty()) forward.applySettings(f);
        if (!b.empty()) feedback.applySettings(b);
        Block<FeedbackMerge>::applySettings(own);
    }
    [[nodiscard]] T processOne(T x) noexcept {
        const T y = kFeedbackIntoSecond ? forward.processOne(x, _state) : forward.processOne(_state, x);
        _state    = feedback.processOne(y);
        return y;
    }
};

// SplitMergeCombine<[OutputSigns<s0, s1, ...>,] Path0, Path1, ...>: the input goes to every path, the (signed) path outputs are summed
// (BlockMerging.hpp:337-520).  Paths are 1-in / 1-out blocks of one sample type; sub-block settings by "path<I>.<setting>"; block.path<I>().
template <auto... Signs>
struct OutputSigns {
    static constexpr std::size_t size = sizeof...(Signs);
    static constexpr std::array<double, sizeof...(Signs)> values{static_cast<double>(Signs)...};
};
namespace detail {
template <typename T>
struct is_output_signs : std::false_type {};
template <auto... S>
struct is_output_signs<OutputSigns<S...>> : std::true_type {};
} // namespace detail

template <typename SignsT, typename... Paths>
struct SplitMergeCombineImpl : Block<SplitMergeCombineImpl<SignsT, Paths...>> {
    static_assert(sizeof...(Paths) >= 2, "SplitMergeCombine needs at least two paths");
    using First = std::tuple_element_t<0, std::tuple<Paths...>>;
    using T     = detail::in_type_t<First>;
    static_assert((std::is_same_v<detail::in_type_t<Paths>, T> && ...) && (std::is_same_v<detail::out_type_t<Paths>, T> && ...), "SplitMergeCombine: one sample type on every path");
    static_assert(SignsT::size == 0 || SignsT::size == sizeof...(Paths), "OutputSigns: one sign per path");
    PortIn<T>  in;
    PortOut<T> out;
    GR_MAKE_REFLECTABLE(SplitMergeCombineImpl, in, out);
    std::tuple<Paths...> _paths{};

    template <std::size_t I> auto&       path() { return std::get<I>(_paths); }
    template <std::size_t I> const auto& path() const { return std::get<I>(_paths); }
    static constexpr double sign(std::size_t i) { return SignsT::size == 0 ? 1.0 : SignsT::values[i]; }

    void applySettings(const property_map& settings) {
        std::array<property_map, sizeof...(Paths)> per{};
        property_map                               own;
        for (const auto& [key, value] : settings) {
            const auto dot = key.find('.');
            if (dot == std::string::npos) { own.emplace(key, value); continue; }
            const std::string prefix = key.substr(0, dot);
            std::size_t       idx    = sizeof...(Paths);
            if (prefix.rfind("path", 0) == 0 && prefix.size() > 4) idx = static_cast<std::size_t>(std::stoul(prefix.substr(4)));
            if (idx >= sizeof...(Paths)) throw std::invalid_argument("unknown sub-block '" + prefix + "' in setting '" + key + "'");
            per[idx].emplace(key.substr(dot + 1), value);
        }
        [&]<std::size_t... I>(std::index_sequence<I...>) { ((per[I].empty() ? void() : std::get<I>(_paths).applySettings(per[I])), ...); }(std::index_sequence_for<Paths...>{});
        Block<SplitMergeCombineImpl>::applySettings(own);
    }
    [[nodiscard]] T processOne(T x) {
        return [&]<std::size_t... I>(std::index_sequence<I...>) { return ((static_cast<T>(sign(I)) * std::get<I>(_paths).processOne(x)) + ...); }(std::index_sequence_for<Paths...>{});
    }
};
namespace detail {
template <typename A, typename... Rest>
struct split_merge_select { using type = SplitMergeCombineImpl<OutputSigns<>, A, Rest...>; };
template <auto... S, typename... Rest>
struct split_merge_select<OutputSigns<S...>, Rest...> { using type = SplitMergeCombineImpl<OutputSigns<S...>, Rest...>; };
} // namespace detail
template <typename A, typename... Rest>
using SplitMergeCombine = typename detail::split_merge_select<A, Rest...>::type;

} // namespace gr
