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
// Settings reach the parts as upstream (BlockMerging.hpp:93-110, 206-210, 641-649): a flat key goes to every part that has a setting of that name, and a
// nested map under the part's name -- {"leftBlock", property_map{{"value", 2.0}}}, "rightBlock", "forward", "feedback", "path<I>" -- goes to that part;
// dotted keys ("leftBlock.value") are accepted as a shorthand for the nested form.  The exposed ports are `in` / `out`; the names the parts gave them
// (a FeedbackMerge over Adder<> exposes the adder's free input: `in1`) resolve to the same ports in Graph::connect.
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

// settings of a merged block -> its parts.  `has(i, key)`: does part i have a setting of that name.  Returns per-part maps; what belongs to the merged
// block itself (name, compute_domain, ...) lands in `own`; a key nobody knows throws like an unknown setting
inline std::vector<property_map> route_settings(const property_map& all, const std::vector<std::string>& part_names, const std::function<bool(std::size_t, std::string_view)>& has,
                                                property_map& own) {
    std::vector<property_map> per(part_names.size());
    const auto part_index = [&](std::string_view name) {
        for (std::size_t i = 0; i < part_names.size(); ++i)
            if (part_names[i] == name) return i;
        return part_names.size();
    };
    for (const auto& [key, value] : all) {
        if (const property_map* nested = value.get_if_map()) { // {"leftBlock", property_map{...}}
            const std::size_t i = part_index(key);
            if (i == part_names.size()) throw std::invalid_argument("unknown sub-block '" + key + "'");
            for (const auto& kv : *nested) per[i].insert_or_assign(kv.first, kv.second);
            continue;
        }
        if (const auto dot = key.find('.'); dot != std::string::npos) { // "leftBlock.value"
            const std::size_t i = part_index(std::string_view(key).substr(0, dot));
            if (i == part_names.size()) throw std::invalid_argument("unknown sub-block '" + key.substr(0, dot) + "' in setting '" + key + "'");
            per[i].insert_or_assign(key.substr(dot + 1), value);
            continue;
        }
        bool taken = false; // flat key: every part that has such a setting takes it (forwardSettings to both parts)
        for (std::size_t i = 0; i < part_names.size(); ++i)
            if (has(i, key)) { per[i].insert_or_assign(key, value); taken = true; }
        if (!taken) own.emplace(key, value);
    }
    return per;
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

    void applySettings(const property_map& settings) { // flat keys, {"leftBlock", map} / {"rightBlock", map}, "leftBlock.<setting>" (see the file header)
        property_map own;
        const auto   per = detail::route_settings(settings, {"leftBlock", "rightBlock"}, [this](std::size_t i, std::string_view k) { return i == 0 ? leftBlock.hasSetting(k) : rightBlock.hasSetting(k); }, own);
        if (!per[0].empty()) leftBlock.applySettings(per[0]);
        if (!per[1].empty()) rightBlock.applySettings(per[1]);
        Block<Merge>::applySettings(own); // name, compute_domain
    }
    [[nodiscard]] bool hasSettingOverride(std::string_view k) { return leftBlock.hasSetting(k) || rightBlock.hasSetting(k); } // (a Merge inside a Merge)
    // the exposed ports under the names the parts gave them: A's input port, B's output port (whatever they are called there)
    static constexpr std::string_view port_alias(std::string_view p) {
        if constexpr (requires { A::port_alias(p); }) { if (A::port_alias(p) == "in") return "in"; }
        if constexpr (requires { B::port_alias(p); }) { if (B::port_alias(p) == "out") return "out"; }
        return p;
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

    void applySettings(const property_map& settings) { // flat keys, {"forward", map} / {"feedback", map}, "feedback.<setting>"
        property_map own;
        const auto   per = detail::route_settings(settings, {"forward", "feedback"}, [this](std::size_t i, std::string_view k) { return i == 0 ? forward.hasSetting(k) : feedback.hasSetting(k); }, own);
        if (!per[0].empty()) forward.applySettings(per[0]);
        if (!per[1].empty()) feedback.applySettings(per[1]);
        Block<FeedbackMerge>::applySettings(own);
    }
    [[nodiscard]] bool hasSettingOverride(std::string_view k) { return forward.hasSetting(k) || feedback.hasSetting(k); }
    // upstream keeps the forward block's name for its free input (the adder's `in1` when the feedback enters at `in2`)
    static constexpr std::string_view port_alias(std::string_view p) { return p == (kFeedbackIntoSecond ? "in1" : "in2") ? std::string_view("in") : p; }
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

    void applySettings(const property_map& settings) { // flat keys (every path that has the setting), {"path<I>", map}, "path<I>.<setting>"
        std::vector<std::string> names;
        for (std::size_t i = 0; i < sizeof...(Paths); ++i) names.push_back("path" + std::to_string(i));
        property_map own;
        const auto   has = [this](std::size_t i, std::string_view k) {
            bool h = false;
            [&]<std::size_t... I>(std::index_sequence<I...>) { ((I == i ? void(h = std::get<I>(_paths).hasSetting(k)) : void()), ...); }(std::index_sequence_for<Paths...>{});
            return h;
        };
        const auto per = detail::route_settings(settings, names, has, own);
        [&]<std::size_t... I>(std::index_sequence<I...>) { ((per[I].empty() ? void() : std::get<I>(_paths).applySettings(per[I])), ...); }(std::index_sequence_for<Paths...>{});
        Block<SplitMergeCombineImpl>::applySettings(own);
    }
    [[nodiscard]] bool hasSettingOverride(std::string_view k) {
        return [&]<std::size_t... I>(std::index_sequence<I...>) { return (std::get<I>(_paths).hasSetting(k) || ...); }(std::index_sequence_for<Paths...>{});
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
