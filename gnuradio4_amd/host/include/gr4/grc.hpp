// gr4/grc.hpp -- a reader for the subset of the GRC / YAML graph format that names blocks and wires them
// (core/include/gnuradio-4.0/Graph_yaml_importer.hpp; shape pinned by core/test/qa_grc.cpp:132-152):
//
//     blocks:
//       - id: gr::filter::fir_filter<float32>
//         parameters:
//           name: lowpass
//           b: [0.5, 0.25, 0.25]
//           compute_domain: "gpu:hip:0"
//     connections:
//       - [source, 0, lowpass, 0]            # [source block, output port, destination block, input port]
//       - [lowpass, out, sink, in]           # ports by index, by [index, sub-index] (vector ports) or by name
//
// Blocks come from a gr::PluginLoader by their registry id (gr4/plugin.hpp), parameters become the property_map the block is created
// with (integers -> int64, reals -> double, true/false -> bool, [a, b, ...] -> vector<double>, !!float32 / !!int32 ... tags honoured, the
// rest strings), blocks are addressed in `connections` by their `name` parameter (or by their id when it is unique).  Everything else of
// the format (sub-graphs, schedulers, UI hints, context-dependent settings) is outside this reader; unknown top-level keys are ignored.
// Errors are returned, not thrown.
#pragma once
#include <sstream>

#include "plugin.hpp"

namespace gr {
namespace grc_detail {
inline std::string trim(std::string_view s) {
    std::size_t a = 0, b = s.size();
    while (a < b && std::isspace(static_cast<unsigned char>(s[a]))) ++a;
    while (b > a && std::isspace(static_cast<unsigned char>(s[b - 1]))) --b;
    return std::string(s.substr(a, b - a));
}
inline std::string unquote(std::string s) {
    if (s.size() >= 2 && ((s.front() == '"' && s.back() == '"') || (s.front() == '\'' && s.back() == '\''))) return s.substr(1, s.size() - 2);
    return s;
}
inline std::string strip_comment(const std::string& line) { // '#' outside quotes starts a comment
    char q = 0;
    for (std::size_t i = 0; i < line.size(); ++i) {
        const char c = line[i];
        if (q) { if (c == q) q = 0; }
        else if (c == '"' || c == '\'') q = c;
        else if (c == '#' && (i == 0 || std::isspace(static_cast<unsigned char>(line[i - 1])))) return line.substr(0, i);
    }
    return line;
}
// split "a, [b, c], 'd, e'" at top-level commas
inline std::vector<std::string> split_flow(std::string_view s) {
    std::vector<std::string> out;
    int                      depth = 0;
    char                     q     = 0;
    std::string              cur;
    for (const char c : s) {
        if (q) { cur += c; if (c == q) q = 0; continue; }
        if (c == '"' || c == '\'') { q = c; cur += c; continue; }
        if (c == '[') ++depth;
        if (c == ']') --depth;
        if (c == ',' && depth == 0) { out.push_back(trim(cur)); cur.clear(); continue; }
        cur += c;
    }
    if (!trim(cur).empty()) out.push_back(trim(cur));
    return out;
}
inline bool is_int(const std::string& s) {
    if (s.empty()) return false;
    std::size_t i = (s[0] == '-' || s[0] == '+') ? 1 : 0;
    if (i == s.size()) return false;
    for (; i < s.size(); ++i)
        if (!std::isdigit(static_cast<unsigned char>(s[i]))) return false;
    return true;
}
inline bool is_real(const std::string& s, double* v) {
    if (s.empty()) return false;
    char* end = nullptr;
    *v        = std::strtod(s.c_str(), &end);
    return end && *end == '\0' && (std::isdigit(static_cast<unsigned char>(s[0])) || s[0] == '-' || s[0] == '+' || s[0] == '.');
}
inline expected<pmt> scalar(std::string text) {
    text = trim(text);
    std::string tag;
    if (text.rfind("!!", 0) == 0) { // explicit type: "!!float32 43"
        const auto sp = text.find(' ');
        tag           = text.substr(2, sp == std::string::npos ? std::string::npos : sp - 2);
        text          = sp == std::string::npos ? std::string() : trim(text.substr(sp + 1));
    }
    if (!text.empty() && text.front() == '[' && text.back() == ']') {
        std::vector<double> v;
        for (const auto& e : split_flow(std::string_view(text).substr(1, text.size() - 2))) {
            double d;
            if (!is_real(e, &d)) return unexpected("list element '" + e + "' is not a number");
            v.push_back(d);
        }
        return pmt(v);
    }
    double d = 0;
    if (tag == "float32") { if (!is_real(text, &d)) return unexpected("'" + text + "' is not a float32"); return pmt(static_cast<float>(d)); }
    if (tag == "float64") { if (!is_real(text, &d)) return unexpected("'" + text + "' is not a float64"); return pmt(d); }
    if (tag.rfind("int", 0) == 0) { if (!is_int(text)) return unexpected("'" + text + "' is not an integer"); return pmt(static_cast<std::int64_t>(std::stoll(text))); }
    if (tag.rfind("uint", 0) == 0) { if (!is_int(text)) return unexpected("'" + text + "' is not an integer"); return pmt(static_cast<std::uint64_t>(std::stoull(text))); }
    if (tag == "bool") return pmt(text == "true");
    if (tag == "str") return pmt(unquote(text));
    if (text.size() >= 2 && (text.front() == '"' || text.front() == '\'')) return pmt(unquote(text));
    if (text == "true" || text == "false") return pmt(text == "true");
    if (is_int(text)) return pmt(static_cast<std::int64_t>(std::stoll(text)));
    if (is_real(text, &d)) return pmt(d);
    return pmt(text);
}
} // namespace grc_detail

struct GrcBlock {
    std::string  id, name;
    property_map parameters;
    BlockModel*  model = nullptr;
};

// reads `yaml`, instantiates the blocks through `loader`, wires them into `graph`; returns the blocks in file order
inline expected<std::vector<GrcBlock>> loadGrc(PluginLoader& loader, Graph& graph, std::string_view yaml) {
    using namespace grc_detail;
    std::vector<GrcBlock>              blocks;
    std::vector<std::vector<std::string>> connections;
    enum class Sec { None, Blocks, Connections, Other } sec = Sec::None;
    bool               in_params = false;
    std::size_t        params_indent = 0;
    std::istringstream in{std::string(yaml)};
    std::string        raw;
    int                lineno = 0;
    const auto fail = [&](const std::string& what) { return unexpected("grc line " + std::to_string(lineno) + ": " + what); };
    while (std::getline(in, raw)) {
        ++lineno;
        const std::string line = strip_comment(raw);
        if (trim(line).empty()) continue;
        const std::size_t indent = line.find_first_not_of(' ');
        std::string       text   = trim(line);
        if (indent == 0) { // top-level key
            in_params = false;
            sec       = text == "blocks:" ? Sec::Blocks : text == "connections:" ? Sec::Connections : Sec::Other;
            continue;
        }
        if (sec == Sec::Connections) {
            if (text.rfind("- ", 0) != 0) return fail("expected '- [src, port, dst, port]'");
            text = trim(text.substr(2));
            if (text.size() < 2 || text.front() != '[' || text.back() != ']') return fail("a connection is a flow sequence [src, port, dst, port]");
            auto parts = split_flow(std::string_view(text).substr(1, text.size() - 2));
            if (parts.size() != 4) return fail("a connection has four entries");
            connections.push_back(std::move(parts));
            continue;
        }
        if (sec != Sec::Blocks) continue;
        if (text.rfind("- ", 0) == 0) { // new block
            blocks.emplace_back();
            in_params = false;
            text      = trim(text.substr(2));
        }
        if (blocks.empty()) return fail("expected '- id: ...'");
        const auto colon = text.find(':');
        if (colon == std::string::npos) return fail("expected 'key: value'");
        const std::string key = unquote(trim(text.substr(0, colon))), value = trim(text.substr(colon + 1));
        if (in_params && indent >= params_indent) {
            auto v = scalar(value);
            if (!v) return fail("parameter '" + key + "': " + v.error().message);
            if (key == "name") blocks.back().name = std::get_if<std::string>(&v.value()) ? std::get<std::string>(v.value()) : value;
            blocks.back().parameters.insert_or_assign(key, v.value());
            continue;
        }
        in_params = false;
        if (key == "id") blocks.back().id = unquote(value);
        else if (key == "parameters") { in_params = true; params_indent = indent + 1; }
        // other per-block keys (ui_constraints, ctx_parameters, ...) are not part of this reader
    }
    // instantiate
    std::map<std::string, std::size_t, std::less<>> by_name;
    for (std::size_t i = 0; i < blocks.size(); ++i) {
        auto& b = blocks[i];
        if (b.id.empty()) return unexpected("grc: block " + std::to_string(i) + " has no id");
        std::unique_ptr<BlockModel> model;
        try {
            model = loader.instantiate(b.id, b.parameters);
        } catch (const std::exception& e) { return unexpected("grc: block '" + b.id + "': " + e.what()); }
        if (!model) return unexpected("grc: no loaded plugin knows block '" + b.id + "'");
        b.model = &graph.addBlock(std::move(model));
        if (b.name.empty()) b.name = b.id;
        if (!by_name.emplace(b.name, i).second) return unexpected("grc: two blocks are called '" + b.name + "' (give them distinct `name` parameters)");
    }
    // wire
    const auto port_of = [](BlockModel& m, const std::string& spec, bool output) -> expected<std::string> {
        std::string s = trim(spec);
        if (!s.empty() && s.front() == '[') { // [index, sub-index]
            const auto parts = split_flow(std::string_view(s).substr(1, s.size() - 2));
            if (parts.size() != 2 || !is_int(parts[0]) || !is_int(parts[1])) return unexpected("port '" + spec + "' is not [index, sub-index]");
            const std::string base = m.port_name(output, static_cast<std::size_t>(std::stoul(parts[0])));
            if (base.empty()) return unexpected("no " + std::string(output ? "output" : "input") + " port " + parts[0]);
            return m.port_is_vector(base) ? base + "#" + parts[1] : base;
        }
        if (is_int(s)) {
            const std::string base = m.port_name(output, static_cast<std::size_t>(std::stoul(s)));
            if (base.empty()) return unexpected("no " + std::string(output ? "output" : "input") + " port " + s);
            return m.port_is_vector(base) ? base + "#0" : base;
        }
        return unquote(s);
    };
    for (const auto& c : connections) {
        const auto si = by_name.find(unquote(c[0])), di = by_name.find(unquote(c[2]));
        if (si == by_name.end() || di == by_name.end()) return unexpected("grc: connection names an unknown block ('" + c[0] + "' -> '" + c[2] + "')");
        BlockModel &s = *blocks[si->second].model, &d = *blocks[di->second].model;
        const auto  sp = port_of(s, c[1], true), dp = port_of(d, c[3], false);
        if (!sp) return unexpected("grc: " + c[0] + ": " + sp.error().message);
        if (!dp) return unexpected("grc: " + c[2] + ": " + dp.error().message);
        if (const auto r = graph.connect(s, sp.value(), d, dp.value()); !r) return unexpected("grc: " + c[0] + "." + sp.value() + " -> " + c[2] + "." + dp.value() + ": " + r.error().message);
    }
    return blocks;
}
} // namespace gr
