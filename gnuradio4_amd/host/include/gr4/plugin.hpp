// gr4/plugin.hpp -- the plugin C ABI of the reference (core/include/gnuradio-4.0/Plugin.hpp:20-85, PluginLoader.hpp:277-305) on top of gr4/core.hpp.
//
// A block library is a shared object that exports
//     extern "C" void gr_plugin_make(gr_plugin_base** plugin);      extern "C" void gr_plugin_free(gr_plugin_base* plugin);
// and hands out a gr_plugin_base whose virtuals list the blocks it knows (under the registry's portable type names, e.g.
// "gr::filter::fir_filter<float32>", BlockRegistry.hpp:76-104) and create them from a property_map.  gr::PluginLoader opens such a
// library, refuses a different ABI version, and instantiates blocks by name -- which is how a graph description that only holds
// strings ("id: gr::filter::fir_filter<float32>", "compute_domain: gpu:hip:0") reaches the device kernels.
#pragma once
#include <dlfcn.h>

#include "core.hpp"

#define GR_PLUGIN_CURRENT_ABI_VERSION 1

struct gr_plugin_metadata { // PluginMetadata.hpp
    std::string_view plugin_name, plugin_author, plugin_license, plugin_version;
};

namespace gr {
struct SchedulerModel { // the slice of the scheduler interface a plugin can hand out
    virtual ~SchedulerModel()                          = default;
    virtual expected<void> exchange(Graph&& g)         = 0;
    virtual expected<void> runAndWait()                = 0;
    virtual Graph&         graph()                     = 0;
};
struct SimpleSchedulerModel final : SchedulerModel {
    scheduler::Simple impl;
    expected<void>    exchange(Graph&& g) override { return impl.exchange(std::move(g)); }
    expected<void>    runAndWait() override { return impl.runAndWait(); }
    Graph&            graph() override { return impl.graph(); }
};
} // namespace gr

class gr_plugin_base {
public:
    gr_plugin_metadata metadata;
    virtual ~gr_plugin_base() = default;
    virtual std::uint8_t                        abiVersion() const                                                          = 0;
    virtual std::vector<std::string>            availableBlocks() const                                                     = 0;
    virtual std::unique_ptr<gr::BlockModel>     createBlock(std::string_view name, const gr::property_map& params)          = 0;
    virtual std::vector<std::string>            availableSchedulers() const                                                 = 0;
    virtual std::unique_ptr<gr::SchedulerModel> createScheduler(std::string_view name, const gr::property_map& params)      = 0;
};

namespace gr {

// name -> factory (GeneralRegistry, BlockRegistry.hpp:60-140)
class BlockRegistry {
    using Factory = std::unique_ptr<BlockModel> (*)(std::string, const property_map&);
    std::map<std::string, Factory, std::less<>> _handlers;

public:
    template <typename TBlock>
    bool insert(std::string_view name) {
        return _handlers
            .insert_or_assign(std::string(name),
                              +[](std::string type, const property_map& params) -> std::unique_ptr<BlockModel> {
                                  auto w = std::make_unique<BlockWrapper<TBlock>>(std::move(type));
                                  w->block.applySettings(params); // throws std::invalid_argument for unknown settings, like Block::init
                                  return w;
                              })
            .second;
    }
    [[nodiscard]] std::unique_ptr<BlockModel> create(std::string_view name, const property_map& params) const {
        const auto it = _handlers.find(name);
        return it == _handlers.end() ? nullptr : it->second(std::string(name), params);
    }
    [[nodiscard]] std::vector<std::string> keys() const {
        std::vector<std::string> k;
        for (const auto& kv : _handlers) k.push_back(kv.first);
        return k;
    }
    [[nodiscard]] bool contains(std::string_view name) const { return _handlers.find(name) != _handlers.end(); }
};

template <std::uint8_t ABI_VERSION = GR_PLUGIN_CURRENT_ABI_VERSION>
class plugin : public gr_plugin_base {
    BlockRegistry registry;

public:
    std::uint8_t                    abiVersion() const override { return ABI_VERSION; }
    std::vector<std::string>        availableBlocks() const override { return registry.keys(); }
    std::unique_ptr<gr::BlockModel> createBlock(std::string_view name, const property_map& params) override { return registry.create(name, params); }
    std::vector<std::string>        availableSchedulers() const override { return {"gr::scheduler::Simple"}; }
    std::unique_ptr<gr::SchedulerModel> createScheduler(std::string_view name, const property_map&) override {
        return name == "gr::scheduler::Simple" ? std::make_unique<SimpleSchedulerModel>() : nullptr;
    }
    operator BlockRegistry&() { return registry; }
};

// dlopen + gr_plugin_make + ABI check (PluginLoader.hpp:277-305); errors are returned, not thrown
class PluginHandler {
    void*           _dl     = nullptr;
    gr_plugin_base* _plugin = nullptr;
    void (*_free)(gr_plugin_base*) = nullptr;
    std::string _error;

public:
    explicit PluginHandler(const std::string& path) {
        _dl = dlopen(path.c_str(), RTLD_LAZY | RTLD_LOCAL);
        if (!_dl) { _error = std::string("dlopen: ") + dlerror(); return; }
        auto make = reinterpret_cast<void (*)(gr_plugin_base**)>(dlsym(_dl, "gr_plugin_make"));
        _free     = reinterpret_cast<void (*)(gr_plugin_base*)>(dlsym(_dl, "gr_plugin_free"));
        if (!make || !_free) { _error = "not a GNU Radio 4 plugin: gr_plugin_make / gr_plugin_free missing"; release(); return; }
        make(&_plugin);
        if (!_plugin) { _error = "gr_plugin_make returned no plugin"; release(); return; }
        if (_plugin->abiVersion() != GR_PLUGIN_CURRENT_ABI_VERSION) {
            _error = "plugin ABI version " + std::to_string(_plugin->abiVersion()) + " != " + std::to_string(GR_PLUGIN_CURRENT_ABI_VERSION);
            release();
        }
    }
    PluginHandler(const PluginHandler&)            = delete;
    PluginHandler& operator=(const PluginHandler&) = delete;
    ~PluginHandler() { release(); }
    void release() {
        if (_plugin && _free) _free(_plugin);
        _plugin = nullptr;
        if (_dl) dlclose(_dl);
        _dl = nullptr;
    }
    [[nodiscard]] explicit   operator bool() const { return _plugin != nullptr; }
    [[nodiscard]] const std::string& status() const { return _error; }
    gr_plugin_base*                  operator->() const { return _plugin; }
};

class PluginLoader {
    std::vector<std::unique_ptr<PluginHandler>> _handlers;
    std::map<std::string, std::string>          _failed; // path -> reason

public:
    expected<void> load(const std::string& path) {
        auto h = std::make_unique<PluginHandler>(path);
        if (!*h) { _failed[path] = h->status(); return unexpected(path + ": " + h->status()); }
        _handlers.push_back(std::move(h));
        return {};
    }
    [[nodiscard]] const std::map<std::string, std::string>& failedPlugins() const { return _failed; }
    [[nodiscard]] std::vector<std::string> availableBlocks() const {
        std::vector<std::string> all;
        for (auto& h : _handlers) { auto k = (*h)->availableBlocks(); all.insert(all.end(), k.begin(), k.end()); }
        return all;
    }
    [[nodiscard]] bool isBlockAvailable(std::string_view name) const {
        const auto all = availableBlocks();
        return std::find(all.begin(), all.end(), name) != all.end();
    }
    // instantiate(name, params) (PluginLoader.hpp): nullptr when no plugin knows the name; a factory that throws (unknown setting) propagates
    [[nodiscard]] std::unique_ptr<BlockModel> instantiate(std::string_view name, const property_map& params = {}) const {
        for (auto& h : _handlers)
            if (auto b = (*h)->createBlock(name, params)) return b;
        return nullptr;
    }
    [[nodiscard]] std::unique_ptr<SchedulerModel> instantiateScheduler(std::string_view name, const property_map& params = {}) const {
        for (auto& h : _handlers)
            if (auto s = (*h)->createScheduler(name, params)) return s;
        return nullptr;
    }
};
} // namespace gr

// GR_PLUGIN("name", "author", "license", "version") (Plugin.hpp:70-81): defines grPluginInstance() and the two C entry points
#define GR_PLUGIN(Name, Author, License, Version)                                                                                           \
    gr::plugin<>& grPluginInstance() {                                                                                                      \
        static gr::plugin<> instance = [] {                                                                                                 \
            gr::plugin<> result;                                                                                                            \
            result.metadata = gr_plugin_metadata{Name, Author, License, Version};                                                           \
            return result;                                                                                                                  \
        }();                                                                                                                                \
        return instance;                                                                                                                    \
    }                                                                                                                                       \
    extern "C" {                                                                                                                            \
    __attribute__((visibility("default"))) void gr_plugin_make(gr_plugin_base** plugin) { *plugin = &grPluginInstance(); }                  \
    __attribute__((visibility("default"))) void gr_plugin_free(gr_plugin_base*) {}                                                          \
    }
