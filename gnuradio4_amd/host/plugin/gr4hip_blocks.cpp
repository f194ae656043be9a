// libgr4hip_blocks.so -- the hot-path blocks as a GNU Radio 4 plugin (core/include/gnuradio-4.0/Plugin.hpp:82-85).
//
// Exports gr_plugin_make / gr_plugin_free; the blocks are registered under the names the reference's registry uses for them
// (GR_REGISTER_BLOCK lines of time_domain_filter.hpp:20-213, Math.hpp:25-71, Rotator.hpp:15, fft.hpp:29; portable type names
// float32 / complex<float32> / int32 ... of meta/utils.hpp:481-490), so a graph description that names
//     gr::filter::fir_filter<float32>   with   compute_domain: "gpu:hip:0"
// resolves to the block whose work() goes through the device seam (gr4/hip.hpp) into libgr4hip.so.  With compute_domain "host" (the
// default) the same block runs its host body.
#include <gr4/hip.hpp>
#include <gr4/plugin.hpp>

GR_PLUGIN("gr4hip hot-path blocks (MI355X)", "gr4-hip", "LGPL-3.0-or-later", "r01")

namespace {
using namespace gr;

template <typename T> constexpr std::string_view portable() {
    if constexpr (std::is_same_v<T, std::uint8_t>) return "uint8"; else if constexpr (std::is_same_v<T, std::uint16_t>) return "uint16";
    else if constexpr (std::is_same_v<T, std::uint32_t>) return "uint32"; else if constexpr (std::is_same_v<T, std::uint64_t>) return "uint64";
    else if constexpr (std::is_same_v<T, std::int8_t>) return "int8"; else if constexpr (std::is_same_v<T, std::int16_t>) return "int16";
    else if constexpr (std::is_same_v<T, std::int32_t>) return "int32"; else if constexpr (std::is_same_v<T, std::int64_t>) return "int64";
    else if constexpr (std::is_same_v<T, float>) return "float32"; else if constexpr (std::is_same_v<T, double>) return "float64";
    else if constexpr (std::is_same_v<T, std::complex<float>>) return "complex<float32>"; else return "complex<float64>";
}
template <typename T> std::string named(std::string_view base, std::string_view extra = "") { return std::string(base) + "<" + std::string(portable<T>()) + std::string(extra) + ">"; }

template <typename T>
void register_math(BlockRegistry& r) {
    using namespace gr::blocks::math;
    r.insert<AddConst<T>>(named<T>("gr::blocks::math::AddConst"));
    r.insert<SubtractConst<T>>(named<T>("gr::blocks::math::SubtractConst"));
    r.insert<MultiplyConst<T>>(named<T>("gr::blocks::math::MultiplyConst"));
    r.insert<DivideConst<T>>(named<T>("gr::blocks::math::DivideConst"));
    r.insert<Add<T>>(named<T>("gr::blocks::math::Add"));
    r.insert<Subtract<T>>(named<T>("gr::blocks::math::Subtract"));
    r.insert<Multiply<T>>(named<T>("gr::blocks::math::Multiply"));
    r.insert<Divide<T>>(named<T>("gr::blocks::math::Divide"));
    r.insert<gr::filter::Decimator<T>>(named<T>("gr::filter::Decimator"));
}
template <typename T>
void register_io(BlockRegistry& r) { // sources / sinks / converters so that a whole graph can come out of the registry
    r.insert<gr::testing::VectorSource<T>>(named<T>("gr::testing::VectorSource"));
    r.insert<gr::testing::VectorSink<T>>(named<T>("gr::testing::VectorSink"));
    r.insert<gr::testing::NullSink<T>>(named<T>("gr::testing::NullSink"));
    r.insert<gr::hip::H2D<T>>(named<T>("gr::hip::H2D"));
    r.insert<gr::hip::D2H<T>>(named<T>("gr::hip::D2H"));
}

const bool registered = [] {
    BlockRegistry& r = grPluginInstance();
    using namespace gr::filter;
    register_math<std::uint8_t>(r); register_math<std::uint16_t>(r); register_math<std::uint32_t>(r); register_math<std::uint64_t>(r);
    register_math<std::int8_t>(r); register_math<std::int16_t>(r); register_math<std::int32_t>(r); register_math<std::int64_t>(r);
    register_math<float>(r); register_math<double>(r); register_math<std::complex<float>>(r); register_math<std::complex<double>>(r);
    register_io<float>(r); register_io<std::complex<float>>(r); register_io<std::int32_t>(r);
    r.insert<fir_filter<float>>(named<float>("gr::filter::fir_filter"));
    r.insert<fir_filter<double>>(named<double>("gr::filter::fir_filter")); // FP64 kernel behind the seam (csrc/f64.hip)
    r.insert<fir_filter<std::complex<float>>>(named<std::complex<float>>("gr::filter::fir_filter"));
    r.insert<iir_filter<float, IIRForm::DF_I>>(named<float>("gr::filter::iir_filter", ", gr::filter::IIRForm::DF_I"));
    r.insert<iir_filter<float, IIRForm::DF_II>>(named<float>("gr::filter::iir_filter", ", gr::filter::IIRForm::DF_II"));
    r.insert<iir_filter<float, IIRForm::DF_I_TRANSPOSED>>(named<float>("gr::filter::iir_filter", ", gr::filter::IIRForm::DF_I_TRANSPOSED"));
    r.insert<iir_filter<float, IIRForm::DF_II_TRANSPOSED>>(named<float>("gr::filter::iir_filter", ", gr::filter::IIRForm::DF_II_TRANSPOSED"));
    r.insert<iir_filter<double, IIRForm::DF_I>>(named<double>("gr::filter::iir_filter", ", gr::filter::IIRForm::DF_I"));
    r.insert<iir_filter<double, IIRForm::DF_II>>(named<double>("gr::filter::iir_filter", ", gr::filter::IIRForm::DF_II"));
    r.insert<iir_filter<double, IIRForm::DF_I_TRANSPOSED>>(named<double>("gr::filter::iir_filter", ", gr::filter::IIRForm::DF_I_TRANSPOSED"));
    r.insert<iir_filter<double, IIRForm::DF_II_TRANSPOSED>>(named<double>("gr::filter::iir_filter", ", gr::filter::IIRForm::DF_II_TRANSPOSED"));
    r.insert<BasicFilter<float>>(named<float>("gr::filter::BasicFilter"));
    r.insert<BasicDecimatingFilter<float>>(named<float>("gr::filter::BasicFilterProto", ", gr::Resampling<1, 1, false>"));
    r.insert<gr::blocks::math::Rotator<std::complex<float>>>(named<std::complex<float>>("gr::blocks::math::Rotator"));
    r.insert<gr::blocks::math::Rotator<std::complex<double>>>(named<std::complex<double>>("gr::blocks::math::Rotator"));
    r.insert<gr::blocks::fft::FFT<double, gr::DataSet<double>>>(named<double>("gr::blocks::fft::FFT"));
    r.insert<gr::blocks::fft::FFT<float>>(named<float>("gr::blocks::fft::FFT"));
    r.insert<gr::blocks::fft::FFT<std::complex<float>>>(named<std::complex<float>>("gr::blocks::fft::FFT"));
    r.insert<gr::blocks::fft::PowerSpectrum<std::complex<float>>>(named<std::complex<float>>("gr::blocks::fft::PowerSpectrum"));
    r.insert<gr::hip::OnDevice<fir_filter<float>>>(named<float>("gr::hip::OnDevice<gr::filter::fir_filter", ">"));
    r.insert<gr::hip::OnDevice<fir_filter<std::complex<float>>>>(named<std::complex<float>>("gr::hip::OnDevice<gr::filter::fir_filter", ">"));
    r.insert<gr::hip::OnDevice<gr::blocks::fft::PowerSpectrum<std::complex<float>>>>(named<std::complex<float>>("gr::hip::OnDevice<gr::blocks::fft::PowerSpectrum", ">"));
    gr::hip::register_provider(); // edges with EdgeParameters{.domain = "gpu:hip:i"} get pinned pages once the plugin is loaded
    return true;
}();
} // namespace
