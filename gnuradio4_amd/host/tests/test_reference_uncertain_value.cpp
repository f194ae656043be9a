// The reference's own meta/UncertainValue.hpp -- included from /root/reference as it is (REF_UNCERTAIN_HPP), its <gnuradio-4.0/meta/utils.hpp> resolved by this host
// layer's forwarding header -- evaluated on a grid of operand pairs: the propagation rules of the oracle (oracle/gr4_oracle.c: op_uf32 / op_uf64), of this layer's
// gr::UncertainValue (gr4/core.hpp) and of the device (csrc/math.hip: apply_uop) are compared with these numbers by tests/test_host_cpp.py.  Nothing of the header
// is kept in the repository; GR4_COMPAT_NO_UNCERTAIN_VALUE keeps the layer's own definition out of this translation unit.
// Without REF_UNCERTAIN_HPP the same table comes from the layer's own type (what the drop-in user gets where the reference's header is not on the include path).
#ifdef REF_UNCERTAIN_HPP
#define GR4_COMPAT_NO_UNCERTAIN_VALUE
#include REF_UNCERTAIN_HPP
#else
#include <gnuradio-4.0/meta/UncertainValue.hpp>
#endif

#include <cstdio>

template <typename F>
void table(const char* tag) {
    using U = gr::UncertainValue<F>;
    static_assert(gr::UncertainValueLike<U> && !gr::UncertainValueLike<F>);
    const F vals[] = {F(10), F(-3.5), F(0.125), F(16), F(4), F(-20), F(1e-3), F(777.25)};
    const F uncs[] = {F(3), F(0.5), F(0), F(2), F(0.25), F(4)};
    for (F av : vals)
        for (F au : uncs)
            for (F bv : vals)
                for (F bu : uncs) {
                    const U a{av, au}, b{bv, bu};
                    const U r[4] = {a + b, a - b, a * b, a / b};
                    std::printf("%s %.17g %.17g %.17g %.17g", tag, double(av), double(au), double(bv), double(bu));
                    for (const U& x : r) std::printf(" %.17g %.17g", double(x.value), double(x.uncertainty));
                    std::printf("\n");
                }
}

int main() {
    table<float>("f32");
    table<double>("f64");
#ifdef REF_UNCERTAIN_HPP
    std::printf("reference UncertainValue.hpp unmodified: done\n");
#else
    std::printf("host layer gr::UncertainValue: done\n");
#endif
    return 0;
}
