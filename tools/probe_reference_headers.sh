#!/bin/bash
# container-only developer tool: how far do the reference's two headline block headers get against this host layer?  Nothing of the reference is copied:
# the headers are #included where they lie under /root/reference.  Prints the FIRST error of each include-path configuration.
R=/root/reference; H=$(cd "$(dirname "$0")/.." && pwd)/gnuradio4_amd/host/include
[ -d $R ] || { echo "no /root/reference here"; exit 0; }
T=$(mktemp -d)
probe() { # name header extra-include-flags...
  local name=$1 hdr=$2; shift 2
  echo "#include <$hdr>" > $T/p.cpp; echo "int main() { return 0; }" >> $T/p.cpp
  local first
  first=$(g++ -std=c++20 -fsyntax-only -I $H "$@" $T/p.cpp 2>&1 | grep -E "error" | head -1 | sed "s#$R/##g; s#$H/##g")
  printf '%-64s %s\n' "$name" "${first:-compiles}"
}
g++ --version | head -1
for hdr in gnuradio-4.0/filter/time_domain_filter.hpp gnuradio-4.0/fourier/fft.hpp; do
  B=$R/blocks/filter/include; [ $hdr = gnuradio-4.0/fourier/fft.hpp ] && B=$R/blocks/fourier/include
  echo "== $hdr"
  probe "this layer's forwarding headers only"                       $hdr -I $B
  probe "+ reference algorithm/include"                              $hdr -I $B -I $R/algorithm/include
  probe "+ third_party/magic_enum (vendored)"                        $hdr -I $B -I $R/algorithm/include -I $R/third_party/magic_enum
  probe "+ reference core/include, meta/include behind this layer's" $hdr -I $B -I $R/algorithm/include -I $R/third_party/magic_enum -I $R/core/include -I $R/meta/include
done
echo "== what the toolchain lacks"
for h in format expected print; do echo "#include <$h>" > $T/q.cpp; printf '<%s>: ' $h; g++ -std=c++20 -fsyntax-only $T/q.cpp 2>&1 | grep -qE "error" && echo "absent (libstdc++ $(g++ -dumpversion))" || echo present; done
grep -n "Tensor<T>" $R/blocks/filter/include/gnuradio-4.0/filter/time_domain_filter.hpp | head -2 | sed "s#$R/##"
rm -rf $T
