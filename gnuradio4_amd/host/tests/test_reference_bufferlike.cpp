// SURVEY.md 8(f) row 1, "GPU-resident BufferLike": the reference's own buffer concepts (core/include/gnuradio-4.0/Buffer.hpp:78-102, included from
// /root/reference where it lies, unmodified -- it only needs the standard library) applied to this layer's HBM ring.  Container-only, compile-time only.
#include <gnuradio-4.0/Buffer.hpp> // the reference's file

#include <gr4/hip.hpp>

static_assert(gr::BufferLike<gr::hip::CircularBuffer<float>>, "hip::CircularBuffer<T> must model the reference's BufferLike");
static_assert(gr::BufferLike<gr::hip::CircularBuffer<std::complex<float>>>);
static_assert(gr::BufferReaderLike<decltype(std::declval<gr::hip::CircularBuffer<float>&>().new_reader())>);
static_assert(gr::BufferWriterLike<decltype(std::declval<gr::hip::CircularBuffer<float>&>().new_writer())>);
int main() { return 0; }
