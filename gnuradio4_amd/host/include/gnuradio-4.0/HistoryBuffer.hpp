// forwarding header: gr::HistoryBuffer<T> for block sources written against the reference's include tree (core/include/gnuradio-4.0/HistoryBuffer.hpp:130-139:
// push_front makes the new sample element [0]; a power-of-two capacity; reads beyond size() see value-initialised samples)
#pragma once
#include "../gr4/compat.hpp"

#include <bit>
#include <vector>

#include <span>

namespace gr {
// HistoryBuffer<T, N>: N == std::dynamic_extent (default) takes its capacity at run time, a fixed N at compile time (HistoryBuffer.hpp:67: the Section of
// FilterTool.hpp:187 uses the two-parameter form)
template <typename T, std::size_t N = std::dynamic_extent>
class HistoryBuffer {
    std::vector<T> _d;
    std::size_t    _cap, _head = 0, _size = 0;

public:
    using value_type = T;
    HistoryBuffer() requires(N != std::dynamic_extent) : _d(2 * std::bit_ceil(N), T{}), _cap(std::bit_ceil(N)) {}
    explicit HistoryBuffer(std::size_t capacity) requires(N == std::dynamic_extent)
        : _d(2 * std::bit_ceil(std::max<std::size_t>(capacity, 1)), T{}), _cap(std::bit_ceil(std::max<std::size_t>(capacity, 1))) {
        if (capacity == 0) throw std::out_of_range("capacity is zero");
    }
    // the storage is mirrored ([0, cap) == [cap, 2 cap)): the newest-first window [head, head + cap) is always contiguous, as upstream
    void push_front(const T& v) noexcept {
        _head          = (_head + _cap - 1) & (_cap - 1);
        _d[_head]      = v;
        _d[_head + _cap] = v;
        if (_size < _cap) ++_size;
    }
    [[nodiscard]] const T& operator[](std::size_t i) const noexcept { return _d[_head + i]; }
    [[nodiscard]] T&       operator[](std::size_t i) noexcept { return _d[_head + i]; }
    [[nodiscard]] std::size_t capacity() const noexcept { return _cap; }
    [[nodiscard]] std::size_t size() const noexcept { return _size; }
    [[nodiscard]] auto begin() const noexcept { return _d.begin() + static_cast<std::ptrdiff_t>(_head); }
    [[nodiscard]] auto end() const noexcept { return begin() + static_cast<std::ptrdiff_t>(_cap); }
    [[nodiscard]] auto cbegin() const noexcept { return begin(); }
    [[nodiscard]] auto cend() const noexcept { return end(); }
    void reset(T v = T{}) noexcept { std::fill(_d.begin(), _d.end(), v); _size = 0; _head = 0; } // HistoryBuffer.hpp: back to the empty state, storage kept
    [[nodiscard]] const T& front() const noexcept { return _d[_head]; }
    [[nodiscard]] const T& back() const noexcept { return _d[_head + (_size ? _size - 1 : 0)]; }
};
} // namespace gr
