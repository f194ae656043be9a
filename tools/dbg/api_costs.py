"""host cost of the runtime calls a DeviceRun makes per piece (round 6, batching at 65 536-item edges)"""
import ctypes as C, time
import gnuradio4_amd as G
L = G.capi.lib()
s_in, s_out, ev = C.c_void_p(), C.c_void_p(), C.c_void_p()
L.gr4hip_stream_create(C.byref(s_in)); L.gr4hip_stream_create(C.byref(s_out)); L.gr4hip_event_create(C.byref(ev))
h, d = C.c_void_p(), C.c_void_p()
L.gr4hip_malloc_host(C.byref(h), 1 << 22); L.gr4hip_malloc(C.byref(d), 1 << 22)
done = C.c_int(0)
def t(f, n=2000):
    L.gr4hip_stream_synchronize(s_in); L.gr4hip_stream_synchronize(s_out)
    t0 = time.perf_counter()
    for _ in range(n): f()
    dt = (time.perf_counter() - t0) / n * 1e6
    L.gr4hip_stream_synchronize(s_in); L.gr4hip_stream_synchronize(s_out)
    tot = (time.perf_counter() - t0) / n * 1e6
    return dt, tot
for kb in (128, 256, 512, 1024):
    print("h2d %4d KiB: %.2f us per call, %.2f us per copy sustained (%.1f GB/s)" % ((kb,) + t(lambda: L.gr4hip_memcpy_h2d(d, h, kb << 10, s_in)) + (kb * 1024 / t(lambda: L.gr4hip_memcpy_h2d(d, h, kb << 10, s_in))[1] / 1e3,)))
    print("d2h %4d KiB: %.2f us per call, %.2f us per copy sustained" % ((kb,) + t(lambda: L.gr4hip_memcpy_d2h(h, d, kb << 10, s_out))))
print("event record: %.2f us" % t(lambda: L.gr4hip_event_record(ev, s_in))[0])
print("event query : %.2f us" % t(lambda: L.gr4hip_event_query(ev, C.byref(done)))[0])
print("noop ctypes call (gr4hip_abi_version): %.2f us" % t(lambda: L.gr4hip_abi_version())[0])
def both():
    L.gr4hip_memcpy_h2d(d, h, 256 << 10, s_in); L.gr4hip_memcpy_d2h(h, d, 128 << 10, s_out)
print("h2d 256 KiB + d2h 128 KiB on two streams: %.2f us per pair call, %.2f us sustained" % t(both))
