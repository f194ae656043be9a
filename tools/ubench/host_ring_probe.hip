// developer probe: is a double-mapped (memfd + two mmaps) host ring registrable with hipHostRegister, and does the copy engine read a span that wraps its end at full rate?
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <unistd.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include <chrono>
#ifndef MFD_CLOEXEC
#define MFD_CLOEXEC 1U
#endif
extern "C" int memfd_create(const char*, unsigned);
int main() {
    const size_t half = (size_t)64 << 20; // 64 MiB ring
    int fd = memfd_create("gr4ring", MFD_CLOEXEC);
    if (fd < 0 || ftruncate(fd, (off_t)half)) { perror("memfd"); return 1; }
    char* base = (char*)mmap(nullptr, 2 * half, PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (base == MAP_FAILED) { perror("reserve"); return 1; }
    if (mmap(base, half, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_FIXED, fd, 0) == MAP_FAILED || mmap(base + half, half, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_FIXED, fd, 0) == MAP_FAILED) { perror("map"); return 1; }
    close(fd);
    for (size_t i = 0; i < half / 4; ++i) ((unsigned*)base)[i] = (unsigned)i * 2654435761u;
    printf("alias ok: %d\n", ((unsigned*)base)[5] == ((unsigned*)(base + half))[5]);
    hipError_t e = hipHostRegister(base, 2 * half, hipHostRegisterDefault);
    printf("hipHostRegister(whole 2x range): %s\n", hipGetErrorString(e));
    if (e != hipSuccess) {
        (void)hipGetLastError();
        hipError_t e1 = hipHostRegister(base, half, hipHostRegisterDefault), e2 = hipHostRegister(base + half, half, hipHostRegisterDefault);
        printf("hipHostRegister(halves): %s / %s\n", hipGetErrorString(e1), hipGetErrorString(e2));
        if (e1 != hipSuccess || e2 != hipSuccess) return 2;
    }
    void* d = nullptr;
    hipMalloc(&d, half);
    hipStream_t st; hipStreamCreate(&st);
    auto time = [&](const char* what, const void* src, size_t n) {
        hipMemcpyAsync(d, src, n, hipMemcpyHostToDevice, st); hipStreamSynchronize(st);
        auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < 10; ++r) hipMemcpyAsync(d, src, n, hipMemcpyHostToDevice, st);
        hipStreamSynchronize(st);
        double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("%-44s %.1f GB/s\n", what, 10.0 * n / s / 1e9);
    };
    const size_t n = half / 2;
    time("ring, span inside the first mapping", base + half / 8, n);
    time("ring, span that wraps the physical end", base + half - n / 2, n);
    std::vector<unsigned> back(n / 4);
    hipMemcpyAsync(d, base + half - n / 2, n, hipMemcpyHostToDevice, st);
    hipMemcpyAsync(back.data(), d, n, hipMemcpyDeviceToHost, st); hipStreamSynchronize(st);
    printf("wrapped span intact: %d\n", memcmp(back.data(), base + half - n / 2, n) == 0);
    void* hp = nullptr; hipHostMalloc(&hp, n, hipHostMallocDefault); memset(hp, 1, n);
    time("hipHostMalloc'ed buffer", hp, n);
    return 0;
}
