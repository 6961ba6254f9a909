// pinned_copy.hip — how fast does a host thread read pinned staging memory? (r04, pool design)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(e_)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void probe(const char* name, void* src, void* dst, size_t bytes)
{
    std::memcpy(dst, src, bytes);
    double t0 = now();
    for (int i = 0; i < 5; i++) std::memcpy(dst, src, bytes);
    double t = (now() - t0) / 5;
    printf("%-44s read  %.2f GB/s\n", name, bytes / t / 1e9);
    t0 = now();
    for (int i = 0; i < 5; i++) std::memcpy(src, dst, bytes);
    t = (now() - t0) / 5;
    printf("%-44s write %.2f GB/s\n", name, bytes / t / 1e9);
}
int main()
{
    const size_t bytes = (size_t)256 << 20;
    void* dst = std::malloc(bytes);
    std::memset(dst, 1, bytes);
    void* plain = std::malloc(bytes);
    std::memset(plain, 2, bytes);
    probe("malloc", plain, dst, bytes);
    void* dev; CK(hipMalloc(&dev, bytes));
    for (unsigned flags : {(unsigned)hipHostMallocDefault, (unsigned)hipHostMallocNonCoherent, (unsigned)hipHostMallocCoherent, (unsigned)hipHostMallocPortable}) {
        void* p; CK(hipHostMalloc(&p, bytes, flags));
        std::memset(p, 3, bytes);
        char nm[64]; snprintf(nm, sizeof nm, "hipHostMalloc flags 0x%x", flags);
        probe(nm, p, dst, bytes);
        CK(hipMemcpy(p, dev, bytes, hipMemcpyDeviceToHost));
        snprintf(nm, sizeof nm, "hipHostMalloc flags 0x%x after D2H", flags);
        probe(nm, p, dst, bytes);
        double t0 = now();
        CK(hipMemcpy(p, dev, bytes, hipMemcpyDeviceToHost));
        printf("   D2H into it: %.2f GB/s\n", bytes / (now() - t0) / 1e9);
        CK(hipHostFree(p));
    }
    void* reg = std::malloc(bytes);
    std::memset(reg, 4, bytes);
    CK(hipHostRegister(reg, bytes, hipHostRegisterDefault));
    probe("malloc + hipHostRegister", reg, dst, bytes);
    double t0 = now();
    CK(hipMemcpy(reg, dev, bytes, hipMemcpyDeviceToHost));
    printf("   D2H into it: %.2f GB/s\n", bytes / (now() - t0) / 1e9);
    t0 = now();
    CK(hipMemcpy(plain, dev, bytes, hipMemcpyDeviceToHost));
    printf("   D2H into pageable malloc: %.2f GB/s\n", bytes / (now() - t0) / 1e9);
    return 0;
}
