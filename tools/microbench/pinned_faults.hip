// pinned_faults.hip — do CPU reads of a pinned buffer fault after every device-to-host DMA into it? (r04, pool design)
#include <hip/hip_runtime.h>
#include <sys/resource.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(e_)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static long minflt() { rusage r; getrusage(RUSAGE_SELF, &r); return r.ru_minflt; }
static double systime() { rusage r; getrusage(RUSAGE_SELF, &r); return r.ru_stime.tv_sec + 1e-6 * r.ru_stime.tv_usec; }
static void run(const char* name, char* host, void* dev, size_t bytes, int threads, hipStream_t s)
{
    std::vector<std::vector<char>> dst((size_t)threads, std::vector<char>(bytes / threads, 1));
    for (int rep = 0; rep < 2; rep++) {
        long f0 = minflt(); double s0 = systime(), t0 = now();
        for (int it = 0; it < 8; it++) {
            CK(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, s));
            CK(hipStreamSynchronize(s));
            std::vector<std::thread> th;
            for (int t = 0; t < threads; t++)
                th.emplace_back([&, t] { std::memcpy(dst[(size_t)t].data(), host + (size_t)t * (bytes / threads), bytes / threads); });
            for (auto& x : th) x.join();
        }
        printf("%-40s threads %3d rep %d: %6.1f ms per round, minor faults per round %ld (pages %zu), sys %.3f s\n", name, threads, rep,
               (now() - t0) / 8 * 1e3, (minflt() - f0) / 8, bytes / 4096, systime() - s0);
    }
}
int main()
{
    const size_t bytes = (size_t)128 << 20;
    void* dev; CK(hipMalloc(&dev, bytes));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    char* p; CK(hipHostMalloc((void**)&p, bytes, hipHostMallocDefault));
    std::memset(p, 3, bytes);
    run("hipHostMalloc default", p, dev, bytes, 1, s);
    run("hipHostMalloc default", p, dev, bytes, 32, s);
    char* q; CK(hipHostMalloc((void**)&q, bytes, hipHostMallocNonCoherent));
    std::memset(q, 3, bytes);
    run("hipHostMalloc non-coherent", q, dev, bytes, 32, s);
    char* r = (char*)aligned_alloc(4096, bytes);
    std::memset(r, 4, bytes);
    CK(hipHostRegister(r, bytes, hipHostRegisterDefault));
    run("malloc + hipHostRegister", r, dev, bytes, 32, s);
    char* m = (char*)aligned_alloc(4096, bytes);
    std::memset(m, 5, bytes);
    run("pageable malloc", m, dev, bytes, 32, s);
    return 0;
}
