// Read-bandwidth ceiling of cmux_kernel's selector stream, by access pattern (no arithmetic).
// 4096 gates x 256 KiB, every byte read once; 256 threads per workgroup, two gates per workgroup, two workgroups per CU
// (64 KiB of dynamic LDS declared), each wave keeps `DEPTH` KiB in flight — the shape of cmux_kernel<4,4,2>.
//   pattern 0: the kernel's layout  — per (row, q): wave w reads [4w, 4w+4) KiB and [8+4w, 8+4w+4) KiB of the 16 KiB
//   pattern 1: wave-contiguous row  — per row: wave w reads 16 KiB contiguous (row layout [w][q][bins])
//   pattern 2: wave-contiguous gate — wave w reads its 128 KiB of the gate contiguously
//   pattern 3: streaming            — workgroup reads its two gates (512 KiB) front to back, 4 KiB per instruction group
// build: hipcc --offload-arch=gfx950 -O3 -o gpurun_out/ggsw_read_patterns tools/microbench/ggsw_read_patterns.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef double v2 __attribute__((ext_vector_type(2)));

template <int PATTERN, int DEPTH, bool NT = true>
__global__ __launch_bounds__(256, 2) void read_kernel(const v2* __restrict__ g, double* sink, uint32_t gates)
{
    extern __shared__ char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, slot = wv >> 1, w = wv & 1;
    const uint32_t gate = blockIdx.x * 2 + slot;
    if (gate >= gates) return;
    const v2* base = g + (size_t)gate * (256 * 1024 / 16);
    v2 acc = {0.0, 0.0};
    constexpr int PIECES = 128; // 1 KiB pieces per wave and gate
    v2 buf[DEPTH];
    auto addr = [&](int i) -> const v2* { // piece i (0..127) of this wave, in consumption order
        if (PATTERN == 0) { // row = i / 16, q = (i / 8) & 1, r = i & 7
            const int row = i >> 4, q = (i >> 3) & 1, r = i & 7;
            return base + (size_t)row * 2048 + q * 1024 + 256 * w + 64 * (r & 3) + 512 * (r >> 2) + lane;
        } else if (PATTERN == 1) {
            const int row = i >> 4, k = i & 15;
            return base + (size_t)row * 2048 + w * 1024 + 64 * k + lane;
        } else if (PATTERN == 2) {
            return base + (size_t)w * 8192 + 64 * i + lane;
        } else { // streaming over the workgroup's two gates: wave wv takes KiB (4 t + wv) of the 512
            const v2* wg = g + (size_t)(blockIdx.x * 2) * (256 * 1024 / 16);
            return wg + (size_t)(4 * i + wv) * 64 + lane;
        }
    };
#pragma unroll
    for (int d = 0; d < DEPTH; d++) buf[d] = NT ? __builtin_nontemporal_load(addr(d)) : *addr(d);
#pragma unroll 1
    for (int i = 0; i < PIECES; i += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
            acc += buf[d];
            const int nx = i + DEPTH + d;
            buf[d] = NT ? __builtin_nontemporal_load(addr(nx < PIECES ? nx : d)) : *addr(nx < PIECES ? nx : d);
        }
    }
    if (acc.x == 12345.678) sink[tid] = acc.y + smem[0];
}


// The whole traffic shape of cmux_kernel, no arithmetic: FLAGS bit 0 = copy a 32 KiB table image from global into LDS per
// workgroup first, bit 1 = read the two 32 KiB operands of the gate (8 bytes per lane, 16-byte stride: one parity per wave),
// bit 2 = write the 32 KiB result the same way, bit 3 = operands / results as 16-byte accesses (both parities in one lane).
template <int FLAGS>
__global__ __launch_bounds__(256, 2) void traffic_kernel(const v2* __restrict__ g, const v2* __restrict__ table, const uint64_t* d0,
                                                         const uint64_t* d1, uint64_t* out, double* sink, uint32_t gates)
{
    extern __shared__ char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, slot = wv >> 1, w = wv & 1;
    const uint32_t gate = blockIdx.x * 2 + slot;
    if (gate >= gates) return;
    uint64_t x[64];
    uint64_t xs = 0;
    if (FLAGS & 2) {
        if (FLAGS & 8) {
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const v2 a = *reinterpret_cast<const v2*>(d1 + (size_t)gate * 4096 + (size_t)(w * 16 + e) * 128 + 2 * lane);
                const v2 b = *reinterpret_cast<const v2*>(d0 + (size_t)gate * 4096 + (size_t)(w * 16 + e) * 128 + 2 * lane);
                x[4 * e] = __double_as_longlong(a.x); x[4 * e + 1] = __double_as_longlong(a.y);
                x[4 * e + 2] = __double_as_longlong(b.x); x[4 * e + 3] = __double_as_longlong(b.y);
            }
        } else {
#pragma unroll
            for (int e = 0; e < 32; e++) {
                x[2 * e] = d1[(size_t)gate * 4096 + e * 128 + 2 * lane + w];
                x[2 * e + 1] = d0[(size_t)gate * 4096 + e * 128 + 2 * lane + w];
            }
        }
        asm volatile("" ::: "memory");
    }
    if (FLAGS & 1) {
        v2* dst = reinterpret_cast<v2*>(smem);
        for (int i = tid; i < 2040; i += 256) dst[i] = table[i];
    }
    if (FLAGS & 2) {
#pragma unroll
        for (int e = 0; e < 64; e++) xs += x[e];
    }
    __syncthreads();
    const v2* base = g + (size_t)gate * (256 * 1024 / 16);
    v2 acc = {0.0, 0.0};
    v2 buf[16];
    auto addr = [&](int i) -> const v2* {
        const int row = i >> 4, q = (i >> 3) & 1, r = i & 7;
        return base + (size_t)row * 2048 + q * 1024 + 256 * w + 64 * (r & 3) + 512 * (r >> 2) + lane;
    };
#pragma unroll
    for (int d = 0; d < 16; d++) buf[d] = __builtin_nontemporal_load(addr(d));
#pragma unroll 1
    for (int i = 0; i < 128; i += 16) {
#pragma unroll
        for (int d = 0; d < 16; d++) {
            acc += buf[d];
            const int nx = i + 16 + d;
            buf[d] = __builtin_nontemporal_load(addr(nx < 128 ? nx : d));
        }
    }
    if (FLAGS & 4) {
        const uint64_t v = xs + __double_as_longlong(acc.x);
        if (FLAGS & 8) {
#pragma unroll
            for (int e = 0; e < 16; e++) {
                v2 o = {__longlong_as_double(v + e), __longlong_as_double(v - e)};
                *reinterpret_cast<v2*>(out + (size_t)gate * 4096 + (size_t)(w * 16 + e) * 128 + 2 * lane) = o;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 32; e++) { if (FLAGS & 16) __builtin_nontemporal_store(v + e, out + (size_t)gate * 4096 + e * 128 + 2 * lane + w); else out[(size_t)gate * 4096 + e * 128 + 2 * lane + w] = v + e; }
        }
    }
    if (acc.x == 12345.678) sink[tid] = acc.y + smem[0] + (double)xs;
}

template <int FLAGS>
int run_traffic(const v2* d, const v2* table, const uint64_t* d0, const uint64_t* d1, uint64_t* out, double* sink, uint32_t gates)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&traffic_kernel<FLAGS>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    float best = 1e9f;
    for (int it = 0; it < 6; it++) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((traffic_kernel<FLAGS>), dim3((gates + 1) / 2), dim3(256), 65536, 0, d, table, d0, d1, out, sink, gates);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (it > 0 && ms < best) best = ms;
    }
    const double bytes = (double)gates * (262144.0 + ((FLAGS & 2) ? 65536.0 : 0.0) + ((FLAGS & 4) ? 32768.0 : 0.0));
    printf("traffic: selector rows%s%s%s%s  %.3f ms  %.0f GB/s (HBM bytes only)\n", (FLAGS & 1) ? " + table copy" : "",
           (FLAGS & 2) ? " + operands" : "", (FLAGS & 4) ? " + result" : "", (FLAGS & 8) ? " (16-byte accesses)" : ((FLAGS & 16) ? " (streaming stores)" : ""), best, bytes / best / 1e6);
    return 0;
}

template <int PATTERN, int DEPTH, bool NT = true>
int run(const v2* d, double* sink, uint32_t gates, const char* name)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&read_kernel<PATTERN, DEPTH, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    float best = 1e9f;
    for (int it = 0; it < 6; it++) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((read_kernel<PATTERN, DEPTH, NT>), dim3((gates + 1) / 2), dim3(256), 65536, 0, d, sink, gates);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (it > 0 && ms < best) best = ms;
    }
    printf("%s pattern %d depth %2d KiB/wave  %-28s %.3f ms  %.0f GB/s\n", NT ? "nontemporal" : "plain      ", PATTERN, DEPTH, name, best, (double)gates * 262144.0 / best / 1e6);
    return 0;
}

int main()
{
    const uint32_t gates = 4096;
    const size_t bytes = (size_t)gates * 262144;
    v2* d; double* sink;
    CK(hipMalloc(&d, bytes)); CK(hipMalloc(&sink, 4096));
    CK(hipMemset(d, 0, bytes));
    int rc = 0;
    rc |= run<0, 16>(d, sink, gates, "kernel layout");
    rc |= run<1, 16>(d, sink, gates, "wave-contiguous row");
    rc |= run<2, 16>(d, sink, gates, "wave-contiguous gate");
    rc |= run<3, 16>(d, sink, gates, "streaming");
    rc |= run<0, 32>(d, sink, gates, "kernel layout");
    rc |= run<2, 32>(d, sink, gates, "wave-contiguous gate");
    rc |= run<3, 32>(d, sink, gates, "streaming");
    rc |= run<0, 16, false>(d, sink, gates, "kernel layout");
    rc |= run<3, 16, false>(d, sink, gates, "streaming");
    rc |= run<0, 8>(d, sink, gates, "kernel layout");
    rc |= run<3, 8>(d, sink, gates, "streaming");
    v2* table; uint64_t *d0, *d1, *out;
    CK(hipMalloc(&table, 32768)); CK(hipMemset(table, 0, 32768));
    CK(hipMalloc(&d0, (size_t)gates * 32768)); CK(hipMalloc(&d1, (size_t)gates * 32768)); CK(hipMalloc(&out, (size_t)gates * 32768));
    CK(hipMemset(d0, 0, (size_t)gates * 32768)); CK(hipMemset(d1, 0, (size_t)gates * 32768));
    rc |= run_traffic<0>(d, table, d0, d1, out, sink, gates);
    rc |= run_traffic<1>(d, table, d0, d1, out, sink, gates);
    rc |= run_traffic<2>(d, table, d0, d1, out, sink, gates);
    rc |= run_traffic<3>(d, table, d0, d1, out, sink, gates);
    rc |= run_traffic<4>(d, table, d0, d1, out, sink, gates);
    rc |= run_traffic<7>(d, table, d0, d1, out, sink, gates);
    rc |= run_traffic<15>(d, table, d0, d1, out, sink, gates);
    rc |= run_traffic<20>(d, table, d0, d1, out, sink, gates);
    rc |= run_traffic<23>(d, table, d0, d1, out, sink, gates);
    return rc;
}
