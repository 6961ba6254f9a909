// lds_valu_overlap.hip — can a wave's LDS exchanges travel under f64 arithmetic on MI355X, and in which arrangement? (r04)
//
// fft_pair_bench shows a transform pair costing (almost) its VALU time PLUS its LDS time.  This probe strips the pattern to
// its bones: per iteration a wave issues NW ds_write_b128, NR ds_read_b128 of what it wrote (own 8 KiB image, the
// conflict-free image of fft512_pair1) and NV independent v_fma_f64, hand-written as inline asm so that the order is exactly
// the one named:
//   valu     NV fma only                                                     (VALU floor)
//   lds      NW writes, NR reads, wait                                       (LDS floor)
//   burst    NW writes, NR reads, then NV fma, then wait                     (what a lone exchange under the other transform's butterflies is)
//   spread   (1 write, NV/NW fma) x NW, (1 read, ...) ... , wait at the end  (one LDS instruction every few fma)
//   wfirst   NW writes, then NV/2 fma, NR reads, NV/2 fma, wait              (the shipped order: reads one block behind their writes)
// with 8 waves per workgroup (two per SIMD) and with 4 (one per SIMD), one workgroup per CU, every CU busy.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/microbench/bin/lds_valu_overlap tools/microbench/lds_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(e_)); exit(1); } } while (0)

typedef double d2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void w128(uint32_t addr, d2 v) { asm volatile("ds_write_b128 %0, %1" :: "v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ d2 r128(uint32_t addr) { d2 v; asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory"); return v; }
__device__ __forceinline__ void fma4(double (&x)[8], double a, double b, int base)
{
    // four independent chains per call
    asm volatile("v_fma_f64 %0, %4, %5, %0\n\tv_fma_f64 %1, %4, %5, %1\n\tv_fma_f64 %2, %4, %5, %2\n\tv_fma_f64 %3, %4, %5, %3"
                 : "+v"(x[base]), "+v"(x[base + 1]), "+v"(x[base + 2]), "+v"(x[base + 3]) : "v"(a), "v"(b));
}
__device__ __forceinline__ void wait_lds() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

template <int MODE, int NV>
__global__ __launch_bounds__(512, 2) void probe(unsigned long long* out, int iters)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t buf = (uint32_t)(size_t)(const __attribute__((address_space(3))) void*)(smem + wv * 8192);
    const int hi3 = lane >> 3, lo3 = lane & 7;
    uint32_t wr[8], rd[8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        wr[r] = buf + (((16 * (64 * hi3 + lo3)) ^ (16 * r)) + 128 * r);
        rd[r] = buf + 1024 * r + 16 * (8 * lo3 + (hi3 ^ lo3));
    }
    d2 v[8];
#pragma unroll
    for (int r = 0; r < 8; r++) v[r] = d2{1.0 + lane + r, 2.0 + wv};
    double x[8];
#pragma unroll
    for (int r = 0; r < 8; r++) x[r] = 0.001 * (lane + r);
    const double a = 0.999999, b = 1e-9;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        if constexpr (MODE == 0) {
#pragma unroll
            for (int k = 0; k < NV / 4; k++) fma4(x, a, b, 4 * (k & 1));
        } else if constexpr (MODE == 1) {
#pragma unroll
            for (int r = 0; r < 8; r++) w128(wr[r], v[r]);
#pragma unroll
            for (int r = 0; r < 8; r++) v[r] = r128(rd[r]);
            wait_lds();
        } else if constexpr (MODE == 2) {
#pragma unroll
            for (int r = 0; r < 8; r++) w128(wr[r], v[r]);
#pragma unroll
            for (int r = 0; r < 8; r++) v[r] = r128(rd[r]);
#pragma unroll
            for (int k = 0; k < NV / 4; k++) fma4(x, a, b, 4 * (k & 1));
            wait_lds();
        } else if constexpr (MODE == 3) {
#pragma unroll
            for (int r = 0; r < 8; r++) {
                w128(wr[r], v[r]);
#pragma unroll
                for (int k = 0; k < NV / 64; k++) fma4(x, a, b, 4 * (k & 1));
            }
#pragma unroll
            for (int r = 0; r < 8; r++) {
                v[r] = r128(rd[r]);
#pragma unroll
                for (int k = 0; k < NV / 64; k++) fma4(x, a, b, 4 * (k & 1));
            }
            wait_lds();
        } else if constexpr (MODE == 4) {
#pragma unroll
            for (int r = 0; r < 8; r++) w128(wr[r], v[r]);
#pragma unroll
            for (int k = 0; k < NV / 8; k++) fma4(x, a, b, 4 * (k & 1));
#pragma unroll
            for (int r = 0; r < 8; r++) v[r] = r128(rd[r]);
#pragma unroll
            for (int k = 0; k < NV / 8; k++) fma4(x, a, b, 4 * (k & 1));
            wait_lds();
        } else if constexpr (MODE == 5) { // writes only, spread: is it the store that does not overlap?
#pragma unroll
            for (int r = 0; r < 8; r++) {
                w128(wr[r], v[r]);
#pragma unroll
                for (int k = 0; k < NV / 32; k++) fma4(x, a, b, 4 * (k & 1));
            }
        } else if constexpr (MODE == 6) { // reads only, spread
#pragma unroll
            for (int r = 0; r < 8; r++) {
                v[r] = r128(rd[r]);
#pragma unroll
                for (int k = 0; k < NV / 32; k++) fma4(x, a, b, 4 * (k & 1));
            }
            wait_lds();
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
#pragma unroll
    for (int r = 0; r < 8; r++) s += x[r] + v[r].x + v[r].y;
    if (lane == 0) {
        out[((size_t)blockIdx.x * 8 + wv) * 2] = t1 - t0;
        out[((size_t)blockIdx.x * 8 + wv) * 2 + 1] = (unsigned long long)__double_as_longlong(s);
    }
}

template <int MODE, int NV> static void run(const char* name, unsigned long long* d_out, int n_cu, int iters, int threads)
{
    const int lds = 8 * 8192;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<MODE, NV>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL((probe<MODE, NV>), dim3(n_cu), dim3(threads), lds, 0, d_out, 16);
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL((probe<MODE, NV>), dim3(n_cu), dim3(threads), lds, 0, d_out, iters);
    CK(hipDeviceSynchronize());
    const int waves = threads / 64;
    std::vector<unsigned long long> h((size_t)n_cu * 16);
    CK(hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> c;
    for (int b = 0; b < n_cu; b++)
        for (int w = 0; w < waves; w++) c.push_back((double)h[((size_t)b * 8 + w) * 2] / iters);
    std::sort(c.begin(), c.end());
    printf("%-10s NV=%3d waves/CU=%d  cycles/iteration: median %7.0f  max %7.0f\n", name, NV, waves, c[c.size() / 2], c.back());
}

template <int NV> static void sweep(unsigned long long* d_out, int n_cu, int iters)
{
    for (int threads : {512, 256}) {
        run<0, NV>("valu", d_out, n_cu, iters, threads);
        run<1, NV>("lds", d_out, n_cu, iters, threads);
        run<2, NV>("burst", d_out, n_cu, iters, threads);
        run<3, NV>("spread", d_out, n_cu, iters, threads);
        run<4, NV>("wfirst", d_out, n_cu, iters, threads);
        run<5, NV>("w-spread", d_out, n_cu, iters, threads);
        run<6, NV>("r-spread", d_out, n_cu, iters, threads);
    }
}

int main(int argc, char** argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    unsigned long long* d_out;
    CK(hipMalloc((void**)&d_out, (size_t)n_cu * 16 * 8));
    printf("%d CUs; per iteration and wave: 8 ds_write_b128 + 8 ds_read_b128 + NV v_fma_f64\n", n_cu);
    sweep<64>(d_out, n_cu, iters);
    sweep<128>(d_out, n_cu, iters);
    return 0;
}
