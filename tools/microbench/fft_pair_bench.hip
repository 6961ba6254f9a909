// fft_pair_bench.hip — what a transform pair costs on one CU, in isolation (r04).
//
// The blind rotation spends ~60 % of a step in `fft512_pair1` (three pairs per CMUX step), at about half the rate its
// own VALU work would allow.  This microbenchmark runs the SAME device code (spf_device.hpp) in a loop, 8 waves per
// 512-thread workgroup, one workgroup per CU, with the register pressure of the real kernel emulated by a live
// accumulator / product set, and reports cycles per pair per wave for a list of variants:
//   0  fft512_pair1<+1, 2>            the shipped schedule (exchange 2 of the second transform in registers)
//   1  fft512_pair1<+1, 0>            every exchange through LDS
//   2  fft512_pair1<+1, 1>            exchange 2 of both transforms in registers
//   3  arithmetic only                the butterflies and twiddles, no exchange at all (wrong results: VALU floor)
//   4  exchanges only                 the LDS traffic of variant 0, no butterflies (LDS floor)
//   5  fft512_pair1s<+1, 2>           variant 0 with the stores of one transform spread through the other's butterflies
// and the same with a workgroup barrier after every pair (SYNC = 1: the lockstep the ring imposes on the real kernel).
// Build:  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -o tools/microbench/bin/fft_pair_bench tools/microbench/fft_pair_bench.hip
// Run:    tools/microbench/bin/fft_pair_bench [iters]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#ifdef SPF_DEVICE_HEADER // (e.g. -DSPF_DEVICE_HEADER='"../../profiles/r06_experimental_sources/dag2/spf_device_dag2.hpp"': the DAG-II transform)
#include SPF_DEVICE_HEADER
#else
#include "../../spf_amd/csrc/spf_device.hpp"
#endif

using namespace spf;

#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(e_)); exit(1); } } while (0)

// the butterflies and twiddles of a pair without any exchange (registers keep stale data between passes)
template <int DIR>
__device__ __forceinline__ void pair_arith_only(c64 (&A)[8], c64 (&B)[8], const c64* tab, int lane)
{
    const int hi3 = lane >> 3;
    radix8<DIR>(A);
#pragma unroll
    for (int k1 = 1; k1 < 8; k1++) A[k1] = cmul_tw<DIR>(A[k1], tab[kT1Off + (k1 - 1) * 64 + lane]);
    sched_fence();
    radix8<DIR>(B);
#pragma unroll
    for (int k1 = 1; k1 < 8; k1++) B[k1] = cmul_tw<DIR>(B[k1], tab[kT1Off + (k1 - 1) * 64 + lane]);
    sched_fence();
    radix8<DIR>(A);
#pragma unroll
    for (int c = 1; c < 8; c++) A[c] = cmul_tw<DIR>(A[c], tab[kT2Off + (c - 1) * 8 + hi3]);
    sched_fence();
    radix8<DIR>(B);
#pragma unroll
    for (int c = 1; c < 8; c++) B[c] = cmul_tw<DIR>(B[c], tab[kT2Off + (c - 1) * 8 + hi3]);
    sched_fence();
    radix8<DIR>(A);
    radix8<DIR>(B);
    sched_fence();
}

// r06, the gate of "DAG-II" (VERDICT r05 task 3 ii): the FMA-folded radix-8 — the previous pass's twiddles applied at the INPUT of
// the first butterfly stage (t = w a; s = t + w' b as two FMAs per component; d = 2 t - s as one), and the W8 rotations' 1/sqrt(2)
// folded into the last stage's additions (v = b0 +- c q as FMAs).  Timing only (the twiddle each lane would need after the
// exchange is taken from the same table rows): 72 f64 instructions per twiddled radix-8 instead of 84, 52 instead of 56 untwiddled.
template <int DIR> __device__ __forceinline__ void radix8_folded(c64 (&v)[8], const c64* w /* 7 input twiddles, or null */)
{
    c64 s[4], t[4];
    if (w) {
        // pair (0, 4): only v4 carries a twiddle
        s[0].re = __builtin_fma(-v[4].im, w[3].im, __builtin_fma(v[4].re, w[3].re, v[0].re));
        s[0].im = __builtin_fma(v[4].im, w[3].re, __builtin_fma(v[4].re, w[3].im, v[0].im));
        t[0].re = __builtin_fma(2.0, v[0].re, -s[0].re);
        t[0].im = __builtin_fma(2.0, v[0].im, -s[0].im);
#pragma unroll
        for (int j = 1; j < 4; j++) {
            const c64 a = cmul_tw<DIR>(v[j], w[j - 1]);
            s[j].re = __builtin_fma(-v[j + 4].im, w[j + 3].im, __builtin_fma(v[j + 4].re, w[j + 3].re, a.re));
            s[j].im = __builtin_fma(v[j + 4].im, w[j + 3].re, __builtin_fma(v[j + 4].re, w[j + 3].im, a.im));
            t[j].re = __builtin_fma(2.0, a.re, -s[j].re);
            t[j].im = __builtin_fma(2.0, a.im, -s[j].im);
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; j++) { s[j] = cadd(v[j], v[j + 4]); t[j] = csub(v[j], v[j + 4]); }
    }
    // the W8 rotations without their 1/sqrt(2): q = the unscaled sum / difference of the two rotated terms
    const double p1 = t[1].re + t[1].im, m1 = t[1].im - t[1].re, p3 = t[3].re + t[3].im, m3 = t[3].im - t[3].re;
    const c64 qb = {p1 + m3, m1 - p3}, qe = {p1 - m3, m1 + p3};
    c64 a0 = cadd(s[0], s[2]), a1 = cadd(s[1], s[3]), a2 = csub(s[0], s[2]), d = csub(s[1], s[3]);
    v[0] = cadd(a0, a1);
    v[4] = csub(a0, a1);
    v[2] = {a2.re + d.im, a2.im - d.re};
    v[6] = {a2.re - d.im, a2.im + d.re};
    const c64 b0 = {t[0].re + t[2].im, t[0].im - t[2].re}, b2 = {t[0].re - t[2].im, t[0].im + t[2].re};
    v[1] = {__builtin_fma(kSqrtHalf, qb.re, b0.re), __builtin_fma(kSqrtHalf, qb.im, b0.im)};
    v[5] = {__builtin_fma(-kSqrtHalf, qb.re, b0.re), __builtin_fma(-kSqrtHalf, qb.im, b0.im)};
    v[3] = {__builtin_fma(kSqrtHalf, qe.im, b2.re), __builtin_fma(-kSqrtHalf, qe.re, b2.im)};
    v[7] = {__builtin_fma(-kSqrtHalf, qe.im, b2.re), __builtin_fma(kSqrtHalf, qe.re, b2.im)};
}

template <int DIR>
__device__ __forceinline__ void pair_arith_only_folded(c64 (&A)[8], c64 (&B)[8], const c64* tab, int lane)
{
    const int hi3 = lane >> 3;
    c64 w[7];
    radix8_folded<DIR>(A, nullptr);
    sched_fence();
    radix8_folded<DIR>(B, nullptr);
    sched_fence();
#pragma unroll
    for (int k = 0; k < 7; k++) w[k] = tab[kT1Off + k * 64 + lane];
    radix8_folded<DIR>(A, w);
    sched_fence();
    radix8_folded<DIR>(B, w);
    sched_fence();
#pragma unroll
    for (int k = 0; k < 7; k++) w[k] = tab[kT2Off + k * 8 + hi3];
    radix8_folded<DIR>(A, w);
    radix8_folded<DIR>(B, w);
    sched_fence();
}

// the LDS traffic of fft512_pair1<., 2> without the butterflies
__device__ __forceinline__ void pair_lds_only(c64 (&A)[8], c64 (&B)[8], char* buf, const c64* tab, int lane)
{
    const int hi3 = lane >> 3, lo3 = lane & 7;
    const uint32_t rd1 = 16 * (8 * lo3 + (hi3 ^ lo3));
    const uint32_t rd2 = 16 * (8 * hi3 + (hi3 ^ lo3));
    const uint32_t wbase = 16 * (64 * hi3 + lo3);
    char* wr[8];
#pragma unroll
    for (int r = 0; r < 8; r++) wr[r] = buf + ((wbase ^ (16 * r)) + 128 * r);
    c64 tw[7];
#pragma unroll
    for (int k = 0; k < 7; k++) tw[k] = tab[kT1Off + k * 64 + lane];
#pragma unroll
    for (int k1 = 0; k1 < 8; k1++) *reinterpret_cast<c64*>(wr[k1]) = A[k1];
    sched_fence();
#pragma unroll
    for (int a = 0; a < 8; a++) A[a] = *reinterpret_cast<const c64*>(buf + 1024 * a + rd1);
    sched_fence();
#pragma unroll
    for (int k1 = 0; k1 < 8; k1++) *reinterpret_cast<c64*>(wr[k1]) = B[k1];
    sched_fence();
#pragma unroll
    for (int k = 0; k < 7; k++) tw[k] = cadd(tw[k], tab[kT2Off + k * 8 + hi3]);
#pragma unroll
    for (int a = 0; a < 8; a++) B[a] = *reinterpret_cast<const c64*>(buf + 1024 * a + rd1);
    sched_fence();
#pragma unroll
    for (int c = 0; c < 8; c++) *reinterpret_cast<c64*>(wr[c]) = A[c];
    sched_fence();
#pragma unroll
    for (int b = 0; b < 8; b++) A[b] = *reinterpret_cast<const c64*>(buf + 1024 * b + rd2);
    sched_fence();
    A[0] = cadd(A[0], tw[0]); A[1] = cadd(A[1], tw[1]); A[2] = cadd(A[2], tw[2]); A[3] = cadd(A[3], tw[3]);
    A[4] = cadd(A[4], tw[4]); A[5] = cadd(A[5], tw[5]); A[6] = cadd(A[6], tw[6]);
}

// variant 5 is spf_device.hpp's fft512_pair1s: the stores of one transform spread through the other's butterflies

// STG: waves 4-7 (the SIMD partners of waves 0-3) sleep STG x 64 cycles at the start of every pair (a small phase offset, so
// that one wave's exchanges fall under the other's butterflies); PRI: 1 = waves 4-7 at s_setprio 1 for the whole pair,
// 2 = for the first half of the pair (the r03c schedule of the real kernel), 3 = waves 0-3 at priority 1 for the second half
// PRESS: registers of emulated live state besides the pair: 2 = accumulator + product (128), 1 = accumulator only (64)
template <int V, int SYNC, int STG = 0, int PRI = 0, int PRESS = 2>
__global__ __launch_bounds__(512, 2) void pair_loop(const c64* tables, unsigned long long* out, int iters)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    c64* tab = reinterpret_cast<c64*>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    char* buf = smem + kTableBytes + wv * 8192;
    for (int i = tid; i < kTableEntries; i += 512) tab[i] = tables[i];
    c64 A[8], B[8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        A[r] = {(double)((lane * 8 + r) * 2654435761u % 65536) - 32768.0, (double)((lane * 8 + r + 7) * 40503u % 65536) - 32768.0};
        B[r] = {(double)((lane * 8 + r + wv) * 2246822519u % 65536) - 32768.0, (double)((lane * 8 + r + 3) * 3266489917u % 65536) - 32768.0};
    }
    // live state of the real kernel: 64 registers of accumulator, 64 of frequency-domain product
    unsigned long long acc[32];
    c64 prod[16];
#pragma unroll
    for (int e = 0; e < 32; e++) acc[e] = (unsigned long long)(tid * 32 + e);
#pragma unroll
    for (int e = 0; e < 16; e++) prod[e] = {1.0 + e, 2.0 + lane};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const bool young = wv >= 4;
    for (int it = 0; it < iters; it++) {
        if constexpr (STG > 0) { if (young) __builtin_amdgcn_s_sleep(STG); }
        if constexpr (PRI == 1 || PRI == 2) { if (young) __builtin_amdgcn_s_setprio(1); }
        if constexpr (V == 0 && PRI >= 2) {
            fft512_pair1<+1, 2>(A, B, buf, tab, lane, [&]() {
                if constexpr (PRI == 2) { if (young) __builtin_amdgcn_s_setprio(0); }
                if constexpr (PRI == 3) { if (young) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(1); }
            });
            if constexpr (PRI == 3) { if (!young) __builtin_amdgcn_s_setprio(0); }
        } else
        if constexpr (V == 0) fft512_pair1<+1, 2>(A, B, buf, tab, lane);
        else if constexpr (V == 1) fft512_pair1<+1, 0>(A, B, buf, tab, lane);
        else if constexpr (V == 2) fft512_pair1<+1, 1>(A, B, buf, tab, lane);
        else if constexpr (V == 3) pair_arith_only<+1>(A, B, tab, lane);
        else if constexpr (V == 4) pair_lds_only(A, B, buf, tab, lane);
        else if constexpr (V == 13) pair_arith_only_folded<+1>(A, B, tab, lane);
        else if constexpr (V == 8) fft512_pair1t<+1, 2>(A, B, buf, tab, lane);
        else if constexpr (V == 9) fft512_pair1t<+1, 1>(A, B, buf, tab, lane);
        else if constexpr (V == 11) fft512_pair1ts<+1, 2>(A, B, buf, tab, lane);
        else fft512_pair1ts2<+1, 2>(A, B, buf, tab, lane);
        // keep magnitudes bounded (exact power-of-two scaling) and the emulated state live
#pragma unroll
        for (int r = 0; r < 8; r++) {
            A[r] = {A[r].re * 0x1p-9, A[r].im * 0x1p-9};
            B[r] = {B[r].re * 0x1p-9, B[r].im * 0x1p-9};
        }
        if ((it & 15) == 15) {
#pragma unroll
            for (int e = 0; e < 16; e++) {
                acc[e] += (unsigned long long)__double_as_longlong(A[e & 7].re);
                acc[16 + e] ^= (unsigned long long)__double_as_longlong(B[e & 7].im);
                if constexpr (PRESS >= 2) {
                    prod[e].re = __builtin_fma(prod[e].re, 0.5, A[e & 7].im);
                    prod[e].im = __builtin_fma(prod[e].im, 0.5, B[e & 7].re);
                }
            }
        }
        if constexpr (PRI == 1) { if (young) __builtin_amdgcn_s_setprio(0); }
        if constexpr (SYNC) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long h = 0;
#pragma unroll
    for (int e = 0; e < 32; e++) h = h * 1099511628211ull + acc[e];
#pragma unroll
    for (int e = 0; e < 16; e++) h = h * 1099511628211ull + (unsigned long long)__double_as_longlong(prod[e].re) + (unsigned long long)__double_as_longlong(prod[e].im);
#pragma unroll
    for (int r = 0; r < 8; r++) h = h * 1099511628211ull + (unsigned long long)__double_as_longlong(A[r].re) + (unsigned long long)__double_as_longlong(B[r].im);
    // wave-level xor of the lanes' hashes
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) h ^= __shfl_xor(h, o);
    if (lane == 0) {
        out[((size_t)blockIdx.x * 8 + wv) * 2] = t1 - t0;
        out[((size_t)blockIdx.x * 8 + wv) * 2 + 1] = h;
    }
}

static c64 root(unsigned long long num, unsigned long long den)
{
    const long double TWO_PI = 6.283185307179586476925286766559005768L;
    long double th = TWO_PI * (long double)(num % den) / (long double)den;
    return {(double)cosl(th), (double)sinl(th)};
}

template <int V, int SYNC, int STG = 0, int PRI = 0, int PRESS = 2> static void run(const char* name, const c64* d_tab, unsigned long long* d_out, int n_cu, int iters)
{
    const int lds = kTableBytes + 8 * 8192;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&pair_loop<V, SYNC, STG, PRI, PRESS>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL((pair_loop<V, SYNC, STG, PRI, PRESS>), dim3(n_cu), dim3(512), lds, 0, d_tab, d_out, 8); // warm-up
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((pair_loop<V, SYNC, STG, PRI, PRESS>), dim3(n_cu), dim3(512), lds, 0, d_tab, d_out, iters);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h((size_t)n_cu * 16);
    CK(hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> old_w, young_w;
    unsigned long long sum = 0;
    for (int b = 0; b < n_cu; b++)
        for (int w = 0; w < 8; w++) {
            (w < 4 ? old_w : young_w).push_back((double)h[((size_t)b * 8 + w) * 2] / iters);
            sum += h[((size_t)b * 8 + w) * 2 + 1];
        }
    std::sort(old_w.begin(), old_w.end());
    std::sort(young_w.begin(), young_w.end());
    if (STG || PRI) printf("[stagger %d x 64 cycles, priority mode %d] ", STG, PRI);
    if (PRESS != 2) printf("[live state %d] ", PRESS);
    printf("%-58s sync=%d  cycles/pair: waves 0-3 %7.0f  waves 4-7 %7.0f   %.3f ms  (%.0f cycles/pair at 2.4 GHz)  checksum %016llx\n", name, SYNC,
           old_w[old_w.size() / 2], young_w[young_w.size() / 2], ms, ms * 1e-3 * 2.4e9 / iters, sum);
}

int main(int argc, char** argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 4000;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    std::vector<c64> t(kTableEntries);
    for (int k1 = 1; k1 < 8; k1++)
        for (int lane = 0; lane < 64; lane++) { c64 r = root((unsigned long long)(lane * k1), 512); t[kT1Off + (k1 - 1) * 64 + lane] = {r.re, -r.im}; }
    for (int c = 1; c < 8; c++)
        for (int b = 0; b < 8; b++) { c64 r = root((unsigned long long)(b * c), 64); t[kT2Off + (c - 1) * 8 + b] = {r.re, -r.im}; }
    for (int k = 0; k < 512; k++) { c64 r = root((unsigned long long)k, 1024); t[kWCOff + k] = {r.re, -r.im}; }
    for (int n = 0; n < 1024; n++) t[kTWOff + n] = root((unsigned long long)n, 4096);
    c64* d_tab; unsigned long long* d_out;
    CK(hipMalloc((void**)&d_tab, kTableBytes));
    CK(hipMemcpy(d_tab, t.data(), kTableBytes, hipMemcpyHostToDevice));
    CK(hipMalloc((void**)&d_out, (size_t)n_cu * 16 * 8));
    printf("%d CUs, %d pairs per wave, 8 waves per workgroup, one workgroup per CU\n", n_cu, iters);
    run<0, 0>("0 fft512_pair1<+1,2> (shipped)", d_tab, d_out, n_cu, iters);
    run<0, 1>("0 fft512_pair1<+1,2> (shipped)", d_tab, d_out, n_cu, iters);
    run<1, 0>("1 fft512_pair1<+1,0> all exchanges via LDS", d_tab, d_out, n_cu, iters);
    run<1, 1>("1 fft512_pair1<+1,0> all exchanges via LDS", d_tab, d_out, n_cu, iters);
    run<2, 0>("2 fft512_pair1<+1,1> exchange 2 in registers (both)", d_tab, d_out, n_cu, iters);
    run<2, 1>("2 fft512_pair1<+1,1> exchange 2 in registers (both)", d_tab, d_out, n_cu, iters);
    run<3, 0>("3 arithmetic only, DAG-I shape (output products) on the new radix-8", d_tab, d_out, n_cu, iters);
    run<3, 1>("3 arithmetic only, DAG-I shape (output products) on the new radix-8", d_tab, d_out, n_cu, iters);
    run<13, 0>("13 arithmetic only, FMA-folded radix-8 (DAG-II gate)", d_tab, d_out, n_cu, iters);
    run<13, 1>("13 arithmetic only, FMA-folded radix-8 (DAG-II gate)", d_tab, d_out, n_cu, iters);
    run<4, 0>("4 LDS traffic only (LDS floor)", d_tab, d_out, n_cu, iters);
    run<4, 1>("4 LDS traffic only (LDS floor)", d_tab, d_out, n_cu, iters);
    run<8, 1>("8 fft512_pair1t<+1,2>: factors requested early, shared", d_tab, d_out, n_cu, iters);
    run<11, 1>("11 fft512_pair1ts: shared twiddles + spread stores", d_tab, d_out, n_cu, iters);
    run<12, 1>("12 fft512_pair1ts2: + reads under the twiddle products", d_tab, d_out, n_cu, iters);
    run<11, 1, 0, 0, 1>("11 shared twiddles + spread stores", d_tab, d_out, n_cu, iters);
    run<12, 1, 0, 0, 1>("12 + reads under the twiddle products", d_tab, d_out, n_cu, iters);
    return 0;
}
