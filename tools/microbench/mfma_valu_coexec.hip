// mfma_valu_coexec — do the f64 matrix pipe and the f64 vector pipe of one SIMD run at the same time?
//
// profiles/r02_mfma_f64_probe.md measured ONE wave: v_fma_f64 4.72 cycles (64 FMA), v_mfma_f64_4x4x4_4b 16.4 (256 FMA),
// v_mfma_f64_16x16x4 64.1 (1024 FMA) — the same FMA rate on either pipe.  Open question (VERDICT r04 task 4 iii): with
// one wave issuing MFMA back to back and a SIBLING wave on the same SIMD issuing v_fma_f64, is the aggregate more than
// either alone (the guide draws the matrix and vector pipes as separate units)?  If f64 MFMA is executed BY the vector
// ALUs (same FMA rate suggests it), there is nothing to gain by moving the external product's MAD to it.
//
// One workgroup of 8 waves on one CU: waves w and w + 4 share SIMD w % 4 (checked with HW_REG_HW_ID).  Wave 0 and wave 4
// run the programmed instruction streams (8 independent accumulators each, 4096 instructions), the other waves exit.
// Every stream is timed with s_memtime inside the wave; both waves start behind the same barrier.
//
// Build: hipcc --offload-arch=gfx950 -O2 -ffp-contract=off tools/microbench/mfma_valu_coexec.hip -o tools/microbench/bin/mfma_valu_coexec
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef double d4 __attribute__((ext_vector_type(4)));

enum Stream { S_NONE = 0, S_FMA = 1, S_MFMA4 = 2, S_MFMA16 = 3, S_MUL_ADD = 4 };

template <int KIND>
__device__ __forceinline__ double run_stream(double x, double y, uint64_t* cyc)
{
    constexpr int kIter = 512;
    double acc[8];
    d4 acc4[8];
    for (int i = 0; i < 8; i++) { acc[i] = (double)i; acc4[i] = d4{(double)i, 1.0, 2.0, 3.0}; }
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    if (KIND == S_FMA) {
        for (int it = 0; it < kIter; it++)
#pragma unroll
            for (int i = 0; i < 8; i++) acc[i] = __builtin_fma(x, y, acc[i]);
    } else if (KIND == S_MFMA4) {
        for (int it = 0; it < kIter; it++)
#pragma unroll
            for (int i = 0; i < 8; i++) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, acc[i], 0, 0, 0);
    } else if (KIND == S_MFMA16) {
        for (int it = 0; it < kIter; it++)
#pragma unroll
            for (int i = 0; i < 8; i++) acc4[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc4[i], 0, 0, 0);
    } else if (KIND == S_MUL_ADD) { // v_mul_f64 / v_add_f64 alternating (the butterflies' mix), 8 chains
        for (int it = 0; it < kIter; it++)
#pragma unroll
            for (int i = 0; i < 8; i++) acc[i] = (i & 1) ? acc[i] * x : acc[i] + y;
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    *cyc = t1 - t0;
    double sum = 0;
    for (int i = 0; i < 8; i++) sum += acc[i] + acc4[i][0] + acc4[i][1] + acc4[i][2] + acc4[i][3];
    return sum;
}

template <int KA, int KB>
__global__ __launch_bounds__(512) void coexec(double* out, uint64_t* cyc, uint32_t* hwid, double x, double y)
{
    const int wave = threadIdx.x / 64;
    if (threadIdx.x % 64 == 0) hwid[wave] = __builtin_amdgcn_s_getreg((31 << 11) | 4); // HW_REG_HW_ID
    __syncthreads();
    double r = 0;
    uint64_t c = 0;
    if (wave == 0) r = run_stream<KA>(x, y, &c);
    else if (wave == 4) r = run_stream<KB>(x, y, &c);
    else return;
    out[threadIdx.x] = r;
    if (threadIdx.x % 64 == 0) cyc[wave] = c;
}

template <int KA, int KB>
static int measure(const char* name, double fma_a, double fma_b)
{
    double* o; uint64_t* cy; uint32_t* hw;
    CK(hipMalloc(&o, 512 * 8)); CK(hipMalloc(&cy, 8 * 8)); CK(hipMalloc(&hw, 8 * 4));
    CK(hipMemset(cy, 0, 64));
    for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL((coexec<KA, KB>), 1, 512, 0, 0, o, cy, hw, 1.0000001, 0.9999999);
    CK(hipDeviceSynchronize());
    uint64_t h[8]; uint32_t id[8];
    CK(hipMemcpy(h, cy, 64, hipMemcpyDeviceToHost));
    CK(hipMemcpy(id, hw, 32, hipMemcpyDeviceToHost));
    const int simd0 = (id[0] >> 4) & 3, simd4 = (id[4] >> 4) & 3;
    // s_memtime ticks (4.46 per v_fma_f64 of one wave: shader cycles on this part); ratios between rows are what matters
    const double ta = (double)h[0], tb = (double)h[4];
    const double wall = ta > tb ? ta : tb;
    // aggregate = all FMAs of both streams over the time until BOTH are done (the two start behind one barrier); a per-wave
    // rate summed over unequal durations would overstate it
    const double agg = ((KA ? fma_a * 4096 : 0) + (KB ? fma_b * 4096 : 0)) / wall;
    printf("%-34s wave0 %8.0f cyc  wave4 %8.0f cyc  (SIMD %d / %d)  FMA/cyc: wave0 alone-rate %6.1f  wave4 %6.1f  AGGREGATE over wall %6.2f\n", name, ta, tb,
           simd0, simd4, KA ? fma_a * 4096 / ta : 0.0, KB ? fma_b * 4096 / tb : 0.0, agg);
    (void)hipFree(o); (void)hipFree(cy); (void)hipFree(hw);
    return 0;
}

int main()
{
    printf("two waves of one SIMD (wave 0 and wave 4 of a 512-thread workgroup), 4096 instructions each, 8 independent accumulators\n");
    if (measure<S_FMA, S_NONE>("v_fma_f64 alone", 64, 0)) return 1;
    if (measure<S_MFMA4, S_NONE>("v_mfma_f64_4x4x4_4b alone", 256, 0)) return 1;
    if (measure<S_MFMA16, S_NONE>("v_mfma_f64_16x16x4 alone", 1024, 0)) return 1;
    if (measure<S_MUL_ADD, S_NONE>("v_mul_f64 / v_add_f64 alone", 64, 0)) return 1;
    if (measure<S_FMA, S_FMA>("v_fma_f64 + v_fma_f64", 64, 64)) return 1;
    if (measure<S_MFMA4, S_MFMA4>("mfma 4x4x4 + mfma 4x4x4", 256, 256)) return 1;
    if (measure<S_MFMA16, S_MFMA16>("mfma 16x16x4 + mfma 16x16x4", 1024, 1024)) return 1;
    if (measure<S_MFMA4, S_FMA>("mfma 4x4x4 + v_fma_f64", 256, 64)) return 1;
    if (measure<S_MFMA16, S_FMA>("mfma 16x16x4 + v_fma_f64", 1024, 64)) return 1;
    if (measure<S_MFMA16, S_MUL_ADD>("mfma 16x16x4 + v_mul/v_add_f64", 1024, 64)) return 1;
    if (measure<S_MFMA4, S_MUL_ADD>("mfma 4x4x4 + v_mul/v_add_f64", 256, 64)) return 1;
    return 0;
}
