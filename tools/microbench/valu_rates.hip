// valu_rates — issue cost of the vector instructions the blind rotation spends its non-FMA time on (gfx950).
//
// The headline kernel is VALU-issue bound (busy 0.72) and only 412 of its 3 176 vector instructions per wave and step are
// v_fma_f64; the rest are v_add_f64 / v_mul_f64 (1 700) and 32- / 64-bit integer, conversion and lane-move instructions
// (960).  profiles/r05_mfma_valu_coexec.md saw the add / multiply mix at 5.3 cycles against 4.46 for v_fma_f64 and left it
// there.  This bench times every instruction kind ALONE: one wave (and two waves of the same SIMD), 16 independent chains,
// 256 x 16 = 4096 instructions, s_memtime inside the wave.  The question behind it: which of the substitutions that leave the
// words unchanged (an add as fma(a, 1.0, b), a 64-bit add as v_lshl_add_u64, f64 fract / convert instead of integer
// shifts) are cheaper on this part.
//
// Build: hipcc --offload-arch=gfx950 -O2 tools/microbench/valu_rates.hip -o tools/microbench/bin/valu_rates
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// sixteen chains c0..c15 (64-bit each: a VGPR pair), two loop-invariant VGPR operands x, y (pairs)
#define CH16(M) M(c0) M(c1) M(c2) M(c3) M(c4) M(c5) M(c6) M(c7) M(c8) M(c9) M(c10) M(c11) M(c12) M(c13) M(c14) M(c15)
// the same chains as 32-bit values d0..d15 (and e0..e15 as a second dword), 32-bit operands x32, y32
#define CH16D(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)

#define I_FMA(c)      asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(c) : "v"(x), "v"(y));
#define I_ADD(c)      asm volatile("v_add_f64 %0, %0, %1" : "+v"(c) : "v"(y));
#define I_MUL(c)      asm volatile("v_mul_f64 %0, %0, %1" : "+v"(c) : "v"(x));
#define I_ADDFMA(c)   asm volatile("v_fma_f64 %0, %0, 1.0, %1" : "+v"(c) : "v"(y));
#define I_MULFMA(c)   asm volatile("v_fma_f64 %0, %0, %1, 0" : "+v"(c) : "v"(x));
#define I_ADD3(c)     asm volatile("v_add_f64 %0, %1, %2" : "=v"(c) : "v"(x), "v"(y));   /* no dependence at all */
#define I_ADDU32(c) asm volatile("v_add_u32 %0, %0, %1" : "+v"(d##c) : "v"(y32));
#define I_XOR(c) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(d##c) : "v"(y32));
#define I_SHL64(c)    asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(c) : "v"(sh));
#define I_ASHR64(c)   asm volatile("v_ashrrev_i64 %0, %1, %0" : "+v"(c) : "v"(sh));
#define I_ADD64(c)    asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(c) : "v"(y));
#define I_ADDCO(c) asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(d##c), "+v"(e##c) : "v"(x32), "v"(y32) : "vcc");
#define I_CMP64(c)    asm volatile("v_cmp_gt_i64 vcc, %0, %1" : : "v"(c), "v"(y) : "vcc");
#define I_CMP32(c) asm volatile("v_cmp_eq_u32 vcc, %0, %1" : : "v"(d##c), "v"(y32) : "vcc");
#define I_BFE(c) asm volatile("v_bfe_u32 %0, %0, 20, 11" : "+v"(d##c));
#define I_MIN3(c) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(d##c) : "v"(x32), "v"(y32));
#define I_CVTF64I32(c) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(c) : "v"(sh));
#define I_CVTU32F64(c) asm volatile("v_cvt_u32_f64 %0, %1" : "=v"(d##c) : "v"(x));
#define I_FRACT(c)    asm volatile("v_fract_f64 %0, %0" : "+v"(c));
#define I_LDEXP(c)    asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(c) : "v"(sh));
#define I_TRUNC(c)    asm volatile("v_trunc_f64 %0, %0" : "+v"(c));
#define I_DPP(c) asm volatile("v_mov_b32_dpp %0, %0 row_ror:8 row_mask:0xf bank_mask:0xc" : "+v"(d##c));
#define I_PERM32(c) asm volatile("v_permlane32_swap_b32_e32 %0, %1" : "+v"(d##c), "+v"(e##c));
#define I_CNDMASK(c) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(d##c) : "v"(y32) : "vcc");
#define I_MOV(c) asm volatile("v_mov_b32 %0, %1" : "=v"(d##c) : "v"(y32));
#define I_ASHR32(c) asm volatile("v_ashrrev_i32 %0, 31, %0" : "+v"(d##c));
#define I_LSHLADD(c) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(d##c) : "v"(y32));
#define I_ANDOR(c) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(d##c) : "v"(x32), "v"(y32));
#define I_PKADD(c)    asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(c) : "v"(y));
#define I_MAD64(c) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(c0) : "v"(d##c), "v"(y32) : "vcc");

enum Kind { K_NONE, K_FMA, K_ADD, K_MUL, K_ADDMUL, K_ADDFMA, K_MULFMA, K_ADD3, K_ADDU32, K_XOR, K_SHL64, K_ASHR64, K_ADD64, K_ADDCO, K_CMP64,
            K_CMP32, K_BFE, K_MIN3, K_CVTF64I32, K_CVTU32F64, K_FRACT, K_LDEXP, K_TRUNC, K_DPP, K_PERM32, K_CNDMASK, K_MOV, K_ASHR32,
            K_LSHLADD, K_ANDOR, K_PKADD, K_MAD64, K_FMAADD, K_COUNT };

template <int KIND>
__device__ __forceinline__ uint64_t run(uint64_t seed, uint64_t* cyc)
{
    uint64_t c0 = seed, c1 = seed + 1, c2 = seed + 2, c3 = seed + 3, c4 = seed + 4, c5 = seed + 5, c6 = seed + 6, c7 = seed + 7;
    uint64_t c8 = seed + 8, c9 = seed + 9, c10 = seed + 10, c11 = seed + 11, c12 = seed + 12, c13 = seed + 13, c14 = seed + 14, c15 = seed + 15;
    uint64_t x = 0x3FF0000000000001ull + seed, y = 0x3FEFFFFFFFFFFFFFull - seed;
    uint32_t sh = (uint32_t)(seed & 3) + 1, x32 = (uint32_t)seed * 3u + 1u, y32 = (uint32_t)seed * 5u + 7u;
    uint32_t d0 = x32, d1 = x32 + 1, d2 = x32 + 2, d3 = x32 + 3, d4 = x32 + 4, d5 = x32 + 5, d6 = x32 + 6, d7 = x32 + 7;
    uint32_t d8 = x32 + 8, d9 = x32 + 9, d10 = x32 + 10, d11 = x32 + 11, d12 = x32 + 12, d13 = x32 + 13, d14 = x32 + 14, d15 = x32 + 15;
    uint32_t e0 = y32, e1 = y32 + 1, e2 = y32 + 2, e3 = y32 + 3, e4 = y32 + 4, e5 = y32 + 5, e6 = y32 + 6, e7 = y32 + 7;
    uint32_t e8 = y32 + 8, e9 = y32 + 9, e10 = y32 + 10, e11 = y32 + 11, e12 = y32 + 12, e13 = y32 + 13, e14 = y32 + 14, e15 = y32 + 15;
    asm volatile("" : "+v"(x), "+v"(y), "+v"(sh), "+v"(x32), "+v"(y32));
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 256; it++) {
        if constexpr (KIND == K_FMA) { CH16(I_FMA) }
        else if constexpr (KIND == K_ADD) { CH16(I_ADD) }
        else if constexpr (KIND == K_MUL) { CH16(I_MUL) }
        else if constexpr (KIND == K_ADDMUL) { I_ADD(c0) I_MUL(c1) I_ADD(c2) I_MUL(c3) I_ADD(c4) I_MUL(c5) I_ADD(c6) I_MUL(c7) I_ADD(c8) I_MUL(c9) I_ADD(c10) I_MUL(c11) I_ADD(c12) I_MUL(c13) I_ADD(c14) I_MUL(c15) }
        else if constexpr (KIND == K_FMAADD) { I_ADD(c0) I_FMA(c1) I_ADD(c2) I_FMA(c3) I_ADD(c4) I_FMA(c5) I_ADD(c6) I_FMA(c7) I_ADD(c8) I_FMA(c9) I_ADD(c10) I_FMA(c11) I_ADD(c12) I_FMA(c13) I_ADD(c14) I_FMA(c15) }
        else if constexpr (KIND == K_ADDFMA) { CH16(I_ADDFMA) }
        else if constexpr (KIND == K_MULFMA) { CH16(I_MULFMA) }
        else if constexpr (KIND == K_ADD3) { CH16(I_ADD3) }
        else if constexpr (KIND == K_ADDU32) { CH16D(I_ADDU32) }
        else if constexpr (KIND == K_XOR) { CH16D(I_XOR) }
        else if constexpr (KIND == K_SHL64) { CH16(I_SHL64) }
        else if constexpr (KIND == K_ASHR64) { CH16(I_ASHR64) }
        else if constexpr (KIND == K_ADD64) { CH16(I_ADD64) }
        else if constexpr (KIND == K_ADDCO) { CH16D(I_ADDCO) }
        else if constexpr (KIND == K_CMP64) { CH16(I_CMP64) }
        else if constexpr (KIND == K_CMP32) { CH16D(I_CMP32) }
        else if constexpr (KIND == K_BFE) { CH16D(I_BFE) }
        else if constexpr (KIND == K_MIN3) { CH16D(I_MIN3) }
        else if constexpr (KIND == K_CVTF64I32) { CH16(I_CVTF64I32) }
        else if constexpr (KIND == K_CVTU32F64) { CH16D(I_CVTU32F64) }
        else if constexpr (KIND == K_FRACT) { CH16(I_FRACT) }
        else if constexpr (KIND == K_LDEXP) { CH16(I_LDEXP) }
        else if constexpr (KIND == K_TRUNC) { CH16(I_TRUNC) }
        else if constexpr (KIND == K_DPP) { CH16D(I_DPP) }
        else if constexpr (KIND == K_PERM32) { CH16D(I_PERM32) }
        else if constexpr (KIND == K_CNDMASK) { CH16D(I_CNDMASK) }
        else if constexpr (KIND == K_MOV) { CH16D(I_MOV) }
        else if constexpr (KIND == K_ASHR32) { CH16D(I_ASHR32) }
        else if constexpr (KIND == K_LSHLADD) { CH16D(I_LSHLADD) }
        else if constexpr (KIND == K_ANDOR) { CH16D(I_ANDOR) }
        else if constexpr (KIND == K_PKADD) { CH16(I_PKADD) }
        else if constexpr (KIND == K_MAD64) { CH16D(I_MAD64) }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    *cyc = t1 - t0;
    const uint32_t dx = d0 ^ d1 ^ d2 ^ d3 ^ d4 ^ d5 ^ d6 ^ d7 ^ d8 ^ d9 ^ d10 ^ d11 ^ d12 ^ d13 ^ d14 ^ d15;
    const uint32_t ex = e0 ^ e1 ^ e2 ^ e3 ^ e4 ^ e5 ^ e6 ^ e7 ^ e8 ^ e9 ^ e10 ^ e11 ^ e12 ^ e13 ^ e14 ^ e15;
    return c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7 ^ c8 ^ c9 ^ c10 ^ c11 ^ c12 ^ c13 ^ c14 ^ c15 ^ dx ^ ((uint64_t)ex << 32);
}

// waves 0 and 4 of a 512-thread workgroup share a SIMD; TWO = 1 runs the stream on both
template <int KIND, int TWO>
__global__ __launch_bounds__(512) void rates(uint64_t* out, uint64_t* cyc, uint64_t seed)
{
    const int wave = threadIdx.x / 64;
    __syncthreads();
    uint64_t r = 0, c = 0;
    if (wave == 0 || (TWO && wave == 4)) r = run<KIND>(seed + threadIdx.x, &c);
    else return;
    out[threadIdx.x] = r;
    if (threadIdx.x % 64 == 0) cyc[wave] = c;
}

template <int KIND>
static int measure(const char* name, int per_stmt)
{
    uint64_t *o, *cy;
    CK(hipMalloc(&o, 512 * 8)); CK(hipMalloc(&cy, 8 * 8));
    uint64_t h1[8], h2[8];
    CK(hipMemset(cy, 0, 64));
    for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL((rates<KIND, 0>), 1, 512, 0, 0, o, cy, (uint64_t)0);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h1, cy, 64, hipMemcpyDeviceToHost));
    CK(hipMemset(cy, 0, 64));
    for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL((rates<KIND, 1>), 1, 512, 0, 0, o, cy, (uint64_t)0);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h2, cy, 64, hipMemcpyDeviceToHost));
    const double n = 4096.0 * per_stmt;
    const double wall2 = (double)(h2[0] > h2[4] ? h2[0] : h2[4]);
    printf("%-44s one wave %6.2f ticks per instruction   two waves of a SIMD %6.2f per instruction (SIMD time / all instructions)\n", name,
           (double)h1[0] / n, wall2 / (2 * n));
    (void)hipFree(o); (void)hipFree(cy);
    return 0;
}

int main()
{
    printf("s_memtime ticks per wave64 instruction, 16 independent chains, 4096 statements\n");
#define M(K, name, n) if (measure<K>(name, n)) return 1;
    M(K_NONE, "(empty loop: overhead, per 4096)", 1)
    M(K_FMA, "v_fma_f64 c, x, y, c", 1)
    M(K_ADD, "v_add_f64 c, c, y", 1)
    M(K_MUL, "v_mul_f64 c, c, x", 1)
    M(K_ADDMUL, "v_add_f64 / v_mul_f64 alternating", 1)
    M(K_FMAADD, "v_add_f64 / v_fma_f64 alternating", 1)
    M(K_ADDFMA, "v_fma_f64 c, c, 1.0, y   (an add)", 1)
    M(K_MULFMA, "v_fma_f64 c, c, x, 0     (a multiply)", 1)
    M(K_ADD3, "v_add_f64 c, x, y        (independent)", 1)
    M(K_ADDU32, "v_add_u32", 1)
    M(K_XOR, "v_xor_b32", 1)
    M(K_MOV, "v_mov_b32", 1)
    M(K_ASHR32, "v_ashrrev_i32", 1)
    M(K_BFE, "v_bfe_u32", 1)
    M(K_MIN3, "v_min3_u32", 1)
    M(K_LSHLADD, "v_lshl_add_u32", 1)
    M(K_ANDOR, "v_and_or_b32", 1)
    M(K_CNDMASK, "v_cndmask_b32", 1)
    M(K_CMP32, "v_cmp_eq_u32 -> vcc", 1)
    M(K_CMP64, "v_cmp_gt_i64 -> vcc", 1)
    M(K_SHL64, "v_lshlrev_b64", 1)
    M(K_ASHR64, "v_ashrrev_i64", 1)
    M(K_ADD64, "v_lshl_add_u64 c, c, 0, y (64-bit add)", 1)
    M(K_ADDCO, "v_add_co_u32 + v_addc_co_u32 (per pair)", 1)
    M(K_MAD64, "v_mad_u64_u32", 1)
    M(K_CVTF64I32, "v_cvt_f64_i32", 1)
    M(K_CVTU32F64, "v_cvt_u32_f64", 1)
    M(K_FRACT, "v_fract_f64", 1)
    M(K_LDEXP, "v_ldexp_f64", 1)
    M(K_TRUNC, "v_trunc_f64", 1)
    M(K_DPP, "v_mov_b32_dpp row_ror:8", 1)
    M(K_PERM32, "v_permlane32_swap", 1)
    M(K_PKADD, "v_pk_add_f32", 1)
    return 0;
}
