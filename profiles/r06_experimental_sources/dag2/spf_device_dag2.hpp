// spf_device.hpp — device-side arithmetic of the gfx950 bootstrap engine.
//
// Everything here is wave64 / CDNA4 code: a 512-point complex FFT lives in ONE wavefront
// (64 lanes x 8 complex f64 registers), exchanges go through a per-wave LDS tile with an
// XOR-swizzled, bank-conflict-free image, and no workgroup barrier is ever needed.
//
// The arithmetic follows the canonical operation order "DAG-I" (DESIGN.md §FFT): every add,
// multiply and fma below is performed on the same operands in the same association as the
// build's definition, so results are reproducible bit for bit.  Compile with
// -ffp-contract=off: the only fused operations are the explicit __builtin_fma calls.
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace spf {

struct c64 {
    double re, im;
};

// Pointers fetched from a table in memory (per-gate operand tables) are generic to the compiler: it emits
// flat_load, which counts on lgkmcnt as well as vmcnt, so every LDS wait in the transforms also waits for the
// key rows in flight.  These views pin the address space to global (global_load/global_store: vmcnt only).
typedef double f64x2_t __attribute__((ext_vector_type(2)));
typedef const f64x2_t __attribute__((address_space(1)))* gc64_ptr;
typedef const uint64_t __attribute__((address_space(1)))* gu64_cptr;
typedef uint64_t __attribute__((address_space(1)))* gu64_ptr;
__device__ __forceinline__ gc64_ptr global_view(const c64* p) { return (gc64_ptr)(uintptr_t)p; }
__device__ __forceinline__ gu64_cptr global_view(const uint64_t* p) { return (gu64_cptr)(uintptr_t)p; }
__device__ __forceinline__ gu64_ptr global_view(uint64_t* p) { return (gu64_ptr)(uintptr_t)p; }
__device__ __forceinline__ c64 gload(gc64_ptr p) { f64x2_t v = *p; return {v.x, v.y}; }
// read-once data (a gate's selector rows): streaming hint, 6.2 -> 7.0 TB/s for the bare read pattern (tools/microbench)
__device__ __forceinline__ c64 gload_stream(gc64_ptr p) { f64x2_t v = __builtin_nontemporal_load(p); return {v.x, v.y}; }

__device__ __forceinline__ c64 cadd(c64 a, c64 b) { return {a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ c64 csub(c64 a, c64 b) { return {a.re - b.re, a.im - b.im}; }

// Complex product as num-complex defines it (4 mul, 1 sub, 1 add; nothing fused) — what
// complex_twist / complex_untwist / complex_mad use
// (sunscreen_tfhe/src/math/simd/scalar.rs:12-35).
__device__ __forceinline__ c64 cmul_nf(c64 a, c64 b)
{
    c64 o;
    o.re = a.re * b.re - a.im * b.im;
    o.im = a.re * b.im + a.im * b.re;
    return o;
}
// a * conj(b), same association as cmul_nf(a, {b.re, -b.im})
__device__ __forceinline__ c64 cmul_nf_conj(c64 a, c64 b)
{
    c64 o;
    o.re = a.re * b.re + a.im * b.im;
    o.im = a.im * b.re - a.re * b.im;
    return o;
}

// FFT-internal twiddle product of DAG-I: one multiply and one fma per component.
template <int DIR> __device__ __forceinline__ c64 cmul_tw(c64 a, c64 w)
{
    c64 o;
    if (DIR > 0) {
        double t1 = a.im * w.im;
        o.re = __builtin_fma(a.re, w.re, -t1);
        double t2 = a.im * w.re;
        o.im = __builtin_fma(a.re, w.im, t2);
    } else { // multiply by conj(w)
        double t1 = a.im * w.im;
        o.re = __builtin_fma(a.re, w.re, t1);
        double t2 = a.im * w.re;
        o.im = __builtin_fma(-a.re, w.im, t2);
    }
    return o;
}

constexpr double kSqrtHalf = 0.70710678118654752440; // 0x3FE6A09E667F3BCD

// ---- the canonical transform, DAG-II (r06; the CPU checker restates it operation for operation) ---------------
// DAG-I (r01-r05) issued 2 112 f64 instructions per wave and blind-rotation step of which 1 700 were one-flop adds or multiplies:
// the instruction mix alone capped the kernel at 0.375-0.45 of the FMA roof.  DAG-II is the same 8 x 8 x 8 transform re-associated:
//   * the twiddle factors between the passes enter at the INPUT of the next pass's first butterfly stage
//       a = w_j v_j;   s = a + w_{j+4} v_{j+4}  (two FMAs per component);   t = 2 a - s  (one FMA per component)
//     — the pass-1 factor W512^{(8a+b) k1} = W64^{a k1} W512^{b k1}: the first part is pass 2's input factor (table T2, indexed by
//     the lane's LOW three bits = k1), the second is common to the eight operands of a pass-2 butterfly, commutes with it and joins
//     pass 2's own W64^{b c} as pass 3's input factor W512^{b (k1 + 8c)} = W512^{b lane} (table T1, as before);
//   * the 1/sqrt(2) of the two W8 rotations multiplies their sum / difference in the last stage (v = b0 +- c q as FMAs).
// 52 f64 instructions per radix-8 without input factors (was 56), 72 with (was 84): 196 per 512-point transform (was 224), and
// the seven twiddle products that used to trail every pass are gone (tools/microbench/fft_pair_bench.hip: the pair's arithmetic
// 4 728 -> 3 812 cycles).  Same tables, same exchange images.

// acc + a * w (DIR > 0) or acc + a * conj(w) (DIR < 0): two fused multiply-adds per component
template <int DIR> __device__ __forceinline__ c64 cfma_tw(c64 acc, c64 a, c64 w)
{
    c64 o;
    if (DIR > 0) {
        o.re = __builtin_fma(-a.im, w.im, __builtin_fma(a.re, w.re, acc.re));
        o.im = __builtin_fma(a.im, w.re, __builtin_fma(a.re, w.im, acc.im));
    } else {
        o.re = __builtin_fma(a.im, w.im, __builtin_fma(a.re, w.re, acc.re));
        o.im = __builtin_fma(a.im, w.re, __builtin_fma(-a.re, w.im, acc.im));
    }
    return o;
}
__device__ __forceinline__ c64 twice_minus(c64 a, c64 s) // 2 a - s
{
    return {__builtin_fma(2.0, a.re, -s.re), __builtin_fma(2.0, a.im, -s.im)};
}

// the three stages of the 8-point DFT (decimation in frequency; DIR=+1: e^{-2 pi i jk/8}, DIR=-1: conjugate), cut apart so that
// callers can issue LDS operations between them.  s / t: sums and differences of operands (j, j + 4); u: second-stage values,
// u[6] / u[7] the W8-rotated pair's sum / difference WITHOUT its 1/sqrt(2).
template <int DIR> __device__ __forceinline__ void radix8_stage1(const c64 (&v)[8], c64 (&s)[4], c64 (&t)[4])
{
#pragma unroll
    for (int i = 0; i < 4; i++) { s[i] = cadd(v[i], v[i + 4]); t[i] = csub(v[i], v[i + 4]); }
}
// ... with the input factors of operands 1..7: tw(k) = the factor of operand k + 1 (operand 0 has none)
#ifndef SPF_STAGE1_FENCE
#define SPF_STAGE1_FENCE 0
#endif
template <int DIR, class TW> __device__ __forceinline__ void radix8_stage1_in(const c64 (&v)[8], TW tw, c64 (&s)[4], c64 (&t)[4])
{
    s[0] = cfma_tw<DIR>(v[0], v[4], tw(3));
    t[0] = twice_minus(v[0], s[0]);
#pragma unroll
    for (int j = 1; j < 4; j++) {
        if constexpr (SPF_STAGE1_FENCE) __builtin_amdgcn_sched_barrier(0); // (keeps the factor reads of the later pairs behind this one: registers)
        const c64 a = cmul_tw<DIR>(v[j], tw(j - 1));
        s[j] = cfma_tw<DIR>(a, v[j + 4], tw(j + 3));
        t[j] = twice_minus(a, s[j]);
    }
}
template <int DIR> __device__ __forceinline__ void radix8_stage2(const c64 (&s)[4], const c64 (&t)[4], c64 (&u)[8])
{
    if (DIR > 0) {
        const double p1 = t[1].re + t[1].im, m1 = t[1].im - t[1].re;
        const double p3 = t[3].re + t[3].im, m3 = t[3].im - t[3].re;
        u[6] = {p1 + m3, m1 - p3};
        u[7] = {p1 - m3, m1 + p3};
    } else {
        const double p1 = t[1].re + t[1].im, m1 = t[1].re - t[1].im;
        const double p3 = t[3].re + t[3].im, m3 = t[3].re - t[3].im;
        u[6] = {m1 - p3, p1 + m3};
        u[7] = {m1 + p3, p1 - m3};
    }
    u[0] = cadd(s[0], s[2]); u[1] = cadd(s[1], s[3]); u[2] = csub(s[0], s[2]); u[3] = csub(s[1], s[3]);
    if (DIR > 0) {
        u[4] = {t[0].re + t[2].im, t[0].im - t[2].re};
        u[5] = {t[0].re - t[2].im, t[0].im + t[2].re};
    } else {
        u[4] = {t[0].re - t[2].im, t[0].im + t[2].re};
        u[5] = {t[0].re + t[2].im, t[0].im - t[2].re};
    }
}
template <int DIR> __device__ __forceinline__ void radix8_stage3(c64 (&v)[8], const c64 (&u)[8])
{
    constexpr double c = kSqrtHalf;
    v[0] = cadd(u[0], u[1]);
    v[4] = csub(u[0], u[1]);
    if (DIR > 0) {
        v[2] = {u[2].re + u[3].im, u[2].im - u[3].re};
        v[6] = {u[2].re - u[3].im, u[2].im + u[3].re};
    } else {
        v[2] = {u[2].re - u[3].im, u[2].im + u[3].re};
        v[6] = {u[2].re + u[3].im, u[2].im - u[3].re};
    }
    v[1] = {__builtin_fma(c, u[6].re, u[4].re), __builtin_fma(c, u[6].im, u[4].im)};
    v[5] = {__builtin_fma(-c, u[6].re, u[4].re), __builtin_fma(-c, u[6].im, u[4].im)};
    if (DIR > 0) {
        v[3] = {__builtin_fma(c, u[7].im, u[5].re), __builtin_fma(-c, u[7].re, u[5].im)};
        v[7] = {__builtin_fma(-c, u[7].im, u[5].re), __builtin_fma(c, u[7].re, u[5].im)};
    } else {
        v[3] = {__builtin_fma(-c, u[7].im, u[5].re), __builtin_fma(c, u[7].re, u[5].im)};
        v[7] = {__builtin_fma(c, u[7].im, u[5].re), __builtin_fma(-c, u[7].re, u[5].im)};
    }
}

// 8-point DFT in place, no input factors (pass 1)
template <int DIR> __device__ __forceinline__ void radix8(c64 (&v)[8])
{
    c64 s[4], t[4], u[8];
    radix8_stage1<DIR>(v, s, t);
    radix8_stage2<DIR>(s, t, u);
    radix8_stage3<DIR>(v, u);
}
// 8-point DFT in place with the input factors tw(0..6) of operands 1..7 (passes 2 and 3)
template <int DIR, class TW> __device__ __forceinline__ void radix8_in(c64 (&v)[8], TW tw)
{
    c64 s[4], t[4], u[8];
    radix8_stage1_in<DIR>(v, tw, s, t);
    radix8_stage2<DIR>(s, t, u);
    radix8_stage3<DIR>(v, u);
}
// the factors of a pass as table reads (entry k at base[stride * k]) or as a register array fetched earlier
struct tw_table {
    const c64* base;
    int stride;
    __device__ __forceinline__ c64 operator()(int k) const { return base[stride * k]; }
};
struct tw_regs {
    const c64 (&w)[7];
    __device__ __forceinline__ c64 operator()(int k) const { return w[k]; }
};

// ---- LDS image of the twiddle tables (one copy per workgroup), in 16-byte complex entries.
//   T1[k1-1][lane]   = W512^{lane*k1}, k1 = 1..7          (7*64 entries)
//   T2[c-1][b]       = W64^{b*c},      c  = 1..7, b < 8   (7*8  entries)
//   WC[k]            = W1024^{k},      k < 512
//   TW[par][n']      = e^{+i pi (2n'+par)/2048}, n' < 512 (negacyclic twist, de-interleaved)
// with W_M = e^{-2 pi i / M}.  The host builds the same image (spf_hip.hip: build_tables).
constexpr int kT1Off = 0;
constexpr int kT2Off = kT1Off + 7 * 64;
constexpr int kWCOff = kT2Off + 7 * 8;
constexpr int kTWOff = kWCOff + 512;
constexpr int kTableEntries = kTWOff + 1024; // 2040 entries = 32640 bytes
constexpr int kTableBytes = kTableEntries * 16;
constexpr int kWaveBufBytes = 16384; // per-wave staging / exchange tile

// wave-local ordering point between LDS writes and the reads of other lanes' data.  DS
// operations of one wave execute in issue order; the asm keeps the compiler from moving
// memory operations across and drains the LDS queue.
__device__ __forceinline__ void wave_lds_fence()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}
// compiler-only ordering point (no instruction): LDS reads that precede it in program order
// stay ahead of the LDS writes that follow it.  The hardware needs nothing here: a wave's DS
// instructions execute in order and all 64 lanes retire an instruction together.
__device__ __forceinline__ void compiler_fence() { asm volatile("" ::: "memory"); }
// compiler-only as well, and stronger: no instruction of any kind is scheduled across it.  Pins the
// source-level interleave of two transforms (and keeps their temporaries from overlapping).
__device__ __forceinline__ void sched_fence()
{
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// One 512-point DFT (same DAG and exchange images as fft512_pair below) held by one wave; used
// by the two-waves-per-ciphertext kernel, where the partner wave on the SIMD hides the LDS
// round trips.  `buf` is the wave's private 8 KiB tile.
// PRE (0..7): how many of each pass's seven twiddles are fetched from the LDS image BEFORE the
// butterflies that precede their use, so the reads travel under ~56 f64 instructions instead of
// being issued one or two at a time right where they are needed (what the compiler does by itself to
// save registers: about five exposed LDS round trips per pass).  Costs 4 PRE live VGPRs across the
// radix-8; the kernels pick what their register budget allows.
template <int DIR, int PRE = 0>
__device__ __forceinline__ void fft512_single(c64 (&V)[8], char* buf, const c64* tab, int lane)
{
    const int hi3 = lane >> 3, lo3 = lane & 7;
    const int rd = 16 * (8 * lo3 + (hi3 ^ lo3));
    // pass 1 (no input factors), exchange 1
    radix8<DIR>(V);
#pragma unroll
    for (int k1 = 0; k1 < 8; k1++)
        *reinterpret_cast<c64*>(buf + 16 * (64 * hi3 + 8 * k1 + (lo3 ^ k1))) = V[k1];
    wave_lds_fence();
    {
        // pass 2: operand a enters with W64^{a k1}, k1 = the lane's low three bits after the exchange
        c64 tw[7];
        if constexpr (PRE > 0) {
#pragma unroll
            for (int k = 0; k < 7; k++) tw[k] = tab[kT2Off + k * 8 + lo3];
        }
#pragma unroll
        for (int a = 0; a < 8; a++) V[a] = *reinterpret_cast<const c64*>(buf + 1024 * a + rd);
        if constexpr (PRE > 0) {
            compiler_fence();
            radix8_in<DIR>(V, tw_regs{tw});
        } else {
            radix8_in<DIR>(V, tw_table{tab + kT2Off + lo3, 8});
        }
    }
    compiler_fence();
#pragma unroll
    for (int c = 0; c < 8; c++)
        *reinterpret_cast<c64*>(buf + 16 * (64 * hi3 + 8 * lo3 + (c ^ lo3))) = V[c];
    wave_lds_fence();
    {
        // pass 3: operand b enters with W512^{b lane}
        c64 tw[7];
        if constexpr (PRE > 0) {
#pragma unroll
            for (int k = 0; k < 7; k++) tw[k] = tab[kT1Off + k * 64 + lane];
        }
#pragma unroll
        for (int b = 0; b < 8; b++) V[b] = *reinterpret_cast<const c64*>(buf + 1024 * b + rd);
        if constexpr (PRE > 0) {
            compiler_fence();
            radix8_in<DIR>(V, tw_regs{tw});
        } else {
            radix8_in<DIR>(V, tw_table{tab + kT1Off + lane, 64});
        }
    }
    compiler_fence();
}

// Two independent 512-point DFTs (the even / odd sample halves of one 1024-point transform),
// DIF 8x8x8, held by one wave and advanced in lockstep so that one transform's LDS round trip
// hides under the other's butterflies.  For each: lane l, register j holds element 64*j + l on
// entry (time order) and on exit (frequency order).  bufE / bufO are private 8 KiB LDS tiles.
//   n' = 64*n1 + n0, n0 = 8a + b;   k' = k1 + 8c + 64d
//   pass 1: radix-8 over n1 -> k1, * W512^{n0 k1} (k1 != 0)     lanes (a,b)  regs k1
//   exchange 1                                                  lanes (b,k1) regs a
//   pass 2: radix-8 over a -> c,  * W64^{b c}   (c != 0)        lanes (b,k1) regs c
//   exchange 2                                                  lanes (c,k1) regs b
//   pass 3: radix-8 over b -> d                                 lanes (c,k1) regs d
// Exchange images (16-byte slots): slot(x, y, z) = 64 x + 8 z + (y ^ z) with (x,y,z) =
// (a,b,k1) resp. (b,c,k1): writes hit 8 distinct slots mod 8 per 8-lane group and reads 16
// distinct slots mod 16 per ds_read_b128 lane group — conflict-free both ways (measured:
// SQ_LDS_BANK_CONFLICT = 0).
template <int DIR>
__device__ __forceinline__ void fft512_pair(c64 (&E)[8], c64 (&O)[8], char* bufE, char* bufO,
                                            const c64* tab, int lane)
{
    const int hi3 = lane >> 3, lo3 = lane & 7;
    const int rd = 16 * (8 * lo3 + (hi3 ^ lo3));
    // pass 1 + exchange-1 writes: writer lane = 8a + b holds reg k1 -> slot 64a + 8k1 + (b ^ k1)
    radix8<DIR>(E);
#pragma unroll
    for (int k1 = 0; k1 < 8; k1++)
        *reinterpret_cast<c64*>(bufE + 16 * (64 * hi3 + 8 * k1 + (lo3 ^ k1))) = E[k1];
    radix8<DIR>(O);
#pragma unroll
    for (int k1 = 0; k1 < 8; k1++)
        *reinterpret_cast<c64*>(bufO + 16 * (64 * hi3 + 8 * k1 + (lo3 ^ k1))) = O[k1];
    wave_lds_fence();
    // exchange-1 reads: reader lane = 8b + k1 wants reg a <- slot 64a + 8k1 + (b ^ k1)
#pragma unroll
    for (int a = 0; a < 8; a++) E[a] = *reinterpret_cast<const c64*>(bufE + 1024 * a + rd);
#pragma unroll
    for (int a = 0; a < 8; a++) O[a] = *reinterpret_cast<const c64*>(bufO + 1024 * a + rd);
    // pass 2 (input factors W64^{a k1}, k1 = lo3) + exchange-2 writes: writer lane = 8b + k1 holds reg c -> slot 64b + 8k1 + (c ^ k1)
    {
        c64 tw[7];
#pragma unroll
        for (int k = 0; k < 7; k++) tw[k] = tab[kT2Off + k * 8 + lo3];
        compiler_fence();
        radix8_in<DIR>(E, tw_regs{tw});
#pragma unroll
        for (int c = 0; c < 8; c++)
            *reinterpret_cast<c64*>(bufE + 16 * (64 * hi3 + 8 * lo3 + (c ^ lo3))) = E[c];
        radix8_in<DIR>(O, tw_regs{tw});
#pragma unroll
        for (int c = 0; c < 8; c++)
            *reinterpret_cast<c64*>(bufO + 16 * (64 * hi3 + 8 * lo3 + (c ^ lo3))) = O[c];
    }
    wave_lds_fence();
    // exchange-2 reads: reader lane = 8c + k1 wants reg b <- slot 64b + 8k1 + (c ^ k1)
#pragma unroll
    for (int b = 0; b < 8; b++) E[b] = *reinterpret_cast<const c64*>(bufE + 1024 * b + rd);
#pragma unroll
    for (int b = 0; b < 8; b++) O[b] = *reinterpret_cast<const c64*>(bufO + 1024 * b + rd);
    // pass 3 (input factors W512^{b lane})
    {
        c64 tw[7];
#pragma unroll
        for (int k = 0; k < 7; k++) tw[k] = tab[kT1Off + k * 64 + lane];
        radix8_in<DIR>(E, tw_regs{tw});
        radix8_in<DIR>(O, tw_regs{tw});
    }
    compiler_fence(); // the tile's next writer stays behind these reads
}

// ---- register <-> lane transposition without LDS ------------------------------------------------
// Swap the three bits of the register index r (V[r], r = 4 r2 + 2 r1 + r0) with lane bits 5, 4, 3:
// afterwards lane (h2 h1 h0 | lo3) register (r2 r1 r0) holds what lane (r2 r1 r0 | lo3) register
// (h2 h1 h0) held.  Bit 5 and bit 4 are one `v_permlane32_swap` / `v_permlane16_swap` per dword and
// register pair (the instruction IS the 2x2 block transpose: upper half of the first operand <->
// lower half of the second), bit 3 has no swap instruction: a copy and two bank-masked DPP moves
// (row_ror:8 reaches lane ^ 8 inside a row of 16).  16 + 16 + 48 32-bit VALU instructions for the eight
// complex registers, no LDS traffic and no waitcnt.
__device__ __forceinline__ void swap_halves32(double& a, double& b)
{
    uint32_t al = (uint32_t)__double2loint(a), ah = (uint32_t)__double2hiint(a);
    uint32_t bl = (uint32_t)__double2loint(b), bh = (uint32_t)__double2hiint(b);
    auto r0 = __builtin_amdgcn_permlane32_swap(al, bl, false, false);
    auto r1 = __builtin_amdgcn_permlane32_swap(ah, bh, false, false);
    a = __hiloint2double((int)r1[0], (int)r0[0]);
    b = __hiloint2double((int)r1[1], (int)r0[1]);
}
__device__ __forceinline__ void swap_rows16(double& a, double& b)
{
    uint32_t al = (uint32_t)__double2loint(a), ah = (uint32_t)__double2hiint(a);
    uint32_t bl = (uint32_t)__double2loint(b), bh = (uint32_t)__double2hiint(b);
    auto r0 = __builtin_amdgcn_permlane16_swap(al, bl, false, false);
    auto r1 = __builtin_amdgcn_permlane16_swap(ah, bh, false, false);
    a = __hiloint2double((int)r1[0], (int)r0[0]);
    b = __hiloint2double((int)r1[1], (int)r0[1]);
}
__device__ __forceinline__ void swap_lane8(double& a, double& b)
{
    // a: lanes with bit 3 set take b from lane ^ 8; b: lanes with bit 3 clear take (the old) a from lane ^ 8
    constexpr int kRowRor8 = 0x128; // DPP row_ror:8
    int al = __double2loint(a), ah = __double2hiint(a);
    int bl = __double2loint(b), bh = __double2hiint(b);
    int nal = __builtin_amdgcn_update_dpp(al, bl, kRowRor8, 0xF, 0xC, false);
    int nah = __builtin_amdgcn_update_dpp(ah, bh, kRowRor8, 0xF, 0xC, false);
    int nbl = __builtin_amdgcn_update_dpp(bl, al, kRowRor8, 0xF, 0x3, false);
    int nbh = __builtin_amdgcn_update_dpp(bh, ah, kRowRor8, 0xF, 0x3, false);
    a = __hiloint2double(nah, nal);
    b = __hiloint2double(nbh, nbl);
}
__device__ __forceinline__ void lane_transpose_hi3(c64 (&V)[8])
{
#if defined(SPF_ABL) && SPF_ABL == 3 // (timing-only ablation: see spf_kernels.hpp)
    return;
#endif
#pragma unroll
    for (int r = 0; r < 4; r++) { // register bit 2 <-> lane bit 5
        swap_halves32(V[r].re, V[r + 4].re);
        swap_halves32(V[r].im, V[r + 4].im);
    }
#pragma unroll
    for (int r = 0; r < 8; r++) { // register bit 1 <-> lane bit 4
        if (r & 2) continue;
        swap_rows16(V[r].re, V[r + 2].re);
        swap_rows16(V[r].im, V[r + 2].im);
    }
#pragma unroll
    for (int r = 0; r < 8; r += 2) { // register bit 0 <-> lane bit 3
        swap_lane8(V[r].re, V[r + 1].re);
        swap_lane8(V[r].im, V[r + 1].im);
    }
}

// fft512_pair, software-pipelined for a wave that has its SIMD to itself (the latency kernels): there
// nothing else covers an LDS round trip, and the lockstep form above costs exactly twice one transform
// (measured 5 774 vs 2 707 cycles).  A wave's DS instructions execute in issue order, so a read issued
// behind the writes of the same image needs no drain: every exchange of one transform is issued and then
// left in flight under a radix-8 pass of the other.  Same butterflies on the same values: same words.
// (Exchange 2 in registers, `lane_transpose_hi3` as in fft512_pair1, trades the LDS store path the four waves of a CU
// share against 80 32-bit VALU instructions per transform: 4.03 -> 3.98 ms on blind_rotate4_kernel, not kept.)
template <int DIR>
__device__ __forceinline__ void fft512_pair_pipelined(c64 (&E)[8], c64 (&O)[8], char* bufE, char* bufO,
                                                      const c64* tab, int lane)
{
    const int hi3 = lane >> 3, lo3 = lane & 7;
    const int rd = 16 * (8 * lo3 + (hi3 ^ lo3));
    c64 tw2[7], tw3[7]; // the input factors of pass 2 (W64^{a k1}, k1 = lo3) and of pass 3 (W512^{b lane})
#pragma unroll
    for (int k = 0; k < 7; k++) tw2[k] = tab[kT2Off + k * 8 + lo3];
#pragma unroll
    for (int k = 0; k < 7; k++) tw3[k] = tab[kT1Off + k * 64 + lane];
    compiler_fence(); // all fourteen requested HERE (r04: without the fence hipcc sank them to their uses, two at a time, each
                      // waited for on the spot — at one wave per SIMD every such round trip is lost time)
    // E pass 1 -> exchange-1 image
    radix8<DIR>(E);
#pragma unroll
    for (int k1 = 0; k1 < 8; k1++)
        *reinterpret_cast<c64*>(bufE + 16 * (64 * hi3 + 8 * k1 + (lo3 ^ k1))) = E[k1];
    compiler_fence();
    // O pass 1, E's exchange-1 reads in flight under it
#pragma unroll
    for (int a = 0; a < 8; a++) E[a] = *reinterpret_cast<const c64*>(bufE + 1024 * a + rd);
    compiler_fence();
    radix8<DIR>(O);
#pragma unroll
    for (int k1 = 0; k1 < 8; k1++)
        *reinterpret_cast<c64*>(bufO + 16 * (64 * hi3 + 8 * k1 + (lo3 ^ k1))) = O[k1];
    compiler_fence();
#pragma unroll
    for (int a = 0; a < 8; a++) O[a] = *reinterpret_cast<const c64*>(bufO + 1024 * a + rd);
    compiler_fence();
    // E pass 2 (O's reads in flight) -> exchange-2 image
    radix8_in<DIR>(E, tw_regs{tw2});
#pragma unroll
    for (int c = 0; c < 8; c++)
        *reinterpret_cast<c64*>(bufE + 16 * (64 * hi3 + 8 * lo3 + (c ^ lo3))) = E[c];
    compiler_fence();
#pragma unroll
    for (int b = 0; b < 8; b++) E[b] = *reinterpret_cast<const c64*>(bufE + 1024 * b + rd);
    compiler_fence();
    // O pass 2 (E's reads in flight)
    radix8_in<DIR>(O, tw_regs{tw2});
#pragma unroll
    for (int c = 0; c < 8; c++)
        *reinterpret_cast<c64*>(bufO + 16 * (64 * hi3 + 8 * lo3 + (c ^ lo3))) = O[c];
    compiler_fence();
#pragma unroll
    for (int b = 0; b < 8; b++) O[b] = *reinterpret_cast<const c64*>(bufO + 1024 * b + rd);
    compiler_fence();
    // pass 3
    radix8_in<DIR>(E, tw_regs{tw3});
    radix8_in<DIR>(O, tw_regs{tw3});
    compiler_fence(); // the tile's next writer stays behind these reads
}

// fft512_pair with ONE 8 KiB exchange image for both transforms — the form that fits two waves per
// SIMD (blind_rotate2p_kernel: 256 registers, 8 KiB of LDS per wave).  A wave's DS instructions execute
// in issue order, so the two transforms can take turns on the same image as long as the program
// order is  A.write, A.read, B.write, B.read, A.write, ...: each read is issued one butterfly pass
// ahead of its use and travels under the OTHER transform's arithmetic, each write drains under it.
// Same DAG as fft512_single.  Exchange images (16-byte slots), chosen so that BOTH exchanges write to
// the same eight lane addresses (8 address registers for the whole transform pair):
//   exchange 1: writer lane (a,b) reg k1 -> 64 a + 8 k1 + (b ^ k1);  reader lane (b,k1) reg a <- 64 a + 8 k1 + (b ^ k1)
//   exchange 2: writer lane (b,k1) reg c -> 64 b + 8 c  + (k1 ^ c);  reader lane (c,k1) reg b <- 64 b + 8 c  + (k1 ^ c)
// i.e. writer (hi3, lo3) reg r -> ((64 hi3 + lo3) ^ r) + 8 r in both; both are conflict-free for
// ds_write_b128 (8 distinct slots mod 8 per 8-lane group) and ds_read_b128 (16 distinct slots mod 16
// per 16-lane group of the b128 read pattern).
// XP = 1: exchange 2 (registers c <-> lane bits 5..3 = b, the lanes are (b, k1) before and (c, k1) after:
// exactly `lane_transpose_hi3`) stays in registers; only exchange 1 goes through the image.
// XP: which transforms keep exchange 2 in registers (registers c <-> lane bits 5..3 = b; the lanes are
// (b, k1) before and (c, k1) after: exactly `lane_transpose_hi3`) instead of sending it through the
// image: 0 none, 1 both, 2 only B (balances the LDS store path against the VALU).
struct no_hook { __device__ __forceinline__ void operator()() const {} };
// `mid` is called once, half-way through the pair (behind pass 2 of A): a hook for the issue-priority schedule of the
// callers, which swap the roles of the two waves of a SIMD in the middle of a long barrier-to-barrier stretch
template <int DIR, int XP = 0, class MID = no_hook>
__device__ __forceinline__ void fft512_pair1(c64 (&A)[8], c64 (&B)[8], char* buf, const c64* tab, int lane, MID mid = MID())
{
    constexpr bool XA = XP == 1, XB = XP == 1 || XP == 2;
    const int hi3 = lane >> 3, lo3 = lane & 7;
    const uint32_t rd1 = 16 * (8 * lo3 + (hi3 ^ lo3));
    const uint32_t rd2 = 16 * (8 * hi3 + (hi3 ^ lo3));
    const uint32_t wbase = 16 * (64 * hi3 + lo3);
    char* wr[8];
#pragma unroll
    for (int r = 0; r < 8; r++) wr[r] = buf + ((wbase ^ (16 * r)) + 128 * r);
    const tw_table t2{tab + kT2Off + lo3, 8};   // pass 2: W64^{a k1}, k1 = lo3
    const tw_table t3{tab + kT1Off + lane, 64}; // pass 3: W512^{b lane}
    // pass 1
    radix8<DIR>(A);
#pragma unroll
    for (int k1 = 0; k1 < 8; k1++) *reinterpret_cast<c64*>(wr[k1]) = A[k1];
    sched_fence();
    radix8<DIR>(B);
    sched_fence();
#pragma unroll
    for (int a = 0; a < 8; a++) A[a] = *reinterpret_cast<const c64*>(buf + 1024 * a + rd1);
    sched_fence();
#pragma unroll
    for (int k1 = 0; k1 < 8; k1++) *reinterpret_cast<c64*>(wr[k1]) = B[k1];
    sched_fence();
    // pass 2 of A; B's exchange-1 reads travel under it
    radix8_in<DIR>(A, t2);
    sched_fence();
    mid();
#pragma unroll
    for (int a = 0; a < 8; a++) B[a] = *reinterpret_cast<const c64*>(buf + 1024 * a + rd1);
    sched_fence();
    if constexpr (XA) {
        lane_transpose_hi3(A);
        radix8_in<DIR>(A, t3); // pass 3 of A
    } else {
#pragma unroll
        for (int c = 0; c < 8; c++) *reinterpret_cast<c64*>(wr[c]) = A[c];
    }
    sched_fence();
    radix8_in<DIR>(B, t2);
    sched_fence();
    if constexpr (!XA) {
#pragma unroll
        for (int b = 0; b < 8; b++) A[b] = *reinterpret_cast<const c64*>(buf + 1024 * b + rd2);
        sched_fence();
    }
    if constexpr (XB) {
        lane_transpose_hi3(B);
        if constexpr (!XA) radix8_in<DIR>(A, t3); // pass 3 of A, its exchange-2 reads having travelled under B's transposition
        radix8_in<DIR>(B, t3);
    } else {
#pragma unroll
        for (int c = 0; c < 8; c++) *reinterpret_cast<c64*>(wr[c]) = B[c];
        sched_fence();
        // pass 3
        radix8_in<DIR>(A, t3);
        sched_fence();
#pragma unroll
        for (int b = 0; b < 8; b++) B[b] = *reinterpret_cast<const c64*>(buf + 1024 * b + rd2);
        radix8_in<DIR>(B, t3);
    }
    sched_fence(); // the image's next writer stays behind these reads
}

// ---- fft512_pair1 with the LDS stores SPREAD through the arithmetic ------------------------------------------------------
// tools/microbench/fft_pair_bench.hip (r04): eight waves per CU running transform pairs in lockstep need 7.9 k cycles per pair,
// the butterflies alone 4.7 k and the LDS traffic alone 4.2 k — the two hardly overlap, because each exchange leaves the wave as
// a burst of eight `ds_write_b128` (13 cycles of the CU's store path each, all eight waves bursting at once) and the other
// transform's table reads queue behind it.  The spread forms issue the eight stores of one transform between the three
// butterfly stages of the OTHER transform; the operations and their order per value are those of `radix8` / `radix8_in`: same
// words.  (r04-r05 also carried fft512_pair1s / pair1e / pair1te — spread stores with per-transform table reads, early reads —
// measured slower than the forms below and removed with DAG-I.)

// radix-8 of X (with the input factors tw, or without: TW = tw_none) with the eight LDS stores op(0) .. op(7) of the caller spread
// through it and `after()` — the reads of the transform just stored — issued behind the last store.  SPF_SPREAD: where the stores
// go — 0: three behind stage 1, three behind stage 2, two behind stage 3; 1: one behind each operand pair of stage 1 (ten f64
// instructions apart: the CU's store path takes ~13 cycles per ds_write_b128), two behind stage 2, two behind stage 3;
// 2: two behind the second and the fourth pair of stage 1, two behind stage 2, two behind stage 3
#ifndef SPF_SPREAD
#define SPF_SPREAD 1
#endif
struct tw_none {};
template <int DIR, class TW, class OP, class AFTER>
__device__ __forceinline__ void radix8_any_spread(c64 (&X)[8], TW tw, OP op, AFTER after)
{
    c64 s[4], t[4], u[8];
    constexpr bool IN = !std::is_same<TW, tw_none>::value;
    if constexpr (SPF_SPREAD == 0) {
        if constexpr (IN) radix8_stage1_in<DIR>(X, tw, s, t); else radix8_stage1<DIR>(X, s, t);
        sched_fence();
        op(0); op(1); op(2);
        sched_fence();
        radix8_stage2<DIR>(s, t, u);
        sched_fence();
        op(3); op(4); op(5);
        sched_fence();
        radix8_stage3<DIR>(X, u);
        sched_fence();
        op(6); op(7);
    } else {
        // stage 1 pair by pair
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if constexpr (IN) {
                if (j == 0) {
                    s[0] = cfma_tw<DIR>(X[0], X[4], tw(3));
                    t[0] = twice_minus(X[0], s[0]);
                } else {
                    const c64 a = cmul_tw<DIR>(X[j], tw(j - 1));
                    s[j] = cfma_tw<DIR>(a, X[j + 4], tw(j + 3));
                    t[j] = twice_minus(a, s[j]);
                }
            } else {
                s[j] = cadd(X[j], X[j + 4]);
                t[j] = csub(X[j], X[j + 4]);
            }
            if constexpr (SPF_SPREAD == 1) {
                sched_fence();
                op(j);
                sched_fence();
            } else if (j == 1 || j == 3) {
                sched_fence();
                op(j - 1); op(j);
                sched_fence();
            }
        }
        radix8_stage2<DIR>(s, t, u);
        sched_fence();
        op(4); op(5);
        sched_fence();
        radix8_stage3<DIR>(X, u);
        sched_fence();
        op(6); op(7);
    }
    sched_fence();
    after();
    sched_fence();
}
template <int DIR, class TW, class OP, class AFTER>
__device__ __forceinline__ void radix8_in_spread2(c64 (&X)[8], TW tw, OP op, AFTER after) { radix8_any_spread<DIR>(X, tw, op, after); }
template <int DIR, class OP, class AFTER>
__device__ __forceinline__ void radix8_spread2(c64 (&X)[8], OP op, AFTER after) { radix8_any_spread<DIR>(X, tw_none{}, op, after); }
template <int DIR, class TW, class OP>
__device__ __forceinline__ void radix8_in_spread(c64 (&X)[8], TW tw, OP op) { radix8_any_spread<DIR>(X, tw, op, [] {}); }
template <int DIR, class OP>
__device__ __forceinline__ void radix8_spread(c64 (&X)[8], OP op) { radix8_any_spread<DIR>(X, tw_none{}, op, [] {}); }

// the factors of a pass for the spread pairs: fetched into registers ahead of the pass and shared by the two transforms (1), or
// read from the table where they are used (0: 28 registers less across the first butterfly stage, which in DAG-II holds the eight
// operands, their factors and the eight results at once)
#ifndef SPF_PAIR_TW_REGS
#define SPF_PAIR_TW_REGS 0
#endif
struct tw_pick {
    const c64 (&w)[7];
    tw_table t;
    __device__ __forceinline__ c64 operator()(int k) const
    {
        if constexpr (SPF_PAIR_TW_REGS) return w[k];
        else return t(k);
    }
};
#define SPF_TW(regs, table) tw_pick{regs, table}

// ---- fft512_pair1 with the pass factors requested EARLY and SHARED by the two transforms -----------------------------------
// The seven factors of a pass are requested once, ahead of the butterflies that use them, and serve both transforms (lane-indexed:
// the same for A and B).  28 registers, live from the request to the second transform's first stage.
template <int DIR, int XP = 2, class MID = no_hook>
__device__ __forceinline__ void fft512_pair1t(c64 (&A)[8], c64 (&B)[8], char* buf, const c64* tab, int lane, MID mid = MID())
{
    static_assert(XP == 1 || XP == 2, "exchange 2 of B (XP = 2) or of both transforms (XP = 1) in registers");
    constexpr bool XA = XP == 1;
    const int hi3 = lane >> 3, lo3 = lane & 7;
    const uint32_t rd1 = 16 * (8 * lo3 + (hi3 ^ lo3));
    const uint32_t rd2 = 16 * (8 * hi3 + (hi3 ^ lo3));
    const uint32_t wbase = 16 * (64 * hi3 + lo3);
    char* wr[8];
#pragma unroll
    for (int r = 0; r < 8; r++) wr[r] = buf + ((wbase ^ (16 * r)) + 128 * r);
    c64 tw[7];
    // pass 1 of both, exchange 1 of A
    radix8<DIR>(A);
#pragma unroll
    for (int k1 = 0; k1 < 8; k1++) *reinterpret_cast<c64*>(wr[k1]) = A[k1];
    sched_fence();
    // pass 2's factors (W64^{a k1}, k1 = lo3) for both transforms, requested ahead of B's pass 1
#pragma unroll
    for (int k = 0; k < 7; k++) tw[k] = tab[kT2Off + k * 8 + lo3];
    compiler_fence();
    radix8<DIR>(B);
    sched_fence();
#pragma unroll
    for (int a = 0; a < 8; a++) A[a] = *reinterpret_cast<const c64*>(buf + 1024 * a + rd1);
    sched_fence();
#pragma unroll
    for (int k1 = 0; k1 < 8; k1++) *reinterpret_cast<c64*>(wr[k1]) = B[k1];
    sched_fence();
    radix8_in<DIR>(A, tw_regs{tw});
    sched_fence();
    mid();
#pragma unroll
    for (int a = 0; a < 8; a++) B[a] = *reinterpret_cast<const c64*>(buf + 1024 * a + rd1);
    sched_fence();
    if constexpr (!XA) {
#pragma unroll
        for (int c = 0; c < 8; c++) *reinterpret_cast<c64*>(wr[c]) = A[c];
        sched_fence();
    }
    radix8_in<DIR>(B, tw_regs{tw});
    sched_fence();
    // pass 3's factors (W512^{b lane}) for both
#pragma unroll
    for (int k = 0; k < 7; k++) tw[k] = tab[kT1Off + k * 64 + lane];
    compiler_fence();
    if constexpr (XA) {
        lane_transpose_hi3(A);
    } else {
#pragma unroll
        for (int b = 0; b < 8; b++) A[b] = *reinterpret_cast<const c64*>(buf + 1024 * b + rd2);
        sched_fence();
    }
    lane_transpose_hi3(B);
    radix8_in<DIR>(A, tw_regs{tw}); // pass 3 of A (its exchange-2 reads having travelled under B's transposition)
    radix8_in<DIR>(B, tw_regs{tw});
    sched_fence(); // the image's next writer stays behind these reads
}

// XP = 0 tail of the pairs below: exchange 2 of B through the image as well, behind A's (A's exchange-2 reads are issued, a wave's
// DS instructions execute in order, so B's stores cannot overtake them): eight stores and eight reads on the LDS pipe instead of
// lane_transpose_hi3's 80 vector instructions — 16 v_permlane*_swap at 8.5 cycles of the SIMD each, 32 DPP moves at 4.5, 16 moves
// (tools/microbench/valu_rates.hip): ~480 of a pair's vector cycles.  B's reads travel under A's last radix-8.
template <int DIR, class TW, class ST, class LD>
__device__ __forceinline__ void pair_tail_through_image(c64 (&A)[8], c64 (&B)[8], TW t3, ST store, LD load, uint32_t rd2)
{
#pragma unroll
    for (int k = 0; k < 8; k++) store(B, k);
    sched_fence();
    load(B, rd2);
    sched_fence();
    radix8_in<DIR>(A, t3);
    sched_fence();
    radix8_in<DIR>(B, t3);
}
// fft512_pair1t with (EARLY) the reads of an exchange issued right behind its own stores and / or (SPREAD) the stores of one
// transform issued between the butterfly stages of the other
template <int DIR, int XP, bool EARLY, bool SPREAD, class MID = no_hook>
__device__ __forceinline__ void fft512_pair1x(c64 (&A)[8], c64 (&B)[8], char* buf, const c64* tab, int lane, MID mid = MID())
{
    static_assert(XP == 2 || XP == 0, "exchange 2 of B in registers (2) or through the image behind A's (0)");
    const int hi3 = lane >> 3, lo3 = lane & 7;
    const uint32_t rd1 = 16 * (8 * lo3 + (hi3 ^ lo3));
    const uint32_t rd2 = 16 * (8 * hi3 + (hi3 ^ lo3));
    const uint32_t wbase = 16 * (64 * hi3 + lo3);
    char* wr[8];
#pragma unroll
    for (int r = 0; r < 8; r++) wr[r] = buf + ((wbase ^ (16 * r)) + 128 * r);
    auto store = [&](c64 (&X)[8], int k) { *reinterpret_cast<c64*>(wr[k]) = X[k]; };
    auto load = [&](c64 (&X)[8], uint32_t rd) {
#pragma unroll
        for (int a = 0; a < 8; a++) X[a] = *reinterpret_cast<const c64*>(buf + 1024 * a + rd);
    };
    c64 tw[7];
    const tw_table t2{tab + kT2Off + lo3, 8}, t3{tab + kT1Off + lane, 64};
    // pass 1 of A
    radix8<DIR>(A);
    // pass 2's factors for both transforms, requested ahead of B's pass 1
    if constexpr (SPF_PAIR_TW_REGS) {
#pragma unroll
        for (int k = 0; k < 7; k++) tw[k] = tab[kT2Off + k * 8 + lo3];
        compiler_fence();
    }
    if constexpr (!SPREAD) {
#pragma unroll
        for (int k = 0; k < 8; k++) store(A, k);
        sched_fence();
        if constexpr (EARLY) { load(A, rd1); sched_fence(); }
        radix8<DIR>(B);
        sched_fence();
    } else {
        sched_fence();
        radix8_spread<DIR>(B, [&](int k) { store(A, k); });
    }
    if constexpr (!EARLY || SPREAD) { load(A, rd1); sched_fence(); }
    // exchange 1 of B, pass 2 of A
    if constexpr (!SPREAD) {
#pragma unroll
        for (int k = 0; k < 8; k++) store(B, k);
        sched_fence();
        if constexpr (EARLY) { load(B, rd1); sched_fence(); }
        radix8_in<DIR>(A, SPF_TW(tw, t2));
        sched_fence();
    } else {
        radix8_in_spread<DIR>(A, SPF_TW(tw, t2), [&](int k) { store(B, k); });
    }
    mid();
    if constexpr (!EARLY || SPREAD) { load(B, rd1); sched_fence(); }
    // exchange 2 of A, pass 2 of B
    if constexpr (!SPREAD) {
#pragma unroll
        for (int k = 0; k < 8; k++) store(A, k);
        sched_fence();
        if constexpr (EARLY) { load(A, rd2); sched_fence(); }
        radix8_in<DIR>(B, SPF_TW(tw, t2));
        sched_fence();
    } else {
        radix8_in_spread<DIR>(B, SPF_TW(tw, t2), [&](int k) { store(A, k); });
    }
    // pass 3's factors for both
    if constexpr (SPF_PAIR_TW_REGS) {
#pragma unroll
        for (int k = 0; k < 7; k++) tw[k] = tab[kT1Off + k * 64 + lane];
        compiler_fence();
    }
    if constexpr (!EARLY || SPREAD) { load(A, rd2); sched_fence(); }
    if constexpr (XP == 2) {
        lane_transpose_hi3(B);
        radix8_in<DIR>(A, SPF_TW(tw, t3));
        radix8_in<DIR>(B, SPF_TW(tw, t3));
    } else {
        pair_tail_through_image<DIR>(A, B, SPF_TW(tw, t3), store, load, rd2);
    }
    sched_fence();
}
// fft512_pair1ts with each exchange's reads issued inside the other transform's pass, right behind the last spread store
template <int DIR, int XP = 2, class MID = no_hook>
__device__ __forceinline__ void fft512_pair1ts2(c64 (&A)[8], c64 (&B)[8], char* buf, const c64* tab, int lane, MID mid = MID())
{
    static_assert(XP == 2 || XP == 0, "exchange 2 of B in registers (2) or through the image behind A's (0)");
    const int hi3 = lane >> 3, lo3 = lane & 7;
    const uint32_t rd1 = 16 * (8 * lo3 + (hi3 ^ lo3));
    const uint32_t rd2 = 16 * (8 * hi3 + (hi3 ^ lo3));
    const uint32_t wbase = 16 * (64 * hi3 + lo3);
    char* wr[8];
#pragma unroll
    for (int r = 0; r < 8; r++) wr[r] = buf + ((wbase ^ (16 * r)) + 128 * r);
    auto store = [&](c64 (&X)[8], int k) { *reinterpret_cast<c64*>(wr[k]) = X[k]; };
    auto load = [&](c64 (&X)[8], uint32_t rd) {
#pragma unroll
        for (int a = 0; a < 8; a++) X[a] = *reinterpret_cast<const c64*>(buf + 1024 * a + rd);
    };
    c64 tw[7];
    const tw_table t2{tab + kT2Off + lo3, 8}, t3{tab + kT1Off + lane, 64};
    radix8<DIR>(A);
    sched_fence();
    // pass 2's factors for both transforms, requested ahead of B's pass 1
    if constexpr (SPF_PAIR_TW_REGS) {
#pragma unroll
        for (int k = 0; k < 7; k++) tw[k] = tab[kT2Off + k * 8 + lo3];
        compiler_fence();
    }
    // pass 1 of B: A's exchange-1 stores in its butterflies, A's exchange-1 reads behind the last store
    radix8_spread2<DIR>(B, [&](int k) { store(A, k); }, [&]() { load(A, rd1); });
    // pass 2 of A: B's exchange-1 stores, then B's exchange-1 reads
    radix8_in_spread2<DIR>(A, SPF_TW(tw, t2), [&](int k) { store(B, k); }, [&]() { load(B, rd1); });
    mid();
    // pass 2 of B: A's exchange-2 stores, then A's exchange-2 reads
    radix8_in_spread2<DIR>(B, SPF_TW(tw, t2), [&](int k) { store(A, k); }, [&]() { load(A, rd2); });
    // pass 3's factors for both
    if constexpr (SPF_PAIR_TW_REGS) {
#pragma unroll
        for (int k = 0; k < 7; k++) tw[k] = tab[kT1Off + k * 64 + lane];
        compiler_fence();
    }
    if constexpr (XP == 2) {
        lane_transpose_hi3(B);
        radix8_in<DIR>(A, SPF_TW(tw, t3));
        radix8_in<DIR>(B, SPF_TW(tw, t3));
    } else {
        pair_tail_through_image<DIR>(A, B, SPF_TW(tw, t3), store, load, rd2);
    }
    sched_fence();
}
template <int DIR, int XP = 2, class MID = no_hook>
__device__ __forceinline__ void fft512_pair1ts(c64 (&A)[8], c64 (&B)[8], char* buf, const c64* tab, int lane, MID mid = MID())
{
    fft512_pair1x<DIR, XP, false, true>(A, B, buf, tab, lane, mid);
}

// round half away from zero, then reduce mod 2^64 into the torus exactly as
// PolynomialFftRef::ifft does (entities/polynomial_fft.rs:82-99 -> simd/scalar.rs:26-35,
// 75-119 -> `x as i64` saturating, math/torus.rs:177-192).
__device__ __forceinline__ uint64_t f64_round_to_torus(double x)
{
    const double q = 18446744073709551616.0;      // 2^64
    const double q_div_2 = 9223372036854775808.0; // 2^63
    double v = __builtin_round(x);
    double m = __builtin_fma(-__builtin_trunc(v * (1.0 / q)), q, v);
    // branch-free form of `if m >= q/2 { m -= q } else if m <= -q/2 { m += q }`
    double adj = (m >= q_div_2) ? -q : ((m <= -q_div_2) ? q : 0.0);
    m += adj;
    // m is now in [-2^63, 2^63]; `as i64` saturates the single out-of-range value +2^63
    bool sat = m >= q_div_2;
    long long r = (long long)(sat ? 0.0 : m);
    r = sat ? 0x7FFFFFFFFFFFFFFFll : r;
    return (uint64_t)r;
}

// Same result as f64_round_to_torus for an input that is already an integer with |v| >= 2^52
// (so round() is the identity): the low 64 bits of v in two's complement via two exact
// floor/fma splits, then the saturating-cast quirk (v mod 2^64 == -2^63 reached from below
// zero becomes +2^63 and saturates to 0x7FFF...F).
__device__ __forceinline__ uint64_t f64_bigint_to_torus(double v)
{
    const double two32 = 4294967296.0, inv32 = 1.0 / 4294967296.0;
    double hi = __builtin_floor(v * inv32);
    double lo = __builtin_fma(hi, -two32, v);   // in [0, 2^32), exact
    double hi2 = __builtin_floor(hi * inv32);
    double hil = __builtin_fma(hi2, -two32, hi); // in [0, 2^32), exact
    uint32_t ulo = (uint32_t)lo, uhi = (uint32_t)hil;
    bool quirk = (v < 0.0) && (ulo == 0u) && (uhi == 0x80000000u);
    uhi = quirk ? 0x7FFFFFFFu : uhi;
    ulo = quirk ? 0xFFFFFFFFu : ulo;
    return ((uint64_t)uhi << 32) | ulo;
}

// Integer form of f64_bigint_to_torus: for an integer-valued v with 2^52 <= |v| < 2^116 the low 64
// bits of v in two's complement are (mantissa << (exponent - 1075)), negated for v < 0 — nine 32-bit
// VALU instructions instead of seven f64 ones and two conversions.  `sh_or` collects the shift
// amounts (the caller checks once per wave that every one is in [0, 64): that IS the magnitude
// test), `quirk_min` becomes 0 if some value hits the saturating-cast quirk (v < 0 and low 64 bits
// == 2^63, which `as i64` turns into 0x7FFF...F): the caller then redoes the wave's values with the
// literal sequence.  Same words as f64_round_to_torus whenever the two checks pass.
__device__ __forceinline__ uint64_t f64_bigint_to_torus_bits(double v, uint32_t& sh_or, uint32_t& quirk_min)
{
    const uint64_t b = (uint64_t)__double_as_longlong(v);
    const uint32_t hi = (uint32_t)(b >> 32), lo = (uint32_t)b;
    const uint32_t sh = ((hi >> 20) & 0x7FFu) - 1075u;
    sh_or |= sh;
    const uint64_t m = ((uint64_t)((hi & 0xFFFFFu) | 0x100000u) << 32) | lo;
    uint64_t r = m << (sh & 63u);
    const uint32_t s32 = (uint32_t)((int32_t)hi >> 31);
    const uint64_t s = ((uint64_t)s32 << 32) | s32;
    r = (r ^ s) - s;
    const uint32_t q = (uint32_t)r | ((uint32_t)(r >> 32) ^ 0x80000000u) | ~s32;
    quirk_min = q < quirk_min ? q : quirk_min;
    return r;
}

// r02's form of the fast path (mantissa extracted, shift amounts OR-ed, quirk by a running minimum): kept for cbs_trace_kernel,
// which is 5 % slower with the form below (4.38 against 4.58 ms per 4096; the blind rotation is 1 % faster with it)
__device__ __forceinline__ void torus_bits16_mantissa(const double (&tv)[16], uint64_t (&t)[16])
{
    uint32_t sh_or = 0, qmin = 0xFFFFFFFFu;
#pragma unroll
    for (int e = 0; e < 16; e++) t[e] = f64_bigint_to_torus_bits(tv[e], sh_or, qmin);
    if (!__all(sh_or < 64u && qmin != 0u)) {
#pragma unroll
        for (int e = 0; e < 16; e++) t[e] = f64_round_to_torus(tv[e]);
    }
}
// untwist_to_torus with an integer conversion on the fast path (r03: 42.8 -> 42.3 ms per 4096 against r02's mantissa-extracting form).  For an integer-valued v with
// 2^64 <= |v| < 2^116 the double's own bits shifted left by (exponent - 1075) mod 64 = (exponent + 13) & 63 ARE the low 64
// bits of |v|: the shift is at least 12, so sign, exponent field and the implicit one all leave the word.  Negated for
// v < 0.  Two checks per wave decide whether the sixteen values take this path: every exponent in [1087, 1138] (that IS the
// magnitude test; products of digits and torus words sit around 2^74..2^85), and no magnitude with the high word
// 0x80000000 — a superset of the saturating-cast quirk (v < 0 and |v| = 2^63 mod 2^64, which `as i64` turns into 0x7FFF...F;
// math/torus.rs:177-192), tested with one compare whose result goes to the scalar unit.  Otherwise the wave redoes its
// values with the literal sequence.  Same words as f64_round_to_torus in every case.
__device__ __forceinline__ void torus_bits16(const double (&tv)[16], uint64_t (&t)[16])
{
    uint32_t emin = 0xFFFFFFFFu, emax = 0u;
    uint64_t quirk = 0;
#pragma unroll
    for (int e = 0; e < 16; e++) {
        const uint64_t b = (uint64_t)__double_as_longlong(tv[e]);
        const uint32_t hi = (uint32_t)(b >> 32);
        const uint32_t e11 = (hi >> 20) & 0x7FFu;
        emin = e11 < emin ? e11 : emin;
        emax = e11 > emax ? e11 : emax;
        const uint64_t r = b << ((e11 + 13u) & 63u);
        quirk |= __ballot((uint32_t)(r >> 32) == 0x80000000u);
        const uint32_t s32 = (uint32_t)((int32_t)hi >> 31);
        const uint64_t sg = ((uint64_t)s32 << 32) | s32;
        t[e] = (r ^ sg) - sg;
    }
    if (!__all(emin >= 1087u && emax <= 1138u) || quirk != 0) {
#pragma unroll
        for (int e = 0; e < 16; e++) t[e] = f64_round_to_torus(tv[e]);
    }
}
#ifndef SPF_UNTWIST_PRE
#define SPF_UNTWIST_PRE 0
#endif
// PRESCALED: the 1/1024 of the inverse transform is already in V — the blind-rotation kernels multiply with a bootstrap key
// whose device image carries it (`scale_bootstrap_key_kernel`); a power of two commutes with every rounding on the way, so the
// words are the same and 32 multiplications per polynomial are not executed
template <bool MANTISSA_FORM = false, bool PRESCALED = false>
__device__ __forceinline__ void untwist_to_torus_bits(const c64 (&V)[8], const c64* twist_lds, uint64_t (&t)[16])
{
    double tv[16];
#if SPF_UNTWIST_PRE
    c64 twf[8];
#pragma unroll
    for (int n1 = 0; n1 < 8; n1++) twf[n1] = twist_lds[64 * n1];
    compiler_fence();
#endif
#pragma unroll
    for (int n1 = 0; n1 < 8; n1++) {
        c64 xs = V[n1];
        if constexpr (!PRESCALED) xs = {V[n1].re * (1.0 / 1024.0), V[n1].im * (1.0 / 1024.0)};
#if SPF_UNTWIST_PRE
        c64 u = cmul_nf_conj(xs, twf[n1]);
#else
        c64 u = cmul_nf_conj(xs, twist_lds[64 * n1]);
#endif
        tv[n1] = u.re;
        tv[8 + n1] = u.im;
    }
    if constexpr (MANTISSA_FORM) torus_bits16_mantissa(tv, t);
    else torus_bits16(tv, t);
}
// a + (uint64_t)b as ONE v_mad_u64_u32 (b * 1 + a; 4.9 cycles of a SIMD): hipcc widens b to a register pair (a move) and adds
// with v_lshl_add_u64, or emits a v_add_co / v_addc pair (9.8 cycles)
__device__ __forceinline__ uint64_t add_u32_to_u64(uint64_t a, uint32_t b)
{
    uint64_t out, carry;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %3" : "=&v"(out), "=s"(carry) : "v"(b), "v"(a));
    return out;
}

// NEGATED-ACCUMULATOR form of untwist_to_torus_bits + `acc += t` (blind_rotate2p_body with SPF_BR_NEG): the caller keeps
// nacc = -acc (mod 2^64) and this does nacc -= t, i.e. nacc += (-t).  Same words as the plain form, fewer VALU cycles
// (tools/microbench/valu_rates.hip, profiles/r05_valu_rates.md: a v_sub_co / v_subb pair costs 9.8 cycles of a SIMD, the
// 64-bit add v_lshl_add_u64 4.8, a VOP2 32-bit operation 2.5, a VOP3 one 4.5):
//   * the untwist product is formed NEGATED for free ((-m1) - m2 and m3 - m4 swapped: source modifiers, exact), so the
//     integer of the negated value is what gets ADDED — no subtraction anywhere;
//   * two's complement of the shifted magnitude bits as (r + s) ^ s with s = the sign spread over 64 bits (a 64-bit shift,
//     a 64-bit add, two XORs) instead of (r ^ s) - s (…, a subtract pair);
//   * the exponent window [1087, 1138] is checked on (high word << 1) — the sign leaves, one VOP2 shift instead of a VOP3
//     bit-field extract — and the saturating-cast quirk by a running signed minimum of the results' high words (INT_MIN
//     <=> some magnitude has the high word 0x80000000; VOP2) instead of one 64-bit compare per value.
// The literal fall-back takes the original value (-tvn, exact) and is subtracted.
// TWIST_AT_ONCE: the eight twist factors requested together ahead of the products (the latency shapes, one wave per SIMD: 3.572 ->
// 3.542 ms per 64, 6.695 -> 6.669 per 512; the four-per-workgroup shape is 0.1 % slower with it)
template <bool PRESCALED, bool TWIST_AT_ONCE = false>
__device__ __forceinline__ void untwist_sub_from_negated(const c64 (&V)[8], const c64* twist_lds, uint64_t (&nacc)[16])
{
    double tvn[16];
    c64 twf[TWIST_AT_ONCE ? 8 : 1];
    if constexpr (TWIST_AT_ONCE) {
#pragma unroll
        for (int n1 = 0; n1 < 8; n1++) twf[n1] = twist_lds[64 * n1];
        compiler_fence();
    }
#pragma unroll
    for (int n1 = 0; n1 < 8; n1++) {
        c64 xs = V[n1];
        if constexpr (!PRESCALED) xs = {V[n1].re * (1.0 / 1024.0), V[n1].im * (1.0 / 1024.0)};
        const c64 tw = TWIST_AT_ONCE ? twf[TWIST_AT_ONCE ? n1 : 0] : twist_lds[64 * n1];
        // cmul_nf_conj negated: re = -(a.re b.re) - (a.im b.im), im = a.re b.im - a.im b.re
        tvn[n1] = -(xs.re * tw.re) - xs.im * tw.im;
        tvn[8 + n1] = xs.re * tw.im - xs.im * tw.re;
    }
    uint32_t kmin = 0xFFFFFFFFu, kmax = 0u;
    int32_t qmin = 0x7FFFFFFF;
    uint64_t t[16];
#pragma unroll
    for (int e = 0; e < 16; e++) {
        const uint64_t b = (uint64_t)__double_as_longlong(tvn[e]);
        const uint32_t hi = (uint32_t)(b >> 32);
        const uint32_t k = hi << 1; // exponent field in bits 31..21, sign gone
        kmin = k < kmin ? k : kmin;
        kmax = k > kmax ? k : kmax;
        const uint64_t r = b << (((hi >> 20) + 13u) & 63u);
        const int32_t rh = (int32_t)(uint32_t)(r >> 32);
        qmin = rh < qmin ? rh : qmin;
        const uint64_t sg = (uint64_t)((int64_t)b >> 63);
        t[e] = (r + sg) ^ sg;
    }
    if (__all(kmin >= (1087u << 21) && kmax < (1139u << 21) && qmin != (int32_t)0x80000000)) {
#pragma unroll
        for (int e = 0; e < 16; e++) nacc[e] += t[e];
    } else {
#pragma unroll
        for (int e = 0; e < 16; e++) nacc[e] -= f64_round_to_torus(-tvn[e]);
    }
}
// the same with the twist factors held in registers (the latency kernels)
__device__ __forceinline__ void untwist_to_torus_bits(const c64 (&V)[8], const c64 (&twist)[8], uint64_t (&t)[16])
{
    double tv[16];
#pragma unroll
    for (int n1 = 0; n1 < 8; n1++) {
        c64 xs = {V[n1].re * (1.0 / 1024.0), V[n1].im * (1.0 / 1024.0)};
        c64 u = cmul_nf_conj(xs, twist[n1]);
        tv[n1] = u.re;
        tv[8 + n1] = u.im;
    }
    torus_bits16(tv, t);
}

// ---- pieces shared by the latency-shape kernels (blind_rotate8 / cmux4) ----------

// RadixDecomposition of one torus word (math/radix.rs:81-113,157-162; simd/scalar.rs:52-71): round to
// the top L*LOGB bits (add the bit below), then L digits, least significant first, each reduced to
// [-B/2, B/2) with the carry going into the next; returned packed, LOGB bits per digit, two's
// complement inside each field.
template <int L, int LOGB>
__device__ __forceinline__ uint32_t gadget_digits_packed(uint64_t x)
{
    static_assert(L * LOGB <= 32, "packed digits need L*LOGB <= 32");
    constexpr int shift = 64 - L * LOGB;
    uint32_t s = (uint32_t)(x >> shift) + (uint32_t)((x >> (shift - 1)) & 1);
    uint32_t packed = 0;
#pragma unroll
    for (int j = 0; j < L; j++) {
        uint32_t d = s & ((1u << LOGB) - 1);
        s >>= LOGB;
        s += d >> (LOGB - 1);
        packed |= d << (j * LOGB);
    }
    return packed;
}

// complex sample of digit j: re from the packed word of coefficient c, im from that of c + N/2,
// sign-extended, converted to f64 and twisted (entities/polynomial.rs:257-274, scalar.rs:19-23)
// The two-digit case (L = 2, LOGB = 16) without the packed form: the rounded top word s itself is kept, and
//   digit 0 = sext16(s),   digit 1 = sext16((s >> 16) + bit15(s)) = (int)(s + 0x8000) >> 16
// (radix.rs:157-162: digit, shift, carry of digit >= B/2 into the next, digits as two's-complement small integers).
__device__ __forceinline__ uint32_t gadget_round_top32(uint64_t x)
{
    return (uint32_t)(x >> 32) + (uint32_t)((x >> 31) & 1);
}
__device__ __forceinline__ c64 twisted_digit_top32(uint32_t s_re, uint32_t s_im, int j, c64 tw)
{
    const int dre = j == 0 ? (int)(int16_t)(s_re & 0xFFFFu) : (int)(s_re + 0x8000u) >> 16;
    const int dim = j == 0 ? (int)(int16_t)(s_im & 0xFFFFu) : (int)(s_im + 0x8000u) >> 16;
    return cmul_nf({(double)dre, (double)dim}, tw);
}

template <int LOGB>
__device__ __forceinline__ c64 twisted_digit(uint32_t packed_re, uint32_t packed_im, int j, c64 tw)
{
    const int sh = j * LOGB;
    int dre = ((int)(packed_re << (32 - LOGB - sh))) >> (32 - LOGB);
    int dim = ((int)(packed_im << (32 - LOGB - sh))) >> (32 - LOGB);
    return cmul_nf({(double)dre, (double)dim}, tw);
}

// inverse side of a transform: (z * 1/1024) * conj(twist), then round / mod 2^64 / saturating cast
// of the 16 values (polynomial_fft.rs:82-99): the short exact path when every value of the wave is
// already an integer of magnitude >= 2^52, else the literal sequence; identical words either way.
__device__ __forceinline__ void untwist_to_torus(const c64 (&V)[8], const c64 (&twist)[8], uint64_t (&t)[16])
{
    double tv[16];
#pragma unroll
    for (int n1 = 0; n1 < 8; n1++) {
        c64 xs = {V[n1].re * (1.0 / 1024.0), V[n1].im * (1.0 / 1024.0)};
        c64 u = cmul_nf_conj(xs, twist[n1]);
        tv[n1] = u.re;
        tv[8 + n1] = u.im;
    }
    double mn = __builtin_fabs(tv[0]);
#pragma unroll
    for (int e = 1; e < 16; e++) mn = __builtin_fmin(mn, __builtin_fabs(tv[e]));
    if (__all(mn >= 4503599627370496.0)) {
#pragma unroll
        for (int e = 0; e < 16; e++) t[e] = f64_bigint_to_torus(tv[e]);
    } else {
#pragma unroll
        for (int e = 0; e < 16; e++) t[e] = f64_round_to_torus(tv[e]);
    }
}

} // namespace spf
