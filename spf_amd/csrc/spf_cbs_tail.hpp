// spf_cbs_tail.hpp — the tail of the circuit bootstrap on gfx950:
//   mod_switch_trace_and_rotate  (ops/bootstrapping/circuit_bootstrapping.rs:260-298)
//     = per gadget level: undo-rotate, X^-i, shift by log2 N, homomorphic trace
//       (ops/automorphisms/mod.rs:53-85: log2 N rounds of automorphism + FFT-domain GLWE
//        keyswitch, ops/fft_ops.rs:457-495)
//   scheme_switch_fft            (ops/fft_ops.rs:225-279, 403-442)  GLEV -> GGSW-FFT
// Both kernels use the two-waves-per-ciphertext layout of the blind rotation (wave w owns the
// complex samples of parity w; one 4 KiB cross exchange per transform) and the same arithmetic
// (DAG-I transforms, AVX-512-order complex_mad, reference rounding).  A work unit is one
// (ciphertext, gadget level) pair; the automorphism / scheme-switch keys are small (2.1 MB /
// 0.5 MB), shared by every unit and read straight from L2 into registers.
#pragma once
#include "spf_kernels.hpp"

namespace spf {

struct PairCtx {
    char* mine;
    char* theirs;
    const c64* tab;
    const c64* twist; // tab + kTWOff + w*512 + lane
    const c64* wc;    // tab + kWCOff + 256*w + lane
    volatile uint32_t* flags;
    int lane, w, me, partner;
};

// Hand-over between the two waves of a unit: the flat-polled word per wave pair.  Unlike the blind rotation the
// four units of a workgroup share nothing (no key ring), so pair-local hand-overs let them drift apart and fill
// each other's waits; a workgroup s_barrier here (compile with -DSPF_TAIL_BARRIER) measured 1.8 % SLOWER on the
// whole circuit bootstrap (54.4 vs 53.5 ms per 4096) although a polled hand-over costs ≈ 2 000 cycles.
__device__ __forceinline__ void tail_sync(const PairCtx& c, uint32_t& seq)
{
#ifdef SPF_TAIL_BARRIER
    (void)seq;
    pair_barrier_w();
#else
    pair_barrier(c.flags, c.me, c.partner, seq);
#endif
}

// value select on the (wave-uniform) parity: written with scalar selects rather than branches on
// purpose — branching over struct copies makes hipcc select between *addresses* of register
// arrays, which pins them in scratch memory.
__device__ __forceinline__ c64 sel(bool pick_b, c64 a, c64 b)
{
    return {pick_b ? b.re : a.re, pick_b ? b.im : a.im};
}

// forward negacyclic transform of the ciphertext's 16 samples per lane pair: wave w brings its 8
// twisted samples V and leaves with its 8 bins X (bin = lane + 64(4w + (r&3)) + 512(r>>2)).
__device__ __forceinline__ void pair_forward(const PairCtx& c, uint32_t& seq, c64 (&V)[8], c64 (&X)[8])
{
    const bool odd = c.w != 0;
    tail_sync(c, seq); // partner is done reading my region
    fft512_single<+1>(V, c.mine, c.tab, c.lane);
    // radix-2 stage across the two waves: wave 0 finishes bins d < 4 (keeps E[0..3], needs O[0..3]),
    // wave 1 bins d >= 4 (keeps O[4..7], needs E[4..7])
#pragma unroll
    for (int i = 0; i < 4; i++) reinterpret_cast<c64*>(c.mine)[i * 64 + c.lane] = sel(odd, V[4 + i], V[i]);
    tail_sync(c, seq);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        c64 got = reinterpret_cast<const c64*>(c.theirs)[i * 64 + c.lane];
        c64 Ei = sel(odd, V[i], got);
        c64 Oi = sel(odd, got, V[4 + i]);
        c64 t = cmul_tw<+1>(Oi, c.wc[64 * i]);
        X[i] = cadd(Ei, t);
        X[i + 4] = csub(Ei, t);
    }
}

// inverse transform of 8 bins per wave back to the wave's 16 coefficients, as rounded-but-not-yet-
// reduced doubles: tv[n1] = coefficient (half 0, n1), tv[8+n1] = (half 1, n1)
__device__ __forceinline__ void pair_inverse(const PairCtx& c, uint32_t& seq, const c64 (&P)[8], double (&tv)[16])
{
    const bool odd = c.w != 0;
    c64 Ep[4], Op[4], V[8];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        Ep[i] = cadd(P[i], P[i + 4]);
        c64 dd = csub(P[i], P[i + 4]);
        Op[i] = cmul_tw<-1>(dd, c.wc[64 * i]);
    }
    tail_sync(c, seq);
    // wave 0 keeps E'[0..3] and needs E'[4..7]; wave 1 keeps O'[4..7] and needs O'[0..3]
#pragma unroll
    for (int i = 0; i < 4; i++) reinterpret_cast<c64*>(c.mine)[i * 64 + c.lane] = sel(odd, Op[i], Ep[i]);
    tail_sync(c, seq);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        c64 got = reinterpret_cast<const c64*>(c.theirs)[i * 64 + c.lane];
        V[i] = sel(odd, Ep[i], got);
        V[4 + i] = sel(odd, got, Op[i]);
    }
    tail_sync(c, seq);
    fft512_single<-1>(V, c.mine, c.tab, c.lane);
#pragma unroll
    for (int n1 = 0; n1 < 8; n1++) {
        c64 xs = {V[n1].re * (1.0 / 1024.0), V[n1].im * (1.0 / 1024.0)};
        c64 t = cmul_nf_conj(xs, c.twist[64 * n1]);
        tv[n1] = t.re;
        tv[8 + n1] = t.im;
    }
}

// round(), mod 2^64, `as i64` of 16 values (the two exact paths: spf_device.hpp)
__device__ __forceinline__ void tv_to_torus(const double (&tv)[16], uint64_t (&out)[16])
{
    double mn = __builtin_fabs(tv[0]);
#pragma unroll
    for (int e = 1; e < 16; e++) mn = __builtin_fmin(mn, __builtin_fabs(tv[e]));
    if (__all(mn >= 4503599627370496.0)) {
#pragma unroll
        for (int e = 0; e < 16; e++) out[e] = f64_bigint_to_torus(tv[e]);
    } else {
#pragma unroll
        for (int e = 0; e < 16; e++) out[e] = f64_round_to_torus(tv[e]);
    }
}

// complex_mad in the reference's AVX-512 order (math/simd/x86_64/avx512.rs:54-57)
__device__ __forceinline__ void mad8(c64 (&acc)[8], const c64 (&k)[8], const c64 (&X)[8])
{
#pragma unroll
    for (int r = 0; r < 8; r++) {
        double re = __builtin_fma(k[r].re, X[r].re, acc[r].re);
        double im = __builtin_fma(k[r].re, X[r].im, acc[r].im);
        acc[r].re = __builtin_fma(-k[r].im, X[r].im, re);
        acc[r].im = __builtin_fma(k[r].im, X[r].re, im);
    }
}

// acc += key_row * X over this wave's 8 bins, the key streamed 4 bins at a time so that only 16
// registers of key are live at once
__device__ __forceinline__ void mad_row(c64 (&acc)[8], const c64* row_w_lane, const c64 (&X)[8])
{
#pragma unroll
    for (int hh = 0; hh < 2; hh++) {
        c64 k[4];
#pragma unroll
        for (int i = 0; i < 4; i++) k[i] = row_w_lane[64 * i + 512 * hh];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int r = i + 4 * hh;
            double re = __builtin_fma(k[i].re, X[r].re, acc[r].re);
            double im = __builtin_fma(k[i].re, X[r].im, acc[r].im);
            acc[r].re = __builtin_fma(-k[i].im, X[r].im, re);
            acc[r].im = __builtin_fma(k[i].im, X[r].re, im);
        }
    }
}

// this wave's 8 bins of a 1024-bin row: index r -> bin lane + 64(4w + (r&3)) + 512(r>>2)
__device__ __forceinline__ void load_bins(c64 (&k)[8], const c64* row_w_lane)
{
#pragma unroll
    for (int r = 0; r < 8; r++) k[r] = row_w_lane[64 * (r & 3) + 512 * (r >> 2)];
}
__device__ __forceinline__ void store_bins(c64* row_w_lane, const c64 (&x)[8])
{
#pragma unroll
    for (int r = 0; r < 8; r++) row_w_lane[64 * (r & 3) + 512 * (r >> 2)] = x[r];
}

// round to the top L*LOGB bits (radix.rs:157-162); the state then yields digits one at a time
template <int L, int LOGB> __device__ __forceinline__ uint64_t radix_round_state(uint64_t x)
{
    constexpr int shift = 64 - L * LOGB;
    return (x >> shift) + ((x >> (shift - 1)) & 1);
}
// vector_next_decomp (scalar.rs:52-71): signed digit in [-B/2, B/2)
template <int LOGB> __device__ __forceinline__ int next_digit(uint64_t& s)
{
    uint32_t d = (uint32_t)s & ((1u << LOGB) - 1);
    s >>= LOGB;
    uint32_t carry = d >> (LOGB - 1);
    s += carry;
    return (int)d - (int)(carry << LOGB);
}

constexpr int kTailLds = kTableBytes + kWavesPerBlock * kWaveBufBytes + 64;
constexpr int kTraceLds = kTailLds + 8 * 8192; // + the parked mask half of each wave's accumulator

struct TraceArgs {
    const uint64_t* glwe_in; // B x 4096: lo-noise GLWE out of the bootstrap
    uint64_t* glev_out;      // B x cbs_count x 4096
    const c64* ak;           // [11][L][2][1024] automorphism keyswitch keys, FFT'd (k = 1: one row)
    const c64* tables;
    uint32_t units;          // B * cbs_count
    uint32_t cbs_count, cbs_radix_log;
};

template <int L, int LOGB> // trace radix: L digits of LOGB bits
__global__ __launch_bounds__(512, 2) void cbs_trace_kernel(TraceArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, cslot = wv >> 1;
    const int w = __builtin_amdgcn_readfirstlane(wv & 1);
    char* tile = smem + kTableBytes + cslot * kWaveBufBytes;
    volatile uint32_t* flags = reinterpret_cast<volatile uint32_t*>(smem + kTableBytes + kWavesPerBlock * kWaveBufBytes);
    PairCtx pc;
    pc.mine = tile + w * 8192; pc.theirs = tile + (w ^ 1) * 8192;
    pc.tab = reinterpret_cast<const c64*>(smem);
    pc.twist = pc.tab + kTWOff + w * 512 + lane; pc.wc = pc.tab + kWCOff + 256 * w + lane;
    pc.flags = flags; pc.lane = lane; pc.w = w;
    pc.me = __builtin_amdgcn_readfirstlane(wv); pc.partner = pc.me ^ 1;
    uint32_t seq = 0;
    {
        const double2* src = reinterpret_cast<const double2*>(a.tables);
        double2* dst = reinterpret_cast<double2*>(smem);
        for (int i = tid; i < kTableEntries; i += 512) dst[i] = src[i];
        if (tid < 8) flags[tid] = 0;
    }
    const uint32_t unit_raw = blockIdx.x * kWavesPerBlock + cslot;
    const bool owns_output = unit_raw < a.units;
    const uint32_t unit = owns_output ? unit_raw : a.units - 1;
    const uint32_t ct = unit / a.cbs_count, lvl = unit % a.cbs_count;
    auto coef2 = [&](int e) -> int { return (e >> 3) * 1024 + (e & 7) * 128 + 2 * lane + w; };

    // x = shr_round( (glwe with body coefficients 0..lvl un-rotated) * X^-lvl , log2 N ).
    // The accumulator's body half lives in registers (acc_b); its mask half, touched only twice
    // per round, is parked in this wave's private 8 KiB of the LDS the key ring uses elsewhere
    // (park[e*64 + lane]): that keeps the kernel inside 256 VGPRs without scratch.
    uint64_t* park = reinterpret_cast<uint64_t*>(smem + kTableBytes + kWavesPerBlock * kWaveBufBytes + 64 + wv * 8192);
    uint64_t acc_b[16];
    {
        const uint64_t* g = a.glwe_in + (size_t)ct * 2 * kN;
#pragma unroll
        for (int p = 0; p < 2; p++)
#pragma unroll
            for (int e = 0; e < 16; e++) {
                uint32_t idx = (uint32_t)coef2(e) + lvl; // X^-lvl: out[c] = +-in[c + lvl]
                uint32_t src = idx & (kN - 1);
                uint64_t v = g[p * kN + src];
                if (p == 1 && src <= lvl) // glwe_rotated.b[t] += encode(1, cbs_radix_log*(t+1)+1), t <= lvl
                    v += (uint64_t)1 << (64 - (a.cbs_radix_log * (src + 1) + 1));
                v = ((idx >> 11) & 1) ? (uint64_t)0 - v : v;
                v = (v >> 11) + ((v >> 10) & 1); // glwe_mod_switch_and_expand_pow_2
                if (p == 0) park[e * 64 + lane] = v; else acc_b[e] = v;
            }
    }
    __syncthreads();

    uint64_t* stage_mine = reinterpret_cast<uint64_t*>(pc.mine);
#pragma unroll 1
    for (uint32_t it = 1; it <= 11; it++) {
        // automorphism X -> X^kk, kk = N/2^(it-1) + 1 (ops/polynomial/mod.rs:62-84): out[d] = +-in[c],
        // c*kk = d mod 2N  <=>  c' = d*kk^-1 mod 2N, c = c' mod N, sign = c' >= N
        const uint32_t kk = (kN >> (it - 1)) + 1;
        uint32_t kinv = kk; // Newton iteration for the inverse of an odd number mod 2^12
        kinv *= 2 - kk * kinv; kinv *= 2 - kk * kinv; kinv *= 2 - kk * kinv; kinv *= 2 - kk * kinv;
        kinv &= 2 * kN - 1;
        uint64_t st[16];
#pragma unroll
        for (int p = 0; p < 2; p++) {
            tail_sync(pc, seq);
#pragma unroll
            for (int e = 0; e < 16; e++)
                stage_mine[(e >> 3) * 512 + (e & 7) * 64 + lane] = p == 0 ? park[e * 64 + lane] : acc_b[e];
            tail_sync(pc, seq);
#pragma unroll
            for (int e = 0; e < 16; e++) {
                uint32_t cp = ((uint32_t)coef2(e) * kinv) & (2 * kN - 1);
                uint32_t src = cp & (kN - 1);
                uint64_t v = reinterpret_cast<const uint64_t*>(tile + (src & 1) * 8192)[src >> 1];
                v = (cp >> 11) ? (uint64_t)0 - v : v;
                if (p == 0) st[e] = radix_round_state<L, LOGB>(v); // keyswitch decomposes the mask
                else acc_b[e] += v;                                 // out.b += trivial(b_k) ...
            }
        }
        c64 prod[2][8];
#pragma unroll
        for (int q = 0; q < 2; q++)
#pragma unroll
            for (int r = 0; r < 8; r++) prod[q][r] = {0.0, 0.0};
        const c64* key = a.ak + (size_t)(it - 1) * (L * 2 * kHalf) + 256 * w + lane;
#pragma unroll 1
        for (int j = 0; j < L; j++) {
            const c64* row = key + (size_t)((L - 1 - j) * 2) * kHalf; // GLEV rows in reverse
            c64 V[8], X[8];
#pragma unroll
            for (int n1 = 0; n1 < 8; n1++) {
                int dre = next_digit<LOGB>(st[n1]);
                int dim = next_digit<LOGB>(st[8 + n1]);
                V[n1] = cmul_nf({(double)dre, (double)dim}, pc.twist[64 * n1]);
            }
            pair_forward(pc, seq, V, X);
            // key rows are shared by every unit (L2-resident); the SIMD partner covers the trip
            mad_row(prod[0], row, X);
            mad_row(prod[1], row + kHalf, X);
        }
        // ... - sum_i <decomp(a_i), glev_i>  (fft_ops.rs:489-494)
#pragma unroll
        for (int q = 0; q < 2; q++) {
            double tv[16];
            uint64_t s[16];
            pair_inverse(pc, seq, prod[q], tv);
            tv_to_torus(tv, s);
#pragma unroll
            for (int e = 0; e < 16; e++) {
                if (q == 0) park[e * 64 + lane] -= s[e]; else acc_b[e] -= s[e];
            }
        }
    }
    if (!owns_output) return;
    uint64_t* out = a.glev_out + (size_t)unit * 2 * kN;
#pragma unroll
    for (int e = 0; e < 16; e++) {
        out[coef2(e)] = park[e * 64 + lane];
        out[kN + coef2(e)] = acc_b[e];
    }
}

struct SchemeSwitchArgs {
    const uint64_t* glev; // units x 4096
    c64* ggsw_out;        // B x [2][cbs_count][2][1024]
    const c64* ssk;       // [L][2][1024] (k = 1: the single pair s*s), FFT'd
    const c64* tables;
    uint32_t units, cbs_count;
};

template <int L, int LOGB> // scheme-switch radix
__global__ __launch_bounds__(512, 2) void scheme_switch_kernel(SchemeSwitchArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, cslot = wv >> 1;
    const int w = __builtin_amdgcn_readfirstlane(wv & 1);
    char* tile = smem + kTableBytes + cslot * kWaveBufBytes;
    volatile uint32_t* flags = reinterpret_cast<volatile uint32_t*>(smem + kTableBytes + kWavesPerBlock * kWaveBufBytes);
    PairCtx pc;
    pc.mine = tile + w * 8192; pc.theirs = tile + (w ^ 1) * 8192;
    pc.tab = reinterpret_cast<const c64*>(smem);
    pc.twist = pc.tab + kTWOff + w * 512 + lane; pc.wc = pc.tab + kWCOff + 256 * w + lane;
    pc.flags = flags; pc.lane = lane; pc.w = w;
    pc.me = __builtin_amdgcn_readfirstlane(wv); pc.partner = pc.me ^ 1;
    uint32_t seq = 0;
    {
        const double2* src = reinterpret_cast<const double2*>(a.tables);
        double2* dst = reinterpret_cast<double2*>(smem);
        for (int i = tid; i < kTableEntries; i += 512) dst[i] = src[i];
        if (tid < 8) flags[tid] = 0;
    }
    const uint32_t unit_raw = blockIdx.x * kWavesPerBlock + cslot;
    const bool owns_output = unit_raw < a.units;
    const uint32_t unit = owns_output ? unit_raw : a.units - 1;
    const uint32_t ct = unit / a.cbs_count, lvl = unit % a.cbs_count;
    auto coef2 = [&](int e) -> int { return (e >> 3) * 1024 + (e & 7) * 128 + 2 * lane + w; };
    const uint64_t* x = a.glev + (size_t)unit * 2 * kN;
    // GGSW-FFT [row][level][poly][bin]; this wave's bins start at 256*w + lane
    c64* out_row0 = a.ggsw_out + (size_t)ct * (2 * a.cbs_count * 2 * kHalf) + (size_t)(lvl * 2) * kHalf + 256 * w + lane;
    c64* out_row1 = out_row0 + (size_t)a.cbs_count * 2 * kHalf;
    __syncthreads();

    // PolynomialRef::fft of a full-range polynomial: u64 -> i64 -> f64 (round to nearest even)
    auto full_fft = [&](int p, c64 (&X)[8]) {
        c64 V[8];
#pragma unroll
        for (int n1 = 0; n1 < 8; n1++) {
            double re = (double)(long long)x[p * kN + coef2(n1)];
            double im = (double)(long long)x[p * kN + coef2(8 + n1)];
            V[n1] = cmul_nf({re, im}, pc.twist[64 * n1]);
        }
        pair_forward(pc, seq, V, X);
    };
    c64 prod[2][8];
    {
        // last row (j == k): plain FFT of both polynomials (fft_ops.rs:243-247)
        c64 Xa[8];
        full_fft(0, Xa);
        if (owns_output) store_bins(out_row1, Xa);
        full_fft(1, prod[0]); // also the start of row 0: y.a[0] = FFT(x.b) (fft_ops.rs:225-241)
        if (owns_output) store_bins(out_row1 + kHalf, prod[0]);
#pragma unroll
        for (int r = 0; r < 8; r++) prod[1][r] = {0.0, 0.0};
    }
    uint64_t st[16];
#pragma unroll
    for (int e = 0; e < 16; e++) st[e] = radix_round_state<L, LOGB>(x[coef2(e)]);
    const c64* key = a.ssk + 256 * w + lane;
#pragma unroll 1
    for (int j = 0; j < L; j++) {
        const c64* row = key + (size_t)((L - 1 - j) * 2) * kHalf;
        c64 V[8], X[8];
#pragma unroll
        for (int n1 = 0; n1 < 8; n1++) {
            int dre = next_digit<LOGB>(st[n1]);
            int dim = next_digit<LOGB>(st[8 + n1]);
            V[n1] = cmul_nf({(double)dre, (double)dim}, pc.twist[64 * n1]);
        }
        pair_forward(pc, seq, V, X);
        mad_row(prod[0], row, X);
        mad_row(prod[1], row + kHalf, X);
    }
    if (!owns_output) return;
    store_bins(out_row0, prod[0]);
    store_bins(out_row0 + kHalf, prod[1]);
}

} // namespace spf
