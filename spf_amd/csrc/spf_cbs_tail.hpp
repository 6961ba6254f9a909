// spf_cbs_tail.hpp — the tail of the circuit bootstrap on gfx950:
//   mod_switch_trace_and_rotate  (ops/bootstrapping/circuit_bootstrapping.rs:260-298)
//     = per gadget level: undo-rotate, X^-i, shift by log2 N, homomorphic trace
//       (ops/automorphisms/mod.rs:53-85: log2 N rounds of automorphism + FFT-domain GLWE
//        keyswitch, ops/fft_ops.rs:457-495)
//   scheme_switch_fft            (ops/fft_ops.rs:225-279, 403-442)  GLEV -> GGSW-FFT
// Both kernels use the two-waves-per-ciphertext layout of the blind rotation (wave w owns the complex samples of
// parity w) and the same arithmetic (DAG-I transforms, AVX-512-order complex_mad, reference rounding), on the schedule
// of the throughput blind rotation: transform pairs, one body per sample parity, bare workgroup barriers, key rows
// through a 64 KiB LDS ring filled by LDS-DMA.  A work unit is one (ciphertext, gadget level) pair, four units per
// 512-thread workgroup, one workgroup per CU.
#pragma once
#include "spf_kernels.hpp"

// The transform pair of each tail kernel (all variants give the same words).  r05 A/B, ms per 4096, two runs each
// (profiles/r05_experiments_other_kernels.md): trace 4.52 / 4.57 / 4.61 / 4.72 and scheme switch 1.149 / 1.111 / 1.126 / 1.130 with
// fft512_pair1 / pair1t / pair1ts / pair1ts2 — the trace kernel has no registers to hold a pass's twiddles across the pair
// (44 B of scratch already), the scheme switch has.
#ifndef SPF_TRACE_PAIR
#define SPF_TRACE_PAIR fft512_pair1
#endif
#ifndef SPF_SS_PAIR
#define SPF_SS_PAIR fft512_pair1t // pass twiddles requested early, shared by the pair (r04: 1.16 -> 1.12 ms)
#endif

namespace spf {

// this wave's 8 bins of a 1024-bin row: index r -> bin lane + 64(4w + (r&3)) + 512(r>>2)
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


struct TraceArgs {
    const uint64_t* glwe_in; // B x 4096: lo-noise GLWE out of the bootstrap
    uint64_t* glev_out;      // B x cbs_count x 4096
    const c64* ak;           // [11][L][2][1024] automorphism keyswitch keys, FFT'd (k = 1: one row)
    const c64* tables;
    uint32_t units;          // B * cbs_count
    uint32_t cbs_count, cbs_radix_log;
    uint64_t* stamps;        // diagnostic builds (-DSPF_STAMPS): [workgroup][wave][16] cycle sums per phase, else null
};

// ---------------------------------------------------------------------------------------------------------------
// cbs_trace_kernel: the trace on the schedule of the throughput blind rotation (r03; r02's kernel — one digit at a time,
// flat-polled pair hand-overs, key rows from L2 into registers, mask half parked in LDS — took 5.86 ms per 4096 against 4.49).
//
// One automorphism round = one GLWE keyswitch in the FFT domain (ops/fft_ops.rs:457-495): the six 7-bit digits of
// the automorphed mask go through THREE transform pairs (`fft512_pair1`, both digits of a pair on one 8 KiB image,
// exchange 2 of the second in registers), the two output polynomials come back as one inverse pair, the body is
// compiled once per sample parity (no value selects), every hand-over is a bare workgroup `s_barrier`, and the
// automorphism key rides a 64 KiB LDS ring: the rows of a digit pair — levels (L-2-2m, L-1-2m), 2 x 2 x 16 KiB,
// contiguous in the reference layout [round][level][poly][bin] — are brought in by LDS-DMA one pair ahead and waited
// for at the cross-exchange barrier, exactly as the bootstrapping key in `blind_rotate2p_kernel`.  That is what lets a
// workgroup barrier replace the flat-polled pair hand-overs (≈ 2 000 cycles each, 22 per round): with the key rows
// coming from L2 straight into registers (r02) every wave arrived at a barrier with its own L2 latency, and a barrier
// tied to eight such waves measured slower than polling.
// The automorphism X -> X^k (k odd) maps each coefficient parity class onto itself, so the gather of a wave reads only
// what the wave itself staged: staging and gather need no hand-over at all.  Eight barriers per round.
// The mask half of the accumulator stays in registers (r02 parked it in the LDS the ring now occupies): after the
// gather only the REMAINING digit state is kept, 28 bits = one register per coefficient once the first pair's digits
// are out.  Same arithmetic in the same order as r02's kernel (and the oracle): same words.
constexpr int kTraceLds = kTableBytes + kWavesPerBlock * kWaveBufBytes + 2 * kBskSlotBytes;

template <int L, int LOGB, int W>
__device__ __forceinline__ void cbs_trace_body(const TraceArgs& a, char* smem)
{
    static_assert(L == 6 && LOGB == 7, "three digit pairs; the state after the first pair fits 32 bits");
#ifdef SPF_STAMPS
    uint64_t st_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t st_prev = __builtin_amdgcn_s_memtime();
#define STAMPT(i) do { uint64_t t_ = __builtin_amdgcn_s_memtime(); st_acc[i] += t_ - st_prev; st_prev = t_; } while (0)
#else
#define STAMPT(i) do { } while (0)
#endif
    constexpr int XP = 2;
    constexpr int NT = 512;
    c64* tab = reinterpret_cast<c64*>(smem);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cslot = wv >> 1;
    constexpr int w = W;
    char* tile = smem + kTableBytes + cslot * kWaveBufBytes;
    char* mine = tile + w * 8192;
    char* theirs = tile + (w ^ 1) * 8192;
    char* ring = smem + kTableBytes + kWavesPerBlock * kWaveBufBytes;
    {
        const double2* src = reinterpret_cast<const double2*>(a.tables);
        double2* dst = reinterpret_cast<double2*>(smem);
        for (int i = tid; i < kTableEntries; i += NT) dst[i] = src[i];
    }
    const uint32_t unit_raw = blockIdx.x * kWavesPerBlock + cslot;
    const bool owns_output = unit_raw < a.units;
    const uint32_t unit = owns_output ? unit_raw : a.units - 1;
    const uint32_t ct = unit / a.cbs_count, lvl = unit % a.cbs_count;
    auto coef2 = [&](int e) -> int { return (e >> 3) * 1024 + (e & 7) * 128 + 2 * lane + w; };

    // key chunk c = 3 (round - 1) + m: the 64 KiB [level L-2-2m | level L-1-2m] of that round, copied as it lies: the
    // first digit of the pair (j = 2m, level L-1-2m) is the ring's second half
    const uint32_t dma_voff = (uint32_t)tid * 16u;
    const uint32_t dma_dst = lds_address(ring) + wv * 1024;
    auto ring_dma = [&](uint32_t chunk) {
        const uint32_t rnd = chunk / 3, m = chunk % 3;
        const char* src = reinterpret_cast<const char*>(a.ak) +
                          (size_t)__builtin_amdgcn_readfirstlane(rnd * L + (L - 2 - 2 * m)) * kBskSlotBytes;
#pragma unroll
        for (int k = 0; k < 2 * kBskSlotBytes / (NT * 16); k++)
            lds_dma_piece(src + k * NT * 16, dma_voff, dma_dst + k * NT * 16);
    };
    ring_dma(0);

    // x = shr_round( (glwe with body coefficients 0..lvl un-rotated) * X^-lvl , log2 N )
    uint64_t accm[16], accb[16];
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
                if (p == 0) accm[e] = v; else accb[e] = v;
            }
    }
    __syncthreads();
    // issue priority swapped inside the long stretches, as in blind_rotate2p_body (there −5 %; here 4.46 -> 4.40 ms):
    // the younger waves lead from the inverse cross exchange to the first twist and through the last forward pair
    const uint32_t is_young = __builtin_amdgcn_readfirstlane(wv >= kWavesPerBlock ? 1u : 0u);
    uint32_t opaque_zero;
    asm volatile("s_mov_b32 %0, 0" : "=s"(opaque_zero));
    auto rendezvous = [&]() { // bare s_barrier; the never-repeating loop gives each phase its own basic block
        do {
            pair_barrier_w();
        } while (opaque_zero != 0);
    };

    uint64_t* stage_mine = reinterpret_cast<uint64_t*>(mine);
    const c64* twist = tab + kTWOff + w * 512 + lane;
    const c64* wc = tab + kWCOff + 256 * w + lane;
    uint32_t chunk = 0;
    constexpr uint32_t total_chunks = 11 * 3;
    // The accumulator (64 registers) is idle from the gather to the end of a round; beside the digit state, one
    // transform pair and the frequency-domain product it does not fit 256 registers, and hipcc's own spilling put
    // 492 B per lane in scratch with reloads in the middle of the transforms.  Its BODY half is parked explicitly
    // instead, in the unit's own 32 KiB of the OUTPUT buffer (written for real only at the very end), fully
    // coalesced: word (w * 16 + e) * 64 + lane.  A padding unit of a ragged last workgroup (a duplicate of the last
    // real unit) does not store there; what it reads back only feeds its own discarded arithmetic.
    gu64_ptr park = global_view(a.glev_out + (size_t)unit * 2 * kN) + (size_t)(w * 16) * 64 + lane;

#pragma unroll 1
    for (uint32_t it = 1; it <= 11; it++) {
        // automorphism X -> X^kk, kk = N/2^(it-1) + 1 (ops/polynomial/mod.rs:62-84): out[d] = +-in[c], c kk = d mod 2N
        // <=> c' = d kk^-1 mod 2N, c = c' mod N, sign = c' >= N.  kk^-1 is odd: c' has the parity of d, so every source
        // lies in this wave's own staging region.
        const uint32_t kk = (kN >> (it - 1)) + 1;
        uint32_t kinv = kk; // Newton iteration for the inverse of an odd number mod 2^12
        kinv *= 2 - kk * kinv; kinv *= 2 - kk * kinv; kinv *= 2 - kk * kinv; kinv *= 2 - kk * kinv;
        kinv &= 2 * kN - 1;
        const uint32_t cp0 = (uint32_t)(2 * lane + w) * kinv;
        uint32_t st[16];  // digit state of the mask coefficients once digits 0 and 1 are out (28 bits)
        uint32_t d01[16]; // digits 0 and 1, sign-extended 16-bit halves
#pragma unroll
        for (int p = 0; p < 2; p++) {
#pragma unroll
            for (int e = 0; e < 16; e++) stage_mine[(e >> 3) * 512 + (e & 7) * 64 + lane] = p == 0 ? accm[e] : accb[e];
            wave_lds_fence(); // own writes, own reads: in-order LDS needs no barrier
            uint64_t gin[16];
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const uint32_t cp = cp0 + (uint32_t)((e >> 3) * 1024 + (e & 7) * 128) * kinv;
                gin[e] = *reinterpret_cast<const uint64_t*>(mine + ((cp << 2) & 0x1FF8u));
            }
            compiler_fence();
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const uint32_t cp = cp0 + (uint32_t)((e >> 3) * 1024 + (e & 7) * 128) * kinv;
                const uint64_t sgn = (uint64_t)((int64_t)((uint64_t)cp << 52) >> 63); // bit 11 of c', spread
                const uint64_t v = (gin[e] ^ sgn) - sgn;
                if (p == 0) { // keyswitch decomposes the mask
                    uint64_t s64 = radix_round_state<L, LOGB>(v);
                    const int d0 = next_digit<LOGB>(s64);
                    const int d1 = next_digit<LOGB>(s64);
                    d01[e] = ((uint32_t)d0 & 0xFFFFu) | ((uint32_t)d1 << 16);
                    st[e] = (uint32_t)s64;
                } else {
                    accb[e] += v; // out.b += trivial(b_k) ...
                }
            }
#ifndef SPF_ABL_NO_PARK // (timing-only ablation: wrong results — what does the parked half's traffic cost?  profiles/r05_experiments_other_kernels.md)
            if (p == 1 && owns_output) {
#pragma unroll
                for (int e = 0; e < 16; e++) park[(size_t)e * 64] = accb[e];
            }
#endif
            wave_lds_fence(); // gathered: the region may be overwritten (next staging / the exchange image)
        }

        STAMPT(0);
        c64 prod[2][8];
#pragma unroll
        for (int m = 0; m < 3; m++, chunk++) {
            c64 VV[2][8];
#pragma unroll
            for (int n1 = 0; n1 < 8; n1++) {
                int dre[2], dim[2];
                if (m == 0) {
                    dre[0] = (int)(int16_t)(d01[n1] & 0xFFFFu); dre[1] = (int)d01[n1] >> 16;
                    dim[0] = (int)(int16_t)(d01[8 + n1] & 0xFFFFu); dim[1] = (int)d01[8 + n1] >> 16;
                } else {
#pragma unroll
                    for (int jj = 0; jj < 2; jj++) {
                        uint64_t sr = st[n1], si = st[8 + n1];
                        dre[jj] = next_digit<LOGB>(sr);
                        dim[jj] = next_digit<LOGB>(si);
                        st[n1] = (uint32_t)sr;
                        st[8 + n1] = (uint32_t)si;
                    }
                }
                const c64 tw = twist[64 * n1];
                VV[0][n1] = cmul_nf({(double)dre[0], (double)dim[0]}, tw);
                VV[1][n1] = cmul_nf({(double)dre[1], (double)dim[1]}, tw);
            }
            // the ring is free since the barrier behind the previous MADs: rows of this pair (those of a round's first
            // pair were requested ahead of the previous round's inverse transforms)
            STAMPT(1);
            if (m == 0) young_prio<0>(is_young);
            if (m == 2) young_prio<1>(is_young);
            if (m > 0) ring_dma(chunk);
            SPF_TRACE_PAIR<+1, XP>(VV[0], VV[1], mine, tab, lane);
            STAMPT(2);
            // radix-2 stage across the two waves, both digits in one exchange
            if constexpr (w == 0) {
#pragma unroll
                for (int j = 0; j < 2; j++)
#pragma unroll
                    for (int i = 0; i < 4; i++) reinterpret_cast<c64*>(mine)[(j * 4 + i) * 64 + lane] = VV[j][4 + i];
            } else {
#pragma unroll
                for (int j = 0; j < 2; j++)
#pragma unroll
                    for (int i = 0; i < 4; i++) reinterpret_cast<c64*>(mine)[(j * 4 + i) * 64 + lane] = VV[j][i];
            }
            if (m == 2) young_prio<0>(is_young);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // my share of the key rows has landed
            __syncthreads();
            STAMPT(3);
            if constexpr (w == 0) {
#pragma unroll
                for (int j = 0; j < 2; j++)
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const c64 in = reinterpret_cast<const c64*>(theirs)[(j * 4 + i) * 64 + lane];
                        const c64 t = cmul_tw<+1>(in, wc[64 * i]);
                        const c64 Ei = VV[j][i];
                        VV[j][i] = cadd(Ei, t);
                        VV[j][i + 4] = csub(Ei, t);
                    }
            } else {
#pragma unroll
                for (int j = 0; j < 2; j++)
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const c64 Ei = reinterpret_cast<const c64*>(theirs)[(j * 4 + i) * 64 + lane];
                        const c64 t = cmul_tw<+1>(VV[j][4 + i], wc[64 * i]);
                        VV[j][i] = cadd(Ei, t);
                        VV[j][i + 4] = csub(Ei, t);
                    }
            }
            if (m == 2) {
                // the parked body half comes back under the last multiply-accumulate and the inverse cross exchange
#ifndef SPF_ABL_NO_PARK
#pragma unroll
                for (int e = 0; e < 16; e++) accb[e] = park[(size_t)e * 64];
#else
#pragma unroll
                for (int e = 0; e < 16; e++) accb[e] = (uint64_t)st[e] * 0x9E3779B97F4A7C15ull; // (something the compiler cannot fold)
#endif
            }
            // ... - sum_j <digit_j(a), glev row L-1-j>, digits in order, both output polynomials (fft_ops.rs:489-494)
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const c64* row = reinterpret_cast<const c64*>(ring + (1 - j) * kBskSlotBytes) + 256 * w + lane;
                c64 kb[2][2];
                auto key2 = [&](int grp, c64 (&dst)[2]) {
#pragma unroll
                    for (int i = 0; i < 2; i++) {
                        const int r = (grp * 2 + i) & 7, q = grp >> 2;
                        dst[i] = row[q * kHalf + 64 * (r & 3) + 512 * (r >> 2)];
                    }
                };
                key2(0, kb[0]);
#pragma unroll
                for (int grp = 0; grp < 8; grp++) {
                    if (grp + 1 < 8) key2(grp + 1, kb[(grp + 1) % 2]);
                    compiler_fence();
#pragma unroll
                    for (int i = 0; i < 2; i++) {
                        const int r = (grp * 2 + i) & 7, q = grp >> 2;
                        const c64 k = kb[grp % 2][i];
                        const bool first = m == 0 && j == 0;
                        double re = __builtin_fma(k.re, VV[j][r].re, first ? 0.0 : prod[q][r].re);
                        double im = __builtin_fma(k.re, VV[j][r].im, first ? 0.0 : prod[q][r].im);
                        prod[q][r].re = __builtin_fma(-k.im, VV[j][r].im, re);
                        prod[q][r].im = __builtin_fma(k.im, VV[j][r].re, im);
                    }
                }
            }
            STAMPT(4);
            __syncthreads(); // every wave is done with the ring and with its partner's cross data
            STAMPT(5);
        }

        // ---- back to the torus, both output polynomials together
        c64 WW[2][8];
#pragma unroll
        for (int q = 0; q < 2; q++)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const c64 wci = wc[64 * i];
                WW[q][i] = cadd(prod[q][i], prod[q][i + 4]);
                WW[q][4 + i] = cmul_tw<-1>(csub(prod[q][i], prod[q][i + 4]), wci);
            }
        // The inverse cross exchange goes through the KEY RING (free between the barrier behind the last MADs and the next
        // refill): wave v writes its outgoing half into slot v (8 KiB), reads slot v^1 behind ONE barrier and then refills
        // exactly that slot with its 8 KiB of the next key chunk — the slot's only reader is the wave that overwrites it,
        // in program order, so no second barrier ("cross reads retired") is needed; the transforms run in the tile.
        {
            c64* slot_mine = reinterpret_cast<c64*>(ring + wv * 8192);
            const c64* slot_theirs = reinterpret_cast<const c64*>(ring + (wv ^ 1) * 8192);
            if constexpr (w == 0) {
#pragma unroll
                for (int q = 0; q < 2; q++)
#pragma unroll
                    for (int i = 0; i < 4; i++) slot_mine[(q * 4 + i) * 64 + lane] = WW[q][4 + i];
            } else {
#pragma unroll
                for (int q = 0; q < 2; q++)
#pragma unroll
                    for (int i = 0; i < 4; i++) slot_mine[(q * 4 + i) * 64 + lane] = WW[q][i];
            }
            rendezvous();
            if constexpr (w == 0) {
#pragma unroll
                for (int q = 0; q < 2; q++)
#pragma unroll
                    for (int i = 0; i < 4; i++) WW[q][4 + i] = slot_theirs[(q * 4 + i) * 64 + lane];
            } else {
#pragma unroll
                for (int q = 0; q < 2; q++)
#pragma unroll
                    for (int i = 0; i < 4; i++) WW[q][i] = slot_theirs[(q * 4 + i) * 64 + lane];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // the slot's contents are in registers
            STAMPT(6);
            young_prio<1>(is_young);
        }
        // the parked accumulator half has landed BEFORE the next key rows are requested: vmcnt counts in order, a wait for
        // it behind the request would wait for the rows as well
#ifndef SPF_TRACE_LATE_PARK
#pragma unroll
        for (int e = 0; e < 16; e++) asm volatile("" : "+v"(accb[e]));
#endif
        if (chunk < total_chunks) { // rows of the next round's first digit pair, this wave's share = the slot it just read
            const uint32_t rnd = chunk / 3;
            const char* src = reinterpret_cast<const char*>(a.ak) +
                              (size_t)__builtin_amdgcn_readfirstlane(rnd * L + (L - 2)) * kBskSlotBytes + (wv ^ 1) * 8192;
            const uint32_t lane16 = (uint32_t)lane * 16u;
            const uint32_t dst = lds_address(ring) + (wv ^ 1) * 8192;
#pragma unroll
            for (int k = 0; k < 8; k++) lds_dma_piece(src + k * 1024, lane16, dst + k * 1024);
        }
        STAMPT(7);
        SPF_TRACE_PAIR<-1, XP>(WW[0], WW[1], mine, tab, lane);
        STAMPT(8);
        {
            uint64_t t[16];
            untwist_to_torus_bits<true>(WW[0], twist, t);
#pragma unroll
            for (int e = 0; e < 16; e++) accm[e] -= t[e];
            untwist_to_torus_bits<true>(WW[1], twist, t);
#pragma unroll
            for (int e = 0; e < 16; e++) accb[e] -= t[e];
        }
        STAMPT(9);
    }
#ifdef SPF_STAMPS
    if (a.stamps && lane == 0) {
#pragma unroll
        for (int i = 0; i < 12; i++) a.stamps[((size_t)blockIdx.x * 8 + wv) * 16 + i] = st_acc[i];
    }
#endif
#undef STAMPT
    if (!owns_output) return;
    uint64_t* out = a.glev_out + (size_t)unit * 2 * kN;
#pragma unroll
    for (int e = 0; e < 16; e++) {
        out[coef2(e)] = accm[e];
        out[kN + coef2(e)] = accb[e];
    }
}

template <int L, int LOGB>
__global__ __launch_bounds__(512, 2) void cbs_trace_kernel(TraceArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) & 1) cbs_trace_body<L, LOGB, 1>(a, smem);
    else cbs_trace_body<L, LOGB, 0>(a, smem);
}

struct SchemeSwitchArgs {
    const uint64_t* glev; // units x 4096
    c64* ggsw_out;        // B x [2][cbs_count][2][1024]
    const c64* ssk;       // [L][2][1024] (k = 1: the single pair s*s), FFT'd
    const c64* tables;
    uint32_t units, cbs_count;
};

// ---------------------------------------------------------------------------------------------------------------
// scheme_switch_kernel: scheme_switch_fft on the same schedule as cbs_trace_kernel (r03; r02 ran one transform at a time with
// flat-polled pair hand-overs and key rows from L2 into registers: 1.035 ms per 4096 against 1.00) — the two full-range
// transforms as one pair, the fifteen 3-bit digits of the mask as seven pairs and a half pair, body per sample parity,
// bare workgroup barriers, the scheme-switch key through the 64 KiB LDS ring (levels (L-2-j, L-1-j) of a digit pair are
// contiguous in [level][poly][bin]).  Same arithmetic in the same order as scheme_switch_kernel: same words.
template <int L, int LOGB, int W>
__device__ __forceinline__ void scheme_switch_body(const SchemeSwitchArgs& a, char* smem)
{
    static_assert(L == 15 && LOGB == 3, "seven digit pairs and one single digit");
    constexpr int XP = 2;
    constexpr int NT = 512;
    constexpr int PAIRS = (L + 1) / 2;
    c64* tab = reinterpret_cast<c64*>(smem);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cslot = wv >> 1;
    constexpr int w = W;
    char* tile = smem + kTableBytes + cslot * kWaveBufBytes;
    char* mine = tile + w * 8192;
    char* theirs = tile + (w ^ 1) * 8192;
    char* ring = smem + kTableBytes + kWavesPerBlock * kWaveBufBytes;
    {
        const double2* src = reinterpret_cast<const double2*>(a.tables);
        double2* dst = reinterpret_cast<double2*>(smem);
        for (int i = tid; i < kTableEntries; i += NT) dst[i] = src[i];
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

    // key chunk c: the rows of digits 2c and 2c+1 = levels (L-2-2c, L-1-2c), 64 KiB as they lie; the first digit of a
    // pair (level L-1-2c) is the ring's second half.  The last chunk holds the single digit L-1 (level 0): second half only.
    const uint32_t dma_voff = (uint32_t)tid * 16u;
    const uint32_t dma_dst = lds_address(ring) + wv * 1024;
    auto ring_dma = [&](int c) {
        if (2 * c + 1 < L) {
            const char* src = reinterpret_cast<const char*>(a.ssk) + (size_t)(L - 2 - 2 * c) * kBskSlotBytes;
#pragma unroll
            for (int k = 0; k < 2 * kBskSlotBytes / (NT * 16); k++)
                lds_dma_piece(src + k * NT * 16, dma_voff, dma_dst + k * NT * 16);
        } else {
            const char* src = reinterpret_cast<const char*>(a.ssk);
#pragma unroll
            for (int k = 0; k < kBskSlotBytes / (NT * 16); k++)
                lds_dma_piece(src + k * NT * 16, dma_voff, dma_dst + kBskSlotBytes + k * NT * 16);
        }
    };
    ring_dma(0);
    __syncthreads();

    const c64* twist = tab + kTWOff + w * 512 + lane;
    const c64* wc = tab + kWCOff + 256 * w + lane;
    // forward transform pair with the radix-2 stage across the two waves; leaves this wave's bins in VV
    auto forward_pair = [&](c64 (&VV)[2][8]) {
        SPF_SS_PAIR<+1, XP>(VV[0], VV[1], mine, tab, lane);
        if constexpr (w == 0) {
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int i = 0; i < 4; i++) reinterpret_cast<c64*>(mine)[(j * 4 + i) * 64 + lane] = VV[j][4 + i];
        } else {
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int i = 0; i < 4; i++) reinterpret_cast<c64*>(mine)[(j * 4 + i) * 64 + lane] = VV[j][i];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // my share of the key rows has landed
        __syncthreads();
        if constexpr (w == 0) {
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const c64 in = reinterpret_cast<const c64*>(theirs)[(j * 4 + i) * 64 + lane];
                    const c64 t = cmul_tw<+1>(in, wc[64 * i]);
                    const c64 Ei = VV[j][i];
                    VV[j][i] = cadd(Ei, t);
                    VV[j][i + 4] = csub(Ei, t);
                }
        } else {
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const c64 Ei = reinterpret_cast<const c64*>(theirs)[(j * 4 + i) * 64 + lane];
                    const c64 t = cmul_tw<+1>(VV[j][4 + i], wc[64 * i]);
                    VV[j][i] = cadd(Ei, t);
                    VV[j][i + 4] = csub(Ei, t);
                }
        }
    };

    c64 prod[2][8];
    {
        // last GGSW row (j == k): plain transforms of both polynomials (fft_ops.rs:243-247); FFT(x.b) also starts
        // row 0: y.a[0] = FFT(x.b) (fft_ops.rs:225-241).  PolynomialRef::fft of a full-range polynomial: u64 -> i64 -> f64.
        c64 VV[2][8];
#pragma unroll
        for (int p = 0; p < 2; p++)
#pragma unroll
            for (int n1 = 0; n1 < 8; n1++) {
                const double re = (double)(long long)x[p * kN + coef2(n1)];
                const double im = (double)(long long)x[p * kN + coef2(8 + n1)];
                VV[p][n1] = cmul_nf({re, im}, twist[64 * n1]);
            }
        forward_pair(VV);
        if (owns_output) {
            store_bins(out_row1, VV[0]);
            store_bins(out_row1 + kHalf, VV[1]);
        }
#pragma unroll
        for (int r = 0; r < 8; r++) prod[0][r] = VV[1][r];
        __syncthreads(); // cross data consumed: the regions may be overwritten
    }
    uint64_t st[16];
#pragma unroll
    for (int e = 0; e < 16; e++) st[e] = radix_round_state<L, LOGB>(x[coef2(e)]);
#pragma unroll 1
    for (int c = 0; c < PAIRS; c++) {
        const bool single = 2 * c + 1 >= L;
        c64 VV[2][8];
#pragma unroll
        for (int n1 = 0; n1 < 8; n1++) {
            const int dre0 = next_digit<LOGB>(st[n1]), dim0 = next_digit<LOGB>(st[8 + n1]);
            const int dre1 = next_digit<LOGB>(st[n1]), dim1 = next_digit<LOGB>(st[8 + n1]);
            const c64 tw = twist[64 * n1];
            VV[0][n1] = cmul_nf({(double)dre0, (double)dim0}, tw);
            VV[1][n1] = cmul_nf({(double)dre1, (double)dim1}, tw); // the half pair transforms whatever the exhausted state yields; unused
        }
        if (c > 0) ring_dma(c); // the ring is free since the barrier behind the previous MADs
        forward_pair(VV);
#pragma unroll
        for (int j = 0; j < 2; j++) {
            if (j == 1 && single) break;
            const c64* row = reinterpret_cast<const c64*>(ring + (1 - j) * kBskSlotBytes) + 256 * w + lane;
            c64 kb[2][2];
            auto key2 = [&](int grp, c64 (&dst)[2]) {
#pragma unroll
                for (int i = 0; i < 2; i++) {
                    const int r = (grp * 2 + i) & 7, q = grp >> 2;
                    dst[i] = row[q * kHalf + 64 * (r & 3) + 512 * (r >> 2)];
                }
            };
            key2(0, kb[0]);
#pragma unroll
            for (int grp = 0; grp < 8; grp++) {
                if (grp + 1 < 8) key2(grp + 1, kb[(grp + 1) % 2]);
                compiler_fence();
#pragma unroll
                for (int i = 0; i < 2; i++) {
                    const int r = (grp * 2 + i) & 7, q = grp >> 2;
                    const c64 k = kb[grp % 2][i];
                    const bool zero = c == 0 && j == 0 && q == 1; // row 0's second polynomial starts from zero
                    double re = __builtin_fma(k.re, VV[j][r].re, zero ? 0.0 : prod[q][r].re);
                    double im = __builtin_fma(k.re, VV[j][r].im, zero ? 0.0 : prod[q][r].im);
                    prod[q][r].re = __builtin_fma(-k.im, VV[j][r].im, re);
                    prod[q][r].im = __builtin_fma(k.im, VV[j][r].re, im);
                }
            }
        }
        __syncthreads(); // every wave is done with the ring and with its partner's cross data
    }
    if (!owns_output) return;
    store_bins(out_row0, prod[0]);
    store_bins(out_row0 + kHalf, prod[1]);
}

template <int L, int LOGB>
__global__ __launch_bounds__(512, 2) void scheme_switch_kernel(SchemeSwitchArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) & 1) scheme_switch_body<L, LOGB, 1>(a, smem);
    else scheme_switch_body<L, LOGB, 0>(a, smem);
}

} // namespace spf
