// spf_kernels.hpp — gfx950 kernels of the bootstrap path.
//
// Blind rotation, TWO WAVEFRONTS PER CIPHERTEXT split by sample parity (wave w owns the complex samples of parity w —
// polynomial coefficients c = half*1024 + 128*n1 + 2*lane + w — exactly the input of one of the two 512-point transforms
// of DAG-I; the forward transform ends with, and the inverse starts with, an exchange between the two waves):
//   blind_rotate2p_kernel   throughput shape: four ciphertexts per 512-thread workgroup, two waves per SIMD, key ring
//   blind_rotate2p2_kernel  the same body with two ciphertexts per workgroup (one to two ciphertexts per CU)
//   blind_rotate8_kernel    latency shape: eight waves per ciphertext (parity x polynomial x digit), at most one ciphertext per CU
// and the CMUX-tree gates (cmux_kernel, cmux4_kernel), the int8-MFMA keyswitch, and the small streaming kernels.
// Each wave keeps its share of the GLWE accumulator and of the frequency-domain product in registers for all n steps.
// The bootstrapping key is read in the reference's own layout (natural DFT bin order): lane l touches bins l + 64 r,
// every key access is a fully coalesced 1 KiB wave access, and all workgroups walk the key in the same order.
// (r01's one-wave-per-ciphertext kernel, the one-digit-at-a-time two-wave kernel and the two-wave latency kernel were
// retired in r03: no dispatch path selected them any more; their measurements are in DESIGN.md §5.)
//
// Reference path reproduced (sunscreen_tfhe/src):
//   ops/bootstrapping/programmable_bootstrapping.rs:342-410  generalized_programmable_bootstrap
//   ops/fft_ops.rs:149-181 cmux, :23-56 glwe_ggsw_mad, :67-98 decomposed_polynomial_glev_mad
//   math/radix.rs:157-162 round, math/simd/scalar.rs:52-71 vector_next_decomp
//   entities/polynomial.rs:171-236 monomial rotation, :257-274 fft
//   entities/polynomial_fft.rs:82-99 ifft, math/simd/scalar.rs:12-35,75-119
#pragma once
#include <type_traits>
#include "spf_device.hpp"

namespace spf {

constexpr int kN = 2048;      // polynomial degree the kernels are built for
constexpr int kHalf = 1024;   // complex bins per polynomial
constexpr int kWavesPerBlock = 4;

struct BlindRotateArgs {
    const uint64_t* lwe_in;   // B x (n+1)
    const uint64_t* lut;      // (k+1)*N words, or B of them
    size_t lut_stride;        // 0 = shared
    const c64* bsk;           // [n][2][L][2][1024]
    const c64* tables;        // kTableEntries
    uint64_t* out;            // B x out_stride
    size_t out_stride;
    uint32_t n;               // LWE dimension
    uint32_t B;
    uint32_t log_chi, log_v;
    uint64_t body_rotate;
    uint32_t sample_extract;  // 0: write GLWE (2N words), 1: write LWE (N+1 words), index 0
    uint64_t* stamps;         // diagnostic builds (-DSPF_STAMPS) only: [workgroup][wave][16] cycle sums per phase
};

// modulus_switch (ops/ciphertext/lwe_ciphertext_ops.rs:130-142) to 2N = 4096
__device__ __forceinline__ uint32_t mod_switch_2n(uint64_t x, uint32_t log_chi, uint32_t log_v)
{
    const uint32_t log_modulus = 12;
    x = x << log_chi;
    uint32_t shift = 64 - (log_modulus - log_v);
    uint64_t round = (x >> (shift - 1)) & 1;
    x = x >> shift;
    return (uint32_t)(((x + round) & ((1u << log_modulus) - 1)) << log_v);
}

constexpr int kBskSlotBytes = 2 * kHalf * 16; // one (row, level) of the bootstrapping key: both output polynomials

// One LDS-DMA piece in the scalar-base form: 64 lanes x 16 bytes from `sbase + voff` (sbase wave-uniform
// in an SGPR pair, voff this lane's 32-bit byte offset) to LDS bytes [lds, lds + 1024) (lds wave-uniform,
// lane l lands at lds + 16 l).  Written out because the builtin, given `base + k * stride + lane offset`,
// keeps one 64-bit per-lane address per piece live (16 VGPRs for an 8-piece refill).  hipcc does not accept
// M0 as a clobber, so the asm saves M0 in an SGPR and puts it back behind the load (the instruction reads M0
// when it issues): whatever the compiler may keep in M0 around this point survives.
__device__ __forceinline__ void lds_dma_piece(const void* sbase, uint32_t voff, uint32_t lds)
{
    uint32_t saved_m0;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(saved_m0) : "v"(voff), "s"(sbase), "s"(lds) : "memory");
}
// the same for the waves whose (wave-uniform) flag is set, as ONE opaque group: a C++ branch on the wave index around a piece makes
// hipcc restructure the step loop (1 060 B of scratch per lane in the three-ciphertext shape, 28.7 ms instead of 9)
__device__ __forceinline__ void lds_dma_piece_if(uint32_t flag, const void* sbase, uint32_t voff, uint32_t lds)
{
    uint32_t saved_m0, tmp;
    asm volatile("v_readfirstlane_b32 %1, %2\n\ts_cmp_eq_u32 %1, 0\n\ts_cbranch_scc1 .Ldmaskip%=\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %3, %4\n\ts_mov_b32 m0, %0\n.Ldmaskip%=:"
                 : "=&s"(saved_m0), "=&s"(tmp) : "v"(flag), "v"(voff), "s"(sbase), "s"(lds) : "memory", "scc");
}
__device__ __forceinline__ uint32_t lds_address(const void* p)
{
    return (uint32_t)(size_t)(const __attribute__((address_space(3))) void*)p;
}

// The 32 640-byte twiddle image global -> LDS by LDS-DMA (NT threads, 16 bytes each per piece; the last piece is partial and
// runs under the lanes' own execution mask).  Asynchronous and without a register round trip: a kernel a workgroup runs ONCE
// issues it before anything else, so it travels under the pointer and operand loads (cmux_kernel: the copy as global loads +
// ds_write queued behind the 64 operand loads of a lane was 6.3 k of a gate's 70 k cycles).  Completion counts on vmcnt.
template <int NT>
__device__ __forceinline__ void table_image_dma(const c64* tables, char* smem, int tid, int wv)
{
    const char* src = reinterpret_cast<const char*>(tables);
    const uint32_t voff = (uint32_t)tid * 16u;
    const uint32_t dst = lds_address(smem) + (uint32_t)wv * 1024u;
    constexpr int kRow = NT * 16, kFull = kTableBytes / kRow, kRest = kTableBytes - kFull * kRow;
#pragma unroll
    for (int k = 0; k < kFull; k++) lds_dma_piece(src + k * kRow, voff, dst + k * kRow);
    if constexpr (kRest > 0) {
        if (tid * 16 < kRest) lds_dma_piece(src + kFull * kRow, voff, dst + kFull * kRow);
    }
}

// Hand-over as a bare workgroup barrier (LDS queue drained first).  Not __syncthreads(): its fence would also drain vmcnt, i.e. wait for the key
// loads in flight.  (The flat-polled words of pair_barrier wait on vmcnt too.)
__device__ __forceinline__ void pair_barrier_w()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

constexpr int kBlindRotate2pLds = kTableBytes + 4 * kWaveBufBytes + 2 * kBskSlotBytes;

// s_setprio PRIO for the waves whose flag is set, as ONE opaque instruction group: a C++ branch on the (runtime, wave-uniform)
// flag makes hipcc restructure the step loop around it (the same unswitching that a conditional barrier provokes)
template <int PRIO>
__device__ __forceinline__ void young_prio(uint32_t flag) {
    uint32_t tmp; // (hipcc hands an "s" INPUT over in a VGPR when it keeps the flag there: read it back explicitly)
    if constexpr (PRIO != 0)
        asm volatile("v_readfirstlane_b32 %0, %1\n\ts_cmp_eq_u32 %0, 0\n\ts_cbranch_scc1 .Lprio%=\n\ts_setprio 1\n.Lprio%=:" : "=&s"(tmp) : "v"(flag) : "scc");
    else
        asm volatile("v_readfirstlane_b32 %0, %1\n\ts_cmp_eq_u32 %0, 0\n\ts_cbranch_scc1 .Lprio%=\n\ts_setprio 0\n.Lprio%=:" : "=&s"(tmp) : "v"(flag) : "scc");
}

// Issue-priority schedule of blind_rotate2p_body: one character per point of a CMUX step, '-' = leave the priority alone,
// '0' / '1' = the younger waves (4-7, the SIMD partners of waves 0-3) set that priority.  Points, in program order:
//    0 top of the step (polynomial 0)      1 p0 staged, before hand-over 1       2 behind hand-over 1, before the gather
//    3 p0 twisted, before hand-over 2      4 before p0's forward pair            5 behind p0's forward pair
//    6 behind p0's key barrier             7 behind p0's MADs                    8 behind p0's ring barrier (p1 staging)
//    9 behind p1's hand-over 1            10 before p1's forward pair           11 behind p1's cross write
//   12 behind p1's key barrier            13 behind p1's ring barrier           14 behind the inverse cross exchange
//   15 behind the inverse pair            16 behind the conversion of q = 0     17 / 18 / 19 half-way through p0's forward
//   pair / p1's forward pair / the inverse pair
// (hand-overs 1 and 2 exist in the mixing instantiation only).  r03c's schedule, tuned on the even-rotation instantiation:
// the younger waves lead from 14 to 4 and from 10 to 11 (r04 re-checked it with the half-way points 17-19: four more
// schedules, 42.35-44.07 ms against 42.29).  The mixing instantiation (the plain PBS, log_v = 0) has two more barriers per
// polynomial, right where r03c's long stretch runs, so its younger waves arrived 4 600 cycles early at hand-over 1; its own
// schedule (r04, twelve timed: profiles/r04_experiments_blind_rotate.md) lets them lead through the FIRST HALF of each of
// the three transform pairs and nowhere else: 46.4 -> 44.9 ms per 4096.
// (r05, after the negated accumulator had shortened the integer phases: every single-position change re-timed, tools/sweep_prio.py +
// tools/gpu_sweep_prio.sh — position 5 -> '0' 38.44-38.53 ms against 38.72-38.79, three runs of nine launches; the mixing schedule stays)
#ifndef SPF_PRIO_SCHED_EVEN
#define SPF_PRIO_SCHED_EVEN "----00----10--1-----"
#endif
#ifndef SPF_PRIO_SCHED_MIX
#define SPF_PRIO_SCHED_MIX "----1-----1---10-00-"
#endif
template <int MIX> constexpr char prio_sched_at(int i) { return MIX ? SPF_PRIO_SCHED_MIX[i] : SPF_PRIO_SCHED_EVEN[i]; }
static_assert(sizeof(SPF_PRIO_SCHED_EVEN) == 21 && sizeof(SPF_PRIO_SCHED_MIX) == 21, "20 schedule points");

// OPT: bit 0 = exchange 2 of BOTH transforms of a pair in registers (lane_transpose_hi3), bit 1 = of the second
// transform only (balances the LDS store path against the VALU), bit 2 = two key pairs in flight in the MAD instead of
// three (8 registers, 32 B of scratch less), bit 3 = inverse cross exchange through the key ring (one barrier instead of
// two).  The library instantiates OPT = 6 — and 10 for the mixing instantiation of the four-per-workgroup shape (SPF_BR_OPT /
// SPF_BR_OPT_MIX / SPF_BR2_OPT below; r03-r04 shipped 14: bit 3 was worth 0.3 ms on the ten-barrier kernel, costs 0.25 ms for even
// rotations today and still pays 0.2 ms in the mixing instantiation); the A/B numbers are in profiles/r05_experiments_blind_rotate.md,
// everything else tried on this kernel in profiles/r02_… and r03_experiments_blind_rotate.md.
// Which transform pair each of the three pairs of a step uses (spf_device.hpp; all give the same words), per instantiation:
// E = even rotations (circuit bootstrap), M = mixing (plain PBS); 0 / 1 = polynomial 0's / 1's forward pair, I = the inverse pair.
// All share the pass twiddles between the two transforms and spread the stores through the butterflies (r04: 42.6 -> 39.9 ms
// per 4096); `…ts2` also issues an exchange's reads under the other transform's twiddle products — it pays where registers are
// slack (39.8 -> 39.4 ms in the even instantiation with polynomial 1's pair left on `…ts`; the mixing one loses 0.2-0.5 ms with it).
#ifndef SPF_PAIR_E0
#define SPF_PAIR_E0 fft512_pair1ts2
#endif
#ifndef SPF_PAIR_E1
#define SPF_PAIR_E1 fft512_pair1ts
#endif
#ifndef SPF_PAIR_EI
#define SPF_PAIR_EI fft512_pair1ts2
#endif
#ifndef SPF_PAIR_M0
#define SPF_PAIR_M0 fft512_pair1ts
#endif
#ifndef SPF_PAIR_M1
#define SPF_PAIR_M1 fft512_pair1ts
#endif
#ifndef SPF_PAIR_MI
#define SPF_PAIR_MI fft512_pair1ts
#endif
#ifndef SPF_BR_OPT
#define SPF_BR_OPT 6  // blind_rotate2p_kernel (four ciphertexts per workgroup): r05 A/B 38.73-38.79 ms per 4096 against 39.01 with 14, plain PBS 40.66-40.75 against 41.06-41.16
#endif
#ifndef SPF_BR_OPT_MIX
#define SPF_BR_OPT_MIX 10 // ... and its mixing instantiation (plain PBS): the inverse cross exchange through the key ring pays there (r05, on the negated-accumulator build: 39.72 / 39.83 ms against 39.98-40.17 with 6)
#endif
#ifndef SPF_BR2_OPT
#define SPF_BR2_OPT 6  // blind_rotate2p2_kernel (two per workgroup): 6.87 ms per 512 against 6.96 with 14 (plain PBS 7.07 / 7.08)
#endif
#ifndef SPF_BSK_PRESCALED
#define SPF_BSK_PRESCALED 1 // the device image of the bootstrap key carries the inverse transform's 1/1024 (scale_bootstrap_key_kernel)
#endif
#ifndef SPF_TWIST_PRE
#define SPF_TWIST_PRE 1     // the eight twist factors of a polynomial requested at once ...
#endif
#ifndef SPF_TWIST_PRE_E4
#define SPF_TWIST_PRE_E4 0  // ... except in the even-rotation instantiation of the four-per-workgroup shape (r05, see the twist)
#endif
#ifndef SPF_GATHER_FENCE
#define SPF_GATHER_FENCE 1  // ... and all sixteen rotation-gather reads out before the first is consumed (together −0.2 / −0.4 %)
#endif
#ifndef SPF_COMBINE_PRE
#define SPF_COMBINE_PRE 0
#endif
// SPF_BR_NEG = 1: the wave keeps the NEGATED accumulator nacc = -acc (mod 2^64) in its registers and stages that.  Every 64-bit
// subtraction of the step becomes an addition — the rotate-and-subtract `(+-acc[src]) - acc[me]` is `(+-nacc[src]) + nacc[me]`
// with the gather's sign flipped, the rounding bit 2^31 and the +1 of the two's complement ride on ONE 32-bit addend
// (v_mad_u64_u32), and the rounded top word is simply the high word of the sum; the torus conversion adds the integer of the
// negated value (untwist_sub_from_negated).  hipcc emits a 9.8-cycle v_sub_co / v_subb pair for a 64-bit subtraction and a
// 4.8-cycle v_lshl_add_u64 for an addition (tools/microbench/valu_rates.hip).  Same words: integer identities only.
#ifndef SPF_BR_NEG
#define SPF_BR_NEG 1
#endif
// SPF_ABL = n: TIMING-ONLY ablations of blind_rotate2p_body (wrong results; never in the library): what does the step cost without
// 1 the torus conversion, 2 the rotation gather + decomposition, 3 the register-side exchange, 4 the multiply-accumulate and its
// key reads, 5 the forward pairs, 6 the inverse pair.  profiles/r05_valu_rates.md, "what the step is made of".
#ifndef SPF_ABL
#define SPF_ABL 0
#endif
template <int L, int LOGB, int OPT, int W, int CTS = 4, int MIX = 1>
__device__ __forceinline__ void blind_rotate2p_body(const BlindRotateArgs& a, char* smem)
{
    constexpr int XP = (OPT & 1) ? 1 : ((OPT & 2) ? 2 : 0);
    // (A/B knobs: the exchange-2 choice of each of the three pairs of a step separately; default = the one OPT selects)
#ifdef SPF_XP_F0
    constexpr int XPF0 = SPF_XP_F0;
#else
    constexpr int XPF0 = XP;
#endif
#ifdef SPF_XP_F1
    constexpr int XPF1 = SPF_XP_F1;
#else
    constexpr int XPF1 = XP;
#endif
#ifdef SPF_XP_I
    constexpr int XPI = SPF_XP_I;
#else
    constexpr int XPI = XP;
#endif
    constexpr bool NEG = SPF_BR_NEG != 0;
#ifdef SPF_STAMPS
    // per-phase wall cycles of this wave (diagnostic build; s_memtime drains lgkmcnt: ~5 % overhead)
    uint64_t st_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t st_prev = __builtin_amdgcn_s_memtime();
#define STAMP(i) do { uint64_t t_ = __builtin_amdgcn_s_memtime(); st_acc[i] += t_ - st_prev; st_prev = t_; } while (0)
#else
#define STAMP(i) do { } while (0)
#endif
    static_assert(L == 2 && LOGB == 16, "two 16-bit digits, taken straight from the rounded top word and processed as a pair");
    constexpr int NT = 128 * CTS; // CTS ciphertexts per workgroup, two waves each
    c64* tab = reinterpret_cast<c64*>(smem);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6); // wave-uniform values live in SGPRs
    const int cslot = wv >> 1;
    constexpr int w = W; // sample parity of this wave: compile-time, the kernel runs one copy of the body per parity
    char* tile = smem + kTableBytes + cslot * kWaveBufBytes;
    char* mine = tile + w * 8192;
    char* theirs = tile + (w ^ 1) * 8192;
    char* bskring = smem + kTableBytes + CTS * kWaveBufBytes;

    {
        const double2* src = reinterpret_cast<const double2*>(a.tables);
        double2* dst = reinterpret_cast<double2*>(smem);
        for (int i = tid; i < kTableEntries; i += NT) dst[i] = src[i];
    }

    const uint32_t ct_raw = blockIdx.x * CTS + cslot;
    const bool owns_output = ct_raw < a.B;
    const uint32_t ct = owns_output ? ct_raw : a.B - 1;
    const uint64_t* lwe = a.lwe_in + (size_t)ct * (a.n + 1);
    const uint64_t* lut = a.lut + (size_t)ct * a.lut_stride;

    // chunk c = 2 step + p: the 64 KiB [level 0 row | level 1 row] of polynomial p of step `step`,
    // copied as it lies: digit j (level L-1-j) is the ring half 1-j
    const uint32_t total_chunks = 2 * a.n;
    const uint32_t dma_voff = (uint32_t)tid * 16u;                   // this lane's bytes inside an 8 KiB piece row
    const uint32_t dma_dst = lds_address(bskring) + wv * 1024;        // this wave's 1 KiB of each piece row
    constexpr int kPieces = 2 * kBskSlotBytes / 1024, NW = 2 * CTS;   // 1 KiB pieces of a chunk; waves of the workgroup
    auto ring_dma = [&](uint32_t chunk) {
        const char* src = reinterpret_cast<const char*>(a.bsk) +
                          (size_t)__builtin_amdgcn_readfirstlane(chunk) * (2 * kBskSlotBytes); // uniform
        if constexpr (kPieces % NW == 0) {
#pragma unroll
            for (int k = 0; k < 2 * kBskSlotBytes / (NT * 16); k++)
                lds_dma_piece(src + k * NT * 16, dma_voff, dma_dst + k * NT * 16);
        } else {
            // (three ciphertexts per workgroup: 64 pieces over six waves — wave v takes pieces v, v + 6, ...)
            const uint32_t lane16 = (uint32_t)lane * 16u, ring0 = lds_address(bskring);
#pragma unroll
            for (int k = 0; k < kPieces / NW; k++) lds_dma_piece(src + (k * NW + wv) * 1024, lane16, ring0 + (k * NW + wv) * 1024);
            lds_dma_piece_if(wv < kPieces % NW ? 1u : 0u, src + ((kPieces / NW) * NW + wv) * 1024, lane16, ring0 + ((kPieces / NW) * NW + wv) * 1024);
        }
    };
    ring_dma(0);

    auto coef2 = [&](int e) -> int { return (e >> 3) * 1024 + (e & 7) * 128 + 2 * lane + w; };

    uint64_t acc[2][16];
    {
        uint32_t bt = mod_switch_2n(lwe[a.n] + a.body_rotate, a.log_chi, a.log_v);
#pragma unroll
        for (int p = 0; p < 2; p++)
#pragma unroll
            for (int e = 0; e < 16; e++) {
                uint32_t idx = (uint32_t)coef2(e) + bt;
                uint64_t v = lut[p * kN + (idx & (kN - 1))];
                if constexpr (NEG) acc[p][e] = ((idx >> 11) & 1) ? v : (uint64_t)0 - v; // (NEG: acc[][] holds -accumulator throughout)
                else acc[p][e] = ((idx >> 11) & 1) ? (uint64_t)0 - v : v;
            }
    }
    __syncthreads();
    // Issue priority, swapped in the middle of each long barrier-to-barrier stretch.  A SIMD holds wave v and wave v + 4 of
    // the workgroup and the arbiter prefers the older one, which then reaches the next barrier thousands of cycles early and
    // idles there while its partner runs ALONE (nobody fills that wave's LDS round trips).  The younger waves take priority
    // 1 for the first part of a stretch and give it back for the rest, so both arrive together: 44.9 -> 42.7 ms per 4096
    // (profiles/r03_experiments_blind_rotate.md, "issue priority"; with ten barriers a step the stretches were too short
    // for this to pay — r02 measured it as a loss).
    constexpr bool PRIO = CTS == 4; // (two waves per SIMD only with four ciphertexts per workgroup)
    const uint32_t is_young = __builtin_amdgcn_readfirstlane(wv >= CTS ? 1u : 0u);
    uint32_t opaque_zero;
    asm volatile("s_mov_b32 %0, 0" : "=s"(opaque_zero));
    // Every hand-over between the two waves of a ciphertext is a bare s_barrier of the whole workgroup
    // (LDS queue drained, vmcnt NOT: key rows stay in flight across it).  The four ciphertexts are tied
    // together by the key ring twice per polynomial anyway, and the hardware barrier costs a few dozen
    // cycles where a flat-polled word per wave pair cost ~2 000 (store -> visible -> load round trips
    // through the vector-memory path): 50.0 -> 47.8 ms per 4096.  The loop never repeats (opaque_zero is
    // 0, which the compiler cannot know); it gives each phase its own basic block — as one straight-line
    // region the step spills three times as much.
    auto rendezvous = [&]() {
        do {
            pair_barrier_w();
        } while (opaque_zero != 0);
    };
#define SPF_PRIO_POINT(i) do { if constexpr (PRIO) { if constexpr (prio_sched_at<MIX>(i) == '1') young_prio<1>(is_young); \
                                                      else if constexpr (prio_sched_at<MIX>(i) == '0') young_prio<0>(is_young); } } while (0)
    // MIX = 0: every rotation amount is even (log_v >= 1), an even rotation keeps the coefficient parity, a wave gathers
    // only what it staged itself and in-order LDS needs no hand-over for that; the block structure stays the same
    auto rendezvous_if_mixing = [&]() {
        do {
            if constexpr (MIX) pair_barrier_w();
            else asm volatile("" ::: "memory");
        } while (opaque_zero != 0);
    };

    uint64_t* stage_mine = reinterpret_cast<uint64_t*>(mine);
    const c64* twist = tab + kTWOff + w * 512 + lane;   // e^{+i pi (2n'+w)/2048}, n' = 64 n1 + lane
    const c64* wc = tab + kWCOff + 256 * w + lane;       // W1024^{lane + 64 (4w + i)}
    uint64_t a_next = lwe[0];
    uint32_t chunk = 0;
    auto stage = [&](int p) {
#pragma unroll
        for (int e = 0; e < 16; e++) stage_mine[(e >> 3) * 512 + (e & 7) * 64 + lane] = acc[p][e];
    };
    for (uint32_t step = 0; step < a.n; step++) {
        const uint32_t at = mod_switch_2n(a_next, a.log_chi, a.log_v);
        a_next = lwe[step + 1];
        STAMP(11);

        // bins lane + 64 (4w + i) + 512 s at index i + 4 s; starts at zero, which the first row's FMAs
        // take as a literal (no zeroed registers live across polynomial 0's transforms)
        c64 prod[2][8];

#pragma unroll
        for (int p = 0; p < 2; p++, chunk++) {
            // my region is free: for p = 0 the partner's last reads of it (inverse cross data) were
            // followed by a rendezvous, for p = 1 by the workgroup barrier behind the MADs
            if (p == 0) SPF_PRIO_POINT(0); else SPF_PRIO_POINT(8);
            stage(p);
            if (p == 0) SPF_PRIO_POINT(1);
            // With log_v >= 1 (MIX = 0: the modulus switch clears the low log_v bits — the circuit bootstrap uses log_v = 2)
            // the two gather hand-overs of a polynomial are not needed: four of the ten barriers of a step go.
            rendezvous_if_mixing(); // both parities staged
            STAMP(0);
            if (p == 0) SPF_PRIO_POINT(2); else SPF_PRIO_POINT(9);
            uint32_t dig[16];
#if SPF_ABL == 2 // (timing-only ablation, wrong results: no rotation gather, no subtraction, no rounding)
#pragma unroll
            for (int e = 0; e < 16; e++) dig[e] = (uint32_t)(acc[p][e] >> 32) ^ at;
#else
            {
                // source coefficient of element e: (c_e - at) mod 2N with c_e = c_0 + 128 m (m = e & 7, +1024
                // for e >= 8): region (parity) and the low address bits do not depend on e
                const uint32_t t0 = (uint32_t)(2 * lane + w) + 2 * kN - at;
                const char* region = tile + (t0 & 1) * 8192;
                uint32_t T0 = (t0 + 2048u) << 20; // (NEG) bit 31 = complement of bit 11 of t0; opaque so that it is re-derived per polynomial
                if constexpr (NEG) asm volatile("" : "+v"(T0));
                uint64_t gin[16];
#pragma unroll
                for (int e = 0; e < 16; e++) {
                    const uint32_t t = t0 + (e >> 3) * 1024 + (e & 7) * 128;
                    gin[e] = *reinterpret_cast<const uint64_t*>(region + ((t << 2) & 0x1FF8u));
                }
#if SPF_GATHER_FENCE
                sched_fence(); // all sixteen reads out before the first is consumed (hipcc otherwise issues them four at a time)
#else
                compiler_fence();
#endif
#pragma unroll
                for (int e = 0; e < 16; e++) {
                    const uint32_t t = t0 + (e >> 3) * 1024 + (e & 7) * 128;
                    if constexpr (NEG) {
                        // gin = nacc[src]; x = rot - acc[me] = (-+ nacc[src]) + nacc[me]: the sign mask m is the complement of bit 11 of t
                        // (= bit 11 of t + 2048, brought to bit 31 once per polynomial: T0 + a literal per element);
                        // x + 2^31 = (gin ^ m) + (nacc[me] + (2^31 + (m ? 1 : 0))), and the rounded top word is its high word.
                        // The small pieces are pinned (opaque values, one v_mad_u64_u32): left to itself hipcc rebuilds the mask from a
                        // bit-field extract, widens the 32-bit addend to a register pair and keeps all of it live across the transforms.
                        const uint32_t m32 = (uint32_t)((int32_t)(T0 + (uint32_t)(((e >> 3) * 1024 + (e & 7) * 128) << 20)) >> 31);
                        const uint64_t m = ((uint64_t)m32 << 32) | m32;
                        uint32_t k32 = 0x80000000u - m32;
                        dig[e] = (uint32_t)(((gin[e] ^ m) + add_u32_to_u64(acc[p][e], k32)) >> 32);
                    } else {
                        const uint64_t sgn = (uint64_t)((int64_t)((uint64_t)t << 52) >> 63); // bit 11 of t, spread
                        const uint64_t rot = (gin[e] ^ sgn) - sgn;
                        dig[e] = gadget_round_top32(rot - acc[p][e]); // the rounded top word; its two digits are taken at the twist
                    }
                }
            }
#endif
            c64 VV[2][8];
            // (r05, on the negated-accumulator build: requested one by one 37.86-37.92 ms per 4096 against 38.20-38.24 for even rotations
            // on the four-per-workgroup shape; the mixing instantiation and the two-per-workgroup shape keep the batch: 39.85 against 39.95, 6.68 against 6.72)
            constexpr bool kTwistBatch = (!MIX && CTS == 4) ? (SPF_TWIST_PRE_E4 != 0) : (SPF_TWIST_PRE != 0);
            if constexpr (kTwistBatch) {
                // the eight twist factors in one go (hipcc fetches them two at a time, each pair waited for on the spot)
                c64 twf[8];
#pragma unroll
                for (int n1 = 0; n1 < 8; n1++) twf[n1] = twist[64 * n1];
                compiler_fence();
#pragma unroll
                for (int n1 = 0; n1 < 8; n1++) {
                    VV[0][n1] = twisted_digit_top32(dig[n1], dig[8 + n1], 0, twf[n1]);
                    VV[1][n1] = twisted_digit_top32(dig[n1], dig[8 + n1], 1, twf[n1]);
                }
            } else {
#pragma unroll
                for (int n1 = 0; n1 < 8; n1++) {
                    const c64 tw = twist[64 * n1];
                    VV[0][n1] = twisted_digit_top32(dig[n1], dig[8 + n1], 0, tw);
                    VV[1][n1] = twisted_digit_top32(dig[n1], dig[8 + n1], 1, tw);
                }
            }
            STAMP(1);
            if (p == 0) SPF_PRIO_POINT(3);
            rendezvous_if_mixing(); // partner is done gathering from my region
            STAMP(2);
            // the ring is free since the barrier behind the last MADs: bring in polynomial 1's rows (those
            // of polynomial 0 were requested ahead of the previous step's inverse transforms)
            // (r03c: polynomial 0: the older waves lead through the forward transforms; polynomial 1: the younger)
            if (p == 0) SPF_PRIO_POINT(4); else SPF_PRIO_POINT(10);
            if (p == 1) ring_dma(chunk);
#if SPF_ABL != 5 // (5: timing-only, the forward transform pairs are not executed)
            if constexpr (MIX) {
                if (p == 0) SPF_PAIR_M0<+1, XPF0>(VV[0], VV[1], mine, tab, lane, [&]() { SPF_PRIO_POINT(17); });
                else SPF_PAIR_M1<+1, XPF1>(VV[0], VV[1], mine, tab, lane, [&]() { SPF_PRIO_POINT(18); });
            } else {
                if (p == 0) SPF_PAIR_E0<+1, XPF0>(VV[0], VV[1], mine, tab, lane, [&]() { SPF_PRIO_POINT(17); });
                else SPF_PAIR_E1<+1, XPF1>(VV[0], VV[1], mine, tab, lane, [&]() { SPF_PRIO_POINT(18); });
            }
#endif
            STAMP(3);
            if (p == 0) SPF_PRIO_POINT(5);
            // radix-2 stage across the two waves, both digits in one exchange: wave 0 finishes bins with
            // d < 4 and sends registers 4..7, wave 1 the other way round
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
            if (p == 1) SPF_PRIO_POINT(11);
#ifdef SPF_STAMPS
            STAMP(4);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            STAMP(2); // diagnostic: the wait for the key rows alone (slot 2 is otherwise empty for even rotations)
#endif
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // my share of the key rows has landed
            __syncthreads();
            STAMP(4);
            if (p == 0) SPF_PRIO_POINT(6); else SPF_PRIO_POINT(12);
            // X[i] = E[i] + W^k O[i], X[i+4] = E[i] - W^k O[i]: wave 0 holds E and receives O, wave 1 the reverse
#if SPF_COMBINE_PRE
            {
                // all eight cross values and the four cross twiddles requested at once behind the barrier
                c64 xin[2][4], wcf[4];
#pragma unroll
                for (int i = 0; i < 4; i++) wcf[i] = wc[64 * i];
#pragma unroll
                for (int j = 0; j < 2; j++)
#pragma unroll
                    for (int i = 0; i < 4; i++) xin[j][i] = reinterpret_cast<const c64*>(theirs)[(j * 4 + i) * 64 + lane];
                compiler_fence();
#pragma unroll
                for (int j = 0; j < 2; j++)
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const c64 Ei = w == 0 ? VV[j][i] : xin[j][i];
                        const c64 t = cmul_tw<+1>(w == 0 ? xin[j][i] : VV[j][4 + i], wcf[i]);
                        VV[j][i] = cadd(Ei, t);
                        VV[j][i + 4] = csub(Ei, t);
                    }
            }
#else
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
#endif
            STAMP(5);
#if SPF_ABL == 4 // (timing-only: no key reads from the ring, no multiply-accumulate)
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int r = 0; r < 8; r++)
#pragma unroll
                    for (int q = 0; q < 2; q++) { // (scaled into the magnitude window of the conversion's short path)
                        const bool first = p == 0 && j == 0;
                        prod[q][r].re = __builtin_fma(VV[j][r].re, 0x1p58, first ? 0.0 : prod[q][r].re);
                        prod[q][r].im = __builtin_fma(VV[j][r].im, 0x1p58, first ? 0.0 : prod[q][r].im);
                    }
#else
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const c64* row = reinterpret_cast<const c64*>(bskring + (1 - j) * kBskSlotBytes) + 256 * w + lane;
#ifdef SPF_MAD_KD
                constexpr int KD = SPF_MAD_KD;
#else
                constexpr int KD = (OPT & 4) ? 2 : 3; // key pairs in flight (bit 2: two — 8 registers fewer across the MAD)
#endif
                c64 kb[KD][2];
                auto key2 = [&](int grp, c64 (&dst)[2]) {
#pragma unroll
                    for (int i = 0; i < 2; i++) {
                        const int r = (grp * 2 + i) & 7, q = grp >> 2;
                        dst[i] = row[q * kHalf + 64 * (r & 3) + 512 * (r >> 2)];
                    }
                };
                key2(0, kb[0]);
                if constexpr (KD >= 3) key2(1, kb[1]);
                if constexpr (KD >= 4) key2(2, kb[2]);
#pragma unroll
                for (int grp = 0; grp < 8; grp++) {
                    if (grp + KD - 1 < 8) key2(grp + KD - 1, kb[(grp + KD - 1) % KD]);
                    compiler_fence();
#pragma unroll
                    for (int i = 0; i < 2; i++) {
                        const int r = (grp * 2 + i) & 7, q = grp >> 2;
                        const c64 k = kb[grp % KD][i];
                        const bool first = p == 0 && j == 0;
                        double re = __builtin_fma(k.re, VV[j][r].re, first ? 0.0 : prod[q][r].re);
                        double im = __builtin_fma(k.re, VV[j][r].im, first ? 0.0 : prod[q][r].im);
                        prod[q][r].re = __builtin_fma(-k.im, VV[j][r].im, re);
                        prod[q][r].im = __builtin_fma(k.im, VV[j][r].re, im);
                    }
                }
            }
#endif
            STAMP(6);
            if (p == 0) SPF_PRIO_POINT(7);
            __syncthreads(); // every wave is done with the ring and with its partner's cross data
            STAMP(7);
            if (p == 1) SPF_PRIO_POINT(13);
        }

        // ---- back to the torus, both output polynomials together
        c64 WW[2][8];
#pragma unroll
        for (int q = 0; q < 2; q++)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const c64 wci = wc[64 * i];
                WW[q][i] = cadd(prod[q][i], prod[q][i + 4]);                      // Ep: kept by wave 0
                WW[q][4 + i] = cmul_tw<-1>(csub(prod[q][i], prod[q][i + 4]), wci); // Op: kept by wave 1
            }
        if constexpr ((OPT & 8) != 0) {
            // The inverse cross exchange goes through the KEY RING (free between the barrier behind the last MADs and the
            // next refill): wave v writes its outgoing half into slot v (8 KiB), reads slot v^1 behind ONE barrier, and then
            // refills exactly that slot with its 8 KiB of the next key chunk — the only reader of the slot is the wave that
            // overwrites it, in program order, so the second barrier of the exchange ("cross reads retired before the
            // image is overwritten") is not needed: the transforms run in the tile, which nobody else touches any more.
            c64* slot_mine = reinterpret_cast<c64*>(bskring + wv * 8192);
            const c64* slot_theirs = reinterpret_cast<const c64*>(bskring + (wv ^ 1) * 8192);
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
            STAMP(8);
            SPF_PRIO_POINT(14); // (r03c: the long stretch starts: the younger waves lead)
            if (chunk < total_chunks) {
                const char* src = reinterpret_cast<const char*>(a.bsk) +
                                  (size_t)__builtin_amdgcn_readfirstlane(chunk) * (2 * kBskSlotBytes);
                const uint32_t lane16 = (uint32_t)lane * 16u;
                const uint32_t ring0 = lds_address(bskring);
#pragma unroll
                for (int k = 0; k < 8; k++) // the slot just read
                    lds_dma_piece(src + (wv ^ 1) * 8192 + k * 1024, lane16, ring0 + (wv ^ 1) * 8192 + k * 1024);
                constexpr int kRest = (2 * kBskSlotBytes - 2 * CTS * 8192) / 1024; // (fewer waves than slots: the rest of the ring)
                if constexpr (kRest % NW == 0) {
#pragma unroll
                    for (int k = 0; k < kRest / NW; k++)
                        lds_dma_piece(src + 2 * CTS * 8192 + (wv * (kRest / NW) + k) * 1024, lane16,
                                      ring0 + 2 * CTS * 8192 + (wv * (kRest / NW) + k) * 1024);
                } else {
#pragma unroll
                    for (int k = 0; k < kRest / NW; k++)
                        lds_dma_piece(src + 2 * CTS * 8192 + (k * NW + wv) * 1024, lane16, ring0 + 2 * CTS * 8192 + (k * NW + wv) * 1024);
                    lds_dma_piece_if(wv < kRest % NW ? 1u : 0u, src + 2 * CTS * 8192 + ((kRest / NW) * NW + wv) * 1024, lane16,
                                     ring0 + 2 * CTS * 8192 + ((kRest / NW) * NW + wv) * 1024);
                }
            }
        } else {
            if constexpr (w == 0) {
    #pragma unroll
                for (int q = 0; q < 2; q++)
    #pragma unroll
                    for (int i = 0; i < 4; i++) reinterpret_cast<c64*>(mine)[(q * 4 + i) * 64 + lane] = WW[q][4 + i];
            } else {
    #pragma unroll
                for (int q = 0; q < 2; q++)
    #pragma unroll
                    for (int i = 0; i < 4; i++) reinterpret_cast<c64*>(mine)[(q * 4 + i) * 64 + lane] = WW[q][i];
            }
            rendezvous();
            if constexpr (w == 0) {
    #pragma unroll
                for (int q = 0; q < 2; q++)
    #pragma unroll
                    for (int i = 0; i < 4; i++) WW[q][4 + i] = reinterpret_cast<const c64*>(theirs)[(q * 4 + i) * 64 + lane];
            } else {
    #pragma unroll
                for (int q = 0; q < 2; q++)
    #pragma unroll
                    for (int i = 0; i < 4; i++) WW[q][i] = reinterpret_cast<const c64*>(theirs)[(q * 4 + i) * 64 + lane];
            }
            rendezvous(); // both cross reads retired before either region is overwritten
            STAMP(8);
            SPF_PRIO_POINT(14);
            if (chunk < total_chunks) ring_dma(chunk); // rows of the next step's polynomial 0
        }
#if SPF_ABL != 6 // (6: timing-only, the inverse transform pair is not executed)
        if constexpr (MIX) SPF_PAIR_MI<-1, XPI>(WW[0], WW[1], mine, tab, lane, [&]() { SPF_PRIO_POINT(19); });
        else SPF_PAIR_EI<-1, XPI>(WW[0], WW[1], mine, tab, lane, [&]() { SPF_PRIO_POINT(19); });
#endif
        STAMP(9);
        SPF_PRIO_POINT(15);
#pragma unroll
        for (int q = 0; q < 2; q++) {
#if SPF_ABL == 1 // (timing-only: no untwist, no conversion to the torus)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[q][e] += (uint64_t)__double_as_longlong(e < 8 ? WW[q][e].re : WW[q][e - 8].im);
#else
            if constexpr (NEG) {
                untwist_sub_from_negated<SPF_BSK_PRESCALED, CTS != 4>(WW[q], twist, acc[q]);
            } else {
                uint64_t t[16];
                untwist_to_torus_bits<false, SPF_BSK_PRESCALED>(WW[q], twist, t);
#pragma unroll
                for (int e = 0; e < 16; e++) acc[q][e] += t[e];
            }
#endif
            if (q == 0) SPF_PRIO_POINT(16);
        }
        STAMP(10);
    }
#ifdef SPF_STAMPS
    if (a.stamps && lane == 0) {
#pragma unroll
        for (int i = 0; i < 12; i++) a.stamps[((size_t)blockIdx.x * (2 * CTS) + wv) * 16 + i] = st_acc[i];
    }
#endif
#undef STAMP
#undef SPF_PRIO_POINT

    if (!owns_output) return;
    uint64_t* out = a.out + (size_t)ct * a.out_stride;
    if constexpr (NEG) {
#pragma unroll
        for (int p = 0; p < 2; p++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[p][e] = (uint64_t)0 - acc[p][e];
    }
    if (!a.sample_extract) {
#pragma unroll
        for (int p = 0; p < 2; p++)
#pragma unroll
            for (int e = 0; e < 16; e++) out[p * kN + coef2(e)] = acc[p][e];
    } else {
#pragma unroll
        for (int e = 0; e < 16; e++) {
            int c = coef2(e);
            if (c == 0) {
                out[0] = acc[0][e];
                out[kN] = acc[1][e];
            } else {
                out[kN - c] = (uint64_t)0 - acc[0][e];
            }
        }
    }
}


// The body is instantiated once per sample parity (w = 0 / 1): every parity-dependent choice (which half
// of the cross exchange a wave keeps, table offsets) is then static — no value selects, no branches that
// merge register arrays (those end up in scratch), parity-dependent LDS offsets as immediates.
template <int L, int LOGB, int OPT, int MIX = 1>
__global__ __launch_bounds__(512, 2) void blind_rotate2p_kernel(BlindRotateArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) & 1) blind_rotate2p_body<L, LOGB, OPT, 1, 4, MIX>(a, smem);
    else blind_rotate2p_body<L, LOGB, OPT, 0, 4, MIX>(a, smem);
}

// The same schedule with TWO ciphertexts per workgroup (four waves, one per SIMD, one workgroup per CU): for batches
// between one and two ciphertexts per CU, where the four-ciphertext shape would leave CUs idle and the four-wave
// latency kernel needs two rounds.  The pair shares each key chunk through the ring; same words.
constexpr int kBlindRotate2p2Lds = kTableBytes + 2 * kWaveBufBytes + 2 * kBskSlotBytes;
template <int L, int LOGB, int OPT, int MIX = 1>
__global__ __launch_bounds__(256, 1) void blind_rotate2p2_kernel(BlindRotateArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) & 1) blind_rotate2p_body<L, LOGB, OPT, 1, 2, MIX>(a, smem);
    else blind_rotate2p_body<L, LOGB, OPT, 0, 2, MIX>(a, smem);
}

// THREE ciphertexts per workgroup (six waves: two SIMDs carry two waves, two carry one; one workgroup per CU): for batches
// between two and three ciphertexts per CU, where the four-ciphertext shape would leave a third of the CUs idle.  Same words.
constexpr int kBlindRotate2p3Lds = kTableBytes + 3 * kWaveBufBytes + 2 * kBskSlotBytes;
template <int L, int LOGB, int OPT, int MIX = 1>
__global__ __launch_bounds__(384, 1) void blind_rotate2p3_kernel(BlindRotateArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) & 1) blind_rotate2p_body<L, LOGB, OPT, 1, 3, MIX>(a, smem);
    else blind_rotate2p_body<L, LOGB, OPT, 0, 3, MIX>(a, smem);
}

// ------------------------------------------------------------------------------------------
// scale_bootstrap_key_kernel: the blind-rotation kernels read the caller's bootstrap-key spectra times 2^-10, so that the product
// spectra come out of the multiply-accumulate already carrying the 1/N of the inverse transform (N/2 = 1024 complex points).
// Exact for every value a forward transform of a torus polynomial can produce; `bad` is raised for a non-zero magnitude
// outside [2^-900, 2^1000) (or a NaN), where scaling first could round differently from scaling last — such a key is refused
// at load time instead of being bootstrapped with differently.
__global__ void scale_bootstrap_key_kernel(const double* key, double* scaled, size_t n, unsigned int* bad)
{
    bool flagged = false;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const double v = key[i];
        const double m = __builtin_fabs(v);
        if (!(m == 0.0 || (m >= 0x1p-900 && m < 0x1p1000))) flagged = true;
        scaled[i] = v * 0x1p-10;
    }
    if (flagged) atomicOr(bad, 1u);
}

// ------------------------------------------------------------------------------------------
// blind_rotate8_kernel: EIGHT waves per ciphertext, one ciphertext per workgroup — the latency shape, for batches of at
// most one ciphertext per CU (B <= #CU).  Wave (w, h, j): sample parity w (as in the two-wave kernels), polynomial h,
// gadget digit j; waves (w, h, 0) and (w, h, 1) are SIMD siblings.
//
// Until r04 the shape was four waves (parity x polynomial, `blind_rotate4_kernel`, now under
// profiles/r04_experimental_sources/), each running BOTH digits' forward transforms: one wave per SIMD, and its stamps
// said the step (15.1 k cycles) was latency.  Here the two polynomials of a CMUX step are still independent until the
// multiply-accumulate, and in addition
//   * the j = 0 wave, which owns the accumulator of (w, h), stages, gathers the rotation, subtracts and decomposes ONCE
//     and hands digit 1 to its sibling as sixteen 16-bit fields (a version in which both waves decomposed was 1 % slower:
//     between the staging barrier and the cross exchange the SIMD of a pair is busy, so duplicated work costs its time);
//   * each wave twists and transforms ONE digit, the sibling the other at the same time;
//   * the multiply-accumulate of output polynomial q = h is split by BINS between the two waves (registers {2j, 2j+1,
//     2j+4, 2j+5}: the pairs the inverse split needs are in one wave); per bin it is the sequential chain the
//     reference's `glwe_ggsw_mad` defines (p = 0: digits 0, 1; p = 1: digits 0, 1), and each wave fetches only its
//     quarter of a key row (16 KiB per wave and step), requested a step ahead in four pieces;
//   * the four waves of polynomial h post their E' / O' halves into the inboxes of the two j = 0 waves, which transform
//     back, untwist, convert and accumulate; the j = 1 wave meanwhile requests its next key rows.
// Same operations in the same order on every value as the other shapes: same words.  Hand-overs are s_barrier among the
// eight waves, five per step (four when every rotation is even: staged [mixing only], digits out, cross data out, spectra
// out, inverse halves out); each LDS region has one use per phase.  LDS: tables + eight 8 KiB exchange images + four
// 16 KiB staging / spectra regions = 160 KiB.  Tried and not kept, both bit-equal: converting the real and the imaginary
// halves in different waves (one more barrier: +3 %), raising the sibling's issue priority (the lag just changes sides).
constexpr int kBlindRotate8Lds = kTableBytes + 8 * 8192 + 4 * 16384;

template <int L, int LOGB, int W, int J, int MIX>
__device__ __forceinline__ void blind_rotate8_body(const BlindRotateArgs& a, char* smem)
{
    static_assert(L == 2 && L * LOGB <= 32, "two digits, one per wave of a pair");
    constexpr bool NEG = SPF_BR_NEG != 0; // negated accumulator, as in blind_rotate2p_body (3.72 -> 3.6x ms: the j = 0 wave's integer work is on the step's critical path)
    c64* tab = reinterpret_cast<c64*>(smem);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6); // = 4 j + 2 h + w
    constexpr int w = W, j = J;
    const int h = (wv >> 1) & 1;
    auto image = [&](int ww, int hh, int jj) -> char* { return smem + kTableBytes + ((jj * 2 + hh) * 2 + ww) * 8192; };
    char* mine = image(w, h, j);
    char* partner = image(w ^ 1, h, j); // same polynomial and digit, other parity
    // 16 KiB region of (w, h): the staged accumulator (rotation source) at the top of a step, the two digits' transforms later
    auto spectra = [&](int ww, int hh) -> char* { return smem + kTableBytes + 8 * 8192 + (hh * 2 + ww) * 16384; };
    auto wg_barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // not __syncthreads(): keep the key loads in flight
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    {
        const double2* src = reinterpret_cast<const double2*>(a.tables);
        double2* dst = reinterpret_cast<double2*>(smem);
        for (int i = tid; i < kTableEntries; i += 512) dst[i] = src[i];
    }
    const uint32_t ct = blockIdx.x; // grid = B
    const uint64_t* lwe = a.lwe_in + (size_t)ct * (a.n + 1);
    const uint64_t* lut = a.lut + (size_t)ct * a.lut_stride;
    auto coef2 = [&](int e) -> int { return (e >> 3) * 1024 + (e & 7) * 128 + 2 * lane + w; };

    uint64_t acc[16]; // polynomial h, parity w: lives in the j = 0 wave only
    if constexpr (J == 0) {
        uint32_t bt = mod_switch_2n(lwe[a.n] + a.body_rotate, a.log_chi, a.log_v);
#pragma unroll
        for (int e = 0; e < 16; e++) {
            uint32_t idx = (uint32_t)coef2(e) + bt;
            uint64_t v = lut[h * kN + (idx & (kN - 1))];
            if constexpr (NEG) acc[e] = ((idx >> 11) & 1) ? v : (uint64_t)0 - v; // (NEG: acc[] holds -accumulator throughout)
            else acc[e] = ((idx >> 11) & 1) ? (uint64_t)0 - v : v;
        }
    }
    __syncthreads(); // twiddle image in place

    // twist and cross-stage factors are read from the LDS tables where they are used: 256 registers per wave here
    const c64* twist_lds = tab + kTWOff + w * 512 + lane;
    const c64* wc_lds = tab + kWCOff + 256 * w + lane;

    // this wave's bins (registers 2j, 2j+1, 2j+4, 2j+5) of OUTPUT polynomial h in the four key rows of a step:
    // [row polynomial p][digit jj][q], q -> register 2j + (q & 1) + 4 (q >> 1).  Requested one row at a time, spread over the step.
    c64 key[2][2][4];
    const c64* key_base = a.bsk + h * kHalf + 256 * w + lane + 128 * j;
    const c64* key_next = key_base;
    auto request_keys = [&](auto piece_c) {
        constexpr int piece = decltype(piece_c)::value;
        constexpr int p = piece >> 1, jj = piece & 1;
        const c64* row = key_next + (size_t)(p * L + (L - 1 - jj)) * (2 * kHalf);
#pragma unroll
        for (int q = 0; q < 4; q++) key[p][jj][q] = row[64 * (q & 1) + 512 * (q >> 1)];
    };
#define SPF_KEY_PIECE(i) request_keys(std::integral_constant<int, i>{})
    SPF_KEY_PIECE(0); SPF_KEY_PIECE(1); SPF_KEY_PIECE(2); SPF_KEY_PIECE(3);
    uint64_t a_next = lwe[0];
#ifdef SPF_STAMPS
    uint64_t st_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t st_prev = __builtin_amdgcn_s_memtime();
#define STAMP8(i) do { uint64_t t_ = __builtin_amdgcn_s_memtime(); st_acc[i] += t_ - st_prev; st_prev = t_; } while (0)
#else
#define STAMP8(i) do { } while (0)
#endif
    // One CMUX step; LAST = the final one, compiled without the requests for a next step's rows (no dead loads)
    auto cmux_step = [&](uint32_t step, auto last_c) {
        constexpr bool LAST = decltype(last_c)::value;
        const uint32_t at = mod_switch_2n(a_next, a.log_chi, a.log_v);
        a_next = lwe[step + 1];
        // ---- rotate, subtract, decompose polynomial h: ONCE, in the j = 0 wave (the two waves of a pair share a SIMD, and between
        // barriers A and B that SIMD has no issue slot to spare: a decomposition done twice costs its full time twice).  The j = 1
        // wave gets its digits as 16-bit fields through its own (idle) exchange image.
        c64 V[8];
        uint4* digits1 = reinterpret_cast<uint4*>(image(w, h, 1));
        if constexpr (J == 0) {
            uint64_t* stage = reinterpret_cast<uint64_t*>(spectra(w, h));
#pragma unroll
            for (int e = 0; e < 16; e++) stage[(e >> 3) * 512 + (e & 7) * 64 + lane] = acc[e];
            STAMP8(0);
            // A: both parities staged.  Not needed when every rotation amount is even (MIX = 0): the wave then gathers only from
            // the region it staged itself
            if constexpr (MIX) wg_barrier();
            else compiler_fence();
            STAMP8(1);
            const uint32_t t0 = (uint32_t)(2 * lane + w) + 2 * kN - at;
            const char* src = spectra((int)(t0 & 1), h);
            uint32_t T0 = (t0 + 2048u) << 20; // (NEG) bit 31 = complement of bit 11 of t0
            if constexpr (NEG) asm volatile("" : "+v"(T0));
            uint64_t gin[16];
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const uint32_t t = t0 + (e >> 3) * 1024 + (e & 7) * 128;
                gin[e] = *reinterpret_cast<const uint64_t*>(src + ((t << 2) & 0x1FF8u));
            }
            c64 twist[8];
#pragma unroll
            for (int n1 = 0; n1 < 8; n1++) twist[n1] = twist_lds[64 * n1];
            sched_fence(); // all reads out before the first is consumed
            uint32_t dig[16];
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const uint32_t t = t0 + (e >> 3) * 1024 + (e & 7) * 128;
                if constexpr (NEG) {
                    // x + 2^31 = (gin ^ m) + (nacc[me] + (2^31 + (m ? 1 : 0))), m = complement of bit 11 of t (see blind_rotate2p_body)
                    const uint32_t m32 = (uint32_t)((int32_t)(T0 + (uint32_t)(((e >> 3) * 1024 + (e & 7) * 128) << 20)) >> 31);
                    const uint64_t m = ((uint64_t)m32 << 32) | m32;
                    const uint32_t k32 = 0x80000000u - m32;
                    const uint32_t s = (uint32_t)(((gin[e] ^ m) + add_u32_to_u64(acc[e], k32)) >> 32); // the rounded top word
                    dig[e] = (s & 0xFFFFu) | ((s + 0x8000u) & 0xFFFF0000u); // digit 0 | digit 1 = (s >> 16) + carry of digit 0, as packed fields
                } else {
                    const uint64_t sgn = (uint64_t)((int64_t)((uint64_t)t << 52) >> 63); // bit 11 of t, spread
                    const uint64_t rot = (gin[e] ^ sgn) - sgn;
                    dig[e] = gadget_digits_packed<L, LOGB>(rot - acc[e]);
                }
            }
            static_assert(LOGB == 16, "digit 1 of the real and of the imaginary element share a 32-bit word");
            uint32_t pk[8];
#pragma unroll
            for (int n1 = 0; n1 < 8; n1++) pk[n1] = (dig[n1] >> 16) | (dig[8 + n1] & 0xFFFF0000u);
            digits1[lane] = uint4{pk[0], pk[1], pk[2], pk[3]};
            digits1[64 + lane] = uint4{pk[4], pk[5], pk[6], pk[7]};
            wg_barrier(); // F: digits handed over
#pragma unroll
            for (int n1 = 0; n1 < 8; n1++) V[n1] = twisted_digit<LOGB>(dig[n1], dig[8 + n1], 0, twist[n1]);
        } else {
            STAMP8(0);
            if constexpr (MIX) wg_barrier(); // A
            c64 twist[8];
#pragma unroll
            for (int n1 = 0; n1 < 8; n1++) twist[n1] = twist_lds[64 * n1];
            wg_barrier(); // F
            STAMP8(1);
            const uint4 lo = digits1[lane], hi = digits1[64 + lane];
            const uint32_t pk[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
            for (int n1 = 0; n1 < 8; n1++)
                V[n1] = cmul_nf({(double)(((int)(pk[n1] << 16)) >> 16), (double)(((int)pk[n1]) >> 16)}, twist[n1]);
        }
        STAMP8(2);
        fft512_single<+1, 7>(V, mine, tab, lane);
        STAMP8(3);
        // radix-2 stage across the parities
#pragma unroll
        for (int i = 0; i < 4; i++)
            reinterpret_cast<c64*>(mine)[i * 64 + lane] = {w == 0 ? V[4 + i].re : V[i].re, w == 0 ? V[4 + i].im : V[i].im};
        wg_barrier(); // B
        STAMP8(4);
        {
            c64 xin[4], wc[4];
#pragma unroll
            for (int i = 0; i < 4; i++) xin[i] = reinterpret_cast<const c64*>(partner)[i * 64 + lane];
#pragma unroll
            for (int i = 0; i < 4; i++) wc[i] = wc_lds[64 * i];
            sched_fence();
            c64 X[8];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const c64 in = xin[i];
                const c64 Ei = {w == 0 ? V[i].re : in.re, w == 0 ? V[i].im : in.im};
                const c64 Oi = {w == 0 ? in.re : V[4 + i].re, w == 0 ? in.im : V[4 + i].im};
                c64 t = cmul_tw<+1>(Oi, wc[i]);
                X[i] = cadd(Ei, t);
                X[i + 4] = csub(Ei, t);
            }
#pragma unroll
            for (int r = 0; r < 8; r++) reinterpret_cast<c64*>(spectra(w, h))[(j * 8 + r) * 64 + lane] = X[r];
        }
        STAMP8(5);
        wg_barrier(); // C: all four transforms of parity w are in the regions (the gathers from them ended before B)

        STAMP8(6);
        // ---- multiply-accumulate, this wave's four bins of output polynomial h, chain order of glwe_ggsw_mad
        c64 P[4];
#pragma unroll
        for (int q = 0; q < 4; q++) P[q] = {0.0, 0.0};
        {
            c64 X[2][2][4];
#pragma unroll
            for (int p = 0; p < 2; p++)
#pragma unroll
                for (int jj = 0; jj < 2; jj++)
#pragma unroll
                    for (int q = 0; q < 4; q++)
                        X[p][jj][q] = reinterpret_cast<const c64*>(spectra(w, p))[(jj * 8 + 2 * j + (q & 1) + 4 * (q >> 1)) * 64 + lane];
            sched_fence();
#pragma unroll
            for (int p = 0; p < 2; p++)
#pragma unroll
                for (int jj = 0; jj < 2; jj++)
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const c64 k = key[p][jj][q];
                        const c64 x = X[p][jj][q];
                        double re = __builtin_fma(k.re, x.re, P[q].re);
                        double im = __builtin_fma(k.re, x.im, P[q].im);
                        P[q].re = __builtin_fma(-k.im, x.im, re);
                        P[q].im = __builtin_fma(k.im, x.re, im);
                    }
        }
        key_next = key_base + (size_t)(step + 1) * (2 * L) * (2 * kHalf);
        if constexpr (!LAST) SPF_KEY_PIECE(0);
        // ---- inverse split; every wave posts its two E' and two O' values into the inboxes of the two waves that transform
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const c64 Ep = cadd(P[i], P[2 + i]);
            const c64 Op = cmul_tw<-1>(csub(P[i], P[2 + i]), wc_lds[64 * (2 * j + i)]);
            reinterpret_cast<c64*>(image(0, h, 0))[(w * 4 + 2 * j + i) * 64 + lane] = Ep;
            reinterpret_cast<c64*>(image(1, h, 0))[(w * 4 + 2 * j + i) * 64 + lane] = Op;
        }
        STAMP8(7);
        wg_barrier(); // D
        STAMP8(8);
        if constexpr (!LAST) SPF_KEY_PIECE(1);
        if constexpr (J == 0) {
            c64 U[8];
#pragma unroll
            for (int r = 0; r < 8; r++) U[r] = reinterpret_cast<const c64*>(mine)[r * 64 + lane];
            sched_fence();
            if constexpr (!LAST) SPF_KEY_PIECE(2);
            STAMP8(9);
            fft512_single<-1, 7>(U, mine, tab, lane); // its exchanges follow the inbox reads in this wave's own LDS queue
            STAMP8(10);
            if constexpr (!LAST) SPF_KEY_PIECE(3);
            if constexpr (NEG) {
                untwist_sub_from_negated<SPF_BSK_PRESCALED, true>(U, twist_lds, acc);
            } else {
                uint64_t t[16];
                untwist_to_torus_bits<false, SPF_BSK_PRESCALED>(U, twist_lds, t);
#pragma unroll
                for (int e = 0; e < 16; e++) acc[e] += t[e];
            }
            STAMP8(11);
        } else {
            if constexpr (!LAST) { SPF_KEY_PIECE(2); SPF_KEY_PIECE(3); }
        }
    };
    for (uint32_t step = 0; step + 1 < a.n; step++) cmux_step(step, std::false_type{});
    cmux_step(a.n - 1, std::true_type{});
#ifdef SPF_STAMPS
    if (a.stamps && lane == 0) {
#pragma unroll
        for (int i = 0; i < 12; i++) a.stamps[((size_t)blockIdx.x * 8 + wv) * 16 + i] = st_acc[i];
    }
#endif
#undef STAMP8
#undef SPF_KEY_PIECE
    if constexpr (J == 0) {
        if constexpr (NEG) {
#pragma unroll
            for (int e = 0; e < 16; e++) acc[e] = (uint64_t)0 - acc[e];
        }
        uint64_t* out = a.out + (size_t)ct * a.out_stride;
        if (!a.sample_extract) {
#pragma unroll
            for (int e = 0; e < 16; e++) out[h * kN + coef2(e)] = acc[e];
        } else {
#pragma unroll
            for (int e = 0; e < 16; e++) {
                int c = coef2(e);
                if (h == 0) {
                    if (c == 0) out[0] = acc[e]; else out[kN - c] = (uint64_t)0 - acc[e];
                } else if (c == 0) {
                    out[kN] = acc[e];
                }
            }
        }
    }
}

// one copy of the body per (parity, digit)
template <int L, int LOGB, int MIX = 1>
__global__ __launch_bounds__(512) void blind_rotate8_kernel(BlindRotateArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wv >> 2) {
        if (wv & 1) blind_rotate8_body<L, LOGB, 1, 1, MIX>(a, smem);
        else blind_rotate8_body<L, LOGB, 0, 1, MIX>(a, smem);
    } else {
        if (wv & 1) blind_rotate8_body<L, LOGB, 1, 0, MIX>(a, smem);
        else blind_rotate8_body<L, LOGB, 0, 0, MIX>(a, smem);
    }
}

// ------------------------------------------------------------------------------------------
// cmux_kernel: batched `cmux` (ops/fft_ops.rs:149-181) with a per-ciphertext GGSW selector,
//   out = d0 + IFFT( sum_{p,j} FFT(digit_j(d1 - d0)_p) . GGSW[p][L-1-j] ),
// the operation `KeylessEvaluation::cmux` performs for every gate of a CMUX tree (GGSW in
// cbs_radix shape, L = 4 digits of 4 bits at DEFAULT_128).  Same two-waves-per-ciphertext
// arithmetic as blind_rotate2p_kernel (one step, no rotation), but every ciphertext brings its own
// 2*L*2 polynomials of key (256 KiB at L = 4), read exactly once straight from HBM into registers
// (streaming loads): algorithmic traffic 256 KiB + 3 x 32 KiB per CMUX makes this kernel HBM-bound.
// tools/microbench/ggsw_read_patterns.hip replays exactly this traffic without any arithmetic: 0.26 ms per
// 4096 gates (5.65 TB/s; the 9 % of the bytes that are WRITES cost a quarter of that time), against
// 0.295 ms for the kernel — 88 % of its own traffic ceiling.
struct CmuxArgs {
    const c64* ggsw;      // B x [2][L][2][1024]
    const uint64_t* d0;   // B x 4096 (selected when the GGSW encrypts 0)
    const uint64_t* d1;   // B x 4096
    uint64_t* out;        // B x 4096
    const c64* tables;
    uint32_t B;           // work units (GLWE pairs)
    uint32_t per_ggsw;    // consecutive units sharing one GGSW: 1 for cmux, l for glev_cmux
    uint32_t d0_zero;     // 1: d0 is the zero ciphertext and is not read (multiply_glwe_ggsw)
    // Scattered operands (the graph executor, spf_graph.hpp): when non-null, unit u takes
    // {ggsw, d0 (null = zero ciphertext), d1, out} from ptrs[4u .. 4u+3] instead of the arrays above.
    const void* const* ptrs;
    uint64_t* stamps;     // diagnostic builds (-DSPF_STAMPS): per-phase cycle counts of cmux4_kernel, else null
};
constexpr int cmux_lds_bytes(int gates) { return kTableBytes + gates * kWaveBufBytes + 64; }

#ifndef SPF_CMUX_INV_PAIR
#define SPF_CMUX_INV_PAIR fft512_pair1
#endif
#ifndef SPF_CMUX_FWD_PRE
#define SPF_CMUX_FWD_PRE 0
#endif
template <int L, int LOGB, int G, int W, bool STREAM>
__device__ __forceinline__ void cmux_body(const CmuxArgs& a, char* smem)
{
    static_assert(L * LOGB <= 32, "packed digits need L*LOGB <= 32");
#ifdef SPF_STAMPS
    uint64_t st_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t st_prev = __builtin_amdgcn_s_memtime();
#define STAMPS_(i) do { uint64_t t_ = __builtin_amdgcn_s_memtime(); st_acc[i] += t_ - st_prev; st_prev = t_; } while (0)
#else
#define STAMPS_(i) do { } while (0)
#endif
    c64* tab = reinterpret_cast<c64*>(smem);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cslot = wv >> 1;
    constexpr int w = W; // sample parity: one copy of the body per parity (no branches that merge register arrays)
    char* tile = smem + kTableBytes + cslot * kWaveBufBytes;
    char* mine = tile + w * 8192;
    char* theirs = tile + (w ^ 1) * 8192;
    // hand-over between the two waves of a gate: every wave of the workgroup runs the same sequence, so a bare
    // s_barrier does it (r01's flat-polled word per pair: 0.366 vs 0.355 ms per 4096 gates)
    auto cmux_sync = [&]() { pair_barrier_w(); };
    table_image_dma<128 * G>(a.tables, smem, tid, wv); // first: it needs nothing but the kernel arguments
    const uint32_t ct_raw = blockIdx.x * G + cslot;
    const bool owns_output = ct_raw < a.B;
    const uint32_t ct = owns_output ? ct_raw : a.B - 1;
    const c64* ggsw;
    const uint64_t *d0, *d1;
    uint64_t* out_ct;
    bool d0_zero = a.d0_zero != 0;
    if (a.ptrs) {
        const void* const* t = a.ptrs + 4 * (size_t)ct;
        ggsw = static_cast<const c64*>(t[0]);
        d1 = static_cast<const uint64_t*>(t[2]);
        d0_zero = t[1] == nullptr;
        d0 = d0_zero ? d1 : static_cast<const uint64_t*>(t[1]);
        out_ct = static_cast<uint64_t*>(const_cast<void*>(t[3]));
    } else {
        ggsw = a.ggsw + (size_t)(ct / a.per_ggsw) * (2 * L * 2 * kHalf);
        d0 = a.d0 + (size_t)ct * 2 * kN;
        d1 = a.d1 + (size_t)ct * 2 * kN;
        out_ct = a.out + (size_t)ct * 2 * kN;
    }
    const gu64_cptr gd0 = global_view(d0), gd1 = global_view(d1);
    const gu64_ptr gout = global_view(out_ct);
    auto coef2 = [&](int e) -> int { return (e >> 3) * 1024 + (e & 7) * 128 + 2 * lane + w; };
#ifdef SPF_STAMPS
    asm volatile("" :: "v"(gd0), "v"(gd1), "v"(gout)); // (diagnostic: the pointers are in)
    STAMPS_(12);
#endif

    // All 64 operand words are requested before anything else happens (left to itself hipcc keeps about fourteen loads
    // in flight and decomposes one value per round trip: 19 us of the 62 us a gate spent in this kernel), the twiddle
    // image is on its way by LDS-DMA since the top, and the decomposition starts when the words are in.
    // (r05, measured and not kept: the words as 16-byte loads of coefficient PAIRS, each wave decomposing both parities of
    // half a polynomial and handing the other parity's digits to its partner through the tile — half the loads, every line
    // touched once; the issue of the loads went from 6.2 k to 4.8 k cycles, the hand-over and the longer decomposition cost
    // more: 46.0 -> 48.0 ms per four 32 x 32 multiplications, batches unchanged.)
    uint64_t x1[2][16], x0[2][16];
#pragma unroll
    for (int p = 0; p < 2; p++)
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const int c = p * kN + coef2(e);
            x1[p][e] = gd1[c];
            x0[p][e] = gd0[c]; // d0 aliases d1 when it is the zero ciphertext: no branch around the load
        }
    sched_fence();
    STAMPS_(13);
    uint32_t dig[2][16];
#pragma unroll
    for (int p = 0; p < 2; p++)
#pragma unroll
        for (int e = 0; e < 16; e++) {
            uint64_t diff = x1[p][e] - (d0_zero ? 0 : x0[p][e]); // sub_glwe_ciphertexts(diff, d_1, d_0) (fft_ops.rs:168)
            constexpr int shift = 64 - L * LOGB;
            uint32_t s = (uint32_t)(diff >> shift) + (uint32_t)((diff >> (shift - 1)) & 1);
            uint32_t packed = 0;
#pragma unroll
            for (int j = 0; j < L; j++) {
                uint32_t d = s & ((1u << LOGB) - 1);
                s >>= LOGB;
                s += d >> (LOGB - 1);
                packed |= d << (j * LOGB);
            }
            dig[p][e] = packed;
        }
    STAMPS_(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // my pieces of the twiddle image have landed (LDS-DMA counts on vmcnt)
    __syncthreads(); // twiddle image ready
    STAMPS_(1);

    const c64* twist = tab + kTWOff + w * 512 + lane;
    const c64* wc = tab + kWCOff + 256 * w + lane;
    c64 prod[2][8];
#pragma unroll
    for (int q = 0; q < 2; q++)
#pragma unroll
        for (int r = 0; r < 8; r++) prod[q][r] = {0.0, 0.0};

    // this wave's bins of key row (p, L-1-j), both output polynomials (2 x 8 KiB per wave), are requested one
    // round ahead: k0 of round m+1 right after round m's first MAD has consumed k0, k1 after the second, so a
    // wave keeps 8-16 KiB of selector in flight through the whole transform instead of waiting on each row
    const gc64_ptr gkey = global_view(ggsw) + 256 * w + lane;
    auto key_row = [&](int m) -> gc64_ptr {
        const int p = m / L, j = m - p * L;
        return gkey + (size_t)((p * L + (L - 1 - j)) * 2) * kHalf;
    };
    // STREAM: the launch's selectors exceed the 256 MB Infinity Cache and are read once — streaming loads (0.317 -> 0.298 ms per
    // 4096 gates); below that, repeated selectors and cache-resident ones are better served by plain loads (0.041 -> 0.038 per 512)
    auto key_load = [&](gc64_ptr p) -> c64 { return STREAM ? gload_stream(p) : gload(p); };
    c64 k0[8], k1[8];
    {
        const gc64_ptr row = key_row(0);
#pragma unroll
        for (int r = 0; r < 8; r++) k0[r] = key_load(row + 64 * (r & 3) + 512 * (r >> 2));
#pragma unroll
        for (int r = 0; r < 8; r++) k1[r] = key_load(row + kHalf + 64 * (r & 3) + 512 * (r >> 2));
    }
    // One round = one digit transform and its two MAD rows.  The LAST round is compiled as its own copy without the requests
    // for a next row pair (until r03 it re-requested its own rows — dead loads, waited for behind the loop with the registers
    // held; a branch around the requests inside ONE copy of the round makes hipcc wait for every row right where it is
    // requested, 0.31 -> 0.61 ms per 4096): no kernel of the library issues a load nobody consumes.
    auto round = [&](int m, auto last_c) {
        constexpr bool LAST = decltype(last_c)::value;
        const int p = m / L, j = m - p * L, sh = j * LOGB;
        const gc64_ptr next = key_row(LAST ? m : m + 1);
        c64 V[8];
#pragma unroll
        for (int n1 = 0; n1 < 8; n1++) {
            uint32_t wre = p ? dig[1][n1] : dig[0][n1];
            uint32_t wim = p ? dig[1][8 + n1] : dig[0][8 + n1];
            int dre = ((int)(wre << (32 - LOGB - sh))) >> (32 - LOGB);
            int dim = ((int)(wim << (32 - LOGB - sh))) >> (32 - LOGB);
            V[n1] = cmul_nf({(double)dre, (double)dim}, twist[64 * n1]);
        }
        STAMPS_(2);
        if (m > 0) cmux_sync(); // partner is done with my last cross data
        STAMPS_(3);
        fft512_single<+1, SPF_CMUX_FWD_PRE>(V, mine, tab, lane);
        STAMPS_(4);
        c64 Ei[4], Oi[4];
        if (w == 0) {
#pragma unroll
            for (int i = 0; i < 4; i++) reinterpret_cast<c64*>(mine)[i * 64 + lane] = V[4 + i];
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++) reinterpret_cast<c64*>(mine)[i * 64 + lane] = V[i];
        }
        cmux_sync();
        STAMPS_(5);
        if (w == 0) {
#pragma unroll
            for (int i = 0; i < 4; i++) { Ei[i] = V[i]; Oi[i] = reinterpret_cast<const c64*>(theirs)[i * 64 + lane]; }
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++) { Ei[i] = reinterpret_cast<const c64*>(theirs)[i * 64 + lane]; Oi[i] = V[4 + i]; }
        }
        c64 X[8];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            c64 t = cmul_tw<+1>(Oi[i], wc[64 * i]);
            X[i] = cadd(Ei[i], t);
            X[i + 4] = csub(Ei[i], t);
        }
        STAMPS_(6);
#ifdef SPF_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // diagnostic: time spent waiting for this round's selector rows
        STAMPS_(7);
#endif
#pragma unroll
        for (int r = 0; r < 8; r++) {
            double re = __builtin_fma(k0[r].re, X[r].re, prod[0][r].re);
            double im = __builtin_fma(k0[r].re, X[r].im, prod[0][r].im);
            prod[0][r].re = __builtin_fma(-k0[r].im, X[r].im, re);
            prod[0][r].im = __builtin_fma(k0[r].im, X[r].re, im);
        }
        if constexpr (!LAST) {
#pragma unroll
            for (int r = 0; r < 8; r++) k0[r] = key_load(next + 64 * (r & 3) + 512 * (r >> 2));
        }
#pragma unroll
        for (int r = 0; r < 8; r++) {
            double re = __builtin_fma(k1[r].re, X[r].re, prod[1][r].re);
            double im = __builtin_fma(k1[r].re, X[r].im, prod[1][r].im);
            prod[1][r].re = __builtin_fma(-k1[r].im, X[r].im, re);
            prod[1][r].im = __builtin_fma(k1[r].im, X[r].re, im);
        }
        if constexpr (!LAST) {
#pragma unroll
            for (int r = 0; r < 8; r++) k1[r] = key_load(next + kHalf + 64 * (r & 3) + 512 * (r >> 2));
        }
        STAMPS_(8);
    };
#pragma unroll 1
    for (int m = 0; m < 2 * L - 1; m++) round(m, std::false_type{});
    {
        // (the round index goes through an opaque move: as a literal it lets hipcc extract the last round's digits in the
        // prologue and carry them across the loop — 24 B of scratch, and a kernel of thousands of short workgroups pays
        // for every byte of scratch it declares)
        int m_last = 2 * L - 1;
        asm volatile("" : "+s"(m_last));
        round(m_last, std::true_type{});
    }

    // ---- both output polynomials back to the torus as ONE transform pair (`fft512_pair1`: each exchange of one
    // transform travels under a butterfly pass of the other; one cross exchange, three hand-overs instead of five)
    // add_glwe_ciphertexts(c, prod, d_0) (fft_ops.rs:180): d_0 is re-read rather than held across the transforms,
    // all words requested here so they land under the inverse transforms.  The lane index goes through an opaque
    // move first: otherwise the 32 output / d_0 addresses are computed in the prologue and parked in scratch, and a
    // kernel of thousands of short workgroups pays for every byte of scratch it declares (0.32 -> 0.40 ms per 4096).
    int lane_late = lane;
    asm volatile("" : "+v"(lane_late));
    auto coef2_late = [&](int e) -> int { return (e >> 3) * 1024 + (e & 7) * 128 + 2 * lane_late + w; };
    uint64_t d0w[2][16];
#pragma unroll
    for (int e = 0; e < 16; e++) d0w[0][e] = gd0[coef2_late(e)];
    c64 WW[2][8];
#pragma unroll
    for (int q = 0; q < 2; q++)
#pragma unroll
        for (int i = 0; i < 4; i++) {
            WW[q][i] = cadd(prod[q][i], prod[q][i + 4]);                             // Ep: kept by wave 0
            WW[q][4 + i] = cmul_tw<-1>(csub(prod[q][i], prod[q][i + 4]), wc[64 * i]); // Op: kept by wave 1
        }
    cmux_sync(); // partner is done with my last cross data
    if constexpr (w == 0) {
#pragma unroll
        for (int q = 0; q < 2; q++)
#pragma unroll
            for (int i = 0; i < 4; i++) reinterpret_cast<c64*>(mine)[(q * 4 + i) * 64 + lane] = WW[q][4 + i];
    } else {
#pragma unroll
        for (int q = 0; q < 2; q++)
#pragma unroll
            for (int i = 0; i < 4; i++) reinterpret_cast<c64*>(mine)[(q * 4 + i) * 64 + lane] = WW[q][i];
    }
    cmux_sync();
    if constexpr (w == 0) {
#pragma unroll
        for (int q = 0; q < 2; q++)
#pragma unroll
            for (int i = 0; i < 4; i++) WW[q][4 + i] = reinterpret_cast<const c64*>(theirs)[(q * 4 + i) * 64 + lane];
    } else {
#pragma unroll
        for (int q = 0; q < 2; q++)
#pragma unroll
            for (int i = 0; i < 4; i++) WW[q][i] = reinterpret_cast<const c64*>(theirs)[(q * 4 + i) * 64 + lane];
    }
    cmux_sync(); // both cross reads retired before either image is overwritten
    STAMPS_(9);
    SPF_CMUX_INV_PAIR<-1, 2>(WW[0], WW[1], mine, tab, lane);
    STAMPS_(10);
    // (the second polynomial's words only now: all 32 across the transform pair do not fit the registers, and a
    // spilled load waits for everything in flight; they land under the first polynomial's conversion)
#pragma unroll
    for (int e = 0; e < 16; e++) d0w[1][e] = gd0[kN + coef2_late(e)];
#pragma unroll
    for (int q = 0; q < 2; q++) {
        uint64_t t[16];
        untwist_to_torus_bits(WW[q], twist, t);
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const uint64_t v = (d0_zero ? 0 : d0w[q][e]) + t[e];
            if (owns_output) gout[q * kN + coef2_late(e)] = v;
        }
    }
    STAMPS_(11);
#ifdef SPF_STAMPS
    if (a.stamps && lane == 0) {
#pragma unroll
        for (int i = 0; i < 16; i++) a.stamps[((size_t)blockIdx.x * (2 * G) + wv) * 16 + i] = st_acc[i];
    }
#endif
#undef STAMPS_
}

template <int L, int LOGB, int G, bool STREAM = false>
__global__ __launch_bounds__(128 * G, 2) void cmux_kernel(CmuxArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) & 1) cmux_body<L, LOGB, G, 1, STREAM>(a, smem);
    else cmux_body<L, LOGB, G, 0, STREAM>(a, smem);

}

// ------------------------------------------------------------------------------------------
// cmux4_kernel: the LATENCY shape of cmux_kernel — four waves per gate, one gate per workgroup, for
// the levels of a gate graph that hold at most one gate per CU (a ripple-carry chain is 1-4 gates per
// level, and its depth, not its width, is what a run waits for).  Four waves per ciphertext (parity x polynomial):
// wave (w, h) = sample parity w x polynomial h.  Each pair of waves decomposes ONE polynomial of
// d1 - d0 and pushes its four digits through two `fft512_pair_pipelined`s; the pairs then publish their four
// transforms in LDS and wave (w, h) runs the whole accumulation chain of OUTPUT polynomial h over the
// eight rows in the reference's order (row polynomial 0 levels 3..0, then polynomial 1: fft_ops.rs:67-98
// reversed GLEV rows) — its own transforms for the rows of polynomial h, the sibling's for the others —
// and transforms that polynomial back.  The selector's rows of output polynomial h go straight from
// HBM/L2 into registers: the first four are requested before anything else, the last four behind the
// forward transforms.  Same operations in the same order on every value as cmux_kernel: same words.
constexpr int kCmux4Lds = kTableBytes + 4 * 4 * 8192;

template <int L, int LOGB, int W>
__device__ __forceinline__ void cmux4_body(const CmuxArgs& a, char* smem)
{
    static_assert(L == 4 && L * LOGB <= 32, "four digits, processed as two pairs");
    c64* tab = reinterpret_cast<c64*>(smem);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int w = W; // sample parity: one copy of the body per parity (see blind_rotate2p_kernel)
    const int h = wv >> 1;
#ifdef SPF_STAMPS
    uint64_t st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t st_prev = __builtin_amdgcn_s_memtime();
#define STAMPC(i) do { uint64_t t_ = __builtin_amdgcn_s_memtime(); st_acc[i] += t_ - st_prev; st_prev = t_; } while (0)
#else
#define STAMPC(i) do { } while (0)
#endif
    // region of wave (w, h): 32 KiB = two exchange images while transforming, then its four transforms
    auto region = [&](int ww, int hh) -> char* { return smem + kTableBytes + (hh * 2 + ww) * 32768; };
    char* mine = region(w, h);
    char* mineB = mine + 8192;
    const char* partner = region(w ^ 1, h);
    auto wg_barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // keeps the selector loads in flight (no vmcnt drain)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    const uint32_t ct = blockIdx.x; // grid = units
    // Load order = need order, because vmcnt retires in issue order: the twiddle image first (it does not depend on
    // the operand table), then d1 / d0 (the decomposition waits for them), the selector rows last (the accumulation
    // chain is far away) — requested the other way round the decomposition waited for 32 KiB of selector per wave.
    constexpr int kTabPerThread = (kTableEntries + 255) / 256;
    f64x2_t tab_img[kTabPerThread];
    {
        const f64x2_t* src = reinterpret_cast<const f64x2_t*>(a.tables);
#pragma unroll
        for (int i = 0; i < kTabPerThread; i++) {
            const int idx = tid + 256 * i;
            tab_img[i] = src[idx < kTableEntries ? idx : kTableEntries - 1];
        }
    }
    const c64* ggsw;
    const uint64_t *d0, *d1;
    uint64_t* out_ct;
    bool d0_zero = a.d0_zero != 0;
    if (a.ptrs) {
        const void* const* t = a.ptrs + 4 * (size_t)ct;
        ggsw = static_cast<const c64*>(t[0]);
        d1 = static_cast<const uint64_t*>(t[2]);
        d0_zero = t[1] == nullptr;
        d0 = d0_zero ? d1 : static_cast<const uint64_t*>(t[1]);
        out_ct = static_cast<uint64_t*>(const_cast<void*>(t[3]));
    } else {
        ggsw = a.ggsw + (size_t)(ct / a.per_ggsw) * (2 * L * 2 * kHalf);
        d0 = a.d0 + (size_t)ct * 2 * kN;
        d1 = a.d1 + (size_t)ct * 2 * kN;
        out_ct = a.out + (size_t)ct * 2 * kN;
    }
    auto coef2 = [&](int e) -> int { return (e >> 3) * 1024 + (e & 7) * 128 + 2 * lane + w; };
    // operand pointers may come from the per-gate table: pin them to global memory so that the loads count on
    // vmcnt only and stay in flight across the LDS waits and barriers of the transforms (see global_view)
    const gc64_ptr gkey = global_view(ggsw) + 256 * w + lane;
    const gu64_cptr gd0 = global_view(d0) + h * kN, gd1 = global_view(d1) + h * kN;
    const gu64_ptr gout = global_view(out_ct) + h * kN;
    // selector row (p, level L-1-j), output polynomial h, this wave's bins
    auto load_row = [&](c64 (&k)[8], int p, int j) {
        const gc64_ptr row = gkey + (size_t)((p * L + (L - 1 - j)) * 2 + h) * kHalf;
#pragma unroll
        for (int r = 0; r < 8; r++) k[r] = gload(row + 64 * (r & 3) + 512 * (r >> 2));
    };
    // d0 aliases d1 when it is the zero ciphertext, so its loads need no branch
    uint64_t x1[16], x0[16];
#pragma unroll
    for (int e = 0; e < 16; e++) x1[e] = gd1[coef2(e)];
#pragma unroll
    for (int e = 0; e < 16; e++) x0[e] = gd0[coef2(e)];
    compiler_fence();
    {
        f64x2_t* dst = reinterpret_cast<f64x2_t*>(smem);
#pragma unroll
        for (int i = 0; i < kTabPerThread; i++) {
            const int idx = tid + 256 * i;
            if (idx < kTableEntries) dst[idx] = tab_img[i];
        }
    }
    compiler_fence();
    c64 key0[L][8], key1[L][8];
#pragma unroll
    for (int j = 0; j < L; j++) load_row(key0[j], 0, j);
    STAMPC(0);
    wg_barrier(); // twiddle image in place
    STAMPC(1);
    c64 twist[8], wc[4];
#pragma unroll
    for (int n1 = 0; n1 < 8; n1++) twist[n1] = tab[kTWOff + w * 512 + lane + 64 * n1];
#pragma unroll
    for (int i = 0; i < 4; i++) wc[i] = tab[kWCOff + 256 * w + lane + 64 * i];

    STAMPC(2);
    uint32_t dig[16];
#pragma unroll
    for (int e = 0; e < 16; e++) // sub_glwe_ciphertexts(diff, d_1, d_0) (fft_ops.rs:168), then the gadget digits
        dig[e] = gadget_digits_packed<L, LOGB>(x1[e] - (d0_zero ? 0 : x0[e]));
    // ---- the four digit transforms of polynomial h, two at a time
    c64 X[L][8];
#pragma unroll
    for (int jj = 0; jj < L; jj += 2) {
#pragma unroll
        for (int j = jj; j < jj + 2; j++)
#pragma unroll
            for (int n1 = 0; n1 < 8; n1++) X[j][n1] = twisted_digit<LOGB>(dig[n1], dig[8 + n1], j, twist[n1]);
        if (jj) wg_barrier(); // partner is done with my last cross data
        fft512_pair_pipelined<+1>(X[jj], X[jj + 1], mine, mineB, tab, lane);
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int i = 0; i < 4; i++)
                reinterpret_cast<c64*>(mine)[(j * 4 + i) * 64 + lane] = {w == 0 ? X[jj + j][4 + i].re : X[jj + j][i].re,
                                                                         w == 0 ? X[jj + j][4 + i].im : X[jj + j][i].im};
        wg_barrier();
        c64 xin[2][4];
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int i = 0; i < 4; i++) xin[j][i] = reinterpret_cast<const c64*>(partner)[(j * 4 + i) * 64 + lane];
        sched_fence();
#pragma unroll
        for (int j = 0; j < 2; j++) {
            c64 Y[8];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const c64 in = xin[j][i];
                const c64 Ei = {w == 0 ? X[jj + j][i].re : in.re, w == 0 ? X[jj + j][i].im : in.im};
                const c64 Oi = {w == 0 ? in.re : X[jj + j][4 + i].re, w == 0 ? in.im : X[jj + j][4 + i].im};
                c64 t = cmul_tw<+1>(Oi, wc[i]);
                Y[i] = cadd(Ei, t);
                Y[i + 4] = csub(Ei, t);
            }
#pragma unroll
            for (int r = 0; r < 8; r++) X[jj + j][r] = Y[r];
        }
    }
    STAMPC(3);
    wg_barrier(); // cross reads retired: the regions can carry the transforms
    // spectra out, and behind each one — into the registers it frees — the matching one of the last four selector
    // rows: the 32 requests trickle into the vector-memory queue between the LDS stores instead of stalling in a block
#pragma unroll
    for (int j = 0; j < L; j++) {
#pragma unroll
        for (int r = 0; r < 8; r++) reinterpret_cast<c64*>(mine)[(j * 8 + r) * 64 + lane] = X[j][r];
        load_row(key1[j], 1, j);
    }
    wg_barrier(); // every wave's four transforms are in its region
    STAMPC(4);

    // ---- accumulation chain of output polynomial h: rows (0, j = 0..3) then (1, j = 0..3)
    c64 V[8];
#pragma unroll
    for (int r = 0; r < 8; r++) V[r] = {0.0, 0.0};
    {
        // row polynomial p: my own transforms when p == h, the sibling's otherwise — both read back from LDS, so that the 128
        // registers of X are free for the selector rows.  The eight spectrum values of row m + 1 are requested before the
        // FMAs of row m (r04: hipcc fetched them two at a time, each pair waited for on the spot — at one wave per SIMD
        // every such round trip is lost time).
        auto row_src = [&](int m) { return reinterpret_cast<const c64*>(smem + kTableBytes + (((m / L) * 2 + w) * 32768)) + ((m % L) * 8) * 64 + lane; };
        c64 sx[2][8];
#pragma unroll
        for (int r = 0; r < 8; r++) sx[0][r] = row_src(0)[r * 64];
#pragma unroll
        for (int m = 0; m < 2 * L; m++) {
            const int p = m / L, j = m % L;
            if (m + 1 < 2 * L) {
#pragma unroll
                for (int r = 0; r < 8; r++) sx[(m + 1) & 1][r] = row_src(m + 1)[r * 64];
            }
            sched_fence();
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const c64 k = {p == 0 ? key0[j][r].re : key1[j][r].re, p == 0 ? key0[j][r].im : key1[j][r].im};
                const c64 x = sx[m & 1][r];
                double re = __builtin_fma(k.re, x.re, V[r].re);
                double im = __builtin_fma(k.re, x.im, V[r].im);
                V[r].re = __builtin_fma(-k.im, x.im, re);
                V[r].im = __builtin_fma(k.im, x.re, im);
            }
        }
    }
    STAMPC(5);
    wg_barrier(); // sibling reads retired; regions free again

    // ---- polynomial h back to the torus, plus d0
    {
        c64 Ep[4], Op[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            Ep[i] = cadd(V[i], V[i + 4]);
            Op[i] = cmul_tw<-1>(csub(V[i], V[i + 4]), wc[i]);
        }
#pragma unroll
        for (int i = 0; i < 4; i++)
            reinterpret_cast<c64*>(mine)[i * 64 + lane] = {w == 0 ? Op[i].re : Ep[i].re, w == 0 ? Op[i].im : Ep[i].im};
        wg_barrier();
        c64 in4[4];
#pragma unroll
        for (int i = 0; i < 4; i++) in4[i] = reinterpret_cast<const c64*>(partner)[i * 64 + lane];
        sched_fence();
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const c64 in = in4[i];
            V[i] = {w == 0 ? Ep[i].re : in.re, w == 0 ? Ep[i].im : in.im};
            V[4 + i] = {w == 0 ? in.re : Op[i].re, w == 0 ? in.im : Op[i].im};
        }
        wg_barrier(); // cross reads retired before the image is overwritten
    }
    // add_glwe_ciphertexts(c, prod, d_0) (fft_ops.rs:180): d_0 re-read under the inverse transform
    uint64_t d0w[16];
#pragma unroll
    for (int e = 0; e < 16; e++) d0w[e] = gd0[coef2(e)];
    STAMPC(6);
    fft512_single<-1, 7>(V, mine, tab, lane);
    uint64_t t[16];
    untwist_to_torus_bits(V, twist, t);
#pragma unroll
    for (int e = 0; e < 16; e++) gout[coef2(e)] = (d0_zero ? 0 : d0w[e]) + t[e];
    STAMPC(7);
#ifdef SPF_STAMPS
    if (a.stamps && lane == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // count the stores' drain
        const uint64_t t_end = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int i = 0; i < 8; i++) a.stamps[((size_t)blockIdx.x * 4 + wv) * 16 + i] = st_acc[i];
        a.stamps[((size_t)blockIdx.x * 4 + wv) * 16 + 8] = t_end - st_prev;
    }
#endif
#undef STAMPC
}

template <int L, int LOGB>
__global__ __launch_bounds__(256, 1) void cmux4_kernel(CmuxArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) & 1) cmux4_body<L, LOGB, 1>(a, smem);
    else cmux4_body<L, LOGB, 0>(a, smem);
}

// ------------------------------------------------------------------------------------------
// LWE keyswitch L1 -> L0 (ops/keyswitch/lwe_keyswitch.rs:23-62; lev_ciphertext_ops.rs:18-42;
// lwe_ciphertext_ops.rs:48-66), batched: out[ct] = (0,..,0,b) - sum_i sum_j d_{i,j} KSK[i][l-1-j].
// A workgroup owns a tile of KS_CT ciphertexts x 256 output columns; each thread owns one
// column and keeps KS_CT 64-bit accumulators in registers, so every key word fetched (coalesced
// across the 256 columns) is used KS_CT times.  Digits are wave-uniform, computed on the
// scalar unit from the input mask word.
constexpr int KS_CT = 32;

struct KeyswitchArgs {
    const uint64_t* in;  // B x (n_in+1)
    const uint64_t* ksk; // [n_in][count][n_out+1]
    uint64_t* out;       // B x (n_out+1)
    uint32_t n_in, n_out, B;
    uint32_t radix_log, count;
};

__global__ __launch_bounds__(256) void keyswitch_kernel(KeyswitchArgs a)
{
    const uint32_t w = a.n_out + 1;
    const uint32_t col = blockIdx.x * 256 + threadIdx.x;
    const uint32_t ct0 = blockIdx.y * KS_CT;
    const bool live = col < w;
    const uint32_t colc = live ? col : 0;
    uint64_t sum[KS_CT];
#pragma unroll
    for (int t = 0; t < KS_CT; t++) sum[t] = 0;

    const uint32_t shift = 64 - a.radix_log * a.count;
    const uint32_t mask = (1u << a.radix_log) - 1;
    for (uint32_t i = 0; i < a.n_in; i++) {
        const uint64_t* lev = a.ksk + (size_t)i * a.count * w + colc;
        uint32_t st[KS_CT];
#pragma unroll
        for (int t = 0; t < KS_CT; t++) {
            uint32_t ct = ct0 + t < a.B ? ct0 + t : a.B - 1;
            uint64_t x = a.in[(size_t)ct * (a.n_in + 1) + i];
            st[t] = (uint32_t)(x >> shift) + (uint32_t)((x >> (shift - 1)) & 1);
        }
        for (uint32_t j = 0; j < a.count; j++) {
            uint64_t kv = lev[(size_t)(a.count - 1 - j) * w];
#pragma unroll
            for (int t = 0; t < KS_CT; t++) {
                uint32_t d = st[t] & mask;
                st[t] >>= a.radix_log;
                uint32_t carry = d >> (a.radix_log - 1);
                st[t] += carry;
                int64_t digit = (int64_t)d - ((int64_t)carry << a.radix_log);
                sum[t] += kv * (uint64_t)digit;
            }
        }
    }
    if (!live) return;
#pragma unroll
    for (int t = 0; t < KS_CT; t++) {
        uint32_t ct = ct0 + t;
        if (ct < a.B) {
            uint64_t base = (col == a.n_out) ? a.in[(size_t)ct * (a.n_in + 1) + a.n_in] : 0;
            a.out[(size_t)ct * w + col] = base - sum[t];
        }
    }
}

// ------------------------------------------------------------------------------------------
// Keyswitch as an int8 block-GEMM on the matrix cores.  The batched keyswitch
//   out[b] = (0,..,0,body_b) - sum_k d[b][k] * KSK_k          (k = i*count + j, wrapping u64)
// is a dense [B x K] x [K x (n_out+1)] product whose left factor holds tiny signed digits.
// Each 64-bit key word is split into its 8 byte planes, stored signed as (byte - 128) so they fit
// v_mfma_i32_32x32x32_i8; the exact integer identity
//   sum_k d_k * word_k = sum_t 2^(8t) * ( sum_k d_k * (byte_{k,t} - 128)  +  128 * sum_k d_k )
// restores the product: the inner sums are the GEMM (int32 accumulators cannot overflow:
// K * 2^(radix_log-1) * 128 < 2^31), the correction needs only each ciphertext's digit sum, and
// the recombination happens in the epilogue in wrapping 64-bit arithmetic.  Results are identical,
// bit for bit, to the scalar definition (integer arithmetic is exact).
//   A  [Mpad][K]  int8   digits, K-contiguous            (ks_digits_kernel, per call)
//   Bt [Npad][K]  int8   key byte planes, n = 8*col + t   (ks_planes_kernel, once per key)
typedef int v4i32 __attribute__((ext_vector_type(4)));
typedef int v16i32 __attribute__((ext_vector_type(16)));

constexpr int KSG_TILE = 128; // block tile (M and N); 4 waves, each 64 x 64

// digits of every mask word: round (radix.rs:157-162) then vector_next_decomp (scalar.rs:52-71)
__global__ __launch_bounds__(256) void ks_digits_kernel(const uint64_t* in, int8_t* dig, int* rowsum,
                                                        uint32_t n_in, uint32_t B, uint32_t radix_log,
                                                        uint32_t count)
{
    const uint32_t ct = blockIdx.x;
    if (ct >= B) return; // padded rows stay zero (buffer is cleared by the host)
    const uint64_t* x = in + (size_t)ct * (n_in + 1);
    int8_t* d = dig + (size_t)ct * n_in * count;
    const uint32_t shift = 64 - radix_log * count;
    const uint64_t mask = ((uint64_t)1 << radix_log) - 1;
    int local = 0;
    for (uint32_t i = threadIdx.x; i < n_in; i += 256) {
        uint64_t v = x[i];
        uint64_t st = (v >> shift) + ((v >> (shift - 1)) & 1);
        for (uint32_t j = 0; j < count; j++) {
            uint64_t dg = st & mask;
            st >>= radix_log;
            uint64_t carry = dg >> (radix_log - 1);
            st += carry;
            int digit = (int)dg - (int)(carry << radix_log);
            d[(size_t)i * count + j] = (int8_t)digit;
            local += digit;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o);
    if ((threadIdx.x & 63) == 0) atomicAdd(&rowsum[ct], local);
}

// key byte planes: Bt[8*col + t][i*count + j] = byte t of KSK[i][count-1-j][col], minus 128
// (LEV rows are consumed in reverse, lev_ciphertext_ops.rs:36).  Rows past the key stay zero.
__global__ __launch_bounds__(256) void ks_planes_kernel(const uint64_t* ksk, int8_t* bt, uint32_t n_in,
                                                        uint32_t w, uint32_t count, size_t K)
{
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; // over (i, j, col), col fastest
    const size_t total = (size_t)n_in * count * w;
    if (idx >= total) return;
    const uint32_t col = (uint32_t)(idx % w);
    const size_t ij = idx / w;
    const uint32_t jrow = (uint32_t)(ij % count), i = (uint32_t)(ij / count);
    const uint64_t word = ksk[idx];
    const size_t k = (size_t)i * count + (count - 1 - jrow);
#pragma unroll
    for (int t = 0; t < 8; t++)
        bt[((size_t)col * 8 + t) * K + k] = (int8_t)((int)((word >> (8 * t)) & 0xFF) - 128);
}

struct KsGemmArgs {
    const int8_t* A;    // [Mpad][K]
    const int8_t* Bt;   // [Npad][K]
    const int* rowsum;  // [Mpad]
    const uint64_t* in; // B x (n_in+1), for the body word
    uint64_t* out;      // B x (n_out+1)
    uint32_t B, n_in, n_out, K;
};

// epilogue of the GEMM kernel
__device__ __forceinline__ void ks_gemm_epilogue(const KsGemmArgs& a, const v16i32& acc00, const v16i32& acc01, const v16i32& acc10,
                                                 const v16i32& acc11, uint32_t m0, uint32_t n0, int r, int h)
{
    // epilogue: C/D map of the 32x32 forms: row = (reg&3) + 8*(reg>>2) + 4*h, col = r.
    // Column n = 8*word + t: shift plane t into place and add the 8 lanes of a word.
    const uint32_t w = a.n_out + 1;
    const int t = r & 7;
    auto finish = [&](const v16i32& acc, uint32_t mbase, uint32_t nbase) {
#pragma unroll
        for (int reg = 0; reg < 16; reg++) {
            const uint32_t row = mbase + (reg & 3) + 8 * (reg >> 2) + 4 * h;
            const long long s = (long long)acc[reg] + 128ll * (long long)a.rowsum[row];
            unsigned long long v = (unsigned long long)s << (8 * t);
            v += __shfl_xor(v, 1);
            v += __shfl_xor(v, 2);
            v += __shfl_xor(v, 4);
            const uint32_t word = (nbase + r) >> 3;
            if (t == 0 && row < a.B && word < w) {
                const uint64_t body = (word == a.n_out) ? a.in[(size_t)row * (a.n_in + 1) + a.n_in] : 0;
                a.out[(size_t)row * w + word] = body - v;
            }
        }
    };
    finish(acc00, m0, n0);
    finish(acc01, m0, n0 + 32);
    finish(acc10, m0 + 32, n0);
    finish(acc11, m0 + 32, n0 + 32);
}

// ks_gemm_lds_kernel: the block-GEMM with the operand tiles staged through LDS.  v_mfma_i32_32x32x32_i8: lane (r, h) supplies
// 16 k values of A row r and of B column r per instruction; any assignment of k values to (h, position) that is the same for
// both operands leaves the dot products unchanged.  A workgroup (4 waves, 128 x 128
// tile, 64 x 64 per wave) brings each 128-row x 128-byte slab of A and of Bt into LDS ONCE per round (LDS-DMA, 1 KiB
// per wave-instruction, double-buffered: the slabs of round i+1 land under the 16 MFMAs per wave of round i) instead
// of every wave fetching its own rows from L2 (r01's direct form, 0.886 against 0.337 ms per 4096, removed in r04): half the
// vector-memory traffic, which is what bounded that form (64 B/clk/CU against 128 B/clk needed at full MFMA rate).  Image: 16-byte slot of (row, chunk) =
// 8 row + (chunk ^ ((row >> 1) & 7)) — the DMA writes slots linearly and chooses WHICH chunk each lane fetches, the
// reads of a 16-lane ds_read_b128 group (16 rows, one chunk) hit 16 distinct slots mod 16: conflict-free.
// Two workgroups per CU (64 KiB of LDS each).
constexpr int kKsLdsBytes = 2 * 2 * KSG_TILE * 128;

__global__ __launch_bounds__(256, 2) void ks_gemm_lds_kernel(KsGemmArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const size_t K = a.K;
    const uint32_t tm0 = blockIdx.y * KSG_TILE, tn0 = blockIdx.x * KSG_TILE;
    // DMA duty of this wave: rows [32 wave, +32) of the A slab and of the B slab, 8 rows per piece
    uint32_t voffA[4], voffB[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t row = 32 * wave + 8 * j + (lane >> 3);
        const uint32_t chunk = (lane & 7) ^ ((row >> 1) & 7);
        voffA[j] = (uint32_t)((size_t)row * K + 16 * chunk);
        voffB[j] = voffA[j];
    }
    const char* gA = reinterpret_cast<const char*>(a.A) + (size_t)tm0 * K;
    const char* gB = reinterpret_cast<const char*>(a.Bt) + (size_t)tn0 * K;
    const uint32_t lds0 = lds_address(smem);
    auto dma_round = [&](size_t k0, int stage) {
        const uint32_t base = lds0 + stage * (2 * KSG_TILE * 128) + (32 * wave) * 128;
#pragma unroll
        for (int j = 0; j < 4; j++) lds_dma_piece(gA + k0, voffA[j], base + j * 1024);
#pragma unroll
        for (int j = 0; j < 4; j++) lds_dma_piece(gB + k0, voffB[j], base + KSG_TILE * 128 + j * 1024);
    };
    // fragment addresses: row of block blk = 64 wm + 32 blk + r (A) / 64 wn + 32 blk + r (B); chunk of step q = 2 q + h
    uint32_t rdA[2], rdB[2], swA[2], swB[2];
#pragma unroll
    for (int blk = 0; blk < 2; blk++) {
        const uint32_t ra = 64 * wm + 32 * blk + r, rb = 64 * wn + 32 * blk + r;
        rdA[blk] = ra * 128; swA[blk] = (ra >> 1) & 7;
        rdB[blk] = KSG_TILE * 128 + rb * 128; swB[blk] = (rb >> 1) & 7;
    }
    v16i32 acc00 = {0}, acc01 = {0}, acc10 = {0}, acc11 = {0};
    dma_round(0, 0);
    const size_t rounds = K / 128;
    for (size_t rd = 0; rd < rounds; rd++) {
        const int stage = (int)(rd & 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // my pieces of this round have landed
        __builtin_amdgcn_s_barrier();                     // ... everyone's; and everyone is done reading the other stage
        if (rd + 1 < rounds) dma_round((rd + 1) * 128, stage ^ 1);
        const char* st = smem + stage * (2 * KSG_TILE * 128);
        v4i32 fa[2][4], fb[2][4];
#pragma unroll
        for (int blk = 0; blk < 2; blk++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                fa[blk][q] = *reinterpret_cast<const v4i32*>(st + rdA[blk] + 16 * ((2 * q + h) ^ swA[blk]));
                fb[blk][q] = *reinterpret_cast<const v4i32*>(st + rdB[blk] + 16 * ((2 * q + h) ^ swB[blk]));
            }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            acc00 = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[0][q], fb[0][q], acc00, 0, 0, 0);
            acc01 = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[0][q], fb[1][q], acc01, 0, 0, 0);
            acc10 = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[1][q], fb[0][q], acc10, 0, 0, 0);
            acc11 = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[1][q], fb[1][q], acc11, 0, 0, 0);
        }
    }
    const uint32_t m0 = tm0 + wm * 64, n0 = tn0 + wn * 64;
    ks_gemm_epilogue(a, acc00, acc01, acc10, acc11, m0, n0, r, h);
}

// sample_extract (ops/ciphertext/glwe_ciphertext_ops.rs:31-76), k = 1, batched
__global__ void sample_extract_kernel(const uint64_t* glwe, uint64_t* lwe, uint32_t B, uint32_t h)
{
    const uint32_t ct = blockIdx.y;
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (ct >= B || j > kN) return;
    const uint64_t* g = glwe + (size_t)ct * 2 * kN;
    uint64_t* o = lwe + (size_t)ct * (kN + 1);
    if (j == kN) o[kN] = g[kN + h];
    else if (j <= h) o[j] = g[h - j];
    else o[j] = (uint64_t)0 - g[h + kN - j];
}

// The three linear `KeylessEvaluation` operations on L1 GLWE ciphertexts (crypto/evaluation.rs:47-66),
// one thread per coefficient, streaming (HBM-bound, 16-24 bytes per coefficient):
//   GLWE_NOT    out = in + trivial_one: the trivial GLWE of the polynomial 1 at one plaintext bit is
//               all zero except body coefficient 0 = 2^63 (encryption.rs:359-364, :132)
//   GLWE_XOR    out = a + b, wrapping (`add_glwe_ciphertexts`, glwe_ciphertext_ops.rs:79-99)
//   GLWE_MUL_XN out = in * X^n mod X^N + 1 on mask and body
//               (`rotate_glwe_positive_monomial_negacyclic`, blind_rotation.rs:126-135 ->
//               `mul_by_positive_monomial_negacyclic`, entities/polynomial.rs:208-236; n taken mod 2N)
enum : uint32_t { GLWE_NOT = 0, GLWE_XOR = 1, GLWE_MUL_XN = 2 };
template <uint32_t OP>
__global__ void glwe_linear_kernel(const uint64_t* a, const uint64_t* b, uint64_t* out, uint32_t B, uint32_t n)
{
    const uint32_t ct = blockIdx.y;
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; // 0 .. 2N-1: polynomial j / N, coefficient j % N
    if (ct >= B || j >= 2 * kN) return;
    const uint64_t* x = a + (size_t)ct * 2 * kN;
    uint64_t* o = out + (size_t)ct * 2 * kN;
    if constexpr (OP == GLWE_NOT) {
        o[j] = x[j] + (j == kN ? (uint64_t)1 << 63 : (uint64_t)0);
    } else if constexpr (OP == GLWE_XOR) {
        o[j] = x[j] + b[(size_t)ct * 2 * kN + j];
    } else {
        const uint32_t poly = j & ~(uint32_t)(kN - 1), i = j & (kN - 1);
        const uint32_t idx = (i + 2 * kN - n) & (2 * kN - 1); // exponent of the source term, mod 2N
        const uint64_t v = x[poly + (idx & (kN - 1))];
        o[j] = (idx & kN) ? (uint64_t)0 - v : v;
    }
}

// words u64 from `src` (pinned host memory, read by the GPU over PCIe, or device memory) to `dst`: the pool's copy-in.  A kernel
// on the batch's own stream instead of a hipMemcpyAsync: the runtime hands those to ONE in-order SDMA queue per direction pair,
// where a host-to-device copy of a new batch stood behind the device-to-host copy of another batch that was still waiting for
// its kernels — up to a whole bootstrap (profiles/r05_pool.md).
__global__ void copy_words_kernel(const uint64_t* __restrict__ src, uint64_t* __restrict__ dst, size_t words)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}


// spins for `ticks` of the 100 MHz constant-rate counter: the pool's probe of how many of its streams really run side by side
__global__ void spin_kernel(uint64_t ticks)
{
    const uint64_t t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

// Row gather for the graph executor: dst row r = the `words` u64 at src[r] (operands of one level of
// a gate graph live wherever their producers wrote them; the batched kernels want them contiguous).
__global__ void gather_rows_kernel(const uint64_t* const* src, uint64_t* dst, uint32_t rows, uint32_t words)
{
    const uint32_t r = blockIdx.y;
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows || j >= words) return;
    dst[(size_t)r * words + j] = src[r][j];
}

// lwe_rotate (ops/homomorphisms/lwe.rs:9-20) is folded into the blind-rotation kernels' body_rotate.

} // namespace spf
