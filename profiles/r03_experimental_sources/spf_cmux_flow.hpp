// cmux_flow_kernel: a whole run of CMUX levels of a gate graph in ONE launch, as dataflow.
//
// The level-by-level executor (spf_graph.hpp) launches one kernel per level: a level of w gates occupies w of the 256 CUs
// (narrow levels: most of a 32 x 32 multiplication's 628) or pays whole 1024-gate rounds of the streaming shape, and every
// level ends with a kernel boundary.  Here the gates of all consecutive CMUX-family levels form one list in level order;
// gridDim.x <= #CU persistent workgroups (four waves per gate, the arithmetic of cmux4_kernel, twiddle image copied once)
// take gates round-robin, and a gate starts as soon as ITS OWN operands are there: every gate has a completion word,
// written (release, agent scope) behind its output stores, and a consumer polls the words of its one or two producers
// (acquire) before it requests its operands.  A gate of level l + 1 therefore runs beside the stragglers of level l, and
// gates with slack fill the CUs that a narrow level leaves idle.
//
// Deadlock-free by construction: workgroup b takes gates b, b + G, b + 2G, ... in increasing order, producers have
// smaller indices than their consumers (level order), and all G workgroups are resident (one per CU, G <= #CU) — the
// in-flight gate with the smallest index never waits.
//
// EXPERIMENT, NOT PART OF THE LIBRARY (nothing includes this file; the host side — regions, dependency table, launch — is
// experimental/graph_flow_executor.patch against spf_graph.hpp / spf_hip.hip).  Same words as the level-by-level executor on
// all 23 gate-graph GPU tests, but SLOWER everywhere: a level of 64 / 256 / 1024 / 2048 gates 24.9 / 28.2 / 92.8 / 180.7 us
// against 20.0 / 23.1 / 53.0 / 96.1; 32-bit addition 6.2 ms (5.85), eight 8 x 8 multiplications 9.3 ms (8.8), four 32 x 32
// multiplications 63.4 ms (48.5).  One gate at a time per CU in the four-wave shape is 20-25 us per gate once operands and
// results bypass the L2s (agent-scope accesses) and the completion word has to travel; the streaming shape keeps four gates
// in flight per CU.  With release / acquire fences instead of agent-scope accesses every gate wrote back and invalidated a
// whole L2: 80 us per gate.  profiles/r03_experiments_blind_rotate.md, "r03f".
//
// Reference: the work replaces the rayon tasks of CircuitProcessor::run_graph_blocking, which also start a node as soon as
// its last operand completes (parasol_runtime/src/circuit_processor/mod.rs:130-253, 573-623); arithmetic: fft_ops.rs:149-181.
#pragma once
#include "../spf_kernels.hpp"

namespace spf {

struct CmuxFlowArgs {
    const void* const* ptrs;   // per unit {selector GGSW, d0 (null = zero ciphertext), d1, out}
    const int2* deps;          // per unit: index of the unit producing d0 / d1 inside this list, -1 = ready before the launch
    uint32_t* done;            // per unit completion word, zeroed before the launch; a finished unit holds `epoch`
    const c64* tables;
    uint32_t n_units;
    uint32_t epoch;
};

__device__ __forceinline__ uint64_t coherent_load(gu64_cptr p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void coherent_store(gu64_ptr p, uint64_t v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int L, int LOGB, int W>
__device__ __forceinline__ void cmux_flow_body(const CmuxFlowArgs& a, char* smem)
{
    static_assert(L == 4 && L * LOGB <= 32, "four digits, processed as two pairs");
    c64* tab = reinterpret_cast<c64*>(smem);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int w = W;
    const int h = wv >> 1;
    auto region = [&](int ww, int hh) -> char* { return smem + kTableBytes + (hh * 2 + ww) * 32768; };
    char* mine = region(w, h);
    char* mineB = mine + 8192;
    const char* partner = region(w ^ 1, h);
    auto wg_barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    {
        const double2* src = reinterpret_cast<const double2*>(a.tables);
        double2* dst = reinterpret_cast<double2*>(smem);
        for (int i = tid; i < kTableEntries; i += 256) dst[i] = src[i];
    }
    __syncthreads();
    for (uint32_t unit = blockIdx.x; unit < a.n_units; unit += gridDim.x) {
        const uint32_t ct = __builtin_amdgcn_readfirstlane(unit);
        const void* const* row = a.ptrs + 4 * (size_t)ct;
        const c64* ggsw = static_cast<const c64*>(row[0]);
        const uint64_t* d1 = static_cast<const uint64_t*>(row[2]);
        const bool d0_zero = row[1] == nullptr;
        const uint64_t* d0 = d0_zero ? d1 : static_cast<const uint64_t*>(row[1]);
        uint64_t* out_ct = static_cast<uint64_t*>(const_cast<void*>(row[3]));
        // The producers of my operands: poll their completion words.  No cache maintenance anywhere: results are written and
        // operands read with agent-scope accesses (they go to the level that is coherent across the eight XCDs' L2s), the
        // words likewise; a wave's loads are issued behind the poll that lets it pass and return in order.  (With release /
        // acquire fences instead every gate wrote back and invalidated a whole L2: 80 us per gate.)
        const int2 dep = a.deps[ct];
        if (dep.x >= 0)
            while (__hip_atomic_load(a.done + dep.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != a.epoch) __builtin_amdgcn_s_sleep(2);
        if (dep.y >= 0)
            while (__hip_atomic_load(a.done + dep.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != a.epoch) __builtin_amdgcn_s_sleep(2);
        asm volatile("" ::: "memory");
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
        for (int e = 0; e < 16; e++) x1[e] = coherent_load(gd1 + coef2(e));
    #pragma unroll
        for (int e = 0; e < 16; e++) x0[e] = coherent_load(gd0 + coef2(e));
        compiler_fence();
        c64 key0[L][8], key1[L][8];
    #pragma unroll
        for (int j = 0; j < L; j++) load_row(key0[j], 0, j);
        c64 twist[8], wc[4];
    #pragma unroll
        for (int n1 = 0; n1 < 8; n1++) twist[n1] = tab[kTWOff + w * 512 + lane + 64 * n1];
    #pragma unroll
        for (int i = 0; i < 4; i++) wc[i] = tab[kWCOff + 256 * w + lane + 64 * i];
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
            compiler_fence();
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

        // ---- accumulation chain of output polynomial h: rows (0, j = 0..3) then (1, j = 0..3)
        c64 V[8];
    #pragma unroll
        for (int r = 0; r < 8; r++) V[r] = {0.0, 0.0};
    #pragma unroll
        for (int p = 0; p < 2; p++)
    #pragma unroll
            for (int j = 0; j < L; j++) {
                // row polynomial p: my own transforms when p == h, the sibling's otherwise — both read back
                // from LDS, so that the 128 registers of X are free for the selector rows
                const char* src = smem + kTableBytes + ((p * 2 + w) * 32768);
                c64 sx[8];
    #pragma unroll
                for (int r = 0; r < 8; r++) sx[r] = reinterpret_cast<const c64*>(src)[(j * 8 + r) * 64 + lane];
    #pragma unroll
                for (int r = 0; r < 8; r++) {
                    const c64 k = {p == 0 ? key0[j][r].re : key1[j][r].re, p == 0 ? key0[j][r].im : key1[j][r].im};
                    const c64 x = sx[r];
                    double re = __builtin_fma(k.re, x.re, V[r].re);
                    double im = __builtin_fma(k.re, x.im, V[r].im);
                    V[r].re = __builtin_fma(-k.im, x.im, re);
                    V[r].im = __builtin_fma(k.im, x.re, im);
                }
            }
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
    #pragma unroll
            for (int i = 0; i < 4; i++) {
                const c64 in = reinterpret_cast<const c64*>(partner)[i * 64 + lane];
                V[i] = {w == 0 ? Ep[i].re : in.re, w == 0 ? Ep[i].im : in.im};
                V[4 + i] = {w == 0 ? in.re : Op[i].re, w == 0 ? in.im : Op[i].im};
            }
            wg_barrier(); // cross reads retired before the image is overwritten
        }
        // add_glwe_ciphertexts(c, prod, d_0) (fft_ops.rs:180): d_0 re-read under the inverse transform
        uint64_t d0w[16];
    #pragma unroll
        for (int e = 0; e < 16; e++) d0w[e] = coherent_load(gd0 + coef2(e));
        fft512_single<-1, 7>(V, mine, tab, lane);
        uint64_t t[16];
        untwist_to_torus_bits(V, twist, t);
    #pragma unroll
        for (int e = 0; e < 16; e++) coherent_store(gout + coef2(e), (d0_zero ? 0 : d0w[e]) + t[e]);

        // the unit is complete when all four waves' stores have been acknowledged: wait, barrier, word
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(a.done + ct, a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

constexpr int kCmuxFlowLds = kCmux4Lds;
template <int L, int LOGB>
__global__ __launch_bounds__(256, 1) void cmux_flow_kernel(CmuxFlowArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) & 1) cmux_flow_body<L, LOGB, 1>(a, smem);
    else cmux_flow_body<L, LOGB, 0>(a, smem);
}

} // namespace spf
