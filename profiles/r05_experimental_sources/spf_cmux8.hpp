// NOT in the library.  r05 experiment (verdict r04 item 6a): the CMUX-tree gate's latency shape over eight waves instead of four.
// Bit-equal with cmux4_kernel on first run (checksums of tools/kernel_bench.py cmux at B = 1 / 4 / 64 / 256), and SLOWER:
//   B = 1 / 4 / 64 / 256:  cmux4 14.7 / 14.9 / 15.4 / 23.7 us per launch, cmux8 16.9 / 17.1 / 17.8 / 29.0 us.
// Why (stamps, profiles/r05_experiments_other_kernels.md): a wave that has its SIMD to itself already issues a transform pair at the SIMD's
// f64 rate (5.0 k cycles a pair alone; cmux4's "decompose + 2 x (pair, cross)" is 12.3 k), and SIMD siblings share that rate: the
// pair plus the wait for the sibling's pair at the cross barrier is 10.2 k here.  What the split saves (2 k) is spent on twice the
// waves to start (entry + table barrier 12.6 k against 7.0 k), 140 B of scratch at 256 registers, and two more barriers.
// To build it: paste this block into spf_kernels.hpp after cmux4_kernel and launch cmux8_kernel<4,4> with 512 threads and kCmux8Lds.
// ------------------------------------------------------------------------------------------
// cmux8_kernel: cmux4_kernel's gate over EIGHT waves — wave (w, h, jj): sample parity w x polynomial h x digit pair jj
// (digits 2jj, 2jj + 1); waves (w, h, 0) and (w, h, 1) are SIMD siblings (wave number 4 jj + 2 h + w).  cmux4's stamps
// (profiles/r05_experiments_other_kernels.md) put 12.3 k of a gate's 35 k cycles into "decompose + 2 x (transform pair, cross)" of a
// wave that has its SIMD to itself: here every wave runs ONE `fft512_pair_pipelined`, the sibling the other beside it.
//   * both waves of a pair load and decompose polynomial h themselves (no hand-over: the loads are the same lines, and a
//     hand-over is a barrier);
//   * the accumulation chain of output polynomial h is split by BINS between the two waves, as in blind_rotate8_kernel
//     (registers {2jj, 2jj+1, 2jj+4, 2jj+5}: the pairs the inverse split needs are in one wave); per bin it is the reference's
//     chain over the eight rows (row polynomial 0 levels 3..0, then polynomial 1), each wave reading half a selector row;
//   * the four waves of polynomial h post their E' / O' halves into the inboxes of the two jj = 0 waves, which transform back,
//     untwist, convert, add d0 and store; the jj = 1 waves are done after posting.
// Same operations in the same order on every value as cmux_kernel / cmux4_kernel: same words.  LDS: tables + eight 16 KiB
// regions (two exchange images while transforming, then the wave's two transforms) = 160 KiB; six barriers.
constexpr int kCmux8Lds = kTableBytes + 8 * 16384;

template <int L, int LOGB, int W, int JJ>
__device__ __forceinline__ void cmux8_body(const CmuxArgs& a, char* smem)
{
    static_assert(L == 4 && L * LOGB <= 32, "four digits, one pair per wave");
    c64* tab = reinterpret_cast<c64*>(smem);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6); // = 4 jj + 2 h + w
    constexpr int w = W, jj = JJ;
    const int h = (wv >> 1) & 1;
#ifdef SPF_STAMPS
    uint64_t st_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t st_prev = __builtin_amdgcn_s_memtime();
#define STAMPC(i) do { uint64_t t_ = __builtin_amdgcn_s_memtime(); st_acc[i] += t_ - st_prev; st_prev = t_; } while (0)
#else
#define STAMPC(i) do { } while (0)
#endif
    auto region = [&](int ww, int hh, int j2) -> char* { return smem + kTableBytes + ((j2 * 2 + hh) * 2 + ww) * 16384; };
    char* mine = region(w, h, jj);
    char* mineB = mine + 8192;
    const char* partner = region(w ^ 1, h, jj);
    auto wg_barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // keeps the selector loads in flight (no vmcnt drain)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    const uint32_t ct = blockIdx.x; // grid = units
    // load order = need order (vmcnt retires in issue order): twiddle image, d1 / d0, selector rows last
    constexpr int kTabPerThread = (kTableEntries + 511) / 512;
    f64x2_t tab_img[kTabPerThread];
    {
        const f64x2_t* src = reinterpret_cast<const f64x2_t*>(a.tables);
#pragma unroll
        for (int i = 0; i < kTabPerThread; i++) {
            const int idx = tid + 512 * i;
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
    const gc64_ptr gkey = global_view(ggsw) + 256 * w + lane + 128 * jj;
    const gu64_cptr gd0 = global_view(d0) + h * kN, gd1 = global_view(d1) + h * kN;
    const gu64_ptr gout = global_view(out_ct) + h * kN;
    // selector row (p, level L-1-j), output polynomial h, this wave's four bins: q -> register 2jj + (q & 1) + 4 (q >> 1)
    auto load_row = [&](c64 (&k)[4], int p, int j) {
        const gc64_ptr row = gkey + (size_t)((p * L + (L - 1 - j)) * 2 + h) * kHalf;
#pragma unroll
        for (int q = 0; q < 4; q++) k[q] = gload(row + 64 * (q & 1) + 512 * (q >> 1));
    };
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
            const int idx = tid + 512 * i;
            if (idx < kTableEntries) dst[idx] = tab_img[i];
        }
    }
    compiler_fence();
    c64 key0[L][4], key1[L][4];
#pragma unroll
    for (int j = 0; j < L; j++) load_row(key0[j], 0, j);
    STAMPC(0);
    wg_barrier(); // twiddle image in place
    STAMPC(1);
    const c64* twist_lds = tab + kTWOff + w * 512 + lane;
    const c64* wc_lds = tab + kWCOff + 256 * w + lane;
    c64 X[2][8];
    {
        c64 twist[8];
#pragma unroll
        for (int n1 = 0; n1 < 8; n1++) twist[n1] = twist_lds[64 * n1];
        uint32_t dig[16];
#pragma unroll
        for (int e = 0; e < 16; e++) // sub_glwe_ciphertexts(diff, d_1, d_0) (fft_ops.rs:168), then the gadget digits
            dig[e] = gadget_digits_packed<L, LOGB>(x1[e] - (d0_zero ? 0 : x0[e]));
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int n1 = 0; n1 < 8; n1++) X[t][n1] = twisted_digit<LOGB>(dig[n1], dig[8 + n1], 2 * jj + t, twist[n1]);
    }
    STAMPC(2);
    fft512_pair_pipelined<+1>(X[0], X[1], mine, mineB, tab, lane);
    STAMPC(3);
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
        for (int i = 0; i < 4; i++)
            reinterpret_cast<c64*>(mine)[(t * 4 + i) * 64 + lane] = {w == 0 ? X[t][4 + i].re : X[t][i].re, w == 0 ? X[t][4 + i].im : X[t][i].im};
    wg_barrier();
    {
        c64 xin[2][4], wc[4];
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int i = 0; i < 4; i++) xin[t][i] = reinterpret_cast<const c64*>(partner)[(t * 4 + i) * 64 + lane];
#pragma unroll
        for (int i = 0; i < 4; i++) wc[i] = wc_lds[64 * i];
        sched_fence();
#pragma unroll
        for (int t = 0; t < 2; t++) {
            c64 Y[8];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const c64 in = xin[t][i];
                const c64 Ei = {w == 0 ? X[t][i].re : in.re, w == 0 ? X[t][i].im : in.im};
                const c64 Oi = {w == 0 ? in.re : X[t][4 + i].re, w == 0 ? in.im : X[t][4 + i].im};
                c64 tt = cmul_tw<+1>(Oi, wc[i]);
                Y[i] = cadd(Ei, tt);
                Y[i + 4] = csub(Ei, tt);
            }
#pragma unroll
            for (int r = 0; r < 8; r++) X[t][r] = Y[r];
        }
    }
    STAMPC(4);
    wg_barrier(); // cross reads retired: the regions can carry the transforms
#pragma unroll
    for (int t = 0; t < 2; t++) {
#pragma unroll
        for (int r = 0; r < 8; r++) reinterpret_cast<c64*>(mine)[(t * 8 + r) * 64 + lane] = X[t][r];
        load_row(key1[2 * t], 1, 2 * t);
        load_row(key1[2 * t + 1], 1, 2 * t + 1);
    }
    wg_barrier(); // every wave's two transforms are in its region
    STAMPC(5);

    // ---- accumulation chain of output polynomial h, this wave's four bins: rows (0, j = 0..3) then (1, j = 0..3)
    c64 V[4];
#pragma unroll
    for (int q = 0; q < 4; q++) V[q] = {0.0, 0.0};
    {
        // transform of digit j of row polynomial p: wave (w, p, j >> 1), its transform j & 1
        auto row_src = [&](int m) {
            const int p = m / L, j = m % L;
            return reinterpret_cast<const c64*>(smem + kTableBytes + (((j >> 1) * 2 + p) * 2 + w) * 16384) + ((j & 1) * 8 + 2 * jj) * 64 + lane;
        };
        // two rows per request group, the next group requested before the FMAs of this one
        c64 sx[2][2][4];
        auto request = [&](int g) {
#pragma unroll
            for (int mm = 0; mm < 2; mm++)
#pragma unroll
                for (int q = 0; q < 4; q++) sx[g & 1][mm][q] = row_src(2 * g + mm)[((q & 1) + 4 * (q >> 1)) * 64];
        };
        request(0);
#pragma unroll
        for (int g = 0; g < L; g++) {
            if (g + 1 < L) request(g + 1);
            sched_fence();
#pragma unroll
            for (int mm = 0; mm < 2; mm++) {
                const int m = 2 * g + mm, p = m / L, j = m % L;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const c64 k = {p == 0 ? key0[j][q].re : key1[j][q].re, p == 0 ? key0[j][q].im : key1[j][q].im};
                    const c64 x = sx[g & 1][mm][q];
                    double re = __builtin_fma(k.re, x.re, V[q].re);
                    double im = __builtin_fma(k.re, x.im, V[q].im);
                    V[q].re = __builtin_fma(-k.im, x.im, re);
                    V[q].im = __builtin_fma(k.im, x.re, im);
                }
            }
        }
    }
    STAMPC(6);
    wg_barrier(); // every wave's chain reads retired: the regions of the jj = 0 waves become the inboxes
    // ---- inverse split; E' halves to the inbox of wave (0, h, 0), O' halves to that of (1, h, 0)
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const c64 Ep = cadd(V[i], V[2 + i]);
        const c64 Op = cmul_tw<-1>(csub(V[i], V[2 + i]), wc_lds[64 * (2 * jj + i)]);
        reinterpret_cast<c64*>(region(0, h, 0))[(w * 4 + 2 * jj + i) * 64 + lane] = Ep;
        reinterpret_cast<c64*>(region(1, h, 0))[(w * 4 + 2 * jj + i) * 64 + lane] = Op;
    }
    STAMPC(7);
    wg_barrier();
    if constexpr (JJ == 0) {
        // add_glwe_ciphertexts(c, prod, d_0) (fft_ops.rs:180): d_0 re-read under the inverse transform
        uint64_t d0w[16];
#pragma unroll
        for (int e = 0; e < 16; e++) d0w[e] = gd0[coef2(e)];
        c64 U[8];
#pragma unroll
        for (int r = 0; r < 8; r++) U[r] = reinterpret_cast<const c64*>(mine)[r * 64 + lane];
        sched_fence();
        STAMPC(8);
        fft512_single<-1, 7>(U, mine, tab, lane); // its exchanges follow the inbox reads in this wave's own LDS queue
        uint64_t t[16];
        untwist_to_torus_bits(U, twist_lds, t);
#pragma unroll
        for (int e = 0; e < 16; e++) gout[coef2(e)] = (d0_zero ? 0 : d0w[e]) + t[e];
        STAMPC(9);
    }
#ifdef SPF_STAMPS
    if (a.stamps && lane == 0) {
#pragma unroll
        for (int i = 0; i < 10; i++) a.stamps[((size_t)blockIdx.x * 8 + wv) * 16 + i] = st_acc[i];
    }
#endif
#undef STAMPC
}

template <int L, int LOGB>
__global__ __launch_bounds__(512) void cmux8_kernel(CmuxArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wv >> 2) {
        if (wv & 1) cmux8_body<L, LOGB, 1, 1>(a, smem);
        else cmux8_body<L, LOGB, 0, 1>(a, smem);
    } else {
        if (wv & 1) cmux8_body<L, LOGB, 1, 0>(a, smem);
        else cmux8_body<L, LOGB, 0, 0>(a, smem);
    }
}

