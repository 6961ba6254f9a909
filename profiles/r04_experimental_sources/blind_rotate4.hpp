// blind_rotate4_kernel: the four-waves-per-ciphertext latency shape shipped in r02-r03, replaced in r04 by
// blind_rotate8_kernel (spf_amd/csrc/spf_kernels.hpp): 3.87 / 3.94 / 4.03 ms against 3.73 / 3.82 / 3.84 ms per launch at
// B = 64 / 256 / 256 plain, bit-equal (profiles/r04_experiments_blind_rotate.md).  Kept here as the record of the
// measured alternative; not compiled into the library.  It needs the helpers of spf_device.hpp / spf_kernels.hpp.

// ------------------------------------------------------------------------------------------
// blind_rotate4_kernel: FOUR waves per ciphertext, one ciphertext per workgroup, one wave per SIMD —
// the shape for batches of at most one ciphertext per CU (B <= #CU), where latency is all that
// counts.  Wave (w, h): sample parity w (as in the two-wave kernels) and polynomial h.  The two
// polynomials of a CMUX step are independent until the multiply-accumulate, so the pair h = 0 and
// the pair h = 1 each rotate, decompose and transform ONE polynomial (both digits together,
// `fft512_pair_pipelined`) at the same time, and each transforms ONE output polynomial back.  The accumulation
//   prod[q] = X00 K00q + X01 K01q + X10 K10q + X11 K11q          (in this order, each term 4 FMAs)
// stays the sequential chain the reference's `glwe_ggsw_mad` defines: the two pairs swap their
// transforms through LDS and wave (w, h) then runs the whole chain of OUTPUT polynomial q = h (its
// own transforms for the rows of polynomial h, the sibling's for the others).  Same operations in
// the same order on every value: same words.  All hand-overs are s_barrier among the four waves
// (four per step: staged, cross data out, spectra out, inverse cross data out — each LDS region has one
// use per step, so nothing waits for "reads retired"); keys go straight from L2 into registers, requested
// a step ahead.  The whole 160 KiB of LDS: twiddles, 4 x 2 exchange images, 4 staging / spectra regions.
constexpr int kBlindRotate4Lds = kTableBytes + 4 * 2 * 8192 + 4 * 16384; // exchange images + staging / spectra regions

template <int L, int LOGB, int W, int MIX = 1>
__device__ __forceinline__ void blind_rotate4_body(const BlindRotateArgs& a, char* smem)
{
    static_assert(L == 2 && L * LOGB <= 32, "two digits, processed as a pair");
    c64* tab = reinterpret_cast<c64*>(smem);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int w = W; // sample parity: compile-time (one copy of the body per parity), so that which half of a
                         // cross exchange a wave keeps is static instead of ~160 v_cndmask per step
    const int h = wv >> 1;
    // region of wave (w, h): two 8 KiB images
    auto region = [&](int ww, int hh) -> char* { return smem + kTableBytes + (hh * 2 + ww) * 16384; };
    char* mine = region(w, h);
    char* mineB = mine + 8192;
    char* partner = region(w ^ 1, h); // same polynomial, other parity
    // second region of wave (w, h), 16 KiB: its staged accumulator (rotation source) at the top of a step, its two
    // transforms for the MADs later.  Having it apart from the exchange images is what lets a step do with four
    // barriers instead of eight: no image is reused while someone may still read it.
    auto spectra = [&](int ww, int hh) -> char* { return smem + kTableBytes + 4 * 16384 + (hh * 2 + ww) * 16384; };
    auto wg_barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // not __syncthreads(): keep the key loads in flight
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    {
        const double2* src = reinterpret_cast<const double2*>(a.tables);
        double2* dst = reinterpret_cast<double2*>(smem);
        for (int i = tid; i < kTableEntries; i += 256) dst[i] = src[i];
    }
    const uint32_t ct = blockIdx.x; // grid = B
    const uint64_t* lwe = a.lwe_in + (size_t)ct * (a.n + 1);
    const uint64_t* lut = a.lut + (size_t)ct * a.lut_stride;
    auto coef2 = [&](int e) -> int { return (e >> 3) * 1024 + (e & 7) * 128 + 2 * lane + w; };

    uint64_t acc[16]; // polynomial h, parity w
    {
        uint32_t bt = mod_switch_2n(lwe[a.n] + a.body_rotate, a.log_chi, a.log_v);
#pragma unroll
        for (int e = 0; e < 16; e++) {
            uint32_t idx = (uint32_t)coef2(e) + bt;
            uint64_t v = lut[h * kN + (idx & (kN - 1))];
            acc[e] = ((idx >> 11) & 1) ? (uint64_t)0 - v : v;
        }
    }
    __syncthreads(); // twiddle image in place

    c64 twist[8], wc[4];
#pragma unroll
    for (int n1 = 0; n1 < 8; n1++) twist[n1] = tab[kTWOff + w * 512 + lane + 64 * n1];
#pragma unroll
    for (int i = 0; i < 4; i++) wc[i] = tab[kWCOff + 256 * w + lane + 64 * i];

#ifdef SPF_STAMPS
    uint64_t st_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t st_prev = __builtin_amdgcn_s_memtime();
#define STAMP4(i) do { uint64_t t_ = __builtin_amdgcn_s_memtime(); st_acc[i] += t_ - st_prev; st_prev = t_; } while (0)
#else
#define STAMP4(i) do { } while (0)
#endif
    // this wave's bins of OUTPUT polynomial h in all four key rows of a step (levels consumed in reverse):
    // [row polynomial p][digit j][r].  32 KiB per wave and step, 128 KiB per CU: about 4 500 cycles of the CU's
    // L2 path (~30 B/clk), and a wave that asks for all of it in one place spends 2 000-3 000 cycles waiting to
    // issue.  So the rows of step s+1 are requested in eight pieces of four loads, spread from behind the MADs
    // of step s (which free the registers) to the transform of step s+1.
    c64 key[2][2][8];
    const c64* key_base = a.bsk + h * kHalf + 256 * w + lane;
    const c64* key_next = key_base; // rows of the step whose pieces are being requested
    auto request_keys = [&](auto piece_c) {
        constexpr int piece = decltype(piece_c)::value;
        constexpr int p = piece >> 2, j = (piece >> 1) & 1, r0 = 4 * (piece & 1);
        const c64* row = key_next + (size_t)(p * L + (L - 1 - j)) * (2 * kHalf);
#pragma unroll
        for (int r = r0; r < r0 + 4; r++) key[p][j][r] = row[64 * (r & 3) + 512 * (r >> 2)];
    };
#define SPF_KEY_PIECE(i) request_keys(std::integral_constant<int, i>{})
    SPF_KEY_PIECE(0); SPF_KEY_PIECE(1); SPF_KEY_PIECE(2); SPF_KEY_PIECE(3); SPF_KEY_PIECE(4);
    uint64_t a_next = lwe[0];
    // One CMUX step.  LAST = the final step, compiled as its own copy WITHOUT the requests for the next step's rows: until r03
    // the last step re-requested its own rows (dead loads, drained behind the loop); a load nobody consumes is a write into a
    // register the allocator considers free, and r03's persistent-CMUX experiment showed how such a kernel goes wrong one
    // gate in a thousand — so no shipped kernel issues one any more.
    auto cmux_step = [&](uint32_t step, auto last_c) {
        constexpr bool LAST = decltype(last_c)::value;
        const uint32_t at = mod_switch_2n(a_next, a.log_chi, a.log_v);
        a_next = lwe[step + 1];
        SPF_KEY_PIECE(5);


        // ---- rotate, subtract, decompose polynomial h
        uint64_t* stage = reinterpret_cast<uint64_t*>(spectra(w, h));
#pragma unroll
        for (int e = 0; e < 16; e++) stage[(e >> 3) * 512 + (e & 7) * 64 + lane] = acc[e];
        // 1: both parities of both polynomials staged.  Not needed when every rotation amount is even (MIX = 0, log_v >= 1:
        // an even rotation keeps the coefficient parity, the wave gathers only from the region it staged itself)
        if constexpr (MIX) wg_barrier();
        else compiler_fence();
        STAMP4(0);
        SPF_KEY_PIECE(6);
        uint32_t dig[16];
        {
            // source coefficient of element e: (c_e - at) mod 2N with c_e = c_0 + 128 m (m = e & 7, +1024 for
            // e >= 8): region (parity) and the low address bits do not depend on e
            const uint32_t t0 = (uint32_t)(2 * lane + w) + 2 * kN - at;
            const char* src = spectra((int)(t0 & 1), h);
            uint64_t gin[16];
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const uint32_t t = t0 + (e >> 3) * 1024 + (e & 7) * 128;
                gin[e] = *reinterpret_cast<const uint64_t*>(src + ((t << 2) & 0x1FF8u));
            }
            sched_fence(); // all sixteen reads out before the first is consumed
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const uint32_t t = t0 + (e >> 3) * 1024 + (e & 7) * 128;
                const uint64_t sgn = (uint64_t)((int64_t)((uint64_t)t << 52) >> 63); // bit 11 of t, spread
                const uint64_t rot = (gin[e] ^ sgn) - sgn;
                dig[e] = gadget_digits_packed<L, LOGB>(rot - acc[e]);
            }
        }
        c64 VV[2][8];
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int n1 = 0; n1 < 8; n1++) VV[j][n1] = twisted_digit<LOGB>(dig[n1], dig[8 + n1], j, twist[n1]);
        SPF_KEY_PIECE(7);
        STAMP4(1);
        STAMP4(2);
        fft512_pair_pipelined<+1>(VV[0], VV[1], mine, mineB, tab, lane);
        STAMP4(3);
        // radix-2 stage across the parities, both digits in one exchange
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int i = 0; i < 4; i++)
                reinterpret_cast<c64*>(mine)[(j * 4 + i) * 64 + lane] = {w == 0 ? VV[j][4 + i].re : VV[j][i].re,
                                                                         w == 0 ? VV[j][4 + i].im : VV[j][i].im};
        wg_barrier(); // 3
        STAMP4(4);
        {
            c64 xin[2][4];
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int i = 0; i < 4; i++) xin[j][i] = reinterpret_cast<const c64*>(partner)[(j * 4 + i) * 64 + lane];
            sched_fence();
#pragma unroll
            for (int j = 0; j < 2; j++) {
                c64 X[8];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const c64 in = xin[j][i];
                    const c64 Ei = {w == 0 ? VV[j][i].re : in.re, w == 0 ? VV[j][i].im : in.im};
                    const c64 Oi = {w == 0 ? in.re : VV[j][4 + i].re, w == 0 ? in.im : VV[j][4 + i].im};
                    c64 t = cmul_tw<+1>(Oi, wc[i]);
                    X[i] = cadd(Ei, t);
                    X[i + 4] = csub(Ei, t);
                }
#pragma unroll
                for (int r = 0; r < 8; r++) VV[j][r] = X[r];
            }
        }
        STAMP4(5);

        // ---- multiply-accumulate.  prod[q] = X00 K00q + X01 K01q + X10 K10q + X11 K11q, in this
        // order (glwe_ggsw_mad): the two pairs swap their transforms through LDS, then wave (w, h)
        // runs the whole chain of output polynomial q = h.
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 8; r++) reinterpret_cast<c64*>(spectra(w, h))[(j * 8 + r) * 64 + lane] = VV[j][r];
        wg_barrier(); // 5: every wave's two transforms are in its region (the gathers from it ended before barrier 3)
        STAMP4(6);
        c64 V[8]; // prod[h]
#pragma unroll
        for (int r = 0; r < 8; r++) V[r] = {0.0, 0.0};
#pragma unroll
        for (int p = 0; p < 2; p++) {
            // row polynomial p: this wave's own transforms when p == h, the sibling's otherwise — both read back
            // from LDS (16 more ds_read_b128 instead of 128 v_cndmask per step)
            const c64* sx = reinterpret_cast<const c64*>(spectra(w, p)) + lane;
            c64 X[2][8];
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int r = 0; r < 8; r++) X[j][r] = sx[(j * 8 + r) * 64];
            sched_fence(); // the sixteen spectrum values of a row polynomial in one go (they were fetched two at a time)
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int r = 0; r < 8; r++) {
                    const c64 k = key[p][j][r];
                    const c64 x = X[j][r];
                    double re = __builtin_fma(k.re, x.re, V[r].re);
                    double im = __builtin_fma(k.re, x.im, V[r].im);
                    V[r].re = __builtin_fma(-k.im, x.im, re);
                    V[r].im = __builtin_fma(k.im, x.re, im);
                }
        }
        key_next = key_base + (size_t)(step + 1) * (2 * L) * (2 * kHalf);
        if constexpr (!LAST) SPF_KEY_PIECE(0);
        STAMP4(7);
        if constexpr (!LAST) SPF_KEY_PIECE(1);

        // ---- polynomial h back to the torus
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
            wg_barrier(); // 7
            if constexpr (!LAST) SPF_KEY_PIECE(2);
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
        }
        STAMP4(8);
        if constexpr (!LAST) SPF_KEY_PIECE(3);
        fft512_single<-1, 7>(V, mineB, tab, lane); // image B: the partner may still be reading the cross data in A
        STAMP4(9);
        if constexpr (!LAST) SPF_KEY_PIECE(4);
        {
            uint64_t t[16];
            untwist_to_torus_bits(V, twist, t);
#pragma unroll
            for (int e = 0; e < 16; e++) acc[e] += t[e];
        }
        STAMP4(10);
        // the next step's staging writes this wave's own image A, which nobody reads after barrier 8
    };
    for (uint32_t step = 0; step + 1 < a.n; step++) cmux_step(step, std::false_type{});
    cmux_step(a.n - 1, std::true_type{});

#ifdef SPF_STAMPS
    if (a.stamps && lane == 0) {
#pragma unroll
        for (int i = 0; i < 12; i++) a.stamps[((size_t)blockIdx.x * 4 + wv) * 16 + i] = st_acc[i];
    }
#endif
#undef STAMP4
#undef SPF_KEY_PIECE
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

// one copy of the body per sample parity (see blind_rotate2p_kernel)
template <int L, int LOGB, int MIX = 1>
__global__ __launch_bounds__(256, 1) void blind_rotate4_kernel(BlindRotateArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) & 1) blind_rotate4_body<L, LOGB, 1, MIX>(a, smem);
    else blind_rotate4_body<L, LOGB, 0, MIX>(a, smem);
}

