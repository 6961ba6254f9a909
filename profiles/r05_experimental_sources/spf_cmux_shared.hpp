// spf_cmux_shared.hpp — r05 experiment, NOT part of the library (compiled by nothing): `cmux_shared_kernel`, the CMUX of a gate-graph
// level whose gates share selectors, as it was built into spf_kernels.hpp / spf_hip.hip / spf_graph.hpp and measured
// (profiles/r05_experiments_other_kernels.md, "Shared-selector CMUX"): bit-equal to cmux_kernel and to the oracle, no faster on the
// 32 x 32 multiplier's levels.  Kept as the record of what was measured.
//
// ---- device side (was in spf_kernels.hpp, between cmux_kernel and cmux4_kernel) ----
// ------------------------------------------------------------------------------------------
// cmux_shared_kernel: the CMUX of a gate-graph level whose gates SHARE selectors.  A level of a `mux_circuits` block tests one
// variable (MuxCircuit::from(&[Bdd]), mux_circuits/src/lib.rs:358-445): BASELINE config 5's 32 x 32 multiplier has 45 gates per
// distinct selector and level on average (tools/graph_selector_sharing.py), and a wide level of four such jobs is ~800 gates on 8
// selectors.  cmux_kernel reads every gate's own 256 KiB of selector (from L2 when it is shared: 58 us per level of 864 gates,
// the HBM-streaming rate); here FOUR gates with the same selector are one workgroup with the blind rotation's layout: two waves
// per gate (sample parity), the selector's rows through the 64 KiB LDS ring by LDS-DMA — one 64 KiB chunk = the two GGSW levels
// of a digit pair of one input polynomial, fetched once for the four gates, the next chunk requested under the current one's
// transforms — the four digits of a polynomial as TWO transform pairs (cmux_kernel: eight single transforms with two hand-overs
// each), two workgroup barriers per chunk.  A CMUX with l = 4 is two blind-rotation steps' worth of forward work and one
// inverse.  Arithmetic, association and accumulation order are cmux_kernel's (fft_ops.rs:149-181, 23-98: rows p ascending,
// digits least significant first against GLEV rows in reverse): same words.
// ptrs: 4 pointers per unit {selector, d0 (null = zero ciphertext), d1, out} as CmuxArgs::ptrs; workgroup g owns units
// 4g .. 4g+3, which all select on unit 4g's GGSW; a unit with out == null is padding (computes on unit 4g's operands, stores nothing).
struct CmuxSharedArgs {
    const void* const* ptrs;
    const c64* tables;
    uint32_t groups;
    uint64_t* stamps; // diagnostic builds (-DSPF_STAMPS): [workgroup][wave][16] cycle sums per phase, else null
};
constexpr int kCmuxSharedLds = kBlindRotate2pLds;

template <int W>
__device__ __forceinline__ void cmux_shared_body(const CmuxSharedArgs& a, char* smem)
{
    constexpr int L = 4, LOGB = 4, CTS = 4, NT = 128 * CTS;
#ifdef SPF_STAMPS
    uint64_t st_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t st_prev = __builtin_amdgcn_s_memtime();
#define STAMPH(i) do { uint64_t t_ = __builtin_amdgcn_s_memtime(); st_acc[i] += t_ - st_prev; st_prev = t_; } while (0)
#else
#define STAMPH(i) do { } while (0)
#endif
    c64* tab = reinterpret_cast<c64*>(smem);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cslot = wv >> 1;
    constexpr int w = W;
    char* tile = smem + kTableBytes + cslot * kWaveBufBytes;
    char* mine = tile + w * 8192;
    char* theirs = tile + (w ^ 1) * 8192;
    char* bskring = smem + kTableBytes + CTS * kWaveBufBytes;

    const void* const* t0 = a.ptrs + 16 * (size_t)blockIdx.x;
    const void* const* tme = t0 + 4 * cslot;
    const char* ggsw = static_cast<const char*>(t0[0]);
    uint64_t* out_ct = static_cast<uint64_t*>(const_cast<void*>(tme[3]));
    const bool owns_output = out_ct != nullptr;
    const void* const* tu = owns_output ? tme : t0;
    const uint64_t* d1 = static_cast<const uint64_t*>(tu[2]);
    const bool d0_zero = tu[1] == nullptr;
    const uint64_t* d0 = d0_zero ? d1 : static_cast<const uint64_t*>(tu[1]);
    const gu64_cptr gd0 = global_view(d0), gd1 = global_view(d1);
    const gu64_ptr gout = global_view(out_ct);
    auto coef2 = [&](int e) -> int { return (e >> 3) * 1024 + (e & 7) * 128 + 2 * lane + w; };

    // chunk c = 2 p + h: rows of input polynomial p for the digits 2h, 2h+1 = GGSW levels L-1-2h, L-2-2h, 64 KiB as they lie
    // (lower level first): digit 2h+jj is the ring half 1-jj, as in the blind rotation
    const uint32_t dma_voff = (uint32_t)tid * 16u;
    const uint32_t dma_dst = lds_address(bskring) + wv * 1024;
    auto ring_dma = [&](int c) {
        const int p = c >> 1, h = c & 1;
        const char* src = ggsw + (size_t)(p * L + (L - 2 - 2 * h)) * kBskSlotBytes;
#pragma unroll
        for (int k = 0; k < 2 * kBskSlotBytes / (NT * 16); k++) lds_dma_piece(src + k * NT * 16, dma_voff, dma_dst + k * NT * 16);
    };
    ring_dma(0);

    uint64_t x1[2][16], x0[2][16];
#pragma unroll
    for (int p = 0; p < 2; p++)
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const int c = p * kN + coef2(e);
            x1[p][e] = gd1[c];
            x0[p][e] = gd0[c];
        }
    sched_fence();
    {
        const double2* src = reinterpret_cast<const double2*>(a.tables);
        double2* dst = reinterpret_cast<double2*>(smem);
        for (int i = tid; i < kTableEntries; i += NT) dst[i] = src[i];
    }
    sched_fence();
    uint32_t dig[2][16];
#pragma unroll
    for (int p = 0; p < 2; p++)
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const uint64_t diff = x1[p][e] - (d0_zero ? 0 : x0[p][e]); // sub_glwe_ciphertexts(diff, d_1, d_0) (fft_ops.rs:168)
            dig[p][e] = gadget_digits_packed<L, LOGB>(diff);
        }
    STAMPH(0);
    __syncthreads(); // twiddle image ready
    STAMPH(1);

    const c64* twist = tab + kTWOff + w * 512 + lane;
    const c64* wc = tab + kWCOff + 256 * w + lane;
    c64 prod[2][8];
#pragma unroll
    for (int q = 0; q < 2; q++)
#pragma unroll
        for (int r = 0; r < 8; r++) prod[q][r] = {0.0, 0.0};

    // Chunks in the order (p, h) = (0, 0), (0, 1), (1, 0), (1, 1) as ONE real loop (the polynomial's digit words picked by
    // selects).  A workgroup runs this code once: what it costs is its SIZE — unrolled (77 KB for the two parity copies, more than
    // the 64 KB instruction cache two CUs share) a level of 232 workgroups took 103 us, most of it instruction fetch.
    {
#pragma unroll 1
        for (int c = 0; c < 4; c++) {
            const int p = c >> 1, h = c & 1;
            if (c > 0) ring_dma(c); // the ring is free since the barrier behind the last chunk's accumulation
            c64 VV[2][8];
            {
                c64 twf[8];
#pragma unroll
                for (int n1 = 0; n1 < 8; n1++) twf[n1] = twist[64 * n1];
                compiler_fence();
                const int sh0 = 32 - LOGB - 2 * h * LOGB; // left shift that brings digit 2h to the top; digit 2h+1: LOGB less
#pragma unroll
                for (int n1 = 0; n1 < 8; n1++)
#pragma unroll
                    for (int jj = 0; jj < 2; jj++) {
                        const uint32_t wre = p ? dig[1][n1] : dig[0][n1];
                        const uint32_t wim = p ? dig[1][8 + n1] : dig[0][8 + n1];
                        const int dre = ((int)(wre << (sh0 - jj * LOGB))) >> (32 - LOGB);
                        const int dim = ((int)(wim << (sh0 - jj * LOGB))) >> (32 - LOGB);
                        VV[jj][n1] = cmul_nf({(double)dre, (double)dim}, twf[n1]);
                    }
            }
            STAMPH(2);
            fft512_pair1ts<+1, 2>(VV[0], VV[1], mine, tab, lane);
            STAMPH(3);
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
            STAMPH(4);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // my share of the chunk has landed
            STAMPH(5);
            __syncthreads();
            STAMPH(6);
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
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const c64* row = reinterpret_cast<const c64*>(bskring + (1 - j) * kBskSlotBytes) + 256 * w + lane;
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
                        double re = __builtin_fma(k.re, VV[j][r].re, prod[q][r].re);
                        double im = __builtin_fma(k.re, VV[j][r].im, prod[q][r].im);
                        prod[q][r].re = __builtin_fma(-k.im, VV[j][r].im, re);
                        prod[q][r].im = __builtin_fma(k.im, VV[j][r].re, im);
                    }
                }
            }
            STAMPH(7);
            __syncthreads(); // every wave is done with the ring and with its partner's cross data
            STAMPH(8);
        }
    }

    // ---- back to the torus: one transform pair for the two output polynomials, out = d_0 + product (fft_ops.rs:180)
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
            WW[q][i] = cadd(prod[q][i], prod[q][i + 4]);
            WW[q][4 + i] = cmul_tw<-1>(csub(prod[q][i], prod[q][i + 4]), wc[64 * i]);
        }
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
    pair_barrier_w();
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
    pair_barrier_w(); // both cross reads retired before either image is overwritten
    STAMPH(9);
    fft512_pair1ts<-1, 2>(WW[0], WW[1], mine, tab, lane);
    STAMPH(10);
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
    STAMPH(11);
#ifdef SPF_STAMPS
    if (a.stamps && lane == 0) {
#pragma unroll
        for (int i = 0; i < 12; i++) a.stamps[((size_t)blockIdx.x * 8 + wv) * 16 + i] = st_acc[i];
    }
#endif
#undef STAMPH
}

__global__ __launch_bounds__(512, 2) void cmux_shared_kernel(CmuxSharedArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) & 1) cmux_shared_body<1>(a, smem);
    else cmux_shared_body<0>(a, smem);
}


// ---- host side (was in spf_hip.hip; declared in include/spf_hip.h as
//      spf_status spf_cmux_shared_dev(spf_ctx *ctx, void *stream, size_t groups, const void *const *d_ptrs);) ----
#if 0
// cmux over scattered operands where every group of FOUR consecutive units selects on the same GGSW (the first unit's
// selector pointer is the group's; a unit with a null `out` is padding): cmux_shared_kernel, one workgroup per group, the
// selector read once per group through the LDS ring.  Same words as spf_cmux_scattered_dev on the same units.
spf_status spf_cmux_shared_dev(spf_ctx* c, void* stream, size_t groups, const void* const* d_ptrs)
{
    if (!c || (groups && !d_ptrs)) return fail(c, SPF_ERR_INVALID_ARGUMENT, "null argument");
    if (c->generic) return fail(c, SPF_ERR_UNSUPPORTED, "the shared-selector CMUX form (gate graphs) is built for DEFAULT_128 only");
    if (c->prm.cbs_radix_log != 4 || c->prm.cbs_radix_count != 4)
        return fail(c, SPF_ERR_UNSUPPORTED, "cmux kernel is built for cbs_radix 4 x 4 bits");
    if (groups == 0) return SPF_OK;
    if (groups > 0x3fffffffu) return fail(c, SPF_ERR_INVALID_ARGUMENT, "batch too large");
    std::lock_guard<std::recursive_mutex> g(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    CmuxSharedArgs a{};
    a.ptrs = d_ptrs; a.tables = c->d_tables; a.groups = (uint32_t)groups;
    c->last_cmux_kernel = "cmux_shared_kernel";
#ifdef SPF_STAMPS
    {
        // diagnostic build: per-phase cycles of the first few launches (median over waves)
        static int reported = 0;
        if (reported < 3) {
            const size_t waves = groups * 8;
            uint64_t* d_st = nullptr;
            (void)hipMalloc(&d_st, waves * 16 * 8);
            (void)hipMemsetAsync(d_st, 0, waves * 16 * 8, (hipStream_t)stream);
            a.stamps = d_st;
            hipLaunchKernelGGL(cmux_shared_kernel, dim3((unsigned)groups), dim3(512), kCmuxSharedLds, (hipStream_t)stream, a);
            (void)hipStreamSynchronize((hipStream_t)stream);
            std::vector<uint64_t> h(waves * 16);
            (void)hipMemcpy(h.data(), d_st, h.size() * 8, hipMemcpyDeviceToHost);
            (void)hipFree(d_st);
            static const char* nm[12] = {"entry: pointers, operand loads, table copy, decomposition", "barrier (table in place)", "digits + twist x4",
                "forward pair x4", "cross write x4", "wait for the chunk x4", "chunk barrier x4", "combine + MAD x4", "ring barrier x4",
                "inverse split + cross exchange", "inverse pair", "untwist + d0 + store issue"};
            fprintf(stderr, "[cmux_shared stamps] groups=%zu (cycles, median over %zu waves)\n", groups, waves);
            double tot = 0;
            for (int i = 0; i < 12; i++) {
                std::vector<uint64_t> v;
                for (size_t wv = 0; wv < waves; wv++) v.push_back(h[wv * 16 + i]);
                std::sort(v.begin(), v.end());
                fprintf(stderr, "[cmux_shared stamps] %-60s %8llu\n", nm[i], (unsigned long long)v[v.size() / 2]);
                tot += (double)v[v.size() / 2];
            }
            fprintf(stderr, "[cmux_shared stamps] total %.0f cycles\n", tot);
            reported++;
            return SPF_OK;
        }
    }
#endif
    hipLaunchKernelGGL(cmux_shared_kernel, dim3((unsigned)groups), dim3(512), kCmuxSharedLds, (hipStream_t)stream, a);
    HIPCHK(c, hipGetLastError());
    return SPF_OK;
}

#endif
// ---- graph planner (was in spf_graph.hpp, plan(): behind the stable_sort of a CMux group's units by selector) ----
#if 0
            // A level wider than one gate per CU whose selectors are shared runs on cmux_shared_kernel: every run of one
            // selector is cut into groups of four units (the last one padded with units that store nothing), one workgroup
            // per group, the selector's 256 KiB read once per group through the LDS ring instead of once per gate.  Below one
            // gate per CU the latency shape (cmux4_kernel, a gate per CU) stays; levels that would be mostly padding stay too.
            // SPF_GRAPH_SHARED=0 switches the form off (A/B).
            {
                const char* env = getenv("SPF_GRAPH_SHARED");
                const bool allow = !(env && env[0] == '0');
                size_t n_groups = 0;
                for (size_t i = 0; i < units.size();) {
                    size_t j = i;
                    while (j < units.size() && units[j].p[0] == units[i].p[0]) j++;
                    n_groups += (j - i + 3) / 4;
                    i = j;
                }
                if (allow && units.size() > (size_t)c->n_cu && n_groups * 4 <= units.size() + units.size() / 4) {
                    gr.shared_groups = n_groups;
                    for (size_t i = 0; i < units.size();) {
                        size_t j = i;
                        while (j < units.size() && units[j].p[0] == units[i].p[0]) j++;
                        for (size_t k = i; k < i + (j - i + 3) / 4 * 4; k++) {
                            Unit u = k < j ? units[k] : units[i];
                            if (k >= j) u.p[3] = nullptr; // padding
                            for (void* q : u.p) table.push_back(q);
                        }
                        i = j;
                    }
                    continue;
                }
            }
#endif
// ---- the parity test it passed on MI355X (was tests/test_gpu_keyless_ops.py) ----
#if 0
def test_cmux_shared_selector_form_against_oracle_and_scattered_form():
    """spf_cmux_shared_dev (cmux_shared_kernel: four gates of one selector per workgroup, the selector through the LDS ring,
    the form a wide gate-graph level takes) — every unit against the oracle's cmux (fft_ops.rs:149-181) and word-equal to
    spf_cmux_scattered_dev on the same units; runs of 5, 4 and 1 units (padding), a zero d0 (multiply_glwe_ggsw)."""
    eng = spf_amd.Engine(to_engine_params(P))
    S = 3
    g = _ggsw(41, S)
    a = random_glwe(42, 10, P.glwe_len)
    b = random_glwe(43, 10, P.glwe_len)
    units = [(0, i, i) for i in range(5)] + [(1, 5 + i, 5 + i) for i in range(4)] + [(2, None, 9)]
    bufs = []

    def up(x):
        x = np.ascontiguousarray(x)
        ptr = eng.device_alloc(x.nbytes)
        bufs.append(ptr)
        eng.device_upload(ptr, x)
        return ptr

    try:
        dg, da, db = up(g), up(a), up(b)
        gs, ws = g.shape[1] * 16, P.glwe_len * 8
        zeros = np.zeros((len(units), P.glwe_len), dtype=np.uint64)
        out_s, out_p = up(zeros), up(zeros)

        def rows(out, padded):
            t = []
            i = 0
            while i < len(units):
                j = i
                while j < len(units) and units[j][0] == units[i][0]:
                    j += 1
                n = (j - i + 3) // 4 * 4 if padded else j - i
                for k in range(i, i + n):
                    s, ia, ib = units[k] if k < j else units[i]
                    t += [dg + s * gs, 0 if ia is None else da + ia * ws, db + ib * ws, out + k * ws if k < j else 0]
                i = j
            return np.array(t, dtype=np.uint64)

        tp, ts = rows(out_p, False), rows(out_s, True)
        assert ts.size == 16 * 4 and tp.size == 4 * len(units)
        eng.cmux_scattered_dev(None, len(units), up(tp))
        eng.cmux_shared_dev(None, ts.size // 16, up(ts))
        assert eng.last_cmux_kernel() == "cmux_shared_kernel"
        got_s, got_p = np.empty_like(zeros), np.empty_like(zeros)
        eng.device_download(None, got_s, out_s)
        eng.device_download(None, got_p, out_p)
    finally:
        for ptr in bufs:
            eng.device_free(ptr)
    assert np.array_equal(got_s, got_p)
    zero = np.zeros(P.glwe_len, dtype=np.uint64)
    for k, (s, ia, ib) in enumerate(units):
        exp = O.cmux(zero if ia is None else a[ia], b[ib], g[s], P.N, P.k, P.cbs_radix_log, P.cbs_count)
        assert np.array_equal(got_s[k], exp), k
#endif
