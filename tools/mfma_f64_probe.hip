// mfma_f64_probe — does v_mfma_f64_4x4x4_4b / v_mfma_f64_16x16x4 accumulate its K = 4 products as the
// sequential chain  fma(a3,b3, fma(a2,b2, fma(a1,b1, fma(a0,b0,c))))  (what glwe_polynomial_mad's
// AVX-512 order needs, sunscreen_tfhe/src/ops/fft_ops.rs:107-124, simd/x86_64/avx512.rs:54-57), or in
// some other association / with some other rounding?
//
// Method: (1) find the operand layout by one-hot probing (which k does a lane's A / B value carry);
// (2) feed every row/column the same (a_k), (b_k), c — integers below 2^31, so every exact sum fits
// __int128 — and compare the instruction's output bit for bit with candidate evaluation orders computed
// on the host.  Build: hipcc --offload-arch=gfx950 -O2 -ffp-contract=off tools/mfma_f64_probe.hip -o mfma_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// one wave; in: a[64], b[64], c[64]  out: d[64]   (4x4x4, 4 blocks: one f64 per lane for A, B and C/D)
__global__ void mfma4(const double* a, const double* b, const double* c, double* d)
{
    int l = threadIdx.x;
    d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], c[l], 0, 0, 0);
}
// 16x16x4: A, B one f64 per lane, C/D four per lane
__global__ void mfma16(const double* a, const double* b, const double* c, double* d)
{
    int l = threadIdx.x;
    typedef double d4 __attribute__((ext_vector_type(4)));
    d4 cc = {c[4 * l], c[4 * l + 1], c[4 * l + 2], c[4 * l + 3]};
    d4 r = __builtin_amdgcn_mfma_f64_16x16x4f64(a[l], b[l], cc, 0, 0, 0);
    for (int i = 0; i < 4; i++) d[4 * l + i] = r[i];
}
// batched order test for the 4x4x4 form: sets[n][9] = a0..a3, b0..b3, c ; ka / kb = the k a lane carries
__global__ void order4(const double* sets, const int* ka, const int* kb, double* out, int n)
{
    int l = threadIdx.x;
    for (int i = 0; i < n; i++) {
        const double* s = sets + 9 * i;
        double r = __builtin_amdgcn_mfma_f64_4x4x4f64(s[ka[l]], s[4 + kb[l]], s[8], 0, 0, 0);
        const double r0 = __shfl(r, 0);
        if (l == 0) out[i] = r;
        else if (__double_as_longlong(r) != __double_as_longlong(r0)) out[n] = (double)l; // every (block, i, j) must agree
    }
}
__global__ void order16(const double* sets, const int* ka, const int* kb, double* out, int n)
{
    int l = threadIdx.x;
    typedef double d4 __attribute__((ext_vector_type(4)));
    for (int i = 0; i < n; i++) {
        const double* s = sets + 9 * i;
        d4 cc = {s[8], s[8], s[8], s[8]};
        d4 r = __builtin_amdgcn_mfma_f64_16x16x4f64(s[ka[l]], s[4 + kb[l]], cc, 0, 0, 0);
        if (l == 0) out[i] = r[0];
    }
}

static double from_i128(__int128 v) // round to nearest even, once
{
    bool neg = v < 0;
    unsigned __int128 u = neg ? (unsigned __int128)(-v) : (unsigned __int128)v;
    if (u == 0) return 0.0;
    int hb = 127;
    while (!((u >> hb) & 1)) hb--;
    double r;
    if (hb <= 52) r = (double)(uint64_t)u;
    else {
        int sh = hb - 52;
        uint64_t m = (uint64_t)(u >> sh);
        unsigned __int128 rem = u & (((unsigned __int128)1 << sh) - 1), half = (unsigned __int128)1 << (sh - 1);
        if (rem > half || (rem == half && (m & 1))) m++;
        r = ldexp((double)m, sh);
    }
    return neg ? -r : r;
}

// cycles per instruction, one wave, 8 independent accumulators, 4096 instructions each
__global__ void rate_kernel(double* out, uint64_t* cyc, double x, double y)
{
    typedef double d4 __attribute__((ext_vector_type(4)));
    double acc[8];
    d4 acc4[8];
    for (int i = 0; i < 8; i++) { acc[i] = (double)i; acc4[i] = d4{(double)i, 1.0, 2.0, 3.0}; }
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 512; it++)
#pragma unroll
        for (int i = 0; i < 8; i++) acc[i] = __builtin_fma(x, y, acc[i]);
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 512; it++)
#pragma unroll
        for (int i = 0; i < 8; i++) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, acc[i], 0, 0, 0);
    uint64_t t2 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 512; it++)
#pragma unroll
        for (int i = 0; i < 8; i++) acc4[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc4[i], 0, 0, 0);
    uint64_t t3 = __builtin_amdgcn_s_memtime();
    double sum = 0;
    for (int i = 0; i < 8; i++) sum += acc[i] + acc4[i][0] + acc4[i][1] + acc4[i][2] + acc4[i][3];
    out[threadIdx.x] = sum;
    if (threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; cyc[2] = t3 - t2; }
}

int main()
{
    {
        double* o; uint64_t* cy;
        CK(hipMalloc(&o, 64 * 8)); CK(hipMalloc(&cy, 3 * 8));
        hipLaunchKernelGGL(rate_kernel, 1, 64, 0, 0, o, cy, 1.0000001, 0.9999999);
        hipLaunchKernelGGL(rate_kernel, 1, 64, 0, 0, o, cy, 1.0000001, 0.9999999);
        uint64_t h[3];
        CK(hipMemcpy(h, cy, 24, hipMemcpyDeviceToHost));
        printf("one wave, 8 independent accumulators, cycles per instruction: v_fma_f64 %.2f (64 FMA), "
               "v_mfma_f64_4x4x4_4b %.2f (256 FMA), v_mfma_f64_16x16x4 %.2f (1024 FMA)\n",
               h[0] / 4096.0, h[1] / 4096.0, h[2] / 4096.0);
    }
    double *da, *db, *dc, *dd;
    CK(hipMalloc(&da, 64 * 8)); CK(hipMalloc(&db, 64 * 8)); CK(hipMalloc(&dc, 256 * 8)); CK(hipMalloc(&dd, 256 * 8));
    std::vector<double> ha(64), hb(64), hc(256, 0.0), hd(256);
    // ---- layout of the 4x4x4_4b form: which k does lane L of A pair with lane M of B
    int ka[64], kb[64];
    {
        // B lane 0 defines k_B(0) =: kref; A lanes with a nonzero product against it share that k
        int pair[64][64];
        for (int L = 0; L < 64; L++)
            for (int M = 0; M < 64; M++) {
                std::fill(ha.begin(), ha.end(), 0.0); std::fill(hb.begin(), hb.end(), 0.0);
                ha[L] = 1.0; hb[M] = 1.0;
                CK(hipMemcpy(da, ha.data(), 512, hipMemcpyHostToDevice));
                CK(hipMemcpy(db, hb.data(), 512, hipMemcpyHostToDevice));
                CK(hipMemcpy(dc, hc.data(), 512, hipMemcpyHostToDevice));
                hipLaunchKernelGGL(mfma4, 1, 64, 0, 0, da, db, dc, dd);
                CK(hipMemcpy(hd.data(), dd, 512, hipMemcpyDeviceToHost));
                int nz = 0;
                for (int i = 0; i < 64; i++) nz += hd[i] != 0.0;
                pair[L][M] = nz;
            }
        // one-hot probing (tools/bin run, r02): A lane L meets B lane M iff L / 16 == M / 16 and (L / 4) % 4 ==
        // (M / 4) % 4, and the product lands in D lane 16 (L % 4) + 4 ((L / 4) % 4) + M % 4 — i.e. for this form
        //   A lane = 16 k + 4 block + i,   B lane = 16 k + 4 block + j,   D lane = 16 i + 4 block + j
        bool ok = true;
        for (int L = 0; L < 64; L++)
            for (int M = 0; M < 64; M++) {
                bool expect = (L / 16 == M / 16) && ((L / 4) % 4 == (M / 4) % 4);
                if ((pair[L][M] != 0) != expect) ok = false;
            }
        printf("4x4x4_4b layout: A lane = 16 k + 4 blk + i, B lane = 16 k + 4 blk + j : %s\n", ok ? "confirmed" : "NOT this");
        if (!ok) return 2;
        for (int l = 0; l < 64; l++) ka[l] = kb[l] = l / 16;
    }
    // ---- order test
    const int N = 200000;
    std::vector<double> sets(9 * (size_t)N);
    std::mt19937_64 rng(0x5EED);
    for (int i = 0; i < N; i++) {
        // integers with wide, independent magnitudes so that roundings and cancellations differ by order
        for (int t = 0; t < 9; t++) {
            int bits = 1 + (int)(rng() % 31);
            int64_t v = (int64_t)(rng() & ((1ull << bits) - 1));
            if (rng() & 1) v = -v;
            sets[9 * (size_t)i + t] = (double)v;
        }
        if (i % 4 == 0) { // force a cancellation: a1 b1 = -a0 b0
            sets[9 * (size_t)i + 1] = -sets[9 * (size_t)i + 0];
            sets[9 * (size_t)i + 5] = sets[9 * (size_t)i + 4];
        }
        if (i % 8 == 1) sets[9 * (size_t)i + 8] *= 4294967296.0 * 1048576.0; // large c (2^52 scale)
    }
    double *dsets, *dout; int *dka, *dkb;
    CK(hipMalloc(&dsets, sets.size() * 8)); CK(hipMalloc(&dout, ((size_t)N + 1) * 8));
    CK(hipMalloc(&dka, 256)); CK(hipMalloc(&dkb, 256));
    CK(hipMemcpy(dsets, sets.data(), sets.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dka, ka, 256, hipMemcpyHostToDevice)); CK(hipMemcpy(dkb, kb, 256, hipMemcpyHostToDevice));
    for (int form = 0; form < 2; form++) {
        CK(hipMemset(dout, 0, ((size_t)N + 1) * 8));
        if (form == 0) hipLaunchKernelGGL(order4, 1, 64, 0, 0, dsets, dka, dkb, dout, N);
        else {
            // 16x16x4: A lane = 16 k + i, B lane = 16 k + j  (k = lane / 16)
            int k16[64];
            for (int l = 0; l < 64; l++) k16[l] = l / 16;
            CK(hipMemcpy(dka, k16, 256, hipMemcpyHostToDevice)); CK(hipMemcpy(dkb, k16, 256, hipMemcpyHostToDevice));
            hipLaunchKernelGGL(order16, 1, 64, 0, 0, dsets, dka, dkb, dout, N);
        }
        CK(hipDeviceSynchronize());
        std::vector<double> out((size_t)N + 1);
        CK(hipMemcpy(out.data(), dout, out.size() * 8, hipMemcpyDeviceToHost));
        const char* names[6] = {"chain k = 0,1,2,3 (fma each, c first)", "chain k = 3,2,1,0", "exact sum of c and the four products, rounded once",
                                "c + exact sum of the four products (two roundings)", "pairwise: fma(a1,b1,a0*b0) + fma(a3,b3,a2*b2), + c", "chain of products first, c last"};
        long match[6] = {0, 0, 0, 0, 0, 0};
        long distinct = 0;
        for (int i = 0; i < N; i++) {
            const double* s = &sets[9 * (size_t)i];
            double cand[6];
            double r = s[8];
            for (int k = 0; k < 4; k++) r = fma(s[k], s[4 + k], r);
            cand[0] = r;
            r = s[8];
            for (int k = 3; k >= 0; k--) r = fma(s[k], s[4 + k], r);
            cand[1] = r;
            __int128 ex = 0;
            for (int k = 0; k < 4; k++) ex += (__int128)(int64_t)s[k] * (__int128)(int64_t)s[4 + k];
            // c is an integer of up to 84 bits: rebuild it exactly from mantissa and exponent
            __int128 ci = 0;
            if (s[8] != 0.0) {
                int e;
                double m = frexp(s[8], &e);
                __int128 mi = (__int128)(int64_t)ldexp(m, 53);
                ci = e >= 53 ? mi * ((__int128)1 << (e - 53)) : mi / ((__int128)1 << (53 - e));
            }
            cand[2] = from_i128(ex + ci);
            cand[3] = s[8] + from_i128(ex);
            cand[4] = (fma(s[1], s[5], s[0] * s[4]) + fma(s[3], s[7], s[2] * s[6])) + s[8];
            r = s[0] * s[4];
            for (int k = 1; k < 4; k++) r = fma(s[k], s[4 + k], r);
            cand[5] = r + s[8];
            static int shown = 0;
            if (form == 0 && shown < 4 && memcmp(&cand[0], &out[i], 8) != 0) {
                shown++;
                printf("  set %d: a = %.0f %.0f %.0f %.0f  b = %.0f %.0f %.0f %.0f  c = %.0f -> mfma %.0f, chain %.0f; partial sums c+p0.. :", i,
                       s[0], s[1], s[2], s[3], s[4], s[5], s[6], s[7], s[8], out[i], cand[0]);
                for (int m = 1; m < 16; m++) {
                    double t = s[8];
                    for (int k = 0; k < 4; k++) if (m >> k & 1) t = fma(s[k], s[4 + k], t);
                    printf(" [%x]%.0f", m, t);
                }
                printf("\n");
            }
            bool any_diff = false;
            for (int c = 0; c < 6; c++) {
                if (memcmp(&cand[c], &out[i], 8) == 0) match[c]++;
                if (memcmp(&cand[c], &cand[0], 8) != 0) any_diff = true;
            }
            distinct += any_diff;
        }
        printf("\n%s: %d operand sets (%ld on which the candidate orders disagree with each other)%s\n",
               form == 0 ? "v_mfma_f64_4x4x4_4b_f64" : "v_mfma_f64_16x16x4_f64", N, distinct,
               form == 0 && out[N] != 0.0 ? "  [outputs of different (block, i, j) disagreed!]" : "");
        for (int c = 0; c < 6; c++) printf("  %-62s %7ld / %d bit-equal\n", names[c], match[c], N);
    }
    return 0;
}
