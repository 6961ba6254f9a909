// spf_generic.hpp — the path for ANY power-of-two polynomial degree 16 <= N <= 2048, any GLWE size k and any gadget radix
// (VERDICT r04 "missing #3").
//
// The specialised kernels of spf_kernels.hpp are built for the one parameter set `parasol_runtime` ships (N = 2048, k = 1,
// PBS radix 2 x 16); the reference's FUNCTIONS are generic (`generalized_programmable_bootstrap`,
// sunscreen_tfhe/src/ops/bootstrapping/programmable_bootstrapping.rs:342-410; `cmux`, ops/fft_ops.rs:149-181) and its own
// functional tests run at TEST_GLWE_DEF_1 = (N 128, k 2) / TEST_RADIX = 3 x 4 bits (high_level.rs:9-58, fft_ops.rs:537-619).
// This family is the correctness path for every other parameter set: one 256-thread workgroup per ciphertext, polynomials and
// spectra in LDS, nothing tuned.  The arithmetic is the oracle's for N != 2048 operation for operation (the C restatement under the checker directory):
//   transform  z_j = (x_j + i x_{j+N/2}) * root(j, 2N) (non-fused), then the in-place radix-2 DIT of `fft_generic`
//              (bit reversal, stages m = 2 .. N/2, butterfly t = cmul_tw(a[k+j+m/2], w), w = root(j, m) conjugated for the
//              forward direction; root(j, m) is read as root(j * (N/2)/m, N/2): the same bits, by construction of the table);
//   MAD        the AVX-512 order, four FMAs (simd/x86_64/avx512.rs:54-57);
//   inverse    (y * (1/(N/2))) * conj(root(j, 2N)) non-fused, round half away, mod 2^64 by fma, saturating cast.
// For N = 2048 the canonical transform is DAG-I (the tuned kernels'): `generic_fft1024_dag1` is its array form, so a context with
// N = 2048 and another radix (the reference's `can_generalized_bootstrap` runs 3 x 4 bits at N = 2048) computes the same bits the
// tuned kernels and the checker define.
#pragma once
#include "spf_device.hpp"

namespace spf {

struct GenericShape {
    uint32_t N, logN, k;      // polynomial degree (power of two), its log2, GLWE size
    const c64* twist;         // [N/2]  e^{+2 pi i j / (2N)}
    const c64* w;             // N < 2048: [N/4] e^{+2 pi i j / (N/2)};  N = 2048: the tuned kernels' table image (DAG-I: T1, T2, WC)
};

constexpr int kGenericThreads = 256;

// LDS of the generic kernels: spectra accumulators (k+1) x N/2 c64, one work transform N/2 c64, digit state N u64,
// and (blind rotation only) the accumulator (k+1) x N u64
__host__ __device__ inline size_t generic_lds_bytes(uint32_t N, uint32_t k, bool with_acc)
{
    // (the digit state doubles as the second image of DAG-I's passes at N = 2048: N u64 = 1024 c64 — same bytes)
    return (size_t)(k + 1) * (N / 2) * 16 + (size_t)(N / 2) * 16 + (size_t)N * 16 + (with_acc ? (size_t)(k + 1) * N * 8 : 0);
}

__device__ inline void generic_fft(c64* a, uint32_t len, uint32_t loglen, int dir, const c64* w_tab);

// DAG-I's 1024-point transform on a[1024] in LDS with tmp[1024] as the second image, array form of the tuned
// kernels' register code: the same butterflies (`radix8`), twiddle products and tables, so the same bits.  128 threads each
// carry one radix-8 of one of the two 512-point transforms per pass.
__device__ inline void generic_fft1024_dag1(c64* a, c64* tmp, int dir, const c64* tab)
{
    const uint32_t tid = threadIdx.x;
    const uint32_t s = tid >> 6, l = tid & 63; // (tid < 128: sub-transform s, task l)
    auto fft512_passes = [&](auto DIRC) {
        constexpr int DIR = decltype(DIRC)::value;
        // entry: sub-transform s has its 512 inputs at a[s * 512 + n'] (time order n' = 64 n1 + n0)
        c64 v[8];
        if (tid < 128) {
#pragma unroll
            for (int n1 = 0; n1 < 8; n1++) v[n1] = a[s * 512 + 64 * n1 + l];
            radix8<DIR>(v);
#pragma unroll
            for (int k1 = 1; k1 < 8; k1++) v[k1] = cmul_tw<DIR>(v[k1], tab[kT1Off + (k1 - 1) * 64 + l]); // W512^{n0 k1}, n0 = l
#pragma unroll
            for (int k1 = 0; k1 < 8; k1++) tmp[s * 512 + l * 8 + k1] = v[k1];                            // y[n0][k1]
        }
        __syncthreads();
        if (tid < 128) {
            const uint32_t k1 = l >> 3, b = l & 7;
#pragma unroll
            for (int aa = 0; aa < 8; aa++) v[aa] = tmp[s * 512 + (8 * aa + b) * 8 + k1];
            radix8<DIR>(v);
#pragma unroll
            for (int c = 1; c < 8; c++) v[c] = cmul_tw<DIR>(v[c], tab[kT2Off + (c - 1) * 8 + b]);          // W64^{b c}
#pragma unroll
            for (int c = 0; c < 8; c++) a[s * 512 + k1 * 64 + b * 8 + c] = v[c];                          // g[k1][b][c]
        }
        __syncthreads();
        if (tid < 128) {
            const uint32_t k1 = l >> 3, c = l & 7;
#pragma unroll
            for (int b = 0; b < 8; b++) v[b] = a[s * 512 + k1 * 64 + b * 8 + c];
            radix8<DIR>(v);
#pragma unroll
            for (int d = 0; d < 8; d++) tmp[s * 512 + k1 + 8 * c + 64 * d] = v[d];                        // X[k1 + 8c + 64d]
        }
        __syncthreads();
        // exit: sub-transform s has its 512 outputs at tmp[s * 512 + k']
    };
    if (dir > 0) {
        // split by parity: E <- a[2n'], O <- a[2n'+1]
        for (uint32_t i = tid; i < 1024; i += kGenericThreads) tmp[(i & 1) * 512 + (i >> 1)] = a[i];
        __syncthreads();
        for (uint32_t i = tid; i < 1024; i += kGenericThreads) a[i] = tmp[i];
        __syncthreads();
        fft512_passes(std::integral_constant<int, +1>{});
        for (uint32_t k = tid; k < 512; k += kGenericThreads) {
            const c64 t = cmul_tw<+1>(tmp[512 + k], tab[kWCOff + k]);
            const c64 E = tmp[k];
            a[k] = cadd(E, t);
            a[k + 512] = csub(E, t);
        }
        __syncthreads();
    } else {
        for (uint32_t k = tid; k < 512; k += kGenericThreads) {
            const c64 x0 = a[k], x1 = a[k + 512];
            tmp[k] = cadd(x0, x1);
            tmp[512 + k] = cmul_tw<-1>(csub(x0, x1), tab[kWCOff + k]);
        }
        __syncthreads();
        for (uint32_t i = tid; i < 1024; i += kGenericThreads) a[i] = tmp[i];
        __syncthreads();
        fft512_passes(std::integral_constant<int, -1>{});
        for (uint32_t i = tid; i < 1024; i += kGenericThreads) a[i] = tmp[(i & 1) * 512 + (i >> 1)];
        __syncthreads();
    }
}

// the transform of a context: the checker's radix-2 DIT for N < 2048, DAG-I for N = 2048
__device__ inline void generic_transform(const GenericShape& g, c64* a, c64* tmp, int dir)
{
    if (g.N == 2048) generic_fft1024_dag1(a, tmp, dir, g.w);
    else generic_fft(a, g.N / 2, g.logN - 1, dir, g.w);
}

// in-place radix-2 DIT of a[len] (LDS), every thread of the block takes part; dir > 0: forward (conjugated twiddles)
__device__ inline void generic_fft(c64* a, uint32_t len, uint32_t loglen, int dir, const c64* w_tab)
{
    const uint32_t tid = threadIdx.x;
    for (uint32_t i = tid; i < len; i += kGenericThreads) {
        const uint32_t j = loglen ? (__brev(i) >> (32 - loglen)) : 0;
        if (i < j) { const c64 t = a[i]; a[i] = a[j]; a[j] = t; }
    }
    __syncthreads();
    uint32_t logm = 1;
    for (uint32_t m = 2; m <= len; m <<= 1, logm++) {
        const uint32_t half = m >> 1;
        for (uint32_t idx = tid; idx < (len >> 1); idx += kGenericThreads) {
            const uint32_t j = idx & (half - 1), base = (idx >> (logm - 1)) << logm;
            c64 w;
            if (m >= 8) w = w_tab[j * (len / m)];
            else if (m == 2) w = {1.0, 0.0};
            else w = j == 0 ? c64{1.0, 0.0} : c64{0.0, 1.0};
            if (dir > 0) w.im = -w.im;
            const c64 t = cmul_tw<+1>(a[base + j + half], w);
            const c64 u = a[base + j];
            a[base + j] = cadd(u, t);
            a[base + j + half] = csub(u, t);
        }
        __syncthreads();
    }
}

__device__ inline uint64_t generic_radix_round(uint64_t x, uint32_t radix_log, uint32_t count)
{
    const uint32_t shift = 64 - radix_log * count;
    return (x >> shift) + ((x >> (shift - 1)) & 1);
}
__device__ inline uint64_t generic_next_digit(uint64_t& s, uint32_t radix_log)
{
    const uint64_t mask = ((uint64_t)1 << radix_log) - 1;
    const uint64_t digit = s & mask;
    s >>= radix_log;
    const uint64_t carry = digit >> (radix_log - 1);
    s += carry;
    return digit - (carry << radix_log);
}

// decomposed_polynomial_glev_mad (fft_ops.rs:67-98 -> :107-124): accf += <decomp(poly), glev>, GLEV entries in reverse;
// `coef(i)` yields coefficient i of the polynomial.  glev: [level<count][poly<k+1][bin<N/2].
template <class COEF>
__device__ inline void generic_glev_mad(const GenericShape& g, c64* accf, c64* buf, uint64_t* state, const c64* glev,
                                        uint32_t radix_log, uint32_t count, COEF coef)
{
    const uint32_t tid = threadIdx.x, N = g.N, h = N / 2, k = g.k;
    {
        for (uint32_t i = tid; i < N; i += kGenericThreads) state[i] = generic_radix_round(coef(i), radix_log, count);
        __syncthreads();
        for (uint32_t j = 0; j < count; j++) {
            // next digit of every coefficient (each state word has ONE owner: thread i mod 256 for i and for i + N/2 alike)
            for (uint32_t t = tid; t < h; t += kGenericThreads) {
                uint64_t s0 = state[t], s1 = state[t + h];
                const double re = (double)(int64_t)generic_next_digit(s0, radix_log);
                const double im = (double)(int64_t)generic_next_digit(s1, radix_log);
                state[t] = s0; state[t + h] = s1;
                buf[t] = cmul_nf({re, im}, g.twist[t]);
            }
            __syncthreads();
            generic_transform(g, buf, reinterpret_cast<c64*>(state + N), +1);
            // GLEV entries are consumed in reverse (fft_ops.rs:92)
            const c64* row = glev + (size_t)(count - 1 - j) * (size_t)(k + 1) * h;
            for (uint32_t q = 0; q <= k; q++)
                for (uint32_t t = tid; t < h; t += kGenericThreads) {
                    const c64 a = row[(size_t)q * h + t], b = buf[t], c = accf[q * h + t];
                    double re = __builtin_fma(a.re, b.re, c.re);
                    double im = __builtin_fma(a.re, b.im, c.im);
                    re = __builtin_fma(-a.im, b.im, re);
                    im = __builtin_fma(a.im, b.re, im);
                    accf[q * h + t] = {re, im};
                }
            __syncthreads();
        }
    }
}

// glwe_ggsw_mad (fft_ops.rs:23-56) into accf (cleared here); `coef(p, i)` yields coefficient i of polynomial p of the GLWE being
// multiplied.  ggsw: [row<k+1][level<count][poly<k+1][bin<N/2].
template <class COEF>
__device__ inline void generic_glwe_ggsw_mad(const GenericShape& g, c64* accf, c64* buf, uint64_t* state, const c64* ggsw,
                                             uint32_t radix_log, uint32_t count, COEF coef)
{
    const uint32_t tid = threadIdx.x, h = g.N / 2, k = g.k;
    for (uint32_t i = tid; i < (k + 1) * h; i += kGenericThreads) accf[i] = {0.0, 0.0};
    __syncthreads();
    for (uint32_t p = 0; p <= k; p++)
        generic_glev_mad(g, accf, buf, state, ggsw + (size_t)p * count * (size_t)(k + 1) * h, radix_log, count,
                         [&](uint32_t i) { return coef(p, i); });
}

// PolynomialRef::fft of a full-range torus polynomial (entities/polynomial.rs:257-274) into spec[N/2]; `coef(i)` its words
template <class COEF>
__device__ inline void generic_poly_fft(const GenericShape& g, c64* spec, c64* tmp, COEF coef)
{
    const uint32_t tid = threadIdx.x, h = g.N / 2;
    for (uint32_t t = tid; t < h; t += kGenericThreads)
        spec[t] = cmul_nf({(double)(int64_t)coef(t), (double)(int64_t)coef(t + h)}, g.twist[t]);
    __syncthreads();
    generic_transform(g, spec, tmp, +1);
}

// PolynomialFftRef::ifft of spectrum q of accf (entities/polynomial_fft.rs:82-99): `sink(i, torus word)` for i < N
template <class SINK>
__device__ inline void generic_poly_ifft(const GenericShape& g, const c64* spec, c64* buf, c64* tmp, SINK sink)
{
    const uint32_t tid = threadIdx.x, h = g.N / 2;
    for (uint32_t t = tid; t < h; t += kGenericThreads) buf[t] = spec[t];
    __syncthreads();
    generic_transform(g, buf, tmp, -1);
    const double n_inv = 1.0 / (double)h;
    for (uint32_t t = tid; t < h; t += kGenericThreads) {
        const c64 y = buf[t], tw = g.twist[t];
        const c64 xs = {y.re * n_inv, y.im * n_inv};
        const c64 u = cmul_nf(xs, {tw.re, -tw.im});
        sink(t, f64_round_to_torus(u.re));
        sink(t + h, f64_round_to_torus(u.im));
    }
    __syncthreads();
}

__device__ inline uint32_t generic_mod_switch(uint64_t x, uint32_t log_chi, uint32_t log_v, uint32_t log_modulus)
{
    x <<= log_chi;
    const uint32_t shift = 64 - (log_modulus - log_v);
    const uint64_t round = (x >> (shift - 1)) & 1;
    x >>= shift;
    return (uint32_t)(((x + round) & (((uint64_t)1 << log_modulus) - 1)) << log_v);
}

struct GenericPbsArgs {
    GenericShape g;
    const uint64_t* lwe_in;  // B x (n + 1)
    const uint64_t* lut;     // (k+1) N words, or B of them
    size_t lut_stride;
    const c64* bsk;          // [n][k+1][count][k+1][N/2]
    uint64_t* out;
    size_t out_stride;
    uint32_t n, B, radix_log, count, log_chi, log_v, sample_extract;
    uint64_t body_rotate;
};

// generalized_programmable_bootstrap (programmable_bootstrapping.rs:342-410), one workgroup per ciphertext
__global__ __launch_bounds__(kGenericThreads) void generic_pbs_kernel(GenericPbsArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const GenericShape& g = a.g;
    const uint32_t tid = threadIdx.x, N = g.N, h = N / 2, k = g.k, len = (k + 1) * N;
    c64* accf = reinterpret_cast<c64*>(smem);
    c64* buf = accf + (size_t)(k + 1) * h;
    uint64_t* state = reinterpret_cast<uint64_t*>(buf + h);
    c64* tmp = reinterpret_cast<c64*>(state + N);
    uint64_t* acc = state + 2 * N;
    const uint32_t ct = blockIdx.x;
    const uint64_t* lwe = a.lwe_in + (size_t)ct * (a.n + 1);
    const uint64_t* lut = a.lut + (size_t)ct * a.lut_stride;
    const uint32_t log_modulus = g.logN + 1;
    // V_0 * X^{-b}: out[i] = +-lut[(i + b) mod N] (entities/polynomial.rs:171-201)
    {
        const uint32_t bt = generic_mod_switch(lwe[a.n] + a.body_rotate, a.log_chi, a.log_v, log_modulus);
        for (uint32_t i = tid; i < len; i += kGenericThreads) {
            const uint32_t p = i / N, c = i % N, idx = c + bt;
            const uint64_t v = lut[p * N + (idx & (N - 1))];
            acc[i] = ((idx >> g.logN) & 1) ? (uint64_t)0 - v : v;
        }
    }
    __syncthreads();
    const size_t ggsw_len = (size_t)(k + 1) * a.count * (k + 1) * h;
    for (uint32_t step = 0; step < a.n; step++) {
        const uint32_t at = generic_mod_switch(lwe[step], a.log_chi, a.log_v, log_modulus);
        // cmux(acc, acc, acc * X^{a}, BSK_step): diff = acc X^a - acc; (acc X^a)[i] = +-acc[(i - a) mod N] (polynomial.rs:208-236)
        generic_glwe_ggsw_mad(g, accf, buf, state, a.bsk + (size_t)step * ggsw_len, a.radix_log, a.count, [&](uint32_t p, uint32_t i) {
            const uint32_t idx = i + 2 * N - at;
            const uint64_t v = acc[p * N + (idx & (N - 1))];
            const uint64_t rot = ((idx >> g.logN) & 1) ? (uint64_t)0 - v : v;
            return rot - acc[p * N + i];
        });
        for (uint32_t q = 0; q <= k; q++)
            generic_poly_ifft(g, accf + (size_t)q * h, buf, tmp, [&](uint32_t i, uint64_t t) { acc[q * N + i] += t; });
    }
    if (!a.sample_extract) {
        uint64_t* out = a.out + (size_t)ct * a.out_stride;
        for (uint32_t i = tid; i < len; i += kGenericThreads) out[i] = acc[i];
    } else { // sample_extract(., 0) (glwe_ciphertext_ops.rs:31-76)
        uint64_t* out = a.out + (size_t)ct * a.out_stride;
        for (uint32_t i = tid; i < k * N; i += kGenericThreads) {
            const uint32_t p = i / N, j = i % N;
            out[i] = j == 0 ? acc[p * N] : (uint64_t)0 - acc[p * N + N - j];
        }
        if (tid == 0) out[k * N] = acc[k * N];
    }
}

struct GenericCmuxArgs {
    GenericShape g;
    const c64* ggsw;      // units / per_ggsw selectors
    const uint64_t* d0;   // low operand (taken when the selector encrypts 0); ignored when d0_zero
    const uint64_t* d1;
    uint64_t* out;
    uint32_t units, per_ggsw, d0_zero, radix_log, count;
    const void* const* ptrs; // non-null: 4 pointers per unit {selector, d0 (null = the zero ciphertext), d1, out} instead (gate graphs, values)
};

// cmux (fft_ops.rs:149-181): out = d0 + IFFT(decomp(d1 - d0) [*] ggsw); with d0_zero: multiply_glwe_ggsw
__global__ __launch_bounds__(kGenericThreads) void generic_cmux_kernel(GenericCmuxArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const GenericShape& g = a.g;
    const uint32_t tid = threadIdx.x, N = g.N, h = N / 2, k = g.k, len = (k + 1) * N;
    c64* accf = reinterpret_cast<c64*>(smem);
    c64* buf = accf + (size_t)(k + 1) * h;
    uint64_t* state = reinterpret_cast<uint64_t*>(buf + h);
    const uint32_t u = blockIdx.x;
    const uint64_t* d0 = a.d0 + (size_t)u * len;
    const uint64_t* d1 = a.d1 + (size_t)u * len;
    uint64_t* out = a.out + (size_t)u * len;
    const c64* ggsw = a.ggsw + (size_t)(u / a.per_ggsw) * (size_t)(k + 1) * a.count * (k + 1) * h;
    bool zero = a.d0_zero != 0;
    if (a.ptrs) {
        const void* const* q = a.ptrs + (size_t)u * 4;
        ggsw = static_cast<const c64*>(q[0]);
        d1 = static_cast<const uint64_t*>(q[2]);
        zero = q[1] == nullptr;
        d0 = zero ? d1 : static_cast<const uint64_t*>(q[1]);
        out = static_cast<uint64_t*>(const_cast<void*>(q[3]));
    }
    generic_glwe_ggsw_mad(g, accf, buf, state, ggsw, a.radix_log, a.count,
                          [&](uint32_t p, uint32_t i) { return zero ? d1[p * N + i] : d1[p * N + i] - d0[p * N + i]; });
    for (uint32_t q = 0; q <= k; q++)
        generic_poly_ifft(g, accf + (size_t)q * h, buf, reinterpret_cast<c64*>(state + N), [&](uint32_t i, uint64_t t) { out[q * N + i] = zero ? t : t + d0[q * N + i]; });
    (void)tid;
}

// sample_extract at index h of every GLWE (glwe_ciphertext_ops.rs:31-76)
__global__ void generic_sample_extract_kernel(const uint64_t* glwe, uint64_t* lwe, uint32_t B, uint32_t N, uint32_t k, uint32_t hidx)
{
    const uint32_t ct = blockIdx.y;
    const uint64_t* a = glwe + (size_t)ct * (k + 1) * N;
    uint64_t* o = lwe + (size_t)ct * (k * N + 1);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i <= k * N; i += gridDim.x * blockDim.x) {
        if (i == k * N) { o[i] = a[k * N + hidx]; continue; }
        const uint32_t p = i / N, j = i % N;
        o[i] = j <= hidx ? a[p * N + hidx - j] : (uint64_t)0 - a[p * N + hidx + N - j];
    }
    (void)B;
}

// KeylessEvaluation::{not, xor, mul_xn} (crypto/evaluation.rs:47-66) for any (N, k): op 0 / 1 / 2
__global__ void generic_linear_kernel(const uint64_t* a, const uint64_t* b, uint64_t* out, uint32_t N, uint32_t logN, uint32_t k,
                                      uint32_t op, uint32_t n)
{
    const uint32_t ct = blockIdx.y, len = (k + 1) * N;
    const uint64_t* x = a + (size_t)ct * len;
    uint64_t* o = out + (size_t)ct * len;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < len; i += gridDim.x * blockDim.x) {
        if (op == 0) o[i] = x[i] + (i == k * N ? (uint64_t)1 << 63 : 0);
        else if (op == 1) o[i] = x[i] + b[(size_t)ct * len + i];
        else { // out = in * X^n: out[c] = +-in[(c - n) mod N]
            const uint32_t p = i / N, c = i % N, idx = c + 2 * N - n;
            const uint64_t v = x[p * N + (idx & (N - 1))];
            o[i] = ((idx >> logN) & 1) ? (uint64_t)0 - v : v;
        }
    }
}

struct GenericTraceArgs {
    GenericShape g;
    const uint64_t* glwe_in; // B x (k+1) N: lo-noise GLWE out of the bootstrap
    uint64_t* glev_out;      // B x cbs_count x (k+1) N
    const c64* ak;           // [log2 N][row<k][level<tr_count][poly<k+1][N/2]
    uint32_t units, cbs_count, cbs_radix_log, tr_radix_log, tr_count;
};
__host__ __device__ inline size_t generic_trace_lds_bytes(uint32_t N, uint32_t k)
{
    return generic_lds_bytes(N, k, true) + (size_t)(k + 1) * N * 8; // + the automorphed copy
}

// mod_switch_trace_and_rotate for one (ciphertext, gadget level) (circuit_bootstrapping.rs:260-298): un-rotate, X^-level,
// shift-round by log2 N, then the trace (automorphisms/mod.rs:53-85) = log2 N rounds of automorphism + FFT-domain GLWE
// keyswitch (fft_ops.rs:457-495)
__global__ __launch_bounds__(kGenericThreads) void generic_trace_kernel(GenericTraceArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const GenericShape& g = a.g;
    const uint32_t tid = threadIdx.x, N = g.N, h = N / 2, k = g.k, len = (k + 1) * N;
    c64* accf = reinterpret_cast<c64*>(smem);
    c64* buf = accf + (size_t)(k + 1) * h;
    uint64_t* state = reinterpret_cast<uint64_t*>(buf + h);
    c64* tmp = reinterpret_cast<c64*>(state + N);
    uint64_t* X = state + 2 * N;
    uint64_t* G = X + len;
    const uint32_t unit = blockIdx.x, ct = unit / a.cbs_count, lvl = unit % a.cbs_count;
    const uint64_t* in = a.glwe_in + (size_t)ct * len;
    for (uint32_t i = tid; i < len; i += kGenericThreads) {
        const uint32_t p = i / N, c = i % N, idx = c + lvl; // X^-lvl: out[c] = +-in[c + lvl]
        const uint32_t src = idx & (N - 1);
        uint64_t v = in[p * N + src];
        if (p == k && src <= lvl) v += (uint64_t)1 << (64 - (a.cbs_radix_log * (src + 1) + 1)); // body coefficients 0..lvl un-rotated
        v = ((idx >> g.logN) & 1) ? (uint64_t)0 - v : v;
        X[i] = (v >> g.logN) + ((v >> (g.logN - 1)) & 1); // glwe_mod_switch_and_expand_pow_2
    }
    __syncthreads();
    const size_t glev_len = (size_t)a.tr_count * (k + 1) * h, ksk_len = (size_t)k * glev_len;
    for (uint32_t it = 1; it <= g.logN; it++) {
        const uint32_t kk = (N >> (it - 1)) + 1;
        // polynomial_pow_k (ops/polynomial/mod.rs:62-84): p_k[(i kk) mod N] = +-p[i], minus when (i kk) / N is odd
        for (uint32_t i = tid; i < len; i += kGenericThreads) {
            const uint32_t p = i / N, c = i % N, prod = c * kk;
            const uint64_t v = X[i];
            G[p * N + (prod & (N - 1))] = ((prod >> g.logN) & 1) ? (uint64_t)0 - v : v;
        }
        for (uint32_t i = tid; i < (k + 1) * h; i += kGenericThreads) accf[i] = {0.0, 0.0};
        __syncthreads();
        const c64* ksk = a.ak + (size_t)(it - 1) * ksk_len;
        for (uint32_t r = 0; r < k; r++)
            generic_glev_mad(g, accf, buf, state, ksk + (size_t)r * glev_len, a.tr_radix_log, a.tr_count,
                             [&](uint32_t i) { return G[r * N + i]; });
        // out += trivial(b_k) - ks
        for (uint32_t q = 0; q <= k; q++)
            generic_poly_ifft(g, accf + (size_t)q * h, buf, tmp, [&](uint32_t i, uint64_t t) {
                X[q * N + i] += (q == k ? G[k * N + i] : (uint64_t)0) - t;
            });
    }
    uint64_t* out = a.glev_out + (size_t)unit * len;
    for (uint32_t i = tid; i < len; i += kGenericThreads) out[i] = X[i];
}

struct GenericSchemeSwitchArgs {
    GenericShape g;
    const uint64_t* glev;  // B x cbs_count x (k+1) N
    c64* ggsw_out;         // B x [row<k+1][level<cbs_count][poly<k+1][N/2]
    const c64* ssk;        // [pair][level<ss_count][poly<k+1][N/2], pairs upper-triangular (entities/scheme_switch_key.rs)
    uint32_t units, cbs_count, ss_radix_log, ss_count;
};

// scheme_switch_fft for one (ciphertext, gadget level): the k+1 rows of that level (fft_ops.rs:225-279, 403-442)
__global__ __launch_bounds__(kGenericThreads) void generic_scheme_switch_kernel(GenericSchemeSwitchArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const GenericShape& g = a.g;
    const uint32_t tid = threadIdx.x, N = g.N, h = N / 2, k = g.k, len = (k + 1) * N;
    c64* y = reinterpret_cast<c64*>(smem);
    c64* buf = y + (size_t)(k + 1) * h;
    uint64_t* state = reinterpret_cast<uint64_t*>(buf + h);
    const uint32_t unit = blockIdx.x, ct = unit / a.cbs_count, lvl = unit % a.cbs_count;
    const uint64_t* x = a.glev + (size_t)unit * len;
    const size_t glwe_fft_len = (size_t)(k + 1) * h, ss_glev_len = (size_t)a.ss_count * glwe_fft_len;
    c64* out_ct = a.ggsw_out + (size_t)ct * (k + 1) * a.cbs_count * glwe_fft_len;
    for (uint32_t j = 0; j <= k; j++) {
        c64* dst = out_ct + ((size_t)j * a.cbs_count + lvl) * glwe_fft_len;
        if (j == k) {
            for (uint32_t p = 0; p <= k; p++) generic_poly_fft(g, y + (size_t)p * h, reinterpret_cast<c64*>(state + N), [&](uint32_t i) { return x[p * N + i]; });
        } else {
            for (uint32_t i = tid; i < (k + 1) * h; i += kGenericThreads) y[i] = {0.0, 0.0};
            __syncthreads();
            generic_poly_fft(g, y + (size_t)j * h, reinterpret_cast<c64*>(state + N), [&](uint32_t i) { return x[k * N + i]; }); // y.a[j] = FFT(x.b)
            for (uint32_t r = 0; r < k; r++) {
                const uint32_t row = j <= r ? j : r, col = j <= r ? r : j; // get_linear_index of the upper triangle
                const size_t pair = (size_t)(k * (k + 1) / 2) - (size_t)(k - row) * ((k - row) + 1) / 2 + col - row;
                generic_glev_mad(g, y, buf, state, a.ssk + pair * ss_glev_len, a.ss_radix_log, a.ss_count,
                                 [&](uint32_t i) { return x[r * N + i]; });
            }
        }
        for (uint32_t i = tid; i < (k + 1) * h; i += kGenericThreads) dst[i] = y[i];
        __syncthreads();
    }
}

} // namespace spf
