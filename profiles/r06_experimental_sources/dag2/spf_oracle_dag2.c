/*
 * spf_oracle.c — CPU ORACLE (test infrastructure, NOT product code).  See spf_oracle.h.
 *
 * Build: gcc -O2 -std=c11 -ffp-contract=off -fno-fast-math -fPIC -shared (oracle/Makefile).
 * -ffp-contract=off is REQUIRED: every fused multiply-add below is an explicit fma() and every
 * other expression must round after each operation, exactly like the Rust reference (rustc
 * never contracts) and exactly like the HIP kernels (hipcc -ffp-contract=off).
 *
 * Citations are relative to /root/reference/.
 */
#define _GNU_SOURCE
#include "spf_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ======================================================================== integer stages */

/* sunscreen_tfhe/src/ops/ciphertext/lwe_ciphertext_ops.rs:130-142 */
uint64_t spfo_modulus_switch(uint64_t x, uint32_t log_chi, uint32_t log_v, uint32_t log_modulus)
{
    const uint64_t one = 1;
    uint64_t mask = (one << log_modulus) - one;
    x = x << log_chi;
    uint32_t shift_amount = 64 - (log_modulus - log_v);
    uint64_t round = (x >> (shift_amount - 1)) & one;
    x = x >> shift_amount;
    return ((x + round) & mask) << log_v;
}

/* lwe_ciphertext_ops.rs:97-128 */
void spfo_lwe_modulus_switch(uint64_t *ct, size_t n_plus_1, uint32_t log_chi, uint32_t log_v,
                             uint32_t log_modulus)
{
    for (size_t i = 0; i < n_plus_1; i++)
        ct[i] = spfo_modulus_switch(ct[i], log_chi, log_v, log_modulus);
}

static void rotate_left_u64(uint64_t *p, size_t len, size_t shift)
{
    /* slice::rotate_left: element at index `shift` becomes first */
    uint64_t *tmp = (uint64_t *)malloc(len * sizeof(uint64_t));
    for (size_t i = 0; i < len; i++) tmp[i] = p[(i + shift) % len];
    memcpy(p, tmp, len * sizeof(uint64_t));
    free(tmp);
}

static void rotate_right_u64(uint64_t *p, size_t len, size_t shift)
{
    /* slice::rotate_right: element at index len-shift becomes first */
    uint64_t *tmp = (uint64_t *)malloc(len * sizeof(uint64_t));
    for (size_t i = 0; i < len; i++) tmp[(i + shift) % len] = p[i];
    memcpy(p, tmp, len * sizeof(uint64_t));
    free(tmp);
}

/* sunscreen_tfhe/src/entities/polynomial.rs:171-201 */
void spfo_poly_mul_neg_monomial(uint64_t *p, size_t len, size_t degree)
{
    degree = degree % (2 * len);
    if (degree == 0) return;
    if (degree == len) {
        for (size_t i = 0; i < len; i++) p[i] = (uint64_t)0 - p[i];
        return;
    }
    size_t shift = degree % len;
    rotate_left_u64(p, len, shift);
    size_t lo, hi;
    if (degree < len) { lo = len - shift; hi = len; } else { lo = 0; hi = len - shift; }
    for (size_t i = lo; i < hi; i++) p[i] = (uint64_t)0 - p[i];
}

/* entities/polynomial.rs:208-236 */
void spfo_poly_mul_pos_monomial(uint64_t *p, size_t len, size_t degree)
{
    degree = degree % (2 * len);
    if (degree == 0) return;
    if (degree == len) {
        for (size_t i = 0; i < len; i++) p[i] = (uint64_t)0 - p[i];
        return;
    }
    size_t shift = degree % len;
    rotate_right_u64(p, len, shift);
    size_t lo, hi;
    if (degree < len) { lo = 0; hi = degree; } else { lo = shift; hi = len; }
    for (size_t i = lo; i < hi; i++) p[i] = (uint64_t)0 - p[i];
}

/* sunscreen_tfhe/src/math/radix.rs:157-162 */
uint64_t spfo_radix_round(uint64_t x, uint32_t radix_log, uint32_t count)
{
    uint32_t shift = 64 - radix_log * count;
    uint64_t round_bit = (x >> (shift - 1)) & 1;
    return (x >> shift) + round_bit;
}

/* sunscreen_tfhe/src/math/simd/scalar.rs:52-71 */
uint64_t spfo_radix_next_digit(uint64_t *s, uint32_t radix_log)
{
    uint64_t mask = ((uint64_t)1 << radix_log) - 1;
    uint64_t digit = *s & mask;
    *s = *s >> radix_log;
    uint64_t carry = digit >> (radix_log - 1);
    *s = *s + carry;
    return digit - (carry << radix_log);
}

/* radix.rs:81-113 */
void spfo_decompose_poly(const uint64_t *poly, size_t N, uint32_t radix_log, uint32_t count,
                         uint64_t *digits)
{
    uint64_t *state = (uint64_t *)malloc(N * sizeof(uint64_t));
    for (size_t i = 0; i < N; i++) state[i] = spfo_radix_round(poly[i], radix_log, count);
    for (uint32_t j = 0; j < count; j++)
        for (size_t i = 0; i < N; i++)
            digits[(size_t)j * N + i] = spfo_radix_next_digit(&state[i], radix_log);
    free(state);
}

/* simd/scalar.rs:134-143 */
void spfo_poly_shr_round(uint64_t *y, const uint64_t *x, size_t len, uint32_t n)
{
    for (size_t i = 0; i < len; i++) {
        uint64_t round_bit = (x[i] >> (n - 1)) & 1;
        y[i] = (x[i] >> n) + round_bit;
    }
}

/* ops/polynomial/mod.rs:62-84 */
void spfo_poly_pow_k(uint64_t *p_k, const uint64_t *p, size_t N, size_t k)
{
    for (size_t i = 0; i < N; i++) {
        size_t i_k = (i * k) % N;
        int neg = (((i * k) / N) % 2) != 0;
        p_k[i_k] = neg ? (uint64_t)0 - p[i] : p[i];
    }
}

/* ops/ciphertext/glwe_ciphertext_ops.rs:31-76 */
void spfo_sample_extract(uint64_t *lwe_out, const uint64_t *glwe, size_t h, size_t N, size_t k)
{
    for (size_t i = 0; i < k; i++) {
        const uint64_t *a = glwe + i * N;
        for (size_t j = 0; j <= h; j++) lwe_out[N * i + j] = a[h - j];
        for (size_t j = h + 1; j < N; j++) lwe_out[N * i + j] = (uint64_t)0 - a[h + N - j];
    }
    lwe_out[k * N] = glwe[k * N + h];
}

/* crypto/evaluation.rs:47-50: output = input + l1glwe_one, where l1glwe_one =
 * trivial_glwe(1, l1_params, PlaintextBits(1)) (encryption.rs:359-376): zero mask, body = encode(1) */
void spfo_glwe_not(uint64_t *out, const uint64_t *in, size_t N, size_t k)
{
    for (size_t i = 0; i < (k + 1) * N; i++) out[i] = in[i];
    out[k * N] += spfo_encode(1, 1);
}

/* crypto/evaluation.rs:52-55 -> add_glwe_ciphertexts (glwe_ciphertext_ops.rs:79-99) */
void spfo_glwe_xor(uint64_t *out, const uint64_t *a, const uint64_t *b, size_t N, size_t k)
{
    for (size_t i = 0; i < (k + 1) * N; i++) out[i] = a[i] + b[i];
}

/* crypto/evaluation.rs:57-65 -> rotate_glwe_monomial_negacyclic (blind_rotation.rs:79-98): clone
 * each mask polynomial and the body, then mul_by_monomial_negacyclic(+n) in place */
void spfo_glwe_mul_xn(uint64_t *out, const uint64_t *in, size_t n, size_t N, size_t k)
{
    for (size_t p = 0; p <= k; p++) {
        for (size_t i = 0; i < N; i++) out[p * N + i] = in[p * N + i];
        spfo_poly_mul_pos_monomial(out + p * N, N, n);
    }
}

/* ops/homomorphisms/lwe.rs:9-20 */
void spfo_lwe_rotate(uint64_t *out, const uint64_t *in, size_t n, uint64_t rot)
{
    for (size_t i = 0; i < n; i++) out[i] = in[i];
    out[n] = in[n] + rot;
}

static uint32_t ceil_log2_sz(size_t v)
{
    /* `if v.is_power_of_two() { v.ilog2() } else { v.ilog2() + 1 }` */
    uint32_t l = 0;
    while (((size_t)1 << (l + 1)) <= v) l++;
    return (((size_t)1 << l) == v) ? l : l + 1;
}

/* ops/bootstrapping/programmable_bootstrapping.rs:129-185 */
void spfo_generate_lut(uint64_t *c, size_t n, const uint64_t *maps, size_t v, uint32_t bits)
{
    size_t p = (size_t)1 << bits;
    uint32_t log_v = ceil_log2_sz(v);
    size_t ceil_v = (size_t)1 << log_v;
    size_t stride = n / p;
    uint32_t delta = 64 - bits;
    for (size_t j = 0; j < p; j++) {
        for (size_t k = 0; k < stride; k++) {
            size_t fn_id = k % ceil_v;
            uint64_t p_i = fn_id < v ? maps[fn_id * p + j] : 0;
            c[j * stride + k] = p_i << delta;
        }
    }
    for (size_t i = 0; i < stride / 2; i++) c[i] = (uint64_t)0 - c[i];
    rotate_left_u64(c, n, stride / 2);
}

/* programmable_bootstrapping.rs:70-126 */
void spfo_generate_negacyclic_lut(uint64_t *c, size_t n, const uint64_t *map, uint32_t bits)
{
    size_t p = (size_t)1 << bits;
    size_t stride = 2 * n / p;
    uint32_t delta = 64 - bits;
    for (size_t j = 0; j <= p / 2; j++) {
        uint64_t p_i = map[j] << delta;
        if (j == 0) {
            for (size_t k = 0; k < stride / 2; k++) c[k] = p_i;
        } else if (j == p / 2) {
            for (size_t k = n - stride / 2; k < n; k++) c[k] = p_i;
        } else {
            for (size_t k = stride / 2 + (j - 1) * stride; k < stride / 2 + j * stride; k++)
                c[k] = p_i;
        }
    }
}

/* math/torus.rs:284-300 */
uint64_t spfo_encode(uint64_t val, uint32_t plain_bits) { return val << (64 - plain_bits); }
uint64_t spfo_decode(uint64_t t, uint32_t plain_bits)
{
    uint64_t round_bit = (t >> (64 - plain_bits - 1)) & 1;
    uint64_t mask = ((uint64_t)1 << plain_bits) - 1;
    return ((t >> (64 - plain_bits)) + round_bit) & mask;
}

/* ops/bootstrapping/circuit_bootstrapping.rs:430-482 */
void spfo_fill_cbs_lut(uint64_t *lut, size_t N, size_t k, uint32_t cbs_radix_log,
                       uint32_t cbs_count)
{
    memset(lut, 0, (k + 1) * N * sizeof(uint64_t));
    uint64_t levels[16];
    for (size_t i = 0; i < 16; i++) {
        levels[i] = 0;
        size_t lvl = i + 1;
        if (lvl * cbs_radix_log + 1 < 64) {
            uint32_t plaintext_bits = (uint32_t)(cbs_radix_log * lvl + 1);
            uint64_t minus_one = ((uint64_t)1 << plaintext_bits) - 1;
            levels[i] = spfo_encode(minus_one, plaintext_bits);
        }
    }
    uint32_t log_v = ceil_log2_sz(cbs_count);
    size_t v = (size_t)1 << log_v;
    uint64_t *b = lut + k * N;
    for (size_t i = 0; i < N; i++) {
        size_t fn_id = i % v;
        b[i] = fn_id < cbs_count ? levels[fn_id] : 0;
    }
}

/* ops/keyswitch/lwe_keyswitch.rs:23-62; lev_ciphertext_ops.rs:18-42; lwe_ciphertext_ops.rs:48-66 */
void spfo_keyswitch_lwe(uint64_t *out, const uint64_t *in, const uint64_t *ksk, size_t n_in,
                        size_t n_out, uint32_t radix_log, uint32_t count)
{
    size_t w = n_out + 1;
    uint64_t *sum = (uint64_t *)calloc(w, sizeof(uint64_t));
    for (size_t i = 0; i < n_in; i++) {
        const uint64_t *lev = ksk + i * (size_t)count * w;
        uint64_t state = spfo_radix_round(in[i], radix_log, count);
        /* LEV rows consumed in reverse: first (least significant) digit <-> row count-1 */
        for (uint32_t j = 0; j < count; j++) {
            uint64_t digit = spfo_radix_next_digit(&state, radix_log);
            const uint64_t *row = lev + (size_t)(count - 1 - j) * w;
            for (size_t t = 0; t < w; t++) sum[t] += row[t] * digit;
        }
    }
    /* output = trivial(b) - sum */
    for (size_t t = 0; t < n_out; t++) out[t] = (uint64_t)0 - sum[t];
    out[n_out] = in[n_in] - sum[n_out];
    free(sum);
}

/* ======================================================================== twiddles */

/* e^{+2*pi*i*num/den}: first-octant cosl/sinl in long double, rounded once to double, then
 * mapped by octant symmetry so that conjugate/mirror entries are exact negations/swaps. */
spfo_c64 spfo_root_of_unity(uint64_t num, uint64_t den)
{
    static const long double TWO_PI = 6.283185307179586476925286766559005768L;
    uint64_t j = num % den;
    uint64_t eighth = den / 8;
    uint64_t oct = j / eighth;
    uint64_t r = j % eighth;
    uint64_t rr = (oct & 1) ? (eighth - r) : r;
    long double theta = TWO_PI * (long double)rr / (long double)den;
    double c = (double)cosl(theta), s = (double)sinl(theta);
    if (rr == 0) { c = 1.0; s = 0.0; }
    spfo_c64 o;
    switch (oct) {
    case 0: o.re = c; o.im = s; break;
    case 1: o.re = s; o.im = c; break;
    case 2: o.re = -s; o.im = c; break;
    case 3: o.re = -c; o.im = s; break;
    case 4: o.re = -c; o.im = -s; break;
    case 5: o.re = -s; o.im = -c; break;
    case 6: o.re = s; o.im = -c; break;
    default: o.re = c; o.im = -s; break;
    }
    return o;
}

/* ======================================================================== complex helpers */

static inline spfo_c64 cadd(spfo_c64 a, spfo_c64 b) { return (spfo_c64){a.re + b.re, a.im + b.im}; }
static inline spfo_c64 csub(spfo_c64 a, spfo_c64 b) { return (spfo_c64){a.re - b.re, a.im - b.im}; }

/* num-complex Mul, non-fused (scalar.rs:12-35 use it): */
static inline spfo_c64 cmul_nf(spfo_c64 a, spfo_c64 b)
{
    spfo_c64 o;
    o.re = a.re * b.re - a.im * b.im;
    o.im = a.re * b.im + a.im * b.re;
    return o;
}

/* FFT-internal twiddle multiply (DAG-I and DAG-II): one mul + one fma per component. */
static inline spfo_c64 cmul_tw(spfo_c64 a, spfo_c64 w)
{
    spfo_c64 o;
    double t1 = a.im * w.im;
    o.re = fma(a.re, w.re, -t1);
    double t2 = a.im * w.re;
    o.im = fma(a.re, w.im, t2);
    return o;
}

static inline spfo_c64 cconj(spfo_c64 a) { return (spfo_c64){a.re, -a.im}; }

/* acc + a * w as DAG-II folds it into a butterfly: two fused multiply-adds per component (w already conjugated for the
 * inverse transform by the caller) */
static inline spfo_c64 cfma_tw(spfo_c64 acc, spfo_c64 a, spfo_c64 w)
{
    spfo_c64 o;
    o.re = fma(-a.im, w.im, fma(a.re, w.re, acc.re));
    o.im = fma(a.im, w.re, fma(a.re, w.im, acc.im));
    return o;
}

/* ======================================================================== canonical FFT (DAG-II, r06)
 *
 * The transform is third-party in the reference (rustfft behind sunscreen_tfhe's negacyclic wrapper,
 * math/fft/negacyclic/mod.rs:96-122): its operation order is this repository's to define, and the HIP kernels follow the
 * definition below operation for operation.  DAG-II (r06) is DAG-I (r01-r05) re-associated so that the f64 stream is mostly
 * fused multiply-adds — of DAG-I's 2 112 f64 instructions per wave and blind-rotation step 1 700 were one-flop adds or
 * multiplies:
 *   - the twiddle factors between the radix-8 passes are applied at the INPUT of the next pass's first butterfly stage:
 *       a = w_j v_j;   s = a + w_{j+4} v_{j+4}  (two FMAs per component);   t = 2 a - s  (one FMA per component)
 *     (pass-1 factor W512^{(8a+b) k1} = W64^{a k1} W512^{b k1}: the first part is pass 2's input factor, the second is
 *     common to the eight operands of a pass-2 butterfly, commutes with it and joins pass 2's own factor W64^{b c} as pass 3's
 *     input factor W512^{b (k1 + 8c)});
 *   - the 1/sqrt(2) of the two W8 rotations inside a radix-8 multiplies their SUM / DIFFERENCE in the last stage
 *     (u = b0 +- c q as FMAs) instead of each rotated term.
 * 8-point DFT butterfly, decimation in frequency, three radix-2 stages.  dir=+1: kernel e^{-2 pi i jk/8}; dir=-1: e^{+2 pi i jk/8}.
 * w: the seven input factors of operands 1..7 (already conjugated for dir < 0), or NULL for none.  The tree below IS the
 * definition: the HIP kernels (spf_amd/csrc/spf_device.hpp: radix8 / radix8_in) perform the same operations on the same operands.
 */
static const double SQRT1_2 = 0.70710678118654752440; /* 0x3FE6A09E667F3BCD */

static void radix8(const spfo_c64 v[8], const spfo_c64 *w, spfo_c64 u[8], int dir)
{
    spfo_c64 s[4], t[4];
    if (w) {
        s[0] = cfma_tw(v[0], v[4], w[3]);
        t[0] = (spfo_c64){fma(2.0, v[0].re, -s[0].re), fma(2.0, v[0].im, -s[0].im)};
        for (int j = 1; j < 4; j++) {
            spfo_c64 a = cmul_tw(v[j], w[j - 1]);
            s[j] = cfma_tw(a, v[j + 4], w[j + 3]);
            t[j] = (spfo_c64){fma(2.0, a.re, -s[j].re), fma(2.0, a.im, -s[j].im)};
        }
    } else {
        for (int j = 0; j < 4; j++) { s[j] = cadd(v[j], v[j + 4]); t[j] = csub(v[j], v[j + 4]); }
    }
    /* t_1 W8, t_3 W8^3 without their 1/sqrt(2): qb = (t1 W8 + t3 W8^3) sqrt(2), qe = (t1 W8 - t3 W8^3) sqrt(2) */
    spfo_c64 qb, qe;
    if (dir > 0) {
        double p1 = t[1].re + t[1].im, m1 = t[1].im - t[1].re;   /* (x+iy)(1-i) */
        double p3 = t[3].re + t[3].im, m3 = t[3].im - t[3].re;   /* (x+iy)(-1-i) = (m3, -p3) */
        qb = (spfo_c64){p1 + m3, m1 - p3};
        qe = (spfo_c64){p1 - m3, m1 + p3};
    } else {
        double p1 = t[1].re + t[1].im, m1 = t[1].re - t[1].im;   /* (x+iy)(1+i) = (m1, p1) */
        double p3 = t[3].re + t[3].im, m3 = t[3].re - t[3].im;   /* (x+iy)(-1+i) = (-p3, m3) */
        qb = (spfo_c64){m1 - p3, p1 + m3};
        qe = (spfo_c64){m1 + p3, p1 - m3};
    }
    /* even outputs: 4-point DFT of s */
    spfo_c64 a0 = cadd(s[0], s[2]), a1 = cadd(s[1], s[3]), a2 = csub(s[0], s[2]), d = csub(s[1], s[3]);
    u[0] = cadd(a0, a1);
    u[4] = csub(a0, a1);
    /* odd outputs: 4-point DFT of (t0, t1 W8, t2 (-/+i), t3 W8^3) */
    spfo_c64 b0, b2;
    const double c = SQRT1_2;
    if (dir > 0) {
        u[2] = (spfo_c64){a2.re + d.im, a2.im - d.re};
        u[6] = (spfo_c64){a2.re - d.im, a2.im + d.re};
        b0 = (spfo_c64){t[0].re + t[2].im, t[0].im - t[2].re};
        b2 = (spfo_c64){t[0].re - t[2].im, t[0].im + t[2].re};
    } else {
        u[2] = (spfo_c64){a2.re - d.im, a2.im + d.re};
        u[6] = (spfo_c64){a2.re + d.im, a2.im - d.re};
        b0 = (spfo_c64){t[0].re - t[2].im, t[0].im + t[2].re};
        b2 = (spfo_c64){t[0].re + t[2].im, t[0].im - t[2].re};
    }
    u[1] = (spfo_c64){fma(c, qb.re, b0.re), fma(c, qb.im, b0.im)};
    u[5] = (spfo_c64){fma(-c, qb.re, b0.re), fma(-c, qb.im, b0.im)};
    if (dir > 0) {   /* b2 -/+ i e, e = c qe */
        u[3] = (spfo_c64){fma(c, qe.im, b2.re), fma(-c, qe.re, b2.im)};
        u[7] = (spfo_c64){fma(-c, qe.im, b2.re), fma(c, qe.re, b2.im)};
    } else {
        u[3] = (spfo_c64){fma(-c, qe.im, b2.re), fma(c, qe.re, b2.im)};
        u[7] = (spfo_c64){fma(c, qe.im, b2.re), fma(-c, qe.re, b2.im)};
    }
}

/* twiddle tables of the canonical transform, forward sign (e^{-2 pi i e/M}); inverse uses exact conjugates */
static spfo_c64 W512_tab[512], W64_tab[64], W1024_tab[512], TWIST2048[1024];
static pthread_once_t tab_once = PTHREAD_ONCE_INIT;
static void init_tables(void)
{
    for (int e = 0; e < 512; e++) W512_tab[e] = cconj(spfo_root_of_unity((uint64_t)e, 512));
    for (int e = 0; e < 64; e++) W64_tab[e] = cconj(spfo_root_of_unity((uint64_t)e, 64));
    for (int e = 0; e < 512; e++) W1024_tab[e] = cconj(spfo_root_of_unity((uint64_t)e, 1024));
    /* negacyclic/mod.rs:56-65: twist_j = (cos, sin)(2 pi j / (2N)), N = 2048 */
    for (int j = 0; j < 1024; j++) TWIST2048[j] = spfo_root_of_unity((uint64_t)j, 4096);
}

/* 512-point DFT, 8x8x8, natural order in and out (DAG-II).
 *   n' = 64*n1 + n0,  n0 = 8a + b,   k' = k1 + 8c + 64d
 *   pass 1: radix-8 over n1 -> k1
 *   pass 2: radix-8 over a  -> c,  operand a enters as  W64^{a*k1}  * y[8a+b][k1]
 *   pass 3: radix-8 over b  -> d,  operand b enters as  W512^{b*(k1+8c)} * g[k1][b][c]
 */
static void fft512(const spfo_c64 *x, spfo_c64 *X, int dir)
{
    static __thread spfo_c64 y[64][8], g[8][8][8];
    spfo_c64 v[8], u[8], w[7];
    for (int n0 = 0; n0 < 64; n0++) {
        for (int n1 = 0; n1 < 8; n1++) v[n1] = x[64 * n1 + n0];
        radix8(v, NULL, u, dir);
        for (int k1 = 0; k1 < 8; k1++) y[n0][k1] = u[k1];
    }
    for (int k1 = 0; k1 < 8; k1++)
        for (int b = 0; b < 8; b++) {
            for (int a = 0; a < 8; a++) v[a] = y[8 * a + b][k1];
            for (int a = 1; a < 8; a++) w[a - 1] = dir > 0 ? W64_tab[a * k1] : cconj(W64_tab[a * k1]);
            radix8(v, w, u, dir);
            for (int c = 0; c < 8; c++) g[k1][b][c] = u[c];
        }
    for (int k1 = 0; k1 < 8; k1++)
        for (int c = 0; c < 8; c++) {
            for (int b = 0; b < 8; b++) v[b] = g[k1][b][c];
            for (int b = 1; b < 8; b++) w[b - 1] = dir > 0 ? W512_tab[b * (k1 + 8 * c)] : cconj(W512_tab[b * (k1 + 8 * c)]);
            radix8(v, w, u, dir);
            for (int d = 0; d < 8; d++) X[k1 + 8 * c + 64 * d] = u[d];
        }
}

/* 1024-point DFT of the canonical transform.
 * forward: split input by parity, FFT-512 each, X[k] = E[k] + W1024^k O[k], X[k+512] = E - W O.
 * inverse: E'[k] = X[k] + X[k+512], O'[k] = (X[k] - X[k+512]) * conj(W1024^k), FFT-512^-1 each,
 *          y[2n'] = e0[n'], y[2n'+1] = e1[n'].  Unnormalised both ways. */
void spfo_fft1024(const spfo_c64 *in, spfo_c64 *out, int dir)
{
    pthread_once(&tab_once, init_tables);
    spfo_c64 e0[512], e1[512], E[512], O[512];
    if (dir > 0) {
        for (int n = 0; n < 512; n++) { e0[n] = in[2 * n]; e1[n] = in[2 * n + 1]; }
        fft512(e0, E, +1);
        fft512(e1, O, +1);
        for (int k = 0; k < 512; k++) {
            spfo_c64 t = cmul_tw(O[k], W1024_tab[k]);
            out[k] = cadd(E[k], t);
            out[k + 512] = csub(E[k], t);
        }
    } else {
        for (int k = 0; k < 512; k++) {
            E[k] = cadd(in[k], in[k + 512]);
            spfo_c64 dd = csub(in[k], in[k + 512]);
            O[k] = cmul_tw(dd, cconj(W1024_tab[k]));
        }
        fft512(E, e0, -1);
        fft512(O, e1, -1);
        for (int n = 0; n < 512; n++) { out[2 * n] = e0[n]; out[2 * n + 1] = e1[n]; }
    }
}

/* plain radix-2 DIT for the other sizes (NOT canonical; serves only the reference's small KATs
 * and small-N functional tests).  len is a power of two >= 1. */
static void fft_generic(spfo_c64 *a, size_t len, int dir)
{
    if (len <= 1) return;
    /* bit reversal */
    for (size_t i = 1, j = 0; i < len; i++) {
        size_t bit = len >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) { spfo_c64 t = a[i]; a[i] = a[j]; a[j] = t; }
    }
    for (size_t m = 2; m <= len; m <<= 1) {
        for (size_t k = 0; k < len; k += m)
            for (size_t j = 0; j < m / 2; j++) {
                spfo_c64 w = (m >= 8) ? spfo_root_of_unity(j, m)
                                      : (m == 2 ? (spfo_c64){1.0, 0.0}
                                                : (j == 0 ? (spfo_c64){1.0, 0.0}
                                                          : (spfo_c64){0.0, 1.0}));
                if (dir > 0) w = cconj(w);
                spfo_c64 t = cmul_tw(a[k + j + m / 2], w);
                spfo_c64 u = a[k + j];
                a[k + j] = cadd(u, t);
                a[k + j + m / 2] = csub(u, t);
            }
    }
}

static spfo_c64 twist_of(size_t j, size_t N)
{
    /* negacyclic/mod.rs:56-65: (cos, sin)(2 pi j / (2N)) */
    if (N == 2048) return TWIST2048[j];
    if (2 * N >= 8) return spfo_root_of_unity(j, 2 * N);
    /* N = 2: 2N = 4, j in {0}: */
    return (spfo_c64){1.0, 0.0};
}

/* negacyclic/mod.rs:96-107 + scalar.rs:19-23 */
void spfo_twisted_fft_forward(const double *x, size_t N, spfo_c64 *out)
{
    pthread_once(&tab_once, init_tables);
    size_t h = N / 2;
    spfo_c64 *z = (spfo_c64 *)malloc(h * sizeof(spfo_c64));
    for (size_t j = 0; j < h; j++) z[j] = cmul_nf((spfo_c64){x[j], x[j + h]}, twist_of(j, N));
    if (N == 2048) {
        spfo_fft1024(z, out, +1);
    } else {
        fft_generic(z, h, +1);
        memcpy(out, z, h * sizeof(spfo_c64));
    }
    free(z);
}

/* negacyclic/mod.rs:109-122 + scalar.rs:26-35.  twist_inv is DEFINED here as the exact
 * conjugate of twist (the reference computes twist.powf(-1) through libm polar form,
 * negacyclic/mod.rs:67-71, which is host-libm dependent). */
void spfo_twisted_fft_reverse(const spfo_c64 *in, size_t N, double *out)
{
    pthread_once(&tab_once, init_tables);
    size_t h = N / 2;
    spfo_c64 *y = (spfo_c64 *)malloc(h * sizeof(spfo_c64));
    if (N == 2048) {
        spfo_fft1024(in, y, -1);
    } else {
        memcpy(y, in, h * sizeof(spfo_c64));
        fft_generic(y, h, -1);
    }
    double n_inv = 1.0 / (double)h;
    for (size_t i = 0; i < h; i++) {
        spfo_c64 xs = {y[i].re * n_inv, y[i].im * n_inv};
        spfo_c64 tmp = cmul_nf(xs, cconj(twist_of(i, N)));
        out[i] = round(tmp.re);
        out[i + h] = round(tmp.im);
    }
    free(y);
}

/* entities/polynomial.rs:257-274 */
void spfo_poly_fft(const uint64_t *poly, size_t N, spfo_c64 *out)
{
    double *xf = (double *)malloc(N * sizeof(double));
    for (size_t i = 0; i < N; i++) xf[i] = (double)(int64_t)poly[i];
    spfo_twisted_fft_forward(xf, N, out);
    free(xf);
}

/* scalar.rs:75-119 with log2_q = 64, then FromF64 for u64 (torus.rs:177-192): `x as i64` */
uint64_t spfo_f64_to_torus(double v)
{
    const double q = 18446744073709551616.0;      /* 2^64 */
    const double q_div_2 = 9223372036854775808.0; /* 2^63 */
    double m = fma(-trunc(v / q), q, v);
    if (m >= q_div_2) m -= q;
    else if (m <= -q_div_2) m += q;
    /* Rust `as i64`: saturating, NaN -> 0 */
    int64_t r;
    if (m != m) r = 0;
    else if (m >= q_div_2) r = INT64_MAX;
    else if (m < -q_div_2) r = INT64_MIN;
    else r = (int64_t)m;
    return (uint64_t)r;
}

/* entities/polynomial_fft.rs:82-99 */
void spfo_poly_ifft(const spfo_c64 *in, size_t N, uint64_t *poly)
{
    double *f = (double *)malloc(N * sizeof(double));
    spfo_twisted_fft_reverse(in, N, f);
    for (size_t i = 0; i < N; i++) poly[i] = spfo_f64_to_torus(f[i]);
    free(f);
}

/* complex_mad: c += a * b.  The reference dispatches on the host CPU
 * (math/simd/x86_64/mod.rs:59-75):
 *   mode 1 (default) — AVX-512 hosts: inline asm with four FMAs per element
 *       (math/simd/x86_64/avx512.rs:54-57):
 *         re(c) = fma(re(a), re(b), re(c));  im(c) = fma(re(a), im(b), im(c));
 *         re(c) = fma(-im(a), im(b), re(c)); im(c) = fma(im(a), re(b), im(c));
 *       This is what the reference executes on any AVX-512 server CPU (including the hosts of
 *       the MI355X boxes), so it is the canonical order of this build.
 *   mode 0 — scalar / AVX2 hosts: `*c += a * b` with num-complex Mul, nothing fused
 *       (math/simd/scalar.rs:12-16, x86_64/avx2.rs:10-16). */
static int g_mad_mode = 1;
void spfo_set_mad_mode(int mode) { g_mad_mode = mode ? 1 : 0; }
int spfo_get_mad_mode(void) { return g_mad_mode; }

void spfo_complex_mad(spfo_c64 *c, const spfo_c64 *a, const spfo_c64 *b, size_t len)
{
    if (g_mad_mode) {
        for (size_t i = 0; i < len; i++) {
            double re = fma(a[i].re, b[i].re, c[i].re);
            double im = fma(a[i].re, b[i].im, c[i].im);
            re = fma(-a[i].im, b[i].im, re);
            im = fma(a[i].im, b[i].re, im);
            c[i].re = re;
            c[i].im = im;
        }
    } else {
        for (size_t i = 0; i < len; i++) {
            spfo_c64 p = cmul_nf(a[i], b[i]);
            c[i].re += p.re;
            c[i].im += p.im;
        }
    }
}

void spfo_negacyclic_mul_exact(uint64_t *c, const uint64_t *a, const uint64_t *b, size_t N)
{
    for (size_t i = 0; i < N; i++) c[i] = 0;
    for (size_t i = 0; i < N; i++)
        for (size_t j = 0; j < N; j++) {
            uint64_t prod = a[i] * b[j];
            size_t t = i + j;
            if (t < N) c[t] += prod; else c[t - N] -= prod;
        }
}

/* ======================================================================== ciphertext ops */

/* ops/fft_ops.rs:23-56 -> :67-98 -> :107-124 */
void spfo_glwe_ggsw_mad(spfo_c64 *c_fft, const uint64_t *a_glwe, const spfo_c64 *ggsw_fft, size_t N,
                        size_t k, uint32_t radix_log, uint32_t count)
{
    size_t h = N / 2;
    size_t glwe_fft_len = (k + 1) * h;
    uint64_t *state = (uint64_t *)malloc(N * sizeof(uint64_t));
    uint64_t *digit = (uint64_t *)malloc(N * sizeof(uint64_t));
    spfo_c64 *dfft = (spfo_c64 *)malloc(h * sizeof(spfo_c64));
    for (size_t p = 0; p <= k; p++) {
        const uint64_t *a_i = a_glwe + p * N;             /* a polynomials then b */
        const spfo_c64 *row = ggsw_fft + p * count * glwe_fft_len; /* GLEV row p */
        for (size_t i = 0; i < N; i++) state[i] = spfo_radix_round(a_i[i], radix_log, count);
        /* GLEV entries consumed in reverse (fft_ops.rs:92) */
        for (uint32_t j = 0; j < count; j++) {
            for (size_t i = 0; i < N; i++) digit[i] = spfo_radix_next_digit(&state[i], radix_log);
            spfo_poly_fft(digit, N, dfft);
            const spfo_c64 *b_glwe = row + (size_t)(count - 1 - j) * glwe_fft_len;
            for (size_t q = 0; q <= k; q++)
                spfo_complex_mad(c_fft + q * h, b_glwe + q * h, dfft, h);
        }
    }
    free(state); free(digit); free(dfft);
}

/* ops/fft_ops.rs:149-181 */
void spfo_cmux(uint64_t *c, const uint64_t *d0, const uint64_t *d1, const spfo_c64 *ggsw_fft,
               size_t N, size_t k, uint32_t radix_log, uint32_t count)
{
    size_t len = (k + 1) * N, h = N / 2;
    uint64_t *diff = (uint64_t *)malloc(len * sizeof(uint64_t));
    uint64_t *prod = (uint64_t *)malloc(len * sizeof(uint64_t));
    spfo_c64 *prod_fft = (spfo_c64 *)calloc((k + 1) * h, sizeof(spfo_c64));
    for (size_t i = 0; i < len; i++) diff[i] = d1[i] - d0[i];
    spfo_glwe_ggsw_mad(prod_fft, diff, ggsw_fft, N, k, radix_log, count);
    for (size_t q = 0; q <= k; q++) spfo_poly_ifft(prod_fft + q * h, N, prod + q * N);
    for (size_t i = 0; i < len; i++) c[i] = prod[i] + d0[i];
    free(diff); free(prod); free(prod_fft);
}

/* ops/bootstrapping/programmable_bootstrapping.rs:342-410 */
void spfo_generalized_pbs(uint64_t *out, const uint64_t *lwe_in, const uint64_t *lut_glwe,
                          const spfo_c64 *bsk_fft, size_t n, size_t N, size_t k, uint32_t radix_log,
                          uint32_t count, uint32_t log_chi, uint32_t log_v)
{
    size_t len = (k + 1) * N, h = N / 2;
    size_t ggsw_len = (k + 1) * count * (k + 1) * h;
    uint32_t two_n = 0;
    while (((size_t)1 << two_n) < N) two_n++;
    two_n += 1; /* degree.ilog2() + 1 */
    uint64_t *ct = (uint64_t *)malloc((n + 1) * sizeof(uint64_t));
    memcpy(ct, lwe_in, (n + 1) * sizeof(uint64_t));
    spfo_lwe_modulus_switch(ct, n + 1, log_chi, log_v, two_n);
    /* V_0 * X^{-b} */
    memcpy(out, lut_glwe, len * sizeof(uint64_t));
    for (size_t p = 0; p <= k; p++) spfo_poly_mul_neg_monomial(out + p * N, N, (size_t)ct[n]);
    uint64_t *tmp = (uint64_t *)malloc(len * sizeof(uint64_t));
    uint64_t *rot = (uint64_t *)malloc(len * sizeof(uint64_t));
    for (size_t i = 0; i < n; i++) {
        memcpy(tmp, out, len * sizeof(uint64_t));
        memcpy(rot, out, len * sizeof(uint64_t));
        for (size_t p = 0; p <= k; p++) spfo_poly_mul_pos_monomial(rot + p * N, N, (size_t)ct[i]);
        spfo_cmux(out, tmp, rot, bsk_fft + i * ggsw_len, N, k, radix_log, count);
    }
    free(ct); free(tmp); free(rot);
}

/* programmable_bootstrapping.rs:291-318 */
void spfo_pbs_univariate(uint64_t *lwe_out, const uint64_t *lwe_in, const uint64_t *lut_glwe,
                         const spfo_c64 *bsk_fft, size_t n, size_t N, size_t k, uint32_t radix_log,
                         uint32_t count)
{
    uint64_t *glwe = (uint64_t *)malloc((k + 1) * N * sizeof(uint64_t));
    spfo_generalized_pbs(glwe, lwe_in, lut_glwe, bsk_fft, n, N, k, radix_log, count, 0, 0);
    spfo_sample_extract(lwe_out, glwe, 0, N, k);
    free(glwe);
}

/* circuit_bootstrapping.rs:387-427 */
void spfo_cbs_pbs(uint64_t *glwe_out, const uint64_t *lwe_in, const spfo_c64 *bsk_fft, size_t n,
                  size_t N, size_t k, uint32_t pbs_radix_log, uint32_t pbs_count,
                  uint32_t cbs_radix_log, uint32_t cbs_count)
{
    uint64_t *lut = (uint64_t *)malloc((k + 1) * N * sizeof(uint64_t));
    uint64_t *rot = (uint64_t *)malloc((n + 1) * sizeof(uint64_t));
    spfo_lwe_rotate(rot, lwe_in, n, spfo_encode(1, 2));
    uint32_t log_v = ceil_log2_sz(cbs_count);
    spfo_fill_cbs_lut(lut, N, k, cbs_radix_log, cbs_count);
    spfo_generalized_pbs(glwe_out, rot, lut, bsk_fft, n, N, k, pbs_radix_log, pbs_count, 0, log_v);
    free(lut); free(rot);
}

/* ======================================================================== keygen subset */

static uint64_t splitmix64(uint64_t *x)
{
    uint64_t z = (*x += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
void spfo_rng_seed(spfo_rng *r, uint64_t seed)
{
    for (int i = 0; i < 4; i++) r->s[i] = splitmix64(&seed);
}
static inline uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
uint64_t spfo_rng_next(spfo_rng *r)
{
    uint64_t *s = r->s;
    uint64_t result = rotl64(s[1] * 5, 7) * 9;
    uint64_t t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
    s[2] ^= t;
    s[3] = rotl64(s[3], 45);
    return result;
}

/* rand.rs:20-31 (distribution only; the reference's sampler is rand_distr::Normal over an
 * unseeded thread_rng).  Box-Muller on two 53-bit uniforms. */
uint64_t spfo_normal_torus(spfo_rng *r, double std)
{
    double u1 = ((double)(spfo_rng_next(r) >> 11) + 0.5) * (1.0 / 9007199254740992.0);
    double u2 = ((double)(spfo_rng_next(r) >> 11) + 0.5) * (1.0 / 9007199254740992.0);
    double g = sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
    double e0 = g * std;
    double q = 18446744073709551616.0;
    double e = round(e0 * q);
    int64_t ei;
    if (e >= 9223372036854775808.0) ei = INT64_MAX;
    else if (e < -9223372036854775808.0) ei = INT64_MIN;
    else ei = (int64_t)e;
    return (uint64_t)ei;
}

void spfo_gen_binary_key(spfo_rng *r, uint64_t *key, size_t len)
{
    for (size_t i = 0; i < len; i++) key[i] = spfo_rng_next(r) % 2;
}

/* ops/encryption/lwe_encryption.rs:36-59 */
void spfo_encrypt_lwe(spfo_rng *r, uint64_t *ct, const uint64_t *sk, size_t n, uint64_t msg,
                      double std)
{
    uint64_t b = 0;
    for (size_t i = 0; i < n; i++) {
        ct[i] = spfo_rng_next(r);
        b += ct[i] * sk[i];
    }
    uint64_t e = spfo_normal_torus(r, std);
    ct[n] = b + msg + e;
}

uint64_t spfo_decrypt_lwe_raw(const uint64_t *ct, const uint64_t *sk, size_t n)
{
    uint64_t acc = 0;
    for (size_t i = 0; i < n; i++) acc += ct[i] * sk[i];
    return ct[n] - acc;
}

/* c += a (*) s  negacyclic, s binary (polynomial_external_mad with a binary key) */
static void poly_mad_binary(uint64_t *c, const uint64_t *a, const uint64_t *s, size_t N)
{
    for (size_t j = 0; j < N; j++) {
        if (!s[j]) continue;
        /* a * X^j */
        for (size_t i = 0; i < N - j; i++) c[i + j] += a[i];
        for (size_t i = N - j; i < N; i++) c[i + j - N] -= a[i];
    }
}

/* ops/encryption/glwe_encryption.rs:22-61 */
void spfo_encrypt_glwe(spfo_rng *r, uint64_t *ct, const uint64_t *sk, const uint64_t *msg, size_t N,
                       size_t k, double std)
{
    uint64_t *b = ct + k * N;
    for (size_t i = 0; i < N; i++) b[i] = 0;
    for (size_t p = 0; p < k; p++) {
        uint64_t *a = ct + p * N;
        for (size_t i = 0; i < N; i++) a[i] = spfo_rng_next(r);
        poly_mad_binary(b, a, sk + p * N, N);
    }
    for (size_t i = 0; i < N; i++) b[i] += msg[i];
    if (std == 0.0) return;
    for (size_t i = 0; i < N; i++) b[i] += spfo_normal_torus(r, std);
}

void spfo_decrypt_glwe_raw(uint64_t *m, const uint64_t *ct, const uint64_t *sk, size_t N, size_t k)
{
    uint64_t *acc = (uint64_t *)calloc(N, sizeof(uint64_t));
    for (size_t p = 0; p < k; p++) poly_mad_binary(acc, ct + p * N, sk + p * N, N);
    for (size_t i = 0; i < N; i++) m[i] = ct[k * N + i] - acc[i];
    free(acc);
}

/* ggsw_encryption.rs:16-72 + glev_encryption.rs:23-77 */
void spfo_encrypt_ggsw_scalar(spfo_rng *r, uint64_t *ggsw, const uint64_t *glwe_sk, uint64_t bit,
                              size_t N, size_t k, uint32_t radix_log, uint32_t count, double std)
{
    size_t glwe_len = (k + 1) * N;
    uint64_t *m = (uint64_t *)malloc(N * sizeof(uint64_t));
    uint64_t *scaled = (uint64_t *)malloc(N * sizeof(uint64_t));
    for (size_t row = 0; row <= k; row++) {
        if (row < k) {
            /* -(msg * s_row), msg = constant polynomial `bit` */
            for (size_t i = 0; i < N; i++) m[i] = (uint64_t)0 - (bit * glwe_sk[row * N + i]);
        } else {
            for (size_t i = 0; i < N; i++) m[i] = 0;
            m[0] = bit;
        }
        for (uint32_t j = 0; j < count; j++) {
            uint64_t factor = (uint64_t)1 << (64 - radix_log * (j + 1));
            for (size_t i = 0; i < N; i++) scaled[i] = m[i] * factor;
            spfo_encrypt_glwe(r, ggsw + (row * count + j) * glwe_len, glwe_sk, scaled, N, k, std);
        }
    }
    free(m); free(scaled);
}

void spfo_ggsw_fft(spfo_c64 *out, const uint64_t *ggsw, size_t N, size_t k, uint32_t count)
{
    size_t polys = (k + 1) * count * (k + 1);
    for (size_t p = 0; p < polys; p++) spfo_poly_fft(ggsw + p * N, N, out + p * (N / 2));
}

/* programmable_bootstrapping.rs:34-58 */
void spfo_gen_bsk_fft(spfo_rng *r, spfo_c64 *bsk_fft, const uint64_t *lwe_sk, size_t n,
                      const uint64_t *glwe_sk, size_t N, size_t k, uint32_t radix_log,
                      uint32_t count, double std)
{
    size_t polys = (k + 1) * count * (k + 1);
    uint64_t *ggsw = (uint64_t *)malloc(polys * N * sizeof(uint64_t));
    for (size_t i = 0; i < n; i++) {
        spfo_encrypt_ggsw_scalar(r, ggsw, glwe_sk, lwe_sk[i], N, k, radix_log, count, std);
        spfo_ggsw_fft(bsk_fft + i * polys * (N / 2), ggsw, N, k, count);
    }
    free(ggsw);
}

/* ops/keyswitch/lwe_keyswitch_key.rs:16-50 */
void spfo_gen_ksk(spfo_rng *r, uint64_t *ksk, const uint64_t *sk_in, size_t n_in,
                  const uint64_t *sk_out, size_t n_out, uint32_t radix_log, uint32_t count,
                  double std)
{
    size_t w = n_out + 1;
    for (size_t i = 0; i < n_in; i++)
        for (uint32_t j = 0; j < count; j++) {
            uint64_t factor = (uint64_t)1 << (64 - radix_log * (j + 1));
            uint64_t msg = factor * sk_in[i];
            spfo_encrypt_lwe(r, ksk + (i * count + j) * w, sk_out, n_out, msg, std);
        }
}

/* ======================================================================== circuit-bootstrap tail
 * (SURVEY.md §8 f2): homomorphic trace over FFT-domain GLWE keyswitches, then the scheme switch
 * that turns the GLEV into a GGSW in the FFT domain. */

/* ops/fft_ops.rs:67-98 for one polynomial: c_fft += <decomp(poly), glev>, rows in reverse */
static void glev_mad(spfo_c64 *c_fft, const uint64_t *poly, const spfo_c64 *glev_fft, size_t N,
                     size_t k, uint32_t radix_log, uint32_t count)
{
    size_t h = N / 2, glwe_fft_len = (k + 1) * h;
    uint64_t *state = (uint64_t *)malloc(N * sizeof(uint64_t));
    uint64_t *digit = (uint64_t *)malloc(N * sizeof(uint64_t));
    spfo_c64 *dfft = (spfo_c64 *)malloc(h * sizeof(spfo_c64));
    for (size_t i = 0; i < N; i++) state[i] = spfo_radix_round(poly[i], radix_log, count);
    for (uint32_t j = 0; j < count; j++) {
        for (size_t i = 0; i < N; i++) digit[i] = spfo_radix_next_digit(&state[i], radix_log);
        spfo_poly_fft(digit, N, dfft);
        const spfo_c64 *b_glwe = glev_fft + (size_t)(count - 1 - j) * glwe_fft_len;
        for (size_t q = 0; q <= k; q++) spfo_complex_mad(c_fft + q * h, b_glwe + q * h, dfft, h);
    }
    free(state); free(digit); free(dfft);
}

/* ops/fft_ops.rs:457-495.  ksk layout [row<k][level<count][poly<k+1][N/2] */
void spfo_keyswitch_glwe_to_glwe(uint64_t *out, const uint64_t *in, const spfo_c64 *ksk_fft, size_t N,
                                 size_t k, uint32_t radix_log, uint32_t count)
{
    size_t h = N / 2, glev_len = (size_t)count * (k + 1) * h;
    spfo_c64 *sum = (spfo_c64 *)calloc((k + 1) * h, sizeof(spfo_c64));
    uint64_t *s = (uint64_t *)malloc((k + 1) * N * sizeof(uint64_t));
    for (size_t i = 0; i < k; i++) glev_mad(sum, in + i * N, ksk_fft + i * glev_len, N, k, radix_log, count);
    for (size_t q = 0; q <= k; q++) spfo_poly_ifft(sum + q * h, N, s + q * N);
    /* output = trivial_encrypt(b) - sum */
    for (size_t i = 0; i < k * N; i++) out[i] = (uint64_t)0 - s[i];
    for (size_t i = 0; i < N; i++) out[k * N + i] = in[k * N + i] - s[k * N + i];
    free(sum); free(s);
}

/* ops/automorphisms/mod.rs:53-85.  ak layout [i<log2 N][glwe ksk] */
void spfo_trace(uint64_t *out, const uint64_t *x, const spfo_c64 *ak_fft, size_t N, size_t k,
                uint32_t radix_log, uint32_t count)
{
    size_t len = (k + 1) * N, ksk_len = k * (size_t)count * (k + 1) * (N / 2);
    uint64_t *glwe_k = (uint64_t *)malloc(len * sizeof(uint64_t));
    uint64_t *ks = (uint64_t *)malloc(len * sizeof(uint64_t));
    uint32_t logn = 0;
    while (((size_t)1 << logn) < N) logn++;
    memcpy(out, x, len * sizeof(uint64_t));
    for (uint32_t i = 1; i <= logn; i++) {
        size_t kk = N / ((size_t)1 << (i - 1)) + 1;
        for (size_t p = 0; p <= k; p++) spfo_poly_pow_k(glwe_k + p * N, out + p * N, N, kk);
        spfo_keyswitch_glwe_to_glwe(ks, glwe_k, ak_fft + (size_t)(i - 1) * ksk_len, N, k, radix_log, count);
        for (size_t t = 0; t < len; t++) out[t] += ks[t];
    }
    free(glwe_k); free(ks);
}

/* ops/bootstrapping/circuit_bootstrapping.rs:260-298: glev is cbs_count GLWEs */
void spfo_mod_switch_trace_and_rotate(uint64_t *glev, const uint64_t *lo_noise_glwe, const spfo_c64 *ak_fft,
                                      size_t N, size_t k, uint32_t tr_radix_log, uint32_t tr_count,
                                      uint32_t cbs_radix_log, uint32_t cbs_count)
{
    size_t len = (k + 1) * N;
    uint32_t shift_amount = 0;
    while (((size_t)1 << shift_amount) < N) shift_amount++;
    uint64_t *rotated = (uint64_t *)malloc(len * sizeof(uint64_t));
    uint64_t *permuted = (uint64_t *)malloc(len * sizeof(uint64_t));
    uint64_t *shifted = (uint64_t *)malloc(len * sizeof(uint64_t));
    memcpy(rotated, lo_noise_glwe, len * sizeof(uint64_t));
    for (uint32_t i = 0; i < cbs_count; i++) {
        uint32_t plaintext_bits = cbs_radix_log * (i + 1) + 1;
        /* undo the rotation applied during the functional bootstrap, coefficient i only */
        rotated[k * N + i] += spfo_encode(1, plaintext_bits);
        memcpy(permuted, rotated, len * sizeof(uint64_t));
        for (size_t p = 0; p <= k; p++) spfo_poly_mul_neg_monomial(permuted + p * N, N, i);
        /* glwe_mod_switch_and_expand_pow_2 (ops/ciphertext/glwe_ciphertext_ops.rs:268-281) */
        spfo_poly_shr_round(shifted, permuted, len, shift_amount);
        spfo_trace(glev + (size_t)i * len, shifted, ak_fft, N, k, tr_radix_log, tr_count);
    }
    free(rotated); free(permuted); free(shifted);
}

static size_t tri_index(size_t i, size_t j, size_t n)
{
    /* entities/scheme_switch_key.rs: get_linear_index */
    size_t row = i <= j ? i : j, col = i <= j ? j : i;
    return (n * (n + 1) / 2) - (n - row) * ((n - row) + 1) / 2 + col - row;
}

/* ops/fft_ops.rs:225-279, 403-442.  out layout [row<k+1][level<ggsw_count][poly<k+1][N/2];
 * ssk layout [pair][level<ss_count][poly<k+1][N/2].  The reference accumulates into rows it
 * assumes zeroed (fresh allocate_ggsw_l1); the restatement clears the output first. */
void spfo_scheme_switch_fft(spfo_c64 *out, const uint64_t *glev, const spfo_c64 *ssk_fft, size_t N,
                            size_t k, uint32_t ggsw_count, uint32_t ss_radix_log, uint32_t ss_count)
{
    size_t h = N / 2, glwe_fft_len = (k + 1) * h, glwe_len = (k + 1) * N;
    size_t ss_glev_len = (size_t)ss_count * glwe_fft_len;
    memset(out, 0, (k + 1) * (size_t)ggsw_count * glwe_fft_len * sizeof(spfo_c64));
    for (size_t j = 0; j <= k; j++)
        for (uint32_t i = 0; i < ggsw_count; i++) {
            spfo_c64 *y = out + (j * ggsw_count + i) * glwe_fft_len;
            const uint64_t *x = glev + (size_t)i * glwe_len;
            if (j == k) {
                for (size_t p = 0; p <= k; p++) spfo_poly_fft(x + p * N, N, y + p * h);
                continue;
            }
            spfo_poly_fft(x + k * N, N, y + j * h); /* y.a[j] = FFT(x.b) */
            for (size_t r = 0; r < k; r++)
                glev_mad(y, x + r * N, ssk_fft + tri_index(j, r, k) * ss_glev_len, N, k, ss_radix_log, ss_count);
        }
}

/* ops/bootstrapping/circuit_bootstrapping.rs:342-385 */
void spfo_circuit_bootstrap(spfo_c64 *ggsw_out, const uint64_t *lwe_in, const spfo_c64 *bsk_fft,
                            const spfo_c64 *ak_fft, const spfo_c64 *ssk_fft, size_t n, size_t N, size_t k,
                            uint32_t pbs_radix_log, uint32_t pbs_count, uint32_t tr_radix_log,
                            uint32_t tr_count, uint32_t ss_radix_log, uint32_t ss_count,
                            uint32_t cbs_radix_log, uint32_t cbs_count)
{
    size_t len = (k + 1) * N;
    uint64_t *glwe = (uint64_t *)malloc(len * sizeof(uint64_t));
    uint64_t *glev = (uint64_t *)malloc((size_t)cbs_count * len * sizeof(uint64_t));
    spfo_cbs_pbs(glwe, lwe_in, bsk_fft, n, N, k, pbs_radix_log, pbs_count, cbs_radix_log, cbs_count);
    spfo_mod_switch_trace_and_rotate(glev, glwe, ak_fft, N, k, tr_radix_log, tr_count, cbs_radix_log, cbs_count);
    spfo_scheme_switch_fft(ggsw_out, glev, ssk_fft, N, k, cbs_count, ss_radix_log, ss_count);
    free(glwe); free(glev);
}

/* ops/keyswitch/glwe_keyswitch_key.rs (encrypt_keyswitch_key_generic) + FFT of every polynomial:
 * row i, level j encrypts original_sk_i * 2^(64 - logB (j+1)) under the new key */
void spfo_gen_glwe_ksk_fft(spfo_rng *r, spfo_c64 *out, const uint64_t *sk_orig, const uint64_t *sk_new,
                           size_t N, size_t k, uint32_t radix_log, uint32_t count, double std)
{
    size_t h = N / 2, len = (k + 1) * N;
    uint64_t *ct = (uint64_t *)malloc(len * sizeof(uint64_t));
    uint64_t *msg = (uint64_t *)malloc(N * sizeof(uint64_t));
    for (size_t i = 0; i < k; i++)
        for (uint32_t j = 0; j < count; j++) {
            uint64_t factor = (uint64_t)1 << (64 - radix_log * (j + 1));
            for (size_t t = 0; t < N; t++) msg[t] = sk_orig[i * N + t] * factor;
            spfo_encrypt_glwe(r, ct, sk_new, msg, N, k, std);
            for (size_t p = 0; p <= k; p++)
                spfo_poly_fft(ct + p * N, N, out + ((i * count + j) * (k + 1) + p) * h);
        }
    free(ct); free(msg);
}

/* ops/automorphisms/mod.rs:18-46 */
void spfo_gen_auto_key_fft(spfo_rng *r, spfo_c64 *ak_fft, const uint64_t *glwe_sk, size_t N, size_t k,
                           uint32_t radix_log, uint32_t count, double std)
{
    size_t ksk_len = k * (size_t)count * (k + 1) * (N / 2);
    uint64_t *sk_k = (uint64_t *)malloc(k * N * sizeof(uint64_t));
    uint32_t logn = 0;
    while (((size_t)1 << logn) < N) logn++;
    for (uint32_t i = 1; i <= logn; i++) {
        size_t kk = N / ((size_t)1 << (i - 1)) + 1;
        for (size_t p = 0; p < k; p++) spfo_poly_pow_k(sk_k + p * N, glwe_sk + p * N, N, kk);
        spfo_gen_glwe_ksk_fft(r, ak_fft + (size_t)(i - 1) * ksk_len, sk_k, glwe_sk, N, k, radix_log, count, std);
    }
    free(sk_k);
}

/* ops/bootstrapping/scheme_switch.rs:22-70: GLEV(s_i * s_j) for the upper-triangular pairs.  The
 * reference multiplies through its FFT; the product of binary polynomials is small, so that is
 * exact and equals the integer negacyclic product used here. */
void spfo_gen_ssk_fft(spfo_rng *r, spfo_c64 *ssk_fft, const uint64_t *glwe_sk, size_t N, size_t k,
                      uint32_t radix_log, uint32_t count, double std)
{
    size_t h = N / 2, len = (k + 1) * N;
    uint64_t *sij = (uint64_t *)malloc(N * sizeof(uint64_t));
    uint64_t *msg = (uint64_t *)malloc(N * sizeof(uint64_t));
    uint64_t *ct = (uint64_t *)malloc(len * sizeof(uint64_t));
    size_t idx = 0;
    for (size_t i = 0; i < k; i++)
        for (size_t j = i; j < k; j++, idx++) {
            spfo_negacyclic_mul_exact(sij, glwe_sk + i * N, glwe_sk + j * N, N);
            for (uint32_t t = 0; t < count; t++) {
                uint64_t factor = (uint64_t)1 << (64 - radix_log * (t + 1));
                for (size_t c = 0; c < N; c++) msg[c] = sij[c] * factor;
                spfo_encrypt_glwe(r, ct, glwe_sk, msg, N, k, std);
                for (size_t p = 0; p <= k; p++)
                    spfo_poly_fft(ct + p * N, N, ssk_fft + ((idx * count + t) * (k + 1) + p) * h);
            }
        }
    free(sij); free(msg); free(ct);
}

/* ======================================================================== cpu_baseline driver */

typedef struct {
    const uint64_t *lwe_in; uint64_t *glwe_out; const spfo_c64 *bsk;
    size_t begin, end, n, N, k; uint32_t prl, pc, crl, cc;
} bench_job;

static void *bench_worker(void *arg)
{
    bench_job *j = (bench_job *)arg;
    for (size_t i = j->begin; i < j->end; i++)
        spfo_cbs_pbs(j->glwe_out + i * (j->k + 1) * j->N, j->lwe_in + i * (j->n + 1), j->bsk, j->n,
                     j->N, j->k, j->prl, j->pc, j->crl, j->cc);
    return NULL;
}

double spfo_bench_cbs_pbs(const uint64_t *lwe_in, size_t count, const spfo_c64 *bsk_fft, size_t n,
                          size_t N, size_t k, uint32_t prl, uint32_t pc, uint32_t crl, uint32_t cc,
                          int threads, uint64_t *glwe_out)
{
    pthread_once(&tab_once, init_tables);
    if (threads < 1) threads = 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
    bench_job *jobs = (bench_job *)malloc(sizeof(bench_job) * (size_t)threads);
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int t = 0; t < threads; t++) {
        jobs[t] = (bench_job){lwe_in, glwe_out, bsk_fft, count * (size_t)t / (size_t)threads,
                              count * (size_t)(t + 1) / (size_t)threads, n, N, k, prl, pc, crl, cc};
        pthread_create(&th[t], NULL, bench_worker, &jobs[t]);
    }
    for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    free(th); free(jobs);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

/* the same driver for the plain PBS (generalized_programmable_bootstrap with any (log_chi, log_v) and a shared
 * or per-ciphertext LUT, programmable_bootstrapping.rs:342-410; `extract` != 0 appends sample_extract(.,0) =
 * programmable_bootstrap_univariate, :291-318): `count` independent bootstraps on `threads` pthreads. */
typedef struct {
    const uint64_t *lwe_in, *lut; uint64_t *out; const spfo_c64 *bsk;
    size_t begin, end, n, N, k, lut_stride, out_stride; uint32_t prl, pc, log_chi, log_v; int extract;
} gpbs_job;

static void *gpbs_worker(void *arg)
{
    gpbs_job *j = (gpbs_job *)arg;
    for (size_t i = j->begin; i < j->end; i++) {
        const uint64_t *lwe = j->lwe_in + i * (j->n + 1), *lut = j->lut + i * j->lut_stride;
        uint64_t *o = j->out + i * j->out_stride;
        if (j->extract) spfo_pbs_univariate(o, lwe, lut, j->bsk, j->n, j->N, j->k, j->prl, j->pc);
        else spfo_generalized_pbs(o, lwe, lut, j->bsk, j->n, j->N, j->k, j->prl, j->pc, j->log_chi, j->log_v);
    }
    return NULL;
}

double spfo_bench_generalized_pbs(const uint64_t *lwe_in, size_t count, const uint64_t *lut, size_t lut_stride,
                                  const spfo_c64 *bsk_fft, size_t n, size_t N, size_t k, uint32_t prl, uint32_t pc,
                                  uint32_t log_chi, uint32_t log_v, int extract, int threads, uint64_t *out)
{
    pthread_once(&tab_once, init_tables);
    if (threads < 1) threads = 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
    gpbs_job *jobs = (gpbs_job *)malloc(sizeof(gpbs_job) * (size_t)threads);
    const size_t out_stride = extract ? k * N + 1 : (k + 1) * N;
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int t = 0; t < threads; t++) {
        jobs[t] = (gpbs_job){lwe_in, lut, out, bsk_fft, count * (size_t)t / (size_t)threads,
                             count * (size_t)(t + 1) / (size_t)threads, n, N, k, lut_stride, out_stride,
                             prl, pc, log_chi, log_v, extract};
        pthread_create(&th[t], NULL, gpbs_worker, &jobs[t]);
    }
    for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    free(th); free(jobs);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
