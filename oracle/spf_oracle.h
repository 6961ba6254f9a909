/*
 * spf_oracle.h — CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the programmable-bootstrap / LWE-keyswitch path of
 * Sunscreen-tech/spf (sunscreen_tfhe).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library, and only as the
 * checker.  The shipped HIP path (spf_amd/csrc) never links or calls it.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - integer stages (modulus switch, radix rounding/decomposition, monomial
 *     rotation, sample extract, keyswitch MAD, LUT generation) are pinned by the
 *     reference's own known-answer tests, transcribed as data in
 *     tests/golden/reference_kats.json.
 *   - floating-point stages: the negacyclic transform is pinned by the
 *     reference's exact KAT `can_negacyclic_conv` and its 1e-12 round-trip test;
 *     the *bit pattern* of FFT outputs at cryptographic magnitudes is
 *     PARITY UNPINNED: the reference delegates the complex FFT to the
 *     un-vendored crate rustfft 6.3.0 (Cargo.lock:2337), whose butterfly
 *     schedule is chosen at run time per host ISA, and no Rust toolchain
 *     exists here to run it.  This file therefore *defines* one canonical
 *     operation order ("DAG-I", below) that the HIP kernels reproduce
 *     bit-for-bit.
 *
 * All citations are relative to /root/reference/.
 */
#ifndef SPF_ORACLE_H
#define SPF_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { double re, im; } spfo_c64;

/* ---------------------------------------------------------------- integer stages */

/* sunscreen_tfhe/src/ops/ciphertext/lwe_ciphertext_ops.rs:130-142 (modulus_switch) */
uint64_t spfo_modulus_switch(uint64_t x, uint32_t log_chi, uint32_t log_v, uint32_t log_modulus);
/* lwe_ciphertext_ops.rs:97-128: applied to the n mask elements and the body. */
void spfo_lwe_modulus_switch(uint64_t *ct, size_t n_plus_1, uint32_t log_chi, uint32_t log_v,
                             uint32_t log_modulus);

/* sunscreen_tfhe/src/entities/polynomial.rs:171-201 / :208-236, in place. */
void spfo_poly_mul_neg_monomial(uint64_t *p, size_t N, size_t degree);
void spfo_poly_mul_pos_monomial(uint64_t *p, size_t N, size_t degree);

/* sunscreen_tfhe/src/math/radix.rs:157-162 (round) */
uint64_t spfo_radix_round(uint64_t x, uint32_t radix_log, uint32_t count);
/* math/simd/scalar.rs:52-71 (vector_next_decomp, one element): updates *state, returns digit */
uint64_t spfo_radix_next_digit(uint64_t *state, uint32_t radix_log);
/* radix.rs:81-113: all `count` digit polynomials of a polynomial, least significant first.
 * digits is count*N u64 (two's-complement small ints). */
void spfo_decompose_poly(const uint64_t *poly, size_t N, uint32_t radix_log, uint32_t count,
                         uint64_t *digits);

/* ops/polynomial/mod.rs:86-96 via simd/scalar.rs:134-143 (vector_shr_round) */
void spfo_poly_shr_round(uint64_t *y, const uint64_t *x, size_t len, uint32_t n);
/* ops/polynomial/mod.rs:62-84 (polynomial_pow_k): p_k[i*k mod N] = ±p[i] */
void spfo_poly_pow_k(uint64_t *p_k, const uint64_t *p, size_t N, size_t k);

/* ops/ciphertext/glwe_ciphertext_ops.rs:31-76 */
void spfo_sample_extract(uint64_t *lwe_out /*k*N+1*/, const uint64_t *glwe /*(k+1)*N*/, size_t h,
                         size_t N, size_t k);

/* parasol_runtime/src/crypto/evaluation.rs:47-66, the linear KeylessEvaluation operations on one
 * L1 GLWE ciphertext ((k+1)*N words):
 *   not    : out = in + trivial_one (crypto/encryption.rs:359-364: the polynomial 1 at one
 *            plaintext bit (:132), zero mask)
 *   xor    : out = a + b (ops/ciphertext/glwe_ciphertext_ops.rs:79-99)
 *   mul_xn : every polynomial * X^n (ops/bootstrapping/blind_rotation.rs:79-98,126-135) */
void spfo_glwe_not(uint64_t *out, const uint64_t *in, size_t N, size_t k);
void spfo_glwe_xor(uint64_t *out, const uint64_t *a, const uint64_t *b, size_t N, size_t k);
void spfo_glwe_mul_xn(uint64_t *out, const uint64_t *in, size_t n, size_t N, size_t k);

/* ops/homomorphisms/lwe.rs:9-20 : out = in with body += rot */
void spfo_lwe_rotate(uint64_t *out, const uint64_t *in, size_t n, uint64_t rot);

/* ops/bootstrapping/programmable_bootstrapping.rs:129-185 (generate_lut), maps given as a
 * table maps[f*p + x] for f < n_maps, x < p = 2^plaintext_bits. out is N u64. */
void spfo_generate_lut(uint64_t *out, size_t N, const uint64_t *maps, size_t n_maps,
                       uint32_t plaintext_bits);
/* programmable_bootstrapping.rs:70-126 (generate_negacyclic_lut), map given as table[p] */
void spfo_generate_negacyclic_lut(uint64_t *out, size_t N, const uint64_t *map,
                                  uint32_t plaintext_bits);
/* ops/bootstrapping/circuit_bootstrapping.rs:430-482: full GLWE (mask zero, body = levels). */
void spfo_fill_cbs_lut(uint64_t *lut_glwe /*(k+1)*N*/, size_t N, size_t k, uint32_t cbs_radix_log,
                       uint32_t cbs_count);

/* ops/keyswitch/lwe_keyswitch.rs:23-62 with lev_ciphertext_ops.rs:18-42 and
 * lwe_ciphertext_ops.rs:48-66.  ksk layout [n_in][count][n_out+1]. */
void spfo_keyswitch_lwe(uint64_t *out /*n_out+1*/, const uint64_t *in /*n_in+1*/,
                        const uint64_t *ksk, size_t n_in, size_t n_out, uint32_t radix_log,
                        uint32_t count);

/* ---------------------------------------------------------------- float stages */

/* math/fft/negacyclic/mod.rs:96-107 (TwistedFft::forward): x is N doubles, out N/2 bins in
 * natural DFT order.  N == 2048 uses the canonical DAG-I; other powers of two (4..4096) use a
 * plain radix-2 schedule that only serves the reference's small-size KATs. */
void spfo_twisted_fft_forward(const double *x, size_t N, spfo_c64 *out);
/* negacyclic/mod.rs:109-122 (TwistedFft::reverse) incl. complex_untwist (scalar.rs:26-35):
 * out is N doubles, already round()ed. */
void spfo_twisted_fft_reverse(const spfo_c64 *in, size_t N, double *out);
/* entities/polynomial.rs:257-274 (PolynomialRef::fft): u64 -> i64 -> f64 -> forward */
void spfo_poly_fft(const uint64_t *poly, size_t N, spfo_c64 *out);
/* entities/polynomial_fft.rs:82-99 (ifft) incl. vector_mod_pow2_q_f64 (scalar.rs:75-119) */
void spfo_poly_ifft(const spfo_c64 *in, size_t N, uint64_t *poly);
/* scalar.rs:75-119 on one value with log2_q = 64, then `as i64` (saturating) -> u64 */
uint64_t spfo_f64_to_torus(double v);
/* complex_mad, c += a*b.  mode 1 (default): the reference's AVX-512 path, four FMAs
 * (simd/x86_64/avx512.rs:54-57); mode 0: its scalar/AVX2 path, non-fused (simd/scalar.rs:12-16). */
void spfo_set_mad_mode(int mode);
int spfo_get_mad_mode(void);
void spfo_complex_mad(spfo_c64 *c, const spfo_c64 *a, const spfo_c64 *b, size_t len);

/* the canonical complex FFT-1024 alone (for table / DAG tests); dir = +1 forward, -1 inverse */
void spfo_fft1024(const spfo_c64 *in, spfo_c64 *out, int dir);
/* table accessors so tests can pin the twiddle definition: e^{+2*pi*i*num/den}, den a power of
 * two between 8 and 8192 */
spfo_c64 spfo_root_of_unity(uint64_t num, uint64_t den);

/* exact integer negacyclic product c = a (*) b mod (X^N + 1, 2^64); second opinion for FFT */
void spfo_negacyclic_mul_exact(uint64_t *c, const uint64_t *a, const uint64_t *b, size_t N);

/* ---------------------------------------------------------------- ciphertext ops */

/* ops/fft_ops.rs:23-56 (glwe_ggsw_mad): c_fft += a [*] ggsw.  ggsw layout
 * [row<k+1][level<l][poly<k+1][bin<N/2]. */
void spfo_glwe_ggsw_mad(spfo_c64 *c_fft /*(k+1)*N/2*/, const uint64_t *a_glwe,
                        const spfo_c64 *ggsw_fft, size_t N, size_t k, uint32_t radix_log,
                        uint32_t count);
/* ops/fft_ops.rs:149-181 (cmux): c = d0 + IFFT(decomp(d1-d0) . ggsw) */
void spfo_cmux(uint64_t *c, const uint64_t *d0, const uint64_t *d1, const spfo_c64 *ggsw_fft,
               size_t N, size_t k, uint32_t radix_log, uint32_t count);
/* ops/bootstrapping/programmable_bootstrapping.rs:342-410.  bsk layout [n][ggsw]. */
void spfo_generalized_pbs(uint64_t *glwe_out, const uint64_t *lwe_in, const uint64_t *lut_glwe,
                          const spfo_c64 *bsk_fft, size_t n, size_t N, size_t k,
                          uint32_t radix_log, uint32_t count, uint32_t log_chi, uint32_t log_v);
/* programmable_bootstrapping.rs:291-318: generalized (0,0) + sample_extract(.,0) */
void spfo_pbs_univariate(uint64_t *lwe_out /*k*N+1*/, const uint64_t *lwe_in,
                         const uint64_t *lut_glwe, const spfo_c64 *bsk_fft, size_t n, size_t N,
                         size_t k, uint32_t radix_log, uint32_t count);
/* circuit_bootstrapping.rs:387-427 (hi_noise_lwe_to_lo_noise_glwe): the PBS part of a circuit
 * bootstrap: rotate by q/4, CBS LUT, generalized PBS with log_v = ceil(log2(cbs_count)). */
void spfo_cbs_pbs(uint64_t *glwe_out, const uint64_t *lwe_in, const spfo_c64 *bsk_fft, size_t n,
                  size_t N, size_t k, uint32_t pbs_radix_log, uint32_t pbs_count,
                  uint32_t cbs_radix_log, uint32_t cbs_count);

/* ---------------------------------------------------------------- circuit-bootstrap tail (§8 f2) */

/* ops/fft_ops.rs:457-495 (keyswitch_glwe_to_glwe); ksk_fft [row<k][level][poly<k+1][N/2] */
void spfo_keyswitch_glwe_to_glwe(uint64_t *out, const uint64_t *in, const spfo_c64 *ksk_fft, size_t N,
                                 size_t k, uint32_t radix_log, uint32_t count);
/* ops/automorphisms/mod.rs:53-85 (trace); ak_fft [i<log2 N][glwe keyswitch key] */
void spfo_trace(uint64_t *out, const uint64_t *x, const spfo_c64 *ak_fft, size_t N, size_t k,
                uint32_t radix_log, uint32_t count);
/* ops/bootstrapping/circuit_bootstrapping.rs:260-298; glev is cbs_count GLWEs */
void spfo_mod_switch_trace_and_rotate(uint64_t *glev, const uint64_t *lo_noise_glwe, const spfo_c64 *ak_fft,
                                      size_t N, size_t k, uint32_t tr_radix_log, uint32_t tr_count,
                                      uint32_t cbs_radix_log, uint32_t cbs_count);
/* ops/fft_ops.rs:403-442 (scheme_switch_fft); out [row<k+1][level<ggsw_count][poly][N/2] */
void spfo_scheme_switch_fft(spfo_c64 *out, const uint64_t *glev, const spfo_c64 *ssk_fft, size_t N,
                            size_t k, uint32_t ggsw_count, uint32_t ss_radix_log, uint32_t ss_count);
/* ops/bootstrapping/circuit_bootstrapping.rs:342-385 (circuit_bootstrap_via_trace_and_scheme_switch) */
void spfo_circuit_bootstrap(spfo_c64 *ggsw_out, const uint64_t *lwe_in, const spfo_c64 *bsk_fft,
                            const spfo_c64 *ak_fft, const spfo_c64 *ssk_fft, size_t n, size_t N, size_t k,
                            uint32_t pbs_radix_log, uint32_t pbs_count, uint32_t tr_radix_log,
                            uint32_t tr_count, uint32_t ss_radix_log, uint32_t ss_count,
                            uint32_t cbs_radix_log, uint32_t cbs_count);
/* ---------------------------------------------------------------- keygen / encrypt subset
 * (self-contained test vectors; the reference uses an unseeded thread_rng (rand.rs:23,34,39),
 * so RNG parity is neither possible nor needed).  PRNG: xoshiro256** seeded by splitmix64. */
typedef struct { uint64_t s[4]; } spfo_rng;
void spfo_rng_seed(spfo_rng *r, uint64_t seed);
uint64_t spfo_rng_next(spfo_rng *r);
/* rand.rs:20-31: round(N(0,std) * 2^64) as i64 */
uint64_t spfo_normal_torus(spfo_rng *r, double std);
void spfo_gen_binary_key(spfo_rng *r, uint64_t *key, size_t len); /* rand.rs:39-48 */
/* ops/encryption/lwe_encryption.rs:36-59 */
void spfo_encrypt_lwe(spfo_rng *r, uint64_t *ct, const uint64_t *sk, size_t n, uint64_t msg,
                      double std);
uint64_t spfo_decrypt_lwe_raw(const uint64_t *ct, const uint64_t *sk, size_t n); /* b - <a,s> */
/* ops/encryption/glwe_encryption.rs:22-61 */
void spfo_encrypt_glwe(spfo_rng *r, uint64_t *ct, const uint64_t *sk /*k*N*/,
                       const uint64_t *msg /*N*/, size_t N, size_t k, double std);
void spfo_decrypt_glwe_raw(uint64_t *msg_out /*N*/, const uint64_t *ct, const uint64_t *sk,
                           size_t N, size_t k);
/* ops/encryption/ggsw_encryption.rs:16-72 + glev_encryption.rs:23-77, message = constant
 * polynomial `bit` (encrypt_ggsw_ciphertext_scalar). out layout [row][level][poly][N] u64 */
void spfo_encrypt_ggsw_scalar(spfo_rng *r, uint64_t *ggsw, const uint64_t *glwe_sk, uint64_t bit,
                              size_t N, size_t k, uint32_t radix_log, uint32_t count, double std);
/* entities ggsw fft: every polynomial through spfo_poly_fft */
void spfo_ggsw_fft(spfo_c64 *out, const uint64_t *ggsw, size_t N, size_t k, uint32_t count);
/* programmable_bootstrapping.rs:34-58 + bootstrap_key.rs fft: BSK_i = FFT(GGSW(s_i)) */
void spfo_gen_bsk_fft(spfo_rng *r, spfo_c64 *bsk_fft, const uint64_t *lwe_sk, size_t n,
                      const uint64_t *glwe_sk, size_t N, size_t k, uint32_t radix_log,
                      uint32_t count, double std);
/* ops/keyswitch/lwe_keyswitch_key.rs:16-50 */
void spfo_gen_ksk(spfo_rng *r, uint64_t *ksk, const uint64_t *sk_in, size_t n_in,
                  const uint64_t *sk_out, size_t n_out, uint32_t radix_log, uint32_t count,
                  double std);
/* keygen for the circuit-bootstrap tail: ops/keyswitch/glwe_keyswitch_key.rs,
 * ops/automorphisms/mod.rs:18-46, ops/bootstrapping/scheme_switch.rs:22-70 — returned FFT'd */
void spfo_gen_glwe_ksk_fft(spfo_rng *r, spfo_c64 *out, const uint64_t *sk_orig, const uint64_t *sk_new,
                           size_t N, size_t k, uint32_t radix_log, uint32_t count, double std);
void spfo_gen_auto_key_fft(spfo_rng *r, spfo_c64 *ak_fft, const uint64_t *glwe_sk, size_t N, size_t k,
                           uint32_t radix_log, uint32_t count, double std);
void spfo_gen_ssk_fft(spfo_rng *r, spfo_c64 *ssk_fft, const uint64_t *glwe_sk, size_t N, size_t k,
                      uint32_t radix_log, uint32_t count, double std);
/* math/torus.rs:284-300 */
uint64_t spfo_encode(uint64_t val, uint32_t plain_bits);
uint64_t spfo_decode(uint64_t torus, uint32_t plain_bits);

/* multi-threaded driver for bench.py's cpu_baseline leg: runs `count` independent
 * spfo_cbs_pbs on `threads` pthreads, returns wall seconds. */
double spfo_bench_cbs_pbs(const uint64_t *lwe_in, size_t count, const spfo_c64 *bsk_fft, size_t n,
                          size_t N, size_t k, uint32_t pbs_radix_log, uint32_t pbs_count,
                          uint32_t cbs_radix_log, uint32_t cbs_count, int threads,
                          uint64_t *glwe_out);

/* the same for generalized_programmable_bootstrap with any (log_chi, log_v) and a shared (lut_stride 0) or
 * per-ciphertext LUT; extract != 0: programmable_bootstrap_univariate (out rows of k*N+1 words) */
double spfo_bench_generalized_pbs(const uint64_t *lwe_in, size_t count, const uint64_t *lut, size_t lut_stride,
                                  const spfo_c64 *bsk_fft, size_t n, size_t N, size_t k, uint32_t pbs_radix_log,
                                  uint32_t pbs_count, uint32_t log_chi, uint32_t log_v, int extract, int threads,
                                  uint64_t *out);

#ifdef __cplusplus
}
#endif
#endif
