/*
 * spf_hip.h — C ABI of the MI355X (gfx950) bootstrap engine.
 *
 * Drop-in boundary for the programmable-bootstrap hot path of Sunscreen-tech/spf.  The
 * reference has no FFI of its own; its seam is the concrete struct
 * `parasol_runtime::Evaluation` (parasol_runtime/src/crypto/evaluation.rs:144-266), whose
 * methods take caller-allocated `&mut` outputs, never fail, and are called concurrently from
 * rayon workers (circuit_processor/mod.rs:201-209).  Every entry point below names the
 * reference method / function whose body it replaces.  Citations are relative to the
 * reference repository root.
 *
 * Conventions
 *   - plain C, no exceptions cross the boundary; every call returns spf_status (0 = OK) and
 *     leaves a message retrievable with spf_last_error().
 *   - all arrays are dense, row-major, little-endian, in the reference's own layouts:
 *       LWE(n)        n mask words then the body                     entities/lwe_ciphertext.rs:24-33
 *       GLWE(k,N)     k mask polynomials then the body polynomial    entities/glwe_ciphertext.rs:32-41
 *       GGSW-FFT      [row<k+1][level<l][poly<k+1][bin<N/2]{re,im}   entities/ggsw_ciphertext_fft.rs:23-29
 *       BSK-FFT       [i<n] GGSW-FFT, natural DFT bin order           entities/bootstrap_key.rs:119-125
 *       KSK           [i<k*N][level<l_ks][n+1]                        entities/lwe_keyswitch_key.rs:27-36
 *     Torus elements are uint64_t (Torus<u64>, math/torus.rs:213); complex bins are
 *     interleaved {double re, double im} (num_complex::Complex<f64>).
 *   - "_batch" entry points take HOST pointers (what a Rust shim holding `&[u64]` passes) and
 *     stage through device memory; "_dev" entry points take DEVICE pointers plus a hipStream_t
 *     (passed as void*) and are asynchronous on that stream.
 *   - a context is bound to one GPU.  Calls on one context are serialised internally (thread-safe); use one context per
 *     stream for concurrency.  A device group (spf_group_*) holds one context per GPU of the node for ONE host process.
 */
#ifndef SPF_HIP_H
#define SPF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int spf_status;
enum {
    SPF_OK = 0,
    SPF_ERR_INVALID_ARGUMENT = 1, /* NULL pointer, size mismatch, unsupported parameter */
    SPF_ERR_HIP = 2,              /* a HIP runtime call failed (message has the HIP error) */
    SPF_ERR_NO_KEY = 3,           /* the key this operation needs has not been loaded */
    SPF_ERR_UNSUPPORTED = 4       /* parameter set outside what the kernels are built for */
};

/* Mirror of `parasol_runtime::Params` (parasol_runtime/src/params.rs:10-100; DEFAULT_128 at
 * :107-134) restricted to the fields the bootstrap / keyswitch path reads. */
typedef struct spf_params {
    uint32_t lwe_dimension;     /* l0_params.dim            = 637  */
    uint32_t polynomial_degree; /* l1_params.dim.polynomial_degree = 2048 */
    uint32_t glwe_size;         /* l1_params.dim.size       = 1    */
    uint32_t pbs_radix_log;     /* pbs_radix.radix_log      = 16   */
    uint32_t pbs_radix_count;   /* pbs_radix.count          = 2    */
    uint32_t cbs_radix_log;     /* cbs_radix.radix_log      = 4    */
    uint32_t cbs_radix_count;   /* cbs_radix.count          = 4    */
    uint32_t ks_radix_log;      /* ks_radix.radix_log       = 2    */
    uint32_t ks_radix_count;    /* ks_radix.count           = 6    */
    uint32_t tr_radix_log;      /* tr_radix.radix_log       = 7    (trace / automorphism keyswitch) */
    uint32_t tr_radix_count;    /* tr_radix.count           = 6    */
    uint32_t ss_radix_log;      /* ss_radix.radix_log       = 3    (scheme switch) */
    uint32_t ss_radix_count;    /* ss_radix.count           = 15   */
} spf_params;

/* DEFAULT_128 (parasol_runtime/src/params.rs:107-134): the parameter set the tuned kernels are built for (lwe_dimension and the
 * keyswitch radix are free).  Any other set with polynomial_degree a power of two in 16 .. 2048 and radices with
 * radix_log * count < 64 whose (glwe_size + 1) polynomials fit one CU's LDS is served by a generic, untuned kernel family with the
 * same results as the reference's generic functions (every entry point: batch and device-pointer forms, the pool by host pointer
 * and by handle, the gate graphs, the device group); anything else — e.g. polynomial_degree 2048 with glwe_size >= 2 — is
 * SPF_ERR_UNSUPPORTED. */
void spf_default_params(spf_params *out);

typedef struct spf_ctx spf_ctx;

/* Replaces `Evaluation::new` (crypto/evaluation.rs:161-197) minus key ownership: creates the
 * per-GPU engine (twiddle tables, scratch).  device_id is the HIP ordinal. */
spf_status spf_create(const spf_params *params, int device_id, spf_ctx **out);
void spf_destroy(spf_ctx *ctx);
/* Message of the last failing call on this context (or, with ctx == NULL, of the last failing
 * spf_create on the calling thread).  The returned pointer is storage of the calling thread: it
 * stays valid until that thread calls spf_last_error again, whatever other threads do. */
const char *spf_last_error(const spf_ctx *ctx);

/* ---- keys: `ComputeKey` fields (crypto/keys.rs:306-318) ------------------------------- */

/* `ComputeKey::bs_key` : BootstrapKeyFft<Complex<f64>>.  n_complex must equal
 * lwe_dimension * (k+1)*l_pbs*(k+1)*N/2.  Host pointer; copied to HBM.  The engine also keeps a copy
 * scaled by 2^-10 for its blind-rotation kernels (the inverse transform's 1/N travels with the key: exact
 * for every spectrum a torus polynomial has, results unchanged), so a key holding a NaN or a non-zero
 * magnitude outside [2^-900, 2^1000) — which no forward transform produces — is refused with
 * SPF_ERR_INVALID_ARGUMENT here and in spf_key_blob_commit(ctx, 0), and leaves the context without a
 * bootstrap key. */
spf_status spf_load_bootstrap_key(spf_ctx *ctx, const double *bsk_fft, size_t n_complex);
/* `ComputeKey::ks_key` : LweKeyswitchKey<u64>.  n_words = k*N * l_ks * (lwe_dimension+1). */
spf_status spf_load_keyswitch_key(spf_ctx *ctx, const uint64_t *ksk, size_t n_words);

/* `ComputeKey::auto_key` : AutomorphismKeyFft<Complex<f64>> = log2(N) GLWE keyswitch keys
 * [i<log2 N][row<k][level<l_tr][poly<k+1][bin<N/2] (entities/automorphism_key.rs). */
spf_status spf_load_automorphism_key(spf_ctx *ctx, const double *ak_fft, size_t n_complex);
/* `ComputeKey::ss_key` : SchemeSwitchKeyFft<Complex<f64>> = k(k+1)/2 GLEVs
 * [pair][level<l_ss][poly<k+1][bin<N/2] (entities/scheme_switch_key.rs). */
spf_status spf_load_scheme_switch_key(spf_ctx *ctx, const double *ssk_fft, size_t n_complex);

/* Multi-GPU key replication (no reference counterpart): the device-resident
 * key blobs, so that a caller can RCCL-broadcast rank 0's keys into every other rank's
 * context.  which: 0 = bootstrap key, 1 = keyswitch key, 2 = automorphism key,
 * 3 = scheme-switch key.  Allocates the blob if needed;
 * after filling it externally call spf_key_blob_commit. */
spf_status spf_key_blob(spf_ctx *ctx, int which, void **dev_ptr, size_t *bytes);
spf_status spf_key_blob_commit(spf_ctx *ctx, int which);

/* ---- the hot path, host-pointer batch forms --------------------------------------------- */

/* B x `Evaluation::keyswitch_lwe_l1_lwe_l0` (crypto/evaluation.rs:246-255) =
 * `keyswitch_lwe_to_lwe` (sunscreen_tfhe/src/ops/keyswitch/lwe_keyswitch.rs:23-62).
 * lwe1_in: B x (k*N+1), lwe0_out: B x (lwe_dimension+1). */
spf_status spf_keyswitch_lwe_l1_lwe_l0_batch(spf_ctx *ctx, size_t B, const uint64_t *lwe1_in,
                                             uint64_t *lwe0_out);

/* B x `generalized_programmable_bootstrap`
 * (sunscreen_tfhe/src/ops/bootstrapping/programmable_bootstrapping.rs:342-410).
 * lwe0_in : B x (lwe_dimension+1)
 * lut_glwe: the UnivariateLookupTable's GLWE, (k+1)*N words; lut_stride = 0 shares one LUT
 *           across the batch, otherwise ciphertext j uses lut_glwe + j*lut_stride.
 * body_rotate is added to each input body first (`lwe_rotate`,
 *           ops/homomorphisms/lwe.rs:9-20; 0 for a plain PBS).
 * glwe_out: B x (k+1)*N. */
spf_status spf_generalized_pbs_batch(spf_ctx *ctx, size_t B, const uint64_t *lwe0_in,
                                     const uint64_t *lut_glwe, size_t lut_stride, uint32_t log_chi,
                                     uint32_t log_v, uint64_t body_rotate, uint64_t *glwe_out);

/* B x `programmable_bootstrap_univariate` (programmable_bootstrapping.rs:291-318):
 * generalized PBS with (log_chi, log_v) = (0, 0) then `sample_extract(., 0)`.
 * lwe1_out: B x (k*N+1). */
spf_status spf_pbs_univariate_batch(spf_ctx *ctx, size_t B, const uint64_t *lwe0_in,
                                    const uint64_t *lut_glwe, size_t lut_stride,
                                    uint64_t *lwe1_out);

/* B x the bootstrap stage of `Evaluation::circuit_bootstrap` (crypto/evaluation.rs:211-226):
 * `hi_noise_lwe_to_lo_noise_glwe` (ops/bootstrapping/circuit_bootstrapping.rs:387-427) =
 * rotate by q/4, multifunctional CBS LUT (:430-482), generalized PBS with
 * log_v = ceil(log2(cbs_radix_count)).  glwe_out: B x (k+1)*N. */
spf_status spf_circuit_bootstrap_pbs_batch(spf_ctx *ctx, size_t B, const uint64_t *lwe0_in,
                                           uint64_t *glwe_out);

/* B x `Evaluation::circuit_bootstrap` (crypto/evaluation.rs:211-226) =
 * `circuit_bootstrap_via_trace_and_scheme_switch` (ops/bootstrapping/circuit_bootstrapping.rs:342-385):
 * the bootstrap above, then `mod_switch_trace_and_rotate` (:260-298) and `scheme_switch_fft`
 * (ops/fft_ops.rs:403-442).  ggsw_fft_out: B x (k+1)*l_cbs*(k+1)*N/2 complex, the layout
 * `L1GgswCiphertext` holds and `cmux` consumes.  Needs all four keys. */
spf_status spf_circuit_bootstrap_batch(spf_ctx *ctx, size_t B, const uint64_t *lwe0_in,
                                       double *ggsw_fft_out);
/* B x `mod_switch_trace_and_rotate` alone: lo-noise GLWE -> GLEV (l_cbs GLWEs per ciphertext). */
spf_status spf_mod_switch_trace_and_rotate_batch(spf_ctx *ctx, size_t B, const uint64_t *glwe_in,
                                                 uint64_t *glev_out);
/* B x `Evaluation::scheme_switch` (crypto/evaluation.rs:231-240) = `scheme_switch_fft`:
 * GLEV (l_cbs GLWEs) -> GGSW-FFT. */
spf_status spf_scheme_switch_batch(spf_ctx *ctx, size_t B, const uint64_t *glev_in, double *ggsw_fft_out);

/* B x `KeylessEvaluation::sample_extract_l1` (crypto/evaluation.rs:126-133) =
 * `sample_extract` (ops/ciphertext/glwe_ciphertext_ops.rs:31-76), same index for the batch. */
spf_status spf_sample_extract_l1_batch(spf_ctx *ctx, size_t B, const uint64_t *glwe_in, size_t idx,
                                       uint64_t *lwe1_out);

/* The linear `KeylessEvaluation` operations on L1 GLWE ciphertexts (B x (k+1)*N words each):
 *   not    (crypto/evaluation.rs:47-50): out = in + trivial_one, i.e. body coefficient 0 += 2^63
 *          (`trivial_glwe_l1_one`, crypto/encryption.rs:359-364, one plaintext bit :132);
 *   xor    (:52-55): out = a + b, wrapping;
 *   mul_xn (:57-65): out = in * X^n mod X^N + 1 = `rotate_glwe_positive_monomial_negacyclic`
 *          (sunscreen_tfhe/src/ops/bootstrapping/blind_rotation.rs:126-135), same n for the batch,
 *          n taken mod 2N (entities/polynomial.rs:208-236). */
spf_status spf_glwe_not_batch(spf_ctx *ctx, size_t B, const uint64_t *glwe_in, uint64_t *glwe_out);
spf_status spf_glwe_xor_batch(spf_ctx *ctx, size_t B, const uint64_t *a, const uint64_t *b,
                              uint64_t *glwe_out);
spf_status spf_glwe_mul_xn_batch(spf_ctx *ctx, size_t B, const uint64_t *glwe_in, size_t n,
                                 uint64_t *glwe_out);

/* B x `KeylessEvaluation::cmux` (crypto/evaluation.rs:68-83) = `cmux`
 * (sunscreen_tfhe/src/ops/fft_ops.rs:149-181) with the GGSW in cbs_radix shape.
 * sel_ggsw_fft: B x (k+1)*l_cbs*(k+1)*N/2 complex; a (selected when 0), b (when 1), out:
 * B x (k+1)*N. */
spf_status spf_cmux_batch(spf_ctx *ctx, size_t B, const double *sel_ggsw_fft, const uint64_t *a,
                          const uint64_t *b, uint64_t *out);

/* B x `KeylessEvaluation::glev_cmux` (crypto/evaluation.rs:86-101) = `glev_cmux`
 * (ops/fft_ops.rs:203-220): one selector per item; a / b / out are GLEVs of l_cbs GLWEs
 * (B x l_cbs x (k+1)*N words). */
spf_status spf_glev_cmux_batch(spf_ctx *ctx, size_t B, const double *sel_ggsw_fft, const uint64_t *a,
                               const uint64_t *b, uint64_t *out);
/* B x `KeylessEvaluation::multiply_glwe_ggsw` (crypto/evaluation.rs:104-123):
 * out = IFFT(glwe [*] ggsw), `glwe_ggsw_mad` into a cleared accumulator (ops/fft_ops.rs:23-56). */
spf_status spf_multiply_glwe_ggsw_batch(spf_ctx *ctx, size_t B, const uint64_t *glwe, const double *ggsw_fft,
                                        uint64_t *out);

/* The north-star "gate": keyswitch L1->L0 then the circuit-bootstrap PBS, fused on device
 * (FheOp::KeyswitchL1toL0 -> FheOp::CircuitBootstrap, circuit_processor/mod.rs:329-340,453-463).
 * lwe1_in: B x (k*N+1); glwe_out: B x (k+1)*N. */
spf_status spf_gate_bootstrap_batch(spf_ctx *ctx, size_t B, const uint64_t *lwe1_in,
                                    uint64_t *glwe_out);
/* B x (FheOp::KeyswitchL1toL0 -> FheOp::CircuitBootstrap) with the whole circuit bootstrap
 * (`Evaluation::keyswitch_lwe_l1_lwe_l0` then `Evaluation::circuit_bootstrap`, crypto/evaluation.rs:211-266):
 * L1 LWE in, L1 GGSW-FFT out (cbs radix); the level-0 LWE in between stays in HBM. */
spf_status spf_keyswitch_circuit_bootstrap_batch(spf_ctx *ctx, size_t B, const uint64_t *lwe1_in,
                                                 double *ggsw_fft_out);

/* ---- device-pointer forms (inputs/outputs resident in HBM, asynchronous on `stream`) -----
 * Contract of every `_dev` entry point: the call only ENQUEUES on `stream`; inputs and outputs must stay valid and
 * unchanged until that work has completed; an output must not overlap any input of the same call.  The entry points that
 * need intermediates (`spf_circuit_bootstrap_dev`, the keyswitch) keep them in buffers of the CONTEXT: enqueue them on ONE
 * stream per context (or order the streams with events) — two such calls running concurrently on different streams would
 * share those buffers.  `spf_mod_switch_trace_and_rotate_dev` additionally uses its OUTPUT as working memory while it runs
 * (the kernel parks half of its accumulator in each unit's 32 KiB of `d_glev_out` between automorphism rounds):
 * `d_glev_out` holds intermediate data until the kernel has completed and must not be read, or alias anything read, by
 * work that may run concurrently with it. */

spf_status spf_keyswitch_lwe_l1_lwe_l0_dev(spf_ctx *ctx, void *stream, size_t B,
                                           const uint64_t *d_lwe1_in, uint64_t *d_lwe0_out);
spf_status spf_generalized_pbs_dev(spf_ctx *ctx, void *stream, size_t B, const uint64_t *d_lwe0_in,
                                   const uint64_t *d_lut_glwe, size_t lut_stride, uint32_t log_chi,
                                   uint32_t log_v, uint64_t body_rotate, uint64_t *d_glwe_out);
spf_status spf_pbs_univariate_dev(spf_ctx *ctx, void *stream, size_t B, const uint64_t *d_lwe0_in,
                                  const uint64_t *d_lut_glwe, size_t lut_stride,
                                  uint64_t *d_lwe1_out);
spf_status spf_circuit_bootstrap_pbs_dev(spf_ctx *ctx, void *stream, size_t B,
                                         const uint64_t *d_lwe0_in, uint64_t *d_glwe_out);
spf_status spf_circuit_bootstrap_dev(spf_ctx *ctx, void *stream, size_t B, const uint64_t *d_lwe0_in,
                                     double *d_ggsw_fft_out);
spf_status spf_mod_switch_trace_and_rotate_dev(spf_ctx *ctx, void *stream, size_t B,
                                               const uint64_t *d_glwe_in, uint64_t *d_glev_out);
spf_status spf_scheme_switch_dev(spf_ctx *ctx, void *stream, size_t B, const uint64_t *d_glev_in,
                                 double *d_ggsw_fft_out);
spf_status spf_sample_extract_l1_dev(spf_ctx *ctx, void *stream, size_t B,
                                     const uint64_t *d_glwe_in, size_t idx, uint64_t *d_lwe1_out);
spf_status spf_glwe_not_dev(spf_ctx *ctx, void *stream, size_t B, const uint64_t *d_in, uint64_t *d_out);
spf_status spf_glwe_xor_dev(spf_ctx *ctx, void *stream, size_t B, const uint64_t *d_a,
                            const uint64_t *d_b, uint64_t *d_out);
spf_status spf_glwe_mul_xn_dev(spf_ctx *ctx, void *stream, size_t B, const uint64_t *d_in, size_t n,
                               uint64_t *d_out);
spf_status spf_cmux_dev(spf_ctx *ctx, void *stream, size_t B, const double *d_sel_ggsw_fft,
                        const uint64_t *d_a, const uint64_t *d_b, uint64_t *d_out);
spf_status spf_glev_cmux_dev(spf_ctx *ctx, void *stream, size_t B, const double *d_sel_ggsw_fft,
                             const uint64_t *d_a, const uint64_t *d_b, uint64_t *d_out);
spf_status spf_multiply_glwe_ggsw_dev(spf_ctx *ctx, void *stream, size_t B, const uint64_t *d_glwe,
                                      const double *d_ggsw_fft, uint64_t *d_out);

/* ciphertext types (`L0LweCiphertext` ... `L1GlevCiphertext`, crypto/encryption.rs:23-110) and the computing variants of `FheOp`
 * (fhe_circuit.rs:65-126): used by the values, the pool's generic submit and the gate graphs below */
typedef enum spf_value_kind {
    SPF_VAL_LWE0 = 0,  /* L0LweCiphertext: n+1 words */
    SPF_VAL_LWE1 = 1,  /* L1LweCiphertext: k*N+1 words */
    SPF_VAL_GLWE1 = 2, /* L1GlweCiphertext: (k+1)*N words */
    SPF_VAL_GGSW1 = 3, /* L1GgswCiphertext, FFT domain: (k+1)*l_cbs*(k+1)*N/2 complex */
    SPF_VAL_GLEV1 = 4  /* L1GlevCiphertext: l_cbs GLWEs */
} spf_value_kind;
typedef enum spf_graph_op {
    SPF_OP_SAMPLE_EXTRACT = 0,
    SPF_OP_KEYSWITCH_L1_TO_L0 = 1,
    SPF_OP_NOT = 2,
    SPF_OP_GLWE_ADD = 3,
    SPF_OP_CMUX = 4,
    SPF_OP_GLEV_CMUX = 5,
    SPF_OP_MULTIPLY_GGSW_GLWE = 6,
    SPF_OP_CIRCUIT_BOOTSTRAP = 7,
    SPF_OP_SCHEME_SWITCH = 8,
    SPF_OP_MUL_XN = 9
} spf_graph_op;

/* ---- call coalescing: many threads, one ciphertext each -> one batch per launch ----------
 * The reference calls `Evaluation` from many rayon workers with a single ciphertext per call
 * (circuit_processor/mod.rs:192-253).  A pool keeps that calling convention — submit, then wait like the synchronous call it
 * replaces — and runs what has gathered of one kind as ONE batch.
 *   Buffers: the INPUT is copied into the pool's pinned staging during submit (it may be released when submit returns); the
 *     OUTPUT buffer must stay valid until spf_pool_wait returns for its ticket — or, for a ticket nobody waits for, until the
 *     pool is destroyed: an output left uncollected for 200 ms is written to its buffer by a pool thread so that the staging
 *     set can be reused (a later wait still returns the status, and never returns while that copy is in progress).
 *   Batching: a calling thread is dealt to one of up to four caller groups on its first submit and stays there; a group's
 *     batch closes when it is full (max_batch, and the staging a batch may pin), when every caller of the group's previous
 *     batch is back, or when nobody has joined it for max_wait_us (stretched to an eighth of the last batch's GPU time, at most
 *     twenty such quiet times after its first member) — but not on the timer while the group's previous batch is still out.
 *     The groups' batches run on the GPU side by side, each on its own stream (profiles/r05_pool.md).
 *   Asynchronous use: a thread may hold many tickets; sixteen staging sets exist (spf_pool_counters.staging_sets), so a caller
 *     that keeps more than sixteen host-pointer batches uncollected waits in submit until a set is collected (or 200 ms old).
 *     Batches by handle give their set back when they complete.
 *   spf_pool_wait returns the status of the batch the operation ran in (first-error-wins per batch, as
 *     circuit_processor/mod.rs:214-223 does per graph).
 *   The library asks the HIP runtime for more hardware queues when it is loaded (GPU_MAX_HW_QUEUES=24 unless set): streams that
 *   share a hardware queue run their kernels one after the other. */
typedef struct spf_pool spf_pool;
spf_status spf_pool_create(spf_ctx *ctx, size_t max_batch, uint32_t max_wait_us, spf_pool **out);
void spf_pool_destroy(spf_pool *pool); /* drains pending work first */
/* `Evaluation::keyswitch_lwe_l1_lwe_l0` for one ciphertext */
spf_status spf_pool_submit_keyswitch(spf_pool *pool, const uint64_t *lwe1_in, uint64_t *lwe0_out, uint64_t *ticket);
/* `Evaluation::circuit_bootstrap` for one ciphertext */
spf_status spf_pool_submit_circuit_bootstrap(spf_pool *pool, const uint64_t *lwe0_in, double *ggsw_fft_out,
                                             uint64_t *ticket);
/* FheOp::KeyswitchL1toL0 -> FheOp::CircuitBootstrap for one ciphertext (L1 LWE in, L1 GGSW out) */
spf_status spf_pool_submit_keyswitch_circuit_bootstrap(spf_pool *pool, const uint64_t *lwe1_in,
                                                       double *ggsw_fft_out, uint64_t *ticket);
/* `KeylessEvaluation::cmux` for one gate */
spf_status spf_pool_submit_cmux(spf_pool *pool, const double *sel_ggsw_fft, const uint64_t *a, const uint64_t *b,
                                uint64_t *out, uint64_t *ticket);
/* The other operations `CircuitProcessor::exec_op` issues per task (circuit_processor/mod.rs:341-540), one ciphertext per call:
 *   FheOp::SampleExtract(idx) -> `KeylessEvaluation::sample_extract_l1` (crypto/evaluation.rs:126-133); idx >= N is
 *                                SPF_ERR_INVALID_ARGUMENT.  Calls with different indices run in different batches.
 *   FheOp::Not / GlweAdd / MulXN(n) -> `not` / `xor` / `mul_xn` (crypto/evaluation.rs:47-66); n is taken mod 2N
 *   FheOp::MultiplyGgswGlwe -> `multiply_glwe_ggsw` (:104-123)         FheOp::GlevCMux -> `glev_cmux` (:86-101), GLEVs of l_cbs GLWEs
 *   FheOp::SchemeSwitch -> `Evaluation::scheme_switch` (:231-240) */
spf_status spf_pool_submit_sample_extract(spf_pool *pool, const uint64_t *glwe_in, size_t idx, uint64_t *lwe1_out, uint64_t *ticket);
spf_status spf_pool_submit_not(spf_pool *pool, const uint64_t *glwe_in, uint64_t *glwe_out, uint64_t *ticket);
spf_status spf_pool_submit_glwe_add(spf_pool *pool, const uint64_t *a, const uint64_t *b, uint64_t *glwe_out, uint64_t *ticket);
spf_status spf_pool_submit_mul_xn(spf_pool *pool, const uint64_t *glwe_in, size_t n, uint64_t *glwe_out, uint64_t *ticket);
spf_status spf_pool_submit_multiply_ggsw_glwe(spf_pool *pool, const double *ggsw_fft, const uint64_t *glwe, uint64_t *glwe_out,
                                              uint64_t *ticket);
spf_status spf_pool_submit_glev_cmux(spf_pool *pool, const double *sel_ggsw_fft, const uint64_t *a, const uint64_t *b,
                                     uint64_t *glev_out, uint64_t *ticket);
spf_status spf_pool_submit_scheme_switch(spf_pool *pool, const uint64_t *glev_in, double *ggsw_fft_out, uint64_t *ticket);
/* Blocks until the operation has run; each ticket can be collected exactly once (an unknown or already
 * collected ticket is SPF_ERR_INVALID_ARGUMENT, never a hang). */
spf_status spf_pool_wait(spf_pool *pool, uint64_t ticket);
/* Flow control (the reference bounds its in-flight operations with a token channel,
 * circuit_processor/mod.rs:139): a submit blocks while `max_inflight` tickets are submitted and not yet
 * collected.  Default 4 x max_batch. */
spf_status spf_pool_set_max_inflight(spf_pool *pool, size_t max_inflight);
/* operations completed and batches launched so far (ops / launches = achieved batch size) */
spf_status spf_pool_stats(spf_pool *pool, uint64_t *ops, uint64_t *launches);

/* counters of a pool (sums over the members of a group pool) */
typedef struct spf_pool_counters {
    uint64_t ops, launches;                  /* as spf_pool_stats: operations completed, batches launched (both forms) */
    uint64_t handle_ops, handle_launches;    /* ... of which by handle */
    uint64_t reclaimed;                      /* outputs a pool thread delivered on behalf of a caller that never collected them */
    uint64_t bootstrap_launches_by_shape[3]; /* circuit-bootstrap batches by blind-rotation shape: eight waves per ciphertext
                                              * (blind_rotate8), two ciphertexts per workgroup (2p2), four (2p) */
    uint64_t staging_sets;                   /* staging sets per pool (a caller may leave that many batches uncollected) */
    uint64_t value_mallocs;                  /* hipMalloc calls of the value arena so far (steady state: no growth) */
    uint64_t stream_concurrency;             /* how many of the pool's streams ran side by side in the probe spf_pool_create makes
                                              * (a 200 us spin kernel on every set's stream at once; minimum over the members;
                                              * the best of three probes).  At 5 or below the resident batches take turns: GPU_MAX_HW_QUEUES took effect too late
                                              * (a host that touched HIP before loading the library) — spf_pool_create then
                                              * still returns SPF_OK and leaves a WARNING in spf_last_error(ctx) */
} spf_pool_counters;
spf_status spf_pool_counters_get(spf_pool *pool, spf_pool_counters *out);

/* ---- device-resident values: the per-operation boundary without PCIe (SURVEY.md §8 b / f3) ---------------------------- *
 *
 * `CircuitProcessor::exec_op` calls `Evaluation` once per `FheOp` (circuit_processor/mod.rs:255-540) with ciphertexts that
 * live in host memory (crypto/encryption.rs:143-165); the GGSW a `CircuitBootstrap` produces is consumed by the ~45 CMux
 * gates behind it (fhe_circuit.rs:473-494).  Through the host-pointer submits above every operand and result crosses PCIe on
 * every call (a CMux: 256 KiB + 2 x 32 KiB in, 32 KiB out).  A *value* is a reference-counted ciphertext in the HBM of one
 * context; the `_v` submits take values and return a value, and nothing crosses PCIe until spf_value_download.  A Rust shim
 * keeps an `Option<SpfValue>` beside (or instead of) the host copy inside `L1GgswCiphertext` & co..
 *
 *   Lifetime: every value returned through an `out` parameter carries ONE reference owned by the caller; spf_value_release
 *     drops it (spf_value_retain adds one).  The pool keeps operands alive while an operation that reads them is queued or
 *     running, so a caller may release an operand right after the submit that used it.  Values may outlive their pool (their
 *     memory is freed on release); they must not be USED with another pool.
 *   Validity: a result value exists as soon as the submit returns, but becomes valid only when its operation has run:
 *     spf_pool_wait has returned SPF_OK for its ticket — exactly when the reference's task output becomes visible to its
 *     dependents (circuit_processor/mod.rs:214-246) — or spf_value_wait has returned SPF_OK for the value.  Reading it earlier
 *     (spf_value_download, spf_value_device_ptr, spf_value_copy_to_member), or after a failed batch, is
 *     SPF_ERR_INVALID_ARGUMENT, never a read of unfinished data.  It must be released in every case.
 *   Deferred operands: a result that is still pending MAY be passed as an operand of a later `_v` submit to the same pool.  The
 *     pool orders the two on the device (stream order, events, or — behind a bootstrap batch — launching the dependent batch
 *     when the producing one has finished) and batches what has been pushed by level: operations on pending operands join the
 *     open batch of their (depth, kind, parameter), depth = 1 + the deepest batch an operand comes from (the bootstrap kinds:
 *     one batch per number of bootstraps they are behind, whatever the depth), and the table is launched in dependency order
 *     when a result in it is waited for or when nothing has joined it for max_wait_us.  A caller can thus push a whole circuit —
 *     every `FheOp` one `_v` submit, from one thread, without a single wait — and wait for the outputs only: the level batching
 *     of spf_graph_run, built while the operations arrive.  An operation whose operand's producer failed fails with that status.
 *     `ticket` may be NULL in the `_v` submits: nobody will spf_pool_wait for that operation (no ticket to collect, no
 *     back-pressure from it); spf_value_wait on the result, or on anything computed from it, takes its place.
 *   Placement: a value lives on ONE member of a group pool (member 0 of a plain pool).  `member` < 0 in spf_value_upload /
 *     spf_value_trivial = the calling thread's home member (as the host-pointer submits deal their callers); an operation
 *     runs on the member its operands live on, its result stays there; operands on different members are
 *     SPF_ERR_INVALID_ARGUMENT — spf_value_copy_to_member makes a copy on another member (peer copy over xGMI).
 *   Memory: the results of one batch share one block of the pool's arena (the kernels write consecutive rows); the block is
 *     reused when its last value is released, so a value pins the block of its batch mates.  Blocks are cached (never
 *     hipFree'd on the steady-state path: hipFree waits for the whole device); spf_pool_trim gives the cache back to the
 *     driver, spf_pool_value_stats reports live values, live bytes (blocks handed out) and cached bytes.
 *     Environment: SPF_VALUE_CACHE_MB bounds the cache (default 32768). */
typedef struct spf_value spf_value;
/* host -> HBM: `host` holds one ciphertext of `kind` in the layout of the conventions above (spf_ciphertext_words(kind) u64
 * words; SPF_VAL_GGSW1: (k+1)*l_cbs*(k+1)*N/2 complex) */
spf_status spf_value_upload(spf_pool *pool, int member, spf_value_kind kind, const void *host, spf_value **out);
/* n ciphertexts of one kind (the bits of an encrypted integer: `host` holds them consecutively) into ONE block with one copy;
 * out receives n values.  Operations that take them in order find them consecutive in HBM (no packing pass). */
spf_status spf_value_upload_batch(spf_pool *pool, int member, spf_value_kind kind, size_t n, const void *host, spf_value **out);
/* n valid values of one kind on one member -> `host`, consecutively (blocking): one copy when they lie consecutively in one block
 * (the results of one batch in slot order, or of spf_value_upload_batch), one copy each otherwise */
spf_status spf_value_download_batch(size_t n, const spf_value *const *values, void *host);
/* FheOp::{Zero,One}{Lwe0,Glwe1,Glev1,Ggsw1} (fhe_circuit.rs:96-116; also LWE1) as values: the trivial encryption of `bit`; the
 * GGSW constants are `Evaluation::l1ggsw_zero / l1ggsw_one` (crypto/evaluation.rs:254-262) and need all four keys */
spf_status spf_value_trivial(spf_pool *pool, int member, spf_value_kind kind, uint64_t bit, spf_value **out);
/* HBM -> host (blocking); `host` has room for the kind's words */
spf_status spf_value_download(const spf_value *value, void *host);
/* blocks until the operation that produces `value` has run (returns at once for a valid value): SPF_OK, or the failure;
 * any thread, any number of times, whether or not the operation has a ticket.  Launches what the value needs: the deferred table,
 * or the open batch the operation sits in (spf_pool_wait leaves such a batch open for other callers to join). */
spf_status spf_value_wait(const spf_value *value);
/* launches what has been pushed so far (the deferred table, see Deferred operands) without waiting for anything: a pusher that
 * knows a long operation is complete — the conversions at the head of a circuit — lets it start while it pushes the rest */
spf_status spf_pool_flush(spf_pool *pool);
spf_status spf_value_retain(spf_value *value);
void spf_value_release(spf_value *value);
/* any of the out pointers may be NULL */
spf_status spf_value_info(const spf_value *value, spf_value_kind *kind, size_t *bytes, int *member, int *valid);
/* the device address of a valid value, for a caller that chains the `_dev` entry points on the member's context
 * (spf_group_ctx); stays valid while the caller holds its reference */
spf_status spf_value_device_ptr(const spf_value *value, void **dev_ptr);
/* a copy of a valid value on `member` of the group pool (device-to-device on the same GPU, peer copy otherwise; blocking) */
spf_status spf_value_copy_to_member(spf_pool *pool, const spf_value *value, int member, spf_value **out);
spf_status spf_pool_value_stats(spf_pool *pool, size_t *live_values, size_t *live_bytes, size_t *cached_bytes);
spf_status spf_pool_trim(spf_pool *pool);

/* The pool's submits by handle: same operations, same batching, same spf_pool_wait as the host-pointer forms above; operands
 * are values of this pool — valid, or still pending (see Deferred operands) —, *out receives the result value (see Validity),
 * `ticket` may be NULL.  Batches by handle never mix with host-pointer callers.  No staging, no copies: the CMUX family reads its operands where they are, the other kinds pack theirs on the device
 * (gather_rows_kernel) unless they already lie consecutively. */
spf_status spf_pool_submit_keyswitch_v(spf_pool *pool, const spf_value *lwe1, spf_value **lwe0_out, uint64_t *ticket);
spf_status spf_pool_submit_circuit_bootstrap_v(spf_pool *pool, const spf_value *lwe0, spf_value **ggsw_out, uint64_t *ticket);
spf_status spf_pool_submit_keyswitch_circuit_bootstrap_v(spf_pool *pool, const spf_value *lwe1, spf_value **ggsw_out,
                                                         uint64_t *ticket);
spf_status spf_pool_submit_cmux_v(spf_pool *pool, const spf_value *sel_ggsw, const spf_value *a, const spf_value *b,
                                  spf_value **out, uint64_t *ticket);
spf_status spf_pool_submit_sample_extract_v(spf_pool *pool, const spf_value *glwe, size_t idx, spf_value **lwe1_out,
                                            uint64_t *ticket);
spf_status spf_pool_submit_not_v(spf_pool *pool, const spf_value *glwe, spf_value **out, uint64_t *ticket);
spf_status spf_pool_submit_glwe_add_v(spf_pool *pool, const spf_value *a, const spf_value *b, spf_value **out, uint64_t *ticket);
spf_status spf_pool_submit_mul_xn_v(spf_pool *pool, const spf_value *glwe, size_t n, spf_value **out, uint64_t *ticket);
spf_status spf_pool_submit_multiply_ggsw_glwe_v(spf_pool *pool, const spf_value *ggsw, const spf_value *glwe, spf_value **out,
                                                uint64_t *ticket);
spf_status spf_pool_submit_glev_cmux_v(spf_pool *pool, const spf_value *sel_ggsw, const spf_value *a, const spf_value *b,
                                       spf_value **glev_out, uint64_t *ticket);
spf_status spf_pool_submit_scheme_switch_v(spf_pool *pool, const spf_value *glev, spf_value **ggsw_out, uint64_t *ticket);
/* `exec_op`'s whole match in one entry (circuit_processor/mod.rs:255-540): the operation as a spf_graph_op, operands in the order
 * spf_graph_add_op takes them (CMUX: selector, low, high), `param` = SampleExtract index / MulXN amount */
spf_status spf_pool_submit_op_v(spf_pool *pool, spf_graph_op op, const spf_value *const *inputs, size_t n_inputs, uint64_t param,
                                spf_value **out, uint64_t *ticket);

/* ---- gate graphs: level-batched, device-resident execution ------------- *
 *
 * The counterpart of `FheCircuit` + `CircuitProcessor::run_graph_blocking`
 * (parasol_runtime/src/fhe_circuit.rs:34-126, circuit_processor/mod.rs:573-623): build a DAG of
 * `FheOp`-like nodes, then run it.  Execution is by topological level: all nodes of one level
 * and one kind are ONE batched launch, intermediates never leave HBM, the whole graph is
 * enqueued on one stream.  Node ids are dense, in creation order; operands must already exist
 * (so the node order is a topological order).  Operand order per operation:
 *   SAMPLE_EXTRACT(param = index) [glwe1] -> lwe1      KEYSWITCH_L1_TO_L0 [lwe1] -> lwe0
 *   CIRCUIT_BOOTSTRAP [lwe0] -> ggsw1                  SCHEME_SWITCH [glev1] -> ggsw1
 *   NOT [glwe1]   GLWE_ADD [glwe1, glwe1]   MUL_XN(param = n) [glwe1]           -> glwe1
 *   CMUX [sel ggsw1, low glwe1 (taken when sel = 0), high glwe1] -> glwe1   (FheEdge::Sel/Low/High)
 *   GLEV_CMUX [sel ggsw1, low glev1, high glev1] -> glev1
 *   MULTIPLY_GGSW_GLWE [ggsw1, glwe1] -> glwe1
 * Wrong arity or operand type is reported by spf_graph_add_op, before anything runs (the
 * reference validates per task, task.rs:26-31).  Input and output host buffers are read /
 * written by every spf_graph_run and must stay valid until it returns; a graph can be run
 * repeatedly with new input contents.  A graph is not thread-safe; distinct graphs on one
 * context may run from different threads (their launches serialise on the context's stream). */
typedef struct spf_graph spf_graph;
spf_status spf_graph_create(spf_ctx *ctx, spf_graph **out);
void spf_graph_destroy(spf_graph *graph);
/* FheOp::Input{Lwe0,Lwe1,Glwe1,Ggsw1,Glev1} */
spf_status spf_graph_add_input(spf_graph *graph, spf_value_kind kind, const void *host, uint32_t *node);
/* FheOp::{Zero,One}{Lwe0,Glwe1,Glev1,Ggsw1} (fhe_circuit.rs:96-116; also LWE1): trivial encryption of a bit;
 * the GGSW constants are the context's circuit bootstraps of the trivial L0 LWE (Evaluation::l1ggsw_zero /
 * l1ggsw_one, crypto/evaluation.rs:161-197, 254-262) and need the bootstrap, automorphism and scheme-switch keys */
spf_status spf_graph_add_trivial(spf_graph *graph, spf_value_kind kind, uint64_t bit, uint32_t *node);
spf_status spf_graph_add_op(spf_graph *graph, spf_graph_op op, const uint32_t *inputs, size_t n_inputs,
                            uint64_t param, uint32_t *node);
/* FheOp::Output*: copy the node's value to `host` at the end of every run.  Plain (pageable) buffers: the run gathers every
 * output on the device and brings them back in ONE copy through the graph's own pinned staging (inputs go up the same way);
 * per-output copies to pageable memory cost ~20 us each on this runtime — 0.68 ms of a 32-bit addition's 5.5. */
spf_status spf_graph_add_output(spf_graph *graph, uint32_t node, void *host);
/* `run_graph_blocking`: returns when every output has been written */
spf_status spf_graph_run(spf_graph *graph);
/* nodes in the graph; levels and kernel launches of the most recent run */
spf_status spf_graph_stats(spf_graph *graph, uint32_t *nodes, uint32_t *levels, uint32_t *launches);

/* cmux over operands that are not contiguous: d_ptrs is a DEVICE array of 4 pointers per unit,
 * {selector GGSW-FFT, a (NULL = the zero ciphertext, i.e. multiply_glwe_ggsw), b, out}. */
spf_status spf_cmux_scattered_dev(spf_ctx *ctx, void *stream, size_t units, const void *const *d_ptrs);
/* dst row r = `words` u64 read from d_src_ptrs[r] (device array of device pointers) */
spf_status spf_gather_rows_dev(spf_ctx *ctx, void *stream, size_t rows, size_t words,
                               const uint64_t *const *d_src_ptrs, uint64_t *d_dst);

/* ---- measurement hooks (bench.py) ------------------------------------------------------- */

/* Average device time in milliseconds of the launches of one kernel family made through this
 * context since timing was enabled (or since the last query, which clears the record), measured
 * with hipEvents recorded on the launch stream around each launch: "pbs" (blind rotation),
 * "keyswitch" (digits + GEMM), "trace" (cbs_trace_kernel), "scheme_switch", "cmux" (contiguous
 * batched CMUX family).  A host-pointer bootstrap of more than one chip round is launched in slices
 * of 1024 ciphertexts: it records one "pbs" entry per slice.  Enable with spf_set_timing(ctx, 1). */
spf_status spf_set_timing(spf_ctx *ctx, int enabled);
spf_status spf_last_kernel_ms(spf_ctx *ctx, const char *kernel /* "pbs" | "keyswitch" | "trace" | "scheme_switch" | "cmux" */,
                              double *avg_ms, int *launches);
/* `generate_lut` (sunscreen_tfhe ops/bootstrapping/programmable_bootstrapping.rs:129-185) for
 * `programmable_bootstrap_univariate`: map_tables[f * 2^bits + x] = f(x) for n_maps functions over a
 * plaintext space of 2^bits values (the reference takes closures; a table is their C form).  Writes the trivial
 * GLWE (zero mask, body = the rotated, half-negated table polynomial), (k+1)*N words.  No GPU involved; ctx-free.
 * A value >= 2^bits is SPF_ERR_INVALID_ARGUMENT (the reference asserts). */
spf_status spf_generate_lut(const spf_params *params, const uint64_t *map_tables, size_t n_maps,
                            uint32_t plaintext_bits, uint64_t *lut_glwe_out);
/* `safe_bincode::deserialize::<ComputeKey>` + upload (parasol_runtime/src/safe_bincode.rs:16-28,
 * crypto/keys.rs:294-318): `bytes` is what the Rust side wrote with bincode DefaultOptions +
 * with_fixint_encoding — four sequences (u64 LE count, elements) in the order bs_key, ks_key, ss_key, auto_key.
 * Every count is checked against the context's parameters before anything is loaded; trailing bytes are allowed. */
spf_status spf_load_compute_key_bincode(spf_ctx *ctx, const uint8_t *bytes, size_t len);
/* Ciphertext wire format (parasol_runtime/src/crypto/encryption.rs:23-110: L0LweCiphertext, L1LweCiphertext,
 * L1GlweCiphertext, L1GlevCiphertext are serde newtypes over entities with one field `data: AVec<Torus<u64>>`,
 * sunscreen_tfhe/src/dst.rs:25-41) as `safe_bincode::deserialize` reads it (safe_bincode.rs:16-28): bincode
 * DefaultOptions + fixint encoding = a u64 little-endian element count, then the words little-endian; the read
 * is bounded by GetSize = (count + 1) * 8 bytes (encryption.rs:454-519), `check_is_valid` then demands exactly
 * the length the parameters give, trailing bytes are allowed.  L1GgswCiphertext is not Serialize in the
 * reference (encryption.rs:93-97): SPF_VAL_GGSW1 is SPF_ERR_UNSUPPORTED.  Host only, no GPU, ctx-free.
 *   spf_ciphertext_words:        number of u64 words of a ciphertext of `kind` (0 for an unknown kind)
 *   spf_ciphertext_from_bincode: bytes -> words_out (caller-allocated, spf_ciphertext_words(kind) words).  A count
 *                                that differs from the parameters' (the reference's malformed-length vector
 *                                safe_bincode.rs:58-66 included) or a truncated body is SPF_ERR_INVALID_ARGUMENT and
 *                                leaves words_out untouched.
 *   spf_ciphertext_to_bincode:   words -> out (capacity `cap` bytes); *written = 8 + 8 * words. */
size_t spf_ciphertext_words(const spf_params *params, spf_value_kind kind);
spf_status spf_ciphertext_from_bincode(const spf_params *params, spf_value_kind kind, const uint8_t *bytes, size_t len,
                                       uint64_t *words_out);
spf_status spf_ciphertext_to_bincode(const spf_params *params, spf_value_kind kind, const uint64_t *words, uint8_t *out,
                                     size_t cap, size_t *written);
/* `Evaluation::l1ggsw_zero()` / `l1ggsw_one()` (crypto/evaluation.rs:254-262): the GGSW (cbs radix, FFT domain,
 * (k+1)*l_cbs*(k+1)*N/2 complex) that `Evaluation::new` obtains by circuit-bootstrapping the trivial L0 LWE of
 * the bit (:161-197).  Computed on first use after the keys were (re)loaded, cached in HBM. */
spf_status spf_l1ggsw_constant(spf_ctx *ctx, int bit, double *ggsw_fft_out);
/* Device buffers for a caller that chains the `_dev` entry points but has no HIP binding of its own (the Rust shim of
 * INTEGRATION.md): hipMalloc / hipFree / hipMemcpy on the context's GPU.  `spf_device_download` first waits for everything
 * enqueued on `stream` (the stream the producing `_dev` call was given; NULL = the default stream), then copies.  No
 * reference counterpart (the reference keeps every ciphertext in host memory, crypto/encryption.rs:143-165). */
spf_status spf_device_alloc(spf_ctx *ctx, size_t bytes, void **dev_ptr);
spf_status spf_device_free(spf_ctx *ctx, void *dev_ptr);
spf_status spf_device_upload(spf_ctx *ctx, void *dev_dst, const void *host_src, size_t bytes);
spf_status spf_device_download(spf_ctx *ctx, void *stream, void *host_dst, const void *dev_src, size_t bytes);
/* Name of the blind-rotation kernel the most recent bootstrap launch of this context used (the shape is
 * picked from the batch size: eight waves per ciphertext, two ciphertexts per workgroup, or the throughput
 * shape).  For measurement records; never NULL. */
const char *spf_last_blind_rotate_kernel(spf_ctx *ctx);
/* The same for the CMUX family (spf_cmux*, spf_glev_cmux*, spf_multiply_glwe_ggsw*, gate graphs): four waves per gate up
 * to one gate per CU, the streaming shape beyond, with streaming (non-temporal) selector loads once a launch's selectors
 * exceed the Infinity Cache.  Never NULL. */
const char *spf_last_cmux_kernel(spf_ctx *ctx);

/* ---- device groups: every GPU of a node from ONE host process (SURVEY.md §8 b / e) -------------------------------- *
 *
 * The caller being replaced is one process: `Evaluation` holds an `Arc<ComputeKey>` (crypto/evaluation.rs:144-197) and is
 * called from the rayon workers of one `CircuitProcessor` (circuit_processor/mod.rs:201-209).  A group is what that one
 * `Evaluation` owns on a multi-GPU node: one context per listed device (a device may be listed more than once: that many
 * contexts on it), the evaluation keys replicated INSIDE the library, and batch entry points that cut a host batch into
 * contiguous ranges of ceil(B / G) — bootstraps are independent units, there is no data-path collective — and drive every
 * member from its own host thread and stream.
 *
 * Keys: the loaders put the key on member 0 (host -> HBM) and replicate it: one RCCL communicator over the distinct
 * devices (`ncclCommInitAll`, single process) and an in-place `ncclBroadcast` of each key blob from member 0 over xGMI;
 * further members on an already served device take a device-to-device copy; every member then derives its own images
 * (keyswitch byte planes, scaled bootstrap key).  librccl.so is loaded on first use (dlopen); transport can be forced with
 * the environment variable SPF_GROUP_TRANSPORT = "rccl" (default whenever the group has more than one member; also taken
 * for a one-member group when set explicitly) or "peer" (hipMemcpyPeerAsync, no RCCL).  A transport NAMED there is never
 * replaced: a missing librccl.so with SPF_GROUP_TRANSPORT=rccl is SPF_ERR_HIP.  When rccl is only the default and librccl.so
 * cannot be loaded the group falls back to peer copies and says so (spf_group_replication_stats: transport
 * "peer (librccl.so could not be loaded)").
 *
 * Failures: a member whose call fails with SPF_ERR_HIP is taken out of rotation and its range is re-queued over the
 * remaining members ("a failed GPU's shard is re-queued by host"); the call fails only when no member is
 * left.  Any other status (invalid argument, missing key) is the caller's error and is returned at once.
 * `spf_group_set_member_enabled` drains or re-admits a device by hand.
 *
 * Thread safety: every entry point may be called from any number of host threads; calls are queued per member. */
typedef struct spf_group spf_group;
spf_status spf_group_create(const spf_params *params, const int *device_ids, int n_devices, spf_group **out);
void spf_group_destroy(spf_group *grp);
int spf_group_size(const spf_group *grp);
/* member i's context: for the `_dev` entry points, gate graphs and measurement hooks on that device.  Owned by the group. */
spf_ctx *spf_group_ctx(spf_group *grp, int member);
/* message of the last failing group call (storage of the calling thread, as spf_last_error) */
const char *spf_group_last_error(const spf_group *grp);

/* `ComputeKey` fields (crypto/keys.rs:306-318): upload to member 0, replicate, derive.  Same argument meaning as the
 * single-context loaders above. */
spf_status spf_group_load_bootstrap_key(spf_group *grp, const double *bsk_fft, size_t n_complex);
spf_status spf_group_load_keyswitch_key(spf_group *grp, const uint64_t *ksk, size_t n_words);
spf_status spf_group_load_automorphism_key(spf_group *grp, const double *ak_fft, size_t n_complex);
spf_status spf_group_load_scheme_switch_key(spf_group *grp, const double *ssk_fft, size_t n_complex);
spf_status spf_group_load_compute_key_bincode(spf_group *grp, const uint8_t *bytes, size_t len);
/* replicate whatever member 0 holds (keys put there through spf_group_ctx(grp, 0), e.g. generated on the device into
 * spf_key_blob + spf_key_blob_commit) */
spf_status spf_group_replicate_keys(spf_group *grp);
/* totals since the group was created: seconds on the wire (communicator set-up counted separately), bytes received per
 * member, ranks of the RCCL communicator (0 = RCCL not used), transport name ("rccl", "peer", "none"; never NULL) */
spf_status spf_group_replication_stats(spf_group *grp, double *wire_seconds, double *comm_init_seconds, size_t *bytes_per_member,
                                       int *rccl_world_size, const char **transport);
/* take a member out of rotation (enabled = 0) or back in (1; also clears a recorded failure) */
spf_status spf_group_set_member_enabled(spf_group *grp, int member, int enabled);
/* members in rotation now */
int spf_group_members_in_rotation(spf_group *grp);
/* testing hook: the next `count` calls dispatched to `member` fail with SPF_ERR_HIP before touching the device */
spf_status spf_group_debug_fail_next(spf_group *grp, int member, int count);

/* The host-pointer batch forms over the group: same arguments as spf_*_batch, results word-identical to one context. */
spf_status spf_group_keyswitch_lwe_l1_lwe_l0_batch(spf_group *grp, size_t B, const uint64_t *lwe1_in, uint64_t *lwe0_out);
spf_status spf_group_generalized_pbs_batch(spf_group *grp, size_t B, const uint64_t *lwe0_in, const uint64_t *lut_glwe,
                                           size_t lut_stride, uint32_t log_chi, uint32_t log_v, uint64_t body_rotate,
                                           uint64_t *glwe_out);
spf_status spf_group_pbs_univariate_batch(spf_group *grp, size_t B, const uint64_t *lwe0_in, const uint64_t *lut_glwe,
                                          size_t lut_stride, uint64_t *lwe1_out);
spf_status spf_group_circuit_bootstrap_pbs_batch(spf_group *grp, size_t B, const uint64_t *lwe0_in, uint64_t *glwe_out);
spf_status spf_group_circuit_bootstrap_batch(spf_group *grp, size_t B, const uint64_t *lwe0_in, double *ggsw_fft_out);
spf_status spf_group_mod_switch_trace_and_rotate_batch(spf_group *grp, size_t B, const uint64_t *glwe_in, uint64_t *glev_out);
spf_status spf_group_scheme_switch_batch(spf_group *grp, size_t B, const uint64_t *glev_in, double *ggsw_fft_out);
spf_status spf_group_sample_extract_l1_batch(spf_group *grp, size_t B, const uint64_t *glwe_in, size_t idx, uint64_t *lwe1_out);
spf_status spf_group_glwe_not_batch(spf_group *grp, size_t B, const uint64_t *glwe_in, uint64_t *glwe_out);
spf_status spf_group_glwe_xor_batch(spf_group *grp, size_t B, const uint64_t *a, const uint64_t *b, uint64_t *glwe_out);
spf_status spf_group_glwe_mul_xn_batch(spf_group *grp, size_t B, const uint64_t *glwe_in, size_t n, uint64_t *glwe_out);
spf_status spf_group_cmux_batch(spf_group *grp, size_t B, const double *sel_ggsw_fft, const uint64_t *a, const uint64_t *b,
                                uint64_t *out);
spf_status spf_group_glev_cmux_batch(spf_group *grp, size_t B, const double *sel_ggsw_fft, const uint64_t *a, const uint64_t *b,
                                     uint64_t *out);
spf_status spf_group_multiply_glwe_ggsw_batch(spf_group *grp, size_t B, const uint64_t *glwe, const double *ggsw_fft,
                                              uint64_t *out);
spf_status spf_group_gate_bootstrap_batch(spf_group *grp, size_t B, const uint64_t *lwe1_in, uint64_t *glwe_out);
spf_status spf_group_keyswitch_circuit_bootstrap_batch(spf_group *grp, size_t B, const uint64_t *lwe1_in, double *ggsw_fft_out);
/* `Evaluation::l1ggsw_zero` / `l1ggsw_one` (identical on every member: same keys, same kernels; taken from member 0) */
spf_status spf_group_l1ggsw_constant(spf_group *grp, int bit, double *ggsw_fft_out);

/* Gate-graph jobs over the group — `CircuitProcessor::run_graph_blocking` for a POOL of independent circuits on a multi-GPU node
 * (circuit_processor/mod.rs:573-623; the multiplier circuits of circuits/mul.rs:90-200; BASELINE config 5).  The unit dealt to a
 * device is the job (one graph): cutting one graph across GPUs would ship a 256 KiB GGSW per selector crossing the cut,
 * independent graphs need nothing.
 *   spf_group_graph_create: a graph that is not bound to a device yet; built with spf_graph_add_* as usual (same validation).
 *   spf_group_run_graphs:   deals the n jobs over the members in rotation by cost, longest processing time first (cost =
 *       50 x circuit bootstraps + other operations of the job; ties by position: deterministic), lowers every member's jobs
 *       into ONE graph on its device — the level-batching executor then sees K jobs x gates per level — runs the members side
 *       by side on their worker threads and returns when every output of every job has been written.  The merged graphs are
 *       kept while the same jobs come again (new input contents, same DAGs).  A member that fails with SPF_ERR_HIP leaves the
 *       rotation and its jobs are dealt again over the others.  Results are word-identical to spf_graph_run of each job on one
 *       context.  spf_graph_run / spf_graph_stats / spf_graph_destroy work on such a graph (run = a pool of one; stats = of
 *       the job as built: levels and launches are those of the merged graph it last ran in).
 *   spf_graph_member:       the member a group's graph last ran on (-1 before its first run; 0 for an ordinary graph). */
spf_status spf_group_graph_create(spf_group *grp, spf_graph **out);
spf_status spf_group_run_graphs(spf_group *grp, spf_graph *const *graphs, size_t n);
int spf_graph_member(const spf_graph *graph);

/* Call coalescing over the group: one pool per member; a calling thread is dealt to a member on its first submit
 * (round-robin over the members in rotation) and stays there, so that the callers of one device keep coming back to the
 * same batch.  The handle is used with the spf_pool_submit_*, spf_pool_wait, spf_pool_stats (sums), spf_pool_set_max_inflight
 * (per member) and spf_pool_destroy entry points above. */
spf_status spf_pool_create_group(spf_group *grp, size_t max_batch, uint32_t max_wait_us, spf_pool **out);

/* Library / kernel build information: version, target, the compile-time options of the blind rotation — and "ABLATION(n: timing
 * only, results are WRONG)" right after the target when the library is a timing-only ablation build (-DSPF_ABL=n). */
const char *spf_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SPF_HIP_H */
