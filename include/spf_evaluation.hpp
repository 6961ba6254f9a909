// spf_evaluation.hpp — C++ host-side mirror of `parasol_runtime::Evaluation`
// (parasol_runtime/src/crypto/evaluation.rs:144-266) over the C ABI in spf_hip.h.
//
// Same method names, argument order and ownership as the reference: the caller allocates the
// output and passes it first (`&mut` there, pointer/span here); methods never return data.
// Where the reference panics (size assertions, `assert_is_valid`) this throws
// spf::Error — a C++-level convenience; nothing is thrown across the C ABI itself.
// Every method also has a batch form (leading size_t B): the engine is batch-native and a
// single ciphertext is simply B = 1.
#pragma once
#include "spf_hip.h"

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

namespace spf {

struct Error : std::runtime_error {
    spf_status status;
    Error(spf_status s, const std::string& m) : std::runtime_error(m), status(s) {}
};

// `ComputeKey` (crypto/keys.rs:306-318): borrowed host arrays in the reference layouts.
struct ComputeKey {
    const double* bs_key;   // BootstrapKeyFft<Complex<f64>>, interleaved re/im
    size_t bs_key_complex;
    const uint64_t* ks_key; // LweKeyswitchKey<u64>
    size_t ks_key_words;
    const double* auto_key = nullptr; // AutomorphismKeyFft<Complex<f64>>
    size_t auto_key_complex = 0;
    const double* ss_key = nullptr;   // SchemeSwitchKeyFft<Complex<f64>>
    size_t ss_key_complex = 0;
};

// One `Evaluation` per process, shared by every worker thread, as in the reference — and it owns EVERY listed GPU: a device
// group (spf_group_*) with the keys replicated inside the library; a batch is cut over the devices, B = 1 lands on the first one.
class Evaluation {
  public:
    // Evaluation::new (evaluation.rs:161-197) over the listed devices (a device may repeat: that many contexts on it)
    Evaluation(const ComputeKey& key, const spf_params& params, const std::vector<int>& devices) : params_(params)
    {
        check_create(spf_group_create(&params, devices.data(), (int)devices.size(), &grp_));
        ctx_ = spf_group_ctx(grp_, 0);
        try { // a throw from here on would skip the destructor: release the group (keys in HBM, streams, threads) first
            check(spf_group_load_bootstrap_key(grp_, key.bs_key, key.bs_key_complex));
            if (key.ks_key) check(spf_group_load_keyswitch_key(grp_, key.ks_key, key.ks_key_words));
            if (key.auto_key) check(spf_group_load_automorphism_key(grp_, key.auto_key, key.auto_key_complex));
            if (key.ss_key) check(spf_group_load_scheme_switch_key(grp_, key.ss_key, key.ss_key_complex));
        } catch (...) {
            spf_group_destroy(grp_);
            grp_ = nullptr;
            throw;
        }
    }
    Evaluation(const ComputeKey& key, const spf_params& params, int device = 0) : Evaluation(key, params, std::vector<int>{device}) {}
    // Evaluation::with_default_params (evaluation.rs:200-204)
    static Evaluation with_default_params(const ComputeKey& key, int device = 0)
    {
        spf_params p;
        spf_default_params(&p);
        return Evaluation(key, p, device);
    }
    Evaluation(Evaluation&& o) noexcept : grp_(o.grp_), ctx_(o.ctx_), params_(o.params_) { o.grp_ = nullptr; o.ctx_ = nullptr; }
    Evaluation(const Evaluation&) = delete;
    Evaluation& operator=(const Evaluation&) = delete;
    ~Evaluation() { spf_group_destroy(grp_); }
    int devices() const { return spf_group_size(grp_); }
    spf_group* group() const { return grp_; }

    // Evaluation::l1ggsw_zero / l1ggsw_one (:254-262): the circuit bootstraps of the trivial L0 LWE of 0 / 1 that
    // Evaluation::new precomputes (:161-197); here made on the GPU at first use and cached in HBM per key set
    void l1ggsw_zero(double* ggsw_fft_out) const { check(spf_group_l1ggsw_constant(grp_, 0, ggsw_fft_out)); }
    void l1ggsw_one(double* ggsw_fft_out) const { check(spf_group_l1ggsw_constant(grp_, 1, ggsw_fft_out)); }

    const spf_params& params() const { return params_; }
    spf_ctx* raw() const { return ctx_; }

    // KeylessEvaluation::not(&mut L1GlweCiphertext, &L1GlweCiphertext) (:48); `not`/`xor` are C++ tokens
    void not_(uint64_t* output, const uint64_t* input, size_t B = 1)
    {
        check(spf_group_glwe_not_batch(grp_, B, input, output));
    }
    // KeylessEvaluation::xor(&mut L1GlweCiphertext, a, b) (:53)
    void xor_(uint64_t* output, const uint64_t* a, const uint64_t* b, size_t B = 1)
    {
        check(spf_group_glwe_xor_batch(grp_, B, a, b, output));
    }
    // KeylessEvaluation::mul_xn(&mut L1GlweCiphertext, &L1GlweCiphertext, n) (:58)
    void mul_xn(uint64_t* output, const uint64_t* input, size_t n, size_t B = 1)
    {
        check(spf_group_glwe_mul_xn_batch(grp_, B, input, n, output));
    }
    // KeylessEvaluation::sample_extract_l1(&mut L1LweCiphertext, &L1GlweCiphertext, idx) (:126)
    void sample_extract_l1(uint64_t* output, const uint64_t* input, size_t idx, size_t B = 1)
    {
        check(spf_group_sample_extract_l1_batch(grp_, B, input, idx, output));
    }
    // Evaluation::keyswitch_lwe_l1_lwe_l0(&mut L0LweCiphertext, &L1LweCiphertext) (:246)
    void keyswitch_lwe_l1_lwe_l0(uint64_t* output, const uint64_t* input, size_t B = 1)
    {
        check(spf_group_keyswitch_lwe_l1_lwe_l0_batch(grp_, B, input, output));
    }
    // Evaluation::circuit_bootstrap(&mut L1GgswCiphertext, &L0LweCiphertext) (:211)
    void circuit_bootstrap(double* output_ggsw_fft, const uint64_t* input_l0, size_t B = 1)
    {
        check(spf_group_circuit_bootstrap_batch(grp_, B, input_l0, output_ggsw_fft));
    }
    // Evaluation::scheme_switch(&mut L1GgswCiphertext, &L1GlevCiphertext) (:231)
    void scheme_switch(double* output_ggsw_fft, const uint64_t* input_glev, size_t B = 1)
    {
        check(spf_group_scheme_switch_batch(grp_, B, input_glev, output_ggsw_fft));
    }
    // bootstrap stage of Evaluation::circuit_bootstrap (:211) = hi_noise_lwe_to_lo_noise_glwe
    void circuit_bootstrap_pbs(uint64_t* output_glwe, const uint64_t* input_l0, size_t B = 1)
    {
        check(spf_group_circuit_bootstrap_pbs_batch(grp_, B, input_l0, output_glwe));
    }
    // sunscreen_tfhe::ops::bootstrapping::programmable_bootstrap_univariate
    void programmable_bootstrap_univariate(uint64_t* output_l1, const uint64_t* input_l0,
                                           const uint64_t* lut_glwe, size_t B = 1, size_t lut_stride = 0)
    {
        check(spf_group_pbs_univariate_batch(grp_, B, input_l0, lut_glwe, lut_stride, output_l1));
    }
    // sunscreen_tfhe::ops::bootstrapping::generalized_programmable_bootstrap
    void generalized_programmable_bootstrap(uint64_t* output_glwe, const uint64_t* input_l0,
                                            const uint64_t* lut_glwe, uint32_t log_chi, uint32_t log_v,
                                            size_t B = 1, size_t lut_stride = 0)
    {
        check(spf_group_generalized_pbs_batch(grp_, B, input_l0, lut_glwe, lut_stride, log_chi, log_v, 0, output_glwe));
    }
    // KeylessEvaluation::cmux(&mut L1GlweCiphertext, &L1GgswCiphertext, a, b) (:68)
    void cmux(uint64_t* output, const double* sel_ggsw_fft, const uint64_t* a, const uint64_t* b, size_t B = 1)
    {
        check(spf_group_cmux_batch(grp_, B, sel_ggsw_fft, a, b, output));
    }
    // FheOp::KeyswitchL1toL0 -> FheOp::CircuitBootstrap, fused on device
    void gate_bootstrap(uint64_t* output_glwe, const uint64_t* input_l1, size_t B = 1)
    {
        check(spf_group_gate_bootstrap_batch(grp_, B, input_l1, output_glwe));
    }

    // KeylessEvaluation::glev_cmux (:86), multiply_glwe_ggsw (:104)
    void glev_cmux(uint64_t* output, const double* sel_ggsw_fft, const uint64_t* a, const uint64_t* b, size_t B = 1)
    {
        check(spf_group_glev_cmux_batch(grp_, B, sel_ggsw_fft, a, b, output));
    }
    void multiply_glwe_ggsw(uint64_t* output, const uint64_t* glwe, const double* ggsw_fft, size_t B = 1)
    {
        check(spf_group_multiply_glwe_ggsw_batch(grp_, B, glwe, ggsw_fft, output));
    }

  private:
    friend class FheCircuit;
    void check(spf_status s) const
    {
        if (s != SPF_OK) throw Error(s, spf_group_last_error(grp_));
    }
    static void check_create(spf_status s)
    {
        if (s != SPF_OK) throw Error(s, spf_last_error(nullptr));
    }
    static void check(spf_status s, const spf_ctx* c)
    {
        if (s != SPF_OK) throw Error(s, spf_last_error(c));
    }
    spf_group* grp_ = nullptr;
    spf_ctx* ctx_ = nullptr; // member 0: gate graphs and the device-pointer forms address one device
    spf_params params_;
};

// `FheCircuit` + `CircuitProcessor::run_graph_blocking` (parasol_runtime/src/fhe_circuit.rs:34-205,
// circuit_processor/mod.rs:573-623) over spf_graph_*: build the DAG node by node, run() executes it
// level by level with every intermediate in HBM.  Node handles are the library's dense ids.
class FheCircuit {
  public:
    using Node = uint32_t;
    // a circuit of the Evaluation's FIRST device ...
    explicit FheCircuit(const Evaluation& ev) : ctx_(ev.raw())
    {
        Evaluation::check(spf_graph_create(ctx_, &g_), ctx_);
    }
    // ... or a JOB of the whole Evaluation (every GPU it owns): placed on a device when it is run — `run_all` deals a pool of
    // jobs over the devices by cost (spf_group_run_graphs; the reference: one CircuitProcessor fed whole FheCircuits,
    // circuit_processor/mod.rs:573-623)
    struct Job {};
    FheCircuit(const Evaluation& ev, Job) : ctx_(ev.raw()), grp_(ev.group())
    {
        if (spf_group_graph_create(grp_, &g_) != SPF_OK) throw Error(SPF_ERR_HIP, spf_group_last_error(grp_));
    }
    static void run_all(const Evaluation& ev, const std::vector<FheCircuit*>& jobs)
    {
        std::vector<spf_graph*> raw;
        for (FheCircuit* j : jobs) raw.push_back(j->g_);
        const spf_status s = spf_group_run_graphs(ev.group(), raw.data(), raw.size());
        if (s != SPF_OK) throw Error(s, spf_group_last_error(ev.group()));
    }
    int device_member() const { return spf_graph_member(g_); }
    FheCircuit(const FheCircuit&) = delete;
    FheCircuit& operator=(const FheCircuit&) = delete;
    ~FheCircuit() { spf_graph_destroy(g_); }

    // FheOp::Input*: `host` is read at every run()
    Node input(spf_value_kind kind, const void* host)
    {
        Node n = 0;
        Evaluation::check(spf_graph_add_input(g_, kind, host, &n), ctx_);
        return n;
    }
    // FheOp::{Zero,One}{Lwe0,Glwe1}
    Node trivial(spf_value_kind kind, uint64_t bit)
    {
        Node n = 0;
        Evaluation::check(spf_graph_add_trivial(g_, kind, bit, &n), ctx_);
        return n;
    }
    Node op(spf_graph_op o, std::initializer_list<Node> operands, uint64_t param = 0)
    {
        Node n = 0;
        Evaluation::check(spf_graph_add_op(g_, o, operands.begin(), operands.size(), param, &n), ctx_);
        return n;
    }
    // FheOp::Output*: `host` is written by every run()
    void output(Node node, void* host) { Evaluation::check(spf_graph_add_output(g_, node, host), ctx_); }
    void run() { Evaluation::check(spf_graph_run(g_), ctx_); }

  private:
    spf_ctx* ctx_ = nullptr;
    spf_group* grp_ = nullptr;
    spf_graph* g_ = nullptr;
};

// ---- the per-operation boundary with the ciphertexts in HBM (spf_value_*, spf_pool_submit_*_v) -----------------------------
//
// The reference's ciphertext types (parasol_runtime/src/crypto/encryption.rs:143-165: L0LweCiphertext, L1LweCiphertext,
// L1GlweCiphertext, L1GlevCiphertext, L1GgswCiphertext) as typed, move-only owners of ONE device-resident value: what a Rust shim
// keeps as `Option<SpfValue>` inside those types.  A wrong operand type is a compile error here, as it is
// there; the C ABI below refuses it at run time.
template <spf_value_kind K> class DeviceCiphertext {
  public:
    static constexpr spf_value_kind kind = K;
    DeviceCiphertext() = default; // empty: an output that has not been written (the reference allocates with `::allocate`)
    DeviceCiphertext(DeviceCiphertext&& o) noexcept : v_(o.v_) { o.v_ = nullptr; }
    DeviceCiphertext& operator=(DeviceCiphertext&& o) noexcept
    {
        if (this != &o) {
            reset();
            v_ = o.v_;
            o.v_ = nullptr;
        }
        return *this;
    }
    DeviceCiphertext(const DeviceCiphertext&) = delete;
    DeviceCiphertext& operator=(const DeviceCiphertext&) = delete;
    ~DeviceCiphertext() { reset(); }
    // `Clone`: another owner of the same immutable value (a reference count, no copy)
    DeviceCiphertext share() const
    {
        DeviceCiphertext c;
        if (v_ && spf_value_retain(v_) == SPF_OK) c.v_ = v_;
        return c;
    }
    explicit operator bool() const { return v_ != nullptr; }
    spf_value* raw() const { return v_; }
    // until the operation that produces the value has run (at once for a valid value; PooledEvaluation in pushed mode hands out
    // ciphertexts that are still pending)
    void wait() const
    {
        const spf_status s = v_ ? spf_value_wait(v_) : SPF_ERR_INVALID_ARGUMENT;
        if (s != SPF_OK) throw Error(s, "DeviceCiphertext::wait: the producing operation failed (or there is no value)");
    }
    // HBM -> host, the reference layout (u64 words; the GGSW: complex f64); waits for a pending value first
    void download(void* host) const
    {
        wait();
        const spf_status s = spf_value_download(v_, host);
        if (s != SPF_OK) throw Error(s, "DeviceCiphertext::download: no valid value");
    }
    void reset()
    {
        if (v_) spf_value_release(v_);
        v_ = nullptr;
    }

  private:
    friend class PooledEvaluation;
    spf_value* v_ = nullptr;
};
using L0LweCiphertext = DeviceCiphertext<SPF_VAL_LWE0>;
using L1LweCiphertext = DeviceCiphertext<SPF_VAL_LWE1>;
using L1GlweCiphertext = DeviceCiphertext<SPF_VAL_GLWE1>;
using L1GlevCiphertext = DeviceCiphertext<SPF_VAL_GLEV1>;
using L1GgswCiphertext = DeviceCiphertext<SPF_VAL_GGSW1>;

// `Evaluation` as `CircuitProcessor`'s workers call it (circuit_processor/mod.rs:255-540): one operation on one ciphertext per
// call, from any number of threads, output first — over a pool of the Evaluation's group, so that the concurrent calls of many
// workers become one launch, with operands and results in HBM.  Each method returns when its output is valid, like the reference's
// — or, in PUSHED mode, at once: the output is a pending ciphertext that later calls take as an operand right away (the pool
// orders and batches what has been pushed by kind and level: spf_hip.h, "Deferred operands"); only `wait()` / `download()` of a
// ciphertext block.  One thread can push a whole circuit that way.
class PooledEvaluation {
  public:
    enum class Mode { Blocking, Pushed };
    explicit PooledEvaluation(const Evaluation& ev, size_t max_batch = 1024, uint32_t max_wait_us = 100, Mode mode = Mode::Blocking)
        : grp_(ev.group()), pushed_(mode == Mode::Pushed)
    {
        if (spf_pool_create_group(grp_, max_batch, max_wait_us, &pool_) != SPF_OK) throw Error(SPF_ERR_HIP, spf_group_last_error(grp_));
    }
    PooledEvaluation(const PooledEvaluation&) = delete;
    PooledEvaluation& operator=(const PooledEvaluation&) = delete;
    ~PooledEvaluation() { spf_pool_destroy(pool_); }
    spf_pool* raw() const { return pool_; }
    // pushed mode: what has been pushed so far is launched now (e.g. once the conversions at the head of a circuit are in: they
    // run under the rest of the push); nothing is waited for
    void flush() const { check(spf_pool_flush(pool_)); }

    // `encrypt` / `trivial_*` land here: host words -> a device ciphertext on the calling thread's member
    template <class C> C upload(const void* host) const
    {
        C c;
        check(spf_value_upload(pool_, -1, C::kind, host, &c.v_));
        return c;
    }
    template <class C> C trivial(uint64_t bit) const
    {
        C c;
        check(spf_value_trivial(pool_, -1, C::kind, bit, &c.v_));
        return c;
    }

    // KeylessEvaluation (crypto/evaluation.rs:47-133)
    void not_(L1GlweCiphertext& output, const L1GlweCiphertext& input) const { run(SPF_OP_NOT, output, {input.raw()}); }
    void xor_(L1GlweCiphertext& output, const L1GlweCiphertext& a, const L1GlweCiphertext& b) const
    {
        run(SPF_OP_GLWE_ADD, output, {a.raw(), b.raw()});
    }
    void mul_xn(L1GlweCiphertext& output, const L1GlweCiphertext& input, size_t n) const { run(SPF_OP_MUL_XN, output, {input.raw()}, n); }
    void cmux(L1GlweCiphertext& output, const L1GgswCiphertext& sel, const L1GlweCiphertext& a, const L1GlweCiphertext& b) const
    {
        run(SPF_OP_CMUX, output, {sel.raw(), a.raw(), b.raw()});
    }
    void glev_cmux(L1GlevCiphertext& output, const L1GgswCiphertext& sel, const L1GlevCiphertext& a, const L1GlevCiphertext& b) const
    {
        run(SPF_OP_GLEV_CMUX, output, {sel.raw(), a.raw(), b.raw()});
    }
    void multiply_glwe_ggsw(L1GlweCiphertext& output, const L1GlweCiphertext& glwe, const L1GgswCiphertext& ggsw) const
    {
        run(SPF_OP_MULTIPLY_GGSW_GLWE, output, {ggsw.raw(), glwe.raw()});
    }
    void sample_extract_l1(L1LweCiphertext& output, const L1GlweCiphertext& input, size_t idx) const
    {
        run(SPF_OP_SAMPLE_EXTRACT, output, {input.raw()}, idx);
    }
    // Evaluation (crypto/evaluation.rs:211-251)
    void keyswitch_lwe_l1_lwe_l0(L0LweCiphertext& output, const L1LweCiphertext& input) const
    {
        run(SPF_OP_KEYSWITCH_L1_TO_L0, output, {input.raw()});
    }
    void circuit_bootstrap(L1GgswCiphertext& output, const L0LweCiphertext& input) const
    {
        run(SPF_OP_CIRCUIT_BOOTSTRAP, output, {input.raw()});
    }
    void scheme_switch(L1GgswCiphertext& output, const L1GlevCiphertext& input) const { run(SPF_OP_SCHEME_SWITCH, output, {input.raw()}); }
    // FheOp::KeyswitchL1toL0 -> FheOp::CircuitBootstrap as one operation (the L0 ciphertext never exists outside the launch)
    void keyswitch_circuit_bootstrap(L1GgswCiphertext& output, const L1LweCiphertext& input) const
    {
        spf_value* out = nullptr;
        uint64_t t = 0;
        check(spf_pool_submit_keyswitch_circuit_bootstrap_v(pool_, input.raw(), &out, pushed_ ? nullptr : &t));
        finish(output, out, t);
    }

  private:
    template <class C> void run(spf_graph_op op, C& output, std::initializer_list<const spf_value*> in, uint64_t param = 0) const
    {
        spf_value* out = nullptr;
        uint64_t t = 0;
        check(spf_pool_submit_op_v(pool_, op, in.begin(), in.size(), param, &out, pushed_ ? nullptr : &t));
        finish(output, out, t);
    }
    template <class C> void finish(C& output, spf_value* out, uint64_t ticket) const
    {
        if (!pushed_) {
            const spf_status s = spf_pool_wait(pool_, ticket);
            if (s != SPF_OK) {
                spf_value_release(out); // never valid: released in every case
                check(s);
            }
        }
        output.reset();
        output.v_ = out;
    }
    void check(spf_status s) const // (the pool leaves its messages with the first member's context)
    {
        if (s != SPF_OK) throw Error(s, spf_last_error(spf_group_ctx(grp_, 0)));
    }
    spf_group* grp_ = nullptr;
    spf_pool* pool_ = nullptr;
    bool pushed_ = false;
};

} // namespace spf
