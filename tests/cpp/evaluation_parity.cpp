// Native parity test of the C++ host mirror (include/spf_evaluation.hpp) — test infrastructure.
// Written the way a test next to parasol_runtime/src/crypto/evaluation.rs:277-337 (`can_lwe_keyswitch`,
// `can_circuit_bootstrap`, `can_cmux`) would read, except that the expectation is the CPU oracle's
// output word for word instead of a decryption.  Links libspf_hip.so (product) and
// libspf_oracle.so (checker); built and run by tests/test_gpu_cpp_host.py.
#include "spf_evaluation.hpp"

extern "C" {
#include "spf_oracle.h"
}

#include <complex>
#include <cstdio>
#include <cstring>
#include <memory>
#include <thread>
#include <vector>

namespace {

int failures = 0;
void expect(bool ok, const char* what)
{
    std::printf("%-58s %s\n", what, ok ? "ok" : "MISMATCH");
    if (!ok) failures++;
}
template <class T> bool same(const std::vector<T>& a, const std::vector<T>& b)
{
    return a.size() == b.size() && std::memcmp(a.data(), b.data(), a.size() * sizeof(T)) == 0;
}

} // namespace

int main()
{
    spf_params p;
    spf_default_params(&p);
    p.lwe_dimension = 10; // a short blind rotation keeps the oracle quick; every other parameter is DEFAULT_128
    const size_t n = p.lwe_dimension, N = p.polynomial_degree, k = p.glwe_size;
    const double lwe_std = 7.25e-5, glwe_std = 7e-16;
    const size_t ggsw_pbs = (k + 1) * p.pbs_radix_count * (k + 1) * (N / 2);
    const size_t ggsw_cbs = (k + 1) * p.cbs_radix_count * (k + 1) * (N / 2);
    size_t logn = 0;
    while (((size_t)1 << logn) < N) logn++;

    spfo_rng r;
    spfo_rng_seed(&r, 0xC0FFEE);
    std::vector<uint64_t> lwe_sk(n), glwe_sk(k * N);
    spfo_gen_binary_key(&r, lwe_sk.data(), n);
    spfo_gen_binary_key(&r, glwe_sk.data(), k * N);
    std::vector<spfo_c64> bsk(n * ggsw_pbs), ak(logn * k * p.tr_radix_count * (k + 1) * (N / 2)),
        ssk(p.ss_radix_count * (k + 1) * (N / 2));
    std::vector<uint64_t> ksk(k * N * p.ks_radix_count * (n + 1));
    spfo_gen_bsk_fft(&r, bsk.data(), lwe_sk.data(), n, glwe_sk.data(), N, k, p.pbs_radix_log, p.pbs_radix_count, glwe_std);
    spfo_gen_ksk(&r, ksk.data(), glwe_sk.data(), k * N, lwe_sk.data(), n, p.ks_radix_log, p.ks_radix_count, lwe_std);
    spfo_gen_auto_key_fft(&r, ak.data(), glwe_sk.data(), N, k, p.tr_radix_log, p.tr_radix_count, glwe_std);
    spfo_gen_ssk_fft(&r, ssk.data(), glwe_sk.data(), N, k, p.ss_radix_log, p.ss_radix_count, glwe_std);

    try {
        spf::ComputeKey key{reinterpret_cast<const double*>(bsk.data()), bsk.size(), ksk.data(), ksk.size(),
                            reinterpret_cast<const double*>(ak.data()), ak.size(),
                            reinterpret_cast<const double*>(ssk.data()), ssk.size()};
        spf::Evaluation ev(key, p, 0);

        // keyswitch_lwe_l1_lwe_l0 on an encryption of 1 under the GLWE key
        std::vector<uint64_t> l1(k * N + 1), l0(n + 1), l0_ref(n + 1);
        spfo_encrypt_lwe(&r, l1.data(), glwe_sk.data(), k * N, spfo_encode(1, 1), glwe_std);
        ev.keyswitch_lwe_l1_lwe_l0(l0.data(), l1.data());
        spfo_keyswitch_lwe(l0_ref.data(), l1.data(), ksk.data(), k * N, n, p.ks_radix_log, p.ks_radix_count);
        expect(same(l0, l0_ref), "Evaluation::keyswitch_lwe_l1_lwe_l0");

        // circuit_bootstrap of that L0 ciphertext
        std::vector<spfo_c64> sel(ggsw_cbs), sel_ref(ggsw_cbs);
        ev.circuit_bootstrap(reinterpret_cast<double*>(sel.data()), l0.data());
        spfo_circuit_bootstrap(sel_ref.data(), l0.data(), bsk.data(), ak.data(), ssk.data(), n, N, k, p.pbs_radix_log,
                               p.pbs_radix_count, p.tr_radix_log, p.tr_radix_count, p.ss_radix_log, p.ss_radix_count,
                               p.cbs_radix_log, p.cbs_radix_count);
        expect(std::memcmp(sel.data(), sel_ref.data(), sel.size() * sizeof(spfo_c64)) == 0, "Evaluation::circuit_bootstrap");

        // cmux with that selector, and the linear operations
        std::vector<uint64_t> a((k + 1) * N), b((k + 1) * N), out((k + 1) * N), ref((k + 1) * N), m(N, 0);
        m[0] = spfo_encode(1, 1);
        spfo_encrypt_glwe(&r, a.data(), glwe_sk.data(), m.data(), N, k, glwe_std);
        m[0] = 0; m[1] = spfo_encode(1, 1);
        spfo_encrypt_glwe(&r, b.data(), glwe_sk.data(), m.data(), N, k, glwe_std);
        ev.cmux(out.data(), reinterpret_cast<const double*>(sel.data()), a.data(), b.data());
        spfo_cmux(ref.data(), a.data(), b.data(), sel_ref.data(), N, k, p.cbs_radix_log, p.cbs_radix_count);
        expect(same(out, ref), "KeylessEvaluation::cmux");
        {
            std::vector<uint64_t> dec(N);
            spfo_decrypt_glwe_raw(dec.data(), out.data(), glwe_sk.data(), N, k);
            expect(spfo_decode(dec[0], 1) == 0 && spfo_decode(dec[1], 1) == 1, "  ... selector 1 picked b (decrypts)");
        }
        ev.not_(out.data(), a.data());
        spfo_glwe_not(ref.data(), a.data(), N, k);
        expect(same(out, ref), "KeylessEvaluation::not");
        ev.xor_(out.data(), a.data(), b.data());
        spfo_glwe_xor(ref.data(), a.data(), b.data(), N, k);
        expect(same(out, ref), "KeylessEvaluation::xor");
        ev.mul_xn(out.data(), a.data(), 2049);
        spfo_glwe_mul_xn(ref.data(), a.data(), 2049, N, k);
        expect(same(out, ref), "KeylessEvaluation::mul_xn");
        std::vector<uint64_t> se(k * N + 1), se_ref(k * N + 1);
        ev.sample_extract_l1(se.data(), a.data(), 7);
        spfo_sample_extract(se_ref.data(), a.data(), 7, N, k);
        expect(same(se, se_ref), "KeylessEvaluation::sample_extract_l1");

        // the same chain as one FheCircuit: glwe -> SampleExtract -> KeyswitchL1toL0 -> CircuitBootstrap -> CMux
        spf::FheCircuit g(ev);
        std::vector<uint64_t> bit((k + 1) * N), gout((k + 1) * N), gref((k + 1) * N);
        std::fill(m.begin(), m.end(), 0);
        m[0] = spfo_encode(1, 1);
        spfo_encrypt_glwe(&r, bit.data(), glwe_sk.data(), m.data(), N, k, glwe_std);
        auto x = g.input(SPF_VAL_GLWE1, bit.data());
        auto ia = g.input(SPF_VAL_GLWE1, a.data());
        auto ib = g.input(SPF_VAL_GLWE1, b.data());
        auto s = g.op(SPF_OP_CIRCUIT_BOOTSTRAP, {g.op(SPF_OP_KEYSWITCH_L1_TO_L0, {g.op(SPF_OP_SAMPLE_EXTRACT, {x}, 0)})});
        g.output(g.op(SPF_OP_CMUX, {s, ia, g.op(SPF_OP_NOT, {ib})}), gout.data());
        g.run();
        {
            std::vector<uint64_t> t1(k * N + 1), t0(n + 1), nb((k + 1) * N);
            std::vector<spfo_c64> ts(ggsw_cbs);
            spfo_sample_extract(t1.data(), bit.data(), 0, N, k);
            spfo_keyswitch_lwe(t0.data(), t1.data(), ksk.data(), k * N, n, p.ks_radix_log, p.ks_radix_count);
            spfo_circuit_bootstrap(ts.data(), t0.data(), bsk.data(), ak.data(), ssk.data(), n, N, k, p.pbs_radix_log,
                                   p.pbs_radix_count, p.tr_radix_log, p.tr_radix_count, p.ss_radix_log,
                                   p.ss_radix_count, p.cbs_radix_log, p.cbs_radix_count);
            spfo_glwe_not(nb.data(), b.data(), N, k);
            spfo_cmux(gref.data(), a.data(), nb.data(), ts.data(), N, k, p.cbs_radix_log, p.cbs_radix_count);
        }
        expect(same(gout, gref), "FheCircuit: SE -> KS -> CBS -> CMux(sel, a, Not(b))");

        // one Evaluation over a device group (two contexts on GPU 0: key replication inside the library, the batch cut in two)
        {
            spf::Evaluation ev2(key, p, std::vector<int>{0, 0});
            const size_t B = 5;
            std::vector<uint64_t> l1b(B * (k * N + 1)), g1(B * (k + 1) * N), g2(B * (k + 1) * N);
            for (size_t i = 0; i < B; i++)
                spfo_encrypt_lwe(&r, l1b.data() + i * (k * N + 1), glwe_sk.data(), k * N, spfo_encode(i & 1, 1), glwe_std);
            ev.gate_bootstrap(g1.data(), l1b.data(), B);
            ev2.gate_bootstrap(g2.data(), l1b.data(), B);
            expect(ev2.devices() == 2 && same(g1, g2), "Evaluation over a device group [0, 0]: same words as one device");

            // gate-graph jobs over the group: four copies of the chain above as four jobs, dealt 2 + 2, same words as the one-device graph
            std::vector<std::vector<uint64_t>> jout(4, std::vector<uint64_t>((k + 1) * N));
            std::vector<std::unique_ptr<spf::FheCircuit>> jobs;
            std::vector<spf::FheCircuit*> raw;
            for (int j = 0; j < 4; j++) {
                jobs.emplace_back(new spf::FheCircuit(ev2, spf::FheCircuit::Job{}));
                spf::FheCircuit& q = *jobs.back();
                auto jx = q.input(SPF_VAL_GLWE1, bit.data());
                auto ja = q.input(SPF_VAL_GLWE1, a.data());
                auto jb = q.input(SPF_VAL_GLWE1, b.data());
                auto js = q.op(SPF_OP_CIRCUIT_BOOTSTRAP, {q.op(SPF_OP_KEYSWITCH_L1_TO_L0, {q.op(SPF_OP_SAMPLE_EXTRACT, {jx}, 0)})});
                q.output(q.op(SPF_OP_CMUX, {js, ja, q.op(SPF_OP_NOT, {jb})}), jout[(size_t)j].data());
                raw.push_back(&q);
            }
            spf::FheCircuit::run_all(ev2, raw);
            int on0 = 0, on1 = 0;
            bool all_same = true;
            for (int j = 0; j < 4; j++) {
                on0 += raw[(size_t)j]->device_member() == 0;
                on1 += raw[(size_t)j]->device_member() == 1;
                all_same = all_same && same(jout[(size_t)j], gref);
            }
            expect(on0 == 2 && on1 == 2 && all_same, "FheCircuit jobs over the group: dealt 2 + 2, oracle's words");
        }

        // the same chain operation by operation, the ciphertexts in HBM: twelve worker threads call PooledEvaluation the way
        // CircuitProcessor's rayon workers call Evaluation (circuit_processor/mod.rs:255-540), one ciphertext per call
        {
            spf::PooledEvaluation pe(ev, 64, 200);
            const int T = 12;
            std::vector<std::vector<uint64_t>> tout((size_t)T, std::vector<uint64_t>((k + 1) * N));
            std::vector<int> ok((size_t)T, 0);
            std::vector<std::thread> th;
            for (int t = 0; t < T; t++)
                th.emplace_back([&, t] {
                    try {
                        auto x = pe.upload<spf::L1GlweCiphertext>(bit.data());
                        auto va = pe.upload<spf::L1GlweCiphertext>(a.data());
                        auto vb = pe.upload<spf::L1GlweCiphertext>(b.data());
                        spf::L1LweCiphertext e1;
                        spf::L0LweCiphertext e0;
                        spf::L1GgswCiphertext sel_v, sel_fused;
                        spf::L1GlweCiphertext nb, o, o2;
                        pe.sample_extract_l1(e1, x, 0);
                        pe.keyswitch_lwe_l1_lwe_l0(e0, e1);
                        pe.circuit_bootstrap(sel_v, e0);
                        pe.not_(nb, vb);
                        pe.cmux(o, sel_v, va, nb);
                        o.download(tout[(size_t)t].data());
                        // the fused key switch + circuit bootstrap gives the same selector
                        pe.keyswitch_circuit_bootstrap(sel_fused, e1);
                        pe.cmux(o2, sel_fused, va, nb);
                        std::vector<uint64_t> again((k + 1) * N);
                        o2.download(again.data());
                        ok[(size_t)t] = again == tout[(size_t)t];
                    } catch (const spf::Error&) {
                        ok[(size_t)t] = 0;
                    }
                });
            for (auto& t : th) t.join();
            bool all = true;
            for (int t = 0; t < T; t++) all = all && ok[(size_t)t] && same(tout[(size_t)t], gref);
            expect(all, "PooledEvaluation by handle, 12 threads: the chain's oracle words");
            size_t live = 1, live_bytes = 1, cached = 0;
            spf_pool_value_stats(pe.raw(), &live, &live_bytes, &cached);
            expect(live == 0 && live_bytes == 0 && cached > 0, "  ... every value released: nothing live, the blocks cached");
            // an output that was never written is refused (the reference cannot express it; the C ABI reports it)
            bool refused = false;
            try {
                spf::L1GlweCiphertext none, o;
                pe.not_(o, none);
            } catch (const spf::Error& e) { refused = e.status == SPF_ERR_INVALID_ARGUMENT; }
            expect(refused, "  ... an empty operand is SPF_ERR_INVALID_ARGUMENT");
        }

        // ... and PUSHED from this one thread: no call blocks, the outputs are pending ciphertexts that the next call takes as
        // operands; only the download waits.  Six independent chains, one launch per kind and level.
        {
            spf::PooledEvaluation pe(ev, 64, 100000, spf::PooledEvaluation::Mode::Pushed);
            const int C = 6;
            std::vector<spf::L1GlweCiphertext> outs((size_t)C);
            spf_pool_counters c0{}, c1{};
            spf_pool_counters_get(pe.raw(), &c0);
            {
                auto x = pe.upload<spf::L1GlweCiphertext>(bit.data());
                auto va = pe.upload<spf::L1GlweCiphertext>(a.data());
                auto vb = pe.upload<spf::L1GlweCiphertext>(b.data());
                for (int c = 0; c < C; c++) {
                    spf::L1LweCiphertext e1;
                    spf::L0LweCiphertext e0;
                    spf::L1GgswCiphertext sel_v;
                    spf::L1GlweCiphertext nb;
                    pe.sample_extract_l1(e1, x, 0);
                    pe.keyswitch_lwe_l1_lwe_l0(e0, e1);
                    pe.circuit_bootstrap(sel_v, e0);
                    pe.not_(nb, vb);
                    pe.cmux(outs[(size_t)c], sel_v, va, nb);
                } // (the intermediates are released here, still pending: the pool keeps what it needs)
            }
            bool all = true;
            for (int c = 0; c < C; c++) {
                std::vector<uint64_t> got((k + 1) * N);
                outs[(size_t)c].download(got.data());
                all = all && same(got, gref);
            }
            spf_pool_counters_get(pe.raw(), &c1);
            expect(all, "PooledEvaluation pushed from one thread: the chain's oracle words");
            expect(c1.handle_ops - c0.handle_ops == 5u * C && c1.handle_launches - c0.handle_launches == 5, "  ... one launch per kind and level (5 for 30 operations)");
        }

        // malformed graph: wrong operand type must throw when the node is added (task.rs:26-31)
        bool threw = false;
        try { g.op(SPF_OP_CIRCUIT_BOOTSTRAP, {ia}); } catch (const spf::Error&) { threw = true; }
        expect(threw, "FheCircuit rejects a GLWE operand for CircuitBootstrap");
    } catch (const spf::Error& e) {
        std::printf("spf::Error: %s\n", e.what());
        return 2;
    }
    std::printf("%s\n", failures ? "FAILED" : "all equal");
    return failures ? 1 : 0;
}
