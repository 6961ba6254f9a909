"""The generic kernel family (spf_amd/csrc/spf_generic.hpp): any power-of-two N in 16 .. 1024, any k, any radix.

The reference's functions are generic (`generalized_programmable_bootstrap`, programmable_bootstrapping.rs:342-410; `cmux`,
fft_ops.rs:149-181) and its own functional tests run at small parameters — TEST_GLWE_DEF_1 = (N 128, k 2), TEST_LWE_DEF_1 =
n 128, TEST_RADIX = 3 x 4 bits (sunscreen_tfhe/src/high_level.rs:9-58).  Here those tests are REPLAYED on the HIP path
(encrypt with the oracle's keygen, compute on the GPU through the C ABI, decrypt, compare with the plaintext expectation —
`can_cmux_fft`, `can_fft_external_product_glwe_ggsw` fft_ops.rs:537-619; `bootstrap_helper` programmable_bootstrapping.rs:708-779)
and every output is also compared word for word with the oracle.
"""
import numpy as np
import pytest

import oracle as O
import spf_amd
from tests.util import random_glwe, random_lwe_batch, to_engine_params

pytestmark = pytest.mark.gpu

# high_level.rs:9-58
TEST1 = O.DEFAULT_128.replace(lwe_n=128, lwe_std=1e-16, N=128, k=2, glwe_std=1e-16, pbs_radix_log=4, pbs_count=3,
                              cbs_radix_log=4, cbs_count=3, ks_radix_log=4, ks_count=3)
# TEST_GLWE_DEF_2 / TEST_LWE_DEF_2
TEST2 = O.DEFAULT_128.replace(lwe_n=20, lwe_std=1e-16, N=256, k=3, glwe_std=1e-16, pbs_radix_log=4, pbs_count=3,
                              cbs_radix_log=8, cbs_count=2, ks_radix_log=4, ks_count=3)
# GLWE_1_1024_128 (sunscreen_tfhe/src/params.rs:248-255) with a short LWE side
P1024 = O.DEFAULT_128.replace(lwe_n=12, N=1024, k=1, glwe_std=7.2e-8, pbs_radix_log=8, pbs_count=3, cbs_radix_log=4, cbs_count=4)
SMALL16 = O.DEFAULT_128.replace(lwe_n=5, lwe_std=0.0, N=16, k=1, glwe_std=0.0, pbs_radix_log=6, pbs_count=2, cbs_radix_log=5,
                                cbs_count=3, ks_radix_log=2, ks_count=6)


def _ggsw(rng, sk, bit, P):
    return O.encrypt_ggsw_fft(rng, sk, bit, P.N, P.k, P.cbs_radix_log, P.cbs_count, P.glwe_std)


def test_reference_can_cmux_fft_replayed_on_the_gpu():
    """fft_ops.rs:577-619 at TEST_GLWE_DEF_1 / TEST_RADIX: cmux(a, b, sel) decrypts to b when sel = 1, else a."""
    P = TEST1
    eng = spf_amd.Engine(to_engine_params(P))
    rng = O.Rng(0xC0DE1)
    sk = O.gen_binary_key(rng, P.k * P.N)
    r = np.random.default_rng(5)
    B = 24
    sels = r.integers(0, 2, B)
    a_pt = r.integers(0, 2, (B, P.N)).astype(np.uint64)
    b_pt = r.integers(0, 2, (B, P.N)).astype(np.uint64)
    enc = lambda pt: O.encrypt_glwe(rng, sk, np.array([O.encode(int(v), 1) for v in pt], dtype=np.uint64), P.N, P.k, P.glwe_std)  # noqa: E731
    a = np.stack([enc(x) for x in a_pt])
    b = np.stack([enc(x) for x in b_pt])
    g = np.stack([_ggsw(rng, sk, int(s), P) for s in sels])
    got = eng.cmux(g, a, b)
    assert eng.last_cmux_kernel() == "generic_cmux_kernel"
    for i in range(B):
        assert np.array_equal(got[i], O.cmux(a[i], b[i], g[i], P.N, P.k, P.cbs_radix_log, P.cbs_count)), i
        dec = np.array([O.decode(int(t), 1) for t in O.decrypt_glwe_raw(got[i], sk, P.N, P.k)], dtype=np.uint64)
        assert np.array_equal(dec, b_pt[i] if sels[i] else a_pt[i]), i


def test_reference_can_fft_external_product_replayed_on_the_gpu():
    """fft_ops.rs:537-575: glwe [*] ggsw decrypts to the GLWE's plaintext when the GGSW encrypts 1, to zero when 0; and
    glev_cmux = cmux over each constituent GLWE."""
    P = TEST1
    eng = spf_amd.Engine(to_engine_params(P))
    rng = O.Rng(0xC0DE2)
    sk = O.gen_binary_key(rng, P.k * P.N)
    r = np.random.default_rng(6)
    B = 16
    sels = r.integers(0, 2, B)
    pt = r.integers(0, 2, (B, P.N)).astype(np.uint64)
    x = np.stack([O.encrypt_glwe(rng, sk, np.array([O.encode(int(v), 1) for v in p], dtype=np.uint64), P.N, P.k, P.glwe_std) for p in pt])
    g = np.stack([_ggsw(rng, sk, int(s), P) for s in sels])
    got = eng.multiply_glwe_ggsw(x, g)
    for i in range(B):
        fft = O.glwe_ggsw_mad(np.zeros(P.glwe_len // 2, dtype=np.complex128), x[i], g[i], P.N, P.k, P.cbs_radix_log, P.cbs_count)
        h = P.N // 2
        exp = np.concatenate([O.poly_ifft(fft[q * h:(q + 1) * h]) for q in range(P.k + 1)])
        assert np.array_equal(got[i], exp), i
        dec = np.array([O.decode(int(t), 1) for t in O.decrypt_glwe_raw(got[i], sk, P.N, P.k)], dtype=np.uint64)
        assert np.array_equal(dec, pt[i] if sels[i] else np.zeros(P.N, dtype=np.uint64)), i
    # glev_cmux (fft_ops.rs:203-220)
    ga = random_glwe(7, 3 * P.cbs_count, P.glwe_len).reshape(3, P.cbs_count, P.glwe_len)
    gb = random_glwe(8, 3 * P.cbs_count, P.glwe_len).reshape(3, P.cbs_count, P.glwe_len)
    gg = eng.glev_cmux(g[:3], ga, gb).reshape(3, P.cbs_count, P.glwe_len)
    for i in range(3):
        for j in range(P.cbs_count):
            assert np.array_equal(gg[i, j], O.cmux(ga[i, j], gb[i, j], g[i], P.N, P.k, P.cbs_radix_log, P.cbs_count)), (i, j)


@pytest.mark.parametrize("P", [TEST1, TEST2, P1024, SMALL16], ids=["N128k2", "N256k3", "N1024k1", "N16k1"])
def test_generic_bootstrap_family_against_the_oracle(P):
    """generalized PBS (shared and per-ciphertext LUTs, several (log_chi, log_v, rotation)), univariate PBS, the circuit
    bootstrap's PBS, keyswitch, sample extract and the linear operations at other parameter sets: every word against the
    oracle."""
    ks = O.gen_keyset(0x5EED0007, P)
    eng = spf_amd.Engine(to_engine_params(P))
    eng.load_bootstrap_key(ks.bsk_fft)
    eng.load_keyswitch_key(ks.ksk)
    B = 5
    lwe = random_lwe_batch(0x6E00 + P.N, B, P.lwe_n)
    lut = random_glwe(0x6E10 + P.N, 1, P.glwe_len)[0]
    luts = random_glwe(0x6E20 + P.N, B, P.glwe_len)
    got = eng.generalized_pbs(lwe, lut, 0, 0, 0)
    assert eng.last_blind_rotate_kernel() == "generic_pbs_kernel"
    for i in range(B):
        assert np.array_equal(got[i], O.generalized_pbs(lwe[i], lut, ks.bsk_fft, P, 0, 0)), i
    got = eng.generalized_pbs(lwe, luts, 1, 2, 1 << 61)
    for i in range(B):
        rot = lwe[i].copy()
        rot[-1] = np.uint64((int(rot[-1]) + (1 << 61)) & ((1 << 64) - 1))
        assert np.array_equal(got[i], O.generalized_pbs(rot, luts[i], ks.bsk_fft, P, 1, 2)), i
    u = eng.pbs_univariate(lwe, lut)
    for i in range(B):
        assert np.array_equal(u[i], O.pbs_univariate(lwe[i], lut, ks.bsk_fft, P)), i
    c = eng.circuit_bootstrap_pbs(lwe)
    for i in range(B):
        assert np.array_equal(c[i], O.cbs_pbs(lwe[i], ks.bsk_fft, P)), i
    lwe1 = random_lwe_batch(0x6E30 + P.N, B, P.k * P.N)
    sw = eng.keyswitch_lwe_l1_lwe_l0(lwe1)
    for i in range(B):
        assert np.array_equal(sw[i], O.keyswitch_lwe(lwe1[i], ks.ksk, P.k * P.N, P.lwe_n, P.ks_radix_log, P.ks_count)), i
    gate = eng.gate_bootstrap(lwe1)
    for i in range(B):
        assert np.array_equal(gate[i], O.cbs_pbs(sw[i], ks.bsk_fft, P)), i
    a = random_glwe(0x6E40 + P.N, B, P.glwe_len)
    b = random_glwe(0x6E50 + P.N, B, P.glwe_len)
    for idx in (0, 1, P.N // 2, P.N - 1):
        se = eng.sample_extract_l1(a, idx)
        for i in range(B):
            assert np.array_equal(se[i], O.sample_extract(a[i], idx, P.N, P.k)), (idx, i)
    n1, x = eng.glwe_not(a), eng.glwe_xor(a, b)
    for amount in (0, 1, P.N - 1, P.N, P.N + 3, 2 * P.N + 5):
        m = eng.glwe_mul_xn(a, amount)
        for i in range(B):
            assert np.array_equal(m[i], O.glwe_mul_xn(a[i], amount, P.N, P.k)), (amount, i)
    for i in range(B):
        assert np.array_equal(n1[i], O.glwe_not(a[i], P.N, P.k))
        assert np.array_equal(x[i], O.glwe_xor(a[i], b[i], P.N, P.k))


def test_reference_bootstrap_helper_shape_replayed_at_small_parameters():
    """programmable_bootstrapping.rs:708-779 (`can_bootstrap`, `can_bootstrap_with_map`) at TEST_LWE_DEF_1 / TEST_GLWE_DEF_1:
    every message of the plaintext space, encrypted with a padding bit, bootstrapped through `generate_lut`'s table, decrypts to
    map(msg)."""
    P = TEST1.replace(pbs_radix_log=8, pbs_count=2)
    EP = to_engine_params(P)
    ks = O.gen_keyset(0x5EED0008, P, with_ksk=False)
    eng = spf_amd.Engine(EP)
    eng.load_bootstrap_key(ks.bsk_fft)
    bits = 2
    for fmap in (lambda v: v, lambda v: (v + 3) % 4):
        lut = spf_amd.generate_lut([fmap], bits, EP)
        assert np.array_equal(lut, O.trivial_lut_glwe(O.generate_lut(P.N, [fmap], bits), P))
        msgs = list(range(1 << bits))
        rng = O.Rng(77)
        cts = np.stack([O.encrypt_lwe(rng, ks.lwe_sk, m << (64 - bits - 1), P.lwe_std) for m in msgs])
        out = eng.pbs_univariate(cts, lut)
        for m, ct_in, ct_out in zip(msgs, cts, out):
            assert np.array_equal(ct_out, O.pbs_univariate(ct_in, lut, ks.bsk_fft, P))
            dec = O.decode(O.decrypt_lwe_raw(ct_out, ks.glwe_sk), bits)
            assert dec == fmap(m), (m, dec)


def _eng_params(P):
    return to_engine_params(P).replace(tr_radix_log=P.tr_radix_log, tr_radix_count=P.tr_count,
                                       ss_radix_log=P.ss_radix_log, ss_radix_count=P.ss_count)


@pytest.mark.parametrize("P", [TEST1.replace(lwe_n=6, tr_radix_log=7, tr_count=6, ss_radix_log=3, ss_count=15),
                               TEST2.replace(lwe_n=4, tr_radix_log=5, tr_count=7, ss_radix_log=4, ss_count=8),
                               SMALL16.replace(tr_radix_log=6, tr_count=5, ss_radix_log=5, ss_count=6)],
                         ids=["N128k2", "N256k3", "N16k1"])
def test_generic_circuit_bootstrap_tail_against_the_oracle(P):
    """mod_switch_trace_and_rotate, scheme_switch_fft and the whole `Evaluation::circuit_bootstrap` at k = 2 / 3 / 1 (the upper-
    triangular scheme-switch key pairs of k > 1 included): every word against the oracle."""
    ks = O.gen_keyset(0x5EED0009, P)
    r = O.Rng(0x7A12)
    ak = O.gen_auto_key_fft(r, ks.glwe_sk, P)
    ssk = O.gen_ssk_fft(r, ks.glwe_sk, P)
    eng = spf_amd.Engine(_eng_params(P))
    eng.load_bootstrap_key(ks.bsk_fft)
    eng.load_keyswitch_key(ks.ksk)
    eng.load_automorphism_key(ak)
    eng.load_scheme_switch_key(ssk)
    B = 3
    glwe = random_glwe(0x7B00 + P.N, B, P.glwe_len)
    got = eng.mod_switch_trace_and_rotate(glwe)
    for i in range(B):
        assert np.array_equal(got[i], O.mod_switch_trace_and_rotate(glwe[i], ak, P)), i
    glev = random_glwe(0x7B10 + P.N, B * P.cbs_count, P.glwe_len).reshape(B, P.cbs_count, P.glwe_len)
    gg = eng.scheme_switch(glev)
    for i in range(B):
        assert np.array_equal(gg[i].view(np.float64), O.scheme_switch_fft(glev[i], ssk, P).view(np.float64)), i
    lwe = random_lwe_batch(0x7B20 + P.N, B, P.lwe_n)
    cb = eng.circuit_bootstrap(lwe)
    lwe1 = random_lwe_batch(0x7B30 + P.N, B, P.k * P.N)
    kcb = eng.keyswitch_circuit_bootstrap(lwe1)
    for i in range(B):
        assert np.array_equal(cb[i].view(np.float64), O.circuit_bootstrap(lwe[i], ks.bsk_fft, ak, ssk, P).view(np.float64)), i
        l0 = O.keyswitch_lwe(lwe1[i], ks.ksk, P.k * P.N, P.lwe_n, P.ks_radix_log, P.ks_count)
        assert np.array_equal(kcb[i].view(np.float64), O.circuit_bootstrap(l0, ks.bsk_fft, ak, ssk, P).view(np.float64)), i
    for bit in (0, 1):
        triv = np.zeros(P.lwe_n + 1, dtype=np.uint64)
        triv[-1] = np.uint64(bit << 63)
        assert np.array_equal(eng.l1ggsw_constant(bit).view(np.float64), O.circuit_bootstrap(triv, ks.bsk_fft, ak, ssk, P).view(np.float64))


@pytest.mark.parametrize("P", [TEST1.replace(lwe_n=5, tr_radix_log=7, tr_count=6, ss_radix_log=3, ss_count=15),
                               SMALL16.replace(tr_radix_log=6, tr_count=5, ss_radix_log=5, ss_count=6)], ids=["N128k2", "N16k1"])
def test_gate_graph_and_values_by_handle_over_a_generic_parameter_set(P):
    """`FheCircuit` + `CircuitProcessor` (fhe_circuit.rs:34-205, circuit_processor/mod.rs:573-623) and the per-operation boundary by
    handle are parameter-agnostic too: the CMUX family over scattered operands runs in the generic kernel.  A graph with every
    operation kind at TEST_GLWE_DEF_1 (k = 2) / N = 16, and the same operations one by one through the pool by handle, against
    the oracle (CMux, conversion chain) and the batch entry points (which the tests above hold to the oracle)."""
    from spf_amd import FheOp, ValueKind
    ks = O.gen_keyset(0x5EED000C, P)
    r = O.Rng(0x7A15)
    ak, ssk = O.gen_auto_key_fft(r, ks.glwe_sk, P), O.gen_ssk_fft(r, ks.glwe_sk, P)
    eng = spf_amd.Engine(_eng_params(P))
    eng.load_bootstrap_key(ks.bsk_fft)
    eng.load_keyswitch_key(ks.ksk)
    eng.load_automorphism_key(ak)
    eng.load_scheme_switch_key(ssk)
    glwe = random_glwe(0x7E00 + P.N, 6, P.glwe_len)
    lwe1 = random_lwe_batch(0x7E01 + P.N, 3, P.k * P.N)
    glev = random_glwe(0x7E02 + P.N, 2 * P.cbs_count, P.glwe_len).reshape(2, -1)

    g = spf_amd.FheCircuit(eng)
    gi = [g.add_input(ValueKind.GLWE1, x) for x in glwe]
    li = [g.add_input(ValueKind.LWE1, x) for x in lwe1]
    vi = [g.add_input(ValueKind.GLEV1, x) for x in glev]
    one, gone = g.add_trivial(ValueKind.GLWE1, 1), g.add_trivial(ValueKind.GGSW1, 1)
    k0 = [g.add_op(FheOp.KeyswitchL1toL0, [x]) for x in li]
    se = g.add_op(FheOp.SampleExtract, [gi[4]], 5)
    nt = g.add_op(FheOp.Not, [gi[1]])
    ad = g.add_op(FheOp.GlweAdd, [gi[2], gi[5]])
    rot = g.add_op(FheOp.MulXN, [gi[0]], 2 * P.N + 7)
    ss = g.add_op(FheOp.SchemeSwitch, [vi[1]])
    cb = [g.add_op(FheOp.CircuitBootstrap, [x]) for x in k0]
    k1 = g.add_op(FheOp.KeyswitchL1toL0, [se])
    mux = g.add_op(FheOp.CMux, [cb[0], nt, ad])
    mul = g.add_op(FheOp.MultiplyGgswGlwe, [cb[1], rot])
    gmux = g.add_op(FheOp.GlevCMux, [cb[2], vi[0], vi[1]])
    mux2 = g.add_op(FheOp.CMux, [ss, one, gi[3]])
    mux3 = g.add_op(FheOp.CMux, [gone, gi[3], gi[4]])
    cb1 = g.add_op(FheOp.CircuitBootstrap, [k1])
    last = g.add_op(FheOp.CMux, [cb1, mux, mul])
    outs = {n: g.add_output(n, k) for n, k in [
        (k0[2], ValueKind.LWE0), (se, ValueKind.LWE1), (ss, ValueKind.GGSW1), (cb[1], ValueKind.GGSW1), (mux, ValueKind.GLWE1),
        (mul, ValueKind.GLWE1), (gmux, ValueKind.GLEV1), (mux2, ValueKind.GLWE1), (mux3, ValueKind.GLWE1), (last, ValueKind.GLWE1)]}
    g.run()
    assert eng.last_cmux_kernel() == "generic_cmux_kernel"

    e_k0 = eng.keyswitch_lwe_l1_lwe_l0(lwe1)
    e_cb = eng.circuit_bootstrap(e_k0)
    e_se = eng.sample_extract_l1(glwe[4:5], 5)
    e_nt, e_ad = eng.glwe_not(glwe[1:2])[0], eng.glwe_xor(glwe[2:3], glwe[5:6])[0]
    e_rot = eng.glwe_mul_xn(glwe[0:1], 2 * P.N + 7)[0]
    e_ss = eng.scheme_switch(glev[1:2])
    e_mux = eng.cmux(e_cb[0:1], e_nt, e_ad)[0]
    e_mul = eng.multiply_glwe_ggsw(e_rot, e_cb[1:2])[0]
    trivial_one = np.zeros(P.glwe_len, dtype=np.uint64)
    trivial_one[P.N * P.k] = 1 << 63
    e_cb1 = eng.circuit_bootstrap(eng.keyswitch_lwe_l1_lwe_l0(e_se))
    assert np.array_equal(outs[k0[2]], O.keyswitch_lwe(lwe1[2], ks.ksk, P.k * P.N, P.lwe_n, P.ks_radix_log, P.ks_count))
    assert np.array_equal(outs[se], O.sample_extract(glwe[4], 5, P.N, P.k))
    assert np.array_equal(outs[ss].view(np.float64), O.scheme_switch_fft(glev[1].reshape(P.cbs_count, -1), ssk, P).view(np.float64))
    assert np.array_equal(outs[cb[1]].view(np.float64), O.circuit_bootstrap(e_k0[1], ks.bsk_fft, ak, ssk, P).view(np.float64))
    assert np.array_equal(outs[mux], O.cmux(e_nt, e_ad, e_cb[0], P.N, P.k, P.cbs_radix_log, P.cbs_count))
    assert np.array_equal(outs[mux], e_mux) and np.array_equal(outs[mul], e_mul)
    assert np.array_equal(outs[gmux], eng.glev_cmux(e_cb[2:3], glev[0:1], glev[1:2]).reshape(-1))
    assert np.array_equal(outs[mux2], eng.cmux(e_ss, trivial_one, glwe[3])[0])
    assert np.array_equal(outs[mux3], eng.cmux(eng.l1ggsw_constant(1)[None], glwe[3], glwe[4])[0])
    assert np.array_equal(outs[last], O.cmux(e_mux, e_mul, e_cb1[0], P.N, P.k, P.cbs_radix_log, P.cbs_count))
    g.close()

    # the same operations one at a time by handle (operands and results stay in HBM between the calls)
    pool = spf_amd.Pool(eng, max_batch=16, max_wait_us=200)
    try:
        v = [pool.upload(ValueKind.GLWE1, x) for x in glwe]
        vl = pool.upload(ValueKind.LWE1, lwe1[0])
        ve = [pool.upload(ValueKind.GLEV1, x) for x in glev]
        sel = pool.run_v(FheOp.CircuitBootstrap, [pool.run_v(FheOp.KeyswitchL1toL0, [vl])])
        assert np.array_equal(sel.download().view(np.float64), e_cb[0].view(np.float64))
        assert np.array_equal(pool.keyswitch_circuit_bootstrap_v(vl).download().view(np.float64), e_cb[0].view(np.float64))
        m = pool.run_v(FheOp.CMux, [sel, pool.run_v(FheOp.Not, [v[1]]), pool.run_v(FheOp.GlweAdd, [v[2], v[5]])])
        assert np.array_equal(m.download(), e_mux)
        assert np.array_equal(pool.run_v(FheOp.MultiplyGgswGlwe, [sel, v[0]]).download(), eng.multiply_glwe_ggsw(glwe[0], e_cb[0:1])[0])
        assert np.array_equal(pool.run_v(FheOp.GlevCMux, [sel, ve[0], ve[1]]).download(), eng.glev_cmux(e_cb[0:1], glev[0:1], glev[1:2]).reshape(-1))
        assert np.array_equal(pool.run_v(FheOp.SchemeSwitch, [ve[1]]).download().view(np.float64), e_ss.reshape(-1).view(np.float64))
        assert np.array_equal(pool.run_v(FheOp.CMux, [pool.trivial(ValueKind.GGSW1, 1), v[3], v[4]]).download(), outs[mux3])
        assert np.array_equal(pool.run_v(FheOp.SampleExtract, [v[4]], 5).download(), e_se[0])
        assert np.array_equal(pool.run_v(FheOp.MulXN, [v[0]], 2 * P.N + 7).download(), e_rot)
        # ... and pushed: pending results as operands, nothing waited for but the last value
        psel = pool.push_v(FheOp.CircuitBootstrap, [pool.push_v(FheOp.KeyswitchL1toL0, [vl])])
        pm = pool.push_v(FheOp.CMux, [psel, pool.push_v(FheOp.Not, [v[1]]), pool.push_v(FheOp.GlweAdd, [v[2], v[5]])])
        plast = pool.push_v(FheOp.CMux, [psel, pm, pool.push_v(FheOp.MultiplyGgswGlwe, [psel, pm])])
        assert np.array_equal(plast.wait().download(), eng.cmux(e_cb[0:1], e_mux, eng.multiply_glwe_ggsw(e_mux, e_cb[0:1])[0])[0])
        assert np.array_equal(pm.download(), e_mux)
    finally:
        import gc
        gc.collect()
        pool.close()


def test_generic_contexts_say_what_they_do_not_do():
    P = to_engine_params(TEST1)
    with pytest.raises(spf_amd.SpfError):
        spf_amd.Engine(P.replace(polynomial_degree=96))        # not a power of two
    with pytest.raises(spf_amd.SpfError):
        spf_amd.Engine(spf_amd.DEFAULT_128.replace(glwe_size=3))   # does not fit the generic kernels' LDS


def test_n2048_with_another_radix_runs_dag1_in_the_generic_family():
    """`can_generalized_bootstrap` (programmable_bootstrapping.rs:925-990) runs TEST_RADIX = 3 x 4 bits at N = 2048: outside the
    tuned kernels (2 x 16).  The generic family carries DAG-I in array form, so the words are the ones the tuned kernels and the
    oracle define: PBS, CMUX, trace, scheme switch and the whole circuit bootstrap against the oracle, and the SAME batch through
    a tuned context where the parameters allow (CMUX at cbs 4 x 4 is radix-independent of the PBS)."""
    P = O.DEFAULT_128.replace(lwe_n=3, pbs_radix_log=4, pbs_count=3, tr_radix_log=6, tr_count=7, ss_radix_log=5, ss_count=9)
    ks = O.gen_keyset(0x5EED000A, P)
    r = O.Rng(0x7A13)
    ak = O.gen_auto_key_fft(r, ks.glwe_sk, P)
    ssk = O.gen_ssk_fft(r, ks.glwe_sk, P)
    eng = spf_amd.Engine(_eng_params(P))
    eng.load_bootstrap_key(ks.bsk_fft)
    eng.load_automorphism_key(ak)
    eng.load_scheme_switch_key(ssk)
    B = 3
    lwe = random_lwe_batch(0x7C00, B, P.lwe_n)
    lut = random_glwe(0x7C01, 1, P.glwe_len)[0]
    got = eng.generalized_pbs(lwe, lut, 0, 1, 0)
    assert eng.last_blind_rotate_kernel() == "generic_pbs_kernel"
    for i in range(B):
        assert np.array_equal(got[i], O.generalized_pbs(lwe[i], lut, ks.bsk_fft, P, 0, 1)), i
    cb = eng.circuit_bootstrap(lwe[:2])
    for i in range(2):
        assert np.array_equal(cb[i].view(np.float64), O.circuit_bootstrap(lwe[i], ks.bsk_fft, ak, ssk, P).view(np.float64)), i
    # the generic CMUX against the tuned one: same selector, same operands, same words
    tuned = spf_amd.Engine(spf_amd.DEFAULT_128.replace(lwe_dimension=3))
    a, b = random_glwe(0x7C02, 2, P.glwe_len), random_glwe(0x7C03, 2, P.glwe_len)
    assert np.array_equal(eng.cmux(cb, a, b), tuned.cmux(cb, a, b))
    assert eng.last_cmux_kernel() == "generic_cmux_kernel" and tuned.last_cmux_kernel().startswith("cmux")


def test_pool_and_group_over_a_generic_parameter_set():
    """The host layers above the kernels are parameter-agnostic: the call-coalescing pool and a device group `[0, 0]` at
    TEST_GLWE_DEF_1 / TEST_LWE_DEF_1 give the words of the single generic context."""
    from concurrent.futures import ThreadPoolExecutor
    P = TEST1.replace(lwe_n=5, tr_radix_log=7, tr_count=6, ss_radix_log=3, ss_count=15)
    EP = _eng_params(P)
    ks = O.gen_keyset(0x5EED000B, P)
    r = O.Rng(0x7A14)
    ak, ssk = O.gen_auto_key_fft(r, ks.glwe_sk, P), O.gen_ssk_fft(r, ks.glwe_sk, P)
    eng, grp = spf_amd.Engine(EP), spf_amd.Group(EP, devices=[0, 0])
    for e in (eng, grp):
        e.load_bootstrap_key(ks.bsk_fft)
        e.load_keyswitch_key(ks.ksk)
        e.load_automorphism_key(ak)
        e.load_scheme_switch_key(ssk)
    n = 9
    lwe1 = random_lwe_batch(0x7D00, n, P.k * P.N)
    a, b = random_glwe(0x7D01, n, P.glwe_len), random_glwe(0x7D02, n, P.glwe_len)
    want_g = eng.keyswitch_circuit_bootstrap(lwe1)
    want_m = eng.cmux(want_g, a, b)
    assert np.array_equal(grp.keyswitch_circuit_bootstrap(lwe1).view(np.float64), want_g.view(np.float64))
    assert np.array_equal(grp.cmux(want_g, a, b), want_m)
    assert np.array_equal(grp.gate_bootstrap(lwe1), eng.gate_bootstrap(lwe1))
    pool = spf_amd.Pool(eng, max_batch=16, max_wait_us=500)
    ggsw = np.zeros((n, EP.cbs_ggsw_complex), dtype=np.complex128)
    mux = np.zeros((n, P.glwe_len), dtype=np.uint64)

    def task(i):
        pool.keyswitch_circuit_bootstrap(ggsw[i], lwe1[i])
        pool.cmux(mux[i], ggsw[i], a[i], b[i])
        return i

    try:
        with ThreadPoolExecutor(max_workers=n) as ex:
            assert sorted(ex.map(task, range(n))) == list(range(n))
    finally:
        pool.close()
        grp.close()
    assert np.array_equal(ggsw.view(np.float64), want_g.view(np.float64))
    assert np.array_equal(mux, want_m)
