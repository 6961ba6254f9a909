"""The level-batching graph executor (`spf_graph_*`) — the counterpart of
`FheCircuit` + `CircuitProcessor::run_graph_blocking` (fhe_circuit.rs:34-205,
circuit_processor/mod.rs:573-623).  A graph must return exactly what the same operations return when
called one by one through the batch entry points, must batch by level, and must reject malformed
graphs when they are built (task.rs:26-31), not when they run."""
import numpy as np
import pytest

import oracle as O
import spf_amd
from spf_amd import FheOp, ValueKind
from tests.util import keyset, random_glwe, random_lwe_batch, to_engine_params

pytestmark = pytest.mark.gpu

SMALL_N = 12


@pytest.fixture(scope="module")
def rig():
    return _make_rig()


def _make_rig():
    ks = keyset(0x5EED0001, SMALL_N)
    P = ks.params
    r = O.Rng(0x6A11)
    eng = spf_amd.Engine(to_engine_params(P))
    eng.load_bootstrap_key(ks.bsk_fft)
    eng.load_keyswitch_key(ks.ksk)
    eng.load_automorphism_key(O.gen_auto_key_fft(r, ks.glwe_sk, P))
    eng.load_scheme_switch_key(O.gen_ssk_fft(r, ks.glwe_sk, P))
    return ks, eng


def test_every_operation_matches_the_batch_entry_points(rig):
    ks, eng = rig
    P = ks.params
    g = spf_amd.FheCircuit(eng)
    glwe = random_glwe(1, 6, P.glwe_len)
    lwe1 = random_lwe_batch(2, 3, P.N * P.k)
    glev = random_glwe(3, 2 * P.cbs_count, P.glwe_len).reshape(2, -1)
    gi = [g.add_input(ValueKind.GLWE1, x) for x in glwe]
    li = [g.add_input(ValueKind.LWE1, x) for x in lwe1]
    vi = [g.add_input(ValueKind.GLEV1, x) for x in glev]
    one = g.add_trivial(ValueKind.GLWE1, 1)

    # level 1: keyswitch x3 (operands interleaved with other inputs -> contiguous among themselves),
    # sample extract of two different indices (two groups), linear ops
    k0 = [g.add_op(FheOp.KeyswitchL1toL0, [x]) for x in li]
    se5 = g.add_op(FheOp.SampleExtract, [gi[4]], 5)
    se0 = [g.add_op(FheOp.SampleExtract, [gi[i]], 0) for i in (3, 0)]       # out of order -> gathered
    nt = g.add_op(FheOp.Not, [gi[1]])
    ad = g.add_op(FheOp.GlweAdd, [gi[2], gi[5]])
    rot = g.add_op(FheOp.MulXN, [gi[0]], 2048 + 77)
    ss = g.add_op(FheOp.SchemeSwitch, [vi[1]])
    # level 2: circuit bootstrap of the three keyswitched values (contiguous) and of se -> ks chain later
    cb = [g.add_op(FheOp.CircuitBootstrap, [x]) for x in k0]
    k1 = g.add_op(FheOp.KeyswitchL1toL0, [se5])
    # level 3
    mux = g.add_op(FheOp.CMux, [cb[0], nt, ad])
    mul = g.add_op(FheOp.MultiplyGgswGlwe, [cb[1], rot])
    gmux = g.add_op(FheOp.GlevCMux, [cb[2], vi[0], vi[1]])
    mux2 = g.add_op(FheOp.CMux, [ss, one, gi[3]])
    cb1 = g.add_op(FheOp.CircuitBootstrap, [k1])
    # level 4
    last = g.add_op(FheOp.CMux, [cb1, mux, mul])

    outs = {n: g.add_output(n, k) for n, k in [
        (k0[2], ValueKind.LWE0), (se5, ValueKind.LWE1), (se0[0], ValueKind.LWE1), (se0[1], ValueKind.LWE1),
        (nt, ValueKind.GLWE1), (ad, ValueKind.GLWE1), (rot, ValueKind.GLWE1), (ss, ValueKind.GGSW1),
        (cb[1], ValueKind.GGSW1), (mux, ValueKind.GLWE1), (mul, ValueKind.GLWE1), (gmux, ValueKind.GLEV1),
        (mux2, ValueKind.GLWE1), (last, ValueKind.GLWE1)]}
    g.run()
    st = g.stats()
    assert st["levels"] == 4 and st["nodes"] == 12 + 20

    # the same values, operation by operation
    e_k0 = eng.keyswitch_lwe_l1_lwe_l0(lwe1)
    e_cb = eng.circuit_bootstrap(e_k0)
    e_se5 = eng.sample_extract_l1(glwe[4:5], 5)
    e_nt = eng.glwe_not(glwe[1:2])[0]
    e_ad = eng.glwe_xor(glwe[2:3], glwe[5:6])[0]
    e_rot = eng.glwe_mul_xn(glwe[0:1], 2048 + 77)[0]
    e_ss = eng.scheme_switch(glev[1:2])
    e_mux = eng.cmux(e_cb[0:1], e_nt, e_ad)[0]
    e_mul = eng.multiply_glwe_ggsw(e_rot, e_cb[1:2])[0]
    e_gmux = eng.glev_cmux(e_cb[2:3], glev[0:1], glev[1:2]).reshape(-1)
    trivial_one = np.zeros(P.glwe_len, dtype=np.uint64)
    trivial_one[P.N * P.k] = 1 << 63
    e_mux2 = eng.cmux(e_ss, trivial_one, glwe[3])[0]
    e_cb1 = eng.circuit_bootstrap(eng.keyswitch_lwe_l1_lwe_l0(e_se5))
    e_last = eng.cmux(e_cb1, e_mux, e_mul)[0]

    assert np.array_equal(outs[k0[2]], e_k0[2])
    assert np.array_equal(outs[se5], e_se5[0])
    assert np.array_equal(outs[se0[0]], eng.sample_extract_l1(glwe[3:4], 0)[0])
    assert np.array_equal(outs[se0[1]], eng.sample_extract_l1(glwe[0:1], 0)[0])
    assert np.array_equal(outs[nt], e_nt) and np.array_equal(outs[ad], e_ad) and np.array_equal(outs[rot], e_rot)
    assert np.array_equal(outs[ss].view(np.float64), e_ss.reshape(-1).view(np.float64))
    assert np.array_equal(outs[cb[1]].view(np.float64), e_cb[1].reshape(-1).view(np.float64))
    assert np.array_equal(outs[mux], e_mux) and np.array_equal(outs[mul], e_mul)
    assert np.array_equal(outs[gmux], e_gmux)
    assert np.array_equal(outs[mux2], e_mux2)
    assert np.array_equal(outs[last], e_last)

    # run again on new input contents: the graph re-reads its input buffers
    g._keep[0][...] = random_glwe(99, 1, P.glwe_len)[0]     # gi[0]
    g.run()
    assert np.array_equal(outs[rot], eng.glwe_mul_xn(g._keep[0][None], 2048 + 77)[0])
    assert np.array_equal(outs[nt], e_nt)
    g.close()


def test_malformed_graphs_are_rejected_when_built(rig):
    ks, eng = rig
    P = ks.params
    g = spf_amd.FheCircuit(eng)
    x = g.add_input(ValueKind.GLWE1, random_glwe(1, 1, P.glwe_len)[0])
    l0 = g.add_input(ValueKind.LWE0, random_lwe_batch(2, 1, P.lwe_n)[0])
    with pytest.raises(spf_amd.SpfError):
        g.add_op(FheOp.CircuitBootstrap, [x])                 # wrong ciphertext type
    with pytest.raises(spf_amd.SpfError):
        g.add_op(FheOp.CMux, [x, x])                          # wrong arity
    with pytest.raises(spf_amd.SpfError):
        g.add_op(FheOp.Not, [1234])                           # not a node
    with pytest.raises(spf_amd.SpfError):
        g.add_op(FheOp.SampleExtract, [x], P.N)               # faults.rs: illegal index
    with pytest.raises(spf_amd.SpfError):
        g.add_input(ValueKind.LWE0, np.zeros(3, dtype=np.uint64))
    with pytest.raises(spf_amd.SpfError):
        g.add_trivial(ValueKind.GGSW1, 2)                     # a constant is a bit
    cb = g.add_op(FheOp.CircuitBootstrap, [l0])               # the graph is still usable
    out = g.add_output(cb, ValueKind.GGSW1)
    g.run()
    assert np.array_equal(out.view(np.float64), eng.circuit_bootstrap(g._keep[1][None]).reshape(-1).view(np.float64))
    g.close()


def test_encrypted_add_32_as_one_graph():
    """BASELINE.json config 3: the whole 32-bit addition — 64 x (SampleExtract -> KeyswitchL1toL0 ->
    CircuitBootstrap) and the ripple-carry CMUX chain — as ONE graph run: one input copy, 68 levels
    enqueued back to back, one output copy."""
    ks = keyset(0x5EED0001, 637)
    P = ks.params
    r = O.Rng(0xADD32)
    eng = spf_amd.Engine(to_engine_params(P))
    eng.load_bootstrap_key(ks.bsk_fft)
    eng.load_keyswitch_key(ks.ksk)
    eng.load_automorphism_key(O.gen_auto_key_fft(r, ks.glwe_sk, P))
    eng.load_scheme_switch_key(O.gen_ssk_fft(r, ks.glwe_sk, P))

    a, b = 0xDEADBEEF, 0x1234ABCD
    bits = [(a >> i) & 1 for i in range(32)] + [(b >> i) & 1 for i in range(32)]
    g = spf_amd.FheCircuit(eng)
    sel = []
    for bit in bits:
        m = np.zeros(P.N, dtype=np.uint64)
        m[0] = O.encode(bit, 1)
        x = g.add_input(ValueKind.GLWE1, O.encrypt_glwe(r, ks.glwe_sk, m, P.N, P.k, P.glwe_std))
        x = g.add_op(FheOp.SampleExtract, [x], 0)
        x = g.add_op(FheOp.KeyswitchL1toL0, [x])
        sel.append(g.add_op(FheOp.CircuitBootstrap, [x]))
    ga, gb = sel[:32], sel[32:]
    zero = g.add_trivial(ValueKind.GLWE1, 0)
    one = g.add_trivial(ValueKind.GLWE1, 1)
    carry = zero
    sums = []
    for i in range(32):
        ncarry = g.add_op(FheOp.Not, [carry])
        # selector b_i: [c, ~c] [~c, c] [0, c] [c, 1];  CMux operands are [sel, low, high]
        l1 = [g.add_op(FheOp.CMux, [gb[i], lo, hi]) for lo, hi in
              [(carry, ncarry), (ncarry, carry), (zero, carry), (carry, one)]]
        sums.append(g.add_output(g.add_op(FheOp.CMux, [ga[i], l1[0], l1[1]]), ValueKind.GLWE1))
        carry = g.add_op(FheOp.CMux, [ga[i], l1[2], l1[3]])
    carry_out = g.add_output(carry, ValueKind.GLWE1)
    g.run()
    st = g.stats()
    # conversion = 3 levels; the carry advances 2 levels per bit (the two carry CMUXes of the first
    # stage do not need ~c), the last sum bit finishes one level after the last carry
    assert st["levels"] == 3 + 2 * 32 + 1
    assert st["launches"] <= 2 * st["levels"], st          # Not and CMux groups share levels; no gathers

    got = 0
    for i, s in enumerate(sums):
        got |= O.decode(int(O.decrypt_glwe_raw(s, ks.glwe_sk, P.N, P.k)[0]), 1) << i
    assert got == (a + b) & 0xFFFFFFFF
    assert O.decode(int(O.decrypt_glwe_raw(carry_out, ks.glwe_sk, P.N, P.k)[0]), 1) == ((a + b) >> 32) & 1
    g.close()


def test_graph_ggsw_and_glev_constants(full_rig=None):
    """FheOp::{Zero,One}Ggsw1 / {Zero,One}Glev1 (fhe_circuit.rs:96-116): the GGSW constants are the context's
    circuit bootstraps of the trivial L0 LWE (Evaluation::new, evaluation.rs:161-197), the GLEV ones the
    trivial gadget encryptions (encryption.rs:434-451).  Checked against the oracle and by what they select."""
    import oracle as O
    from tests.util import keyset, random_glwe, to_engine_params
    ks = keyset(0x5EED0001, 12)
    P = ks.params
    r = O.Rng(0x7A11)
    eng = spf_amd.Engine(to_engine_params(P))
    ak, ssk = O.gen_auto_key_fft(r, ks.glwe_sk, P), O.gen_ssk_fft(r, ks.glwe_sk, P)
    eng.load_bootstrap_key(ks.bsk_fft)
    eng.load_automorphism_key(ak)
    eng.load_scheme_switch_key(ssk)
    triv = [np.zeros(P.lwe_n + 1, dtype=np.uint64) for _ in range(2)]
    triv[1][-1] = np.uint64(1 << 63)
    for bit in (0, 1):
        exp = O.circuit_bootstrap(triv[bit], ks.bsk_fft, ak, ssk, P)
        assert np.array_equal(eng.l1ggsw_constant(bit).view(np.float64), np.asarray(exp).view(np.float64)), bit
    a, b = random_glwe(5, 1, P.glwe_len)[0], random_glwe(6, 1, P.glwe_len)[0]
    g = spf_amd.FheCircuit(eng)
    na, nb = g.add_input(ValueKind.GLWE1, a), g.add_input(ValueKind.GLWE1, b)
    outs = []
    for bit in (0, 1):
        sel = g.add_trivial(ValueKind.GGSW1, bit)
        outs.append(g.add_output(g.add_op(FheOp.CMux, [sel, na, nb]), ValueKind.GLWE1))
        glev = g.add_trivial(ValueKind.GLEV1, bit)
        outs.append(g.add_output(g.add_op(FheOp.SchemeSwitch, [glev]), ValueKind.GGSW1))
    g.run()
    for bit in (0, 1):
        ggsw = eng.l1ggsw_constant(bit)
        assert np.array_equal(outs[2 * bit], eng.cmux(ggsw[None], a[None], b[None])[0])
        glev = np.zeros((P.cbs_count, 2, P.N), dtype=np.uint64)
        for j in range(P.cbs_count):
            glev[j, 1, 0] = np.uint64(bit << (64 - P.cbs_radix_log * (j + 1)))
        exp = O.scheme_switch_fft(glev.reshape(-1), ssk, P)
        assert np.array_equal(outs[2 * bit + 1].view(np.float64), np.asarray(exp).view(np.float64)), bit
    g.close()
    # keys loaded again => constants are rebuilt (a different key must not serve stale constants)
    eng.load_scheme_switch_key(ssk)
    assert np.array_equal(eng.l1ggsw_constant(1).view(np.float64),
                          np.asarray(O.circuit_bootstrap(triv[1], ks.bsk_fft, ak, ssk, P)).view(np.float64))


def _replay_rounds(eng, P):
    g = spf_amd.FheCircuit(eng)
    lwe1 = random_lwe_batch(31, 4, P.N * P.k)
    a, b = random_glwe(32, 4, P.glwe_len), random_glwe(33, 4, P.glwe_len)
    li = [g.add_input(ValueKind.LWE1, lwe1[i]) for i in range(4)]
    ai = [g.add_input(ValueKind.GLWE1, a[i]) for i in range(4)]
    bi = [g.add_input(ValueKind.GLWE1, b[i]) for i in range(4)]
    sel = [g.add_op(FheOp.CircuitBootstrap, [g.add_op(FheOp.KeyswitchL1toL0, [x])]) for x in li]
    m = [g.add_op(FheOp.CMux, [sel[i], ai[i], bi[i]]) for i in range(4)]
    top = g.add_op(FheOp.CMux, [sel[0], m[1], g.add_op(FheOp.Not, [m[2]])])
    outs = [g.add_output(x, ValueKind.GLWE1) for x in m + [top]]

    def expected():
        gg = eng.keyswitch_circuit_bootstrap(lwe1)
        mm = eng.cmux(gg, a, b)
        tt = eng.cmux(gg[:1], mm[1:2], eng.glwe_not(mm[2:3]))
        return list(mm) + [tt[0]]

    for rnd in range(5):
        if rnd:
            lwe1[...] = random_lwe_batch(40 + rnd, 4, P.N * P.k)    # same buffers, new contents
            a[...] = random_glwe(50 + rnd, 4, P.glwe_len)
        if rnd == 3:
            eng.circuit_bootstrap(random_lwe_batch(60, 64, P.lwe_n))  # grows the context's scratch: re-capture
        g.run()
        for got, exp in zip(outs, expected()):
            assert np.array_equal(got, exp), rnd
    g.close()


def test_graph_reruns_on_new_inputs(rig):
    """inputs are read from the callers' buffers at every run: new contents, new (correct) outputs"""
    ks, eng = rig
    _replay_rounds(eng, ks.params)


def test_replayed_hip_graph_equals_eager_runs():
    """SPF_GRAPH_CAPTURE=1 (read once per process, hence the child process): spf_graph_run captures the planned
    launch sequence as a hipGraph on its second run and replays it afterwards; the replay must give the eager
    results on new inputs, and a context scratch buffer that grows in between must force a re-capture."""
    import os
    import subprocess
    import sys
    code = ("import tests.test_gpu_graph as t\n"
            "from tests.util import keyset\n"
            "ks, eng = t._make_rig()\n"
            "t._replay_rounds(eng, ks.params)\nprint('replay ok')\n")
    env = dict(os.environ, SPF_GRAPH_CAPTURE="1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=root, env=env, timeout=600)
    assert r.returncode == 0 and "replay ok" in r.stdout, r.stderr[-3000:]


def test_encrypted_add_32_via_mux_circuits_adder():
    """BASELINE config 3 with the reference's OWN adder circuit: `mux_circuits::add::ripple_carry_adder(32, 32, false)`
    (one multiplexer per BDD node of every sum bit: 1 679 CMUX in 64 levels, rebuilt by spf_amd.mux_circuits) fed the way
    `add_circuit` feeds it (parasol_runtime/src/circuits/add.rs:10-32: inputs a0, b0, a1, b1, ... each L1 GLWE ->
    SampleExtract -> KeyswitchL1toL0 -> CircuitBootstrap), one gate graph, DEFAULT_128, decrypt == a + b."""
    import oracle as O
    from spf_amd.gate_pool import circuit_jobs_as_one_graph
    from spf_amd.mux_circuits import ripple_carry_adder
    from tests.util import keyset, to_engine_params
    ks = keyset(0x5EED0001, 637)
    P = ks.params
    r = O.Rng(0xADD33)
    eng = spf_amd.Engine(to_engine_params(P))
    eng.load_bootstrap_key(ks.bsk_fft)
    eng.load_keyswitch_key(ks.ksk)
    eng.load_automorphism_key(O.gen_auto_key_fft(r, ks.glwe_sk, P))
    eng.load_scheme_switch_key(O.gen_ssk_fft(r, ks.glwe_sk, P))
    adder = ripple_carry_adder(32, 32, False)
    assert adder.metrics() == {"mux_gates": 1679, "inputs": 64, "outputs": 33} and adder.depth() == 64
    a, b = 0xFFFF0F37, 0x9E3779B9
    cts = []
    for i in range(32):
        for bit in ((a >> i) & 1, (b >> i) & 1):
            m = np.zeros(P.N, dtype=np.uint64)
            m[0] = O.encode(bit, 1)
            cts.append(O.encrypt_glwe(r, ks.glwe_sk, m, P.N, P.k, P.glwe_std))
    g, outs = circuit_jobs_as_one_graph(eng, adder, np.stack(cts)[None])
    g.run()
    got = 0
    for i, o in enumerate(outs[0]):
        got |= O.decode(int(O.decrypt_glwe_raw(o, ks.glwe_sk, P.N, P.k)[0]), 1) << i
    g.close()
    assert got == a + b
