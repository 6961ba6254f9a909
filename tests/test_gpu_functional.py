"""Functional tests in the reference's own style — encrypt, run the operation on the GPU, decrypt,
compare plaintexts — at the full DEFAULT_128 parameter set, plus the noise measurement of
parasol_runtime/examples/op_noise/noise.rs:15-38 (normalised torus distance after asserting
decode equality).  Mirrors: `can_bootstrap_with_map` (programmable_bootstrapping.rs:709-789, all
eight 3-bit messages), `keyswitch_lwe` (lwe_keyswitch.rs:71-95), `can_cmux` / `can_circuit_bootstrap`
(crypto/evaluation.rs:277-337) and `can_and` of the circuit processor (circuit_processor/tests/mod.rs:
224-270).  The oracle only generates keys, encrypts and decrypts here."""
import numpy as np
import pytest

import oracle as O
import spf_amd
from spf_amd import FheOp, ValueKind
from tests.util import M64, keyset, to_engine_params

pytestmark = pytest.mark.gpu


def _torus_distance(a: int, b: int) -> float:
    d = (a - b) & M64
    return min(d, (1 << 64) - d) / 2.0 ** 64


@pytest.fixture(scope="module")
def full():
    ks = keyset(0x5EED0001, 637)
    P = ks.params
    r = O.Rng(0xF00D)
    eng = spf_amd.Engine(to_engine_params(P))
    eng.load_bootstrap_key(ks.bsk_fft)
    eng.load_keyswitch_key(ks.ksk)
    eng.load_automorphism_key(O.gen_auto_key_fft(r, ks.glwe_sk, P))
    eng.load_scheme_switch_key(O.gen_ssk_fft(r, ks.glwe_sk, P))
    return ks, eng


def test_can_bootstrap_with_map_all_3_bit_messages(full):
    ks, eng = full
    P = ks.params
    bits = 3
    f = lambda x: (x + 3) % 8                                       # noqa: E731  (the reference's map)
    lut = O.trivial_lut_glwe(O.generate_lut(P.N, [f], bits), P)
    msgs = list(range(8)) * 2
    lwe = np.stack([O.encrypt_lwe(O.Rng(900 + i), ks.lwe_sk, O.encode(m, bits + 1), P.lwe_std)
                    for i, m in enumerate(msgs)])
    out = eng.pbs_univariate(lwe, lut)                              # L0 LWE -> L1 LWE under the GLWE key
    worst = 0.0
    for i, m in enumerate(msgs):
        phase = O.decrypt_lwe_raw(out[i], ks.glwe_sk)
        # the input carries a padding bit, the LUT output does not (bootstrap_helper, :743-770)
        assert O.decode(phase, bits) == f(m), (i, m)
        worst = max(worst, _torus_distance(phase, O.encode(f(m), bits)))
    # bootstrap output noise: far inside the decoding radius 2^-4 of a 3-bit message space
    assert worst < 2.0 ** -12, worst


def test_keyswitch_decrypts_and_noise(full):
    ks, eng = full
    P = ks.params
    msgs = [0, 1] * 25                                               # 50 trials, as the reference
    lwe1 = np.stack([O.encrypt_lwe(O.Rng(1200 + i), ks.glwe_sk, O.encode(m, 1), P.glwe_std)
                     for i, m in enumerate(msgs)])
    out = eng.keyswitch_lwe_l1_lwe_l0(lwe1)
    worst = 0.0
    for i, m in enumerate(msgs):
        phase = O.decrypt_lwe_raw(out[i], ks.lwe_sk)
        assert O.decode(phase, 1) == m
        worst = max(worst, _torus_distance(phase, O.encode(m, 1)))
    # 2048 x 6 digit x key-row products at sigma_lwe = 7.25e-5: sigma ~ 0.01 of the torus; the
    # decoding radius of one bit is 0.25
    assert worst < 0.08, worst


def test_can_circuit_bootstrap_then_cmux_selects(full):
    ks, eng = full
    P = ks.params
    r = O.Rng(77)
    sel_bits = [0, 1, 1, 0]
    lwe0 = np.stack([O.encrypt_lwe(O.Rng(1500 + i), ks.lwe_sk, O.encode(b, 1), P.lwe_std)
                     for i, b in enumerate(sel_bits)])
    ggsw = eng.circuit_bootstrap(lwe0)
    m0 = np.zeros(P.N, dtype=np.uint64)
    m1 = np.zeros(P.N, dtype=np.uint64)
    m0[:8] = [O.encode(v & 1, 1) for v in range(8)]
    m1[:8] = [O.encode((v >> 1) & 1, 1) for v in range(8)]
    a = np.stack([O.encrypt_glwe(r, ks.glwe_sk, m0, P.N, P.k, P.glwe_std)] * len(sel_bits))
    b = np.stack([O.encrypt_glwe(r, ks.glwe_sk, m1, P.N, P.k, P.glwe_std)] * len(sel_bits))
    out = eng.cmux(ggsw, a, b)
    worst = 0.0
    for i, s in enumerate(sel_bits):
        dec = O.decrypt_glwe_raw(out[i], ks.glwe_sk, P.N, P.k)
        want = m1 if s else m0
        for c in range(8):
            assert O.decode(int(dec[c]), 1) == O.decode(int(want[c]), 1), (i, c)
            worst = max(worst, _torus_distance(int(dec[c]), int(want[c])))
    assert worst < 2.0 ** -8, worst


def test_can_and_as_a_gate_graph(full):
    ks, eng = full
    P = ks.params
    r = O.Rng(4242)
    a_bits, b_bits = [0, 1, 1, 0], [1, 0, 1, 0]
    g = spf_amd.FheCircuit(eng)

    def bit_to_ggsw(bit):
        m = np.full(P.N, O.encode(bit, 1), dtype=np.uint64)          # the reference fills every coefficient
        x = g.add_input(ValueKind.GLWE1, O.encrypt_glwe(r, ks.glwe_sk, m, P.N, P.k, P.glwe_std))
        x = g.add_op(FheOp.SampleExtract, [x], 0)
        x = g.add_op(FheOp.KeyswitchL1toL0, [x])
        return g.add_op(FheOp.CircuitBootstrap, [x])

    zero = g.add_trivial(ValueKind.GLWE1, 0)
    one = g.add_trivial(ValueKind.GLWE1, 1)
    outs = []
    for x, y in zip(a_bits, b_bits):
        sa, sb = bit_to_ggsw(x), bit_to_ggsw(y)
        inner = g.add_op(FheOp.CMux, [sb, zero, one])                # b ? 1 : 0
        outs.append(g.add_output(g.add_op(FheOp.CMux, [sa, zero, inner]), ValueKind.GLWE1))   # a ? b : 0
    g.run()
    for x, y, o in zip(a_bits, b_bits, outs):
        assert O.decode(int(O.decrypt_glwe_raw(o, ks.glwe_sk, P.N, P.k)[0]), 1) == (x & y)
    assert g.stats()["levels"] == 5
    g.close()
