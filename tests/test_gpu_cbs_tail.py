"""GPU parity of the circuit-bootstrap tail: homomorphic trace and scheme switch,
and the whole `Evaluation::circuit_bootstrap`, bit for bit against the oracle; plus the reference's
functional check (circuit_bootstrapping.rs:721-805): the produced GGSW drives a CMUX correctly."""
import numpy as np
import pytest

import oracle as O
import spf_amd
from tests.util import keyset, random_glwe, random_lwe_batch, to_engine_params

pytestmark = pytest.mark.gpu

SMALL_N = 12


@pytest.fixture(scope="module")
def tail():
    ks = keyset(0x5EED0001, SMALL_N)
    P = ks.params
    r = O.Rng(0x7A11)
    ak = O.gen_auto_key_fft(r, ks.glwe_sk, P)
    ssk = O.gen_ssk_fft(r, ks.glwe_sk, P)
    eng = spf_amd.Engine(to_engine_params(P))
    eng.load_bootstrap_key(ks.bsk_fft)
    eng.load_keyswitch_key(ks.ksk)
    eng.load_automorphism_key(ak)
    eng.load_scheme_switch_key(ssk)
    return ks, ak, ssk, eng


@pytest.mark.parametrize("B", [1, 3])
def test_mod_switch_trace_and_rotate_parity(tail, B):
    ks, ak, _, eng = tail
    P = ks.params
    glwe = random_glwe(60 + B, B, P.glwe_len)      # arbitrary torus words: parity needs no valid ciphertext
    got = eng.mod_switch_trace_and_rotate(glwe)
    for i in range(B):
        exp = O.mod_switch_trace_and_rotate(glwe[i], ak, P)
        assert np.array_equal(got[i], exp), i


@pytest.mark.parametrize("B", [1, 2])
def test_scheme_switch_parity(tail, B):
    ks, _, ssk, eng = tail
    P = ks.params
    glev = random_glwe(70 + B, B * P.cbs_count, P.glwe_len).reshape(B, P.cbs_count, P.glwe_len)
    got = eng.scheme_switch(glev)
    for i in range(B):
        exp = O.scheme_switch_fft(glev[i], ssk, P)
        assert np.array_equal(got[i].view(np.float64), exp.view(np.float64)), i


def test_circuit_bootstrap_parity_and_cmux_select(tail):
    ks, ak, ssk, eng = tail
    P = ks.params
    bits = [0, 1, 1, 0, 1]
    lwe = np.stack([O.encrypt_lwe(O.Rng(800 + i), ks.lwe_sk, O.encode(b, 1), P.lwe_std) for i, b in enumerate(bits)])
    got = eng.circuit_bootstrap(lwe)
    rng = O.Rng(17)
    msgs = [np.array([O.encode(int(v), 3) for v in np.random.default_rng(s).integers(0, 8, P.N)], dtype=np.uint64)
            for s in (1, 2)]
    d = [O.encrypt_glwe(rng, ks.glwe_sk, m, P.N, P.k, P.glwe_std) for m in msgs]
    for i, b in enumerate(bits):
        exp = O.circuit_bootstrap(lwe[i], ks.bsk_fft, ak, ssk, P)
        assert np.array_equal(got[i].view(np.float64), exp.view(np.float64)), i
    # the GGSWs out of the GPU circuit bootstrap select in the GPU CMUX
    a = np.stack([d[0]] * len(bits))
    b = np.stack([d[1]] * len(bits))
    sel = eng.cmux(got, a, b)
    for i, bit in enumerate(bits):
        dec = [O.decode(int(v), 3) for v in O.decrypt_glwe_raw(sel[i], ks.glwe_sk, P.N, P.k)]
        assert dec == [O.decode(int(v), 3) for v in msgs[bit]], i


def test_circuit_bootstrap_random_words(tail):
    ks, ak, ssk, eng = tail
    lwe = random_lwe_batch(5, 2, SMALL_N)
    got = eng.circuit_bootstrap(lwe)
    for i in range(2):
        exp = O.circuit_bootstrap(lwe[i], ks.bsk_fft, ak, ssk, ks.params)
        assert np.array_equal(got[i].view(np.float64), exp.view(np.float64)), i


def test_tail_needs_its_keys():
    ks = keyset(0x5EED0001, SMALL_N)
    eng = spf_amd.Engine(to_engine_params(ks.params))
    eng.load_bootstrap_key(ks.bsk_fft)
    with pytest.raises(spf_amd.SpfError):
        eng.circuit_bootstrap(random_lwe_batch(1, 1, SMALL_N))
