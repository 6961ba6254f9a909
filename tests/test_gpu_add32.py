"""End-to-end on the GPU at DEFAULT_128: a 32-bit encrypted addition as a ripple-carry CMUX circuit
(BASELINE.json config 3 in spirit; the reference builds it with mux_circuits' BDD compiler,
parasol_runtime/src/circuits/add.rs:10-32).

Data flow, exactly the reference's (fhe_circuit.rs:473-494): every input bit arrives as an L1
GLWE ciphertext -> SampleExtract(0) -> KeyswitchL1toL0 -> CircuitBootstrap -> L1 GGSW selector;
the 64 conversions are mutually independent and run as ONE batch.  The adder itself is a chain of
CMUX gates over GLWE-encoded carries (`not` = add a trivial one, crypto/evaluation.rs:48-51).
Everything between encryption and decryption runs in the HIP library; the oracle only makes keys,
encrypts and decrypts."""
import numpy as np
import pytest

import oracle as O
import spf_amd
from tests.util import keyset, to_engine_params

pytestmark = pytest.mark.gpu


def _trivial(bit, P):
    g = np.zeros(P.glwe_len, dtype=np.uint64)
    g[P.N] = O.encode(bit, 1)
    return g


def test_encrypted_add_32():
    ks = keyset(0x5EED0001, 637)
    P = ks.params
    r = O.Rng(0xADD32)
    ak = O.gen_auto_key_fft(r, ks.glwe_sk, P)
    ssk = O.gen_ssk_fft(r, ks.glwe_sk, P)
    eng = spf_amd.Engine(to_engine_params(P))
    eng.load_bootstrap_key(ks.bsk_fft)
    eng.load_keyswitch_key(ks.ksk)
    eng.load_automorphism_key(ak)
    eng.load_scheme_switch_key(ssk)

    a, b = 0xDEADBEEF, 0x1234ABCD
    bits = [(a >> i) & 1 for i in range(32)] + [(b >> i) & 1 for i in range(32)]
    # inputs: L1 GLWE encryptions, the bit in coefficient 0 (PlaintextBits(1))
    glwe_in = []
    for bit in bits:
        m = np.zeros(P.N, dtype=np.uint64)
        m[0] = O.encode(bit, 1)
        glwe_in.append(O.encrypt_glwe(r, ks.glwe_sk, m, P.N, P.k, P.glwe_std))
    glwe_in = np.stack(glwe_in)

    # 64 x (SampleExtract -> KeyswitchL1toL0 -> CircuitBootstrap), batched
    lwe1 = eng.sample_extract_l1(glwe_in, 0)
    lwe0 = eng.keyswitch_lwe_l1_lwe_l0(lwe1)
    ggsw = eng.circuit_bootstrap(lwe0)                 # 64 x GGSW-FFT
    ga, gb = ggsw[:32], ggsw[32:]

    one = _trivial(1, P)
    zero = _trivial(0, P)
    carry = zero.copy()
    sums = []
    for i in range(32):
        ncarry = (carry + one).astype(np.uint64)       # KeylessEvaluation::not
        # first level, selector b_i:  [c, ~c] [~c, c] [0, c] [c, 1]
        d0 = np.stack([carry, ncarry, zero, carry])
        d1 = np.stack([ncarry, carry, carry, one])
        lvl1 = eng.cmux(np.stack([gb[i]] * 4), d0, d1)
        # second level, selector a_i:  sum = a ? (b ? c : ~c) : (b ? ~c : c);  carry' = a ? (b ? 1 : c) : (b ? c : 0)
        lvl2 = eng.cmux(np.stack([ga[i]] * 2), np.stack([lvl1[0], lvl1[2]]), np.stack([lvl1[1], lvl1[3]]))
        sums.append(lvl2[0])
        carry = lvl2[1]

    got = 0
    for i, s in enumerate(sums):
        got |= O.decode(int(O.decrypt_glwe_raw(s, ks.glwe_sk, P.N, P.k)[0]), 1) << i
    carry_out = O.decode(int(O.decrypt_glwe_raw(carry, ks.glwe_sk, P.N, P.k)[0]), 1)
    assert got == (a + b) & 0xFFFFFFFF
    assert carry_out == ((a + b) >> 32) & 1
