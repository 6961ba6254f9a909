"""BASELINE config 5's building block on the GPU at DEFAULT_128: an encrypted 8 x 8 -> 16-bit multiplication through
the REFERENCE's multiplier block (mux_circuits' `unsigned_multiplier(8, 8)`, 3 228 CMUX gates in 126 levels, read
from its bincode blob) lowered into one gate graph exactly as `mul_impl` feeds it (circuits/mul.rs:104-117): per
input bit L1 GLWE -> SampleExtract -> KeyswitchL1toL0 -> CircuitBootstrap -> selector of the block's CMUX tree.
Two multiplications in ONE graph (levels batch across jobs); the oracle only makes keys, encrypts and decrypts.
(32 x 32 in the reference = four 16 x 16 blocks + a reduction circuit its BDD compiler generates at run time.)"""
import os

import numpy as np
import pytest

import oracle as O
import spf_amd
from spf_amd.gate_pool import multiply_jobs_as_one_graph, run_sharded
from spf_amd.mux_circuits import parse_mux_circuit
from tests.util import keyset, to_engine_params

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "spf_amd", "data", "mux_multiplier_n8_m8.bincode")


def test_encrypted_multiply_8x8_through_the_reference_block():
    ks = keyset(0x5EED0001, 637)
    P = ks.params
    r = O.Rng(0x3A5)
    eng = spf_amd.Engine(to_engine_params(P))
    eng.load_bootstrap_key(ks.bsk_fft)
    eng.load_keyswitch_key(ks.ksk)
    eng.load_automorphism_key(O.gen_auto_key_fft(r, ks.glwe_sk, P))
    eng.load_scheme_switch_key(O.gen_ssk_fft(r, ks.glwe_sk, P))
    circuit = parse_mux_circuit(open(GOLDEN, "rb").read())
    jobs = [(0xB7, 0x5D), (255, 255)]

    def run_batch(mine):
        cts = []
        for a, b in mine:
            for bit in [(a >> i) & 1 for i in range(8)] + [(b >> i) & 1 for i in range(8)]:
                m = np.zeros(P.N, dtype=np.uint64)
                m[0] = O.encode(bit, 1)
                cts.append(O.encrypt_glwe(r, ks.glwe_sk, m, P.N, P.k, P.glwe_std))
        g, outs = multiply_jobs_as_one_graph(eng, circuit, np.stack(cts).reshape(len(mine), 16, -1))
        g.run()
        st = g.stats()
        assert st["levels"] == 3 + 126 and st["nodes"] >= len(mine) * (16 * 4 + 3228)
        res = [np.stack(o) for o in outs]
        g.close()
        return res

    results = run_sharded(jobs, [3228] * len(jobs), 0, 1, run_batch)
    for (a, b), out in zip(jobs, results):
        got = 0
        for i in range(16):
            got |= O.decode(int(O.decrypt_glwe_raw(out[i], ks.glwe_sk, P.N, P.k)[0]), 1) << i
        assert got == a * b, (a, b, got)


def test_config5_encrypted_multiply_32x32():
    """BASELINE config 5's circuit on ONE GPU at DEFAULT_128: 32 x 32 -> 64 bits exactly as `append_uint_multiply`
    builds it (circuits/mul.rs:75-200): 64 input bits -> GGSW, four 16 x 16 multiplier blocks (29 500 CMUX, 510 levels
    each), 128 partial-product bits converted back to GGSW (SampleExtract -> KeyswitchL1toL0 -> CircuitBootstrap), the
    gradeschool reduction circuit; ~127 k CMUX and 192 circuit bootstraps in one gate graph.  Decrypt == a * b."""
    from spf_amd import FheCircuit, ValueKind
    from spf_amd.mux_circuits import GraphBuilder, append_uint_multiply
    ks = keyset(0x5EED0001, 637)
    P = ks.params
    r = O.Rng(0x32A32)
    eng = spf_amd.Engine(to_engine_params(P))
    eng.load_bootstrap_key(ks.bsk_fft)
    eng.load_keyswitch_key(ks.ksk)
    eng.load_automorphism_key(O.gen_auto_key_fft(r, ks.glwe_sk, P))
    eng.load_scheme_switch_key(O.gen_ssk_fft(r, ks.glwe_sk, P))
    blk16 = parse_mux_circuit(open(GOLDEN.replace("n8_m8", "n16_m16"), "rb").read())
    a, b = 0xC0FFEE11, 0x9E3779B9
    g = FheCircuit(eng)
    builder = GraphBuilder(g)
    sel = []
    for bit in [(a >> i) & 1 for i in range(32)] + [(b >> i) & 1 for i in range(32)]:
        m = np.zeros(P.N, dtype=np.uint64)
        m[0] = O.encode(bit, 1)
        x = g.add_input(ValueKind.GLWE1, O.encrypt_glwe(r, ks.glwe_sk, m, P.N, P.k, P.glwe_std))
        sel.append(builder.to_ggsw(x))
    prod = append_uint_multiply(builder, sel[:32], sel[32:], lambda n, m: {(16, 16): blk16}[(n, m)])
    outs = [g.add_output(n, ValueKind.GLWE1) for n in prod]
    g.run()
    st = g.stats()
    assert st["nodes"] > 127000
    got = 0
    for i, o in enumerate(outs):
        got |= O.decode(int(O.decrypt_glwe_raw(o, ks.glwe_sk, P.N, P.k)[0]), 1) << i
    g.close()
    assert got == a * b, (hex(got), hex(a * b))
