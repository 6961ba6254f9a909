"""Pins the CPU oracle against every known-answer test the reference holds for the
bootstrap / keyswitch path, plus the oracle's own second opinions
(exact-integer negacyclic product, numpy DFT) and functional decrypt-equality — the way the
reference tests itself (e.g. programmable_bootstrapping.rs:709-789, lwe_keyswitch.rs:71-95).
CPU only."""
import json
import os

import numpy as np
import pytest

import oracle as O

M64 = (1 << 64) - 1


def _h(s):
    return int(s, 16)


def _s(vals):
    """signed python ints -> u64 array (wrapping_neg for negatives)"""
    return np.array([v & M64 for v in vals], dtype=np.uint64)


@pytest.fixture(scope="module")
def kats(golden_dir):
    with open(os.path.join(golden_dir, "reference_kats.json")) as f:
        return json.load(f)


def test_can_modulus_switch(kats):
    k = kats["modulus_switch"]
    for c in k["cases"]:
        got = O.modulus_switch(_h(k["x"]), c["log_chi"], c["log_v"], c["log_modulus"])
        assert got == _h(c["expect"])


def test_can_round_values(kats):
    for c in kats["radix_round"]["cases"]:
        assert O.radix_round(_h(c["x"]), c["radix_log"], c["count"]) == _h(c["expect"])


def test_can_decompose(kats):
    for c in kats["decompose"]["cases"]:
        poly = np.array([O.encode(v, b) for v, b in c["poly_encode"]], dtype=np.uint64)
        got = O.decompose_poly(poly, c["radix_log"], c["count"])
        exp = np.stack([_s(row) for row in c["digits"]])
        assert np.array_equal(got, exp)


def test_can_decompose_recompose():
    # property test of radix.rs:286-340
    rng = np.random.default_rng(11)
    for _ in range(50):
        radix = int(rng.integers(1, 8))
        while True:
            count = int(rng.integers(1, 8))
            if radix * count < 64:
                break
        x = rng.integers(0, 1 << 64, size=8, dtype=np.uint64)
        lsb = 64 - radix * count
        expected = []
        for c in x.tolist():
            rnd = (c >> (lsb - 1)) & 1
            mask = (M64 << lsb) & M64
            expected.append(((c & mask) + (rnd << lsb)) & M64)
        digits = O.decompose_poly(x, radix, count)
        res = [0] * 8
        cur = 1 << lsb
        for j in range(count):
            for i in range(8):
                res[i] = (res[i] + int(digits[j, i]) * cur) & M64
            cur = (cur << radix) & M64
        assert res == expected


def test_can_negacyclic_conv(kats):
    k = kats["negacyclic_conv"]
    y = O.twisted_fft_forward(np.array(k["x"]))
    out = O.twisted_fft_reverse(y * y)
    assert out.tolist() == k["expect"]


def test_can_roundtrip_negacyclic_fft(kats):
    k = kats["negacyclic_roundtrip"]
    x = np.array(k["x"])
    back = O.twisted_fft_reverse(O.twisted_fft_forward(x))
    assert np.all(np.abs(back - x) < k["tol"])


@pytest.mark.parametrize("which", ["pos_monomial", "neg_monomial"])
def test_monomial_rotation(kats, which):
    k = kats[which]
    fn = O.poly_mul_pos_monomial if which == "pos_monomial" else O.poly_mul_neg_monomial
    for deg, exp in k["cases"].items():
        assert np.array_equal(fn(_s(k["p"]), int(deg)), _s(exp)), (which, deg)


def test_glwe_linear_ops_follow_the_polynomial_kats(kats):
    """KeylessEvaluation::{not, xor, mul_xn} (crypto/evaluation.rs:47-66): mul_xn is the monomial KAT applied
    to mask and body; not adds the trivial one (body coefficient 0 += 2^63); xor is the wrapping sum."""
    k = kats["pos_monomial"]
    p = _s(k["p"])
    N = p.size
    glwe = np.concatenate([p, p[::-1].copy()])
    for deg, exp in k["cases"].items():
        got = O.glwe_mul_xn(glwe, int(deg), N, 1)
        assert np.array_equal(got[:N], _s(exp)), deg
        assert np.array_equal(got[N:], O.poly_mul_pos_monomial(p[::-1].copy(), int(deg))), deg
    n1 = O.glwe_not(glwe, N, 1)
    assert n1[N] == (glwe[N] + (1 << 63)) & M64 and np.array_equal(np.delete(n1, N), np.delete(glwe, N))
    assert np.array_equal(O.glwe_not(n1, N, 1), glwe)
    assert np.array_equal(O.glwe_xor(glwe, n1, N, 1), glwe + n1)


def test_can_polynomial_pow_k(kats):
    k = kats["poly_pow_k"]
    p = np.zeros(k["N"], dtype=np.uint64)
    for i, v in k["in"].items():
        p[int(i)] = v
    exp = np.zeros(k["N"], dtype=np.uint64)
    for i, v in k["out"].items():
        exp[int(i)] = v & M64
    assert np.array_equal(O.poly_pow_k(p, k["k"]), exp)


def test_can_polynomial_shift_round(kats):
    k = kats["poly_shr_round"]
    assert O.poly_shr_round(k["x"], k["n"]).tolist() == k["expect"]


def test_can_generate_negacyclic_lut(kats):
    k = kats["negacyclic_lut_formula"]
    N, bits = k["N"], k["plaintext_bits"]
    p = 1 << bits
    got = O.generate_negacyclic_lut(N, lambda x: x, bits)

    def div_rounded(a, b):  # sunscreen_tfhe/src/math/basic.rs:11-19
        q, r = divmod(a, b)
        return q + 1 if r >= b // 2 else q

    exp = [((div_rounded(p * j, 2 * N) % p) << (64 - bits)) & M64 for j in range(N)]
    assert got.tolist() == exp


def test_complex_mad_matches_definition():
    # simd/x86_64/mod.rs:231-251 (can_scalar_mad_complex_f64_slice): SIMD == `c += a*b` exactly
    rng = np.random.default_rng(5)
    mk = lambda: (rng.integers(0, 1 << 64, 16, dtype=np.uint64).astype(np.float64)
                  + 1j * rng.integers(0, 1 << 64, 16, dtype=np.uint64).astype(np.float64))
    a, b, c = mk(), mk(), mk()
    assert O.get_mad_mode() == 1     # the build's canonical order = the reference's AVX-512 path
    try:
        # mode 0: scalar / AVX2 path, `*c += a * b`, nothing fused (simd/scalar.rs:12-16)
        O.set_mad_mode(0)
        got = O.complex_mad(c, a, b)
        exp = np.array([complex(c[i].real + (a[i].real * b[i].real - a[i].imag * b[i].imag),
                                c[i].imag + (a[i].real * b[i].imag + a[i].imag * b[i].real))
                        for i in range(16)])
        assert np.array_equal(got, exp)
    finally:
        O.set_mad_mode(1)
    # mode 1: AVX-512 path, four FMAs in this order (simd/x86_64/avx512.rs:54-57).  Exact
    # rational arithmetic gives the correctly-rounded fma the asm performs.
    from fractions import Fraction as F

    def fma(x, y, z):
        return float(F(x) * F(y) + F(z))   # float(Fraction) rounds to nearest-even

    got = O.complex_mad(c, a, b)
    for i in range(16):
        re = fma(a[i].real, b[i].real, c[i].real)
        im = fma(a[i].real, b[i].imag, c[i].imag)
        re = fma(-a[i].imag, b[i].imag, re)
        im = fma(a[i].imag, b[i].real, im)
        assert (got[i].real, got[i].imag) == (re, im)


# ------------------------------------------------------------------ second opinions on the FFT


def test_fft1024_against_numpy():
    rng = np.random.default_rng(0)
    a = rng.standard_normal(1024) + 1j * rng.standard_normal(1024)
    f = O.fft1024(a, +1)
    assert np.abs(f - np.fft.fft(a)).max() < 1e-12 * np.abs(f).max()
    g = O.fft1024(a, -1)
    assert np.abs(g - np.fft.ifft(a) * 1024).max() < 1e-12 * np.abs(g).max()


def test_twiddle_symmetry_exact():
    # the table definition: conjugate / quarter-turn symmetries hold bit-exactly
    for den in (64, 512, 1024, 4096):
        for j in range(0, den, max(1, den // 128)):
            w = O.root_of_unity(j, den)
            wc = O.root_of_unity(den - j, den)
            assert (w.real, w.imag) == (wc.real, -wc.imag) or j == 0
            wq = O.root_of_unity(j + den // 4, den)
            assert (wq.real, wq.imag) == (-w.imag, w.real)
        assert O.root_of_unity(0, den) == 1 + 0j
        assert O.root_of_unity(den // 4, den) == 1j


def test_negacyclic_product_exact_when_small():
    # FFT path == exact u64 negacyclic product while the result stays below 2^53
    rng = np.random.default_rng(1)
    N = 2048
    p = rng.integers(-(1 << 15), 1 << 15, N).astype(np.int64).astype(np.uint64)
    q = rng.integers(0, 1 << 20, N).astype(np.uint64)
    got = O.poly_ifft(O.poly_fft(p) * O.poly_fft(q))
    assert np.array_equal(got, O.negacyclic_mul_exact(p, q))


def test_f64_to_torus_edges():
    # scalar.rs:85-118 + `as i64` saturation (torus.rs:177-192)
    assert O.f64_to_torus(0.0) == 0
    assert O.f64_to_torus(-1.0) == M64
    assert O.f64_to_torus(2.0 ** 63) == 1 << 63          # >= q/2 -> -2^63 -> 0x8000..
    assert O.f64_to_torus(-(2.0 ** 63)) == (1 << 63) - 1  # <= -q/2 -> +2^63 -> saturates
    assert O.f64_to_torus(2.0 ** 64) == 0
    assert O.f64_to_torus(2.0 ** 64 + 4096.0) == 4096
    assert O.f64_to_torus(-(2.0 ** 70) - 2.0 ** 30) == (-(1 << 30)) & M64
    assert O.f64_to_torus(3 * 2.0 ** 63) == 1 << 63
    assert O.f64_to_torus(-3 * 2.0 ** 63) == (1 << 63) - 1


# ------------------------------------------------------------------ functional (decrypt-equal)

SMALL = O.DEFAULT_128.replace(lwe_n=12)


@pytest.fixture(scope="module")
def small_keys():
    return O.gen_keyset(0x5EED0001, SMALL)


def test_can_bootstrap_identity(small_keys):
    # mirrors programmable_bootstrapping.rs:709-748 (can_bootstrap) with 1 message bit + padding
    P = SMALL
    lut = O.trivial_lut_glwe(O.generate_lut(P.N, [lambda x: (x + 1) % 2], 1), P)
    for msg in (0, 1):
        ct = O.encrypt_lwe(O.Rng(40 + msg), small_keys.lwe_sk, O.encode(msg, 2), P.lwe_std)
        out = O.pbs_univariate(ct, lut, small_keys.bsk_fft, P)
        assert O.decode(O.decrypt_lwe_raw(out, small_keys.glwe_sk), 1) == (msg + 1) % 2


def test_cbs_pbs_levels(small_keys):
    # circuit_bootstrapping.rs:387-427: coefficient i < 4 holds (2b-1) * 2^(64-(4(i+1)+1))
    P = SMALL
    for bit in (0, 1):
        ct = O.encrypt_lwe(O.Rng(50 + bit), small_keys.lwe_sk, O.encode(bit, 1), P.lwe_std)
        glwe = O.cbs_pbs(ct, small_keys.bsk_fft, P)
        m = O.decrypt_glwe_raw(glwe, small_keys.glwe_sk, P.N, P.k)
        for i in range(P.cbs_count):
            mag = 1 << (64 - (P.cbs_radix_log * (i + 1) + 1))
            exp = mag if bit else (-mag) & M64
            err = (int(m[i]) - exp + (1 << 63)) % (1 << 64) - (1 << 63)
            assert abs(err) < mag // 4, (bit, i, hex(int(m[i])))


def test_keyswitch_lwe(small_keys):
    # lwe_keyswitch.rs:71-95
    P = SMALL
    for t in range(8):
        bit = t & 1
        ct1 = O.encrypt_lwe(O.Rng(90 + t), small_keys.glwe_sk, O.encode(bit, 1), P.glwe_std)
        ct0 = O.keyswitch_lwe(ct1, small_keys.ksk, P.N * P.k, P.lwe_n, P.ks_radix_log, P.ks_count)
        assert O.decode(O.decrypt_lwe_raw(ct0, small_keys.lwe_sk), 1) == bit


def test_cmux_selects(small_keys):
    # fft_ops.rs:537-619 (can_cmux_fft) at the PBS shape
    P = SMALL
    rng = O.Rng(77)
    msgs = [np.array([O.encode(int(v), 3) for v in np.random.default_rng(s).integers(0, 8, P.N)],
                     dtype=np.uint64) for s in (1, 2)]
    d = [O.encrypt_glwe(rng, small_keys.glwe_sk, m, P.N, P.k, P.glwe_std) for m in msgs]
    for sel in (0, 1):
        g = O.encrypt_ggsw_fft(rng, small_keys.glwe_sk, sel, P.N, P.k, P.pbs_radix_log,
                               P.pbs_count, P.glwe_std)
        out = O.cmux(d[0], d[1], g, P.N, P.k, P.pbs_radix_log, P.pbs_count)
        dec = O.decrypt_glwe_raw(out, small_keys.glwe_sk, P.N, P.k)
        got = [O.decode(int(v), 3) for v in dec]
        assert got == [O.decode(int(v), 3) for v in msgs[sel]]


# ------------------------------------------------------------------ circuit-bootstrap tail (§8 f2)


@pytest.fixture(scope="module")
def tail_keys(small_keys):
    r = O.Rng(0x7A11)
    return O.gen_auto_key_fft(r, small_keys.glwe_sk, SMALL), O.gen_ssk_fft(r, small_keys.glwe_sk, SMALL)


def test_trace_keeps_only_the_constant_term(small_keys, tail_keys):
    # ops/automorphisms/mod.rs tests: trace zeroes every coefficient but the constant one and
    # multiplies it by N; pre-dividing by N (shr_round by log2 N) makes it the identity there
    P = SMALL
    msg = np.array([O.encode(int(v), 4) for v in np.random.default_rng(3).integers(0, 16, P.N)], dtype=np.uint64)
    ct = O.encrypt_glwe(O.Rng(1), small_keys.glwe_sk, msg, P.N, P.k, P.glwe_std)
    tr = O.trace(O.poly_shr_round(ct, 11), tail_keys[0], P)
    dec = [O.decode(int(v), 4) for v in O.decrypt_glwe_raw(tr, small_keys.glwe_sk, P.N, P.k)]
    assert dec[0] == O.decode(int(msg[0]), 4) and not any(dec[1:])


def test_can_circuit_bootstrap_via_trace_ss(small_keys, tail_keys):
    # circuit_bootstrapping.rs:721-805 / evaluation.rs:277-300: the GGSW a circuit bootstrap
    # produces must drive a CMUX like a fresh GGSW encryption of the same bit
    P = SMALL
    ak, ssk = tail_keys
    rng = O.Rng(9)
    m = [np.array([O.encode(int(v), 3) for v in np.random.default_rng(s).integers(0, 8, P.N)], dtype=np.uint64)
         for s in (1, 2)]
    d = [O.encrypt_glwe(rng, small_keys.glwe_sk, x, P.N, P.k, P.glwe_std) for x in m]
    for bit in (0, 1):
        lwe = O.encrypt_lwe(O.Rng(30 + bit), small_keys.lwe_sk, O.encode(bit, 1), P.lwe_std)
        g = O.circuit_bootstrap(lwe, small_keys.bsk_fft, ak, ssk, P)
        out = O.cmux(d[0], d[1], g, P.N, P.k, P.cbs_radix_log, P.cbs_count)
        dec = [O.decode(int(v), 3) for v in O.decrypt_glwe_raw(out, small_keys.glwe_sk, P.N, P.k)]
        assert dec == [O.decode(int(v), 3) for v in m[bit]]


def test_external_product_fft_error_at_cryptographic_magnitude():
    """SURVEY §8(c)(ii) on the CPU side: a one-step blind rotation (= one GGSW (x) GLWE external product
    at the PBS shape) with uniform 64-bit GGSW rows, against the exact integer negacyclic product.  The
    f64 FFT round trip must stay at the reference's error scale (~2^-30 of the torus); the same check
    runs against the HIP output in tests/test_gpu_configs.py."""
    P = O.DEFAULT_128.replace(lwe_n=1)
    rng = np.random.default_rng(0xE47)
    N = P.N
    G = rng.integers(0, 1 << 64, (2, 2, 2, N), dtype=np.uint64)
    bsk = np.stack([O.poly_fft(G[p, lvl, q]) for p in range(2) for lvl in range(2) for q in range(2)]).reshape(-1)
    a_t = 37
    lwe = np.array([np.uint64(a_t << 52), 0], dtype=np.uint64)
    d0 = rng.integers(0, 1 << 64, 2 * N, dtype=np.uint64)
    got = O.generalized_pbs(lwe, d0, bsk, P)
    rot = np.concatenate([O.poly_mul_pos_monomial(d0[:N], a_t), O.poly_mul_pos_monomial(d0[N:], a_t)])
    diff = rot - d0
    exact = d0.copy()
    for p in range(2):
        digs = O.decompose_poly(diff[p * N:(p + 1) * N], P.pbs_radix_log, P.pbs_count)
        for j in range(2):
            for q in range(2):
                exact[q * N:(q + 1) * N] += O.negacyclic_mul_exact(digs[j], G[p, 1 - j, q])
    dist = np.abs((got - exact).astype(np.int64).astype(np.float64)) / 2.0 ** 64
    assert 0.0 < dist.max() < 2.0 ** -26
    assert np.sqrt((dist ** 2).mean()) < 2.0 ** -29
