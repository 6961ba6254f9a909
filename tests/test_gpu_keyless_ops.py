"""glev_cmux and multiply_glwe_ggsw (KeylessEvaluation, crypto/evaluation.rs:86-123) against the oracle."""
import numpy as np
import pytest

import oracle as O
import spf_amd
from tests.util import random_glwe, to_engine_params

pytestmark = pytest.mark.gpu
P = O.DEFAULT_128.replace(lwe_n=1)


def _ggsw(seed, B):
    r = np.random.default_rng(seed)
    n = P.cbs_ggsw_fft_len
    return (r.standard_normal((B, n)) + 1j * r.standard_normal((B, n))) * 2.0 ** 58


def test_glev_cmux_parity():
    eng = spf_amd.Engine(to_engine_params(P))
    B = 3
    g = _ggsw(1, B)
    a = random_glwe(2, B * P.cbs_count, P.glwe_len).reshape(B, P.cbs_count, P.glwe_len)
    b = random_glwe(3, B * P.cbs_count, P.glwe_len).reshape(B, P.cbs_count, P.glwe_len)
    got = eng.glev_cmux(g, a, b).reshape(B, P.cbs_count, P.glwe_len)
    for i in range(B):
        for j in range(P.cbs_count):   # glev_cmux: cmux over each constituent GLWE (fft_ops.rs:211-219)
            exp = O.cmux(a[i, j], b[i, j], g[i], P.N, P.k, P.cbs_radix_log, P.cbs_count)
            assert np.array_equal(got[i, j], exp), (i, j)


def test_multiply_glwe_ggsw_parity():
    eng = spf_amd.Engine(to_engine_params(P))
    B = 4
    g = _ggsw(5, B)
    x = random_glwe(6, B, P.glwe_len)
    got = eng.multiply_glwe_ggsw(x, g)
    for i in range(B):
        fft = O.glwe_ggsw_mad(np.zeros(P.glwe_len // 2, dtype=np.complex128), x[i], g[i], P.N, P.k,
                              P.cbs_radix_log, P.cbs_count)
        exp = np.concatenate([O.poly_ifft(fft[:P.N // 2]), O.poly_ifft(fft[P.N // 2:])])
        assert np.array_equal(got[i], exp), i


# The three linear operations (crypto/evaluation.rs:47-66).  The reference's own tests for them are
# functional (`evaluation.rs` tests decrypt after not/xor; `can_rotate_*` in blind_rotation.rs:390-470);
# here the bar is bit equality with the oracle plus the algebraic properties those tests rely on.
def test_not_xor_parity_and_involution():
    eng = spf_amd.Engine(to_engine_params(P))
    B = 5
    a = random_glwe(11, B, P.glwe_len)
    b = random_glwe(12, B, P.glwe_len)
    n1 = eng.glwe_not(a)
    x = eng.glwe_xor(a, b)
    for i in range(B):
        assert np.array_equal(n1[i], O.glwe_not(a[i], P.N, P.k))
        assert np.array_equal(x[i], O.glwe_xor(a[i], b[i], P.N, P.k))
    assert np.array_equal(eng.glwe_not(n1), a)                       # 2 * 2^63 = 0 mod 2^64
    assert np.array_equal(eng.glwe_xor(b, a), x) and np.array_equal(x, a + b)   # commutes, wraps
    # only body coefficient 0 moves
    d = n1 - a
    assert d[:, P.N * P.k].tolist() == [1 << 63] * B and np.count_nonzero(d) == B


@pytest.mark.parametrize("n", [0, 1, 7, 2047, 2048, 2049, 4095, 4096, 4097 + 4096 * 3])
def test_mul_xn_parity(n):
    eng = spf_amd.Engine(to_engine_params(P))
    B = 3
    a = random_glwe(20 + n % 7, B, P.glwe_len)
    got = eng.glwe_mul_xn(a, n)
    for i in range(B):
        assert np.array_equal(got[i], O.glwe_mul_xn(a[i], n, P.N, P.k)), i
    # X^n * X^(2N - n) = 1
    back = eng.glwe_mul_xn(got, (2 * P.N - n % (2 * P.N)) % (2 * P.N))
    assert np.array_equal(back, a)


def test_evaluation_mirror_linear_ops():
    key = spf_amd.ComputeKey(bs_key=np.zeros((P.lwe_n, (P.k + 1) ** 2 * P.pbs_count * (P.N // 2)), dtype=np.complex128), ks_key=None)
    ev = spf_amd.Evaluation(key, to_engine_params(P))
    a = random_glwe(31, 1, P.glwe_len)[0]
    b = random_glwe(32, 1, P.glwe_len)[0]
    out = np.zeros_like(a)
    ev.not_(out, a)
    assert np.array_equal(out, O.glwe_not(a, P.N, P.k))
    ev.xor(out, a, b)
    assert np.array_equal(out, O.glwe_xor(a, b, P.N, P.k))
    ev.mul_xn(out, a, 5)
    assert np.array_equal(out, O.glwe_mul_xn(a, 5, P.N, P.k))


def test_batches_beyond_one_grid_slice():
    """The one-grid-row-per-ciphertext kernels run batches above 32 768 in slices (grid.y limit)."""
    eng = spf_amd.Engine(to_engine_params(P))
    B, h = 33000, 5
    glwe = (np.arange(B * P.glwe_len, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)).reshape(B, P.glwe_len)
    got = eng.sample_extract_l1(glwe, h)
    # sample_extract (glwe_ciphertext_ops.rs:31-76), vectorised over the batch
    N = P.N
    exp = np.empty((B, N + 1), dtype=np.uint64)
    exp[:, :h + 1] = glwe[:, h::-1][:, :h + 1]
    exp[:, h + 1:N] = np.uint64(0) - glwe[:, N - 1:h:-1]
    exp[:, N] = glwe[:, N + h]
    assert np.array_equal(got, exp)
    assert np.array_equal(got[B - 1], O.sample_extract(glwe[B - 1], h, P.N, P.k))
    del got, exp
    n1 = eng.glwe_not(glwe)
    assert np.array_equal(n1[:, N], glwe[:, N] + np.uint64(1 << 63)) and np.array_equal(n1[:, :N], glwe[:, :N])
