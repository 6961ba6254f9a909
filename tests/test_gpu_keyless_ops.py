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
