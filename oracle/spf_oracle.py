"""ctypes/numpy front end of the CPU oracle (oracle/spf_oracle.c).

TEST INFRASTRUCTURE ONLY — see oracle/spf_oracle.h for scope, citations and pinning status.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(native: bool = False) -> str:
    """Compile the oracle with gcc (no reference sources involved). Returns the .so path."""
    target = "libspf_oracle_native.so" if native else "libspf_oracle.so"
    subprocess.run(["make", "-s", "-C", _HERE, target], check=True)
    return os.path.join(_HERE, target)


def library_path() -> str:
    """Path of the built (portable) oracle library, for native tests that link it."""
    _load()
    return os.path.join(_HERE, "libspf_oracle.so")


def _load(native: bool = False):
    global _LIB
    if _LIB is not None and not native:
        return _LIB
    name = "libspf_oracle_native.so" if native else "libspf_oracle.so"
    path = os.path.join(_HERE, name)
    src = os.path.join(_HERE, "spf_oracle.c")
    if not os.path.exists(path) or (
        os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(path)
    ):
        build(native)
    lib = C.CDLL(path)
    _declare(lib)
    if not native:
        _LIB = lib
    return lib


u64 = C.c_uint64
u32 = C.c_uint32
sz = C.c_size_t
dbl = C.c_double
P = C.c_void_p


class _Rng(C.Structure):
    _fields_ = [("s", u64 * 4)]


class _C64(C.Structure):
    _fields_ = [("re", dbl), ("im", dbl)]


def _declare(lib):
    def f(name, res, *args):
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = list(args)

    f("spfo_modulus_switch", u64, u64, u32, u32, u32)
    f("spfo_lwe_modulus_switch", None, P, sz, u32, u32, u32)
    f("spfo_poly_mul_neg_monomial", None, P, sz, sz)
    f("spfo_poly_mul_pos_monomial", None, P, sz, sz)
    f("spfo_radix_round", u64, u64, u32, u32)
    f("spfo_radix_next_digit", u64, P, u32)
    f("spfo_decompose_poly", None, P, sz, u32, u32, P)
    f("spfo_poly_shr_round", None, P, P, sz, u32)
    f("spfo_poly_pow_k", None, P, P, sz, sz)
    f("spfo_sample_extract", None, P, P, sz, sz, sz)
    f("spfo_glwe_not", None, P, P, sz, sz)
    f("spfo_glwe_xor", None, P, P, P, sz, sz)
    f("spfo_glwe_mul_xn", None, P, P, sz, sz, sz)
    f("spfo_lwe_rotate", None, P, P, sz, u64)
    f("spfo_generate_lut", None, P, sz, P, sz, u32)
    f("spfo_generate_negacyclic_lut", None, P, sz, P, u32)
    f("spfo_fill_cbs_lut", None, P, sz, sz, u32, u32)
    f("spfo_keyswitch_lwe", None, P, P, P, sz, sz, u32, u32)
    f("spfo_twisted_fft_forward", None, P, sz, P)
    f("spfo_twisted_fft_reverse", None, P, sz, P)
    f("spfo_poly_fft", None, P, sz, P)
    f("spfo_poly_ifft", None, P, sz, P)
    f("spfo_f64_to_torus", u64, dbl)
    f("spfo_complex_mad", None, P, P, P, sz)
    f("spfo_set_mad_mode", None, C.c_int)
    f("spfo_get_mad_mode", C.c_int)
    f("spfo_fft1024", None, P, P, C.c_int)
    f("spfo_root_of_unity", _C64, u64, u64)
    f("spfo_negacyclic_mul_exact", None, P, P, P, sz)
    f("spfo_glwe_ggsw_mad", None, P, P, P, sz, sz, u32, u32)
    f("spfo_cmux", None, P, P, P, P, sz, sz, u32, u32)
    f("spfo_generalized_pbs", None, P, P, P, P, sz, sz, sz, u32, u32, u32, u32)
    f("spfo_pbs_univariate", None, P, P, P, P, sz, sz, sz, u32, u32)
    f("spfo_cbs_pbs", None, P, P, P, sz, sz, sz, u32, u32, u32, u32)
    f("spfo_rng_seed", None, P, u64)
    f("spfo_rng_next", u64, P)
    f("spfo_normal_torus", u64, P, dbl)
    f("spfo_gen_binary_key", None, P, P, sz)
    f("spfo_encrypt_lwe", None, P, P, P, sz, u64, dbl)
    f("spfo_decrypt_lwe_raw", u64, P, P, sz)
    f("spfo_encrypt_glwe", None, P, P, P, P, sz, sz, dbl)
    f("spfo_decrypt_glwe_raw", None, P, P, P, sz, sz)
    f("spfo_encrypt_ggsw_scalar", None, P, P, P, u64, sz, sz, u32, u32, dbl)
    f("spfo_ggsw_fft", None, P, P, sz, sz, u32)
    f("spfo_gen_bsk_fft", None, P, P, P, sz, P, sz, sz, u32, u32, dbl)
    f("spfo_gen_ksk", None, P, P, P, sz, P, sz, u32, u32, dbl)
    f("spfo_keyswitch_glwe_to_glwe", None, P, P, P, sz, sz, u32, u32)
    f("spfo_trace", None, P, P, P, sz, sz, u32, u32)
    f("spfo_mod_switch_trace_and_rotate", None, P, P, P, sz, sz, u32, u32, u32, u32)
    f("spfo_scheme_switch_fft", None, P, P, P, sz, sz, u32, u32, u32)
    f("spfo_circuit_bootstrap", None, P, P, P, P, P, sz, sz, sz, u32, u32, u32, u32, u32, u32, u32, u32)
    f("spfo_gen_glwe_ksk_fft", None, P, P, P, P, sz, sz, u32, u32, dbl)
    f("spfo_gen_auto_key_fft", None, P, P, P, sz, sz, u32, u32, dbl)
    f("spfo_gen_ssk_fft", None, P, P, P, sz, sz, u32, u32, dbl)
    f("spfo_encode", u64, u64, u32)
    f("spfo_decode", u64, u64, u32)
    f("spfo_bench_cbs_pbs", dbl, P, sz, P, sz, sz, sz, u32, u32, u32, u32, C.c_int, P)
    f("spfo_bench_generalized_pbs", dbl, P, sz, P, sz, P, sz, sz, sz, u32, u32, u32, u32, C.c_int, C.c_int, P)


def _p(a: np.ndarray):
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(P)


def _u(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.uint64)


def _c(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.complex128)


# ----------------------------------------------------------------------------- parameters


@dataclass(frozen=True)
class Params:
    """Mirror of parasol_runtime/src/params.rs:107-134 (DEFAULT_128) — only the fields the
    bootstrap / keyswitch path reads."""

    lwe_n: int = 637                 # sunscreen_tfhe/src/params.rs:219-222
    lwe_std: float = 7.25e-5
    N: int = 2048                    # sunscreen_tfhe/src/params.rs:258-264
    k: int = 1
    glwe_std: float = 7e-16
    pbs_radix_log: int = 16          # parasol_runtime/src/params.rs:114-117
    pbs_count: int = 2
    cbs_radix_log: int = 4           # :110-113
    cbs_count: int = 4
    ks_radix_log: int = 2            # :122-125
    ks_count: int = 6
    tr_radix_log: int = 7            # :130-133 (trace / automorphism keyswitch)
    tr_count: int = 6
    ss_radix_log: int = 3            # :126-129 (scheme switch)
    ss_count: int = 15

    @property
    def glwe_len(self) -> int:
        return (self.k + 1) * self.N

    @property
    def ggsw_fft_len(self) -> int:   # complex bins of one PBS-radix GGSW
        return (self.k + 1) * self.pbs_count * (self.k + 1) * (self.N // 2)

    @property
    def cbs_ggsw_fft_len(self) -> int:  # complex bins of one cbs-radix GGSW
        return (self.k + 1) * self.cbs_count * (self.k + 1) * (self.N // 2)

    @property
    def ak_fft_len(self) -> int:       # log2(N) GLWE keyswitch keys
        return (self.N.bit_length() - 1) * self.k * self.tr_count * (self.k + 1) * (self.N // 2)

    @property
    def ssk_fft_len(self) -> int:
        return (self.k * (self.k + 1) // 2) * self.ss_count * (self.k + 1) * (self.N // 2)

    def replace(self, **kw) -> "Params":
        d = dict(self.__dict__)
        d.update(kw)
        return Params(**d)


DEFAULT_128 = Params()


# ----------------------------------------------------------------------------- integer stages


def modulus_switch(x: int, log_chi: int, log_v: int, log_modulus: int) -> int:
    return int(_load().spfo_modulus_switch(x, log_chi, log_v, log_modulus))


def lwe_modulus_switch(ct, log_chi, log_v, log_modulus) -> np.ndarray:
    out = _u(ct).copy()
    _load().spfo_lwe_modulus_switch(_p(out), out.size, log_chi, log_v, log_modulus)
    return out


def poly_mul_neg_monomial(p, degree: int) -> np.ndarray:
    out = _u(p).copy()
    _load().spfo_poly_mul_neg_monomial(_p(out), out.size, degree)
    return out


def poly_mul_pos_monomial(p, degree: int) -> np.ndarray:
    out = _u(p).copy()
    _load().spfo_poly_mul_pos_monomial(_p(out), out.size, degree)
    return out


def radix_round(x: int, radix_log: int, count: int) -> int:
    return int(_load().spfo_radix_round(x, radix_log, count))


def decompose_poly(poly, radix_log: int, count: int) -> np.ndarray:
    poly = _u(poly)
    out = np.zeros((count, poly.size), dtype=np.uint64)
    _load().spfo_decompose_poly(_p(poly), poly.size, radix_log, count, _p(out))
    return out


def poly_shr_round(x, n: int) -> np.ndarray:
    x = _u(x)
    y = np.zeros_like(x)
    _load().spfo_poly_shr_round(_p(y), _p(x), x.size, n)
    return y


def poly_pow_k(p, k: int) -> np.ndarray:
    p = _u(p)
    out = np.zeros_like(p)
    _load().spfo_poly_pow_k(_p(out), _p(p), p.size, k)
    return out


def glwe_not(glwe, N: int, k: int) -> np.ndarray:
    glwe = _u(glwe)
    out = np.zeros_like(glwe)
    _load().spfo_glwe_not(_p(out), _p(glwe), N, k)
    return out


def glwe_xor(a, b, N: int, k: int) -> np.ndarray:
    a, b = _u(a), _u(b)
    out = np.zeros_like(a)
    _load().spfo_glwe_xor(_p(out), _p(a), _p(b), N, k)
    return out


def glwe_mul_xn(glwe, n: int, N: int, k: int) -> np.ndarray:
    glwe = _u(glwe)
    out = np.zeros_like(glwe)
    _load().spfo_glwe_mul_xn(_p(out), _p(glwe), n, N, k)
    return out


def sample_extract(glwe, h: int, N: int, k: int) -> np.ndarray:
    glwe = _u(glwe)
    out = np.zeros(k * N + 1, dtype=np.uint64)
    _load().spfo_sample_extract(_p(out), _p(glwe), h, N, k)
    return out


def generate_lut(N: int, maps, plaintext_bits: int) -> np.ndarray:
    """maps: list of python callables x -> y (as reference's generate_lut)."""
    p = 1 << plaintext_bits
    tab = _u([[m(x) for x in range(p)] for m in maps])
    out = np.zeros(N, dtype=np.uint64)
    _load().spfo_generate_lut(_p(out), N, _p(tab), len(maps), plaintext_bits)
    return out


def generate_negacyclic_lut(N: int, fmap, plaintext_bits: int) -> np.ndarray:
    p = 1 << plaintext_bits
    tab = _u([fmap(x) for x in range(p)])
    out = np.zeros(N, dtype=np.uint64)
    _load().spfo_generate_negacyclic_lut(_p(out), N, _p(tab), plaintext_bits)
    return out


def fill_cbs_lut(params: Params = DEFAULT_128) -> np.ndarray:
    out = np.zeros(params.glwe_len, dtype=np.uint64)
    _load().spfo_fill_cbs_lut(_p(out), params.N, params.k, params.cbs_radix_log, params.cbs_count)
    return out


def trivial_lut_glwe(lut_poly, params: Params = DEFAULT_128) -> np.ndarray:
    """UnivariateLookupTable::trivial_from_fn: mask zero, body = LUT polynomial."""
    out = np.zeros(params.glwe_len, dtype=np.uint64)
    out[params.k * params.N:] = _u(lut_poly)
    return out


def keyswitch_lwe(ct, ksk, n_in: int, n_out: int, radix_log: int, count: int) -> np.ndarray:
    ct, ksk = _u(ct), _u(ksk)
    assert ct.size == n_in + 1 and ksk.size == n_in * count * (n_out + 1)
    out = np.zeros(n_out + 1, dtype=np.uint64)
    _load().spfo_keyswitch_lwe(_p(out), _p(ct), _p(ksk), n_in, n_out, radix_log, count)
    return out


# ----------------------------------------------------------------------------- float stages


def twisted_fft_forward(x) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.zeros(x.size // 2, dtype=np.complex128)
    _load().spfo_twisted_fft_forward(_p(x), x.size, _p(out))
    return out


def twisted_fft_reverse(X) -> np.ndarray:
    X = _c(X)
    out = np.zeros(X.size * 2, dtype=np.float64)
    _load().spfo_twisted_fft_reverse(_p(X), X.size * 2, _p(out))
    return out


def poly_fft(poly) -> np.ndarray:
    poly = _u(poly)
    out = np.zeros(poly.size // 2, dtype=np.complex128)
    _load().spfo_poly_fft(_p(poly), poly.size, _p(out))
    return out


def poly_ifft(X) -> np.ndarray:
    X = _c(X)
    out = np.zeros(X.size * 2, dtype=np.uint64)
    _load().spfo_poly_ifft(_p(X), X.size * 2, _p(out))
    return out


def f64_to_torus(v: float) -> int:
    return int(_load().spfo_f64_to_torus(float(v)))


def set_mad_mode(mode: int):
    """1 = the reference's AVX-512 complex_mad (four FMAs; default, canonical for this build),
    0 = its scalar/AVX2 complex_mad (non-fused)."""
    _load().spfo_set_mad_mode(int(mode))


def get_mad_mode() -> int:
    return int(_load().spfo_get_mad_mode())


def complex_mad(c, a, b) -> np.ndarray:
    c, a, b = _c(c).copy(), _c(a), _c(b)
    _load().spfo_complex_mad(_p(c), _p(a), _p(b), c.size)
    return c


def fft1024(x, direction: int) -> np.ndarray:
    x = _c(x)
    assert x.size == 1024
    out = np.zeros(1024, dtype=np.complex128)
    _load().spfo_fft1024(_p(x), _p(out), direction)
    return out


def root_of_unity(num: int, den: int) -> complex:
    r = _load().spfo_root_of_unity(num, den)
    return complex(r.re, r.im)


def negacyclic_mul_exact(a, b) -> np.ndarray:
    a, b = _u(a), _u(b)
    c = np.zeros_like(a)
    _load().spfo_negacyclic_mul_exact(_p(c), _p(a), _p(b), a.size)
    return c


# ----------------------------------------------------------------------------- ciphertext ops


def glwe_ggsw_mad(c_fft, a_glwe, ggsw_fft, N, k, radix_log, count) -> np.ndarray:
    c_fft = _c(c_fft).copy()
    a_glwe, ggsw_fft = _u(a_glwe), _c(ggsw_fft)
    _load().spfo_glwe_ggsw_mad(_p(c_fft), _p(a_glwe), _p(ggsw_fft), N, k, radix_log, count)
    return c_fft


def cmux(d0, d1, ggsw_fft, N, k, radix_log, count) -> np.ndarray:
    d0, d1, ggsw_fft = _u(d0), _u(d1), _c(ggsw_fft)
    assert ggsw_fft.size == (k + 1) * count * (k + 1) * (N // 2)
    out = np.zeros((k + 1) * N, dtype=np.uint64)
    _load().spfo_cmux(_p(out), _p(d0), _p(d1), _p(ggsw_fft), N, k, radix_log, count)
    return out


def generalized_pbs(lwe_in, lut_glwe, bsk_fft, params: Params, log_chi=0, log_v=0) -> np.ndarray:
    lwe_in, lut_glwe, bsk_fft = _u(lwe_in), _u(lut_glwe), _c(bsk_fft)
    n = lwe_in.size - 1
    assert bsk_fft.size == n * params.ggsw_fft_len
    out = np.zeros(params.glwe_len, dtype=np.uint64)
    _load().spfo_generalized_pbs(_p(out), _p(lwe_in), _p(lut_glwe), _p(bsk_fft), n, params.N,
                                 params.k, params.pbs_radix_log, params.pbs_count, log_chi, log_v)
    return out


def pbs_univariate(lwe_in, lut_glwe, bsk_fft, params: Params) -> np.ndarray:
    lwe_in, lut_glwe, bsk_fft = _u(lwe_in), _u(lut_glwe), _c(bsk_fft)
    n = lwe_in.size - 1
    out = np.zeros(params.k * params.N + 1, dtype=np.uint64)
    _load().spfo_pbs_univariate(_p(out), _p(lwe_in), _p(lut_glwe), _p(bsk_fft), n, params.N,
                                params.k, params.pbs_radix_log, params.pbs_count)
    return out


def cbs_pbs(lwe_in, bsk_fft, params: Params) -> np.ndarray:
    lwe_in, bsk_fft = _u(lwe_in), _c(bsk_fft)
    n = lwe_in.size - 1
    out = np.zeros(params.glwe_len, dtype=np.uint64)
    _load().spfo_cbs_pbs(_p(out), _p(lwe_in), _p(bsk_fft), n, params.N, params.k,
                         params.pbs_radix_log, params.pbs_count, params.cbs_radix_log,
                         params.cbs_count)
    return out


def bench_cbs_pbs(lwe_batch, bsk_fft, params: Params, threads: int, native: bool = True):
    """cpu_baseline leg: returns (seconds, outputs)."""
    lib = _load(native=native)
    lwe_batch, bsk_fft = _u(lwe_batch), _c(bsk_fft)
    count, n = lwe_batch.shape[0], lwe_batch.shape[1] - 1
    out = np.zeros((count, params.glwe_len), dtype=np.uint64)
    secs = lib.spfo_bench_cbs_pbs(_p(lwe_batch), count, _p(bsk_fft), n, params.N, params.k,
                                  params.pbs_radix_log, params.pbs_count, params.cbs_radix_log,
                                  params.cbs_count, threads, _p(out))
    return float(secs), out


def bench_generalized_pbs(lwe_batch, lut, bsk_fft, params: Params, threads: int, log_chi: int = 0, log_v: int = 0,
                          extract: bool = False, native: bool = False):
    """Many independent generalized / univariate bootstraps on `threads` host threads: returns (seconds, outputs).
    `lut` is one GLWE (shared) or one per ciphertext."""
    lib = _load(native=native)
    lwe_batch, bsk_fft, lut = _u(lwe_batch), _c(bsk_fft), _u(lut)
    count, n = lwe_batch.shape[0], lwe_batch.shape[1] - 1
    stride = 0 if lut.ndim == 1 else params.glwe_len
    assert lut.size == (params.glwe_len if stride == 0 else count * params.glwe_len)
    out = np.zeros((count, params.k * params.N + 1 if extract else params.glwe_len), dtype=np.uint64)
    secs = lib.spfo_bench_generalized_pbs(_p(lwe_batch), count, _p(lut), stride, _p(bsk_fft), n, params.N, params.k,
                                          params.pbs_radix_log, params.pbs_count, log_chi, log_v, int(extract), threads,
                                          _p(out))
    return float(secs), out


# ----------------------------------------------------------------------------- keygen subset


class Rng:
    def __init__(self, seed: int):
        self._r = _Rng()
        _load().spfo_rng_seed(C.byref(self._r), seed)

    @property
    def ref(self):
        return C.byref(self._r)

    def next(self) -> int:
        return int(_load().spfo_rng_next(self.ref))

    def uniform(self, n: int) -> np.ndarray:
        return np.array([self.next() for _ in range(n)], dtype=np.uint64)


def encode(val: int, plain_bits: int) -> int:
    return int(_load().spfo_encode(val, plain_bits))


def decode(t: int, plain_bits: int) -> int:
    return int(_load().spfo_decode(int(t), plain_bits))


def gen_binary_key(rng: Rng, n: int) -> np.ndarray:
    key = np.zeros(n, dtype=np.uint64)
    _load().spfo_gen_binary_key(rng.ref, _p(key), n)
    return key


def encrypt_lwe(rng: Rng, sk, msg_torus: int, std: float) -> np.ndarray:
    sk = _u(sk)
    ct = np.zeros(sk.size + 1, dtype=np.uint64)
    _load().spfo_encrypt_lwe(rng.ref, _p(ct), _p(sk), sk.size, msg_torus, std)
    return ct


def decrypt_lwe_raw(ct, sk) -> int:
    ct, sk = _u(ct), _u(sk)
    return int(_load().spfo_decrypt_lwe_raw(_p(ct), _p(sk), sk.size))


def encrypt_glwe(rng: Rng, sk, msg, N: int, k: int, std: float) -> np.ndarray:
    sk, msg = _u(sk), _u(msg)
    ct = np.zeros((k + 1) * N, dtype=np.uint64)
    _load().spfo_encrypt_glwe(rng.ref, _p(ct), _p(sk), _p(msg), N, k, std)
    return ct


def decrypt_glwe_raw(ct, sk, N: int, k: int) -> np.ndarray:
    ct, sk = _u(ct), _u(sk)
    out = np.zeros(N, dtype=np.uint64)
    _load().spfo_decrypt_glwe_raw(_p(out), _p(ct), _p(sk), N, k)
    return out


def encrypt_ggsw_fft(rng: Rng, glwe_sk, bit: int, N, k, radix_log, count, std) -> np.ndarray:
    glwe_sk = _u(glwe_sk)
    polys = (k + 1) * count * (k + 1)
    ggsw = np.zeros(polys * N, dtype=np.uint64)
    _load().spfo_encrypt_ggsw_scalar(rng.ref, _p(ggsw), _p(glwe_sk), bit, N, k, radix_log, count,
                                     std)
    out = np.zeros(polys * (N // 2), dtype=np.complex128)
    _load().spfo_ggsw_fft(_p(out), _p(ggsw), N, k, count)
    return out


def gen_bsk_fft(rng: Rng, lwe_sk, glwe_sk, params: Params) -> np.ndarray:
    lwe_sk, glwe_sk = _u(lwe_sk), _u(glwe_sk)
    out = np.zeros(lwe_sk.size * params.ggsw_fft_len, dtype=np.complex128)
    _load().spfo_gen_bsk_fft(rng.ref, _p(out), _p(lwe_sk), lwe_sk.size, _p(glwe_sk), params.N,
                             params.k, params.pbs_radix_log, params.pbs_count, params.glwe_std)
    return out


def gen_ksk(rng: Rng, sk_in, sk_out, radix_log: int, count: int, std: float) -> np.ndarray:
    sk_in, sk_out = _u(sk_in), _u(sk_out)
    out = np.zeros(sk_in.size * count * (sk_out.size + 1), dtype=np.uint64)
    _load().spfo_gen_ksk(rng.ref, _p(out), _p(sk_in), sk_in.size, _p(sk_out), sk_out.size,
                         radix_log, count, std)
    return out


@dataclass
class KeySet:
    params: Params
    lwe_sk: np.ndarray      # n
    glwe_sk: np.ndarray     # k*N (== the L1 LWE key, sample-extract convention)
    bsk_fft: np.ndarray     # n * ggsw_fft_len complex128, reference layout (natural bin order)
    ksk: np.ndarray         # (k*N) * ks_count * (n+1) u64


def gen_keyset(seed: int, params: Params = DEFAULT_128, with_ksk: bool = True) -> KeySet:
    """Synthetic key set of SURVEY.md §8(d): binary LWE / GLWE keys, BSK = GGSW(s_i) FFT'd by
    the oracle, KSK L1->L0."""
    rng = Rng(seed)
    lwe_sk = gen_binary_key(rng, params.lwe_n)
    glwe_sk = gen_binary_key(rng, params.k * params.N)
    bsk = gen_bsk_fft(rng, lwe_sk, glwe_sk, params)
    ksk = (gen_ksk(rng, glwe_sk, lwe_sk, params.ks_radix_log, params.ks_count, params.lwe_std)
           if with_ksk else np.zeros(0, dtype=np.uint64))
    return KeySet(params, lwe_sk, glwe_sk, bsk, ksk)


def encrypt_bits_l0(seed: int, keys: KeySet, bits) -> np.ndarray:
    """Batch of L0 LWE encryptions of bits at 2^63 (PlaintextBits(1))."""
    rng = Rng(seed)
    return np.stack([encrypt_lwe(rng, keys.lwe_sk, encode(int(b), 1), keys.params.lwe_std)
                     for b in bits])


# ----------------------------------------------------------------------------- circuit-bootstrap tail


def keyswitch_glwe_to_glwe(ct, ksk_fft, N, k, radix_log, count) -> np.ndarray:
    ct, ksk_fft = _u(ct), _c(ksk_fft)
    out = np.zeros((k + 1) * N, dtype=np.uint64)
    _load().spfo_keyswitch_glwe_to_glwe(_p(out), _p(ct), _p(ksk_fft), N, k, radix_log, count)
    return out


def trace(x, ak_fft, params: Params = DEFAULT_128) -> np.ndarray:
    x, ak_fft = _u(x), _c(ak_fft)
    assert ak_fft.size == params.ak_fft_len
    out = np.zeros(params.glwe_len, dtype=np.uint64)
    _load().spfo_trace(_p(out), _p(x), _p(ak_fft), params.N, params.k, params.tr_radix_log, params.tr_count)
    return out


def mod_switch_trace_and_rotate(lo_noise_glwe, ak_fft, params: Params = DEFAULT_128) -> np.ndarray:
    g, ak_fft = _u(lo_noise_glwe), _c(ak_fft)
    out = np.zeros((params.cbs_count, params.glwe_len), dtype=np.uint64)
    _load().spfo_mod_switch_trace_and_rotate(_p(out), _p(g), _p(ak_fft), params.N, params.k, params.tr_radix_log,
                                             params.tr_count, params.cbs_radix_log, params.cbs_count)
    return out


def scheme_switch_fft(glev, ssk_fft, params: Params = DEFAULT_128) -> np.ndarray:
    glev, ssk_fft = _u(glev), _c(ssk_fft)
    assert ssk_fft.size == params.ssk_fft_len
    out = np.zeros(params.cbs_ggsw_fft_len, dtype=np.complex128)
    _load().spfo_scheme_switch_fft(_p(out), _p(glev), _p(ssk_fft), params.N, params.k, params.cbs_count,
                                   params.ss_radix_log, params.ss_count)
    return out


def circuit_bootstrap(lwe_in, bsk_fft, ak_fft, ssk_fft, params: Params = DEFAULT_128) -> np.ndarray:
    """Evaluation::circuit_bootstrap: L0 LWE -> L1 GGSW-FFT (cbs_radix shape)."""
    lwe_in, bsk_fft, ak_fft, ssk_fft = _u(lwe_in), _c(bsk_fft), _c(ak_fft), _c(ssk_fft)
    out = np.zeros(params.cbs_ggsw_fft_len, dtype=np.complex128)
    _load().spfo_circuit_bootstrap(_p(out), _p(lwe_in), _p(bsk_fft), _p(ak_fft), _p(ssk_fft), lwe_in.size - 1,
                                   params.N, params.k, params.pbs_radix_log, params.pbs_count, params.tr_radix_log,
                                   params.tr_count, params.ss_radix_log, params.ss_count, params.cbs_radix_log,
                                   params.cbs_count)
    return out


def gen_auto_key_fft(rng: Rng, glwe_sk, params: Params = DEFAULT_128) -> np.ndarray:
    glwe_sk = _u(glwe_sk)
    out = np.zeros(params.ak_fft_len, dtype=np.complex128)
    _load().spfo_gen_auto_key_fft(rng.ref, _p(out), _p(glwe_sk), params.N, params.k, params.tr_radix_log,
                                  params.tr_count, params.glwe_std)
    return out


def gen_ssk_fft(rng: Rng, glwe_sk, params: Params = DEFAULT_128) -> np.ndarray:
    glwe_sk = _u(glwe_sk)
    out = np.zeros(params.ssk_fft_len, dtype=np.complex128)
    _load().spfo_gen_ssk_fft(rng.ref, _p(out), _p(glwe_sk), params.N, params.k, params.ss_radix_log,
                             params.ss_count, params.glwe_std)
    return out
