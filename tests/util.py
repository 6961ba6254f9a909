"""Shared helpers for the parity tests (test infrastructure; may import the oracle)."""
import functools

import numpy as np

import oracle as O
import spf_amd

M64 = (1 << 64) - 1


def to_engine_params(p: O.Params) -> spf_amd.Params:
    return spf_amd.Params(lwe_dimension=p.lwe_n, polynomial_degree=p.N, glwe_size=p.k,
                          pbs_radix_log=p.pbs_radix_log, pbs_radix_count=p.pbs_count,
                          cbs_radix_log=p.cbs_radix_log, cbs_radix_count=p.cbs_count,
                          ks_radix_log=p.ks_radix_log, ks_radix_count=p.ks_count)


@functools.lru_cache(maxsize=4)
def keyset(seed: int, lwe_n: int, with_ksk: bool = True) -> O.KeySet:
    return O.gen_keyset(seed, O.DEFAULT_128.replace(lwe_n=lwe_n), with_ksk=with_ksk)


def random_lwe_batch(seed: int, B: int, n: int) -> np.ndarray:
    """uniformly random LWE words (parity does not need valid encryptions)"""
    rng = np.random.default_rng(seed)
    return rng.integers(0, 1 << 64, size=(B, n + 1), dtype=np.uint64)


def random_glwe(seed: int, B: int, words: int) -> np.ndarray:
    rng = np.random.default_rng(seed)
    return rng.integers(0, 1 << 64, size=(B, words), dtype=np.uint64)


def gpu_available() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def dev_bootstrap(eng, lwe, lut=None, log_chi=0, log_v=0, body_rotate=0, extract=False):
    """ONE launch of the whole batch through the device-pointer entry points (the host-pointer forms cut a batch into
    slices of one chip round, 4 x #CU ciphertexts, so a ragged last workgroup of the throughput shape never forms there).
    lut None: spf_circuit_bootstrap_pbs_dev; extract: spf_pbs_univariate_dev; else spf_generalized_pbs_dev.
    Device buffers through the library's own spf_device_* helpers, default stream."""
    P = eng.params
    lwe = np.ascontiguousarray(lwe, dtype=np.uint64)
    B = lwe.shape[0]
    out = np.empty((B, P.lwe1_words if extract else P.glwe_words), dtype=np.uint64)
    bufs = []

    def up(a):
        p = eng.device_alloc(a.nbytes)
        bufs.append(p)
        eng.device_upload(p, a)
        return p

    try:
        d_lwe = up(lwe)
        d_out = eng.device_alloc(out.nbytes)
        bufs.append(d_out)
        if lut is None:
            eng.circuit_bootstrap_pbs_dev(None, B, d_lwe, d_out)
        else:
            lut = np.ascontiguousarray(lut, dtype=np.uint64)
            stride = 0 if lut.ndim == 1 else P.glwe_words
            d_lut = up(lut)
            if extract:
                eng.pbs_univariate_dev(None, B, d_lwe, d_lut, stride, d_out)
            else:
                eng.generalized_pbs_dev(None, B, d_lwe, d_lut, stride, log_chi, log_v, body_rotate, d_out)
        eng.device_download(None, out, d_out)
    finally:
        for p in bufs:
            eng.device_free(p)
    return out
