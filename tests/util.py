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
