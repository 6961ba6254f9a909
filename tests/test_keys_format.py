"""ComputeKey bincode layout (keys.rs:294-318 + safe_bincode.rs:16-28): round trip, exact byte
layout on a tiny parameter set, and the size guards."""
import struct

import numpy as np
import pytest

import spf_amd
from spf_amd.keys import KeyFormatError, parse_compute_key, serialize_compute_key

P = spf_amd.DEFAULT_128.replace(lwe_dimension=2)


def _ck(seed=0):
    r = np.random.default_rng(seed)
    c = lambda n: r.standard_normal(n) + 1j * r.standard_normal(n)
    return spf_amd.ComputeKey(bs_key=c(P.bsk_complex), ks_key=r.integers(0, 1 << 64, P.ksk_words, dtype=np.uint64),
                              ss_key=c(P.ssk_complex), auto_key=c(P.ak_complex))


def test_roundtrip_and_layout():
    ck = _ck()
    buf = serialize_compute_key(ck)
    assert len(buf) == 4 * 8 + 16 * (P.bsk_complex + P.ssk_complex + P.ak_complex) + 8 * P.ksk_words
    # u64 LE length, then interleaved re/im little-endian doubles
    assert struct.unpack_from("<Q", buf, 0)[0] == P.bsk_complex
    assert struct.unpack_from("<dd", buf, 8) == (ck.bs_key[0].real, ck.bs_key[0].imag)
    off = 8 + 16 * P.bsk_complex
    assert struct.unpack_from("<Q", buf, off)[0] == P.ksk_words
    assert struct.unpack_from("<Q", buf, off + 8)[0] == int(ck.ks_key[0])
    back = parse_compute_key(buf + b"trailing bytes are allowed", P)
    for f in ("bs_key", "ks_key", "ss_key", "auto_key"):
        assert np.array_equal(getattr(back, f), getattr(ck, f))


def test_size_guards():
    buf = serialize_compute_key(_ck())
    with pytest.raises(KeyFormatError):
        parse_compute_key(buf, spf_amd.DEFAULT_128)          # wrong parameter set
    with pytest.raises(KeyFormatError):
        parse_compute_key(buf[: len(buf) // 2], P)           # truncated
    with pytest.raises(KeyFormatError):
        parse_compute_key(b"\x01\x00", P)
