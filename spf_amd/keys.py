"""`ComputeKey` wire format: what `bincode::DefaultOptions().with_fixint_encoding()`
produces for `parasol_runtime::ComputeKey` (parasol_runtime/src/crypto/keys.rs:294-318, read by
`safe_bincode::deserialize`, parasol_runtime/src/safe_bincode.rs:16-28).

Every key entity is a struct with one field `data: AVec<T>` (sunscreen_tfhe/src/dst.rs:25-41), which
serde writes as a sequence: u64 little-endian element count, then the elements — `Complex<f64>` as
two little-endian f64 (re, im), `Torus<u64>` as one little-endian u64.  Field order:
bs_key, ks_key, ss_key, auto_key.  No Rust toolchain exists in this environment, so the format is
restated from the source, not checked against bytes written by the Rust code."""
from __future__ import annotations

import struct

import numpy as np

from .evaluation import ComputeKey
from .params import Params, DEFAULT_128


class KeyFormatError(ValueError):
    pass


def _expected_counts(p: Params):
    return (p.bsk_complex, p.ksk_words, p.ssk_complex, p.ak_complex)


def parse_compute_key(buf: bytes, params: Params = DEFAULT_128) -> ComputeKey:
    """Deserialize; like safe_bincode it refuses lengths that do not match `params` (the
    reference bounds the read by `GetSize` and then runs `check_is_valid`)."""
    mv = memoryview(buf)
    off = 0
    out = []
    for name, want, dtype, esz in zip(("bs_key", "ks_key", "ss_key", "auto_key"), _expected_counts(params),
                                      (np.complex128, np.uint64, np.complex128, np.complex128), (16, 8, 16, 16)):
        if off + 8 > len(mv):
            raise KeyFormatError(f"truncated before the length of {name}")
        (n,) = struct.unpack_from("<Q", mv, off)
        off += 8
        if n != want:
            raise KeyFormatError(f"{name}: {n} elements, parameters need {want}")
        end = off + n * esz
        if end > len(mv):
            raise KeyFormatError(f"truncated inside {name}")
        out.append(np.frombuffer(mv[off:end], dtype=np.dtype(dtype).newbyteorder("<")).astype(dtype, copy=True))
        off = end
    return ComputeKey(bs_key=out[0], ks_key=out[1], ss_key=out[2], auto_key=out[3])   # trailing bytes allowed


def serialize_compute_key(ck: ComputeKey) -> bytes:
    parts = []
    for arr, dtype in ((ck.bs_key, np.complex128), (ck.ks_key, np.uint64), (ck.ss_key, np.complex128),
                       (ck.auto_key, np.complex128)):
        a = np.ascontiguousarray(arr, dtype=dtype).reshape(-1)
        parts.append(struct.pack("<Q", a.size))
        parts.append(a.astype(np.dtype(dtype).newbyteorder("<"), copy=False).tobytes())
    return b"".join(parts)
