"""ComputeKey bincode layout (keys.rs:294-318 + safe_bincode.rs:16-28): round trip, exact byte
layout on a tiny parameter set, and the size guards."""
import os
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
    with pytest.raises(KeyFormatError):   # the reference's malformed-length vector (rejects_malformed_keys, safe_bincode.rs:98-117)
        parse_compute_key(bytes([253, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0x1, 0x2, 0x3, 0x4]), P)


@pytest.mark.gpu
def test_bincode_loader_behind_the_c_abi_equals_field_by_field_loading():
    """f4: `spf_load_compute_key_bincode` (safe_bincode.rs:16-28): one blob -> all four keys; same gate output as
    loading the fields one by one; wrong counts / truncation rejected before anything is touched."""
    import struct
    import oracle as O
    import spf_amd
    from spf_amd.keys import serialize_compute_key
    from tests.util import keyset, random_lwe_batch, to_engine_params
    ks = keyset(0x5EED0001, 12)
    P = ks.params
    r = O.Rng(0x7A11)
    ak, ssk = O.gen_auto_key_fft(r, ks.glwe_sk, P), O.gen_ssk_fft(r, ks.glwe_sk, P)
    blob = serialize_compute_key(spf_amd.ComputeKey(bs_key=ks.bsk_fft, ks_key=ks.ksk, ss_key=ssk, auto_key=ak))
    ref = spf_amd.Engine(to_engine_params(P))
    ref.load_bootstrap_key(ks.bsk_fft)
    ref.load_keyswitch_key(ks.ksk)
    ref.load_automorphism_key(ak)
    ref.load_scheme_switch_key(ssk)
    eng = spf_amd.Engine(to_engine_params(P))
    eng.load_compute_key_bincode(blob + b"trailing bytes are allowed")
    lwe1 = random_lwe_batch(3, 4, P.N * P.k)
    assert np.array_equal(eng.keyswitch_circuit_bootstrap(lwe1).view(np.float64),
                          ref.keyswitch_circuit_bootstrap(lwe1).view(np.float64))
    bad = bytearray(blob)
    bad[0:8] = struct.pack("<Q", ks.bsk_fft.size + 1)
    fresh = spf_amd.Engine(to_engine_params(P))
    # ... and the reference's own key-shaped negative vector (rejects_malformed_keys, safe_bincode.rs:98-117: a length
    # prefix of 0xFF..FF introduced by 253, then four stray bytes)
    ref_bad = bytes(_kats()["bytes"])
    for broken in (bytes(bad), blob[:-9], blob[:4], ref_bad):
        with pytest.raises(spf_amd.SpfError):
            fresh.load_compute_key_bincode(broken)
    with pytest.raises(spf_amd.SpfError):           # nothing was loaded by the failed attempts
        fresh.keyswitch_circuit_bootstrap(lwe1)


# ---- ciphertext half of f4: the serde newtypes of crypto/encryption.rs:23-110 through the C ABI (host only) ----

_KINDS = {"L0LweCiphertext": 0, "L1LweCiphertext": 1, "L1GlweCiphertext": 2, "L1GlevCiphertext": 4}


def _kats():
    import json
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_kats.json")) as f:
        return json.load(f)["safe_bincode_malformed_length"]


def test_ciphertext_bincode_sizes_match_getsize():
    """GetSize (encryption.rs:454-519) = (words + 1) * 8 bytes; words at DEFAULT_128 as the entities define them."""
    spf_amd.build_library()
    for name, words in _kats()["words_default128"].items():
        assert spf_amd.ciphertext_words(_KINDS[name]) == words, name
        blob = spf_amd.ciphertext_to_bincode(_KINDS[name], np.zeros(words, dtype=np.uint64))
        assert len(blob) == (words + 1) * 8
        assert blob[:8] == struct.pack("<Q", words)


def test_ciphertext_bincode_round_trip_like_can_safe_deserialize_ciphertexts():
    """safe_bincode.rs:41-53 with non-trivial contents: serialize -> deserialize is the identity for all four
    serializable newtypes; words are little-endian u64; trailing bytes are allowed (allow_trailing_bytes)."""
    rng = np.random.default_rng(0xB1C0DE)
    for name, kind in _KINDS.items():
        n = spf_amd.ciphertext_words(kind)
        w = rng.integers(0, 1 << 64, n, dtype=np.uint64)
        blob = spf_amd.ciphertext_to_bincode(kind, w)
        assert blob[8:16] == struct.pack("<Q", int(w[0]))
        assert np.array_equal(spf_amd.ciphertext_from_bincode(kind, blob), w), name
        assert np.array_equal(spf_amd.ciphertext_from_bincode(kind, blob + b"\x01\x02\x03"), w), name


def test_ciphertext_bincode_rejects_the_references_malformed_vector():
    """rejects_malformed_serialized_ciphertext (safe_bincode.rs:58-77): the reference's own negative vector, for
    every serializable kind; plus a count off by one either way, a truncated body, and a truncated count."""
    bad = bytes(_kats()["bytes"])
    for name, kind in _KINDS.items():
        with pytest.raises(spf_amd.SpfError):
            spf_amd.ciphertext_from_bincode(kind, bad)
        n = spf_amd.ciphertext_words(kind)
        good = spf_amd.ciphertext_to_bincode(kind, np.arange(n, dtype=np.uint64))
        for broken in (struct.pack("<Q", n + 1) + good[8:] + b"\0" * 8,   # longer than GetSize allows
                       struct.pack("<Q", n - 1) + good[8:],               # check_is_valid: wrong length
                       good[:-1], good[:5], b""):
            with pytest.raises(spf_amd.SpfError):
                spf_amd.ciphertext_from_bincode(kind, broken)
    # a ciphertext of one kind is not accepted as another
    l0 = spf_amd.ciphertext_to_bincode(0, np.zeros(638, dtype=np.uint64))
    with pytest.raises(spf_amd.SpfError):
        spf_amd.ciphertext_from_bincode(1, l0)


def test_l1ggsw_has_no_wire_format():
    """L1GgswCiphertext derives only Clone (encryption.rs:93-97): refused, not invented."""
    with pytest.raises(spf_amd.SpfError) as e:
        spf_amd.ciphertext_from_bincode(3, b"\0" * 64)
    assert e.value.status == 4
