"""ctypes binding of the C ABI in include/spf_hip.h.

``Engine`` wraps one ``spf_ctx`` (one GPU).  Host-array methods take / return numpy arrays and
go through the ``*_batch`` entry points; ``*_dev`` methods take raw device pointers (e.g.
``torch.Tensor.data_ptr()``) and a stream handle and are asynchronous.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

from .params import Params, DEFAULT_128

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class SpfError(RuntimeError):
    def __init__(self, status: int, msg: str):
        super().__init__(f"spf_hip status {status}: {msg}")
        self.status = status


class _CParams(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in (
        "lwe_dimension", "polynomial_degree", "glwe_size", "pbs_radix_log", "pbs_radix_count",
        "cbs_radix_log", "cbs_radix_count", "ks_radix_log", "ks_radix_count", "tr_radix_log",
        "tr_radix_count", "ss_radix_log", "ss_radix_count")]


def lib_path() -> str:
    # SPF_HIP_LIBRARY lets experiments (e.g. timing-only ablation builds) swap the library
    return os.environ.get("SPF_HIP_LIBRARY") or os.path.join(_HERE, "lib", "libspf_hip.so")


# every symbol include/spf_hip.h declares: (name, restype, argtypes)
_P, _SZ, _U32, _U64, _I = C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint64, C.c_int
SYMBOLS = [
    ("spf_default_params", None, [C.POINTER(_CParams)]),
    ("spf_create", _I, [C.POINTER(_CParams), _I, C.POINTER(_P)]),
    ("spf_destroy", None, [_P]),
    ("spf_last_error", C.c_char_p, [_P]),
    ("spf_load_bootstrap_key", _I, [_P, _P, _SZ]),
    ("spf_load_keyswitch_key", _I, [_P, _P, _SZ]),
    ("spf_load_automorphism_key", _I, [_P, _P, _SZ]),
    ("spf_load_scheme_switch_key", _I, [_P, _P, _SZ]),
    ("spf_key_blob", _I, [_P, _I, C.POINTER(_P), C.POINTER(_SZ)]),
    ("spf_key_blob_commit", _I, [_P, _I]),
    ("spf_keyswitch_lwe_l1_lwe_l0_batch", _I, [_P, _SZ, _P, _P]),
    ("spf_generalized_pbs_batch", _I, [_P, _SZ, _P, _P, _SZ, _U32, _U32, _U64, _P]),
    ("spf_pbs_univariate_batch", _I, [_P, _SZ, _P, _P, _SZ, _P]),
    ("spf_circuit_bootstrap_pbs_batch", _I, [_P, _SZ, _P, _P]),
    ("spf_circuit_bootstrap_batch", _I, [_P, _SZ, _P, _P]),
    ("spf_mod_switch_trace_and_rotate_batch", _I, [_P, _SZ, _P, _P]),
    ("spf_scheme_switch_batch", _I, [_P, _SZ, _P, _P]),
    ("spf_sample_extract_l1_batch", _I, [_P, _SZ, _P, _SZ, _P]),
    ("spf_glwe_not_batch", _I, [_P, _SZ, _P, _P]),
    ("spf_glwe_xor_batch", _I, [_P, _SZ, _P, _P, _P]),
    ("spf_glwe_mul_xn_batch", _I, [_P, _SZ, _P, _SZ, _P]),
    ("spf_glwe_not_dev", _I, [_P, _P, _SZ, _P, _P]),
    ("spf_glwe_xor_dev", _I, [_P, _P, _SZ, _P, _P, _P]),
    ("spf_glwe_mul_xn_dev", _I, [_P, _P, _SZ, _P, _SZ, _P]),
    ("spf_cmux_batch", _I, [_P, _SZ, _P, _P, _P, _P]),
    ("spf_glev_cmux_batch", _I, [_P, _SZ, _P, _P, _P, _P]),
    ("spf_multiply_glwe_ggsw_batch", _I, [_P, _SZ, _P, _P, _P]),
    ("spf_glev_cmux_dev", _I, [_P, _P, _SZ, _P, _P, _P, _P]),
    ("spf_multiply_glwe_ggsw_dev", _I, [_P, _P, _SZ, _P, _P, _P]),
    ("spf_gate_bootstrap_batch", _I, [_P, _SZ, _P, _P]),
    ("spf_keyswitch_circuit_bootstrap_batch", _I, [_P, _SZ, _P, _P]),
    ("spf_keyswitch_lwe_l1_lwe_l0_dev", _I, [_P, _P, _SZ, _P, _P]),
    ("spf_generalized_pbs_dev", _I, [_P, _P, _SZ, _P, _P, _SZ, _U32, _U32, _U64, _P]),
    ("spf_pbs_univariate_dev", _I, [_P, _P, _SZ, _P, _P, _SZ, _P]),
    ("spf_circuit_bootstrap_pbs_dev", _I, [_P, _P, _SZ, _P, _P]),
    ("spf_circuit_bootstrap_dev", _I, [_P, _P, _SZ, _P, _P]),
    ("spf_mod_switch_trace_and_rotate_dev", _I, [_P, _P, _SZ, _P, _P]),
    ("spf_scheme_switch_dev", _I, [_P, _P, _SZ, _P, _P]),
    ("spf_sample_extract_l1_dev", _I, [_P, _P, _SZ, _P, _SZ, _P]),
    ("spf_cmux_dev", _I, [_P, _P, _SZ, _P, _P, _P, _P]),
    ("spf_pool_create", _I, [_P, _SZ, _U32, C.POINTER(_P)]),
    ("spf_pool_destroy", None, [_P]),
    ("spf_pool_submit_keyswitch", _I, [_P, _P, _P, C.POINTER(_U64)]),
    ("spf_pool_submit_circuit_bootstrap", _I, [_P, _P, _P, C.POINTER(_U64)]),
    ("spf_pool_submit_keyswitch_circuit_bootstrap", _I, [_P, _P, _P, C.POINTER(_U64)]),
    ("spf_pool_submit_cmux", _I, [_P, _P, _P, _P, _P, C.POINTER(_U64)]),
    ("spf_pool_submit_sample_extract", _I, [_P, _P, _SZ, _P, C.POINTER(_U64)]),
    ("spf_pool_submit_not", _I, [_P, _P, _P, C.POINTER(_U64)]),
    ("spf_pool_submit_glwe_add", _I, [_P, _P, _P, _P, C.POINTER(_U64)]),
    ("spf_pool_submit_mul_xn", _I, [_P, _P, _SZ, _P, C.POINTER(_U64)]),
    ("spf_pool_submit_multiply_ggsw_glwe", _I, [_P, _P, _P, _P, C.POINTER(_U64)]),
    ("spf_pool_submit_glev_cmux", _I, [_P, _P, _P, _P, _P, C.POINTER(_U64)]),
    ("spf_pool_submit_scheme_switch", _I, [_P, _P, _P, C.POINTER(_U64)]),
    ("spf_pool_wait", _I, [_P, _U64]),
    ("spf_pool_set_max_inflight", _I, [_P, _SZ]),
    ("spf_pool_stats", _I, [_P, C.POINTER(_U64), C.POINTER(_U64)]),
    ("spf_graph_create", _I, [_P, C.POINTER(_P)]),
    ("spf_graph_destroy", None, [_P]),
    ("spf_graph_add_input", _I, [_P, _I, _P, C.POINTER(_U32)]),
    ("spf_graph_add_trivial", _I, [_P, _I, _U64, C.POINTER(_U32)]),
    ("spf_graph_add_op", _I, [_P, _I, C.POINTER(_U32), _SZ, _U64, C.POINTER(_U32)]),
    ("spf_graph_add_output", _I, [_P, _U32, _P]),
    ("spf_graph_run", _I, [_P]),
    ("spf_graph_stats", _I, [_P, C.POINTER(_U32), C.POINTER(_U32), C.POINTER(_U32)]),
    ("spf_cmux_scattered_dev", _I, [_P, _P, _SZ, _P]),
    ("spf_gather_rows_dev", _I, [_P, _P, _SZ, _SZ, _P, _P]),
    ("spf_set_timing", _I, [_P, _I]),
    ("spf_last_kernel_ms", _I, [_P, C.c_char_p, C.POINTER(C.c_double), C.POINTER(_I)]),
    ("spf_last_blind_rotate_kernel", C.c_char_p, [_P]),
    ("spf_last_cmux_kernel", C.c_char_p, [_P]),
    ("spf_device_alloc", _I, [_P, _SZ, C.POINTER(_P)]),
    ("spf_device_free", _I, [_P, _P]),
    ("spf_device_upload", _I, [_P, _P, _P, _SZ]),
    ("spf_device_download", _I, [_P, _P, _P, _P, _SZ]),
    ("spf_l1ggsw_constant", _I, [_P, _I, _P]),
    ("spf_generate_lut", _I, [C.POINTER(_CParams), _P, _SZ, _U32, _P]),
    ("spf_load_compute_key_bincode", _I, [_P, _P, _SZ]),
    ("spf_ciphertext_words", _SZ, [C.POINTER(_CParams), _I]),
    ("spf_ciphertext_from_bincode", _I, [C.POINTER(_CParams), _I, _P, _SZ, _P]),
    ("spf_ciphertext_to_bincode", _I, [C.POINTER(_CParams), _I, _P, _P, _SZ, C.POINTER(_SZ)]),
    ("spf_version", C.c_char_p, []),
    # device groups (one host process, every GPU of the node)
    ("spf_group_create", _I, [C.POINTER(_CParams), C.POINTER(_I), _I, C.POINTER(_P)]),
    ("spf_group_destroy", None, [_P]),
    ("spf_group_size", _I, [_P]),
    ("spf_group_ctx", _P, [_P, _I]),
    ("spf_group_last_error", C.c_char_p, [_P]),
    ("spf_group_load_bootstrap_key", _I, [_P, _P, _SZ]),
    ("spf_group_load_keyswitch_key", _I, [_P, _P, _SZ]),
    ("spf_group_load_automorphism_key", _I, [_P, _P, _SZ]),
    ("spf_group_load_scheme_switch_key", _I, [_P, _P, _SZ]),
    ("spf_group_load_compute_key_bincode", _I, [_P, _P, _SZ]),
    ("spf_group_replicate_keys", _I, [_P]),
    ("spf_group_replication_stats", _I, [_P, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(_SZ), C.POINTER(_I),
                                         C.POINTER(C.c_char_p)]),
    ("spf_group_set_member_enabled", _I, [_P, _I, _I]),
    ("spf_group_members_in_rotation", _I, [_P]),
    ("spf_group_debug_fail_next", _I, [_P, _I, _I]),
    ("spf_group_keyswitch_lwe_l1_lwe_l0_batch", _I, [_P, _SZ, _P, _P]),
    ("spf_group_generalized_pbs_batch", _I, [_P, _SZ, _P, _P, _SZ, _U32, _U32, _U64, _P]),
    ("spf_group_pbs_univariate_batch", _I, [_P, _SZ, _P, _P, _SZ, _P]),
    ("spf_group_circuit_bootstrap_pbs_batch", _I, [_P, _SZ, _P, _P]),
    ("spf_group_circuit_bootstrap_batch", _I, [_P, _SZ, _P, _P]),
    ("spf_group_mod_switch_trace_and_rotate_batch", _I, [_P, _SZ, _P, _P]),
    ("spf_group_scheme_switch_batch", _I, [_P, _SZ, _P, _P]),
    ("spf_group_sample_extract_l1_batch", _I, [_P, _SZ, _P, _SZ, _P]),
    ("spf_group_glwe_not_batch", _I, [_P, _SZ, _P, _P]),
    ("spf_group_glwe_xor_batch", _I, [_P, _SZ, _P, _P, _P]),
    ("spf_group_glwe_mul_xn_batch", _I, [_P, _SZ, _P, _SZ, _P]),
    ("spf_group_cmux_batch", _I, [_P, _SZ, _P, _P, _P, _P]),
    ("spf_group_glev_cmux_batch", _I, [_P, _SZ, _P, _P, _P, _P]),
    ("spf_group_multiply_glwe_ggsw_batch", _I, [_P, _SZ, _P, _P, _P]),
    ("spf_group_gate_bootstrap_batch", _I, [_P, _SZ, _P, _P]),
    ("spf_group_keyswitch_circuit_bootstrap_batch", _I, [_P, _SZ, _P, _P]),
    ("spf_group_l1ggsw_constant", _I, [_P, _I, _P]),
    ("spf_pool_create_group", _I, [_P, _SZ, _U32, C.POINTER(_P)]),
    ("spf_group_graph_create", _I, [_P, C.POINTER(_P)]),
    ("spf_group_run_graphs", _I, [_P, C.POINTER(_P), _SZ]),
    ("spf_graph_member", _I, [_P]),
    # device-resident values and the pool's submits by handle
    ("spf_pool_counters_get", _I, [_P, C.POINTER(_U64 * 11)]),
    ("spf_value_upload", _I, [_P, _I, _I, _P, C.POINTER(_P)]),
    ("spf_value_upload_batch", _I, [_P, _I, _I, _SZ, _P, C.POINTER(_P)]),
    ("spf_value_download_batch", _I, [_SZ, C.POINTER(_P), _P]),
    ("spf_value_trivial", _I, [_P, _I, _I, _U64, C.POINTER(_P)]),
    ("spf_value_download", _I, [_P, _P]),
    ("spf_value_wait", _I, [_P]),
    ("spf_pool_flush", _I, [_P]),
    ("spf_value_retain", _I, [_P]),
    ("spf_value_release", None, [_P]),
    ("spf_value_info", _I, [_P, C.POINTER(_I), C.POINTER(_SZ), C.POINTER(_I), C.POINTER(_I)]),
    ("spf_value_device_ptr", _I, [_P, C.POINTER(_P)]),
    ("spf_value_copy_to_member", _I, [_P, _P, _I, C.POINTER(_P)]),
    ("spf_pool_value_stats", _I, [_P, C.POINTER(_SZ), C.POINTER(_SZ), C.POINTER(_SZ)]),
    ("spf_pool_trim", _I, [_P]),
    ("spf_pool_submit_keyswitch_v", _I, [_P, _P, C.POINTER(_P), C.POINTER(_U64)]),
    ("spf_pool_submit_circuit_bootstrap_v", _I, [_P, _P, C.POINTER(_P), C.POINTER(_U64)]),
    ("spf_pool_submit_keyswitch_circuit_bootstrap_v", _I, [_P, _P, C.POINTER(_P), C.POINTER(_U64)]),
    ("spf_pool_submit_cmux_v", _I, [_P, _P, _P, _P, C.POINTER(_P), C.POINTER(_U64)]),
    ("spf_pool_submit_sample_extract_v", _I, [_P, _P, _SZ, C.POINTER(_P), C.POINTER(_U64)]),
    ("spf_pool_submit_not_v", _I, [_P, _P, C.POINTER(_P), C.POINTER(_U64)]),
    ("spf_pool_submit_glwe_add_v", _I, [_P, _P, _P, C.POINTER(_P), C.POINTER(_U64)]),
    ("spf_pool_submit_mul_xn_v", _I, [_P, _P, _SZ, C.POINTER(_P), C.POINTER(_U64)]),
    ("spf_pool_submit_multiply_ggsw_glwe_v", _I, [_P, _P, _P, C.POINTER(_P), C.POINTER(_U64)]),
    ("spf_pool_submit_glev_cmux_v", _I, [_P, _P, _P, _P, C.POINTER(_P), C.POINTER(_U64)]),
    ("spf_pool_submit_scheme_switch_v", _I, [_P, _P, C.POINTER(_P), C.POINTER(_U64)]),
    ("spf_pool_submit_op_v", _I, [_P, _I, C.POINTER(_P), _SZ, _U64, C.POINTER(_P), C.POINTER(_U64)]),
]


def load_library(path: Optional[str] = None):
    """dlopen the HIP library and declare every prototype.  Raises if it has not been built —
    the product path never falls back to a CPU implementation."""
    global _LIB
    if _LIB is not None and path is None:
        return _LIB
    p = path or lib_path()
    if not os.path.exists(p):
        raise SpfError(-1, f"{p} is missing: build it with spf_amd.build_library() "
                           "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    lib = C.CDLL(p)
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)  # AttributeError if the header and the library disagree
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _LIB = lib
    return lib


def _u64(a, shape=None) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint64)
    if shape is not None and a.shape != shape:
        raise ValueError(f"expected shape {shape}, got {a.shape}")
    return a


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def _same_rows(what: str, *arrays):
    """the C side copies B * row_size bytes of every operand: a batch mismatch would be an out-of-bounds
    host read (the reference panics in assert_is_valid instead)"""
    rows = {a.shape[0] for a in arrays}
    if len(rows) != 1:
        raise SpfError(-2, f"{what}: operand batches differ in size {[a.shape[0] for a in arrays]}")


def _out(what: str, output, dtype, words: int) -> np.ndarray:
    """caller-allocated output of a pool call: right dtype, C-contiguous, exactly `words` elements"""
    if not isinstance(output, np.ndarray) or output.dtype != np.dtype(dtype) or not output.flags["C_CONTIGUOUS"] \
            or not output.flags["WRITEABLE"] or output.size != words:
        raise SpfError(-2, f"{what}: output must be a writable C-contiguous {np.dtype(dtype).name} array of {words} "
                           f"elements, got {getattr(output, 'dtype', type(output))} x {getattr(output, 'size', '?')}")
    return output


def _in(what: str, a, dtype, words: int) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=dtype)
    if a.size != words:
        raise SpfError(-2, f"{what}: input has {a.size} elements, expected {words}")
    return a


def generate_lut(maps, plaintext_bits: int, params: Params = DEFAULT_128) -> np.ndarray:
    """`generate_lut` (programmable_bootstrapping.rs:129-185) as the trivial GLWE `pbs_univariate` takes.
    `maps`: callables x -> f(x) on [0, 2^plaintext_bits), or already tabulated rows.  Host only, no GPU."""
    lib = load_library()
    p = 1 << plaintext_bits
    table = np.array([[m(x) for x in range(p)] if callable(m) else list(m) for m in maps], dtype=np.uint64).reshape(len(maps), p)
    out = np.empty(params.glwe_words, dtype=np.uint64)
    cp = _CParams(*[getattr(params, n) for n, _ in _CParams._fields_])
    st = lib.spf_generate_lut(C.byref(cp), _ptr(table), len(maps), plaintext_bits, _ptr(out))
    if st != 0:
        raise SpfError(st, (lib.spf_last_error(None) or b"").decode())
    return out


def _cparams(params: Params) -> _CParams:
    return _CParams(*[getattr(params, n) for n, _ in _CParams._fields_])


def ciphertext_words(kind: int, params: Params = DEFAULT_128) -> int:
    return int(load_library().spf_ciphertext_words(C.byref(_cparams(params)), int(kind)))


def ciphertext_from_bincode(kind: int, blob: bytes, params: Params = DEFAULT_128) -> np.ndarray:
    """`safe_bincode::deserialize::<L0Lwe | L1Lwe | L1Glwe | L1Glev Ciphertext>` (safe_bincode.rs:16-28,
    encryption.rs:23-110): bytes -> u64 words.  Host only.  Malformed input raises SpfError."""
    lib = load_library()
    n = ciphertext_words(kind, params)
    out = np.empty(max(n, 1), dtype=np.uint64)
    buf = np.frombuffer(bytes(blob) if len(blob) else b"\0", dtype=np.uint8)
    st = lib.spf_ciphertext_from_bincode(C.byref(_cparams(params)), int(kind), _ptr(buf), len(blob), _ptr(out))
    if st != 0:
        raise SpfError(st, (lib.spf_last_error(None) or b"").decode())
    return out[:n]


def ciphertext_to_bincode(kind: int, words, params: Params = DEFAULT_128) -> bytes:
    """`bincode::serialize` of one of the serializable ciphertext newtypes (fixint): u64 count + words, little-endian."""
    lib = load_library()
    n = ciphertext_words(kind, params)
    w = _in("ciphertext", words, np.uint64, max(n, 1)) if n else np.zeros(1, dtype=np.uint64)
    out = np.empty(8 + 8 * n, dtype=np.uint8)
    written = _SZ(0)
    st = lib.spf_ciphertext_to_bincode(C.byref(_cparams(params)), int(kind), _ptr(w), _ptr(out), out.size, C.byref(written))
    if st != 0:
        raise SpfError(st, (lib.spf_last_error(None) or b"").decode())
    return out[:written.value].tobytes()


class Engine:
    """One GPU's bootstrap engine (`spf_ctx`)."""

    def __init__(self, params: Params = DEFAULT_128, device: int = 0):
        self._lib = load_library()
        self.params = params
        self.device = device
        cp = _CParams(*[getattr(params, n) for n, _ in _CParams._fields_])
        h = C.c_void_p()
        st = self._lib.spf_create(C.byref(cp), device, C.byref(h))
        if st != 0:
            raise SpfError(st, (self._lib.spf_last_error(None) or b"").decode())
        self._h = h

    # -- plumbing
    def _ck(self, st: int):
        if st != 0:
            raise SpfError(st, (self._lib.spf_last_error(self._h) or b"").decode())

    def close(self):
        if getattr(self, "_h", None):
            self._lib.spf_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def version(self) -> str:
        return self._lib.spf_version().decode()

    # -- keys (ComputeKey fields, crypto/keys.rs:306-318)
    def load_bootstrap_key(self, bsk_fft: np.ndarray):
        a = np.ascontiguousarray(bsk_fft, dtype=np.complex128).reshape(-1)
        self._ck(self._lib.spf_load_bootstrap_key(self._h, _ptr(a), a.size))

    def load_keyswitch_key(self, ksk: np.ndarray):
        a = _u64(ksk).reshape(-1)
        self._ck(self._lib.spf_load_keyswitch_key(self._h, _ptr(a), a.size))

    def load_automorphism_key(self, ak_fft: np.ndarray):
        a = np.ascontiguousarray(ak_fft, dtype=np.complex128).reshape(-1)
        self._ck(self._lib.spf_load_automorphism_key(self._h, _ptr(a), a.size))

    def load_scheme_switch_key(self, ssk_fft: np.ndarray):
        a = np.ascontiguousarray(ssk_fft, dtype=np.complex128).reshape(-1)
        self._ck(self._lib.spf_load_scheme_switch_key(self._h, _ptr(a), a.size))

    def key_blob(self, which: int):
        """(device pointer, bytes) of the device-resident key; for RCCL broadcast."""
        p, n = C.c_void_p(), C.c_size_t()
        self._ck(self._lib.spf_key_blob(self._h, which, C.byref(p), C.byref(n)))
        return p.value, n.value

    def key_blob_commit(self, which: int):
        self._ck(self._lib.spf_key_blob_commit(self._h, which))

    # -- host-array batch forms
    def keyswitch_lwe_l1_lwe_l0(self, lwe1: np.ndarray) -> np.ndarray:
        x = _u64(lwe1)
        x = x.reshape(-1, self.params.lwe1_words)
        out = np.empty((x.shape[0], self.params.lwe0_words), dtype=np.uint64)
        self._ck(self._lib.spf_keyswitch_lwe_l1_lwe_l0_batch(self._h, x.shape[0], _ptr(x), _ptr(out)))
        return out

    def _lut(self, lut, B):
        lut = _u64(lut)
        if lut.ndim == 1:
            if lut.size != self.params.glwe_words:
                raise ValueError("lut must hold one GLWE")
            return lut, 0
        if lut.shape != (B, self.params.glwe_words):
            raise ValueError("per-ciphertext luts must be B x glwe_words")
        return lut, self.params.glwe_words

    def generalized_pbs(self, lwe0, lut_glwe, log_chi=0, log_v=0, body_rotate=0) -> np.ndarray:
        x = _u64(lwe0).reshape(-1, self.params.lwe0_words)
        lut, stride = self._lut(lut_glwe, x.shape[0])
        out = np.empty((x.shape[0], self.params.glwe_words), dtype=np.uint64)
        self._ck(self._lib.spf_generalized_pbs_batch(self._h, x.shape[0], _ptr(x), _ptr(lut), stride,
                                                     log_chi, log_v, body_rotate, _ptr(out)))
        return out

    def pbs_univariate(self, lwe0, lut_glwe) -> np.ndarray:
        x = _u64(lwe0).reshape(-1, self.params.lwe0_words)
        lut, stride = self._lut(lut_glwe, x.shape[0])
        out = np.empty((x.shape[0], self.params.lwe1_words), dtype=np.uint64)
        self._ck(self._lib.spf_pbs_univariate_batch(self._h, x.shape[0], _ptr(x), _ptr(lut), stride,
                                                    _ptr(out)))
        return out

    def circuit_bootstrap_pbs(self, lwe0, out: Optional[np.ndarray] = None) -> np.ndarray:
        """`out`: caller-allocated (B, glwe_words) uint64 array, as the C ABI's caller does — a fresh np.empty is
        untouched memory, and its first-touch page faults land inside the device-to-host copy"""
        x = _u64(lwe0).reshape(-1, self.params.lwe0_words)
        if out is None:
            out = np.empty((x.shape[0], self.params.glwe_words), dtype=np.uint64)
        else:
            _out("circuit_bootstrap_pbs", out, np.uint64, x.shape[0] * self.params.glwe_words)
        self._ck(self._lib.spf_circuit_bootstrap_pbs_batch(self._h, x.shape[0], _ptr(x), _ptr(out)))
        return out

    def circuit_bootstrap(self, lwe0) -> np.ndarray:
        """Evaluation::circuit_bootstrap: L0 LWE -> L1 GGSW-FFT (B x cbs_ggsw_complex)."""
        x = _u64(lwe0).reshape(-1, self.params.lwe0_words)
        out = np.empty((x.shape[0], self.params.cbs_ggsw_complex), dtype=np.complex128)
        self._ck(self._lib.spf_circuit_bootstrap_batch(self._h, x.shape[0], _ptr(x), _ptr(out)))
        return out

    def mod_switch_trace_and_rotate(self, glwe) -> np.ndarray:
        x = _u64(glwe).reshape(-1, self.params.glwe_words)
        out = np.empty((x.shape[0], self.params.cbs_radix_count, self.params.glwe_words), dtype=np.uint64)
        self._ck(self._lib.spf_mod_switch_trace_and_rotate_batch(self._h, x.shape[0], _ptr(x), _ptr(out)))
        return out

    def scheme_switch(self, glev) -> np.ndarray:
        x = _u64(glev).reshape(-1, self.params.cbs_radix_count * self.params.glwe_words)
        out = np.empty((x.shape[0], self.params.cbs_ggsw_complex), dtype=np.complex128)
        self._ck(self._lib.spf_scheme_switch_batch(self._h, x.shape[0], _ptr(x), _ptr(out)))
        return out

    def sample_extract_l1(self, glwe, idx: int) -> np.ndarray:
        x = _u64(glwe).reshape(-1, self.params.glwe_words)
        out = np.empty((x.shape[0], self.params.lwe1_words), dtype=np.uint64)
        self._ck(self._lib.spf_sample_extract_l1_batch(self._h, x.shape[0], _ptr(x), idx, _ptr(out)))
        return out

    def glwe_not(self, glwe) -> np.ndarray:
        x = _u64(glwe).reshape(-1, self.params.glwe_words)
        out = np.empty_like(x)
        self._ck(self._lib.spf_glwe_not_batch(self._h, x.shape[0], _ptr(x), _ptr(out)))
        return out

    def glwe_xor(self, a, b) -> np.ndarray:
        a = _u64(a).reshape(-1, self.params.glwe_words)
        b = _u64(b).reshape(-1, self.params.glwe_words)
        if a.shape != b.shape:
            raise SpfError(-2, "glwe_xor: operand batches differ in size")
        out = np.empty_like(a)
        self._ck(self._lib.spf_glwe_xor_batch(self._h, a.shape[0], _ptr(a), _ptr(b), _ptr(out)))
        return out

    def glwe_mul_xn(self, glwe, n: int) -> np.ndarray:
        x = _u64(glwe).reshape(-1, self.params.glwe_words)
        out = np.empty_like(x)
        self._ck(self._lib.spf_glwe_mul_xn_batch(self._h, x.shape[0], _ptr(x), n, _ptr(out)))
        return out

    def cmux(self, sel_ggsw_fft, a, b) -> np.ndarray:
        g = np.ascontiguousarray(sel_ggsw_fft, dtype=np.complex128).reshape(-1, self.params.cbs_ggsw_complex)
        a = _u64(a).reshape(-1, self.params.glwe_words)
        b = _u64(b).reshape(-1, self.params.glwe_words)
        _same_rows("cmux", g, a, b)
        out = np.empty_like(a)
        self._ck(self._lib.spf_cmux_batch(self._h, a.shape[0], _ptr(g), _ptr(a), _ptr(b), _ptr(out)))
        return out

    def glev_cmux(self, sel_ggsw_fft, a, b) -> np.ndarray:
        n = self.params.cbs_radix_count * self.params.glwe_words
        g = np.ascontiguousarray(sel_ggsw_fft, dtype=np.complex128).reshape(-1, self.params.cbs_ggsw_complex)
        a = _u64(a).reshape(-1, n)
        b = _u64(b).reshape(-1, n)
        _same_rows("glev_cmux", g, a, b)
        out = np.empty_like(a)
        self._ck(self._lib.spf_glev_cmux_batch(self._h, a.shape[0], _ptr(g), _ptr(a), _ptr(b), _ptr(out)))
        return out

    def multiply_glwe_ggsw(self, glwe, ggsw_fft) -> np.ndarray:
        g = np.ascontiguousarray(ggsw_fft, dtype=np.complex128).reshape(-1, self.params.cbs_ggsw_complex)
        x = _u64(glwe).reshape(-1, self.params.glwe_words)
        _same_rows("multiply_glwe_ggsw", g, x)
        out = np.empty_like(x)
        self._ck(self._lib.spf_multiply_glwe_ggsw_batch(self._h, x.shape[0], _ptr(x), _ptr(g), _ptr(out)))
        return out

    def keyswitch_circuit_bootstrap(self, lwe1) -> np.ndarray:
        """KeyswitchL1toL0 -> CircuitBootstrap (L1 LWE in, L1 GGSW-FFT out); the L0 LWE stays on the device"""
        x = _u64(lwe1).reshape(-1, self.params.lwe1_words)
        out = np.empty((x.shape[0], self.params.cbs_ggsw_complex), dtype=np.complex128)
        self._ck(self._lib.spf_keyswitch_circuit_bootstrap_batch(self._h, x.shape[0], _ptr(x), _ptr(out)))
        return out

    def gate_bootstrap(self, lwe1, out: Optional[np.ndarray] = None) -> np.ndarray:
        x = _u64(lwe1).reshape(-1, self.params.lwe1_words)
        if out is None:
            out = np.empty((x.shape[0], self.params.glwe_words), dtype=np.uint64)
        else:
            _out("gate_bootstrap", out, np.uint64, x.shape[0] * self.params.glwe_words)
        self._ck(self._lib.spf_gate_bootstrap_batch(self._h, x.shape[0], _ptr(x), _ptr(out)))
        return out

    # -- device-pointer forms (ints are raw device addresses; stream is a hipStream_t handle)
    def keyswitch_dev(self, stream: int, B: int, d_in: int, d_out: int):
        self._ck(self._lib.spf_keyswitch_lwe_l1_lwe_l0_dev(self._h, stream, B, d_in, d_out))

    def generalized_pbs_dev(self, stream, B, d_lwe, d_lut, lut_stride, log_chi, log_v, body_rotate, d_out):
        self._ck(self._lib.spf_generalized_pbs_dev(self._h, stream, B, d_lwe, d_lut, lut_stride,
                                                   log_chi, log_v, body_rotate, d_out))

    def pbs_univariate_dev(self, stream, B, d_lwe, d_lut, lut_stride, d_out):
        self._ck(self._lib.spf_pbs_univariate_dev(self._h, stream, B, d_lwe, d_lut, lut_stride, d_out))

    def circuit_bootstrap_pbs_dev(self, stream: int, B: int, d_lwe: int, d_out: int):
        self._ck(self._lib.spf_circuit_bootstrap_pbs_dev(self._h, stream, B, d_lwe, d_out))

    def circuit_bootstrap_dev(self, stream, B, d_lwe, d_ggsw_out):
        self._ck(self._lib.spf_circuit_bootstrap_dev(self._h, stream, B, d_lwe, d_ggsw_out))

    def mod_switch_trace_and_rotate_dev(self, stream, B, d_glwe, d_glev_out):
        self._ck(self._lib.spf_mod_switch_trace_and_rotate_dev(self._h, stream, B, d_glwe, d_glev_out))

    def scheme_switch_dev(self, stream, B, d_glev, d_ggsw_out):
        self._ck(self._lib.spf_scheme_switch_dev(self._h, stream, B, d_glev, d_ggsw_out))

    def cmux_dev(self, stream, B, d_sel_ggsw_fft, d_a, d_b, d_out):
        self._ck(self._lib.spf_cmux_dev(self._h, stream, B, d_sel_ggsw_fft, d_a, d_b, d_out))

    def cmux_scattered_dev(self, stream, units, d_ptrs):
        """`units` CMUXes over scattered operands: d_ptrs = device array of 4 pointers per unit {selector, a (0 = zero), b, out}"""
        self._ck(self._lib.spf_cmux_scattered_dev(self._h, stream, units, d_ptrs))

    def glwe_not_dev(self, stream, B, d_in, d_out):
        self._ck(self._lib.spf_glwe_not_dev(self._h, stream, B, d_in, d_out))

    def glwe_xor_dev(self, stream, B, d_a, d_b, d_out):
        self._ck(self._lib.spf_glwe_xor_dev(self._h, stream, B, d_a, d_b, d_out))

    def glwe_mul_xn_dev(self, stream, B, d_in, n, d_out):
        self._ck(self._lib.spf_glwe_mul_xn_dev(self._h, stream, B, d_in, n, d_out))

    def sample_extract_l1_dev(self, stream, B, d_glwe, idx, d_out):
        self._ck(self._lib.spf_sample_extract_l1_dev(self._h, stream, B, d_glwe, idx, d_out))

    # -- device buffers (for chaining the _dev forms without a HIP binding of one's own)
    def device_alloc(self, nbytes: int) -> int:
        p = C.c_void_p()
        self._ck(self._lib.spf_device_alloc(self._h, int(nbytes), C.byref(p)))
        return p.value or 0

    def device_free(self, ptr: int):
        self._ck(self._lib.spf_device_free(self._h, ptr))

    def device_upload(self, ptr: int, host: np.ndarray):
        host = np.ascontiguousarray(host)
        self._ck(self._lib.spf_device_upload(self._h, ptr, _ptr(host), host.nbytes))

    def device_download(self, stream, host: np.ndarray, ptr: int):
        """waits for `stream` (None = the default stream), then copies host.nbytes bytes from `ptr`"""
        assert host.flags.c_contiguous
        self._ck(self._lib.spf_device_download(self._h, stream, _ptr(host), ptr, host.nbytes))

    # -- measurement
    def set_timing(self, enabled: bool):
        self._ck(self._lib.spf_set_timing(self._h, 1 if enabled else 0))

    def load_compute_key_bincode(self, blob: bytes):
        """the bytes `bincode` wrote for a parasol_runtime::ComputeKey (safe_bincode.rs:16-28), all four keys"""
        buf = np.frombuffer(blob, dtype=np.uint8)
        self._ck(self._lib.spf_load_compute_key_bincode(self._h, _ptr(buf), buf.size))

    def l1ggsw_constant(self, bit: int) -> np.ndarray:
        """Evaluation::l1ggsw_zero / l1ggsw_one: circuit bootstrap of the trivial L0 LWE of `bit` (cached per key set)"""
        out = np.empty(self.params.cbs_ggsw_complex, dtype=np.complex128)
        self._ck(self._lib.spf_l1ggsw_constant(self._h, int(bit), _ptr(out)))
        return out

    def last_blind_rotate_kernel(self) -> str:
        return (self._lib.spf_last_blind_rotate_kernel(self._h) or b"").decode()

    def last_cmux_kernel(self) -> str:
        return (self._lib.spf_last_cmux_kernel(self._h) or b"").decode()

    def last_kernel_ms(self, kernel: str = "pbs"):
        ms, n = C.c_double(), C.c_int()
        self._ck(self._lib.spf_last_kernel_ms(self._h, kernel.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value


class _GroupLib:
    """the library seen through a group handle: `spf_X` resolves to `spf_group_X` where the group has that entry point, so
    that every host-array method of Engine runs unchanged over the group"""

    def __init__(self, lib):
        self._lib = lib

    def __getattr__(self, name):
        if name.startswith("spf_") and not name.startswith("spf_group_"):
            fn = getattr(self._lib, "spf_group_" + name[4:], None)
            if fn is not None:
                return fn
            if name not in ("spf_version",):   # anything else would take the group handle for a context
                raise SpfError(-2, f"{name} has no group form: use group.member(i)")
        return getattr(self._lib, name)


class _MemberEngine(Engine):
    """member i of a group as an Engine (the `_dev` forms, gate graphs, measurement hooks on that device); the context
    belongs to the group"""

    def __init__(self, group: "Group", member: int):
        self._lib = group._raw
        self.params = group.params
        self.device = group.devices[member]
        self._group = group   # keeps the owner alive
        self._h = C.c_void_p(group._raw.spf_group_ctx(group._h, member))
        if not self._h:
            raise SpfError(-2, f"group has no member {member}")

    def close(self):
        self._h = None


class Group(Engine):
    """Every listed GPU behind one handle (`spf_group`): keys replicated inside the library (RCCL broadcast from member 0),
    host batches cut into contiguous ranges of ceil(B / G).  Has every host-array method of Engine; the device-pointer
    forms belong to a member: `group.member(i)`."""

    def __init__(self, params: Params = DEFAULT_128, devices=(0,)):
        self._raw = load_library()
        self._lib = _GroupLib(self._raw)
        self.params = params
        self.devices = [int(d) for d in devices]
        self.device = self.devices[0] if self.devices else -1
        cp = _cparams(params)
        ids = (C.c_int * len(self.devices))(*self.devices)
        h = C.c_void_p()
        st = self._raw.spf_group_create(C.byref(cp), ids, len(self.devices), C.byref(h))
        if st != 0:
            raise SpfError(st, (self._raw.spf_last_error(None) or b"").decode())
        self._h = h

    def _ck(self, st: int):
        if st != 0:
            raise SpfError(st, (self._raw.spf_group_last_error(self._h) or b"").decode())

    def close(self):
        if getattr(self, "_h", None):
            self._raw.spf_group_destroy(self._h)
            self._h = None

    def __len__(self):
        return int(self._raw.spf_group_size(self._h))

    def member(self, i: int) -> Engine:
        return _MemberEngine(self, i)

    def replicate_keys(self):
        self._ck(self._raw.spf_group_replicate_keys(self._h))

    def replication_stats(self) -> dict:
        w, ci, b, ws, t = C.c_double(), C.c_double(), C.c_size_t(), C.c_int(), C.c_char_p()
        self._ck(self._raw.spf_group_replication_stats(self._h, C.byref(w), C.byref(ci), C.byref(b), C.byref(ws), C.byref(t)))
        return {"wire_seconds": w.value, "comm_init_seconds": ci.value, "bytes_per_member": b.value,
                "rccl_world_size": ws.value, "transport": (t.value or b"").decode()}

    def set_member_enabled(self, member: int, enabled: bool):
        self._ck(self._raw.spf_group_set_member_enabled(self._h, member, 1 if enabled else 0))

    def members_in_rotation(self) -> int:
        return int(self._raw.spf_group_members_in_rotation(self._h))

    def debug_fail_next(self, member: int, count: int = 1):
        self._ck(self._raw.spf_group_debug_fail_next(self._h, member, count))

    def run_graphs(self, graphs):
        """`spf_group_run_graphs`: graphs are `spf_amd.FheCircuit`s made over this group; dealt to the members by cost,
        every member's jobs lowered into one graph, all members side by side"""
        arr = (C.c_void_p * len(graphs))(*[g._g for g in graphs])
        self._ck(self._raw.spf_group_run_graphs(self._h, arr, len(graphs)))

    # the per-context hooks have no group form
    def key_blob(self, which):
        raise SpfError(-2, "key blobs belong to a member: group.member(i).key_blob(which)")

    key_blob_commit = key_blob


class Pool:
    """Call-coalescing front end (`spf_pool`): many threads submit single-ciphertext operations and
    block in wait(); a worker thread runs what is pending as one batch.  The synchronous methods
    below are what a per-op caller (the reference's rayon task) would use."""

    def __init__(self, engine: Engine, max_batch: int = 4096, max_wait_us: int = 200):
        self._lib = load_library()
        self.engine = engine
        h = C.c_void_p()
        if isinstance(engine, Group):   # one pool per member, callers dealt across the devices
            st = self._lib.spf_pool_create_group(engine._h, max_batch, max_wait_us, C.byref(h))
        else:
            st = self._lib.spf_pool_create(engine._h, max_batch, max_wait_us, C.byref(h))
        if st != 0:
            raise SpfError(st, "spf_pool_create failed")
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._lib.spf_pool_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_max_inflight(self, n: int):
        st = self._lib.spf_pool_set_max_inflight(self._h, n)
        if st != 0:
            raise SpfError(st, "spf_pool_set_max_inflight failed")

    def submit_keyswitch(self, output: np.ndarray, input: np.ndarray) -> int:
        """asynchronous form: returns the ticket; buffers must stay alive until wait(ticket) returns"""
        t = C.c_uint64()
        x = _in("pool keyswitch", input, np.uint64, self.engine.params.lwe1_words)
        _out("pool keyswitch", output, np.uint64, self.engine.params.lwe0_words)
        st = self._lib.spf_pool_submit_keyswitch(self._h, _ptr(x), _ptr(output), C.byref(t))
        if st != 0:
            raise SpfError(st, "submit failed")
        self._keep = getattr(self, "_keep", {})
        self._keep[t.value] = (x, output)
        return t.value

    def wait(self, ticket: int):
        try:
            self._wait(ticket)
        finally:
            getattr(self, "_keep", {}).pop(ticket, None)

    def _wait(self, ticket):
        st = self._lib.spf_pool_wait(self._h, ticket)
        if st != 0:
            if isinstance(self.engine, Group):
                eng = self.engine.member(min(ticket >> 56, len(self.engine.devices) - 1))
                raise SpfError(st, (self._lib.spf_last_error(eng._h) or b"").decode())
            raise SpfError(st, (self._lib.spf_last_error(self.engine._h) or b"").decode())

    def keyswitch_lwe_l1_lwe_l0(self, output: np.ndarray, input: np.ndarray):
        t = C.c_uint64()
        x = _in("pool keyswitch", input, np.uint64, self.engine.params.lwe1_words)
        _out("pool keyswitch", output, np.uint64, self.engine.params.lwe0_words)
        st = self._lib.spf_pool_submit_keyswitch(self._h, _ptr(x), _ptr(output), C.byref(t))
        if st != 0:
            raise SpfError(st, "submit failed")
        self._wait(t.value)

    def circuit_bootstrap(self, output: np.ndarray, input: np.ndarray):
        t = C.c_uint64()
        x = _in("pool circuit_bootstrap", input, np.uint64, self.engine.params.lwe0_words)
        _out("pool circuit_bootstrap", output, np.complex128, self.engine.params.cbs_ggsw_complex)
        st = self._lib.spf_pool_submit_circuit_bootstrap(self._h, _ptr(x), _ptr(output), C.byref(t))
        if st != 0:
            raise SpfError(st, "submit failed")
        self._wait(t.value)

    def keyswitch_circuit_bootstrap(self, output: np.ndarray, input_l1: np.ndarray):
        t = C.c_uint64()
        x = _in("pool keyswitch_circuit_bootstrap", input_l1, np.uint64, self.engine.params.lwe1_words)
        _out("pool keyswitch_circuit_bootstrap", output, np.complex128, self.engine.params.cbs_ggsw_complex)
        st = self._lib.spf_pool_submit_keyswitch_circuit_bootstrap(self._h, _ptr(x), _ptr(output), C.byref(t))
        if st != 0:
            raise SpfError(st, "submit failed")
        self._wait(t.value)

    def cmux(self, output: np.ndarray, sel: np.ndarray, a: np.ndarray, b: np.ndarray):
        t = C.c_uint64()
        P = self.engine.params
        s_ = _in("pool cmux", sel, np.complex128, P.cbs_ggsw_complex)
        a_, b_ = _in("pool cmux", a, np.uint64, P.glwe_words), _in("pool cmux", b, np.uint64, P.glwe_words)
        _out("pool cmux", output, np.uint64, P.glwe_words)
        st = self._lib.spf_pool_submit_cmux(self._h, _ptr(s_), _ptr(a_), _ptr(b_), _ptr(output), C.byref(t))
        if st != 0:
            raise SpfError(st, "submit failed")
        self._wait(t.value)

    # -- the other FheOp kinds of CircuitProcessor::exec_op (synchronous forms: submit + wait)
    def _run(self, what, fn, *args):
        t = C.c_uint64()
        st = fn(self._h, *args, C.byref(t))
        if st != 0:
            raise SpfError(st, f"pool {what}: submit failed")
        self._wait(t.value)

    def sample_extract_l1(self, output: np.ndarray, glwe: np.ndarray, idx: int):
        P = self.engine.params
        x = _in("pool sample_extract", glwe, np.uint64, P.glwe_words)
        _out("pool sample_extract", output, np.uint64, P.lwe1_words)
        self._run("sample_extract", self._lib.spf_pool_submit_sample_extract, _ptr(x), int(idx), _ptr(output))

    def glwe_not(self, output: np.ndarray, glwe: np.ndarray):
        P = self.engine.params
        x = _in("pool not", glwe, np.uint64, P.glwe_words)
        _out("pool not", output, np.uint64, P.glwe_words)
        self._run("not", self._lib.spf_pool_submit_not, _ptr(x), _ptr(output))

    def glwe_add(self, output: np.ndarray, a: np.ndarray, b: np.ndarray):
        P = self.engine.params
        a_, b_ = _in("pool glwe_add", a, np.uint64, P.glwe_words), _in("pool glwe_add", b, np.uint64, P.glwe_words)
        _out("pool glwe_add", output, np.uint64, P.glwe_words)
        self._run("glwe_add", self._lib.spf_pool_submit_glwe_add, _ptr(a_), _ptr(b_), _ptr(output))

    def mul_xn(self, output: np.ndarray, glwe: np.ndarray, n: int):
        P = self.engine.params
        x = _in("pool mul_xn", glwe, np.uint64, P.glwe_words)
        _out("pool mul_xn", output, np.uint64, P.glwe_words)
        self._run("mul_xn", self._lib.spf_pool_submit_mul_xn, _ptr(x), int(n), _ptr(output))

    def multiply_glwe_ggsw(self, output: np.ndarray, glwe: np.ndarray, ggsw: np.ndarray):
        P = self.engine.params
        g = _in("pool multiply_glwe_ggsw", ggsw, np.complex128, P.cbs_ggsw_complex)
        x = _in("pool multiply_glwe_ggsw", glwe, np.uint64, P.glwe_words)
        _out("pool multiply_glwe_ggsw", output, np.uint64, P.glwe_words)
        self._run("multiply_glwe_ggsw", self._lib.spf_pool_submit_multiply_ggsw_glwe, _ptr(g), _ptr(x), _ptr(output))

    def glev_cmux(self, output: np.ndarray, sel: np.ndarray, a: np.ndarray, b: np.ndarray):
        P = self.engine.params
        n = P.cbs_radix_count * P.glwe_words
        s_ = _in("pool glev_cmux", sel, np.complex128, P.cbs_ggsw_complex)
        a_, b_ = _in("pool glev_cmux", a, np.uint64, n), _in("pool glev_cmux", b, np.uint64, n)
        _out("pool glev_cmux", output, np.uint64, n)
        self._run("glev_cmux", self._lib.spf_pool_submit_glev_cmux, _ptr(s_), _ptr(a_), _ptr(b_), _ptr(output))

    def scheme_switch(self, output: np.ndarray, glev: np.ndarray):
        P = self.engine.params
        x = _in("pool scheme_switch", glev, np.uint64, P.cbs_radix_count * P.glwe_words)
        _out("pool scheme_switch", output, np.complex128, P.cbs_ggsw_complex)
        self._run("scheme_switch", self._lib.spf_pool_submit_scheme_switch, _ptr(x), _ptr(output))

    def stats(self):
        ops, launches = C.c_uint64(), C.c_uint64()
        self._lib.spf_pool_stats(self._h, C.byref(ops), C.byref(launches))
        return ops.value, launches.value

    def counters(self) -> dict:
        a = (C.c_uint64 * 11)()
        self._ck(self._lib.spf_pool_counters_get(self._h, C.byref(a)), "spf_pool_counters_get")
        return {"ops": a[0], "launches": a[1], "handle_ops": a[2], "handle_launches": a[3], "reclaimed": a[4],
                "bootstrap_launches_by_shape": {"blind_rotate8": a[5], "blind_rotate2p2": a[6], "blind_rotate2p": a[7]},
                "staging_sets": a[8], "value_mallocs": a[9], "stream_concurrency": a[10]}

    # -- device-resident values (spf_value_*): the operands and results of the `_v` submits ---------------------------------
    def _ck(self, st: int, what: str):
        if st != 0:
            eng = self.engine.member(0) if isinstance(self.engine, Group) else self.engine
            raise SpfError(st, f"{what}: " + (self._lib.spf_last_error(eng._h) or b"").decode())

    _VALUE_DTYPE = {3: np.complex128}

    def upload(self, kind: int, array: np.ndarray, member: int = -1) -> "Value":
        P = self.engine.params
        words = {0: P.lwe0_words, 1: P.lwe1_words, 2: P.glwe_words, 3: 2 * P.cbs_ggsw_complex, 4: P.cbs_radix_count * P.glwe_words}[int(kind)]
        a = np.ascontiguousarray(array)
        if a.nbytes != words * 8:
            raise SpfError(1, f"value of kind {int(kind)} must have {words * 8} bytes, got {a.nbytes}")
        h = C.c_void_p()
        self._ck(self._lib.spf_value_upload(self._h, member, int(kind), _ptr(a), C.byref(h)), "spf_value_upload")
        return Value(self, h)

    def _words(self, kind: int) -> int:
        P = self.engine.params
        return {0: P.lwe0_words, 1: P.lwe1_words, 2: P.glwe_words, 3: 2 * P.cbs_ggsw_complex, 4: P.cbs_radix_count * P.glwe_words}[int(kind)]

    def upload_batch(self, kind: int, arrays: np.ndarray, member: int = -1):
        """`spf_value_upload_batch`: arrays is [n, words of the kind]; one block, one copy -> list of n values"""
        a = np.ascontiguousarray(arrays)
        n = a.shape[0]
        if a.nbytes != n * self._words(kind) * 8:
            raise SpfError(1, f"{n} values of kind {int(kind)} must have {n * self._words(kind) * 8} bytes, got {a.nbytes}")
        hs = (C.c_void_p * n)()
        self._ck(self._lib.spf_value_upload_batch(self._h, member, int(kind), n, _ptr(a), hs), "spf_value_upload_batch")
        return [Value(self, C.c_void_p(h)) for h in hs]

    def download_batch(self, values) -> np.ndarray:
        """`spf_value_download_batch`: -> [n, words] (complex128 for GGSW)"""
        i = values[0].info()
        n = len(values)
        dtype = np.complex128 if i["kind"] == 3 else np.uint64
        out = np.empty((n, i["bytes"] // np.dtype(dtype).itemsize), dtype=dtype)
        hs = (C.c_void_p * n)(*[v._h for v in values])
        self._ck(self._lib.spf_value_download_batch(n, hs, _ptr(out)), "spf_value_download_batch")
        return out

    def trivial(self, kind: int, bit: int, member: int = -1) -> "Value":
        h = C.c_void_p()
        self._ck(self._lib.spf_value_trivial(self._h, member, int(kind), int(bit), C.byref(h)), "spf_value_trivial")
        return Value(self, h)

    def copy_to_member(self, value: "Value", member: int) -> "Value":
        h = C.c_void_p()
        self._ck(self._lib.spf_value_copy_to_member(self._h, value._h, member, C.byref(h)), "spf_value_copy_to_member")
        return Value(self, h)

    def submit_v(self, op: int, inputs, param: int = 0):
        """`spf_pool_submit_op_v`: op is a spf_graph_op (spf_amd.FheOp); returns (result value, ticket) — the value is valid
        once wait(ticket) has returned"""
        arr = (C.c_void_p * len(inputs))(*[v._h for v in inputs])
        h, t = C.c_void_p(), C.c_uint64()
        self._ck(self._lib.spf_pool_submit_op_v(self._h, int(op), arr, len(inputs), int(param), C.byref(h), C.byref(t)), "spf_pool_submit_op_v")
        return Value(self, h), t.value

    def push_v(self, op: int, inputs, param: int = 0) -> "Value":
        """`spf_pool_submit_op_v` without a ticket: the operands may be results that are still pending (the pool orders and
        batches what is pushed by level); Value.wait() on the result — or on anything computed from it — makes it run"""
        arr = (C.c_void_p * len(inputs))(*[v._h for v in inputs])
        h = C.c_void_p()
        self._ck(self._lib.spf_pool_submit_op_v(self._h, int(op), arr, len(inputs), int(param), C.byref(h), None), "spf_pool_submit_op_v")
        return Value(self, h)

    def flush(self):
        """`spf_pool_flush`: what has been pushed so far is launched (nothing is waited for)"""
        self._ck(self._lib.spf_pool_flush(self._h), "spf_pool_flush")

    def run_v(self, op: int, inputs, param: int = 0) -> "Value":
        v, t = self.submit_v(op, inputs, param)
        try:
            self._wait(t)
        except SpfError:
            v.release()
            raise
        return v

    def keyswitch_circuit_bootstrap_v(self, lwe1: "Value") -> "Value":
        h, t = C.c_void_p(), C.c_uint64()
        self._ck(self._lib.spf_pool_submit_keyswitch_circuit_bootstrap_v(self._h, lwe1._h, C.byref(h), C.byref(t)),
                 "spf_pool_submit_keyswitch_circuit_bootstrap_v")
        v = Value(self, h)
        self._wait(t.value)
        return v

    def value_stats(self) -> dict:
        a, b, c = C.c_size_t(), C.c_size_t(), C.c_size_t()
        self._ck(self._lib.spf_pool_value_stats(self._h, C.byref(a), C.byref(b), C.byref(c)), "spf_pool_value_stats")
        return {"live_values": a.value, "live_bytes": b.value, "cached_bytes": c.value}

    def trim(self):
        self._ck(self._lib.spf_pool_trim(self._h), "spf_pool_trim")


class Value:
    """one device-resident ciphertext (`spf_value`): holds ONE reference, dropped by release() / garbage collection"""

    def __init__(self, pool: Pool, handle):
        self._pool = pool
        self._lib = pool._lib
        self._h = handle

    def info(self) -> dict:
        k, b, m, r = C.c_int(), C.c_size_t(), C.c_int(), C.c_int()
        st = self._lib.spf_value_info(self._h, C.byref(k), C.byref(b), C.byref(m), C.byref(r))
        if st != 0:
            raise SpfError(st, "spf_value_info")
        return {"kind": k.value, "bytes": b.value, "member": m.value, "valid": bool(r.value)}

    def download(self) -> np.ndarray:
        i = self.info()
        out = np.empty(i["bytes"] // (16 if i["kind"] == 3 else 8), dtype=np.complex128 if i["kind"] == 3 else np.uint64)
        st = self._lib.spf_value_download(self._h, _ptr(out))
        if st != 0:
            raise SpfError(st, "spf_value_download: the value is not valid (wait for its ticket first)")
        return out

    def wait(self) -> "Value":
        """`spf_value_wait`: until the operation that produces the value has run"""
        st = self._lib.spf_value_wait(self._h)
        if st != 0:
            raise SpfError(st, "spf_value_wait: the producing operation failed")
        return self

    def device_ptr(self) -> int:
        p = C.c_void_p()
        st = self._lib.spf_value_device_ptr(self._h, C.byref(p))
        if st != 0:
            raise SpfError(st, "spf_value_device_ptr: the value is not valid")
        return p.value

    def release(self):
        if getattr(self, "_h", None):
            self._lib.spf_value_release(self._h)
            self._h = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass
