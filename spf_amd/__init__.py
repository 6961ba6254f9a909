"""spf_amd — MI355X (gfx950) engine for the programmable-bootstrap path of Sunscreen-tech/spf.

The compute lives in hand-written HIP kernels behind a C ABI (include/spf_hip.h,
spf_amd/csrc/).  This package is plumbing over that ABI: a ctypes binding and a host-side
mirror of ``parasol_runtime::Evaluation`` (parasol_runtime/src/crypto/evaluation.rs:144-266).
There is no CPU fallback: importing works anywhere, but creating an ``Engine`` without the
compiled library or without a GPU raises.
"""
import os as _os

# read by the HIP runtime when it initialises; the library's loader sets the same default, but a process may touch the GPU (torch)
# between importing this package and opening the library (spf_amd/csrc/spf_hip.hip, spf_ask_for_hw_queues)
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")

from .params import Params, DEFAULT_128  # noqa: F401,E402
from ._ffi import (Engine, Group, Pool, SpfError, Value, ciphertext_from_bincode, ciphertext_to_bincode, ciphertext_words,  # noqa: F401
                   generate_lut, lib_path, load_library)
from .evaluation import Evaluation, ComputeKey  # noqa: F401
from .graph import FheCircuit, FheOp, RecordedCircuit, ValueKind  # noqa: F401
from .build import build_library  # noqa: F401

__all__ = ["Params", "DEFAULT_128", "Engine", "Group", "Pool", "Value", "SpfError", "Evaluation", "ComputeKey", "FheCircuit", "FheOp", "ValueKind", "RecordedCircuit",
           "build_library", "ciphertext_from_bincode", "ciphertext_to_bincode", "ciphertext_words", "generate_lut", "lib_path", "load_library"]
