"""Host-side mirror of ``parasol_runtime::Evaluation`` / ``KeylessEvaluation``
(parasol_runtime/src/crypto/evaluation.rs:26-266) for the bootstrap path.

Method names, argument order and meaning follow the reference: the caller allocates ``output``
and the method writes into it (the reference takes ``&mut`` outputs and returns nothing).
Inputs may be single ciphertexts (1-D arrays, as the reference's typed newtypes) or batches
(2-D, leading dimension = batch): the engine underneath is batch-native.  All compute happens in
the HIP library; nothing here touches ciphertext words.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from ._ffi import Engine
from .params import Params, DEFAULT_128


@dataclass
class ComputeKey:
    """``parasol_runtime::ComputeKey`` (crypto/keys.rs:306-318): the fields this path uses."""

    bs_key: np.ndarray   # BootstrapKeyFft<Complex<f64>>, complex128, reference layout
    ks_key: np.ndarray   # LweKeyswitchKey<u64>
    auto_key: np.ndarray = None  # AutomorphismKeyFft<Complex<f64>>
    ss_key: np.ndarray = None    # SchemeSwitchKeyFft<Complex<f64>>


class Evaluation:
    def __init__(self, compute_key: ComputeKey, params: Params = DEFAULT_128, device: int = 0):
        # Evaluation::new (evaluation.rs:161-197): load the keys; the two circuit bootstraps that give
        # l1ggsw_zero / l1ggsw_one run on the GPU at first use (they need all of bs / auto / ss keys)
        self.params = params
        self.engine = Engine(params, device)
        self.engine.load_bootstrap_key(compute_key.bs_key)
        if compute_key.ks_key is not None and np.size(compute_key.ks_key):
            self.engine.load_keyswitch_key(compute_key.ks_key)
        if compute_key.auto_key is not None:
            self.engine.load_automorphism_key(compute_key.auto_key)
        if compute_key.ss_key is not None:
            self.engine.load_scheme_switch_key(compute_key.ss_key)

    # Evaluation::l1ggsw_zero / l1ggsw_one (evaluation.rs:254-262)
    def l1ggsw_zero(self) -> np.ndarray:
        return self.engine.l1ggsw_constant(0)

    def l1ggsw_one(self) -> np.ndarray:
        return self.engine.l1ggsw_constant(1)

    @staticmethod
    def _store(output: np.ndarray, result: np.ndarray):
        if output.dtype != np.uint64:
            raise TypeError("output must be uint64")
        output[...] = result.reshape(output.shape)

    # KeylessEvaluation::not (evaluation.rs:47-50); `not` is a Python keyword, hence the underscore
    def not_(self, output: np.ndarray, input: np.ndarray):
        self._store(output, self.engine.glwe_not(input))

    # KeylessEvaluation::xor (evaluation.rs:52-55)
    def xor(self, output: np.ndarray, a: np.ndarray, b: np.ndarray):
        self._store(output, self.engine.glwe_xor(a, b))

    # KeylessEvaluation::mul_xn (evaluation.rs:57-65)
    def mul_xn(self, output: np.ndarray, input: np.ndarray, n: int):
        self._store(output, self.engine.glwe_mul_xn(input, n))

    # KeylessEvaluation::sample_extract_l1 (evaluation.rs:126-133)
    def sample_extract_l1(self, output: np.ndarray, input: np.ndarray, idx: int):
        self._store(output, self.engine.sample_extract_l1(input, idx))

    # Evaluation::keyswitch_lwe_l1_lwe_l0 (evaluation.rs:246-255)
    def keyswitch_lwe_l1_lwe_l0(self, output: np.ndarray, input: np.ndarray):
        self._store(output, self.engine.keyswitch_lwe_l1_lwe_l0(input))

    # Evaluation::circuit_bootstrap (evaluation.rs:211-226): L0 LWE -> L1 GGSW (FFT domain)
    def circuit_bootstrap(self, output: np.ndarray, input: np.ndarray):
        if output.dtype != np.complex128:
            raise TypeError("an L1 GGSW ciphertext is complex128 (FFT domain)")
        output[...] = self.engine.circuit_bootstrap(input).reshape(output.shape)

    # Evaluation::scheme_switch (evaluation.rs:231-240): L1 GLEV -> L1 GGSW
    def scheme_switch(self, output: np.ndarray, input: np.ndarray):
        output[...] = self.engine.scheme_switch(input).reshape(output.shape)

    # the bootstrap stage of Evaluation::circuit_bootstrap (evaluation.rs:211-226):
    # hi_noise_lwe_to_lo_noise_glwe (circuit_bootstrapping.rs:387-427).  Output: L1 GLWE whose
    # first cbs_radix.count coefficients hold the gadget levels of the input bit.
    def circuit_bootstrap_pbs(self, output: np.ndarray, input: np.ndarray):
        self._store(output, self.engine.circuit_bootstrap_pbs(input))

    # sunscreen_tfhe::ops::bootstrapping::programmable_bootstrap_univariate
    # (programmable_bootstrapping.rs:291-318)
    def programmable_bootstrap_univariate(self, output: np.ndarray, input: np.ndarray, lut: np.ndarray):
        self._store(output, self.engine.pbs_univariate(input, lut))

    # generalized_programmable_bootstrap (programmable_bootstrapping.rs:342-410)
    def generalized_programmable_bootstrap(self, output, input, lut, log_chi: int, log_v: int):
        self._store(output, self.engine.generalized_pbs(input, lut, log_chi, log_v, 0))

    # KeylessEvaluation::cmux (evaluation.rs:68-83)
    def cmux(self, output: np.ndarray, sel: np.ndarray, a: np.ndarray, b: np.ndarray):
        self._store(output, self.engine.cmux(sel, a, b))

    # KeylessEvaluation::glev_cmux (evaluation.rs:86-101): L1 GLEV = cbs_radix.count consecutive GLWEs
    def glev_cmux(self, output: np.ndarray, sel: np.ndarray, a: np.ndarray, b: np.ndarray):
        self._store(output, self.engine.glev_cmux(sel, a, b))

    # KeylessEvaluation::multiply_glwe_ggsw (evaluation.rs:104-123)
    def multiply_glwe_ggsw(self, output: np.ndarray, glwe: np.ndarray, ggsw: np.ndarray):
        self._store(output, self.engine.multiply_glwe_ggsw(glwe, ggsw))

    # FheOp::KeyswitchL1toL0 -> FheOp::CircuitBootstrap fused (circuit_processor/mod.rs:329-340,453-463)
    def gate_bootstrap(self, output: np.ndarray, input_l1: np.ndarray):
        self._store(output, self.engine.gate_bootstrap(input_l1))
