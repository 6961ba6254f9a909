#!/usr/bin/env python3
"""Generates the committed golden vectors under tests/golden/ from the CPU oracle.

The reference (Rust) cannot be built or run in this environment and holds no golden ciphertext
vectors for this path (its tests use an unseeded thread_rng), so these fixtures pin the
*oracle's* outputs: a regression net under both the oracle and the HIP kernels.  Inputs are
stored explicitly (not re-derived from a PRNG) so the files stay valid if generators change.

    python tests/golden/gen_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle as O  # noqa: E402


def main():
    P = O.DEFAULT_128.replace(lwe_n=2)
    keys = O.gen_keyset(0x60D, P)
    rng = np.random.default_rng(0x60D)

    # 1. one CMUX at the PBS shape (N=2048, k=1, l=2, logB=16): fft_ops.rs:149-181
    g = O.encrypt_ggsw_fft(O.Rng(11), keys.glwe_sk, 1, P.N, P.k, P.pbs_radix_log, P.pbs_count, P.glwe_std)
    d0 = rng.integers(0, 1 << 64, P.glwe_len, dtype=np.uint64)
    d1 = rng.integers(0, 1 << 64, P.glwe_len, dtype=np.uint64)
    out = O.cmux(d0, d1, g, P.N, P.k, P.pbs_radix_log, P.pbs_count)
    np.savez_compressed(os.path.join(HERE, "cmux_pbs_shape.npz"), ggsw_fft=g, d0=d0, d1=d1, out=out)

    # 2. two-step bootstraps (n = 2): generalized PBS with the CBS LUT and with a random LUT
    lwe = rng.integers(0, 1 << 64, (3, P.lwe_n + 1), dtype=np.uint64)
    lut = rng.integers(0, 1 << 64, P.glwe_len, dtype=np.uint64)
    cbs = np.stack([O.cbs_pbs(x, keys.bsk_fft, P) for x in lwe])
    gen = np.stack([O.generalized_pbs(x, lut, keys.bsk_fft, P, 1, 1) for x in lwe])
    uni = np.stack([O.pbs_univariate(x, lut, keys.bsk_fft, P) for x in lwe])
    np.savez_compressed(os.path.join(HERE, "pbs_n2.npz"), bsk_fft=keys.bsk_fft, lwe=lwe, lut=lut,
                        cbs_out=cbs, gen_out_chi1_v1=gen, univariate_out=uni)

    # 3. LWE keyswitch 2048 -> 2 (ks_radix 6 x 2 bits)
    lwe1 = rng.integers(0, 1 << 64, (4, P.N + 1), dtype=np.uint64)
    ks = np.stack([O.keyswitch_lwe(x, keys.ksk, P.N, P.lwe_n, P.ks_radix_log, P.ks_count) for x in lwe1])
    np.savez_compressed(os.path.join(HERE, "keyswitch_n2.npz"), ksk=keys.ksk, lwe1=lwe1, out=ks)

    # 4. the canonical FFT-1024 and its twiddles on a fixed vector (pins DAG-I itself)
    x = (rng.standard_normal(1024) + 1j * rng.standard_normal(1024)) * 2.0 ** 40
    np.savez_compressed(os.path.join(HERE, "fft1024.npz"), x=x, fwd=O.fft1024(x, +1), inv=O.fft1024(x, -1),
                        twist=np.array([O.root_of_unity(j, 4096) for j in range(1024)]))
    # 5. DEFAULT_128 (n = 637): three ciphertexts in / out.  The 83 MB key is not stored: it is re-derived
    #    from the seed recipe (O.gen_keyset(key_seed, DEFAULT_128)), pinned by a checksum of its bits.
    P128 = O.DEFAULT_128
    k128 = O.gen_keyset(0x5EED0001, P128, with_ksk=False)
    lwe128 = O.encrypt_bits_l0(0x5EED0200, k128, [0, 1, 1])
    lwe128[2] = rng.integers(0, 1 << 64, P128.lwe_n + 1, dtype=np.uint64)   # and one arbitrary word vector
    out128 = np.stack([O.cbs_pbs(x, k128.bsk_fft, P128) for x in lwe128])
    #    plus the plain PBS (log_chi = 0, log_v = 0: the other instantiation of every blind-rotation kernel): the
    #    negacyclic identity LUT of one plaintext bit + padding (generate_lut, programmable_bootstrapping.rs:129-185)
    #    as GLWE out and through sample_extract(., 0) (programmable_bootstrap_univariate, :291-318)
    plain_lut = O.trivial_lut_glwe(O.generate_lut(P128.N, [lambda x: x], 1), P128)
    #    inputs: the bits 0, 1 with a padding bit (message at 2^62, as the reference's PBS tests encode them) and the
    #    arbitrary word vector again
    plain_lwe = np.stack([O.encrypt_lwe(O.Rng(0x5EED0300 + b), k128.lwe_sk, O.encode(b, 2), P128.lwe_std) for b in (0, 1)]
                         + [lwe128[2]])
    gen128 = np.stack([O.generalized_pbs(x, plain_lut, k128.bsk_fft, P128, 0, 0) for x in plain_lwe])
    uni128 = np.stack([O.pbs_univariate(x, plain_lut, k128.bsk_fft, P128) for x in plain_lwe])
    np.savez_compressed(os.path.join(HERE, "pbs_default128.npz"), key_seed=np.uint64(0x5EED0001),
                        lwe_n=np.uint32(P128.lwe_n), lwe=lwe128, cbs_out=out128, plain_lut=plain_lut, plain_lwe=plain_lwe,
                        gen_out_chi0_v0=gen128, univariate_out=uni128,
                        bsk_checksum=k128.bsk_fft.view(np.uint64).sum(dtype=np.uint64))
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
