"""Mirror of ``parasol_runtime::Params`` (parasol_runtime/src/params.rs:10-134) restricted to the
fields the bootstrap / keyswitch path reads.  Field names follow include/spf_hip.h."""
from __future__ import annotations

from dataclasses import dataclass, replace


@dataclass(frozen=True)
class Params:
    lwe_dimension: int = 637        # l0_params (sunscreen_tfhe/src/params.rs:219-222)
    polynomial_degree: int = 2048   # l1_params (sunscreen_tfhe/src/params.rs:258-264)
    glwe_size: int = 1
    pbs_radix_log: int = 16         # parasol_runtime/src/params.rs:114-117
    pbs_radix_count: int = 2
    cbs_radix_log: int = 4          # :110-113
    cbs_radix_count: int = 4
    ks_radix_log: int = 2           # :122-125
    ks_radix_count: int = 6
    tr_radix_log: int = 7           # :130-133
    tr_radix_count: int = 6
    ss_radix_log: int = 3           # :126-129
    ss_radix_count: int = 15

    # derived sizes, in 64-bit words / complex bins
    @property
    def lwe0_words(self) -> int:
        return self.lwe_dimension + 1

    @property
    def lwe1_words(self) -> int:
        return self.glwe_size * self.polynomial_degree + 1

    @property
    def glwe_words(self) -> int:
        return (self.glwe_size + 1) * self.polynomial_degree

    @property
    def bsk_complex(self) -> int:
        k1 = self.glwe_size + 1
        return self.lwe_dimension * k1 * self.pbs_radix_count * k1 * (self.polynomial_degree // 2)

    @property
    def ksk_words(self) -> int:
        return self.glwe_size * self.polynomial_degree * self.ks_radix_count * self.lwe0_words

    @property
    def cbs_ggsw_complex(self) -> int:
        k1 = self.glwe_size + 1
        return k1 * self.cbs_radix_count * k1 * (self.polynomial_degree // 2)

    @property
    def ak_complex(self) -> int:
        k1 = self.glwe_size + 1
        logn = self.polynomial_degree.bit_length() - 1
        return logn * self.glwe_size * self.tr_radix_count * k1 * (self.polynomial_degree // 2)

    @property
    def ssk_complex(self) -> int:
        k = self.glwe_size
        return (k * (k + 1) // 2) * self.ss_radix_count * (k + 1) * (self.polynomial_degree // 2)

    def replace(self, **kw) -> "Params":
        return replace(self, **kw)


DEFAULT_128 = Params()
