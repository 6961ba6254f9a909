"""CPU-side checks of the boundary: the C-ABI library builds for gfx950, loads, exports every
symbol include/spf_hip.h declares, and refuses to run without a GPU (no silent fallback)."""
import os
import re

import pytest

import spf_amd
from spf_amd import _ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    spf_amd.build_library()
    return spf_amd.load_library()


def test_header_symbols_all_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "spf_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(spf_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 20
    bound = {name for name, _, _ in _ffi.SYMBOLS}
    assert declared == bound, (declared ^ bound)
    for name in declared:
        assert hasattr(lib, name), name


def test_default_params_match_reference(lib):
    p = _ffi._CParams()
    lib.spf_default_params(p)
    got = {n: getattr(p, n) for n, _ in _ffi._CParams._fields_}
    assert got == dict(lwe_dimension=637, polynomial_degree=2048, glwe_size=1, pbs_radix_log=16,
                       pbs_radix_count=2, cbs_radix_log=4, cbs_radix_count=4, ks_radix_log=2,
                       ks_radix_count=6, tr_radix_log=7, tr_radix_count=6, ss_radix_log=3,
                       ss_radix_count=15)  # parasol_runtime/src/params.rs:107-134
    assert spf_amd.DEFAULT_128.bsk_complex * 16 == 83_492_864
    assert spf_amd.DEFAULT_128.ksk_words * 8 == 62_717_952


def test_unsupported_params_rejected(lib):
    import ctypes as C
    p = _ffi._CParams()
    lib.spf_default_params(p)
    p.polynomial_degree = 4096           # beyond the tuned kernels (2048) and the generic ones (16 .. 2048)
    h = C.c_void_p()
    st = lib.spf_create(p, 0, h)
    assert st == 4 and not h.value
    assert b"2048" in lib.spf_last_error(None)
    lib.spf_default_params(p)
    p.glwe_size = 3                      # (k+1) polynomials of degree 2048 do not fit the generic kernels' LDS
    st = lib.spf_create(p, 0, h)
    assert st == 4 and not h.value and b"LDS" in lib.spf_last_error(None)
    lib.spf_default_params(p)
    p.pbs_radix_log, p.pbs_radix_count = 16, 4   # l * log B must stay below 64
    st = lib.spf_create(p, 0, h)
    assert st == 4 and not h.value


def test_no_gpu_means_loud_failure():
    from tests.util import gpu_available
    if gpu_available():
        pytest.skip("GPU present")
    with pytest.raises(spf_amd.SpfError):
        spf_amd.Engine()


def test_group_entry_points_validate_before_touching_a_device(lib):
    """spf_group_*: null / empty arguments are refused with a message, and without a GPU creation fails loudly (no fallback)."""
    import ctypes as C
    from tests.util import gpu_available
    p = _ffi._CParams()
    lib.spf_default_params(p)
    h = C.c_void_p()
    ids = (C.c_int * 2)(0, 0)
    assert lib.spf_group_create(p, ids, 0, h) == 1 and not h.value          # no devices
    assert lib.spf_group_create(p, None, 2, h) == 1 and not h.value         # null list
    assert b"n_devices" in lib.spf_last_error(None)
    assert lib.spf_group_size(None) == 0 and lib.spf_group_ctx(None, 0) is None
    assert lib.spf_group_replicate_keys(None) == 1
    assert lib.spf_group_gate_bootstrap_batch(None, 1, None, None) == 1
    if not gpu_available():
        st = lib.spf_group_create(p, ids, 2, h)
        assert st == 2 and not h.value and b"no HIP device" in lib.spf_last_error(None)
        with pytest.raises(spf_amd.SpfError):
            spf_amd.Group(devices=[0, 0])


def test_product_never_imports_oracle():
    # the oracle is test infrastructure: nothing under spf_amd/ may reference it
    for dirpath, _, files in os.walk(os.path.join(ROOT, "spf_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "spf_oracle" not in src, f


def test_c_header_is_plain_c_and_cpp_mirror_compiles(tmp_path):
    import subprocess
    inc = os.path.join(ROOT, "include")
    c = tmp_path / "t.c"
    c.write_text('#include "spf_hip.h"\nint main(void){spf_params p; spf_default_params(&p); return (int)p.glwe_size - 1;}\n')
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", inc, "-fsyntax-only", str(c)], check=True)
    cpp = tmp_path / "t.cpp"
    cpp.write_text('#include "spf_evaluation.hpp"\nint main(){ spf::ComputeKey k{nullptr,0,nullptr,0,nullptr,0,nullptr,0}; (void)k; return 0; }\n')
    subprocess.run(["g++", "-std=c++17", "-Wall", "-Werror", "-I", inc, "-fsyntax-only", str(cpp)], check=True)
    # the typed device ciphertexts: the by-handle mirror instantiates, and a wrong operand type is a compile error
    # (as `&L1GlweCiphertext` for a `&L1GgswCiphertext` is in the reference)
    body = ('#include "spf_evaluation.hpp"\nvoid f(spf::PooledEvaluation& pe, spf::L1GgswCiphertext& s, spf::L1GlweCiphertext& a,'
            ' spf::L1GlweCiphertext& o, spf::L1LweCiphertext& e1, spf::L0LweCiphertext& e0){ pe.sample_extract_l1(e1, a, 0);'
            ' pe.keyswitch_lwe_l1_lwe_l0(e0, e1); pe.circuit_bootstrap(s, e0); %s }\nint main(){return 0;}\n')
    good, bad = tmp_path / "good.cpp", tmp_path / "bad.cpp"
    good.write_text(body % "pe.cmux(o, s, a, a);")
    bad.write_text(body % "pe.cmux(o, a, a, a);")
    subprocess.run(["g++", "-std=c++17", "-Wall", "-Werror", "-I", inc, "-fsyntax-only", str(good)], check=True)
    assert subprocess.run(["g++", "-std=c++17", "-I", inc, "-fsyntax-only", str(bad)], capture_output=True).returncode != 0
    # and a real link + run against the built library (no GPU call: version + params only)
    exe = tmp_path / "t"
    subprocess.run(["gcc", "-std=c99", "-I", inc, str(c), "-o", str(exe), "-L", os.path.dirname(spf_amd.lib_path()),
                    "-lspf_hip", "-Wl,-rpath," + os.path.dirname(spf_amd.lib_path())], check=True)
    assert subprocess.run([str(exe)]).returncode == 0


def test_ffi_operand_validation_helpers():
    """a batch mismatch or a wrong pool output buffer must raise before any pointer is taken
    (the C side copies B * row_size bytes)."""
    import numpy as np
    from spf_amd import _ffi
    a, b = np.zeros((3, 8), dtype=np.uint64), np.zeros((4, 8), dtype=np.uint64)
    with pytest.raises(spf_amd.SpfError):
        _ffi._same_rows("cmux", a, b)
    _ffi._same_rows("cmux", a, a)
    good = np.zeros(16, dtype=np.uint64)
    assert _ffi._out("x", good, np.uint64, 16) is good
    for bad in (np.zeros(15, dtype=np.uint64), np.zeros(16, dtype=np.int32), np.zeros((16, 2), dtype=np.uint64)[:, 0],
                [0] * 16):
        with pytest.raises(spf_amd.SpfError):
            _ffi._out("x", bad, np.uint64, 16)
    ro = np.zeros(16, dtype=np.uint64)
    ro.setflags(write=False)
    with pytest.raises(spf_amd.SpfError):
        _ffi._out("x", ro, np.uint64, 16)
    with pytest.raises(spf_amd.SpfError):
        _ffi._in("x", np.zeros(5), np.uint64, 4)


def test_generate_lut_matches_the_oracle_and_rejects_out_of_range_maps():
    """a4': `generate_lut` (programmable_bootstrapping.rs:129-185) on the product side, host only."""
    import numpy as np
    import oracle as O
    P = O.DEFAULT_128
    for bits, fns in [(1, [lambda x: (x + 1) % 2]), (3, [lambda x: (x + 3) % 8]), (4, [lambda x: (5 * x + 1) % 16]),
                      (2, [lambda x: x, lambda x: (3 * x) % 4, lambda x: (x + 1) % 4])]:
        got = spf_amd.generate_lut(fns, bits)
        assert np.array_equal(got, O.trivial_lut_glwe(O.generate_lut(P.N, fns, bits), P)), (bits, len(fns))
    with pytest.raises(spf_amd.SpfError):
        spf_amd.generate_lut([lambda x: 8], 3)


def test_generated_rust_binding_covers_the_header(lib):
    """tools/gen_rust_ffi.py (VERDICT r05 task 4): the committed include/spf_hip.rs is what the generator makes of the header
    today, holds every function, struct and constant the header declares with pointer constness carried over, and the library
    exports exactly the header's functions (`make -C spf_amd/csrc check`)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_rust_ffi", os.path.join(ROOT, "tools", "gen_rust_ffi.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    hdr_path = os.path.join(ROOT, "include", "spf_hip.h")
    h = gen.Header(hdr_path)
    rust = h.rust()
    assert rust == open(os.path.join(ROOT, "include", "spf_hip.rs")).read(), "include/spf_hip.rs is stale: make -C spf_amd/csrc rust-ffi"
    hdr = re.sub(r"/\*.*?\*/", "", open(hdr_path).read(), flags=re.S)
    declared = set(re.findall(r"\b(spf_[a-z0-9_]+)\s*\(", hdr))
    assert {n for n, _, _ in h.functions} == declared
    for name in declared:
        assert re.search(rf"pub fn {name}\(", rust), name
    for const in re.findall(r"\b(SPF_[A-Z0-9_]+)\b", hdr):
        if const != "SPF_HIP_H":
            assert f"pub const {const}:" in rust, const
    # spot checks of the type mapping
    assert "pub fn spf_create(params: *const spf_params, device_id: c_int, out: *mut *mut spf_ctx) -> spf_status;" in rust
    assert "pub fn spf_destroy(ctx: *mut spf_ctx);" in rust
    assert "pub fn spf_cmux_scattered_dev(ctx: *mut spf_ctx, stream: *mut c_void, units: usize, d_ptrs: *const *const c_void) -> spf_status;" in rust
    assert "pub fn spf_pool_submit_cmux_v(pool: *mut spf_pool, sel_ggsw: *const spf_value, a: *const spf_value, b: *const spf_value, out: *mut *mut spf_value, ticket: *mut u64) -> spf_status;" in rust
    assert "pub bootstrap_launches_by_shape: [u64; 3]," in rust
    assert gen.exported_symbols(spf_amd.lib_path()) == sorted(declared)


def test_makefile_describes_the_same_build():
    """spf_amd/csrc/Makefile (the build a non-Python host uses) compiles the same translation unit with the same flags as
    spf_amd/build.py; `make -n` resolves (hipcc need not run here)."""
    import subprocess
    from spf_amd import build
    mk = os.path.join(ROOT, "spf_amd", "csrc", "Makefile")
    out = subprocess.run(["make", "-n", "-B", "-f", mk, "all"], capture_output=True, text=True, check=True).stdout
    line = next(l for l in out.splitlines() if "hipcc" in l)
    for flag in build.HIPCC_FLAGS:
        assert flag in line.split(), flag
    assert line.rstrip().endswith("spf_hip.hip") and "libspf_hip.so.tmp" in line
    pc = subprocess.run(["make", "-n", "-f", mk, "install", "PREFIX=/tmp/spf_prefix"], capture_output=True, text=True, check=True).stdout
    assert "spf_hip.pc" in pc and "/tmp/spf_prefix/include" in pc and "spf_evaluation.hpp" in pc


def test_integration_md_mentions_every_entry_point():
    """the integration guide names every function of the header (its §6 index is generated)."""
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "spf_hip.h")).read(), flags=re.S)
    declared = set(re.findall(r"\b(spf_[a-z0-9_]+)\s*\(", hdr))
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    missing = sorted(n for n in declared if f"`{n}`" not in text)
    assert not missing, missing
