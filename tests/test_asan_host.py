"""Sanitizers on the host code: tools/asan_host.sh builds the library's translation unit with
AddressSanitizer + UndefinedBehaviorSanitizer for the HOST side and runs the mutation fuzz of tools/fuzz_host.cpp over the
untrusted-bytes parsers, the parameter validators and the graph builder — seeded with the reference's own malformed vector
(parasol_runtime/src/safe_bincode.rs:58-66).  Short form here (no GPU); the 10^6-case soak is logged in profiles/r06_asan_fuzz.md.
GPU ASan / XNACK are not available on this pool and are not attempted."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def fuzz_binary():
    # (builds on first use: the whole translation unit, ~50 s)
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "asan_host.sh"), "0"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return os.path.join(ROOT, "tools", "bin", "fuzz_host")


def test_host_fuzz_under_asan_and_ubsan(fuzz_binary):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([fuzz_binary, "15000", "20261005"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "15000 cases" in r.stdout and "no sanitizer report" in r.stdout
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr


def test_the_sanitizer_watches_the_library(fuzz_binary):
    """a caller that lies about `len` makes the parser read past its buffer: the instrumented build must report it"""
    env = dict(os.environ, ASAN_OPTIONS="abort_on_error=0:exitcode=66")
    r = subprocess.run([fuzz_binary, "--selftest-overflow"], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode != 0 and "heap-buffer-overflow" in r.stderr, r.stdout + r.stderr[-2000:]
    assert "spf_ciphertext_from_bincode" in r.stderr
