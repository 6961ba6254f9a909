"""The C++ host mirror (include/spf_evaluation.hpp) exercised from native code: a C++ program built
with g++ against the product library and checked, word for word, against the C oracle."""
import os
import subprocess

import pytest

import oracle as O
import spf_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_cpp_evaluation_and_fhe_circuit_match_the_oracle(tmp_path):
    libdir = os.path.dirname(spf_amd.lib_path())
    oracle_so = O.library_path()
    exe = tmp_path / "evaluation_parity"
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-pthread", "-I", os.path.join(ROOT, "include"),
                    "-I", os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests", "cpp", "evaluation_parity.cpp"),
                    "-o", str(exe), "-L", libdir, "-lspf_hip", oracle_so,
                    "-Wl,-rpath," + libdir, "-Wl,-rpath," + os.path.dirname(oracle_so)], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all equal" in r.stdout
