"""Host-side self-test of csrc/glphi.hpp: Goldilocks elements as a + b*2^32 with signed 64-bit
components, the arithmetic of the opt-in `NTT_PHI` butterfly networks of csrc/ntt.hip (plonky2
field/src/fft.rs `fft_classic` as used by `circuit_data.prove`, prove_action.rs:96).  Every network
shape the kernels instantiate is run on the CPU against a from-the-definition DFT in canonical
arithmetic, on random and extreme words, with the run-time magnitudes checked against the compile-time
bounds that place the renormalisations; once more under UBSan (signed overflow is the failure mode)."""
import os
import subprocess

import pytest

from conftest import ROOT

SRC = os.path.join(ROOT, "acvm-backend-plonky2_amd", "csrc", "tests", "phi_selftest.cpp")


@pytest.mark.parametrize("flags,iters", [(["-O2"], "1500"), (["-O1", "-g", "-fsanitize=undefined,address", "-fno-sanitize-recover=undefined"], "150")])
def test_phi_networks_against_the_definition(tmp_path, flags, iters):
    exe = str(tmp_path / "phi_selftest")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", *flags, "-o", exe, SRC])
    r = subprocess.run([exe, iters], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "failures: 0" in r.stdout and "runtime error" not in r.stderr
