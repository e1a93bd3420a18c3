"""CPU-side hardening (the sanitizer row): the host-only parsers of the
product -- verifier-key blob, uncompressed proofs, the reference's compressed on-disk format -- and
the oracle, under mutation fuzzing, normally and under AddressSanitizer + UBSan.  None of this needs a
GPU: the verifier side of the library is plain host code (`p2gpu_verifier_create`, `p2gpu_verify`,
`p2gpu_proof_{de,}compress`), the counterpart of the reference's `verify` action
(plonky2-backend/src/actions/verify_action.rs:11-17)."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from test_verifier import vk_blob

CSRC = os.path.join(ROOT, "acvm-backend-plonky2_amd", "csrc")


@pytest.fixture(scope="module")
def corpus(pkg, orc, tmp_path_factory):
    """One accepted (vk, proof, compressed proof) triple of a circuit with every gate kind and public inputs."""
    d = tmp_path_factory.mktemp("fuzz")
    blob, wires, pis = pkg.make_circuit(7, "ecdsa", 4, num_public_inputs=3)
    oc = orc.OracleCircuit(blob)
    vk = vk_blob(blob, oc.cap(), oc.digest())
    proof, _ = oc.prove(wires, public_inputs=pis)
    comp = pkg.VerifierCircuitData(vk).compress(proof)
    for name, data in (("vk.blob", vk), ("proof.bin", proof), ("comp.bin", comp)):
        (d / name).write_bytes(data)
    return d, vk, proof


def _gate_words(vk, index):
    return np.frombuffer(vk[256 + 48 * index:256 + 48 * (index + 1)], dtype=np.uint32).copy()


def test_blob_gate_parameters_are_validated(pkg, corpus):
    """a vk blob with ComparisonGate num_chunks = 0 used to divide by zero inside
    p2gpu_verifier_create, and an ArithmeticGate with 2^20 ops indexed wires far beyond the row."""
    _, vk, proof = corpus
    hdr = np.frombuffer(vk[:256], dtype=np.uint32)
    kinds = [int(_gate_words(vk, i)[0]) for i in range(int(hdr[23]))]

    def with_gate(i, **kw):
        g = _gate_words(vk, i)
        for k, v in kw.items():
            g[{"kind": 0, "p0": 1, "p1": 2, "p2": 3, "ncons": 8, "nconst": 10, "gs": 6, "ge": 7, "deg": 9}[k]] = v
        return vk[:256 + 48 * i] + g.tobytes() + vk[256 + 48 * (i + 1):]

    bad = [
        with_gate(kinds.index(11), p1=0),                                  # ComparisonGate: num_chunks = 0 (SIGFPE before)
        with_gate(kinds.index(11), p0=64, p1=1, ncons=6 + 5 + 64),         # chunk_bits = 64
        with_gate(kinds.index(3), p0=1 << 20, ncons=1 << 20),              # ArithmeticGate far wider than the row (OOB before)
        with_gate(kinds.index(3), p0=0, ncons=0),
        with_gate(kinds.index(4), p0=1),                                   # BaseSum base 1
        with_gate(kinds.index(5), p0=7),                                   # RandomAccess bits 7
        with_gate(kinds.index(5), p1=0, ncons=int(_gate_words(vk, kinds.index(5))[3])),
        with_gate(kinds.index(7), p0=7, ncons=7 * 36),                     # 7 U32Arithmetic ops need 266 wires > 234
        with_gate(kinds.index(10), p0=14, ncons=14 * 17),                  # 14 range-check limbs need 238 wires
        with_gate(kinds.index(1), nconst=99),                              # more gate constants than constant columns
        with_gate(0, gs=5),                                                # gate 0 outside its own selector group
        with_gate(kinds.index(3), kind=12),                                # kind outside the registry
    ]
    for b in bad:
        with pytest.raises(pkg.P2GpuError) as ei:
            pkg.VerifierCircuitData(b)
        assert ei.value.code == -1
    pkg.VerifierCircuitData(vk).verify(proof)  # the untouched blob still works


def test_mutation_fuzz_of_the_host_parsers(built, corpus):
    d, _, _ = corpus
    exe = os.path.join(ROOT, "acvm-backend-plonky2_amd", "p2gpu-fuzz-host")
    r = subprocess.run([exe, str(d / "vk.blob"), str(d / "proof.bin"), str(d / "comp.bin"), "4000", "20260930"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "ACCEPTED MUTANTS 0" in r.stdout


def test_host_parsers_under_asan_ubsan(built, corpus):
    """`make asan`: hostcore.hip + verify.hip + proofio.hip compiled host-only (no device code) with
    -fsanitize=address,undefined, the same fuzzer linked against that build."""
    d, _, _ = corpus
    subprocess.check_call(["make", "-s", "-C", CSRC, "asan"])
    exe = os.path.join(CSRC, "build_asan", "fuzz_host_asan")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe, str(d / "vk.blob"), str(d / "proof.bin"), str(d / "comp.bin"), "700", "7"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "ACCEPTED MUTANTS 0" in r.stdout and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr


def test_oracle_under_asan_ubsan(pkg, tmp_path):
    """The oracle's prover + verifier (and 200 mutated proofs) under the sanitizers (`make -C oracle asan`)."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"])
    blob, wires = pkg.make_circuit(6, "ecdsa", 9)
    (tmp_path / "c.blob").write_bytes(blob.tobytes())
    (tmp_path / "w.bin").write_bytes(wires.tobytes())
    env = dict(os.environ, OMP_NUM_THREADS="2", ASAN_OPTIONS="detect_leaks=1")
    r = subprocess.run([os.path.join(ROOT, "oracle", "asan_check"), str(tmp_path / "c.blob"), str(tmp_path / "w.bin"), "200", "3"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout + r.stderr[-4000:]
    assert "ACCEPTED 0" in r.stdout and "runtime error" not in r.stderr
