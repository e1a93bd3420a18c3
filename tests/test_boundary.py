"""The C-ABI boundary on a CPU-only host: the library loads, exports every symbol that
include/p2gpu.h declares, and fails loudly (no CPU fallback) when no GPU is present."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "p2gpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(p2gpu_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_are_exported(pkg):
    lib = pkg.load_library()
    syms = _declared_symbols()
    assert {"p2gpu_init", "p2gpu_circuit_create", "p2gpu_prove", "p2gpu_prove_dev", "p2gpu_circuit_destroy",
            "p2gpu_last_error", "p2gpu_ifft_batch", "p2gpu_lde_batch", "p2gpu_commit_values"} <= set(syms)
    for s in syms:
        assert hasattr(lib, s), s


def test_library_is_in_tree_and_gfx950(pkg):
    path = pkg.lib_path()
    assert path.startswith(ROOT) and os.path.exists(path)
    data = open(path, "rb").read()
    assert b"gfx950" in data  # the code object is built for MI355X only
    # the product never links, loads or names the checker: no oracle symbol / file name in the binary
    for needle in (b"liboracle", b"pyoracle", b"orc_prove", b"orc_verify", b"orc_circuit"):
        assert needle not in data, needle
    import subprocess
    needed = subprocess.run(["readelf", "-d", path], capture_output=True, text=True).stdout
    assert "oracle" not in needed.lower()


def test_product_sources_do_not_reference_the_oracle():
    src = os.path.join(ROOT, "acvm-backend-plonky2_amd")
    for dirpath, _, files in os.walk(src):
        if "build" in dirpath.split(os.sep):
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "liboracle" not in text and "pyoracle" not in text and "orc_" not in text, f


def test_no_cpu_fallback_without_gpu(pkg):
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    blob, wires = pkg.make_circuit(5, "arith", 1)
    with pytest.raises(pkg.P2GpuError) as ei:
        pkg.CircuitData(blob)
    assert ei.value.code == -3 and "no CPU fallback" in str(ei.value)
    with pytest.raises(pkg.P2GpuError):
        pkg.ifft_batch(np.zeros((1, 8), dtype=np.uint64))


def test_blob_header_layout(pkg):
    """include/p2gpu.h blob: wide_ecc_config shape (circuit_translation/mod.rs:69)."""
    blob, wires = pkg.make_circuit(6, "sha", 3)
    h = blob[:256].view(np.uint32)
    assert h[0] == 0x43473250 and h[1] == 1 and h[2] == 6
    assert (h[3], h[4], h[7], h[8], h[9], h[10], h[11], h[12]) == (234, 80, 2, 8, 3, 4, 16, 28)
    assert h[26] == 9 and wires.shape == (234, 64) and wires.dtype == np.uint64
    ng, nc = int(h[23]), int(h[5])
    assert blob.nbytes == 256 + 48 * ng + 8 * (80 + (nc + 80) * 64)


def test_numbers_table_is_generated_from_the_committed_bench_lines():
    """profiles/NUMBERS.md is the one place current numbers live (DESIGN.md quotes none that it does not need): it must be what
    profiles/numbers.py makes of the newest committed bench lines, not a hand-edited copy."""
    import subprocess
    import sys

    path = os.path.join(ROOT, "profiles", "NUMBERS.md")
    before = open(path).read()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "profiles", "numbers.py")], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-500:]
    after = open(path).read()
    if after != before:
        open(path, "w").write(before)
    assert after == before
