"""Synthetic circuits with the reference's shape (workload generator, host only).

See csrc/synth.cpp: wide_ecc_config (plonky2-backend/src/circuit_translation/mod.rs:69)
with gate-mix presets `arith`, `sha`, `ecdsa`; satisfying wire values from a seed.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def synth_lib_path():
    return os.path.join(_HERE, "libp2synth.so")


def _lib():
    global _LIB
    if _LIB is None:
        path = synth_lib_path()
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run __graft_entry__.build()")
        _LIB = ctypes.CDLL(path)
        _LIB.p2synth_free.argtypes = [ctypes.c_void_p]
        _LIB.p2synth_free.restype = None
    return _LIB


def make_circuit(degree_bits, mix="arith", seed=1, num_public_inputs=0, num_wires=234, hasher=0, pi_row_routed_only=False):
    """Returns (blob: np.uint8[...], wires: np.uint64[num_wires][2^degree_bits]) and, when
    num_public_inputs > 0, additionally the public input values (np.uint64[num_public_inputs]).
    num_wires: 234 (wide_ecc_config, the translator's shape) or 135 (standard_recursion_config).
    hasher: 0 = KeccakHash<25> (the reference's KeccakGoldilocksConfig), 1 = PoseidonHash
    (PoseidonGoldilocksConfig: Poseidon Merkle trees, challenger and circuit digest) -- header word 22.
    pi_row_routed_only: the unused NON-routed wires of the PublicInputGate row stay zero instead of holding the
    random values plonky2's build() gives them (then the routed columns determine the whole matrix)."""
    lib = _lib()
    blob = ctypes.POINTER(ctypes.c_uint8)()
    blen = ctypes.c_size_t()
    wires = ctypes.POINTER(ctypes.c_uint64)()
    nw = ctypes.c_uint32()
    pis = np.zeros(max(num_public_inputs, 1), dtype=np.uint64)
    rc = lib.p2synth_make2(ctypes.c_uint(degree_bits), mix.encode(), ctypes.c_uint64(seed),
                           ctypes.c_uint32(num_public_inputs), ctypes.c_uint32(num_wires), ctypes.c_uint32(1 if pi_row_routed_only else 0),
                           ctypes.byref(blob), ctypes.byref(blen),
                           ctypes.byref(wires), ctypes.byref(nw), ctypes.c_void_p(pis.ctypes.data))
    if rc != 0:
        raise ValueError(f"p2synth_make({degree_bits}, {mix!r}) failed: {rc}")
    try:
        b = np.ctypeslib.as_array(blob, (blen.value,)).copy()
        b[:256].view(np.uint32)[22] = hasher
        w = np.ctypeslib.as_array(wires, (nw.value, 1 << degree_bits)).copy()
    finally:
        lib.p2synth_free(blob)
        lib.p2synth_free(wires)
    if num_public_inputs:
        return b, w, pis[:num_public_inputs].copy()
    return b, w


def poseidon_gate_row(inputs):
    """The 135 wires of one PoseidonGate row for twelve inputs, swap = 0 (csrc/synth.cpp p2synth_poseidon_gate_row)."""
    lib = _lib()
    x = np.ascontiguousarray(np.array([int(v) for v in inputs], dtype=np.uint64))
    assert x.size == 12
    out = np.zeros(135, dtype=np.uint64)
    lib.p2synth_poseidon_gate_row.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lib.p2synth_poseidon_gate_row.restype = None
    lib.p2synth_poseidon_gate_row(x.ctypes.data, out.ctypes.data)
    return out
