"""ctypes loader for oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() import
this module; the product package never does.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
UINT64_MAX = (1 << 64) - 1


class Trace(ctypes.Structure):
    _fields_ = [
        ("pi_hash", ctypes.c_uint64 * 4),
        ("betas", ctypes.c_uint64 * 4),
        ("gammas", ctypes.c_uint64 * 4),
        ("alphas", ctypes.c_uint64 * 4),
        ("zeta", ctypes.c_uint64 * 2),
        ("alpha_fri", ctypes.c_uint64 * 2),
        ("fri_betas", (ctypes.c_uint64 * 2) * 8),
        ("pow_witness", ctypes.c_uint64),
        ("query_indices", ctypes.c_uint32 * 64),
        ("t_wires", ctypes.c_double),
        ("t_zs", ctypes.c_double),
        ("t_quotient", ctypes.c_double),
        ("t_openings", ctypes.c_double),
        ("t_fri", ctypes.c_double),
        ("t_total", ctypes.c_double),
    ]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = ctypes.CDLL(path)
        vp, sz = ctypes.c_void_p, ctypes.c_size_t
        L.orc_circuit_create.argtypes = [vp, sz, ctypes.POINTER(vp)]
        L.orc_circuit_create_verifier.argtypes = [vp, sz, vp, vp, ctypes.POINTER(vp)]
        L.orc_circuit_destroy.argtypes = [vp]
        L.orc_circuit_destroy.restype = None
        L.orc_circuit_cap.argtypes = [vp, vp]
        L.orc_circuit_cap.restype = None
        L.orc_circuit_digest.argtypes = [vp, vp]
        L.orc_circuit_digest.restype = None
        L.orc_prove.argtypes = [vp, vp, vp, ctypes.c_uint32, ctypes.c_uint64, vp, ctypes.POINTER(sz), ctypes.POINTER(Trace)]
        L.orc_verify.argtypes = [vp, vp, sz, ctypes.POINTER(Trace)]
        L.orc_fill_witness.argtypes = [vp, vp]
        L.orc_ntt.argtypes = [vp, ctypes.c_uint, ctypes.c_int]
        L.orc_ntt.restype = None
        L.orc_coset_lde.argtypes = [vp, ctypes.c_uint, ctypes.c_uint, vp]
        L.orc_coset_lde.restype = None
        L.orc_keccak256.argtypes = [vp, sz, vp]
        L.orc_keccak256.restype = None
        L.orc_keccak_permutation.argtypes = [vp]
        L.orc_keccak_permutation.restype = None
        L.orc_poseidon_permute.argtypes = [vp]
        L.orc_poseidon_permute.restype = None
        L.orc_poseidon_round_constants.argtypes = [vp]
        L.orc_poseidon_round_constants.restype = None
        L.orc_poseidon_hash_no_pad.argtypes = [vp, sz, vp]
        L.orc_poseidon_hash_no_pad.restype = None
        L.orc_commit_values.argtypes = [vp, sz, ctypes.c_uint, ctypes.c_uint, ctypes.c_uint, vp]
        L.orc_commit_values.restype = None
        L.orc_merkle_cap.argtypes = [vp, sz, sz, ctypes.c_uint, vp]
        L.orc_merkle_cap.restype = None
        L.orc_gate_eval.argtypes = [ctypes.c_uint32, vp, vp, vp, vp, vp]
        L.orc_gate_eval_ext.argtypes = [ctypes.c_uint32, vp, vp, vp, vp, vp]
        for f in ("orc_gl_mul", "orc_gl_pow"):
            getattr(L, f).argtypes = [ctypes.c_uint64, ctypes.c_uint64]
            getattr(L, f).restype = ctypes.c_uint64
        L.orc_gl_inv.argtypes = [ctypes.c_uint64]
        L.orc_gl_inv.restype = ctypes.c_uint64
        L.orc_challenger_squeeze.argtypes = [vp, sz, vp, sz]
        L.orc_challenger_squeeze.restype = None
        _LIB = L
    return _LIB


def _u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


class OracleCircuit:
    def __init__(self, blob, verifier_cap=None, verifier_digest=None):
        """Full (prover + verifier) handle, or verifier-only when the constants_sigmas cap and the
        circuit digest are supplied (plonky2 VerifierCircuitData: no commitment is recomputed)."""
        self._blob = np.ascontiguousarray(blob, dtype=np.uint8)
        self._h = ctypes.c_void_p()
        if verifier_cap is not None:
            cap = np.frombuffer(verifier_cap, dtype=np.uint8)
            dg = np.frombuffer(verifier_digest, dtype=np.uint8)
            rc = lib().orc_circuit_create_verifier(self._blob.ctypes.data, self._blob.nbytes, cap.ctypes.data,
                                                   dg.ctypes.data, ctypes.byref(self._h))
        else:
            rc = lib().orc_circuit_create(self._blob.ctypes.data, self._blob.nbytes, ctypes.byref(self._h))
        if rc != 0:
            raise ValueError(f"orc_circuit_create failed: {rc}")
        hdr = self._blob[:256].view(np.uint32)
        self.cap_height = int(hdr[10])
        self.hash_bytes = 32 if int(hdr[22]) == 1 else 25   # PoseidonHash digests are 4 field elements

    def cap(self):
        out = np.zeros(self.hash_bytes << self.cap_height, dtype=np.uint8)
        lib().orc_circuit_cap(self._h, out.ctypes.data)
        return out.tobytes()

    def digest(self):
        out = np.zeros(self.hash_bytes, dtype=np.uint8)
        lib().orc_circuit_digest(self._h, out.ctypes.data)
        return out.tobytes()

    def prove(self, wires, public_inputs=(), pow_hint=UINT64_MAX):
        w = _u64(wires)
        pis = _u64(np.array(list(public_inputs), dtype=np.uint64))
        out = np.zeros(1 << 21, dtype=np.uint8)
        plen = ctypes.c_size_t(out.nbytes)
        tr = Trace()
        rc = lib().orc_prove(self._h, w.ctypes.data, pis.ctypes.data, len(pis), ctypes.c_uint64(pow_hint),
                             out.ctypes.data, ctypes.byref(plen), ctypes.byref(tr))
        if rc != 0:
            raise RuntimeError(f"orc_prove failed: {rc}")
        return out[:plen.value].tobytes(), tr

    def fill_witness(self, wires):
        w = _u64(wires).copy()
        rc = lib().orc_fill_witness(self._h, w.ctypes.data)
        if rc != 0:
            raise RuntimeError(f"orc_fill_witness failed: {rc}")
        return w

    def verify(self, proof):
        p = np.frombuffer(proof, dtype=np.uint8)
        tr = Trace()
        rc = lib().orc_verify(self._h, p.ctypes.data, p.nbytes, ctypes.byref(tr))
        return rc == 0

    def close(self):
        if self._h and self._h.value:
            lib().orc_circuit_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def ntt(a, inverse=False):
    a = _u64(a).copy()
    lib().orc_ntt(a.ctypes.data, a.size.bit_length() - 1, 1 if inverse else 0)
    return a


def coset_lde(coeffs, rate_bits=3):
    c = _u64(coeffs)
    d = c.size.bit_length() - 1
    out = np.zeros(c.size << rate_bits, dtype=np.uint64)
    lib().orc_coset_lde(c.ctypes.data, d, rate_bits, out.ctypes.data)
    return out


def keccak256(data):
    b = np.frombuffer(bytes(data), dtype=np.uint8) if len(data) else np.zeros(0, dtype=np.uint8)
    out = np.zeros(32, dtype=np.uint8)
    lib().orc_keccak256(b.ctypes.data if b.size else None, b.size, out.ctypes.data)
    return out.tobytes()


def commit_values(vals, rate_bits=3, cap_height=4):
    v = _u64(vals)
    ncols, n = v.shape
    out = np.zeros(25 << cap_height, dtype=np.uint8)
    lib().orc_commit_values(v.ctypes.data, ncols, n.bit_length() - 1, rate_bits, cap_height, out.ctypes.data)
    return out.tobytes()


def hash_rows(rows):
    """hash_or_noop of each row via a one-leaf-per-cap-entry Merkle cap."""
    r = _u64(rows)
    out = []
    for row in r:
        buf = np.zeros(25, dtype=np.uint8)
        lib().orc_merkle_cap(np.ascontiguousarray(row).ctypes.data, 1, row.size, 0, buf.ctypes.data)
        out.append(buf)
    return np.stack(out)


def gate_eval(kind, params, wires, consts=(), pi_hash=(0, 0, 0, 0), ext=False):
    p = np.array(list(params) + [0] * (4 - len(params)), dtype=np.uint32)
    w = _u64(wires)
    c = _u64(np.array(list(consts) + [0] * 8, dtype=np.uint64))
    pih = _u64(np.array(pi_hash, dtype=np.uint64))
    out = np.zeros(4096, dtype=np.uint64)
    fn = lib().orc_gate_eval_ext if ext else lib().orc_gate_eval
    k = fn(kind, p.ctypes.data, w.ctypes.data, c.ctypes.data, pih.ctypes.data, out.ctypes.data)
    return out[: k * (2 if ext else 1)].copy()
