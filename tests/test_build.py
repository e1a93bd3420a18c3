"""N2: the GPU-free part of `builder.build::<C>()` -- selector columns and groups, sigma
polynomials from the copy constraints, k_is, FRI arities -- as the product's p2gpu_build_blob (host code,
csrc/hostcore.hip) and the oracle's orc_build_blob (oracle/build.c).  Reference call sites:
plonky2-backend/src/circuit_translation/mod.rs:80-82, actions/write_vk_action.rs:76.

Pinned to the reference: the two circuits recovered from the proofs the reference ships
(tests/golden/reference_proofs.py) hold plonky2's own selector columns (one group for basic_if, two for
basic_div with its degree-7 PoseidonGate) and sigma polynomials; fed back as gate rows + copy constraints,
both implementations must return those columns bit for bit."""
import ctypes
import sys

import numpy as np
import pytest

from conftest import GOLDEN, P

sys.path.insert(0, GOLDEN)
import reference_proofs as rp  # noqa: E402


class _Params(ctypes.Structure):
    _fields_ = [(k, ctypes.c_uint32) for k in ("degree_bits", "num_wires", "num_routed_wires", "num_challenges", "quotient_degree_factor",
                                                "rate_bits", "cap_height", "proof_of_work_bits", "num_query_rounds", "num_public_inputs")]


class _Gate(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_uint32), ("p", ctypes.c_uint32 * 4), ("degree", ctypes.c_uint32), ("num_constants", ctypes.c_uint32)]


def decompose(blob):
    """A circuit blob -> the inputs of build(): params, gate declarations, row -> gate, gate constants, copy pairs."""
    h = blob[:256].view(np.uint32)
    d, W, R, NC, nsel, ng = int(h[2]), int(h[3]), int(h[4]), int(h[5]), int(h[6]), int(h[23])
    n = 1 << d
    params = _Params(d, W, R, int(h[7]), int(h[8]), int(h[9]), int(h[10]), int(h[11]), int(h[12]), int(h[24]))
    gt = blob[256:256 + 48 * ng].view(np.uint32).reshape(ng, 12)
    gates = (_Gate * ng)()
    for i in range(ng):
        gates[i].kind = int(gt[i, 0])
        for j in range(4):
            gates[i].p[j] = int(gt[i, 1 + j])
        gates[i].degree, gates[i].num_constants = int(gt[i, 9]), int(gt[i, 10])
    off = 256 + 48 * ng
    k_is = blob[off:off + 8 * R].view(np.uint64)
    off += 8 * R
    consts = blob[off:off + 8 * NC * n].view(np.uint64).reshape(NC, n)
    off += 8 * NC * n
    sig = blob[off:off + 8 * R * n].view(np.uint64).reshape(R, n)
    sel = consts[:nsel]
    row_gate = np.zeros(n, dtype=np.uint32)
    for r in range(n):
        vals = [int(v) for v in sel[:, r] if nsel == 1 or int(v) != 0xFFFFFFFF]
        assert len(vals) == 1
        row_gate[r] = vals[0]
    gconst = np.ascontiguousarray(consts[nsel:])
    # copy pairs = the edges x -> sigma(x) of every cycle
    g = rp.root_of_unity(d)
    pos, x = {}, 1
    sub = []
    for i in range(n):
        sub.append(x)
        x = x * g % P
    for c in range(R):
        kc = int(k_is[c])
        for r in range(n):
            pos[kc * sub[r] % P] = (r, c)
    pairs = []
    for c in range(R):
        for r in range(n):
            r2, c2 = pos[int(sig[c, r])]
            if (r2, c2) != (r, c):
                pairs.append((r, c, r2, c2))
    copies = np.array(pairs, dtype=np.uint32).reshape(-1, 4)
    return params, gates, ng, row_gate, gconst, copies


def build_with(fn, params, gates, ng, row_gate, gconst, copies):
    ln = ctypes.c_size_t(0)
    args = [ctypes.byref(params), gates, ng, row_gate.ctypes.data_as(ctypes.c_void_p),
            gconst.ctypes.data_as(ctypes.c_void_p) if gconst.size else None,
            copies.ctypes.data_as(ctypes.c_void_p) if copies.size else None, ctypes.c_size_t(len(copies))]
    assert fn(*args, None, ctypes.byref(ln)) == 0
    out = np.zeros(ln.value, dtype=np.uint8)
    assert fn(*args, out.ctypes.data_as(ctypes.c_void_p), ctypes.byref(ln)) == 0
    return out[:ln.value]


def _fns(pkg, orc):
    a, b = pkg.load_library().p2gpu_build_blob, orc.lib().orc_build_blob
    for f in (a, b):
        f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                      ctypes.c_size_t, ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t)]
    return a, b


@pytest.mark.parametrize("name", ["basic_if", "basic_div"])
def test_build_reproduces_the_reference_circuits(pkg, orc, name):
    """plonky2's own build() output, recovered from the reference's proof files: selector columns (basic_div:
    two groups, UNUSED = 2^32 - 1 outside a gate's group), sigma cycles in WirePartition order, k_is."""
    case = rp.ReferenceCase(name)
    blob = case.blob()
    inputs = decompose(blob)
    for fn in _fns(pkg, orc):
        assert build_with(fn, *inputs).tobytes() == blob.tobytes()
    assert int(blob[:256].view(np.uint32)[6]) == (2 if name == "basic_div" else 1)


@pytest.mark.parametrize("d,mix,npi,nw", [(5, "arith", 0, 234), (7, "ecdsa", 0, 234), (8, "ecdsa", 9, 234), (9, "sha", 4, 135), (11, "ecdsa", 0, 234)])
def test_build_matches_the_workload_generator_and_the_oracle(pkg, orc, d, mix, npi, nw):
    """Three implementations: the synthetic workload generator (csrc/synth.cpp), the product and the oracle."""
    blob = pkg.make_circuit(d, mix, 77, num_public_inputs=npi, num_wires=nw)[0]
    inputs = decompose(blob)
    a, b = (build_with(fn, *inputs) for fn in _fns(pkg, orc))
    assert a.tobytes() == blob.tobytes() and b.tobytes() == blob.tobytes()


def test_build_rejects_bad_input(pkg, orc):
    blob = pkg.make_circuit(5, "ecdsa", 1)[0]
    params, gates, ng, row_gate, gconst, copies = decompose(blob)
    fn = _fns(pkg, orc)[0]
    ln = ctypes.c_size_t(1 << 24)
    out = np.zeros(1 << 24, dtype=np.uint8)

    def call(rg=row_gate, cp=copies, gc=gconst, g=gates):
        return fn(ctypes.byref(params), g, ng, rg.ctypes.data_as(ctypes.c_void_p), gc.ctypes.data_as(ctypes.c_void_p),
                  cp.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(len(cp)), out.ctypes.data_as(ctypes.c_void_p), ctypes.byref(ln))

    bad_rows = row_gate.copy()
    bad_rows[3] = 99
    assert call(rg=bad_rows) != 0
    bad_copies = copies.copy()
    bad_copies[0, 1] = 200            # a column beyond the routed wires
    ln.value = 1 << 24
    assert call(cp=bad_copies) != 0
    bad_c = gconst.copy()
    bad_c[0, 0] = P                   # non-canonical constant
    ln.value = 1 << 24
    assert call(gc=bad_c) != 0
    ln.value = 1 << 24
    assert call() == 0


def test_python_mirror_build_blob(pkg):
    """pkg.build_blob (the `builder.build()` mirror of the host-side package) on a hand-written circuit."""
    blob = pkg.make_circuit(6, "sha", 5)[0]
    params, gates, ng, row_gate, gconst, copies = decompose(blob)
    decl = [(g.kind, tuple(g.p), g.degree, g.num_constants) for g in gates]
    out = pkg.build_blob(6, decl, row_gate, gconst, copies)
    assert out.tobytes() == blob.tobytes()
