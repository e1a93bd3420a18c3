"""Parity of the HIP path (through the C ABI) with the CPU oracle -- bit-exact, integer work.

Small/medium sizes: stage-by-stage and whole-proof byte equality on the same seeded inputs.
Full BASELINE size (2^17 gates = 2^20 LDE rows): size-independent properties -- the oracle's
verifier accepts the GPU proof, the transcript is reproducible, LDE/iNTT round-trip identities.
"""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, P

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu(pkg):
    import torch

    assert torch.cuda.is_available(), "these tests need the MI355X"
    info = pkg.device_info()
    assert "gfx950" in info["name"]
    return info


def _rand(shape, seed):
    return np.random.default_rng(seed).integers(0, P, size=shape, dtype=np.uint64)


def test_field_primitives_match_portable_code(pkg, gpu):
    """The device forms of canon / add / sub / reduce128 / mul / mul_add / x*2^e (e = 1..95) / Acc160 are
    carry-chain inline assembly; the host, the verifier and the oracle run the portable code.  Every wrap
    case: all pairs of an edge set around 0, 2^32, 2^63, p, 2^64 plus random words (any u64, not only
    canonical ones: canon and reduce128 take arbitrary words)."""
    eps = (1 << 32) - 1
    pts = [0, 1, 2, eps - 1, eps, eps + 1, eps + 2, 1 << 33, (1 << 63) - 1, 1 << 63, (1 << 63) + 1,
           P - eps - 1, P - eps, P - eps + 1, P - 2, P - 1, P, P + 1, P + eps - 2, P + eps - 1,
           (1 << 64) - 2, (1 << 64) - 1, 0xFFFFFFFE00000000, 0xFFFFFFFEFFFFFFFF, 0x00000000FFFFFFFF,
           0x0000000100000000, 0xFFFFFFFF, 0x8000000080000000, 0x7FFFFFFF7FFFFFFF]
    for s in (1, 7, 12, 24, 31, 33, 36, 48, 60, 63):
        pts += [(1 << s) - 1, 1 << s, (P - (1 << s)) % (1 << 64), ((1 << 64) - (1 << s))]
    edge = np.array(sorted(set(pts)), dtype=np.uint64)
    a = np.repeat(edge, edge.size)
    b = np.tile(edge, edge.size)
    rng = np.random.default_rng(5)
    ra = rng.integers(0, 1 << 64, size=1 << 20, dtype=np.uint64)
    rb = rng.integers(0, 1 << 64, size=1 << 20, dtype=np.uint64)
    # random words with a sparse high or low half hit the carry corner cases far more often
    ra[::3] &= np.uint64(0xFFFFFFFF00000000)
    rb[1::3] |= np.uint64(0xFFFFFFFF00000000)
    ra[2::5] |= np.uint64(0x00000000FFFFFFFF)
    a = np.concatenate([a, ra, np.repeat(edge, 64)])
    b = np.concatenate([b, rb, rng.integers(0, 1 << 64, size=64 * edge.size, dtype=np.uint64)])
    bad = pkg.field_selftest(a, b)
    # slots 8-13: the congruent-word forms of round 3 (gl_mul_nc, gl_mul_add_nc, gl_reduce128_nc, gl_add / gl_sub with a first
    # operand in [p, 2^64), chains of them) -- the edge set and the random words above hand them operands >= p directly,
    # which a proof reaches with probability 2^-32 per operation
    assert not bad.any(), dict(zip("canon add sub reduce128 mul mul_add mul_pow2 acc160 mul_nc mul_add_nc reduce128_nc add_nc sub_nc chain_nc - -".split(), bad.tolist()))
    # the round-1 entry point keeps its eight-word contract (a caller built against the old header passes uint64_t[8])
    lib = pkg.load_library()
    out = np.full(16, 0xA5A5A5A5A5A5A5A5, dtype=np.uint64)
    assert lib.p2gpu_field_selftest(a.ctypes.data, b.ctypes.data, 4096, out.ctypes.data) == 0
    assert not out[:8].any() and (out[8:] == 0xA5A5A5A5A5A5A5A5).all()


@pytest.mark.parametrize("d", [0, 1, 2, 5, 8, 11, 12, 13, 16])
def test_ifft_and_lde_match_oracle(pkg, orc, gpu, d):
    # 12 = one LDS pass, 13 = first two-pass size, ragged column counts
    cols = 3 if d < 16 else 2
    v = _rand((cols, 1 << d), d)
    v[0, 0] = P - 1  # maximum canonical value
    if d:
        v[1, :] = 0   # all-zero column
    coeffs = pkg.ifft_batch(v)
    exp = np.stack([orc.ntt(r, inverse=True) for r in v])
    assert np.array_equal(coeffs, exp)
    lde = pkg.lde_batch(exp, 3)
    assert np.array_equal(lde, np.stack([orc.coset_lde(r, 3) for r in exp]))


_FUSED_STAGE = r"""
import sys, numpy as np
sys.path.insert(0, {root!r})
import __graft_entry__ as ge
pkg, orc = ge.load_package(), ge.load_oracle()
P = 0xFFFFFFFF00000001
for d, cols in ((13, 8), (14, 11), (16, 9), (17, 19), (19, 9)):
    rng = np.random.default_rng(1000 + d)
    v = rng.integers(0, P, size=(cols, 1 << d), dtype=np.uint64)
    v[0, 0] = P - 1
    v[cols - 1, :] = 0
    coeffs = pkg.ifft_batch(v)
    exp = np.stack([orc.ntt(r, inverse=True) for r in v])
    assert np.array_equal(coeffs, exp), ("ifft", d)
    rate = 3 if d <= 17 else 1
    lde = pkg.lde_batch(exp, rate)
    assert np.array_equal(lde, np.stack([orc.coset_lde(r, rate) for r in exp])), ("lde", d)
print("fused-stage-ok")
"""


def test_two_pass_transforms_ragged_column_counts(gpu):
    """>= 8 columns of the two-pass sizes (13 <= d): ragged column counts (not a multiple of the 8 XCD slots the 1-D grid is dealt
    over), every word of iNTT and LDE against the oracle.  Own process: a fresh library, fresh plans."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _FUSED_STAGE.format(root=root)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "fused-stage-ok" in r.stdout, r.stderr[-2000:]


@pytest.mark.parametrize("d", [12, 15, 17, 18, 19, 20])
def test_direct_dit_passes_match_oracle(pkg, orc, gpu, d):
    """The direct DIT passes of round 5 (ntt.hip ntt_dit_head_kernel / ntt_dit_strided_kernel: first round from global memory, last
    round to it with shift twiddles, folded table in between), one size per shape of the strided pass -- 12: head only; 15: [3];
    17: [3,2]; 18: [3,3]; 19: [3,2,2]; 20: [3,3,2] (21 = [3,3,3] and 22 = two strided passes: test_deep_transforms) -- every word of
    the coset LDE against the oracle, three columns (one of edge words, one zero)."""
    v = _rand((3, 1 << d), 100 + d)
    v[0, :8] = [P - 1, 0, 1, P - 2, 1 << 32, (1 << 32) - 1, P - (1 << 32), 2]
    v[2, :] = 0
    rate = 3 if d <= 17 else 1
    lde = pkg.lde_batch(v, rate)
    assert np.array_equal(lde, np.stack([orc.coset_lde(r, rate) for r in v]))
    # ... and their mirror images for values -> coefficients (ntt_dif_strided_kernel, ntt_dif_tail2_kernel)
    assert np.array_equal(pkg.ifft_batch(v), np.stack([orc.ntt(r, inverse=True) for r in v]))


def test_ab_switches_keep_every_byte(gpu):
    """The A/B switches of round 5, each in a process of its own (they are read once): P2GPU_NTT_DIRECT=0 sends every pass through
    ntt_pass_kernel (the round 1-4 path), P2GPU_NTT_HEAD=1 takes the head pass with its first two rounds in LDS (ntt_dit_head_kernel),
    P2GPU_LEAF_LEVELS=0 leaves every tree level to merkle_level / merkle_tail instead of
    building the first two inside the leaf-hash launch.  Same LDE words, same proof bytes (virtual-column wires tree, plain
    Z / quotient trees, a 231-dense-column witness) as the default build."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import sys, hashlib, numpy as np
sys.path.insert(0, {root!r})
import __graft_entry__ as ge
pkg = ge.load_package()
rng = np.random.default_rng(77)
P = 0xFFFFFFFF00000001
for d in (12, 15, 17, 19, 21):
    v = rng.integers(0, P, size=(2, 1 << d), dtype=np.uint64)
    print("LDE", d, hashlib.sha256(pkg.lde_batch(v, 3 if d <= 17 else 1).tobytes()).hexdigest())
    print("LDE-ifft", d, hashlib.sha256(pkg.ifft_batch(v).tobytes()).hexdigest())
for d, mix in ((8, "sha"), (11, "sha"), (14, "ecdsa")):
    blob, wires = pkg.make_circuit(d, mix, 5)
    cd = pkg.CircuitData(blob)
    print("PROOF", d, mix, hashlib.sha256(cd.prove(wires).to_bytes()).hexdigest())
    cd.set("virtual_columns", 0)
    print("PROOF-novirt", d, mix, hashlib.sha256(cd.prove(wires).to_bytes()).hexdigest())
""".format(root=root)
    outs = []
    for env in ({}, {"P2GPU_NTT_DIRECT": "0"}, {"P2GPU_LEAF_LEVELS": "0"}, {"P2GPU_NTT_HEAD": "1"}):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append([ln for ln in r.stdout.splitlines() if ln.startswith(("LDE", "PROOF"))])
    assert len(outs[0]) == 16 and outs[0] == outs[1] == outs[2] == outs[3]


def test_measurement_switches_keep_every_byte(pkg, orc, gpu):
    """Every other environment switch of INTEGRATION.md section 7 (the host-witness pipeline's, the gate-group balance, the staged /
    unstaged gate sums, the round-4 column groups), each in a process of its own: the proofs of a `sha` circuit through the
    host-matrix entry and of an `ecdsa` circuit with the half-domain route forced are the ORACLE's bytes under every one of them."""
    import hashlib
    import os
    import subprocess
    import sys

    want = []
    for d, mix in ((11, "sha"), (12, "ecdsa")):
        blob, wires = pkg.make_circuit(d, mix, 9)
        want.append("PROOF %d %s %s" % (d, mix, hashlib.sha256(orc.OracleCircuit(blob).prove(wires)[0]).hexdigest()))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import sys, hashlib, numpy as np, torch
sys.path.insert(0, {root!r})
import __graft_entry__ as ge
pkg = ge.load_package()
for d, mix in ((11, "sha"), (12, "ecdsa")):
    blob, wires = pkg.make_circuit(d, mix, 9)
    cd = pkg.CircuitData(blob)
    cd.set("half_gates", 2)
    host = cd.prove(wires).to_bytes()                                                  # p2gpu_prove: chunked upload, host scan
    assert cd.prove(torch.from_numpy(wires.view(np.int64)).cuda()).to_bytes() == host  # p2gpu_prove_dev
    print("PROOF", d, mix, hashlib.sha256(host).hexdigest())
""".format(root=root)
    envs = ({}, {"P2GPU_CHUNK_BLOCKS": "1"}, {"P2GPU_CHUNK_BLOCKS": "5"}, {"P2GPU_HOST_PRESCAN": "0"}, {"P2GPU_HALF_GATES": "0"},
            {"P2GPU_SUMS_STAGE": "0"}, {"P2GPU_GATE_GROUPS": "4", "P2GPU_GATE_GROUPS_HALF": "4", "P2GPU_SUMS_GROUPS": "4"},
            {"P2GPU_GATE_GROUPS": "1", "P2GPU_GATE_GROUPS_HALF": "1", "P2GPU_SUMS_GROUPS": "1"}, {"P2GPU_PERM_COST": "100000"},
            {"P2GPU_PERM_COST": "1", "P2GPU_GATE_GROUPS": "4", "P2GPU_GATE_GROUPS_HALF": "4"}, {"P2GPU_NTT_GROUP_MB": "1"})
    for env in envs:
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (env, r.stderr[-2000:])
        assert [ln for ln in r.stdout.splitlines() if ln.startswith("PROOF")] == want, env


@pytest.mark.parametrize("d", [21, 22])
def test_deep_transforms(pkg, orc, gpu, d):
    """2^21 points: 12 + 9 layers (one strided pass with 64-byte runs); 2^22: 12 + 5 + 5 (two strided
    passes) -- the plan's three-pass shape."""
    v = _rand((1, 1 << d), d)
    coeffs = pkg.ifft_batch(v)
    assert np.array_equal(coeffs[0], orc.ntt(v[0], inverse=True))
    lde = pkg.lde_batch(coeffs[:, : 1 << d], 1)
    assert np.array_equal(lde[0], orc.coset_lde(coeffs[0], 1))


def test_large_transform_roundtrip(pkg, orc, gpu):
    """2^20 points (three LDS passes): iNTT against the oracle on one column and the identity
    LDE(iNTT(v))[8k] relation is checked through the oracle's LDE on the same coefficients."""
    v = _rand((2, 1 << 20), 77)
    coeffs = pkg.ifft_batch(v)
    assert np.array_equal(coeffs[0], orc.ntt(v[0], inverse=True))
    assert np.array_equal(coeffs[1], orc.ntt(v[1], inverse=True))


@pytest.mark.parametrize("ncols", [1, 2, 3, 4, 16, 17, 18, 33, 34, 35, 234])
def test_keccak_rows_match_oracle(pkg, orc, gpu, ncols):
    # <= 3 elements: hash_or_noop copies; 17 = exactly one rate block; 234 = a wires leaf (14 blocks)
    rows = _rand((7, ncols), ncols)
    assert np.array_equal(pkg.hash_rows(rows), orc.hash_rows(rows))


@pytest.mark.parametrize("d,ncols", [(1, 4), (4, 20), (9, 5), (12, 84), (13, 3)])
def test_commit_cap_matches_oracle(pkg, orc, gpu, d, ncols):
    v = _rand((ncols, 1 << d), d * 31 + ncols)
    assert pkg.commit_values(v, 3, 4) == orc.commit_values(v, 3, 4)


@pytest.mark.parametrize("d,mix,seed", [
    (5, "arith", 1), (5, "ecdsa", 2), (6, "sha", 3), (8, "ecdsa", 4), (9, "arith", 5),
    (11, "sha", 6), (12, "ecdsa", 7), (13, "ecdsa", 8), (13, "arith", 9), (14, "sha", 10), (7, "grammar", 12), (12, "grammar", 13),
])
def test_proof_bytes_match_oracle(pkg, orc, gpu, d, mix, seed):
    blob, wires = pkg.make_circuit(d, mix, seed)
    cd = pkg.CircuitData(blob)
    oc = orc.OracleCircuit(blob)
    assert cd.constants_sigmas_cap() == oc.cap()
    assert cd.circuit_digest() == oc.digest()
    expect, tr = oc.prove(wires)
    got = cd.prove(wires)
    assert got.timings["pow_witness"] == tr.pow_witness
    assert got.to_bytes() == expect
    assert oc.verify(got.to_bytes())
    # the library's own verifier, on the prover handle and on the exported verifier key
    cd.verify(got)
    vd = cd.verifier_data()
    vd.verify(got)
    assert vd.circuit_digest() == oc.digest() and vd.constants_sigmas_cap() == oc.cap()
    # device-resident witness (torch tensor) goes through p2gpu_prove_dev: same bytes
    import torch

    wd = torch.from_numpy(wires.view(np.int64)).cuda()
    assert cd.prove(wd).to_bytes() == expect
    # proving twice on one handle is repeatable (no state leaks between proofs)
    assert cd.prove(wires).to_bytes() == expect
    cd.close()


@pytest.mark.parametrize("d,mix,routed_only", [(5, "arith", False), (10, "sha", False), (10, "sha", True), (13, "sha", False),
                                               (13, "arith", True), (13, "ecdsa", False)])
def test_structured_wire_columns_do_not_change_the_proof(pkg, orc, gpu, d, mix, routed_only):
    """The wires no gate of a circuit uses (154 of 234 in the sha / arith mixes, as in the reference's circuits
    without ECC gates) hold ONE value: the random one build() puts in the PublicInputGate row
    (randomize_unused_pi_wires) -- or nothing at all when a caller leaves them zero.  Their transforms are not
    computed: zeros are stored / the handle's transform of that row's unit column is scaled.  Same bytes with the
    knob off, same bytes as the oracle, on the resident and the chunked host-witness entry points."""
    import torch

    blob, wires = pkg.make_circuit(d, mix, 77, pi_row_routed_only=routed_only)
    w = wires.reshape(234, -1)
    per_col = (w != 0).sum(axis=1)
    if mix == "ecdsa":
        assert (per_col > 1).sum() > 200
    else:
        assert (per_col > 1).sum() == 80 and (per_col[80:] == (0 if routed_only else 1)).all()
    cd, oc = pkg.CircuitData(blob), orc.OracleCircuit(blob)
    expect, _ = oc.prove(wires)
    wd = torch.from_numpy(wires.view(np.int64)).cuda()
    # "virtual_columns": structured columns that no gate reads get no LDE in memory at all -- the leaf hash and the
    # query gather recompute val * LDE(unit column); toggling it between proofs must not leave stale buffers behind
    for knob, virt in ((1, 1), (1, 0), (0, 1), (1, 1), (1, 0)):
        cd.set("zero_columns", knob)
        cd.set("virtual_columns", virt)
        assert cd.prove(wd).to_bytes() == expect          # p2gpu_prove_dev
        assert cd.prove(wires).to_bytes() == expect       # p2gpu_prove: column chunks, classes per chunk
        if routed_only:
            assert cd.prove_routed(wires[:80]).to_bytes() == expect
    # a witness whose column structure changes from proof to proof on the same handle: stale zeros or stale
    # values of the previous proof must not survive in the coefficient / LDE buffers
    cd.set("self_check", 0)
    pi_row = int(np.nonzero(w[233])[0][0]) if not routed_only and mix != "ecdsa" else 1
    for cols, rows in (([5, 17, 79, 100], None), ([0, 40, 233], None), ([90, 200], 7), ([], None)):
        w2 = wires.copy()
        for c_ in cols:
            if rows is None:
                w2[c_, :] = 0                       # zero column
            else:
                w2[c_, (pi_row + rows) % w.shape[1]] = 5     # a second non-zero row: dense
        want, _ = oc.prove(w2)   # an unsatisfied witness still yields (unverifiable) bytes, on both sides
        cd.set("virtual_columns", 1)
        assert cd.prove(torch.from_numpy(w2.view(np.int64)).cuda()).to_bytes() == want, cols
        assert cd.prove(w2).to_bytes() == want, cols
        cd.set("virtual_columns", 0)
        assert cd.prove(w2).to_bytes() == want, cols
        cd.set("virtual_columns", 1)
        assert cd.prove(torch.from_numpy(w2.view(np.int64)).cuda()).to_bytes() == want, cols
    cd.close()


@pytest.mark.parametrize("d,mix,npi", [(10, "sha", 4), (12, "sha", 20), (11, "arith", 1), (13, "sha", 40), (9, "ecdsa", 9)])
def test_structured_columns_with_public_inputs(pkg, orc, gpu, d, mix, npi):
    """With public inputs the circuit has PoseidonGate rows (the public-input hash, one per 8 inputs): wires 80..134
    are then non-zero in those rows and in the PublicInputGate row only.  Such columns are a linear combination of up
    to four unit columns (class 3: the PublicInputGate row and the first three PoseidonGate rows) and are written by
    the fill kernel instead of being transformed; with more Poseidon rows than that they are plain dense columns.
    Same bytes as the oracle with the shortcuts on and off, on every entry point."""
    import torch

    blob, wires, pis = pkg.make_circuit(d, mix, 31, num_public_inputs=npi)
    w = wires.reshape(234, -1)
    per_col = (w != 0).sum(axis=1)
    if mix != "ecdsa":
        assert (per_col[135:] == 1).all() and (per_col[80:135] >= 2).all() and (per_col[80:135] <= 1 + (npi + 7) // 8).all()
    cd, oc = pkg.CircuitData(blob), orc.OracleCircuit(blob)
    expect, _ = oc.prove(wires, public_inputs=pis)
    wd = torch.from_numpy(wires.view(np.int64)).cuda()
    pi_row = int(np.nonzero(w[233])[0][0]) if mix != "ecdsa" else 0
    for knob, virt in ((1, 1), (0, 1), (1, 0), (1, 1)):
        cd.set("zero_columns", knob)
        cd.set("virtual_columns", virt)
        assert cd.prove(wd, public_inputs=pis).to_bytes() == expect
        assert cd.prove(wires, public_inputs=pis).to_bytes() == expect
        if mix != "ecdsa":
            assert cd.prove_sparse(wires, 135, pi_row, public_inputs=pis).to_bytes() == expect
    # a Poseidon-row value turning up in a column that was class 1, and a class 3 column losing its second row
    cd.set("self_check", 0)
    if mix != "ecdsa":
        rows3 = np.nonzero(w[100])[0]
        w2 = wires.copy().reshape(234, -1)
        w2[200, rows3[-1]] = 7
        w2[90, rows3[rows3 != pi_row][0]] = 0
        w2 = np.ascontiguousarray(w2)
        want, _ = oc.prove(w2, public_inputs=pis)
        assert cd.prove(torch.from_numpy(w2.view(np.int64)).cuda(), public_inputs=pis).to_bytes() == want
        assert cd.prove(w2, public_inputs=pis).to_bytes() == want
    cd.close()


@pytest.mark.parametrize("d,mix,ncols", [(10, "sha", 80), (10, "sha", 97), (13, "arith", 80), (9, "sha", 234), (11, "ecdsa", 231),
                                         (8, "sha", 0), (12, "sha", 17)])
def test_prove_sparse_matches_the_full_matrix(pkg, orc, gpu, d, mix, ncols):
    """p2gpu_prove_sparse: the first `ncols` columns from the host, the others as ONE value each (their value in
    the PublicInputGate row, all plonky2 leaves in the wires no gate uses) and written in HBM.  Same bytes as the
    oracle and as p2gpu_prove on the matrix the compact form stands for -- also when the split is not the
    circuit's (ragged ncols, no tail at all, a tail that cuts into used wires: the library classifies what it is
    given)."""
    blob, wires = pkg.make_circuit(d, mix, 5)
    w = wires.reshape(234, -1)
    nz = np.nonzero(w[233])[0]
    row = int(nz[0]) if nz.size == 1 else 3
    full = w.copy()
    tail = full[ncols:, row].copy()
    full[ncols:, :] = 0
    full[ncols:, row] = tail
    cd, oc = pkg.CircuitData(blob), orc.OracleCircuit(blob)
    if not np.array_equal(full, w):
        cd.set("self_check", 0)        # cutting into used wires leaves an unsatisfied witness: bytes still comparable
    else:
        assert ncols >= 80
    expect, _ = oc.prove(np.ascontiguousarray(full))
    assert cd.prove_sparse(w, ncols, row).to_bytes() == expect
    assert cd.prove_sparse(np.ascontiguousarray(full[:ncols]), ncols, row, tail=tail).to_bytes() == expect
    assert cd.prove(np.ascontiguousarray(full)).to_bytes() == expect
    for bad in (lambda: cd.prove_sparse(w, 235, row), lambda: cd.prove_sparse(w, ncols, w.shape[1])):
        with pytest.raises(pkg.P2GpuError):
            bad()
    cd.close()


@pytest.mark.parametrize("d,mix", [(9, "ecdsa"), (13, "sha")])
def test_page_locked_host_witness(pkg, orc, gpu, d, mix):
    """p2gpu_host_alloc / p2gpu_host_free: a wire matrix built in page-locked memory proves to the same bytes through every
    host entry (full matrix, compact form), the block can be reused for the next witness and released."""
    blob, wires = pkg.make_circuit(d, mix, 11)
    cd, oc = pkg.CircuitData(blob), orc.OracleCircuit(blob)
    expect, _ = oc.prove(wires)
    wp = pkg.host_array(wires.shape)
    assert wp.dtype == np.uint64 and wp.shape == wires.shape and wp.flags.c_contiguous
    wp[...] = wires
    assert cd.prove(wp).to_bytes() == expect
    w2 = wp.reshape(234, -1)
    if mix == "sha":
        nz = np.nonzero(w2[233])[0]
        assert cd.prove_sparse(w2, 80, int(nz[0]) if nz.size == 1 else 0).to_bytes() == expect
    blob2, wires2 = pkg.make_circuit(d, mix, 12)     # same shape, another witness in the same block
    wp[...] = wires2
    cd2, oc2 = pkg.CircuitData(blob2), orc.OracleCircuit(blob2)
    assert cd2.prove(wp).to_bytes() == oc2.prove(wires2)[0]
    pkg.host_free(wp)
    with pytest.raises(pkg.P2GpuError):
        pkg.host_free(np.zeros(4, dtype=np.uint64))
    cd.close()
    cd2.close()


@pytest.mark.parametrize("d,mix,npi", [(6, "arith", 1), (8, "sha", 4), (9, "ecdsa", 9), (12, "ecdsa", 20)])
def test_public_inputs_proof_bytes_match_oracle(pkg, orc, gpu, d, mix, npi):
    """PoseidonGate rows + Poseidon public_inputs_hash (InnerHasher) on the GPU path."""
    blob, wires, pis = pkg.make_circuit(d, mix, 17, num_public_inputs=npi)
    cd, oc = pkg.CircuitData(blob), orc.OracleCircuit(blob)
    assert cd.constants_sigmas_cap() == oc.cap()
    expect, _ = oc.prove(wires, public_inputs=pis)
    got = cd.prove(wires, public_inputs=pis).to_bytes()
    assert got == expect and oc.verify(got)
    assert got[-8 * npi:] == pis.tobytes()
    with pytest.raises(pkg.P2GpuError):
        cd.prove(wires)  # public inputs missing
    cd.close()


@pytest.mark.parametrize("d,mix,npi", [(6, "ecdsa", 0), (11, "ecdsa", 5), (13, "sha", 0)])
def test_standard_recursion_config_proofs(pkg, orc, gpu, d, mix, npi):
    """135-wire shape (standard_recursion_config): different column counts, gate op counts and
    Keccak block tail than the 234-wire shape."""
    out = pkg.make_circuit(d, mix, 19, num_public_inputs=npi, num_wires=135, pi_row_routed_only=True)
    blob, wires = out[0], out[1]
    pis = out[2] if npi else ()
    cd, oc = pkg.CircuitData(blob), orc.OracleCircuit(blob)
    expect, _ = oc.prove(wires, public_inputs=pis)
    assert cd.prove(wires, public_inputs=pis).to_bytes() == expect
    assert cd.prove_routed(wires[:80], public_inputs=pis).to_bytes() == expect
    cd.close()


def test_golden_proof_digests_on_gpu(pkg, gpu):
    """Committed regression vectors (tests/golden/proof_digests.json) without running the oracle."""
    with open(os.path.join(GOLDEN, "proof_digests.json")) as f:
        gold = json.load(f)
    for g in gold:
        blob, wires = pkg.make_circuit(g["degree_bits"], g["mix"], g["seed"])
        cd = pkg.CircuitData(blob)
        assert hashlib.sha256(cd.constants_sigmas_cap()).hexdigest() == g["constants_sigmas_cap_sha256"]
        assert cd.circuit_digest().hex() == g["circuit_digest"]
        proof = cd.prove(wires)
        assert len(proof) == g["proof_len"]
        assert hashlib.sha256(proof.to_bytes()).hexdigest() == g["proof_sha256"]
        cd.close()


def test_two_circuits_interleaved(pkg, orc, gpu):
    """Distinct handles are independent (one in-flight prove per handle)."""
    b1, w1 = pkg.make_circuit(7, "ecdsa", 21)
    b2, w2 = pkg.make_circuit(9, "sha", 22)
    c1, c2 = pkg.CircuitData(b1), pkg.CircuitData(b2)
    p1a, p2a = c1.prove(w1).to_bytes(), c2.prove(w2).to_bytes()
    p2b, p1b = c2.prove(w2).to_bytes(), c1.prove(w1).to_bytes()
    assert p1a == p1b == orc.OracleCircuit(b1).prove(w1)[0]
    assert p2a == p2b == orc.OracleCircuit(b2).prove(w2)[0]


def test_unsatisfied_witness_gives_rejected_proof(pkg, orc, gpu):
    """Like the reference (no trim check fires when quotient_degree_factor == 2^rate_bits) an
    unsatisfied witness still yields bytes -- which the verifier rejects; and they are the same
    bytes the oracle produces."""
    blob, wires = pkg.make_circuit(8, "ecdsa", 5)
    bad = wires.copy()
    bad[3, 5] = (int(bad[3, 5]) + 1) % P
    cd, oc = pkg.CircuitData(blob), orc.OracleCircuit(blob)
    # default: the library's self-check (the verifier's identity at zeta) refuses the witness
    with pytest.raises(pkg.P2GpuError) as ei:
        cd.prove(bad)
    assert ei.value.code == -5
    assert len(cd.prove(wires)) > 0  # the handle stays usable
    cd.set("self_check", 0)
    proof = cd.prove(bad).to_bytes()
    assert proof == oc.prove(bad)[0]
    assert not oc.verify(proof)
    with pytest.raises(pkg.P2GpuError) as ei:
        cd.verify(proof)
    assert ei.value.code == -9


def test_error_paths(pkg, gpu):
    blob, wires = pkg.make_circuit(5, "arith", 1)
    bad = blob.copy()
    bad[0] ^= 0xFF
    with pytest.raises(pkg.P2GpuError) as ei:
        pkg.CircuitData(bad)
    assert ei.value.code == -1
    with pytest.raises(pkg.P2GpuError):
        pkg.CircuitData(blob[:2000])
    cd = pkg.CircuitData(blob)
    with pytest.raises(pkg.P2GpuError):
        cd.prove(wires[:, :16])          # wrong shape
    with pytest.raises(pkg.P2GpuError):
        cd.prove(wires, public_inputs=[1])  # the circuit has no public inputs
    cd.set("pow_hint", 0)                # a wrong PoW witness is refused, not emitted
    with pytest.raises(pkg.P2GpuError):
        cd.prove(wires)
    cd.set("pow_hint", (1 << 64) - 1)
    assert len(cd.prove(wires)) > 0


@pytest.mark.parametrize("d,mix", [(19, "ecdsa"), (21, "arith"), (21, "grammar")])
def test_larger_configs_are_accepted(pkg, orc, gpu, d, mix):
    """BASELINE configs[3] / configs[4] sizes on ONE GPU: 2^22 LDE rows with every gate kind,
    2^24 LDE rows (12 + 9 layer NTT, ~55 GB resident; `grammar` = SURVEY 8(d)'s gate mix for configs[4]: arithmetic,
    base-2 / base-4 sums and RandomAccessGate memory reads).  Property check: the verifier accepts."""
    blob, wires = pkg.make_circuit(d, mix, 2)
    cd = pkg.CircuitData(blob)
    proof = cd.prove(wires)
    ov = orc.OracleCircuit(blob, verifier_cap=cd.constants_sigmas_cap(), verifier_digest=cd.circuit_digest())
    assert ov.verify(proof.to_bytes())
    cd.verify(proof)
    bad = bytearray(proof.to_bytes())
    bad[len(bad) // 2] ^= 1
    assert not ov.verify(bytes(bad))
    with pytest.raises(pkg.P2GpuError):
        cd.verify(bytes(bad))
    cd.close()


@pytest.mark.parametrize("mix", ["sha", "ecdsa"])
def test_full_size_proof_is_accepted(pkg, orc, gpu, mix):
    """BASELINE size: 2^17 gates -> 2^20 LDE rows.  The oracle prover would take minutes here, so
    the check is the size-independent one the reference's tests use: the verifier accepts.  The
    verifier handle takes the constants_sigmas cap from the GPU (VerifierCircuitData), and the
    cap itself is cross-checked at 2^13 in test_proof_bytes_match_oracle."""
    d = 17
    blob, wires = pkg.make_circuit(d, mix, 1)
    cd = pkg.CircuitData(blob)
    proof = cd.prove(wires)
    ov = orc.OracleCircuit(blob, verifier_cap=cd.constants_sigmas_cap(), verifier_digest=cd.circuit_digest())
    assert ov.verify(proof.to_bytes())
    cd.verifier_data().verify(proof)
    comp = cd.compress(proof)          # the reference's on-disk format
    assert len(comp) < len(proof) and cd.decompress(comp).to_bytes() == proof.to_bytes()
    cd.verify_compressed(comp)
    # reproducible, and a tampered opening is rejected
    assert cd.prove(wires).to_bytes() == proof.to_bytes()
    bad = bytearray(proof.to_bytes())
    bad[3 * 16 * 25 + 24] ^= 1
    assert not ov.verify(bytes(bad))
    with pytest.raises(pkg.P2GpuError):
        cd.verify(bytes(bad))
    # an unsatisfied witness at full size is rejected too
    w2 = wires.copy()
    w2[7, 12345] = (int(w2[7, 12345]) + 1) % P
    with pytest.raises(pkg.P2GpuError) as ei:
        cd.prove(w2)
    assert ei.value.code == -5
    cd.set("self_check", 0)
    assert not ov.verify(cd.prove(w2).to_bytes())
    cd.close()


@pytest.mark.parametrize("d,mix,npi", [(7, "arith", 0), (9, "ecdsa", 9), (12, "ecdsa", 0), (13, "sha", 4)])
def test_fill_witness_matches_oracle(pkg, orc, gpu, d, mix, npi):
    """N1: the gates' row-local generators on the GPU.  From the routed columns alone the filled
    matrix equals the generator's full witness, and equals the oracle's fill word for word."""
    import torch

    out = pkg.make_circuit(d, mix, 23, num_public_inputs=npi, pi_row_routed_only=True)
    blob, wires = out[0], out[1]
    pis = out[2] if npi else ()
    cd, oc = pkg.CircuitData(blob), orc.OracleCircuit(blob)
    part = wires.copy()
    part[80:, :] = 0                       # drop every non-routed (gate-internal) column
    part[3, :] ^= part[3, :] & 0           # (routed columns untouched)
    dev = torch.from_numpy(part.view(np.int64)).cuda()
    cd.fill_witness(dev)
    got = dev.cpu().numpy().view(np.uint64)
    assert np.array_equal(got, oc.fill_witness(part))
    assert np.array_equal(got, wires)
    # proving from the routed columns only gives the same proof as from the full matrix
    full = cd.prove(wires, public_inputs=pis).to_bytes()
    assert cd.prove_routed(wires[:80], public_inputs=pis).to_bytes() == full
    cd.close()


def test_standalone_c_caller(pkg, orc, gpu, tmp_path):
    """`p2gpu-prove` is a plain-C program on the C ABI (no Python/torch in the process): same bytes."""
    import subprocess

    from conftest import ROOT

    exe = os.path.join(ROOT, "acvm-backend-plonky2_amd", "p2gpu-prove")
    assert os.path.exists(exe)
    blob, wires, pis = pkg.make_circuit(9, "ecdsa", 29, num_public_inputs=3, pi_row_routed_only=True)
    (tmp_path / "c.blob").write_bytes(blob.tobytes())
    (tmp_path / "w.bin").write_bytes(wires.tobytes())
    (tmp_path / "r.bin").write_bytes(np.ascontiguousarray(wires[:80]).tobytes())
    (tmp_path / "pi.bin").write_bytes(pis.tobytes())
    expect, _ = orc.OracleCircuit(blob).prove(wires, public_inputs=pis)
    # --sparse: the first 231 columns + the value of each other column in row 0 (their only non-zero row here or not:
    # the tool proves the matrix that compact form stands for, which is `wires` when the tail columns are zero elsewhere)
    w2 = wires.reshape(234, -1)
    tail_only_row0 = not w2[231:, 1:].any()
    (tmp_path / "s.bin").write_bytes(np.ascontiguousarray(w2[:231]).tobytes() + np.ascontiguousarray(w2[231:, 0]).tobytes())
    runs = [(["w.bin", "--vk", str(tmp_path / "vk.blob")], "p1.bin"), (["r.bin", "--routed"], "p2.bin")]
    if tail_only_row0:
        runs.append((["s.bin", "--sparse", "231", "0"], "p4.bin"))
    for args, out in runs:
        cmd = [exe, str(tmp_path / "c.blob"), str(tmp_path / args[0]), str(tmp_path / out), str(tmp_path / "pi.bin")] + args[1:]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        assert (tmp_path / out).read_bytes() == expect
    # `p2gpu-verify <vk> <proof>`: the reference's write_vk -> prove -> verify round trip in plain C
    vexe = os.path.join(ROOT, "acvm-backend-plonky2_amd", "p2gpu-verify")
    r = subprocess.run([vexe, str(tmp_path / "vk.blob"), str(tmp_path / "p1.bin")], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "accepted" in r.stderr, r.stderr
    tampered = bytearray(expect)
    tampered[-1] ^= 1
    (tmp_path / "bad.bin").write_bytes(bytes(tampered))
    r = subprocess.run([vexe, str(tmp_path / "vk.blob"), str(tmp_path / "bad.bin")], capture_output=True, text=True, timeout=60)
    assert r.returncode == 3 and "rejected" in r.stderr
    # --reference-format: hex of the compressed proof, the file format of the reference's CLI
    # (+ --vk: the key in the reference's VK file layout, which p2gpu-verify tells from its own blob by the magic)
    cmd = [exe, str(tmp_path / "c.blob"), str(tmp_path / "w.bin"), str(tmp_path / "p.hex"), str(tmp_path / "pi.bin"), "--reference-format",
           "--vk", str(tmp_path / "vk.ref")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    hx = (tmp_path / "p.hex").read_text()
    assert len(hx) // 2 < len(expect) and bytes.fromhex(hx)[:25] == expect[:25]
    # CommonCircuitData first (usize config.num_wires), VerifierOnlyCircuitData (cap height, cap, digest) last
    vkref = (tmp_path / "vk.ref").read_bytes()
    assert vkref[:8] == (234).to_bytes(8, "little") and vkref[-(8 + 17 * 25):][:8] == (4).to_bytes(8, "little")
    for key in ("vk.blob", "vk.ref"):
        r = subprocess.run([vexe, str(tmp_path / key), str(tmp_path / "p.hex")], capture_output=True, text=True, timeout=60)
        assert r.returncode == 0 and "accepted" in r.stderr, r.stderr
    # a broken blob is reported through the error code + message, not a crash
    (tmp_path / "bad.blob").write_bytes(b"\0" * 300)
    r = subprocess.run([exe, str(tmp_path / "bad.blob"), str(tmp_path / "w.bin"), str(tmp_path / "p3.bin")],
                       capture_output=True, text=True, timeout=60)
    assert r.returncode == 2 and "p2gpu_circuit_create" in r.stderr


# ---- one proof sharded over two ranks (coset sharding) -----------------------------
def _shard_worker(rank, world, port, d, mix, npi, q):
    import os
    import sys

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from conftest import ROOT
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    import __graft_entry__ as entry

    torch.cuda.set_device(0)  # both ranks share the one GPU of the test box; collectives over gloo
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = entry.load_package()
    out = pkg.make_circuit(d, mix, 31, num_public_inputs=npi)
    blob, wires, pis = (out + ((),))[:3] if npi == 0 else out
    cd = pkg.CircuitData(blob)
    cd.set_shard(rank, world)
    same = True
    p1 = None
    wd = torch.from_numpy(wires.view(np.int64)).cuda()
    # host witness in sharded mode: a rank reads only ITS block of columns -- the rest
    # of the matrix it is handed may be garbage, the blocks are all-gathered between the GPUs
    cpr = -(-wires.shape[0] // world)
    own = wires.copy()
    own[:rank * cpr] = 0xDEADBEEF
    own[(rank + 1) * cpr:] = 0xDEADBEEF
    # knob shard_intt: the inverse transforms of the wires / Z-PP columns replicated (0) or column-sharded with an all-gather of
    # the coefficient blocks (1) -- the same bytes either way, through every entry point, and with the column classes off
    # ... and knob shard_reduce: the FRI batch reduction replicated (0) or column-sharded with an all-gather + sum of the partial sums (1)
    # ... and knob shard_zs: the chunk quotients of the permutation argument on every rank (0) or row-sharded + all-gathered in place (1)
    for intt, zc, red, zs in ((0, 1, 0, 0), (1, 1, 1, 1), (1, 0, 0, 1), (0, 1, 1, 0)):
        cd.set("shard_intt", intt)
        cd.set("zero_columns", zc)
        cd.set("shard_reduce", red)
        cd.set("shard_zs", zs)
        got = [cd.prove(wires, public_inputs=pis).to_bytes(), cd.prove(wd, public_inputs=pis).to_bytes(), cd.prove(own, public_inputs=pis).to_bytes()]
        p1 = p1 or got[0]
        same = same and got == [p1] * 3
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, p1, same))


@pytest.mark.parametrize("world,d,mix,npi", [(2, 8, "ecdsa", 0), (2, 13, "sha", 4), (4, 9, "arith", 0), (8, 10, "ecdsa", 0), (4, 11, "sha", 0),
                                             (8, 12, "sha", 3), (8, 14, "grammar", 0), (2, 17, "sha", 0)])
def test_coset_sharded_proof_matches_oracle(pkg, orc, gpu, world, d, mix, npi):
    """`world` processes (one per GPU on a real node; here they share the GPU and talk gloo) each
    keep 8/world LDE cosets of the per-proof oracles; caps, quotient interpolants and query
    openings are all-gathered -- and, with the knob `shard_intt`, the coefficient blocks of the column-sharded inverse
    transforms.  Every rank must return the oracle's proof bytes."""
    import socket

    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_shard_worker, args=(r, world, port, d, mix, npi, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    out = pkg.make_circuit(d, mix, 31, num_public_inputs=npi)
    blob, wires = out[0], out[1]
    pis = out[2] if npi else ()
    expect, _ = orc.OracleCircuit(blob).prove(wires, public_inputs=pis)
    for rank, proof, same in res:
        assert same, rank
        assert proof == expect, rank


# ---- bit-exactness at the BASELINE sizes, without the oracle on the box ----------------------------
def _gold(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def _assert_matches_gold(pkg, blob, wires, pis, g):
    import sys

    sys.path.insert(0, GOLDEN)
    import proof_stages

    assert hashlib.sha256(blob.tobytes()).hexdigest() == g["blob_sha256"]
    assert hashlib.sha256(wires.tobytes()).hexdigest() == g["wires_sha256"]
    cd = pkg.CircuitData(blob)
    assert hashlib.sha256(cd.constants_sigmas_cap()).hexdigest() == g["constants_sigmas_cap_sha256"]
    assert cd.circuit_digest().hex() == g["circuit_digest"]
    proof = cd.prove(wires, public_inputs=pis)
    assert proof.timings["pow_witness"] == g["pow_witness"]
    assert len(proof) == g["proof_len"]
    bad = proof_stages.first_difference(blob, proof.to_bytes(), g["stages"])
    assert bad is None, f"first diverging prover stage: {bad}"
    assert hashlib.sha256(proof.to_bytes()).hexdigest() == g["proof_sha256"]
    cd.verify(proof)
    # the other entry points and the shortcuts' off switches at this size: resident witness, the compact witness
    # when the matrix has that form (the unused wires non-zero in one common row only), everything dense
    import torch

    want = g["proof_sha256"]
    assert hashlib.sha256(cd.prove(torch.from_numpy(wires.view(np.int64)).cuda(), public_inputs=pis).to_bytes()).hexdigest() == want
    wm = wires.reshape(cd.num_wires, -1)
    nzc = (wm != 0).sum(axis=1)
    ncols = int(np.max(np.nonzero(nzc > 1)[0])) + 1 if (nzc > 1).any() else 0
    rows = {int(np.nonzero(wm[j])[0][0]) for j in range(ncols, cd.num_wires) if nzc[j] == 1}
    if len(rows) == 1 and ncols < cd.num_wires:
        assert hashlib.sha256(cd.prove_sparse(wires, ncols, rows.pop(), public_inputs=pis).to_bytes()).hexdigest() == want
    if wires.size <= 234 << 17:
        cd.set("zero_columns", 0)
        assert hashlib.sha256(cd.prove(wires, public_inputs=pis).to_bytes()).hexdigest() == want
    cd.close()


@pytest.mark.parametrize("idx", range(6))
def test_baseline_size_proofs_are_bit_exact(pkg, gpu, idx):
    """BASELINE.json configs[2] (2^20 LDE rows: `sha` = the bench workload, `ecdsa` = every gate kind,
    `sha` + 4 public inputs = PoseidonGate rows), configs[3] (2^22 LDE rows, every gate kind) and -- round 5 -- configs[4]
    (2^24 LDE rows: the `grammar` mix, and the bench workload's mix at that size): the GPU proof must equal the ORACLE's,
    stage by stage and as a whole.  The oracle ran in the build container (tests/golden/gen_proof_digests.py --large /
    --xlarge; minutes of CPU, the 2^24-row LDEs spilled to disk); only SHA-256 digests travel."""
    gold = _gold("proof_digests_large.json")
    if idx >= len(gold):
        pytest.skip("no such entry in proof_digests_large.json")
    g = gold[idx]
    out = pkg.make_circuit(g["degree_bits"], g["mix"], g["seed"], num_public_inputs=g["public_inputs"])
    blob, wires = out[0], out[1]
    pis = out[2] if g["public_inputs"] else ()
    _assert_matches_gold(pkg, blob, wires, pis, g)


def test_hand_written_acir_circuits_on_gpu(pkg, gpu):
    """BASELINE.json configs[0] counterpart: the fibonacci example program as a hand-written
    ACIR-equivalent circuit (tests/golden/mini_builder.py), proved on the GPU: same bytes as the oracle's."""
    import sys

    sys.path.insert(0, GOLDEN)
    import mini_builder

    gold = {g["name"]: g for g in _gold("proof_digests_hand.json")}
    for name, fn in (("fibonacci", mini_builder.fibonacci), ("quadratic_example", mini_builder.quadratic_example)):
        blob, wires = fn()
        _assert_matches_gold(pkg, blob, wires, (), gold[name])


# ---- RCCL inside the library ------------------------------------------------------------------------
@pytest.mark.parametrize("d,mix,npi", [(9, "ecdsa", 0), (13, "sha", 4)])
def test_rccl_transport_single_rank(pkg, orc, gpu, d, mix, npi):
    """p2gpu_circuit_set_shard_rccl on the one GPU of the test box: librccl is resolved with dlopen, a
    1-rank communicator is created from an ncclUniqueId, and with `shard_exercise` every exchange step of
    a sharded proof (cap all-gather, quotient interpolants, PoW minimum, query openings) really goes
    through ncclAllGather on the circuit's stream.  Same bytes as the oracle, and as the plain path."""
    out = pkg.make_circuit(d, mix, 37, num_public_inputs=npi)
    blob, wires = out[0], out[1]
    pis = out[2] if npi else ()
    cd = pkg.CircuitData(blob)
    plain = cd.prove(wires, public_inputs=pis).to_bytes()
    cd.set_shard(0, 1, transport="rccl")
    cd.set("shard_exercise", 1)
    got = cd.prove(wires, public_inputs=pis).to_bytes()
    assert got == plain == orc.OracleCircuit(blob).prove(wires, public_inputs=pis)[0]
    cd.set("shard_zs", 1)       # one rank computes every row's chunk quotients; the (one-block) all-gather is in place
    cd.set("shard_reduce", 1)   # one rank sums every column; its "partial" sum goes through the all-gather and the adding kernel
    assert cd.prove(wires, public_inputs=pis).to_bytes() == plain
    cd.set("shard_intt", 1)     # one rank owns every block: the transform runs block-wise; the exchange has no peer, but the grouped
                                # ncclSend / ncclRecv pair still executes once, to the rank itself (transport.hip, world 1 + shard_exercise)
    assert cd.prove(wires, public_inputs=pis).to_bytes() == plain
    cd.set("shard_intt", 0)
    cd.set("shard_reduce", 0)
    cd.set("shard_zs", 0)
    cd.set("shard_exercise", 0)
    assert cd.prove(wires, public_inputs=pis).to_bytes() == plain
    cd.close()


def _rccl_worker(rank, world, port, d, mix, q):
    import os
    import sys

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    from conftest import ROOT
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    import __graft_entry__ as entry

    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    pkg = entry.load_package()
    import ctypes
    dev = (ctypes.c_int * 1)(rank)
    assert pkg.load_library().p2gpu_init(dev, 1) == 0
    blob, wires = pkg.make_circuit(d, mix, 31)
    cd = pkg.CircuitData(blob)
    cd.set_shard(rank, world)           # nccl backend -> RCCL inside the library
    p1 = cd.prove(wires).to_bytes()
    p2 = cd.prove(torch.from_numpy(wires.view(np.int64)).cuda()).to_bytes()
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, p1, p1 == p2))


@pytest.mark.parametrize("world,d,mix", [(2, 12, "ecdsa"), (4, 13, "sha"), (8, 14, "arith")])
def test_coset_sharded_proof_over_rccl(pkg, orc, gpu, world, d, mix):
    """One proof over `world` GPUs with RCCL (xGMI) as the transport: runs whenever the box has that many
    GPUs (the single-GPU test box skips it; the transport's single-rank form is the test above, and the
    sharding logic itself is covered for 2/4/8 ranks over gloo in test_coset_sharded_proof_matches_oracle)."""
    import socket

    import torch
    import torch.multiprocessing as mp

    if torch.cuda.device_count() < world:
        pytest.skip(f"needs {world} GPUs, this box has {torch.cuda.device_count()}")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rccl_worker, args=(r, world, port, d, mix, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=900) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    blob, wires = pkg.make_circuit(d, mix, 31)
    expect, _ = orc.OracleCircuit(blob).prove(wires)
    for rank, proof, same in res:
        assert same and proof == expect, rank


# ---- X1: PoseidonGoldilocksConfig (Poseidon Merkle trees, challenger, digest) ---------------------------
@pytest.mark.parametrize("d,mix,npi", [(5, "arith", 0), (8, "ecdsa", 3), (10, "sha", 0), (13, "ecdsa", 0), (14, "sha", 4)])
def test_poseidon_hasher_proofs_match_oracle(pkg, orc, gpu, d, mix, npi):
    """hasher = 1 in the blob: every Merkle tree (leaves absorbed 8 elements per Poseidon permutation, nodes =
    compress(l, r)), the Fiat-Shamir challenger, the PoW grind and the circuit digest use Poseidon-Goldilocks
    instead of Keccak-256/25 -- the north_star's "Poseidon-GL Merkle-tree hashing" mode (the reference itself
    runs KeccakGoldilocksConfig, lib.rs:13).  GPU proof == oracle proof, both verifiers accept, and the
    compressed format round-trips with 32-byte digests."""
    out = pkg.make_circuit(d, mix, 43, num_public_inputs=npi, hasher=1)
    blob, wires = out[0], out[1]
    pis = out[2] if npi else ()
    cd, oc = pkg.CircuitData(blob), orc.OracleCircuit(blob)
    assert cd.hash_bytes() == 32
    assert cd.constants_sigmas_cap() == oc.cap() and cd.circuit_digest() == oc.digest()
    expect, tr = oc.prove(wires, public_inputs=pis)
    got = cd.prove(wires, public_inputs=pis)
    assert got.timings["pow_witness"] == tr.pow_witness
    assert got.to_bytes() == expect
    assert oc.verify(expect)
    cd.verify(got)
    vd = cd.verifier_data()
    vd.verify(got)
    comp = cd.compress(got)
    assert cd.decompress(comp).to_bytes() == expect
    vd.verify_compressed(comp)
    import torch
    assert cd.prove(torch.from_numpy(wires.view(np.int64)).cuda(), public_inputs=pis).to_bytes() == expect
    bad = bytearray(expect)
    bad[40] ^= 1
    with pytest.raises(pkg.P2GpuError):
        cd.verify(bytes(bad))
    # the same circuit under the reference's Keccak configuration gives a different (and shorter) proof
    kb = blob.copy()
    kb[:256].view(np.uint32)[22] = 0
    ck = pkg.CircuitData(kb)
    assert ck.hash_bytes() == 25 and len(ck.prove(wires, public_inputs=pis)) < len(expect)
    ck.close()
    cd.close()


def test_poseidon_hasher_golden_digests_at_full_size(pkg, gpu):
    """PoseidonGoldilocksConfig (X1) bit-exact at BASELINE configs[2]'s size: the ORACLE proved the bench workload under hasher 1 in the build
    container (tests/golden/gen_poseidon_digest.py; its Poseidon is the plain form, round by round -- the device runs the partial rounds'
    linear layers three at a time, poseidon.hpp); the GPU proof must have the same bytes, through the host-witness and the resident entry."""
    import torch

    for g in _gold("proof_digests_poseidon.json"):
        out = pkg.make_circuit(g["degree_bits"], g["mix"], g["seed"], num_public_inputs=g["public_inputs"], hasher=g["hasher"])
        blob, wires = out[0], out[1]
        pis = out[2] if g["public_inputs"] else ()
        assert hashlib.sha256(blob.tobytes()).hexdigest() == g["blob_sha256"] and hashlib.sha256(wires.tobytes()).hexdigest() == g["wires_sha256"]
        cd = pkg.CircuitData(blob)
        assert cd.hash_bytes() == 32
        assert hashlib.sha256(cd.constants_sigmas_cap()).hexdigest() == g["constants_sigmas_cap_sha256"] and cd.circuit_digest().hex() == g["circuit_digest"]
        proof = cd.prove(wires, public_inputs=pis)
        assert proof.timings["pow_witness"] == g["pow_witness"] and len(proof) == g["proof_len"]
        assert hashlib.sha256(proof.to_bytes()).hexdigest() == g["proof_sha256"]
        assert hashlib.sha256(cd.prove(torch.from_numpy(wires.view(np.int64)).cuda(), public_inputs=pis).to_bytes()).hexdigest() == g["proof_sha256"]
        cd.verify(proof)
        cd.close()


def test_poseidon_hasher_full_size_and_sharded_exercise(pkg, orc, gpu):
    """2^20 LDE rows with the Poseidon hasher: verifier acceptance (the oracle would take minutes), and the
    exchange steps of a sharded proof through RCCL with 32-byte digests."""
    blob, wires = pkg.make_circuit(17, "sha", 1, hasher=1)
    cd = pkg.CircuitData(blob)
    proof = cd.prove(wires)
    ov = orc.OracleCircuit(blob, verifier_cap=cd.constants_sigmas_cap(), verifier_digest=cd.circuit_digest())
    assert ov.verify(proof.to_bytes())
    cd.verifier_data().verify(proof)
    cd.set_shard(0, 1, transport="rccl")
    cd.set("shard_exercise", 1)
    assert cd.prove(wires).to_bytes() == proof.to_bytes()
    cd.close()


# ---- one process, several devices: the device group behind p2gpu_init(ids, n > 1) ----------------------------------------
@pytest.mark.parametrize("world", [2, 4, 8])
def test_exchange_plan_matches_the_library(pkg, gpu, world):
    """parallel.exchange_plan (the host-side restatement DESIGN.md section 7's table and the CPU test are built on) against what
    the library itself counts for a sharded proof: `profile` = 2 brackets every shard_allgather with events and files it under
    its payload class -- number of exchanges and bytes over all ranks must agree, resident and host witness."""
    import torch

    def classes(plan):
        out = {"small": [0, 0], "mid": [0, 0], "big": [0, 0]}
        for _, b in plan:
            mx, total = pkg.parallel.exchange_bytes(b, world)
            k = "small" if mx <= 4096 else ("mid" if mx < (1 << 20) else "big")
            out[k][0] += 1
            out[k][1] += total
        return out

    try:
        pkg.init([0] * world)
        d = 15
        blob, wires = pkg.make_circuit(d, "sha", 9)
        hdr = blob[:256].view(np.uint32)
        cd = pkg.CircuitData(blob)
        wd = torch.from_numpy(wires.view(np.int64)).cuda()
        # the dense wire columns as the library classifies them: anything that is non-zero outside the one row the unused wires share
        wm = wires.reshape(int(hdr[3]), -1)
        nzrows = [np.nonzero(col)[0] for col in wm]
        single = [int(r[0]) for r in nzrows if len(r) == 1]
        pi_row = max(set(single), key=single.count) if single else -1
        dense_list = [j for j, r in enumerate(nzrows) if len(r) > 1 or (len(r) == 1 and int(r[0]) != pi_row)]
        for host, w, ncols, intt, red, zs in ((False, wd, None, 0, 0, 0), (True, wires, None, 0, 1, 1), (False, wd, None, 1, 1, 0), (True, wires, None, 1, 0, 1)):
            cd.set("shard_intt", intt)
            cd.set("shard_reduce", red)
            cd.set("shard_zs", zs)
            plan = pkg.parallel.exchange_plan(d, world, num_wires=int(hdr[3]), num_constants_sigmas=int(hdr[5]) + int(hdr[4]), host_witness=host,
                                              dense_columns=ncols, shard_intt=bool(intt), dense_list=dense_list, shard_reduce=bool(red), shard_zs=bool(zs))
            assert len(plan) == 8 + int(host) + 2 * intt + red + zs
            want = classes(plan)
            cd.prove(w)
            cd.set("profile", 2)
            cd.prove(w)
            st = cd.kernel_stats()
            cd.set("profile", 0)
            got = {"small": [0, 0], "mid": [0, 0], "big": [0, 0]}
            for k, v in st.items():
                if k.startswith("exchange["):
                    key = "small" if "<=4KB" in k else ("mid" if "<1MB" in k else "big")
                    got[key] = [v["launches"], int(v["bytes"])]
            extra_pow = got["small"][0] - want["small"][0]      # a second grinding batch adds one 8-byte exchange (rare)
            assert 0 <= extra_pow <= 2, (got, want)
            want["small"][0] += extra_pow
            want["small"][1] += 8 * world * extra_pow
            assert got == want, (host, ncols, got, want)
        cd.close()
    finally:
        pkg.init([0])


@pytest.mark.parametrize("world", [2, 4, 8])
def test_single_process_device_group(pkg, orc, gpu, world):
    """p2gpu_init with several ids makes every circuit handle a device group: ONE caller thread, one proof coset-sharded
    over the ranks, exchanges as peer copies between the ranks' streams (no RCCL, no second process) -- what a single
    `circuit_data.prove` call site (prove_action.rs:96) can drive.  On this one-GPU box the ids repeat (the ranks share
    the GPU): every code path of the group runs -- rank threads, rendezvous, peer all-gathers of caps / quotient
    interpolants / PoW minima / query openings / the witness column blocks -- and the bytes equal the oracle's, through
    every entry point, with and without public inputs, and after a failing proof."""
    import torch

    try:
        pkg.init([0] * world)
        for d, mix, npi in ((9, "ecdsa", 0), (12, "sha", 3)):
            out = pkg.make_circuit(d, mix, 41, num_public_inputs=npi, pi_row_routed_only=True)
            blob, wires = out[0], out[1]
            pis = out[2] if npi else ()
            want = orc.OracleCircuit(blob).prove(wires, public_inputs=pis)[0]
            cd = pkg.CircuitData(blob)
            assert cd.prove(wires, public_inputs=pis).to_bytes() == want                      # host witness: column blocks per rank + peer all-gather
            wd = torch.from_numpy(wires.view(np.int64)).cuda()
            assert cd.prove(wd, public_inputs=pis).to_bytes() == want                         # resident witness
            assert cd.prove_routed(np.ascontiguousarray(wires[:80]), public_inputs=pis).to_bytes() == want
            wm = wires.reshape(cd.num_wires, -1)
            nz = (wm != 0).sum(axis=1)
            ncols = int(np.max(np.nonzero(nz > 1)[0])) + 1 if (nz > 1).any() else 0
            if ncols < cd.num_wires and not wm[ncols:, 1:].any():
                sp = cd.prove_sparse(np.ascontiguousarray(wm[:ncols]).reshape(-1), ncols, 0, public_inputs=pis, tail=np.ascontiguousarray(wm[ncols:, 0]))
                assert sp.to_bytes() == want
            cd.verify(want)
            # column-sharded inverse transforms: the coefficient blocks go rank to rank as peer copies, in place
            cd.set("shard_intt", 1)
            cd.set("shard_reduce", 1)   # ... and the FRI batch reduction column-sharded, partial sums all-gathered and added
            cd.set("shard_zs", 1)       # ... and the permutation argument's chunk quotients row-sharded, blocks all-gathered in place
            assert cd.prove(wires, public_inputs=pis).to_bytes() == want
            assert cd.prove(wd, public_inputs=pis).to_bytes() == want
            assert cd.prove_routed(np.ascontiguousarray(wires[:80]), public_inputs=pis).to_bytes() == want
            cd.set("zero_columns", 0)
            assert cd.prove(wd, public_inputs=pis).to_bytes() == want
            cd.set("zero_columns", 1)
            cd.set("shard_intt", 0)
            cd.set("shard_reduce", 0)
            cd.set("shard_zs", 0)
            # an unsatisfied witness fails on every rank at the same point; the group is usable afterwards
            bad = wires.copy()
            bad[0, 1] = (int(bad[0, 1]) + 1) % P
            with pytest.raises(pkg.P2GpuError) as e:
                cd.prove(bad, public_inputs=pis)
            assert e.value.code == -5
            assert cd.prove(wires, public_inputs=pis).to_bytes() == want
            with pytest.raises(pkg.P2GpuError):
                cd.set_shard(0, 1)       # the sharding of a group is fixed
            # the same process can still make PLAIN handles on a named device of the list (p2gpu_circuit_create_on): replicas
            # next to the group, each proving whole proofs from its own thread
            import threading
            plain = [pkg.CircuitData(blob, device=0) for _ in range(2)]
            got = [None, None]

            def work(i):
                got[i] = plain[i].prove(wires, public_inputs=pis).to_bytes()
            th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
            for t in th:
                t.start()
            assert cd.prove(wd, public_inputs=pis).to_bytes() == want    # the group proves while the replicas do
            for t in th:
                t.join()
            assert got == [want, want]
            with pytest.raises(pkg.P2GpuError):
                pkg.CircuitData(blob, device=5)    # not in the init list
            for h in plain:
                h.close()
            cd.close()
        assert pkg.peer_access() == [[-1] * world for _ in range(world)]   # every id is device 0 here: no pair to enable
    finally:
        pkg.init([0])
    # back to one device: ordinary handles again
    blob, wires = pkg.make_circuit(8, "arith", 3)
    cd = pkg.CircuitData(blob)
    assert cd.prove(wires).to_bytes() == orc.OracleCircuit(blob).prove(wires)[0]
    cd.close()


# ---- FRI / commitment parameters other than the reference's (cap 2^4, 16 PoW bits) -------------------------------------
@pytest.mark.parametrize("d,mix,cap_h,pow_bits,queries", [(8, "ecdsa", 3, 16, 28), (8, "sha", 5, 8, 10), (9, "arith", 7, 16, 28),
                                                           (8, "ecdsa", 11, 12, 5), (10, "sha", 12, 16, 28), (9, "sha", 4, 21, 28)])
def test_other_cap_heights_and_pow_bits(pkg, orc, gpu, d, mix, cap_h, pow_bits, queries):
    """The pinned staging arena of a proof is sized from the circuit: 3 + n_steps trees stage 2^cap_h digests
    each, and the PoW loop takes its staging words once however many candidate batches it grinds through (21 bits:
    ~2^21 candidates expected, several batches).  Same circuit rebuilt with other cap heights / PoW bits / query counts
    through p2gpu_build_blob (cap heights from rate_bits -- a coset owns whole cap subtrees -- up to rate_bits + d, where the cap
    IS the leaf level): GPU bytes == oracle bytes, both verifiers accept."""
    import ctypes
    import test_build as tb

    blob0 = pkg.make_circuit(d, mix, 61)[0]
    wires = pkg.make_circuit(d, mix, 61)[1]
    params, gates, ng, row_gate, gconst, copies = tb.decompose(blob0)
    params.cap_height, params.proof_of_work_bits, params.num_query_rounds = cap_h, pow_bits, queries
    fn = pkg.load_library().p2gpu_build_blob
    fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                   ctypes.c_size_t, ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t)]
    blob = tb.build_with(fn, params, gates, ng, row_gate, gconst, copies)
    h = blob[:256].view(np.uint32)
    assert int(h[10]) == cap_h and int(h[11]) == pow_bits and int(h[12]) == queries
    cd = pkg.CircuitData(blob)
    oc = orc.OracleCircuit(blob)
    want, tr = oc.prove(wires)
    for _ in range(2):       # twice on one handle: the arena is reset, not regrown
        got = cd.prove(wires)
        assert got.to_bytes() == want and got.timings["pow_witness"] == tr.pow_witness
    assert oc.verify(want)
    cd.verify(want)
    # the stage-level commitment operator has its own small arena
    v = _rand((5, 1 << d), 7)
    assert pkg.commit_values(v, 3, cap_h) == orc.commit_values(v, 3, cap_h)
    cd.close()
    oc.close()


@pytest.mark.parametrize("d,mix,npi,W", [(10, "ecdsa", 0, 234), (12, "ecdsa", 3, 234), (13, "grammar", 0, 234), (11, "ecdsa", 2, 135), (15, "ecdsa", 0, 234)])
def test_half_domain_gates_keep_every_byte(pkg, orc, gpu, d, mix, npi, W):
    """Round 6: the folded constraint sums of the gates of degree <= 4 (the reference's five custom gates, BaseSum<4>) are
    evaluated on the four even LDE cosets only and extended to the odd ones by interpolation (plonk.hip gate_sums_kernel) --
    a polynomial of degree < 4n is fixed by 4n values, so every quotient word, and with it every proof byte, is the one the
    direct evaluation gives: knob `half_gates` 1 (default) / 0, against the oracle (which evaluates every gate on every row),
    with public inputs (PoseidonGate rows next to them), on the 135-wire configuration, and on an unsatisfied witness."""
    out = pkg.make_circuit(d, mix, 53, num_public_inputs=npi, num_wires=W)
    blob, wires = out[0], out[1]
    pis = out[2] if npi else ()
    want = orc.OracleCircuit(blob).prove(wires, public_inputs=pis)[0]
    cd = pkg.CircuitData(blob)
    cd.set("profile", 2)
    assert cd.prove(wires, public_inputs=pis).to_bytes() == want
    ran = any(k.startswith("gate_sums_kernel") for k in cd.kernel_stats())
    # default (1): only where it pays -- constraints of such gates x gates >= 12 M: the heavy mix from 2^14 gates on
    assert ran == (mix == "ecdsa" and d >= 15), "default: the half-domain path runs only on circuits large enough for it to pay"
    cd.set("profile", 0)
    cd.set("half_gates", 2)     # whatever the size
    cd.set("profile", 2)
    assert cd.prove(wires, public_inputs=pis).to_bytes() == want
    ran = any(k.startswith("gate_sums_kernel") for k in cd.kernel_stats())
    assert ran == (mix == "ecdsa"), "forced: the half-domain path runs exactly where a gate of degree <= 4 has >= 48 constraints"
    cd.set("profile", 0)
    cd.set("half_gates", 0)
    cd.set("profile", 2)
    assert cd.prove(wires, public_inputs=pis).to_bytes() == want
    assert not any(k.startswith("gate_sums_kernel") for k in cd.kernel_stats())
    cd.set("profile", 0)
    cd.set("half_gates", 2)
    # a witness that breaks a range-checked limb of a half-domain gate: the identity fails at zeta whichever way the sums were made
    import torch
    bad = wires.copy()
    rows = np.nonzero((bad[W - 40] > 0) & (bad[W - 40] < 4))[0]    # a 2-bit limb of a U32 gate row (not the PublicInputGate row's random filler)
    if len(rows):
        bad[W - 40, rows[0]] = (int(bad[W - 40, rows[0]]) + 1) % P
        with pytest.raises(pkg.P2GpuError) as e:
            cd.prove(bad, public_inputs=pis)
        assert e.value.code == -5
    assert cd.prove(torch.from_numpy(wires.view(np.int64)).cuda(), public_inputs=pis).to_bytes() == want
    cd.close()


@pytest.mark.parametrize("env", [{"P2GPU_HOST_PRESCAN": "0"}, {"P2GPU_NTT_DIRECT": "0", "P2GPU_LEAF_LEVELS": "0"}, {"P2GPU_HALF_GATES": "0"}])
def test_measurement_switches_keep_the_bytes(pkg, gpu, env):
    """The A/B switches select the older kernels / paths (the whole wire matrix over PCIe; the generic NTT passes and no tree
    levels inside the leaf hash -- the one-lane tree tails of rounds 1-2 behind P2GPU_COOP_TAIL left the tree in round 5): read once per process, so a child process proves a golden circuit with the switch set and the digest
    must still be the committed one."""
    import subprocess
    import sys

    g = max(_gold("proof_digests.json"), key=lambda x: x["degree_bits"])
    code = (
        "import hashlib, sys; sys.path.insert(0, %r); import __graft_entry__ as ge; pkg = ge.load_package();"
        "out = pkg.make_circuit(%d, %r, %d, num_public_inputs=%d); cd = pkg.CircuitData(out[0]);"
        "p = cd.prove(out[1], public_inputs=(out[2] if %d else ())); print('DIGEST', hashlib.sha256(p.to_bytes()).hexdigest())"
        % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), g["degree_bits"], g["mix"], g["seed"], g["public_inputs"],
           g["public_inputs"]))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env={**os.environ, **env})
    assert r.returncode == 0, r.stderr[-2000:]
    assert ("DIGEST " + g["proof_sha256"]) in r.stdout, r.stdout[-500:]


def test_seeded_differential_fuzz(pkg, orc, gpu):
    """scratch/fuzz_parity.py as a bounded, seeded test: random (degree, gate mix, seed, public inputs,
    width), random structure of the unused wire columns (zeroed, extra rows, stray values), random knobs toggled between
    proofs on one handle (zero_columns / virtual_columns), all four entry points (host matrix, resident, routed,
    sparse with a random split): every proof byte-equal to the oracle's; verifier and compress / decompress round trip."""
    import time

    import torch

    rng = np.random.default_rng(20260930)
    t0, done = time.time(), 0
    for it in range(14):
        if time.time() - t0 > 25 and done >= 6:
            break
        d = int(rng.integers(5, 13))
        mix = ["arith", "sha", "ecdsa"][int(rng.integers(0, 3))]
        seed = int(rng.integers(1, 1 << 30))
        npi = int(rng.choice([0, 0, 1, 4, 9, 17]))
        nw = int(rng.choice([234, 234, 135]))
        routed_only = bool(rng.integers(0, 2))
        if nw == 135 and mix == "ecdsa":
            mix = "sha"   # the ECC gates need the wide config
        out = pkg.make_circuit(d, mix, seed, num_public_inputs=npi, num_wires=nw, pi_row_routed_only=routed_only)
        blob, wires = out[0], out[1]
        pis = out[2] if npi else ()
        oc, cd = orc.OracleCircuit(blob), pkg.CircuitData(blob)
        W, n = nw, 1 << d
        try:
            for rnd in range(2):
                w = wires.copy().reshape(W, n)
                mutated = rnd > 0
                if mutated:  # perturb the structure of random columns (the witness may stop satisfying the circuit: bytes still comparable)
                    for c_ in rng.integers(0, W, size=int(rng.integers(1, 6))):
                        k = int(rng.integers(0, 4))
                        if k == 0:
                            w[c_, :] = 0
                        elif k == 1:
                            w[c_, int(rng.integers(0, n))] = int(rng.integers(1, 1 << 62))
                        elif k == 2:
                            w[c_, :] = 0
                            w[c_, int(rng.integers(0, n))] = 5
                        else:
                            w[c_, rng.integers(0, n, size=3)] = 9
                w = np.ascontiguousarray(w)
                cd.set("self_check", 0 if mutated else 1)
                cd.set("zero_columns", int(rng.integers(0, 4) != 0))
                cd.set("virtual_columns", int(rng.integers(0, 3) != 0))
                cd.set("half_gates", int(rng.integers(0, 3)))      # 0 never / 1 where it pays / 2 always
                expect, _ = oc.prove(w, public_inputs=pis)
                tag = (it, rnd, d, mix, seed, npi, nw, routed_only)
                assert cd.prove(w, public_inputs=pis).to_bytes() == expect, ("host",) + tag
                assert cd.prove(torch.from_numpy(w.view(np.int64)).cuda(), public_inputs=pis).to_bytes() == expect, ("dev",) + tag
                if routed_only and not mutated:
                    assert cd.prove_routed(w[:80], public_inputs=pis).to_bytes() == expect, ("routed",) + tag
                row = int(rng.integers(0, n))
                dense = np.nonzero(np.delete(w, row, axis=1).any(axis=1))[0]
                lo = int(dense.max()) + 1 if dense.size else 0
                ncols = int(rng.integers(lo, W + 1))
                assert cd.prove_sparse(w, ncols, row, public_inputs=pis).to_bytes() == expect, ("sparse",) + tag
                if not mutated:
                    cd.verify(expect)
                    assert cd.decompress(cd.compress(expect)).to_bytes() == expect
        finally:
            cd.close()
            oc.close()
        done += 1
    assert done >= 6
