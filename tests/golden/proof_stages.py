"""Cuts a proof (plonky2 `ProofWithPublicInputs::to_bytes`, uncompressed) into its
prover stages and hashes each one, so that a parity failure at the BASELINE sizes names the stage
that diverged instead of just "the bytes differ".  Pure Python; needs only the circuit blob header
(include/p2gpu.h).  Used by gen_proof_digests.py (oracle side, build container) and by the GPU tests
(product side) -- the two never run in the same place at these sizes.

Stages, in byte order:
  wires_cap | zs_partial_products_cap | quotient_polys_cap      (prove steps 1, 2, 3)
  openings                                                       (OpeningSet at zeta / g zeta)
  fri_commit_caps                                                (one cap per arity-16 fold step)
  fri_queries                                                    (28 x initial-tree rows + paths, step leaves + paths)
  final_poly | pow_witness | public_inputs
"""
import hashlib

import numpy as np


def header(blob):
    h = np.frombuffer(bytes(blob[:256]), dtype=np.uint32)
    assert h[0] == 0x43473250
    return dict(d=int(h[2]), W=int(h[3]), R=int(h[4]), NC=int(h[5]), K=int(h[7]), QF=int(h[8]), rate_bits=int(h[9]),
                cap_h=int(h[10]), queries=int(h[12]), steps=int(h[13]), arity=[int(x) for x in h[14:14 + int(h[13])]],
                n_pi=int(h[24]), PP=int(h[26]))


def stages(blob, proof):
    """{stage name: bytes}; the concatenation in insertion order is the proof."""
    c = header(blob)
    cap = 25 << c["cap_h"]
    n_open = c["NC"] + c["R"] + c["W"] + 2 * c["K"] + c["K"] * c["PP"] + c["K"] * c["QF"]
    final_len = 16 << (c["d"] - sum(c["arity"]))
    tail = final_len + 8 + 8 * c["n_pi"]
    off, out = 0, {}
    for name, size in (("wires_cap", cap), ("zs_partial_products_cap", cap), ("quotient_polys_cap", cap),
                       ("openings", 16 * n_open), ("fri_commit_caps", cap * c["steps"])):
        out[name] = proof[off:off + size]
        off += size
    assert len(proof) >= off + tail
    out["fri_queries"] = proof[off:len(proof) - tail]
    off = len(proof) - tail
    out["final_poly"] = proof[off:off + final_len]
    out["pow_witness"] = proof[off + final_len:off + final_len + 8]
    out["public_inputs"] = proof[off + final_len + 8:]
    assert b"".join(out.values()) == proof
    return out


def stage_digests(blob, proof):
    return {k: hashlib.sha256(v).hexdigest() for k, v in stages(blob, proof).items()}


def first_difference(blob, proof, expected_digests):
    """Name of the first stage whose SHA-256 differs from `expected_digests` (None if all match)."""
    for k, v in stage_digests(blob, proof).items():
        if expected_digests[k] != v:
            return k
    return None
