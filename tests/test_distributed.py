"""N > 1 path on CPU: world_size-2 gloo processes (one process per GPU on the real node).
Replica scheduling of independent proofs (no data-path collective) and the Merkle-cap
all-gather of the coset-sharded scheme."""
import hashlib
import os
import socket
import sys

import pytest

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2")
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    import __graft_entry__ as entry

    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg, orc = entry.load_package(), entry.load_oracle()
    par = pkg.parallel
    # (1) replicas: 5 independent proofs over 2 ranks
    mine = list(par.partition(5, world, rank))
    digests = {}
    for u in mine:
        blob, wires = pkg.make_circuit(5, "arith", seed=100 + u)
        oc = orc.OracleCircuit(blob)
        proof, _ = oc.prove(wires)
        assert oc.verify(proof)
        digests[u] = hashlib.sha256(proof).hexdigest()
    gathered = [None] * world
    dist.all_gather_object(gathered, digests)
    # (2) cap all-gather of the coset-sharded scheme
    blob, wires = pkg.make_circuit(5, "sha", seed=3)
    full = orc.commit_values(wires[:20], 3, 4)
    own = par.owned_cap_entries(world, rank)
    local = b"".join(full[25 * i: 25 * i + 25] for i in own)
    cap = par.all_gather_cap(local, world, rank)
    t = par.max_over_ranks(float(rank + 1))
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, mine, gathered, cap == full, t))


def test_two_rank_gloo():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == [0, 1, 2] and res[1][1] == [3, 4]
    merged = {}
    for d in res[0][2]:
        merged.update(d)
    assert sorted(merged) == [0, 1, 2, 3, 4] and len(set(merged.values())) == 5
    assert res[0][2] == res[1][2]
    assert res[0][3] and res[1][3]
    assert res[0][4] == res[1][4] == 2.0


def test_partition_and_ownership(pkg):
    par = pkg.parallel
    for n in (0, 1, 7, 8, 9):
        for w in (1, 2, 4, 8):
            parts = [list(par.partition(n, w, r)) for r in range(w)]
            assert sum(parts, []) == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    for w in (1, 2, 4, 8):
        owned = sorted(sum((par.owned_cap_entries(w, r) for r in range(w)), []))
        assert owned == list(range(16))
    assert par.owned_cap_entries(8, 1) == [8, 9]  # coset 1 -> bitrev3(1) = 4 -> entries 8, 9
