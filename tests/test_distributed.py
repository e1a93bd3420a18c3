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
    # each rank contributes the subtree roots of ITS cosets (Python restatement of the ownership map);
    # the library's own host code (p2gpu_shard_assemble_cap, used by the prover's exchange path) orders them
    cap = par.all_gather_cap(par.local_roots(full, world, rank), world, rank)
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
    # library assembly (product code) vs the Python ownership map, for every world size, no collective needed
    import ctypes

    import numpy as np

    lib = pkg.load_library()
    lib.p2gpu_shard_assemble_cap.argtypes = [ctypes.c_int, ctypes.c_uint, ctypes.c_uint, ctypes.c_void_p, ctypes.c_void_p]
    full = bytes(np.random.default_rng(5).integers(0, 256, size=25 * 16, dtype=np.uint8))
    for w in (1, 2, 4, 8):
        g = np.frombuffer(b"".join(par.local_roots(full, w, r) for r in range(w)), dtype=np.uint8).copy()
        out = np.zeros(25 * 16, dtype=np.uint8)
        assert lib.p2gpu_shard_assemble_cap(w, 3, 4, g.ctypes.data, out.ctypes.data) == 0
        assert out.tobytes() == full
    assert lib.p2gpu_shard_assemble_cap(3, 3, 4, g.ctypes.data, out.ctypes.data) != 0  # 8 cosets over 3 ranks


def _bench(*argv, env=None):
    import subprocess

    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e.update(OMP_NUM_THREADS="2", **(env or {}))
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True, timeout=600, env=e)


def test_bench_gpus_flag_launches_its_own_ranks():
    """`bench.py --gpus 2` outside torchrun starts two ranks itself (the flag used to be parsed and ignored).
    --dry --backend gloo runs the rank plumbing only -- rendezvous, barrier, the max-over-ranks reduction -- with no GPU."""
    import json

    r = _bench("--gpus", "2", "--backend", "gloo", "--dry")
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["ranks"] == [0, 1] and line["launched_by"] == "bench.py"
    assert line["max_over_ranks_s"] >= 0.02   # rank 1 sleeps 20 ms: the reduction really took the MAX over both ranks
    # one rank, no launcher
    r1 = _bench("--dry")
    assert r1.returncode == 0 and json.loads(r1.stdout.splitlines()[-1])["n_gpus"] == 1


def test_bench_refuses_more_gpus_than_the_box_has():
    """On a box with fewer devices than --gpus asks for the bench fails loudly -- exit code 2, no JSON line -- instead of
    printing a line with a smaller n_gpus (this container has no GPU at all; the GPU boxes of the test pool have one)."""
    import torch

    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    r = _bench("--gpus", str(have + 1 if have else 2))
    assert r.returncode == 2, (r.returncode, r.stderr[-500:])
    assert "HIP device" in r.stderr and not any(ln.startswith("{") for ln in r.stdout.splitlines())
    # a rank whose torchrun world disagrees with --gpus refuses as well (the line's n_gpus must be the N asked for)
    r2 = _bench("--gpus", "4", "--dry", env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(_free_port())})
    assert r2.returncode == 2 and "WORLD_SIZE=2" in r2.stderr


@pytest.mark.parametrize("d", [17, 19, 21])
@pytest.mark.parametrize("world", [2, 4, 8])
def test_exchange_plan_of_a_sharded_proof(pkg, d, world):
    """DESIGN.md section 7's table, walked without a GPU: the exchanges of one coset-sharded proof, their
    number and their bytes, for the BASELINE sizes on 2 / 4 / 8 GPUs.  `exchange_plan` restates the shard_allgather call sites of
    csrc/prover.hip; the GPU suite checks it against what the library counts (test_exchange_plan_matches_the_library)."""
    n, G = 1 << d, world
    plan = pkg.parallel.exchange_plan(d, world)
    names = [p[0] for p in plan]
    assert names == ["wires cap", "Z / partial products cap", "quotient interpolants", "quotient cap", "opening partial sums",
                     "first FRI tree cap", "PoW minima", "query rows and paths"]
    by = dict(plan)
    # the north_star's "all-gather to reassemble Merkle caps": 16 entries x 32 B over all ranks, four commitments
    assert by["wires cap"] * G == 16 * 32 and sum(1 for p in plan if p[0].endswith("cap")) == 4
    # the quotient's per-coset interpolants: 2 challenges x N x 8 B over all ranks (16.8 MB at d = 17)
    assert by["quotient interpolants"] * G == 2 * 8 * n * 8
    # 354 + 2 opened polynomials, 16 partial sums of 16 B each at these sizes, split over the ranks: 91 KB in total
    assert by["opening partial sums"] == -(-356 // G) * 16 * 16
    assert by["PoW minima"] == 8
    steps = len([1 for db in range(d, 5, -4)])
    per_query = (84 + 234 + 20 + 16) + 4 * 4 * (d - 1) + sum(32 + 4 * (d - 4 * (s + 1) - 1) for s in range(steps))
    assert by["query rows and paths"] == 8 * 28 * per_query
    # everything but the interpolants is latency: < 0.5 MB per rank and proof
    assert sum(b for nm, b in plan if nm != "quotient interpolants") < 512 * 1024
    # the host-witness entry point adds exactly one exchange: W / G columns per rank
    hw = pkg.parallel.exchange_plan(d, world, host_witness=True)
    assert len(hw) == len(plan) + 1 and hw[0] == ("witness column blocks", 8 * -(-234 // G) * n) and hw[1:] == plan
    # ... the compact entry point (p2gpu_prove_sparse): every rank uploads the dense columns itself, nothing is exchanged
    assert pkg.parallel.exchange_plan(d, world, host_witness=True, dense_columns=80) == plan
    # knob shard_intt: two more exchanges, the coefficient blocks of the column-sharded inverse transforms --
    # unequal blocks, bytes per rank; only dense columns (and what lies between two of one block) travel
    par = pkg.parallel
    dense = list(range(80)) + list(range(135, 155))          # e.g. routed wires + a second run of gate wires
    sp = par.exchange_plan(d, world, shard_intt=True, dense_list=dense)
    assert [p[0] for p in sp] == ["wires coefficient blocks", "wires cap", "Z / partial products coefficient blocks"] + names[1:]
    blocks = par.intt_blocks(dense, G)
    owned = [[c for c in dense if lo <= c < hi] for lo, hi in blocks]
    assert sum(owned, []) == dense and max(map(len, owned)) - min(map(len, owned)) <= 1
    assert all(a[1] <= b[0] for a, b in zip(blocks, blocks[1:]) if b != (0, 0))      # disjoint, in order
    wb = dict(sp)["wires coefficient blocks"]
    assert isinstance(wb, tuple) and len(wb) == G and wb == tuple(8 * (hi - lo) * n for lo, hi in blocks)
    assert 8 * len(dense) * n <= sum(wb) <= 8 * (len(dense) + 55) * n     # at most one gap (the 55 structured columns) rides along
    zb = dict(sp)["Z / partial products coefficient blocks"]
    assert sum(zb) == 8 * 20 * n and max(zb) == 8 * -(-20 // G) * n
    assert par.exchange_bytes(wb, G) == (max(wb), sum(wb)) and par.exchange_bytes(64, G) == (64, 64 * G)
    assert par.intt_blocks([], 4) == [(0, 0)] * 4 and par.intt_blocks([5], 2) == [(0, 0), (5, 6)]
    assert par.exchange_budget(sp, world) > par.exchange_budget(plan, world)
    # knob shard_zs: the chunk quotients of the permutation argument by rows, 2 x (10 + 1) columns of n / G rows per rank
    zp_ = par.exchange_plan(d, world, shard_zs=True)
    assert [p[0] for p in zp_] == names[:1] + ["Z chunk quotient blocks"] + names[1:] and dict(zp_)["Z chunk quotient blocks"] == 8 * 22 * (n // G)
    # knob shard_reduce: one more exchange, the partial sums of the column-sharded FRI batch reduction
    rp = par.exchange_plan(d, world, shard_reduce=True)
    assert [p[0] for p in rp] == names[:5] + ["FRI batch-reduce partial sums"] + names[5:] and dict(rp)["FRI batch-reduce partial sums"] == 16 * n
    # budget: eight exchanges at ~60 us each + the interpolants over G - 1 links at once
    t = pkg.parallel.exchange_budget(plan, world)
    assert 8 * 60e-6 < t < 8 * 60e-6 + 2 * 8 * n * 8 / G / 153e9 + 1e-4
