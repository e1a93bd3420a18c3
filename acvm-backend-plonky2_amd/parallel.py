"""One process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on the GPU box,
"gloo" in the CPU tests).

Two ways the `prove` path spreads over the GPUs of a node:

* replicas (throughput, proofs/sec): proofs are independent units -- rank r proves the units
  `partition(n, world, r)`; no data-path collective.  This is what bench.py runs for N > 1.
* coset sharding of ONE proof (latency): GPU q owns the LDE cosets {r : r mod G == q}; because
  plonky2's leaf index is bitrev(8k + r) = bitrev3(r) * n + bitrev(k), coset r owns whole cap
  subtrees -- cap entries 2*bitrev3(r) and 2*bitrev3(r)+1 -- so the only exchange a commitment
  needs is an all-gather of 16 x 25-byte cap entries (`all_gather_cap`), after which every rank
  runs the Fiat-Shamir transcript replicated.  This mode is implemented inside the library
  (`CircuitData.set_shard(rank, world)` -> p2gpu_circuit_set_shard); the helpers below state the
  ownership map and are what the CPU-only gloo test exercises.
"""
import numpy as np


def partition(n_units, world_size, rank):
    """Contiguous block partition of `n_units` independent proofs; sizes differ by at most 1."""
    base, rem = divmod(n_units, world_size)
    lo = rank * base + min(rank, rem)
    return range(lo, lo + base + (1 if rank < rem else 0))


def _bitrev(x, bits):
    r = 0
    for _ in range(bits):
        r = (r << 1) | (x & 1)
        x >>= 1
    return r


def owned_cosets(world_size, rank, rate_bits=3):
    return [r for r in range(1 << rate_bits) if r % world_size == rank]


def owned_cap_entries(world_size, rank, rate_bits=3, cap_height=4):
    """plonky2 cap indices owned by `rank` under coset sharding (whole subtrees, no halo)."""
    per = 1 << (cap_height - rate_bits)
    out = []
    for r in owned_cosets(world_size, rank, rate_bits):
        base = _bitrev(r, rate_bits) * per
        out += list(range(base, base + per))
    return out


def all_gather_cap(local_entries, world_size, rank, rate_bits=3, cap_height=4, group=None):
    """Reassemble a Merkle cap from each rank's owned entries (bytes, 25 B per entry, in the
    order of `owned_cap_entries`).  One small all-gather; returns the full cap bytes."""
    import torch
    import torch.distributed as dist

    n_own = len(owned_cap_entries(world_size, rank, rate_bits, cap_height))
    assert len(local_entries) == 25 * n_own
    backend = dist.get_backend(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    mine = torch.frombuffer(bytearray(local_entries), dtype=torch.uint8).to(dev)
    gathered = torch.empty(world_size * mine.numel(), dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(gathered, mine, group=group)
    g = gathered.cpu().numpy().reshape(world_size, n_own, 25)
    cap = np.zeros((1 << cap_height, 25), dtype=np.uint8)
    for q in range(world_size):
        for j, idx in enumerate(owned_cap_entries(world_size, q, rate_bits, cap_height)):
            cap[idx] = g[q, j]
    return cap.tobytes()


def max_over_ranks(seconds, group=None):
    """Timing reduction used by bench.py: the slowest rank defines the step time."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return seconds
    backend = dist.get_backend(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    t = torch.tensor([seconds], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
