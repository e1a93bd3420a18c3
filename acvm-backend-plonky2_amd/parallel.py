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
  (`CircuitData.set_shard(rank, world)` -> p2gpu_circuit_set_shard_rccl: RCCL called by the library on
  its own stream).  `local_roots` / `owned_cap_entries` restate the ownership map in Python;
  `all_gather_cap` runs the same exchange with torch.distributed as the transport and the library's own
  cap assembly (p2gpu_shard_assemble_cap) -- which is what the CPU-only gloo test exercises.
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


def local_roots(full_cap, world_size, rank, rate_bits=3, cap_height=4):
    """The subtree roots rank `rank` holds after building its cosets' trees, cut out of a complete cap
    (bytes, 25 B per entry): [local coset z][k] -> cap entry bitrev(rank + z * world) * per + bitrev(k),
    each padded to the 32 B a digest occupies in device memory.  (What a rank sends into the all-gather.)"""
    per = 1 << (cap_height - rate_bits)
    lgp = cap_height - rate_bits
    out = bytearray()
    for r in owned_cosets(world_size, rank, rate_bits):
        for k in range(per):
            idx = _bitrev(r, rate_bits) * per + _bitrev(k, lgp)
            out += full_cap[25 * idx:25 * idx + 25] + bytes(7)
    return bytes(out)


def all_gather_cap(local_roots32, world_size, rank, rate_bits=3, cap_height=4, group=None):
    """The commitment-time collective of a coset-sharded proof with torch.distributed as the transport:
    all-gather every rank's local subtree roots (32 B each, `local_roots` order) and let the LIBRARY put
    the cap into plonky2 order -- p2gpu_shard_assemble_cap is the same host code the prover's own
    exchange path runs after its ncclAllGather (csrc/hostcore.hip shard_assemble_cap)."""
    import ctypes

    import torch
    import torch.distributed as dist

    from .prover import _check, load_library

    backend = dist.get_backend(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    mine = torch.frombuffer(bytearray(local_roots32), dtype=torch.uint8).to(dev)
    gathered = torch.empty(world_size * mine.numel(), dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(gathered, mine, group=group)
    g = np.ascontiguousarray(gathered.cpu().numpy())
    assert g.size == 32 << cap_height
    cap = np.zeros(25 << cap_height, dtype=np.uint8)
    lib = load_library()
    lib.p2gpu_shard_assemble_cap.argtypes = [ctypes.c_int, ctypes.c_uint, ctypes.c_uint, ctypes.c_void_p, ctypes.c_void_p]
    _check(lib.p2gpu_shard_assemble_cap(world_size, rate_bits, cap_height, g.ctypes.data, cap.ctypes.data))
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


def intt_blocks(dense, world_size):
    """Column blocks of the column-sharded inverse transform (knob `shard_intt`; csrc/transport.hip intt_blocks): the sorted list of
    dense columns is dealt out in `world_size` contiguous runs whose sizes differ by at most one; rank p's block is the column
    range [first, last + 1) of its run -- what it transforms and what it sends (structured columns in between ride along).
    -> [(lo, hi)] per rank, (0, 0) for a rank with no dense column."""
    dense = sorted(dense)
    nd, G = len(dense), world_size
    out = []
    for p in range(G):
        s, e = p * nd // G, (p + 1) * nd // G
        out.append((dense[s], dense[e - 1] + 1) if e > s else (0, 0))
    return out


def exchange_bytes(entry_bytes, world_size):
    """(bytes the busiest rank sends, bytes over all ranks) of one plan entry: an int is an all-gather of equal blocks, a tuple
    holds the per-rank bytes of an all-gather of unequal blocks (the coefficient blocks of `shard_intt`)."""
    if isinstance(entry_bytes, tuple):
        return max(entry_bytes), sum(entry_bytes)
    return entry_bytes, entry_bytes * world_size


def exchange_plan(degree_bits, world_size, num_wires=234, num_constants_sigmas=84, num_challenges=2, partial_products=9,
                  quotient_degree_factor=8, rate_bits=3, cap_height=4, num_queries=28, host_witness=False, dense_columns=None,
                  shard_intt=False, dense_list=None, shard_reduce=False, shard_zs=False, num_routed=80):
    """The exchange steps of ONE coset-sharded proof (csrc/prover.hip shard_allgather call sites), in
    order: [(what, bytes each rank sends)].  Every step is an all-gather over the `world_size` ranks, so a rank receives
    (world_size - 1) x those bytes (a tuple: unequal blocks, bytes per rank -- `exchange_bytes`); nothing else crosses between the GPUs.  Host-side restatement for tests and budgets: the
    GPU tests compare it with what the library counts (`profile` = 2: the exchange[...] pseudo-kernels of p2gpu_kernel_stats)."""
    d, G, K = degree_bits, world_size, num_challenges
    n, C = 1 << d, 1 << rate_bits
    assert C % G == 0, "GPU q owns the cosets r = q (mod G): G divides 2^rate_bits"
    cl = C // G                                    # cosets per rank
    cap_bytes = cl * ((1 << cap_height) // C) * 32  # local subtree roots, 32 B per digest in device memory
    nzp, nq = K * (1 + partial_products), K * quotient_degree_factor
    nall = num_constants_sigmas + num_wires + nzp + nq
    arity = []
    db = d
    while db > 5 and db + rate_bits - 4 >= cap_height:   # ConstantArityBits(4, 5) as build() derives it (hostcore.hip:290)
        arity.append(4)
        db -= 4
    parts = 1
    while parts < 16 and (n // (parts * 2)) >= 1024:
        parts *= 2
    plan = []
    if host_witness and (dense_columns is None or dense_columns >= num_wires):
        # p2gpu_prove (the whole matrix in host RAM): every rank pulls W / G columns over its own PCIe link and the blocks are
        # exchanged.  Through p2gpu_prove_sparse (dense_columns < W: the caller says which wires are unused) every rank uploads
        # the dense columns itself and nothing is exchanged
        plan.append(("witness column blocks", 8 * -(-num_wires // G) * n))
    if shard_intt:
        # every rank runs the inverse transform of ITS block of the dense wire columns and the coefficient
        # blocks are all-gathered in place (unequal blocks: a tuple of per-rank bytes); structured columns never travel unless they
        # lie between two dense columns of one block.  dense_list: the dense wire columns (default: the first `dense_columns`)
        dl = list(dense_list) if dense_list is not None else list(range(num_wires if dense_columns is None else dense_columns))
        plan.append(("wires coefficient blocks", tuple(8 * (hi - lo) * n for lo, hi in intt_blocks(dl, G))))
    plan.append(("wires cap", cap_bytes))
    if shard_zs:
        # the chunk quotients of the permutation argument (and the row products behind them) are computed
        # for n / G rows per rank and all-gathered in place; the scan runs on every rank
        nchunks = -(-num_routed // quotient_degree_factor)
        plan.append(("Z chunk quotient blocks", 8 * K * (nchunks + 1) * (n // G)))
    if shard_intt:
        plan.append(("Z / partial products coefficient blocks", tuple(8 * (hi - lo) * n for lo, hi in intt_blocks(range(nzp), G))))
    plan.append(("Z / partial products cap", cap_bytes))
    plan.append(("quotient interpolants", K * cl * n * 8))
    plan.append(("quotient cap", cap_bytes))
    plan.append(("opening partial sums", -(-(nall + K) // G) * parts * 2 * 8))
    if shard_reduce:
        # SURVEY 8(e) step 8, second half: every rank batch-reduces only its block of the concatenated columns; the partial sums
        # F0_q (an extension-field polynomial: 2 n words) are all-gathered and added
        plan.append(("FRI batch-reduce partial sums", 16 * n))
    plan.append(("first FRI tree cap", cap_bytes))      # later FRI trees are small and built whole on every rank
    plan.append(("PoW minima", 8))                      # one per grinding batch; the first batch almost always holds a witness
    per_query = 0
    for cols in (num_constants_sigmas, num_wires, nzp, nq):
        per_query += cols + 4 * (d - 1)                  # the row + its Merkle path (leaves n per coset -> 2 per coset)
    dcur = d
    for ab in arity:
        per_query += 2 * (1 << ab) + 4 * (dcur - ab - 1)
        dcur -= ab
    plan.append(("query rows and paths", 8 * num_queries * per_query))
    return plan


def exchange_budget(plan, world_size, link_gbps=153.0, latency_us=60.0):
    """Seconds of one sharded proof spent in its exchanges under a simple model: every rank drives all its world_size - 1 xGMI
    links at once (direct peer copies / grouped send-recv, not a ring), `latency_us` per exchange (measured between two ranks on one
    GPU: 60 us below 1 MB, profiles/NUMBERS.md), `link_gbps` per link and direction."""
    return sum(latency_us * 1e-6 + exchange_bytes(b, world_size)[0] / (link_gbps * 1e9) for _, b in plan)
