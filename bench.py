#!/usr/bin/env python3
"""bench.py -- proofs/sec and prove latency of the MI355X `prove` hot path.

One "step" = one proof of the BASELINE.json workload (configs[2]: SHA256-like circuit,
2^17 gates = 2^20 LDE rows, wide_ecc_config shape), witness matrix resident in HBM, proof
bytes back on the host (the boundary of p2gpu_prove_dev).  N > 1: one process per GPU
(torch.distributed / RCCL), every rank proves its own independent proofs (replicas, weak
scaling, no data-path collective); time = max over ranks between two barriers.

BASELINE.json's metric has two halves and the line carries both: `value` = proofs/sec, the throughput of
a proving service that keeps --in-flight proofs (default 4; separate circuit handles / HIP streams, one
host thread each) in flight per GPU, K steps = K proofs, ms_per_step = wall time / K; and
`latency_ms_single_proof` = the prove latency, measured in its own pass with exactly one proof on the GPU
at a time (`single_proof` also gives that pass's proofs/sec -- the round-1 headline).  A lone proof leaves
the GPU idle at its eleven Fiat-Shamir sync points and under-filled in the top levels of every Merkle tree
(one dependent Keccak-f per level); other proofs' kernels fill those holes.

The timed region runs the production path only: no per-launch events, no oracle.  After it, in
the same process, separate passes collect what the JSON line reports beside `value`:
  * `single_proof`: one proof at a time (latency);
  * `host_witness`: the same proofs through p2gpu_prove (witness in host RAM, the 245 MB H2D inside
    the call) -- SURVEY.md 8(d)'s boundary-to-boundary number, never `value`;
  * `pipelined` (only with --in-flight 1): three proofs in flight on one GPU;
  * `roofline` + `kernel_ms_per_proof`: per-launch HIP events on the library's launch stream
    (`profile` knob) over a few extra proofs;
  * `cpu_baseline` (N = 1): the oracle = CPU port, on THIS workload, all host cores, unscaled;
  * `cold_process` (N = 1): a fresh process that reads the circuit and the witness, creates the handle and proves ONCE
    (the plain-C `p2gpu-prove --timing`) -- what one invocation of the reference's CLI would see (prove_action.rs:27-43).
Prints ONE JSON line (< 6 KB: compact_line) on rank 0; the full result goes to bench_detail.json beside it (--detail).

`--gpus N` with N > 1 is self-contained: started WITHOUT torchrun the command itself launches N ranks (one process per
device, RCCL) and exits non-zero if the box has fewer than N devices; started BY torchrun (WORLD_SIZE set) it is one of
the ranks and insists that WORLD_SIZE == N.  One run measures both halves of BASELINE.json's metric: `value` = replicas
(proofs/s over all N devices), and ONE proof sharded over the N devices -- `latency_ms_sharded` through RCCL called by
the library from the N ranks, `latency_ms_sharded_group` through a device group of ONE process (peer copies), with the
per-exchange microseconds (HIP events) and the peer-access matrix of the group.
"""
import argparse
import glob
import json
import os
import sys
import time

# The MI355X boxes show 256 hardware threads and grant a cgroup quota of 16 CPUs: thread pools sized by the former (OpenMP in
# torch / numpy's BLAS: 256 spinning workers per process) run into the quota and every thread of the process group -- the
# prover's host threads too -- is frozen for the rest of the 100 ms period (cpu.stat nr_throttled; eight ranks make it eight times
# worse).  Nothing here needs them; the oracle leg sets its own count (_oracle_leg).
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "4")

import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

import bench_support as bs  # noqa: E402
from bench_support import (HBM_PEAK_GBPS, LINE_LIMIT, cold_process, compact_line, dry_run, effective_cores,  # noqa: E402,F401
                           exchange_stats, group_probe, issue_roofline, launch_ranks, newest, pmc_traffic, processed_bytes, step_of,
                           survey_bytes, valu_floor)


# ---- the CPU baseline: the ONLY place outside tests/ and smoke() that runs anything under oracle/ (timed as a baseline, never the product) ----
def _oracle_leg(d, mix, n_pi, threads, timeout=900):
    """One oracle proof of synth(d, mix) in a fresh process with OMP_NUM_THREADS=threads -> (seconds, phase seconds)."""
    import subprocess
    code = ("import sys,time,json;sys.path.insert(0,%r);import __graft_entry__ as e;p=e.load_package();o=e.load_oracle();"
            "m=p.make_circuit(%d,%r,seed=1,num_public_inputs=%d);c=o.OracleCircuit(m[0]);t=time.perf_counter();"
            "pr,tr=c.prove(m[1],public_inputs=(m[2] if %d else ()));dt=time.perf_counter()-t;"
            "print(json.dumps([dt,tr.t_wires,tr.t_zs,tr.t_quotient,tr.t_openings,tr.t_fri,o.lib().orc_num_threads()]))"
            % (ROOT, d, mix, n_pi, n_pi))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=timeout,
                       env=dict(os.environ, OMP_NUM_THREADS=str(threads), OMP_PROC_BIND="false"))
    v = json.loads(r.stdout.strip().splitlines()[-1])
    return v[0], dict(zip(["wires", "zs", "quotient", "openings", "fri"], v[1:6])), int(v[6])


def cpu_baseline(pkg, d, mix, n_pi, single_thread_bits):
    """The oracle (CPU port of the same path, oracle/) on the SAME circuit the GPU was timed on, on the host CPUs this
    process may use (cgroup quota, see effective_cores), one full proof, no scaling; plus a one-thread leg on a bounded
    smaller sample (a full one-thread proof at 2^20 rows takes ~40 s).  Test infrastructure, timed as a baseline --
    never the target."""
    import subprocess
    cores = effective_cores()
    try:
        cpu = subprocess.run(["sh", "-c", "grep -m1 'model name' /proc/cpuinfo | cut -d: -f2"], capture_output=True, text=True).stdout.strip()
    except Exception:
        cpu = ""
    try:
        dt, phases, used = _oracle_leg(d, mix, n_pi, cores)
    except Exception as e:  # the baseline is informative; never fail the bench for it
        return {"error": str(e)[:300]}
    out = {
        "value": 1.0 / dt, "unit": "proofs/sec", "cores": int(used), "kind": "port",
        "sample": f"1 full proof of synth(d={d},{mix}) = 2^{d + 3} LDE rows (the benchmarked circuit, unscaled) in {dt:.2f} s on "
                  f"{used} OpenMP threads = the CPU quota of this box ({os.cpu_count()} hardware threads visible; {cpu}); oracle/ C "
                  f"restatement, not upstream plonky2 (no AVX2 field/Keccak kernels)",
        "seconds": dt,
        "phase_seconds": phases,
    }
    if single_thread_bits:
        ds = min(17 if single_thread_bits < 0 else single_thread_bits, d)
        try:
            t1, ph1, _ = _oracle_leg(ds, mix, n_pi if ds == d else 0, 1, timeout=900)
            scale = 1 << (d - ds)
            out["single_thread"] = {"seconds": t1 * scale, "proofs_per_sec": 1.0 / (t1 * scale), "cores": 1, "phase_seconds": ph1,
                                    "sample": (f"1 full proof of synth(d={ds},{mix}) = 2^{ds + 3} LDE rows, OMP_NUM_THREADS=1" +
                                               ("" if scale == 1 else f", x{scale} (linear in rows; ignores the NTT log factor, which favours the CPU)")),
                                    "scaled": scale != 1, "parallel_speedup": (t1 * scale) / dt}
        except Exception as e:
            out["single_thread"] = {"error": str(e)[:200]}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=192,
                    help="proofs in the timed region (default 192 = 0.9 s at 2^20 rows: the clock governor needs a few tenths of a second of "
                         "sustained load, 48 proofs read 4 % low)")
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--clock-warmup-ms", type=float, default=400.0,
                    help="untimed proofs continue after the W warm-up ones until this much time has passed (0: off): steady-state "
                         "shader clock in the timed region however small W is; `clock_warmup_proofs` in the output")
    ap.add_argument("--degree-bits", type=int, default=17)
    ap.add_argument("--mix", default="sha")
    ap.add_argument("--workload", choices=["synth", "sha256"], default="synth",
                    help="synth: synthetic circuit with the reference's shape (SURVEY 8(d)); sha256: the circuit the reference's "
                         "translator builds for --sha-blocks chained Sha256Compression opcodes (translate.py; 4 blocks = 2^17 gates "
                         "= 2^20 LDE rows; translating takes ~15 s per block in Python, outside every timed region)")
    ap.add_argument("--sha-blocks", type=int, default=4)
    ap.add_argument("--hasher", choices=["keccak", "poseidon"], default="keccak",
                    help="keccak = KeccakGoldilocksConfig (the reference, default); poseidon = PoseidonGoldilocksConfig "
                         "(Poseidon Merkle trees / challenger / digest; synth workload only)")
    ap.add_argument("--public-inputs", type=int, default=0,
                    help="public inputs of the synthetic circuit (> 0 adds PoseidonGate rows to the circuit and the "
                         "PoseidonGate to the gate set every LDE row evaluates); 0 = the BASELINE parity shape")
    ap.add_argument("--in-flight", type=int, default=4,
                    help="independent proofs in flight per GPU IN THE TIMED REGION (separate circuit handles / HIP streams, "
                         "one host thread each; measured on MI355X at 2^20 rows, round 3's final code, one box: 1 -> 172, 2 -> 201, 3 -> 212, "
                         "4 -> 213, 6 -> 216, 8 -> 213 proofs/s); 1 = strictly one proof at a time, so ms_per_step is a prove latency.  Capped so that the "
                         "handles fit HBM (2^23 rows: 2, 2^24 rows: 1); the prove latency is measured in its own pass either way")
    ap.add_argument("--pipelined", type=int, default=3,
                    help="N = 1: after the timed region, also measure this many proofs in flight (0 = skip)")
    ap.add_argument("--mode", choices=["replicas", "sharded"], default="replicas",
                    help="N > 1: `replicas` = independent proofs per GPU (throughput, weak scaling, default); "
                         "`sharded` = ONE proof coset-sharded over the N GPUs (latency, strong scaling; RCCL "
                         "all-gathers of caps / quotient interpolants / query openings)")
    ap.add_argument("--backend", default=os.environ.get("P2GPU_BENCH_BACKEND", "nccl"),
                    help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to exercise the "
                         "multi-rank flow on a box with fewer GPUs than ranks)")
    ap.add_argument("--group", default="",
                    help="comma-separated HIP device ids driven by THIS ONE process as a device group (p2gpu_init with several "
                         "ids): every proof is coset-sharded over them, exchanges are peer copies between the ranks' streams. "
                         "Not combined with --gpus > 1.  An id may repeat (ranks sharing a GPU: functional check on a one-GPU box)")
    ap.add_argument("--blocking-sync", choices=["auto", "0", "1"], default="auto",
                    help="knob blocking_sync of every handle: 1 = the host threads sleep at the transcript sync points instead of "
                         "spinning in hipStreamSynchronize.  auto: 1 when ranks x proofs in flight exceed half the CPUs this process "
                         "may use (cgroup quota), e.g. 8 ranks x 4 threads on a 16-CPU quota; 0 otherwise (lowest latency)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-single-thread-bits", type=int, default=-1,
                    help="degree bits of the one-thread oracle leg: -1 (default) = the benchmarked circuit itself up to 2^17 gates "
                         "(~27 s at 2^20 LDE rows, unscaled), 0 = skip the one-thread leg")
    ap.add_argument("--profile-steps", type=int, default=4)
    ap.add_argument("--min-timed-s", type=float, default=None,
                    help="repeat the K-proof timed region until at least this much proving has been timed (default 1.5 s; 0 with "
                         "--timed-only): ms_per_step is the MEDIAN over the repeats, min / max / repeats are in the line")
    ap.add_argument("--sharded-steps", type=int, default=6,
                    help="N > 1: proofs of the sharded-latency half (ONE proof over the N devices; 0 = skip that half)")
    ap.add_argument("--no-group-probe", action="store_true",
                    help="N > 1: skip the single-process device-group half (rank 0 starts `bench.py --group 0..N-1 --group-probe`)")
    ap.add_argument("--group-probe", action="store_true", help="(internal) --group ids: print the device-group latency JSON and exit")
    ap.add_argument("--no-cold-process", action="store_true", help="N = 1: skip the fresh-process prove (cold_process)")
    ap.add_argument("--dry", action="store_true",
                    help="rank plumbing only, no GPU work: rendezvous + barrier + max-over-ranks, one JSON line naming the ranks "
                         "(the CPU-side test of `--gpus N`)")
    ap.add_argument("--detail", default="",
                    help="where the full result goes (default bench_detail.json beside bench.py); stdout is ONE line under 6 KB")
    ap.add_argument("--timed-only", action="store_true",
                    help="warm-up + timed region only (no host-witness / pipelined / event-profile / CPU passes): the command "
                         "rocprofv3 wraps (scratch/prof.sh), so that its per-kernel averages are those of the timed path")
    args = ap.parse_args()
    if args.min_timed_s is None:
        args.min_timed_s = 0.0 if args.timed_only else 1.5
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    # --gpus N > 1 outside torchrun: this process launches the N ranks itself
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.group:
        sys.exit(launch_ranks(args, sys.argv[1:]))
    if args.dry:
        sys.exit(dry_run(args))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and not args.group:
        # the line's n_gpus must be the N that was asked for: a mismatch between torchrun's world and --gpus is an error
        print(f"bench.py --gpus {args.gpus} is running as one of WORLD_SIZE={world} ranks: launch it with --nproc-per-node {args.gpus} "
              f"(or without torchrun: it starts its own ranks)", file=sys.stderr, flush=True)
        sys.exit(2)
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    if args.backend == "nccl" and world > torch.cuda.device_count():
        print(f"bench.py --gpus {world}: {torch.cuda.device_count()} HIP device(s) on this box", file=sys.stderr, flush=True)
        sys.exit(2)
    if args.backend != "nccl":  # test mode: ranks may share a device
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
    # one rank per node (re)builds; the others wait so that concurrent `make`s never race on the .so
    if int(os.environ.get("LOCAL_RANK", "0")) == 0:
        entry.build()
    if world > 1:
        dist.barrier()
    pkg = entry.load_package()
    lib = pkg.load_library()
    import ctypes
    group = [int(x) for x in args.group.split(",") if x.strip() != ""]
    if group:
        assert world == 1, "--group is the single-process multi-GPU mode: run it without torchrun"
        pkg.init(group)
        args.in_flight = 1
    else:
        dev = (ctypes.c_int * 1)(local_rank)
        assert lib.p2gpu_init(dev, 1) == 0, lib.p2gpu_last_error()

    d, mix = args.degree_bits, args.mix
    bs.PROFILE_KEY = (("poseidon" if args.hasher == "poseidon" and mix == "sha" else mix) if args.workload == "synth" else "sha256", d)
    # every rank proves its own witness of the same circuit shape (independent proofs)
    sharded = args.mode == "sharded" and world > 1
    if args.workload == "sha256":
        # a message of --sha-blocks 64-byte blocks hashed by chained compression opcodes (each one's output state
        # is the next one's hash_values), through the restated translator; every rank hashes its own message
        cb = pkg.translate.CircuitBuilderFromAcirToPlonky2()
        nb = args.sha_blocks
        ops, w0 = [], 24 * 0
        state_w = list(range(16 * nb, 16 * nb + 8))
        nxt = 16 * nb + 8
        for i in range(nb):
            outs = list(range(nxt, nxt + 8))
            ops.append(("sha256_compression", list(range(16 * i, 16 * i + 16)), state_w, outs))
            state_w, nxt = outs, nxt + 8
        cb.translate_circuit(ops)
        rng = np.random.default_rng(1 + rank)
        wit = {i: int(v) for i, v in enumerate(rng.integers(0, 1 << 32, size=16 * nb))}
        wit.update({16 * nb + i: v for i, v in enumerate([0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19])})
        blob, wires = cb.build(wit)
        pis = ()
        d = int(blob[:256].view(np.uint32)[2])
        mix = f"sha256 x{nb} blocks (translated)"
    else:
        made = pkg.make_circuit(d, mix, seed=1 if sharded else 1 + rank, num_public_inputs=args.public_inputs,
                                hasher=1 if args.hasher == "poseidon" else 0)
        blob, wires = made[0], made[1]
        pis = made[2] if args.public_inputs else ()
    if args.group_probe:
        assert group, "--group-probe needs --group"
        print(json.dumps(group_probe(pkg, args, group, blob, wires, pis)), flush=True)
        return
    # proofs in flight = handles: as many as asked for, as long as HBM holds them (a handle allocates everything at create:
    # 0.4 GB at 2^20 rows, ~55 GB at 2^24 -- 288 GB take four of those; rounds 1-5 ran 2^24 rows one proof at a time)
    S_want = 1 if sharded else max(1, min(args.in_flight, args.steps, 8))
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    cds = [pkg.CircuitData(blob)]
    handle_bytes = max(1, free0 - torch.cuda.mem_get_info()[0])
    while len(cds) < S_want and torch.cuda.mem_get_info()[0] > 1.1 * handle_bytes + wires.nbytes + (4 << 30):
        cds.append(pkg.CircuitData(blob))
    S = len(cds)
    blocking = (world * S * 2 > effective_cores()) if args.blocking_sync == "auto" else args.blocking_sync == "1"
    if blocking:
        for cd_ in cds:
            cd_.set("blocking_sync", 1)
    if os.environ.get("P2GPU_BENCH_NO_SELF_CHECK"):  # kernel-timing experiments with deliberately wrong kernels (scratch/) only
        for cd_ in cds:
            cd_.set("self_check", 0)
    if args.mode == "sharded":
        # world 1: same code path as a replica (nothing to exchange); world > 1: RCCL inside the library
        cds[0].set_shard(rank, world, transport="rccl" if args.backend == "nccl" else None)
    wires_dev = torch.from_numpy(wires.view(np.int64)).cuda()
    torch.cuda.synchronize()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    import threading

    def run(handles, n_proofs, collect=None, w=None):
        """n_proofs proofs over the handles (thread i proves every len(handles)-th proof)."""
        w = wires_dev if w is None else w
        k = len(handles)

        def work(i):
            last = None
            for _ in range(i, n_proofs, k):
                if isinstance(w, tuple):   # ("sparse", dense columns, ncols, row, tail): p2gpu_prove_sparse
                    last = handles[i].prove_sparse(w[1], w[2], w[3], public_inputs=pis, tail=w[4])
                else:
                    last = handles[i].prove(w, public_inputs=pis)
                if collect is not None:
                    collect.append(last.timings)
            results[i] = last
        results = [None] * k
        if k == 1:
            work(0)
        else:
            th = [threading.Thread(target=work, args=(i,)) for i in range(k)]
            for t in th:
                t.start()
            for t in th:
                t.join()
        return next(r for r in results if r is not None)

    # ---- warm-up, then the timed region: K proofs, nothing but the production path ----
    tw0 = time.perf_counter()
    proof = run(cds, max(args.warmup, S))
    # clock warm-up (untimed, reported): the governor raises the shader clock over a few tenths of a second of sustained load
    # (the same binary reads 209 proofs/s over 48 timed proofs and 218 over 192); a short W would leave the timed
    # region on the ramp, so proofs continue until --clock-warmup-ms have passed since the first one
    extra_warm = 0
    # (not in the sharded mode: there every proof is a collective and the ranks must agree on their number)
    while (time.perf_counter() - tw0) * 1e3 < args.clock_warmup_ms and d <= 17 and not sharded and not args.group:
        run(cds, 2 * S)
        extra_warm += 2 * S
    # The timed region: K proofs between two barriers, MAX over ranks -- repeated until >= --min-timed-s of proving has been
    # timed (a 20-proof region is 0.09 s at 2^20 rows: boxes differ by 10 % on a sample that short, and an external sampler
    # cannot even see it); ms_per_step = the MEDIAN repeat.  Every rank takes the same number of repeats: the decision uses
    # the reduced time.
    tms = []
    rep_s = []
    while True:
        barrier()
        t0 = time.perf_counter()
        proof = run(cds, args.steps, tms)
        barrier()
        rep_s.append(pkg.parallel.max_over_ranks(time.perf_counter() - t0))
        if sum(rep_s) >= args.min_timed_s or len(rep_s) >= 200:
            break
    srt = sorted(rep_s)
    dt = srt[len(srt) // 2] if len(srt) % 2 else 0.5 * (srt[len(srt) // 2 - 1] + srt[len(srt) // 2])
    phase = {}
    for t in tms:
        for k, v in t.items():
            if k.endswith("_ms"):
                phase[k] = phase.get(k, 0.0) + v / len(rep_s)

    if args.timed_only:
        if rank == 0:
            print(json.dumps({"metric": f"proofs/sec at 2^{d + 3} LDE rows, {S} proof(s) in flight per GPU", "value": (args.steps if sharded else world * args.steps) / dt,
                              "unit": "proofs/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
                              "proofs_in_process": args.steps * len(rep_s) + max(args.warmup, S) + extra_warm, "clock_warmup_proofs": extra_warm, "repeats": len(rep_s), "timed_only": True}), flush=True)
        for c_ in cds:
            c_.close()
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- separate passes (not part of `value`) ----
    cd = cds[0]
    # (0) prove latency: exactly one proof on the GPU at a time (same handle, same resident witness)
    if S == 1:
        single_ms = dt / args.steps * 1e3
    else:
        nlat = max(3, min(args.steps, 8))
        run([cd], 1)
        barrier()
        t1 = time.perf_counter()
        run([cd], nlat)
        barrier()
        single_ms = pkg.parallel.max_over_ranks(time.perf_counter() - t1) / nlat * 1e3
    # (a) per-launch HIP events on the library's launch streams -> roofline of the dominant kernel.  Two passes: one proof at a
    # time (`..._lone`: what a latency budget is made of), and the SAME number of proofs in flight as the timed region (what
    # `value` is made of: events are per stream, so a kernel's span includes the time it shares the chip with the other
    # proofs' kernels -- the sum over kernels is <= ms_per_step x in-flight)
    P = max(1, args.profile_steps)

    def profiled(handles, n_proofs):
        for h in handles:
            h.set("profile", 2)   # 2: every launch the library wraps in a scope (1: only those moving >= 32 MB)
        run(handles, n_proofs)
        acc = {}
        for h in handles:
            for k_, v_ in h.kernel_stats().items():
                a = acc.setdefault(k_, {"ms": 0.0, "launches": 0, "bytes": 0.0})
                a["ms"] += v_["ms"]
                a["launches"] += v_["launches"]
                a["bytes"] += v_["bytes"]
            h.set("profile", 0)
        return acc
    stats_lone = profiled([cd], P)
    if S > 1:
        run(cds, S)
        spans = profiled(cds, P * S)
        P_inflight = P * S
        # With S streams kept busy, S kernels share the chip at any moment and an event pair measures a kernel's SPAN, about S times
        # the chip time it costs.  span / S is its share of the GPU time of a proof: those shares sum to ms_per_step (the spans sum to
        # ms_per_step x S), and a roofline at the operating point of `value` is bytes / share.  The raw spans are kept beside them.
        stats = {k_: {"ms": v_["ms"] / S, "launches": v_["launches"], "bytes": v_["bytes"]} for k_, v_ in spans.items()}
    else:
        stats, spans, P_inflight = stats_lone, stats_lone, P
    # (b) the same proofs with the witness in HOST memory: p2gpu_prove, H2D inside the call (N = 1 only: PCIe is shared)
    host = None
    HW = 8 if d <= 17 else 3   # proofs per host-witness measurement (per handle in the in-flight legs)
    if world == 1:
        run([cd], 1, w=wires)
        th = []
        t1 = time.perf_counter()
        run([cd], HW, th, w=wires)
        host_ms = (time.perf_counter() - t1) / HW * 1e3
        host_pipe = None
        if S > 1:  # the same S handles / threads as the timed region, every proof's witness crossing PCIe inside the call
            run(cds, S, w=wires)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            run(cds, HW * S, w=wires)
            torch.cuda.synchronize()
            host_pipe = HW * S / (time.perf_counter() - t1)
        # the same witness in its compact form (p2gpu_prove_sparse): the wires no gate uses are one value each
        # (plonky2's randomize_unused_pi_wires) and never cross PCIe
        sparse = None
        wm = wires.reshape(cd.num_wires, -1)
        nzc = (wm != 0).sum(axis=1)
        ncols = int(np.max(np.nonzero(nzc > 1)[0])) + 1 if (nzc > 1).any() else 0
        rows = {int(np.nonzero(wm[j])[0][0]) for j in range(ncols, cd.num_wires) if nzc[j] == 1}
        if ncols < cd.num_wires and len(rows) <= 1:
            row = rows.pop() if rows else 0
            sw = ("sparse", np.ascontiguousarray(wm[:ncols]).reshape(-1), ncols, row, np.ascontiguousarray(wm[ncols:, row]))
            assert run([cd], 1, w=sw).to_bytes() == proof.to_bytes()
            t1 = time.perf_counter()
            run([cd], HW, w=sw)
            sp_ms = (time.perf_counter() - t1) / HW * 1e3
            sp_pipe = None
            if S > 1:
                run(cds, S, w=sw)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                run(cds, HW * S, w=sw)
                torch.cuda.synchronize()
                sp_pipe = HW * S / (time.perf_counter() - t1)
            sparse = {"entry_point": "p2gpu_prove_sparse (dense columns in host RAM + one value per unused wire)", "dense_columns": ncols,
                      "bytes_over_pcie": int(8 * ncols * wm.shape[1]), "ms_per_proof": sp_ms, "proofs_per_sec": 1e3 / sp_ms,
                      "proofs_per_sec_in_flight": sp_pipe, "in_flight": S, "same_proof_bytes": True}
        # the same call with the wire matrix in page-locked memory (p2gpu_host_alloc): direct DMA, no staging on the thread
        pinned = None
        try:
            wp = pkg.host_array(wires.shape)
            wp[...] = wires
            assert run([cd], 1, w=wp).to_bytes() == proof.to_bytes()
            tp = []
            t1 = time.perf_counter()
            run([cd], HW, tp, w=wp)
            pin_ms = (time.perf_counter() - t1) / HW * 1e3
            pin_pipe = None
            if S > 1:
                run(cds, S, w=wp)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                run(cds, HW * S, w=wp)
                torch.cuda.synchronize()
                pin_pipe = HW * S / (time.perf_counter() - t1)
            pinned = {"entry_point": "p2gpu_prove, wire matrix in p2gpu_host_alloc memory", "ms_per_proof": pin_ms,
                      "proofs_per_sec": 1e3 / pin_ms, "proofs_per_sec_in_flight": pin_pipe, "in_flight": S,
                      "h2d_ms": sum(t["h2d_ms"] for t in tp) / HW, "same_proof_bytes": True}
            pkg.host_free(wp)
        except Exception as e:  # reported, not fatal: the pageable leg above is the contract's number
            pinned = {"error": str(e)}
        host = {"entry_point": "p2gpu_prove (witness in host RAM -> proof bytes in host RAM)", "ms_per_proof": host_ms,
                "pinned": pinned,
                "proofs_per_sec": 1e3 / host_ms, "proofs_per_sec_in_flight": host_pipe, "in_flight": S,
                "h2d_ms": sum(t["h2d_ms"] for t in th) / HW, "sparse": sparse,
                "witness_bytes": int(wires.nbytes), "note": "H2D runs in column chunks on a copy stream, overlapped with the "
                "transforms / leaf hashing of the chunks already on the device; h2d_ms is the copy stream's span"}
    # (c) several proofs in flight on one GPU (throughput mode of a proving service)
    pipe = None
    if world == 1 and not sharded and args.pipelined > 1 and S == 1 and d <= 19:
        extra = [pkg.CircuitData(blob) for _ in range(args.pipelined - 1)]
        hs = [cd] + extra
        run(hs, 2 * len(hs))
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        npipe = max(args.steps, 3 * len(hs))
        run(hs, npipe)
        torch.cuda.synchronize()
        dtp = time.perf_counter() - t1
        pipe = {"in_flight": len(hs), "proofs": npipe, "proofs_per_sec": npipe / dtp, "ms_per_proof": dtp / npipe * 1e3}
        for h in extra:
            h.close()

    # (d) N > 1, replicas run: the other half of the metric in the SAME run -- ONE proof over the N devices.
    #     All ranks prove the seed-1 circuit / witness together (every proof is a collective): RCCL called by the library on
    #     each rank's own stream (backend nccl), or the host-callback transport (gloo: functional only).
    sharded_half = None
    wedged = False   # a collective of the sharded half did not come back: no further collective may be attempted
    if world > 1 and not sharded and args.sharded_steps > 0 and args.workload == "synth":
        # The replicas number is in hand.  The sharded half is collectives inside the library on hardware this code has
        # never been timed on: it runs under a watchdog, so that a transport that does not come back costs this half of the line
        # (an `error` entry), not the whole run -- the JSON line is printed either way and every rank leaves with os._exit.
        box = {}

        def sharded_leg():
          nonlocal cds
          try:
            torch.cuda.set_device(local_rank)   # the current device is per thread
            made1 = pkg.make_circuit(d, mix, seed=1, num_public_inputs=args.public_inputs, hasher=1 if args.hasher == "poseidon" else 0)
            blob1, wires1 = made1[0], made1[1]
            pis1 = made1[2] if args.public_inputs else ()
            for c_ in cds[1:]:
                c_.close()   # their HBM back before another handle is made
            cds = cds[:1]
            csh = pkg.CircuitData(blob1)
            if blocking:
                csh.set("blocking_sync", 1)
            csh.set_shard(rank, world, transport="rccl" if args.backend == "nccl" else None)
            w1 = torch.from_numpy(wires1.view(np.int64)).cuda()
            for _ in range(2):
                pr1 = csh.prove(w1, public_inputs=pis1)
            barrier()
            t1 = time.perf_counter()
            for _ in range(args.sharded_steps):
                pr1 = csh.prove(w1, public_inputs=pis1)
            barrier()
            sh_ms = pkg.parallel.max_over_ranks(time.perf_counter() - t1) / args.sharded_steps * 1e3
            csh.set("profile", 2)
            for _ in range(2):
                csh.prove(w1, public_inputs=pis1)
            xst = exchange_stats(csh.kernel_stats(), 2)
            csh.set("profile", 0)
            barrier()
            sharded_half_ = {"latency_ms_sharded": sh_ms, "latency_ms_sharded_intt": None,
                            "proofs": args.sharded_steps, "rccl_ranks": world if args.backend == "nccl" else 0,
                            "transport": ("RCCL called by the library on each rank's stream (grouped ncclSend/ncclRecv >= 1 MB, ncclAllGather below)"
                                          if args.backend == "nccl" else f"host callback over torch.distributed/{args.backend} (functional check, ranks may share a GPU)"),
                            "exchanges_rank0": xst, "proof_bytes": len(pr1),
                            "witness": "resident on every rank's device (p2gpu_prove_dev)"}
            box["out"] = sharded_half_   # (what is measured so far survives a later leg that raises or hangs)
            # ... and the same proof from ONE process driving all N devices (device group, peer copies): rank 0 starts it while the
            # other ranks wait at the barrier with idle GPUs
            if not args.no_group_probe:
                gp = None
                if rank == 0:
                    import subprocess
                    ndev = torch.cuda.device_count()
                    ids = ",".join(str(i % ndev) for i in range(world))
                    cmd = [sys.executable, os.path.abspath(__file__), "--group", ids, "--group-probe", "--degree-bits", str(d), "--mix", mix,
                           "--public-inputs", str(args.public_inputs), "--hasher", args.hasher, "--sharded-steps", str(args.sharded_steps)]
                    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT",
                                                                            "GROUP_RANK", "LOCAL_WORLD_SIZE", "ROLE_RANK", "TORCHELASTIC_RUN_ID")}
                    try:
                        r = subprocess.run(cmd, capture_output=True, text=True, timeout=120 * scale_, env=env)
                        gp = json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else {"error": (r.stderr or r.stdout)[-400:]}
                    except Exception as e:
                        gp = {"error": str(e)[:300]}
                barrier()
                sharded_half_["group"] = gp
            box["out"] = sharded_half_
            # the same proof with the inverse transforms of the wires / Z-PP column-sharded and the coefficient blocks all-gathered
            # (knob shard_intt) instead of replicated: same bytes, the first multi-GPU lease decides the default.  LAST:
            # grouped send / receive with per-peer sizes has never run between real peers -- if it wedges, everything above is already in `box`
            try:
                csh.set("shard_intt", 1)
                pr2 = csh.prove(w1, public_inputs=pis1)
                barrier()
                t1 = time.perf_counter()
                for _ in range(args.sharded_steps):
                    pr2 = csh.prove(w1, public_inputs=pis1)
                barrier()
                sharded_half_["latency_ms_sharded_intt"] = pkg.parallel.max_over_ranks(time.perf_counter() - t1) / args.sharded_steps * 1e3
                sharded_half_["shard_intt_same_bytes"] = pr2.to_bytes() == pr1.to_bytes()
                csh.set("profile", 2)
                for _ in range(2):
                    csh.prove(w1, public_inputs=pis1)
                sharded_half_["exchanges_rank0_shard_intt"] = exchange_stats(csh.kernel_stats(), 2)
                csh.set("profile", 0)
                # ... and with every step of SURVEY 8(e)'s table sharded: + the permutation argument's chunk quotients by rows (shard_zs),
                # + the FRI batch reduction by columns (shard_reduce)
                csh.set("shard_zs", 1)
                csh.set("shard_reduce", 1)
                pr3 = csh.prove(w1, public_inputs=pis1)
                barrier()
                t1 = time.perf_counter()
                for _ in range(args.sharded_steps):
                    pr3 = csh.prove(w1, public_inputs=pis1)
                barrier()
                sharded_half_["latency_ms_sharded_all_steps"] = pkg.parallel.max_over_ranks(time.perf_counter() - t1) / args.sharded_steps * 1e3
                sharded_half_["all_steps_same_bytes"] = pr3.to_bytes() == pr1.to_bytes()
                for k_ in ("shard_intt", "shard_zs", "shard_reduce"):
                    csh.set(k_, 0)
                barrier()
            except Exception as e:   # (a library error is raised on every rank alike: the ranks stay in step)
                sharded_half_["shard_intt_error"] = repr(e)[:300]
            csh.close()
            del w1
            torch.cuda.empty_cache()
            box["out"] = sharded_half_
          except Exception as e:   # (the other ranks then wait in a collective until their own watchdog fires)
            box["exc"] = repr(e)

        import threading
        scale_ = 1 << max(0, d - 17)
        deadline = 300.0 * scale_
        th_ = threading.Thread(target=sharded_leg, daemon=True)
        th_.start()
        th_.join(deadline)
        if th_.is_alive() or "exc" in box or "out" not in box:
            wedged = th_.is_alive()
            sharded_half = dict(box.get("out") or {"latency_ms_sharded": None, "rccl_ranks": world if args.backend == "nccl" else 0})
            sharded_half["error"] = (f"the sharded half did not finish within {deadline:.0f} s (a collective is stuck): what had been measured until then is kept" if wedged
                                     else "the sharded half raised: " + str(box.get("exc", "?"))[:300])
        else:
            sharded_half = box["out"]

    # (e) N = 1: what ONE invocation of a CLI-shaped caller sees (the reference's `prove` re-translates, re-builds and proves
    #     once per process, prove_action.rs:27-43): fresh process, circuit create, first prove
    cold = None
    if world == 1 and not group and not args.no_cold_process and d <= 19:
        try:
            cold = cold_process(pkg, blob, wires, pis)
        except Exception as e:
            cold = {"error": str(e)[:300]}

    if rank == 0:
        total_proofs = args.steps if sharded else world * args.steps
        hdr = blob[:256].view(np.uint32)
        W, CS = int(hdr[3]), int(hdr[5]) + int(hdr[4])
        steps_b, total_b = survey_bytes(d, W, CS)
        ms_step = dt / args.steps * 1e3
        # The roofline is that of ONE kernel, and every figure in it divides by that kernel's LONE launch time (one proof on the GPU
        # at a time: per-launch HIP events on the library's launch stream, what `rocprofv3 --kernel-trace --stats` of scratch/prof.sh
        # agrees with).  With S proofs in flight an event pair measures a span the kernel shares with the other proofs' kernels;
        # span / S is an attribution, not a duration (it printed an HBM fraction of 1.124), so the in-flight
        # spans are reported in the detail file as spans and nothing is divided by them.  The dominant kernel is the one with
        # the largest lone time per proof (stable from run to run; the in-flight shares flipped between two kernels).
        name, st = max(stats_lone.items(), key=lambda kv: kv[1]["ms"])
        avg_ms = st["ms"] / st["launches"]
        launches_per_proof = st["launches"] / P
        step = step_of(name)
        # SURVEY 8(d) bytes of the kernel's step, shared by every kernel symbol of that step in proportion to its (lone) time
        step_ms = sum(v["ms"] for k, v in stats_lone.items() if step_of(k) == step) or st["ms"]
        alg_per_launch = (steps_b.get(step, 0.0) * (st["ms"] / step_ms)) / launches_per_proof if step else st["bytes"] / st["launches"]
        impl_per_launch = st["bytes"] / st["launches"]
        gbps_alg = alg_per_launch / (avg_ms * 1e-3) / 1e9
        gbps_impl = impl_per_launch / (avg_ms * 1e-3) / 1e9
        traffic, traffic_src = pmc_traffic(name)
        issue = issue_roofline(name, 1e3 / avg_ms)
        # columns of the witness that are not "zero outside one row": what the transforms and the leaf hash really touch
        wm_ = wires.reshape(W, -1)
        dense_w = int(((wm_ != 0).sum(axis=1) > 1).sum())
        proc_b = processed_bytes(d, W, CS, dense_w)
        # every profiled kernel: LONE time, algorithmic bytes (what the launcher counts for the columns it was given) and, from the
        # newest committed PMC summary, measured HBM traffic and its ratio to the algorithmic bytes; the in-flight span beside it
        per_kernel = {}
        for k_, v_ in sorted(stats_lone.items(), key=lambda kv: -kv[1]["ms"]):
            tr_, _ = pmc_traffic(k_)
            alg_ = v_["bytes"] / v_["launches"]
            iss_ = issue_roofline(k_, v_["launches"] / (v_["ms"] * 1e-3)) if v_["ms"] > 0 else None
            per_kernel[k_] = {"ms_per_proof_lone": round(v_["ms"] / P, 4), "launches_per_proof": v_["launches"] / P,
                              "avg_launch_ms_lone": v_["ms"] / v_["launches"],
                              "span_ms_per_proof_in_flight": round(spans[k_]["ms"] / P_inflight, 4) if k_ in spans else None,
                              "algorithmic_bytes_per_launch": alg_, "traffic_bytes_per_launch": tr_,
                              "traffic_over_algorithmic": (tr_ / alg_) if (tr_ and alg_) else None,
                              "hbm_frac": alg_ / (v_["ms"] / v_["launches"] * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                              "valu_lane_instr_per_s": iss_["achieved"] if iss_ else None,
                              "valu_frac_of_peak": iss_["frac"] if iss_ else None,
                              "valu_frac_of_mix_ceiling": iss_["frac_of_mix_ceiling"] if iss_ else None}
        out = {
            "metric": (f"prove latency (ms) + proofs/sec at 2^{d + 3} LDE rows, {world} GPU(s): value = proofs/sec, witness resident in HBM, {S} proof(s) in "
                       f"flight per GPU; latency_ms_single_proof = lone resident proof; *_host_witness = p2gpu_prove from host RAM (PCIe inside the call)"
                       + ("; latency_ms_sharded[_group] = ONE proof over the N GPUs (RCCL ranks / one-process device group)" if world > 1 else "")),
            "value": total_proofs / dt,
            "unit": "proofs/sec",
            "n_gpus": world,
            "ranks_share_devices": bool(world > 1 and args.backend != "nccl"),   # gloo test mode: functional run, not a scaling point
            "steps": args.steps,
            "warmup": args.warmup,
            "clock_warmup_proofs": extra_warm,   # untimed proofs beyond W (--clock-warmup-ms): steady-state clocks in the timed region
            "ms_per_step": ms_step,
            "ms_per_step_min": min(rep_s) / args.steps * 1e3,
            "ms_per_step_max": max(rep_s) / args.steps * 1e3,
            "repeats": len(rep_s),
            "timed_seconds": sum(rep_s),
            "higher_is_better": True,
            "scaling": "strong" if sharded else "weak",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {
                "workload": (f"synth(d={d},{mix})" if args.workload == "synth" else f"SHA-256 of a {args.sha_blocks}-block message: {args.sha_blocks} "
                             f"chained Sha256Compression opcodes through the restated reference translator (translate.py)") +
                            f": {1 << d} gates -> 2^{d + 3} LDE rows, {W} wires / 80 routed, "
                            f"{'Poseidon' if args.hasher == 'poseidon' else 'Keccak'}GoldilocksConfig, rate 8, cap 2^4, 28 queries, 16 PoW bits (BASELINE configs[2] shape)",
                "degree_bits": d, "lde_rows": 1 << (d + 3), "mix": mix, "public_inputs": args.public_inputs,
                "parallelism": (f"ONE process, device group {group}: each proof coset-sharded over {len(group)} ranks (8/{len(group)} LDE cosets each), "
                                f"peer-to-peer exchanges between the ranks' streams" if group else
                                f"one proof coset-sharded over {world} GPUs (8/{world} LDE cosets each; RCCL all-gather of caps, "
                                f"quotient interpolants, query openings)" if sharded else
                                f"replicas x{world} (independent proofs per GPU, no data-path collective), "
                                f"{S} proof(s) in flight per GPU"),
                "proof_bytes": len(proof),
                "witness": "resident in HBM when the timed region starts (p2gpu_prove_dev); proof bytes returned to host",
            },
            "roofline": {
                "kernel": name,
                # The contract's form: ALGORITHMIC bytes per launch / the kernel's average LONE launch duration / 8 TB/s.  The kernel
                # is in fact bound by VALU issue (counters + microbenchmarks): `valu_issue` carries that side
                "bound": "hbm",
                "achieved": gbps_alg,
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": gbps_alg / HBM_PEAK_GBPS,
                "traffic": traffic,
                "traffic_source": traffic_src,
                "avg_launch_ms": avg_ms,
                "launches_per_proof": launches_per_proof,
                "timing": f"per-launch HIP events on the library's launch stream, one proof on the GPU at a time, {P} proofs after the timed region",
                "algorithmic_bytes_per_launch": alg_per_launch,
                "algorithmic_bytes_definition": f"SURVEY.md 8(d) bytes of prover step '{step}' ({steps_b.get(step, 0) / 1e9:.3f} GB per proof) "
                                                f"/ launches of that step per proof",
                "implementation_bytes_per_launch": impl_per_launch,
                "frac_traffic": (traffic / (avg_ms * 1e-3) / 1e9 if traffic else gbps_impl) / HBM_PEAK_GBPS,
                "valu_issue": issue,
                "whole_proof": {"algorithmic_bytes": total_b, "achieved": total_b / (single_ms * 1e-3) / 1e9, "unit": "GB/s",
                                "frac": total_b / (single_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                "note": "B(N) / t_prove of SURVEY.md 8(d) with t_prove = latency_ms_single_proof",
                                "frac_at_throughput": total_b * (total_proofs / world / dt) / 1e9 / HBM_PEAK_GBPS,
                                "valu_floor": valu_floor(ms_step * (1 if sharded else world))},
                # every prover step by SURVEY 8(d)'s bytes over the summed LONE HIP-event time of its kernels (per proof)
                # `frac` divides the bytes of the columns the step PROCESSED (dense wire columns: `dense_wire_columns`);
                # `frac_incl_elided` is SURVEY 8(d)'s figure for all 270 columns over the same time
                "steps": {stp: {"algorithmic_bytes": steps_b[stp], "processed_bytes": proc_b.get(stp, steps_b[stp]), "kernel_ms": ms_,
                                "achieved_GBps": proc_b.get(stp, steps_b[stp]) / (ms_ * 1e-3) / 1e9,
                                "frac": proc_b.get(stp, steps_b[stp]) / (ms_ * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                "frac_incl_elided": steps_b[stp] / (ms_ * 1e-3) / 1e9 / HBM_PEAK_GBPS}
                          for stp in steps_b
                          for ms_ in [sum(v["ms"] for k, v in stats_lone.items() if step_of(k) == stp) / P] if ms_ > 0},
                "dense_wire_columns": dense_w,
                "kernels": per_kernel,
                "counter_summaries": {"pmc": os.path.basename(newest("r*_pmc_summary.json") or "") or None,
                                      "sq": os.path.basename(newest("r*_sq_summary.json") or "") or None,
                                      "keyed_by": f"(mix, degree_bits) = {bs.PROFILE_KEY}: null fields where this workload has no committed counter pass"},
            },
            "latency_ms_single_proof": single_ms,
            "latency_ms_sharded": sharded_half["latency_ms_sharded"] if sharded_half else None,
            "latency_ms_sharded_group": (sharded_half.get("group") or {}).get("latency_ms_sharded_group") if sharded_half else None,
            "latency_ms_sharded_intt": sharded_half.get("latency_ms_sharded_intt") if sharded_half else None,
            "latency_ms_sharded_group_intt": (sharded_half.get("group") or {}).get("latency_ms_sharded_group_intt") if sharded_half else None,
            "latency_ms_sharded_all_steps": sharded_half.get("latency_ms_sharded_all_steps") if sharded_half else None,
            "latency_ms_sharded_group_all_steps": (sharded_half.get("group") or {}).get("latency_ms_sharded_group_all_steps") if sharded_half else None,
            "rccl_ranks": sharded_half["rccl_ranks"] if sharded_half else (world if sharded and args.backend == "nccl" else 0),
            "peer_access": (sharded_half.get("group") or {}).get("peer_access") if sharded_half else (pkg.peer_access() if group else None),
            "sharded": sharded_half,
            "cold_process_ms": cold.get("cold_process_ms") if cold else None,
            "cold_process": cold,
            # SURVEY 8(d)'s boundary-to-boundary prove (p2gpu_prove: witness in host RAM -> proof bytes in host RAM, H2D inside
            # the call).  Reported beside `value`; the measurement contract keeps `value` on the resident-witness entry
            "latency_ms_single_proof_host_witness": host["ms_per_proof"] if host else None,
            "value_host_witness": host["proofs_per_sec_in_flight"] or host["proofs_per_sec"] if host else None,
            "single_proof": {"in_flight": 1, "ms_per_proof": single_ms, "proofs_per_sec": (1 if sharded else world) * 1e3 / single_ms,
                             "note": "one proof on the GPU at a time (the round-1 headline configuration)"},
            "in_flight_per_gpu": S,
            "resident_bytes_per_handle": handle_bytes,
            "blocking_sync": bool(blocking),
            "host_witness": host,
            "pipelined": pipe,
            "phase_ms": {k: v / args.steps for k, v in sorted(phase.items())},
            "kernel_ms_per_proof_lone": {k: round(v["ms"] / P, 4) for k, v in sorted(stats_lone.items(), key=lambda kv: -kv[1]["ms"])},
            "kernel_span_ms_per_proof": {k: round(v["ms"] / P_inflight, 4) for k, v in sorted(spans.items(), key=lambda kv: -kv[1]["ms"])},
            "kernel_ms_sum": {"lone": round(sum(v["ms"] for v in stats_lone.values()) / P, 4),
                              "spans_in_flight": round(sum(v["ms"] for v in spans.values()) / P_inflight, 4),
                              "ms_per_step_per_gpu": ms_step * (1 if sharded else world),
                              "note": f"HIP events are per stream.  kernel_ms_per_proof_lone = one proof at a time (the operating point of latency_ms_single_proof; "
                                      f"every roofline figure divides by these).  kernel_span_ms_per_proof = event spans with {S} proof(s) in flight: a span includes "
                                      f"the time the kernel shares the chip with the other proofs' kernels (the spans sum to <= ms_per_step x {S} per GPU); "
                                      f"spans are not durations and nothing is divided by them"},
            "kernel_profile": f"{P} extra proofs one at a time (..._lone, roofline) and {P_inflight} with {S} in flight (kernel_span_ms_per_proof) "
                              f"after the timed region, per-launch HIP events on the library's launch streams (profile = 2: every scoped launch)",
            "device": pkg.device_info()["name"],
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(pkg, d, mix if args.workload == "synth" else "sha", args.public_inputs, args.cpu_single_thread_bits)
        # stdout carries ONE short line (the driver keeps 8 KB of stdout: round 4's 21 KB line came back unparsed); everything
        # else goes to the detail file beside it
        detail_path = args.detail or os.path.join(ROOT, "bench_detail.json")
        try:
            with open(detail_path, "w") as fh:
                json.dump(out, fh, indent=1)
        except OSError as e:
            print(f"bench.py: could not write {detail_path}: {e}", file=sys.stderr)
            detail_path = None
        print(compact_line(out, os.path.basename(detail_path) if detail_path else None), flush=True)
    if wedged:   # a collective never came back: the line is out, leave without touching the process group again
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)
    for c_ in cds:
        c_.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
