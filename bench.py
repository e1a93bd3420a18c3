#!/usr/bin/env python3
"""bench.py -- proofs/sec and prove latency of the MI355X `prove` hot path.

One "step" = one proof of the BASELINE.json workload (configs[2]: SHA256-like circuit,
2^17 gates = 2^20 LDE rows, wide_ecc_config shape), witness matrix resident in HBM, proof
bytes back on the host (the boundary of p2gpu_prove_dev).  N > 1: one process per GPU
(torch.distributed / RCCL), every rank proves its own independent proofs (replicas, weak
scaling, no data-path collective); time = max over ranks between two barriers.

Prints ONE JSON line on rank 0 (see the round contract): metric/value/unit, ms_per_step,
`roofline` for the dominant kernel (HIP-event time measured inside the library on its launch
stream over the timed region, algorithmic bytes from DESIGN.md) and `cpu_baseline` (the
oracle = CPU port, timed on this host's cores on a bounded sample; N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the newest committed rocprofv3 PMC summary
    (profiles/rNN*_pmc_summary.json: separate --pmc FETCH_SIZE / WRITE_SIZE passes of this same
    command, gfx950 read-side x2 correction).  None when no summary is committed."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")))
    if not files:
        return None, None
    try:
        with open(files[-1]) as f:
            k = json.load(f)["kernels"].get(kernel)
        return (k["hbm_bytes_per_launch"] if k else None), os.path.basename(files[-1])
    except Exception:
        return None, None


VALU_PEAK = 256 * 4 * 16 * 2.4e9  # lane-instructions/s: 256 CUs x 4 SIMDs x 16 lanes/cycle x 2.4 GHz (MI355X_MICROARCH.md)


def issue_roofline(kernel, achieved_gbps):
    """What actually bounds the integer kernels: VALU lane-instructions per HBM byte of
    `kernel` from the committed rocprofv3 passes (SQ_INSTS_VALU x 64 lanes / (FETCH + WRITE bytes), same
    command), times the live byte rate, against the chip's VALU issue peak.  None without a summary."""
    import glob
    sq = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_sq_summary.json")))
    pm = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")))
    if not sq or not pm:
        return None
    try:
        with open(sq[-1]) as f:
            insts = json.load(f)["kernels"][kernel]["SQ_INSTS_VALU"] * 64.0
        with open(pm[-1]) as f:
            nbytes = json.load(f)["kernels"][kernel]["hbm_bytes_per_launch"]
        per_byte = insts / nbytes
        rate = per_byte * achieved_gbps * 1e9
        return {"bound": "valu-issue", "lane_instr_per_byte": per_byte, "achieved": rate, "peak": VALU_PEAK,
                "unit": "lane-instr/s", "frac": rate / VALU_PEAK, "source": os.path.basename(sq[-1])}
    except Exception:
        return None


def cpu_baseline(pkg, d_sample, d_target, mix):
    """Oracle (CPU port of the same path) on a bounded sample: one proof at 2^d_sample gates,
    scaled linearly to the target size (NTT log factor ignored -> favours the CPU)."""
    orc = entry.load_oracle()
    cores = orc.lib().orc_num_threads()
    blob, wires = pkg.make_circuit(d_sample, mix, seed=1)
    oc = orc.OracleCircuit(blob)          # circuit precompute is outside the timed call, as on the GPU
    t0 = time.perf_counter()
    proof, tr = oc.prove(wires)
    dt = time.perf_counter() - t0
    scale = float(1 << (d_target - d_sample))
    return {
        "value": 1.0 / (dt * scale),
        "unit": "proofs/sec",
        "cores": int(cores),
        "kind": "port",
        "sample": f"1 proof of synth(d={d_sample},{mix}) = 2^{d_sample + 3} LDE rows in {dt:.2f} s on {cores} OpenMP threads, "
                  f"scaled x{int(scale)} (linear in rows) to 2^{d_target + 3} LDE rows; oracle/ C restatement, not upstream plonky2",
        "seconds_sample": dt,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--degree-bits", type=int, default=17)
    ap.add_argument("--mix", default="sha")
    ap.add_argument("--public-inputs", type=int, default=0,
                    help="public inputs of the synthetic circuit (> 0 adds PoseidonGate rows to the circuit and the "
                         "PoseidonGate to the gate set every LDE row evaluates); 0 = the BASELINE parity shape")
    ap.add_argument("--in-flight", type=int, default=1,
                    help="independent proofs in flight per GPU (separate circuit handles / HIP streams, one host "
                         "thread each): hides the latency-bound Merkle-tree tails and host round trips of one proof "
                         "behind the kernels of another; 1 = strictly one proof at a time")
    ap.add_argument("--mode", choices=["replicas", "sharded"], default="replicas",
                    help="N > 1: `replicas` = independent proofs per GPU (throughput, weak scaling, default); "
                         "`sharded` = ONE proof coset-sharded over the N GPUs (latency, strong scaling; RCCL "
                         "all-gathers of caps / quotient interpolants / query openings)")
    ap.add_argument("--backend", default=os.environ.get("P2GPU_BENCH_BACKEND", "nccl"),
                    help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to exercise the "
                         "multi-rank flow on a box with fewer GPUs than ranks)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-bits", type=int, default=14)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
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
    dev = (ctypes.c_int * 1)(local_rank)
    assert lib.p2gpu_init(dev, 1) == 0, lib.p2gpu_last_error()

    d, mix = args.degree_bits, args.mix
    # every rank proves its own witness of the same circuit shape (independent proofs)
    sharded = args.mode == "sharded" and world > 1
    made = pkg.make_circuit(d, mix, seed=1 if sharded else 1 + rank, num_public_inputs=args.public_inputs)
    blob, wires = made[0], made[1]
    pis = made[2] if args.public_inputs else ()
    S = 1 if sharded else max(1, min(args.in_flight, args.steps))
    cds = [pkg.CircuitData(blob) for _ in range(S)]
    if sharded:
        cds[0].set_shard(rank, world)
    wires_dev = torch.from_numpy(wires.view(np.int64)).cuda()
    torch.cuda.synchronize()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    import threading

    def run(n_proofs, collect=None):
        """n_proofs proofs over the S handles (thread i proves every S-th proof)."""
        def work(i):
            last = None
            for _ in range(i, n_proofs, S):
                last = cds[i].prove(wires_dev, public_inputs=pis)
                if collect is not None:
                    collect.append(last.timings)
            results[i] = last
        results = [None] * S
        th = [threading.Thread(target=work, args=(i,)) for i in range(S)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        return next(r for r in results if r is not None)

    proof = run(max(args.warmup, S))
    for cd in cds:
        cd.set("profile", 1)  # per-launch HIP events on each handle's stream, over the timed region
    barrier()
    t0 = time.perf_counter()
    tms = []
    proof = run(args.steps, tms)
    barrier()
    dt = time.perf_counter() - t0
    dt = pkg.parallel.max_over_ranks(dt)
    stats = {}
    for cd in cds:
        for k, v in cd.kernel_stats().items():
            a = stats.setdefault(k, {"ms": 0.0, "bytes": 0.0, "launches": 0})
            for f in a:
                a[f] += v[f]
        cd.set("profile", 0)
    phase = {}
    for t in tms:
        for k, v in t.items():
            if k.endswith("_ms"):
                phase[k] = phase.get(k, 0.0) + v
    # single-proof latency (nothing else in flight), measured after the timed region
    cds[0].prove(wires_dev, public_inputs=pis)
    torch.cuda.synchronize()
    tl = time.perf_counter()
    for _ in range(3):
        cds[0].prove(wires_dev, public_inputs=pis)
    latency_ms = (time.perf_counter() - tl) / 3 * 1e3
    cd = cds[0]

    if rank == 0:
        total_proofs = args.steps if sharded else world * args.steps
        name, st = max(stats.items(), key=lambda kv: kv[1]["ms"])
        avg_ms = st["ms"] / st["launches"]
        gbps = (st["bytes"] / st["launches"]) / (avg_ms * 1e-3) / 1e9
        traffic, traffic_src = pmc_traffic(name)
        out = {
            "metric": f"proofs/sec at 2^{d + 3} LDE rows (prove latency = ms_per_step)",
            "value": total_proofs / dt,
            "unit": "proofs/sec",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if sharded else "weak",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {
                "workload": f"synth(d={d},{mix}): {1 << d} gates -> 2^{d + 3} LDE rows, 234 wires / 80 routed, "
                            f"KeccakGoldilocksConfig, rate 8, cap 2^4, 28 queries, 16 PoW bits (BASELINE configs[2] shape)",
                "degree_bits": d, "lde_rows": 1 << (d + 3), "mix": mix, "public_inputs": args.public_inputs,
                "parallelism": (f"one proof coset-sharded over {world} GPUs (8/{world} LDE cosets each; all-gather of caps, "
                                f"quotient interpolants, query openings)" if sharded else
                                f"replicas x{world} (independent proofs per GPU, no data-path collective), "
                                f"{S} proofs in flight per GPU"),
                "proof_bytes": len(proof),
                "witness": "resident in HBM (p2gpu_prove_dev); proof bytes returned to host",
            },
            "roofline": {
                "kernel": name,
                "bound": "hbm",
                "achieved": gbps,
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": gbps / HBM_PEAK_GBPS,
                "traffic": traffic,
                "traffic_source": traffic_src,
                "avg_launch_ms": avg_ms,
                "launches_per_proof": st["launches"] / args.steps,
                "algorithmic_bytes_per_launch": st["bytes"] / st["launches"],
                "issue": issue_roofline(name, gbps),
            },
            "latency_ms_single_proof": latency_ms,
            "in_flight_per_gpu": S,
            "phase_ms": {k: v / args.steps for k, v in sorted(phase.items())},
            "kernel_ms_per_proof": {k: round(v["ms"] / args.steps, 4) for k, v in sorted(stats.items(), key=lambda kv: -kv[1]["ms"])},
            "device": pkg.device_info()["name"],
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(pkg, min(args.cpu_sample_bits, d), d, mix)
        print(json.dumps(out), flush=True)
    for c_ in cds:
        c_.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
