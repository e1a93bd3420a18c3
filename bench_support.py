"""bench_support.py -- what bench.py's measurement rests on, apart from the run itself: SURVEY 8(d)'s byte counts per prover step, the look-up
of the committed counter summaries (profiles/rNN_<mix><d>_{pmc,sq}_summary.json) behind `roofline.traffic` / `valu_issue` / `valu_floor`, the
ONE stdout line (`compact_line`), the rank launcher of
`--gpus N`, the device-group probe and the cold-process measurement.  bench.py keeps the run: warm-up, timed region, passes, result."""
import glob
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

HBM_PEAK_GBPS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md; 6.29 TB/s is what a float4 copy reaches)
# VALU issue ceilings (profiles/r03_ubench.txt: counter-clocked >= 16 ms microbenchmarks, scratch/ubench/gen_issue.py, gen_bank.py).
# The guide's 2 cycles per wave64 instruction hold -- 78.6 T lane-instr/s at 2.4 GHz, 71 T measured (2.2 cycles at 2.38 GHz) --
# for a SUBSET of opcodes (v_add/sub_u32, v_and/or/xor_b32, v_lshrrev_b32, v_mov_b32, v_bitop3_b32, v_fma/mul_f32); everything
# else (carry chains, compares, v_cndmask, every multiply, v_alignbit_b32, all other VOP3 integer ops, 64-bit ops, SGPR sources)
# retires at half that: 39.3 T nominal, 37.7 T measured.  `roofline.issue.peak` is the guide's figure; `mix_ceiling` is what the
# kernel's own instruction mix allows (profiles/r03_isa_mix.json; for the Keccak kernels the round's 120 v_bitop3 : 58 v_alignbit
# measured as an interleaved stream).
VALU_PEAK = 78.6e12
VALU_FULL_MEASURED, VALU_HALF_MEASURED = 71.0e12, 37.7e12
KECCAK_MIX_CEILING = 49.9e12   # "bitop3, bitop3, alignbit" stream at 8 waves per SIMD: 3.13 cycles per instruction (48.2 T at 4 waves)


def survey_bytes(d, W, CS, K=2, PP=9, QF=8):
    """ALGORITHMIC bytes of one proof per prover step, n = 2^d gates, N = 8n LDE rows.
    Returns ({step: bytes}, total).  Steps are named after the kernel that does the bulk of them."""
    n, N = 1 << d, 8 << d
    zp, q = K * (1 + PP), K * QF
    cols = W + zp + q                                 # columns committed per proof (wires, Z/PP, quotient chunks)
    steps = {
        "intt": (W + zp) * n * 16 + 2 * N * 16,       # values -> coefficients (read + write); quotient coset-iNTT
        "lde": cols * (n * 8 + N * 8),                # coefficients in, 8x LDE values out
        "leaf_hash": cols * N * 8 + 3 * 32 * N,       # LDE rows in, leaf digests out
        "merkle": 3 * 32 * N,                         # digests in, inner nodes out
        "zs": 2 * 80 * n * 8 + zp * n * 8,            # wires + sigmas in, Z/PP out
        "quotient": (W + CS + zp) * N * 8 + 2 * N * 8,
        "openings_fri_reduce": (CS + cols) * n * 8,   # one pass over every coefficient (openings + batch reduce)
        "fri": int(0.01 * 8660 * N),
    }
    return steps, sum(steps.values())


# kernel symbol (prefix) -> prover step of survey_bytes()
KERNEL_STEP = [
    ("ntt_pass_kernel<1,", "lde"), ("ntt_dit_", "lde"), ("lde_", "lde"), ("ntt_pass_kernel<0,", "intt"), ("ntt_dif_", "intt"), ("intt_", "intt"),
    ("hash_lde", "leaf_hash"), ("merkle", "merkle"), ("zs_", "zs"), ("quotient", "quotient"), ("poseidon_gate", "quotient"), ("gate_sums", "quotient"),
    ("eval_columns", "openings_fri_reduce"), ("reduce_columns", "openings_fri_reduce"),
    ("structured_fill", "lde"), ("column_nonzero", "intt"),   # the structured columns' share of those steps
]


def processed_bytes(d, W, CS, dense_w, K=2, PP=9, QF=8):
    """Bytes of the columns a step actually PROCESSED: structured wire columns (zero outside the
    PublicInputGate row) are not transformed, their LDE is not stored and the leaf hash recomputes them from a
    scalar -- survey_bytes() counts them (SURVEY 8(d) is about the reference's algorithm), this does not."""
    n, N = 1 << d, 8 << d
    zp, q = K * (1 + PP), K * QF
    cols = dense_w + zp + q
    return {
        "intt": (dense_w + zp) * n * 16 + W * n * 8 + 2 * N * 16,   # + the classification pass over the whole witness
        "lde": cols * (n * 8 + N * 8),
        "leaf_hash": cols * N * 8 + 3 * 32 * N,
    }


def step_of(kernel):
    k = kernel.replace(" ", "")
    for pre, step in KERNEL_STEP:
        if k.startswith(pre.replace(" ", "")):
            return step
    return None


PROFILE_KEY = ("sha", 17)   # (mix, degree_bits) of the workload this run benches: selects the committed counter summaries


def newest(pattern):
    """Newest committed counter summary of THIS workload: profiles/rNN[x]_<mix><degree_bits>_<kind>.json (round 4 on), e.g.
    r04_sha17_pmc_summary.json, r04_ecdsa19_sq_summary.json.  Rounds 1-3 profiled 2^20 rows only and named the files
    rNN_<kind> (sha) / rNN_ecdsa_<kind>: accepted for exactly those two workloads.  None when this (mix, degree_bits) was
    never profiled -- the fields that need counters are then null instead of borrowed from another size."""
    import re
    kind = pattern.split("*_", 1)[1]
    mix, d = PROFILE_KEY
    rx = [re.compile(r"^r\d+[a-z]?_" + re.escape(f"{mix}{d}_{kind}") + "$")]
    if d == 17 and mix in ("sha", "ecdsa"):
        rx.append(re.compile(r"^r\d+[a-z]?_" + re.escape(("ecdsa_" if mix == "ecdsa" else "") + kind) + "$"))
    for r in rx:
        files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", pattern)) if r.match(os.path.basename(f)))
        if files:
            return files[-1]
    return None


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the newest committed rocprofv3 PMC summary
    (profiles/rNN*_pmc_summary.json: separate --pmc FETCH_SIZE / WRITE_SIZE passes of this same
    command, gfx950 read-side x2 correction).  None when the summary does not hold the kernel."""
    f = newest("r*_pmc_summary.json")
    if not f:
        return None, None
    try:
        with open(f) as fh:
            k = json.load(fh)["kernels"].get(kernel)
        return (k["hbm_bytes_per_launch"] if k else None), os.path.basename(f)
    except Exception:
        return None, None


def mix_ceiling(kernel):
    """Ceiling of the kernel's own VALU mix in lane-instr/s (see VALU_PEAK above) and where it comes from."""
    if kernel.startswith(("hash_lde", "hash_fri", "merkle", "pow_kernel")) and "<1" not in kernel.split(",")[0]:
        return KECCAK_MIX_CEILING, "Keccak-f round = 120 v_bitop3_b32 (full rate) + 58 v_alignbit_b32 (half rate): measured as an interleaved stream, profiles/r03_ubench.txt"
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_isa_mix.json")))   # static: one file for every workload
    f = files[-1] if files else None
    if f:
        try:
            with open(f) as fh:
                k = json.load(fh)["kernels"].get(kernel)
            if k:
                return k["mix_ceiling_lane_instr_per_s"], f"{os.path.basename(f)}: {k['full_rate']} full-rate + {k['half_rate']} half-rate VALU (static)"
        except Exception:
            pass
    return VALU_HALF_MEASURED, "half-rate class (field arithmetic: carry chains + v_mad_u64_u32)"


def counter_clock(kernel, wave_instr_per_launch, launches_per_sec):
    """Cycle-level view of a kernel from the committed counter pass (profiles/rNN_clock.txt = scratch/clock_pmc.sh: GRBM_GUI_ACTIVE
    per dispatch): cycles per VALU wave-instruction per SIMD -- what the issue rules of profiles/r03_ubench.txt bound, whatever
    the clock -- the shader clock that pass saw, and the clock THIS run's kernel time implies at the same cycles per instruction.
    The chip lowers its clock under these kernels (1.7-2.0 GHz in the Keccak streams, 2.1-2.3 in the field arithmetic): the gap
    between frac_of_mix_ceiling (instructions per SECOND) and 1 is mostly that, not stalls."""
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_clock.txt")))
    if not files:
        return {}
    if PROFILE_KEY[1] != 17 or PROFILE_KEY[0] not in ("sha", "ecdsa"):
        return {}   # the clock pass was taken at 2^20 rows only
    section = PROFILE_KEY[0]
    cur = None
    for line in open(files[-1]):
        if line.startswith("=="):
            cur = "ecdsa" if "ecdsa" in line else "sha"
        m = re.match(r"^(.*?)\s+launches\s+\d+ avg\s+([\d.]+) us\s+clock ([\d.]+) GHz\s+cycles/VALU wave-instr/SIMD ([\d.]+)", line)
        if m and cur == section and m.group(1).strip() == kernel:
            if float(m.group(2)) < 100.0:
                # GRBM_GUI_ACTIVE / 8 / duration is not a clock for a dispatch this short (round 4 read 2.5-3.3 GHz on a 2.4 GHz
                # part below ~100 us: the counter window is wider than the kernel) -- no cycle figures for such kernels
                return {}
            cpi = float(m.group(4))
            return {"cycles_per_valu_wave_instr": cpi, "clock_ghz_counter_pass": float(m.group(3)),
                    "clock_ghz_implied_this_run": cpi * wave_instr_per_launch / 1024.0 * launches_per_sec / 1e9,
                    "cycles_source": os.path.basename(files[-1]) + " (GRBM_GUI_ACTIVE / 8 XCDs per dispatch, SQ_INSTS_VALU / 1024 SIMDs)",
                    "stream_cycles_per_instr_ubench": 3.16 if "hash_lde" in kernel or "merkle" in kernel else None}
    return {}


def valu_floor(ms_per_proof_per_gpu):
    """What the proof is really bound by: its VALU instructions.  Lane-instructions per proof by class from the committed
    SQ pass of THIS workload and the time they need at the ceilings of their mixes (profiles/numbers.py budget_of: Keccak-f stream
    49.9 T lane-instr/s, field arithmetic 37.7 T) -- the floor of one proof on one GPU whatever the overlap -- next to the
    measured time per proof at `value`.  None when this (mix, degree_bits) has no committed SQ pass."""
    f = newest("r*_sq_summary.json")
    if not f:
        return None
    try:
        import importlib.util
        spec = importlib.util.spec_from_file_location("p2_numbers", os.path.join(ROOT, "profiles", "numbers.py"))
        nb = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(nb)
        with open(f) as fh:
            tot, floor_ms = nb.budget_of(json.load(fh))
        if tot is None:
            return None
        return {"lane_instr_per_proof": sum(tot.values()), "floor_ms_per_proof": floor_ms, "measured_ms_per_proof_at_value": ms_per_proof_per_gpu,
                "frac_of_floor": floor_ms / ms_per_proof_per_gpu if ms_per_proof_per_gpu else None, "source": os.path.basename(f)}
    except Exception:
        return None


def issue_roofline(kernel, launches_per_sec):
    """VALU issue side of a kernel: lane-instructions per launch from the committed SQ pass (SQ_INSTS_VALU x 64) x live
    launches/s of kernel time, against the guide's peak and against the ceiling of the kernel's own instruction mix."""
    f = newest("r*_sq_summary.json")
    if not f:
        return None
    try:
        with open(f) as fh:
            insts = json.load(fh)["kernels"][kernel]["SQ_INSTS_VALU"] * 64.0
        rate = insts * launches_per_sec
        ceil, why = mix_ceiling(kernel)
        out = {"bound": "valu-issue", "lane_instr_per_launch": insts, "achieved": rate, "peak": VALU_PEAK,
               "unit": "lane-instr/s", "frac": rate / VALU_PEAK, "source": os.path.basename(f),
               "peak_source": "MI355X_MICROARCH.md (2 cycles per wave64 VALU at 2.4 GHz), reproduced for the full-rate opcodes in profiles/r03_ubench.txt",
               "mix_ceiling": ceil, "frac_of_mix_ceiling": rate / ceil, "mix_ceiling_source": why}
        out.update(counter_clock(kernel, insts / 64.0, launches_per_sec))
        return out
    except Exception:
        return None


def _short(x, nd=4):
    """Floats to `nd` significant digits, recursively: the stdout line is a summary, the detail file keeps full precision."""
    if isinstance(x, float):
        return float(f"{x:.{nd}g}")
    if isinstance(x, dict):
        return {k: _short(v, nd) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_short(v, nd) for v in x]
    return x


LINE_LIMIT = 6000   # bytes of the stdout line (the driver's record keeps the last 8 KB of stdout)


def compact_line(out, detail_name="bench_detail.json"):
    """The ONE stdout line: the contract's fields, the latency / host-boundary numbers, `roofline` for ONE kernel and
    `cpu_baseline`, a few hundred bytes each.  `out` is the full result (what goes to the detail file)."""
    r = out.get("roofline") or {}
    iss = r.get("valu_issue") or {}
    wp = r.get("whole_proof") or {}
    cfg = out.get("config") or {}
    line = {k: out.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "ms_per_step_min", "ms_per_step_max",
                                    "repeats", "timed_seconds", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    line["metric"] = str(line["metric"])[:400]
    line["config"] = {"workload": str(cfg.get("workload"))[:300], "degree_bits": cfg.get("degree_bits"), "lde_rows": cfg.get("lde_rows"),
                      "mix": cfg.get("mix"), "public_inputs": cfg.get("public_inputs"), "in_flight_per_gpu": out.get("in_flight_per_gpu"),
                      "parallelism": str(cfg.get("parallelism"))[:200], "proof_bytes": cfg.get("proof_bytes")}
    for k in ("latency_ms_single_proof", "latency_ms_single_proof_host_witness", "value_host_witness", "cold_process_ms"):
        line[k] = out.get(k)
    if (out.get("n_gpus") or 1) > 1 or out.get("peer_access") is not None:
        for k in ("latency_ms_sharded", "latency_ms_sharded_intt", "latency_ms_sharded_all_steps", "latency_ms_sharded_group", "latency_ms_sharded_group_intt",
                  "latency_ms_sharded_group_all_steps", "rccl_ranks", "peer_access",
                  "ranks_share_devices"):
            line[k] = out.get(k)
        sh = out.get("sharded") or {}
        if sh.get("error"):
            line["sharded_error"] = str(sh["error"])[:200]
        gr = sh.get("group") or {}
        if gr.get("error"):
            line["sharded_group_error"] = str(gr["error"])[:200]
        opr = gr.get("one_process_replicas") or {}
        if opr:
            line["one_process_replicas_proofs_per_sec"] = opr.get("proofs_per_sec")
    if r:
        line["roofline"] = {
            "kernel": r.get("kernel"), "bound": r.get("bound"), "achieved": r.get("achieved"), "peak": r.get("peak"), "unit": r.get("unit"),
            "frac": r.get("frac"), "traffic": r.get("traffic"), "traffic_source": r.get("traffic_source"),
            "avg_launch_ms": r.get("avg_launch_ms"), "launches_per_proof": r.get("launches_per_proof"),
            "algorithmic_bytes_per_launch": r.get("algorithmic_bytes_per_launch"),
            "timing": "lone launches: per-launch HIP events on the launch stream, one proof on the GPU at a time",
            "hbm": {"frac": r.get("frac")},
            "valu_issue": ({"frac": iss.get("frac"), "frac_of_mix_ceiling": iss.get("frac_of_mix_ceiling"), "peak": iss.get("peak"),
                            "unit": iss.get("unit"), "source": iss.get("source")} if iss else None),
            "whole_proof": {"algorithmic_bytes": wp.get("algorithmic_bytes"), "frac_lone": wp.get("frac"), "frac_at_value": wp.get("frac_at_throughput"),
                            "valu_floor": wp.get("valu_floor")},
        }
    cb = out.get("cpu_baseline")
    if cb:
        if "error" in cb:
            line["cpu_baseline"] = {"error": str(cb["error"])[:200]}
        else:
            st = cb.get("single_thread") or {}
            line["cpu_baseline"] = {"value": cb.get("value"), "unit": cb.get("unit"), "cores": cb.get("cores"), "kind": cb.get("kind"),
                                    "sample": str(cb.get("sample"))[:330], "seconds": cb.get("seconds"),
                                    "single_thread": ({"seconds": st.get("seconds"), "cores": 1, "scaled": st.get("scaled")} if "seconds" in st else None)}
    line["device"] = out.get("device")
    line["detail"] = detail_name
    txt = json.dumps(_short(line), separators=(",", ":"))
    if len(txt) > LINE_LIMIT:   # cannot happen with the caps above; if it ever does, drop prose before numbers
        for k in ("timing", "traffic_source"):
            line.get("roofline", {}).pop(k, None)
        line["metric"] = line["metric"][:120]
        txt = json.dumps(_short(line), separators=(",", ":"))
    assert len(txt) <= LINE_LIMIT, len(txt)
    return txt


def effective_cores():
    """CPUs this process may actually use: the cgroup CPU quota where there is one (the MI355X boxes show 256 hardware
    threads and a 16-CPU quota -- 128 OpenMP threads there run 2x SLOWER than 16), else the affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return n


def _free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def launch_ranks(args, argv):
    """`bench.py --gpus N` started without torchrun: this process is the launcher.  N ranks, one per device (LOCAL_RANK = device
    id), rendezvous on 127.0.0.1; rank 0 prints the JSON line on this process's stdout.  Fewer than N devices: exit code 2, no line."""
    import subprocess
    n = args.gpus
    if not args.dry and args.backend == "nccl":
        import torch
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < n:
            print(f"bench.py --gpus {n}: this box has {have} HIP device(s); refusing to report n_gpus = {n} from fewer devices "
                  f"(--backend gloo lets ranks share a device for functional checks only)", file=sys.stderr, flush=True)
            return 2
    env = dict(os.environ, WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), P2GPU_BENCH_LAUNCHED="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = []
    for r in range(n):
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for pr in procs:
        rc = pr.wait() or rc
    return rc


def dry_run(args):
    """--dry: the rank plumbing without a GPU (CPU-side test of `--gpus N`): rendezvous, barrier, the MAX-over-ranks
    reduction of the timed region, one JSON line from rank 0 naming every rank that took part."""
    import torch.distributed as dist
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world != args.gpus:
        print(f"bench.py --gpus {args.gpus} running as one of WORLD_SIZE={world} ranks", file=sys.stderr)
        return 2
    seen = [rank]
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo" if args.backend == "nccl" else args.backend, rank=rank, world_size=world)
        seen = [None] * world
        dist.all_gather_object(seen, rank)
        dist.barrier()
    pkg = entry.load_package()
    t0 = time.perf_counter()
    time.sleep(0.01 * (1 + rank))
    dt = pkg.parallel.max_over_ranks(time.perf_counter() - t0)
    if rank == 0:
        print(json.dumps({"dry": True, "n_gpus": world, "ranks": sorted(seen), "launched_by": "bench.py" if os.environ.get("P2GPU_BENCH_LAUNCHED") else "torchrun",
                          "backend": args.backend, "max_over_ranks_s": dt}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def exchange_stats(stats, proofs):
    """The pseudo-kernels the library records around every exchange of a sharded proof (`profile` = 2): microseconds per
    exchange from HIP events on the rank's own stream (enqueue of the exchange -> its last copy / collective done)."""
    out = {}
    for k, v in stats.items():
        if k.startswith("exchange["):
            out[k] = {"per_proof": v["launches"] / proofs, "avg_us": v["ms"] / v["launches"] * 1e3, "bytes_all_ranks_avg": v["bytes"] / v["launches"]}
    return out


def group_probe(pkg, args, group, blob, wires, pis):
    """ONE process driving the devices of `group` (p2gpu_init with several ids): latency of one proof coset-sharded over them,
    resident witness, + exchange timings + the peer-access matrix.  Returns a dict (printed as JSON in --group-probe mode)."""
    import torch
    cd = pkg.CircuitData(blob)
    wd = torch.from_numpy(wires.view(np.int64)).to(f"cuda:{group[0]}")
    for _ in range(3):
        ref = cd.prove(wd, public_inputs=pis)
    k = max(3, args.sharded_steps)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        cd.prove(wd, public_inputs=pis)
    ms = (time.perf_counter() - t0) / k * 1e3
    # knob shard_intt: column-sharded inverse transforms + all-gather of the coefficient blocks (peer copies, in place)
    ms_intt, ms_all, same = None, None, None
    try:
        cd.set("shard_intt", 1)
        same = cd.prove(wd, public_inputs=pis).to_bytes() == ref.to_bytes()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(k):
            cd.prove(wd, public_inputs=pis)
        ms_intt = (time.perf_counter() - t0) / k * 1e3
        cd.set("shard_zs", 1)        # every step of SURVEY 8(e)'s table sharded
        cd.set("shard_reduce", 1)
        same = same and cd.prove(wd, public_inputs=pis).to_bytes() == ref.to_bytes()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(k):
            cd.prove(wd, public_inputs=pis)
        ms_all = (time.perf_counter() - t0) / k * 1e3
    except Exception as e:
        same = repr(e)[:200]
    for k_ in ("shard_intt", "shard_zs", "shard_reduce"):
        cd.set(k_, 0)
    cd.set("profile", 2)
    for _ in range(2):
        cd.prove(wd, public_inputs=pis)
    st = cd.kernel_stats()
    cd.set("profile", 0)
    out = {"devices": group, "latency_ms_sharded_group": ms, "latency_ms_sharded_group_intt": ms_intt, "latency_ms_sharded_group_all_steps": ms_all, "shard_intt_same_bytes": same,
           "proofs": k, "peer_access": pkg.peer_access(),
           "exchanges": exchange_stats(st, 2), "proof_bytes": len(ref),
           "transport": "hipMemcpyPeerAsync between the ranks' streams, one host thread per rank inside p2gpu_prove_dev"}
    cd.close()
    # ... and replicas from the same single process: plain handles on named devices (p2gpu_circuit_create_on), two per device
    # entry of the list, one host thread each -- the throughput half of the metric without a process per GPU
    import threading
    per = 2
    hs = [(pkg.CircuitData(blob, device=dev), dev) for dev in group for _ in range(per)]
    wdev = {dev: torch.from_numpy(wires.view(np.int64)).to(f"cuda:{dev}") for dev in set(group)}
    n_each = max(4, args.sharded_steps * 2)

    def work(h, dev, n):
        for _ in range(n):
            h.prove(wdev[dev], public_inputs=pis)
    for n in (2, n_each):   # warm-up, then timed
        th = [threading.Thread(target=work, args=(h, dev, n)) for h, dev in hs]
        for dev in set(group):
            torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        dt = time.perf_counter() - t0
    out["one_process_replicas"] = {"proofs_per_sec": len(hs) * n_each / dt, "handles": len(hs), "per_device_entry": per, "proofs": len(hs) * n_each,
                                   "entry_point": "p2gpu_circuit_create_on + p2gpu_prove_dev from one host thread per handle"}
    for h, _ in hs:
        h.close()
    return out


def cold_process(pkg, blob, wires, pis):
    """A fresh process -> p2gpu_init -> p2gpu_circuit_create -> ONE p2gpu_prove -> exit: the plain-C caller with --timing
    (tools/p2gpu_prove.c), inputs on a RAM disk so that `cold_process_ms` (init + create + first prove) excludes file I/O."""
    import subprocess
    import tempfile
    tool = os.path.join(os.path.dirname(pkg.lib_path()), "p2gpu-prove")
    base = "/dev/shm" if os.path.isdir("/dev/shm") else None
    tmp = tempfile.mkdtemp(prefix="p2gpu_cold_", dir=base)
    try:
        bp, wp, pp, pip = (os.path.join(tmp, x) for x in ("c.blob", "w.bin", "proof.bin", "pi.bin"))
        np.asarray(blob).tofile(bp)
        np.asarray(wires).tofile(wp)
        cmd = [tool, bp, wp, pp]
        if len(pis):
            np.asarray(pis, dtype=np.uint64).tofile(pip)
            cmd.append(pip)
        runs = []
        for _ in range(5):
            r = subprocess.run(cmd + ["--timing"], capture_output=True, text=True, timeout=600)
            if r.returncode != 0:
                return {"error": (r.stderr or r.stdout)[-300:]}
            runs.append(json.loads(r.stdout.strip().splitlines()[-1]))
        best = min(runs, key=lambda x: x["cold_process_ms"])
        best["runs_cold_process_ms"] = [x["cold_process_ms"] for x in runs]
        # the same box, the same minute: what a HIP process that does nothing of ours pays (tools/cold_floor.hip)
        floor_tool = os.path.join(os.path.dirname(pkg.lib_path()), "p2gpu-cold-floor")
        if os.path.exists(floor_tool):
            fl = []
            for _ in range(5):
                r = subprocess.run([floor_tool], capture_output=True, text=True, timeout=120)
                if r.returncode == 0:
                    fl.append(json.loads(r.stdout.strip().splitlines()[-1]))
            if fl:
                best["hip_floor"] = {k: min(x[k] for x in fl) for k in fl[0]}
                best["hip_floor"]["what"] = ("per-field min of 5 fresh processes that only call the HIP runtime: hipGetDeviceCount (runtime start-up) is what "
                                             "p2gpu_init pays, hipSetDevice + the first stream is the head of p2gpu_circuit_create, before any library work")
                best["hip_start_up_floor_ms"] = best["hip_floor"]["hipGetDeviceCount_ms"] + best["hip_floor"]["setdevice_stream_ms"]
                # what is the library's own: everything after p2gpu_init, minus the context / first-stream creation no HIP program avoids
                best["library_ms_after_hip_start_up"] = best["circuit_create_ms"] + best["first_prove_ms"] - best["hip_floor"]["setdevice_stream_ms"]
        best["what"] = ("fresh process (plain C on the C ABI, no Python): p2gpu_init + p2gpu_circuit_create + the first p2gpu_prove, witness in host RAM; "
                        "best of 5 processes (the HIP runtime's own start-up inside p2gpu_init varies between 50 and 250 ms from process to process on one box); "
                        "read_inputs_ms / write_proof_ms (RAM disk) are outside cold_process_ms")
        return best
    finally:
        import shutil
        shutil.rmtree(tmp, ignore_errors=True)


