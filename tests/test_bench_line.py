"""bench.py's stdout contract on a CPU-only host: ONE JSON line, well under the 8 KB of stdout the driver's record keeps
(round 4's line had grown to 21 KB and came back unparsed), carrying the contract's fields, `roofline` for ONE kernel and
`cpu_baseline`; everything else lives in the detail file."""
import importlib.util
import json
import os

from conftest import ROOT


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _canned(world=8):
    """A worst case: 8 GPUs, every optional half present, long strings everywhere, 40 profiled kernels."""
    kernels = {f"some_kernel_name_with_template_args<{i}, false, 12>": {"ms_per_proof_lone": 0.1 * i, "launches_per_proof": 3.0,
                                                                         "avg_launch_ms_lone": 0.0333 * i, "hbm_frac": 0.1} for i in range(40)}
    return {
        "metric": "prove latency (ms) + proofs/sec at 2^20 LDE rows, 8 GPU(s): " + "x" * 900,
        "value": 1765.4321, "unit": "proofs/sec", "n_gpus": world, "ranks_share_devices": False, "steps": 192, "warmup": 16,
        "clock_warmup_proofs": 80, "ms_per_step": 4.5312345, "ms_per_step_min": 4.51, "ms_per_step_max": 4.56, "repeats": 2,
        "timed_seconds": 1.74, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": "synth(d=17,sha): " + "w" * 700, "degree_bits": 17, "lde_rows": 1 << 20, "mix": "sha", "public_inputs": 0,
                   "parallelism": "replicas x8 " + "p" * 500, "proof_bytes": 170972, "witness": "resident " * 40},
        "roofline": {"kernel": "hash_lde_leaves_kf_kernel<true>", "bound": "hbm", "achieved": 1215.3, "peak": 8000.0, "unit": "GB/s",
                     "frac": 0.1519, "traffic": 714e6, "traffic_source": "r05_sha17_pmc_summary.json", "avg_launch_ms": 1.643,
                     "launches_per_proof": 1.0, "timing": "t" * 300, "algorithmic_bytes_per_launch": 1.9965e9,
                     "algorithmic_bytes_definition": "d" * 300, "implementation_bytes_per_launch": 7.1e8, "frac_traffic": 0.054,
                     "valu_issue": {"bound": "valu-issue", "achieved": 41.5e12, "peak": 78.6e12, "unit": "lane-instr/s", "frac": 0.528,
                                    "frac_of_mix_ceiling": 0.83, "source": "r05_sha17_sq_summary.json", "mix_ceiling_source": "m" * 400},
                     "whole_proof": {"algorithmic_bytes": 9073957273, "achieved": 1620.0, "frac": 0.2026, "frac_at_throughput": 0.2511, "note": "n" * 200},
                     "steps": {f"step{i}": {"algorithmic_bytes": 1e9, "kernel_ms": 1.0, "frac": 0.1} for i in range(8)},
                     "kernels": kernels, "counter_summaries": {"pmc": "a", "sq": "b", "keyed_by": "k" * 100}},
        "latency_ms_single_proof": 5.6, "latency_ms_sharded": 3.1, "latency_ms_sharded_group": 2.9, "rccl_ranks": world,
        "peer_access": [[1] * world for _ in range(world)],
        "sharded": {"latency_ms_sharded": 3.1, "error": "e" * 1000, "exchanges_rank0": {f"exchange[{i}]": {"avg_us": 60.0} for i in range(8)},
                    "group": {"error": "g" * 1000, "one_process_replicas": {"proofs_per_sec": 1700.0}}},
        "cold_process_ms": 127.3, "cold_process": {"what": "c" * 600, "runs_cold_process_ms": [1.0] * 5},
        "latency_ms_single_proof_host_witness": 6.93, "value_host_witness": 207.2, "in_flight_per_gpu": 4,
        "host_witness": {"note": "h" * 500}, "phase_ms": {f"p{i}_ms": 0.1 for i in range(30)},
        "kernel_ms_per_proof_lone": {k: 0.1 for k in kernels}, "kernel_span_ms_per_proof": {k: 0.4 for k in kernels},
        "kernel_ms_sum": {"note": "s" * 800}, "kernel_profile": "k" * 300, "device": "AMD Instinct MI355X",
        "cpu_baseline": {"value": 0.444, "unit": "proofs/sec", "cores": 16, "kind": "port", "sample": "1 full proof " + "s" * 800, "seconds": 2.25,
                         "phase_seconds": {"wires": 1.4}, "single_thread": {"seconds": 27.5, "proofs_per_sec": 0.036, "cores": 1, "scaled": False,
                                                                            "sample": "o" * 300}},
    }


def test_stdout_line_is_short_and_complete():
    bench = _bench()
    line = bench.compact_line(_canned())
    assert "\n" not in line and len(line) < 8000 and len(line) <= bench.LINE_LIMIT
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline", "latency_ms_single_proof", "latency_ms_single_proof_host_witness", "value_host_witness"):
        assert k in d, k
    assert d["config"]["workload"].startswith("synth(d=17,sha)") and "model" not in d["config"]
    r = d["roofline"]
    assert set(("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(r)
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == "GB/s"
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0 < r["frac"] <= 1
    assert "kernels" not in r and "steps" not in r                      # ONE kernel on the line; the table is in the detail file
    assert r["valu_issue"]["frac"] == 0.528
    cb = d["cpu_baseline"]
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(cb) and cb["kind"] in ("port", "reference")
    assert cb["single_thread"]["seconds"] == 27.5
    # the multi-GPU half rides on the same line
    assert d["latency_ms_sharded"] == 3.1 and d["rccl_ranks"] == 8 and len(d["peer_access"]) == 8
    assert len(d["sharded_error"]) <= 200
    assert d["detail"] == "bench_detail.json"


def test_single_gpu_line_has_no_multi_gpu_fields():
    bench = _bench()
    c = _canned(world=1)
    c.update(n_gpus=1, peer_access=None, sharded=None, latency_ms_sharded=None, latency_ms_sharded_group=None, rccl_ranks=0)
    d = json.loads(bench.compact_line(c))
    assert "peer_access" not in d and "latency_ms_sharded" not in d
    assert len(json.dumps(d)) < 4000


def test_committed_bench_lines_of_this_round_fit():
    """Every stdout line committed under profiles/ from round 5 on (rNN_line*.json) is one parseable JSON line under 8 KB."""
    import glob
    import re
    for p in glob.glob(os.path.join(ROOT, "profiles", "r*_line*.json")):
        if int(re.match(r"r(\d+)", os.path.basename(p)).group(1)) < 5:
            continue
        lines = [ln for ln in open(p).read().splitlines() if ln.strip()]
        assert len(lines) == 1 and len(lines[0]) < 8000, p
        d = json.loads(lines[0])
        assert 0 < d["roofline"]["frac"] <= 1.0 and d["roofline"]["bound"] in ("hbm", "mfma"), p
        if re.match(r"r\d+_line\.json$", os.path.basename(p)):     # the driver's command: the side configurations skip the CPU legs
            assert d["cpu_baseline"]["kind"] in ("port", "reference") and d["cpu_baseline"]["cores"] >= 1, p
