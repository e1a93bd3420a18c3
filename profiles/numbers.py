#!/usr/bin/env python3
"""profiles/NUMBERS.md from the newest committed bench lines (profiles/rNN_bench*.json): ONE table of current numbers, so that no
figure has to be repeated (and drift) in DESIGN.md / README.md.  Usage: python profiles/numbers.py [tag]   (default: newest rNN)"""
import glob
import json
import os
import re
import sys

here = os.path.dirname(os.path.abspath(__file__))


def load(path):
    """A full result (bench.py --detail: one JSON document, round 5 on) or, rounds 1-4, the stdout line."""
    try:
        return json.load(open(path))
    except Exception:
        pass
    try:
        lines = [ln for ln in open(path) if ln.startswith("{")]
        return json.loads(lines[-1]) if lines else None
    except Exception:
        return None


def f(x, nd=2):
    return "–" if x is None else (f"{x:.{nd}f}" if isinstance(x, float) else str(x))


# ---- instruction budget per proof and the floor it implies ---------------------------------------------
# Ceilings (profiles/r03_ubench.txt, MI355X at 2.38 GHz, 8 waves per SIMD): half-rate class 37.7 T lane-instr/s (every carry,
# multiply, compare: all of the field arithmetic); the Keccak-f round mix (120 v_bitop3 + 58 v_alignbit) 49.9 T as an interleaved stream.
HALF_T, KECCAK_T = 37.7e12, 49.9e12
CLASSES = [("Keccak (leaf hashes, tree levels, FRI leaves, PoW)", ("hash_lde_leaves_kf", "hash_lde_absorb_kf", "merkle_level_kf", "merkle_levels_kf", "merkle_coop_kernel",
                                                                 "hash_fri_leaves_kernel<0", "hash_lde_leaves_kernel<0", "hash_rows", "merkle_level_kernel<0", "pow_kernel"), KECCAK_T),
           ("Poseidon hashing (PoseidonGoldilocksConfig)", ("hash_lde_leaves_kernel<1", "hash_fri_leaves_kernel<1", "merkle_level_kernel<1", "merkle_coop_poseidon", "hash_fri_leaves_coop_poseidon"), HALF_T),
           ("NTT (inverse transforms, LDE, FRI)", ("ntt_", "structured_fill", "column_"), HALF_T),
           ("quotient (permutation argument, gates, Z / partial products)", ("quotient", "gate_sums", "poseidon_gate", "zs_", "scan_"), HALF_T)]


def budget_of(sq):
    """(lane-instructions per proof by class, floor in ms) of one SQ summary (profiles/summarize.py's <tag>_<workload>_sq_summary.json)."""
    ks = sq["kernels"]
    proofs = max((v["launches"] for k, v in ks.items() if k.startswith(("quotient_kernel", "zs_chunk_kernel"))), default=0)
    if not proofs:
        return None, None
    tot = {c[0]: 0.0 for c in CLASSES}
    rest = "rest (openings, FRI reduce / fold / quotient, gather)"
    tot[rest] = 0.0
    for k, v in ks.items():
        li = v["SQ_INSTS_VALU"] * 64.0 * v["launches"] / proofs
        for name, pres, _ in CLASSES:
            if k.startswith(pres):
                tot[name] += li
                break
        else:
            tot[rest] += li
    ceil = {c[0]: c[2] for c in CLASSES}
    return tot, sum(v / ceil.get(k, HALF_T) for k, v in tot.items()) * 1e3


def instruction_budget(tag):
    """Lane-instructions per proof by class from the committed SQ passes (profiles/<tag>_<workload>_sq_summary.json: SQ_INSTS_VALU per
    launch x 64 lanes x launches / proofs) and the time each class needs at the ceiling of its instruction mix: the floor of a proof
    on one MI355X whatever the overlap -- what `ms_per_step` is to be judged against."""
    rows = []
    for wl, label, bench in (("sha17", "synth(17, sha)", "bench"), ("ecdsa17", "synth(17, ecdsa)", "bench_d17_ecdsa"), ("grammar17", "synth(17, grammar)", None),
                             ("poseidon17", "synth(17, sha), Poseidon hasher", "bench_poseidon")):
        sq = load(os.path.join(here, f"{tag}_{wl}_sq_summary.json"))
        if not sq:
            continue
        tot, floor_ms = budget_of(sq)
        if tot is None:
            continue
        ceil = {c[0]: c[2] for c in CLASSES}
        b = load(os.path.join(here, f"{tag}_{bench}.json")) if bench else None
        rows.append((label, tot, floor_ms, b))
        if wl in ("grammar17", "sha17"):   # the 2^24-row configurations: 16 x the rows; the transforms also grow by the layer count
            mixn = wl[:-2]
            big = {k: v * 16 * (24.0 / 20.0 if k.startswith("NTT") else 1.0) for k, v in tot.items()}
            fl = sum(v / ceil.get(k, HALF_T) for k, v in big.items()) * 1e3
            rows.append((f"synth(21, {mixn}) — scaled from the 2^20-row pass", big, fl, load(os.path.join(here, f"{tag}_bench_d21_{mixn}.json"))))
    if not rows:
        return []
    out = ["", "## Instruction budget per proof and the floor it implies\n",
           "Lane-instructions per proof by class (SQ_INSTS_VALU x 64 from the committed SQ passes, `profiles/" + tag + "_*_sq_summary.json`) and the time each class "
           "needs at the ceiling of its own instruction mix (half-rate class 37.7 T lane-instr/s; the Keccak-f round mix 49.9 T: `profiles/r03_ubench.txt`): the floor of "
           "one proof on one MI355X whatever the overlap.  2^24 rows: x16 for everything but the transforms, whose count per element grows with the "
           "layers (x16 x 24/20 for the LDE).\n",
           "| workload | " + " | ".join(k.split(" (")[0] for k in rows[0][1]) + " | total G lane-instr | floor ms | measured ms/step at `value` (lone) | floor / measured |",
           "|---|" + "---|" * (len(rows[0][1]) + 4)]
    for label, tot, floor_ms, b in rows:
        cells = " | ".join(f"{v / 1e9:.1f} G" for v in tot.values())
        ms = b.get("ms_per_step") if b else None
        lone = b.get("latency_ms_single_proof") if b else None
        out.append(f"| {label} | {cells} | {sum(tot.values()) / 1e9:.1f} | {floor_ms:.2f} | {f(ms, 2)} ({f(lone, 2)}) | {f(floor_ms / ms if ms else None, 2)} |")
    return out


def main():
    tags = sorted({re.match(r"^(r\d+[a-z]?)_bench", os.path.basename(p)).group(1) for p in glob.glob(os.path.join(here, "r*_bench*.json"))})
    tag = sys.argv[1] if len(sys.argv) > 1 else tags[-1]
    rows = [("bench", "synth(17, sha) — BASELINE configs[2], the bench line"), ("bench_sha256x4", "SHA-256 of 4 blocks through the restated translator (2^20 rows)"),
            ("bench_pi4", "synth(17, sha) + 4 public inputs"), ("bench_d17_ecdsa", "synth(17, ecdsa): every gate kind, 231 dense columns"),
            ("bench_d13_arith", "synth(13, arith) — 2^16 rows"), ("bench_d19_ecdsa", "synth(19, ecdsa) — configs[3], 2^22 rows"),
            ("bench_d21_sha", "synth(21, sha) — 2^24 rows"), ("bench_d21_grammar", "synth(21, grammar) — configs[4], 2^24 rows"),
            ("bench_poseidon", "synth(17, sha), PoseidonGoldilocksConfig")]
    out = [f"# Current numbers ({tag}; one MI355X; generated by profiles/numbers.py — do not edit)\n",
           "`value` = proofs/s, witness resident in HBM, `in flight` proofs per GPU, median of timed regions totalling ≥ 1.5 s; lone = one proof on the GPU at a",
           "time; host = `p2gpu_prove` with the wire matrix in host RAM (SURVEY §8(d)'s boundary, PCIe inside the call).  Sources: `profiles/" + tag + "_bench*.json`.\n",
           "| workload | value (proofs/s) | in flight | ms/step (min–max, repeats) | lone ms | host lone ms | host proofs/s | dominant kernel (lone launches): ms per proof, frac of HBM peak by SURVEY §8(d)'s bytes, VALU issue frac (of its mix ceiling) |",
           "|---|---|---|---|---|---|---|---|"]
    main_line = None
    for name, label in rows:
        d = load(os.path.join(here, f"{tag}_{name}.json"))
        if not d:
            continue
        if name == "bench":
            main_line = d
        r = d.get("roofline", {})
        k = r.get("kernel", "")
        km = (d.get("kernel_ms_per_proof_lone") or d.get("kernel_ms_per_proof") or {}).get(k)
        iss = r.get("valu_issue") or r.get("issue") or {}
        hb = r.get("frac") if r.get("bound") == "hbm" else (r.get("hbm") or {}).get("frac")
        out.append(f"| {label} | {f(d['value'], 1)} | {d.get('in_flight_per_gpu')} | {f(d['ms_per_step'], 3)} ({f(d.get('ms_per_step_min'), 3)}–{f(d.get('ms_per_step_max'), 3)}, {d.get('repeats')}) | "
                   f"{f(d.get('latency_ms_single_proof'), 2)} | {f(d.get('latency_ms_single_proof_host_witness'), 2)} | {f(d.get('value_host_witness'), 1)} | "
                   f"`{k}` {f(km, 3)}, {f(hb, 3)}, {f(iss.get('frac'), 3)} ({f(iss.get('frac_of_mix_ceiling'), 2)}) |")
    if main_line:
        d = main_line
        out += ["", "## The bench line's workload in detail\n",
                "Every duration below is a LONE launch time (one proof on the GPU at a time: per-launch HIP events on the launch stream); the span column is the event span "
                "with the timed region's proofs in flight — a span includes the time a kernel shares the chip with the other proofs' kernels, so nothing is divided by it.\n",
                "| kernel | lone ms per proof | launches per proof | span with proofs in flight (ms per proof) | frac of HBM peak (launcher's bytes) | PMC traffic / those bytes |", "|---|---|---|---|---|---|"]
        kk = (d.get("roofline") or {}).get("kernels") or {}
        for k, v in list(d["kernel_ms_per_proof_lone"].items())[:14]:
            q = kk.get(k, {})
            out.append(f"| `{k}` | {f(v, 3)} | {f(q.get('launches_per_proof'), 1)} | {f(d.get('kernel_span_ms_per_proof', {}).get(k), 3)} | {f(q.get('hbm_frac'), 3)} | {f(q.get('traffic_over_algorithmic'), 2)} |")
        ks = d.get("kernel_ms_sum", {})
        out.append(f"| sum of all {len(d['kernel_ms_per_proof_lone'])} kernels | {f(ks.get('lone'), 3)} (lone proof {f(d['latency_ms_single_proof'], 2)}) | | {f(ks.get('spans_in_flight'), 3)} (ms_per_step {f(d['ms_per_step'], 3)} x {d.get('in_flight_per_gpu')}) | | |")
        r = d["roofline"]
        iss = r.get("valu_issue") or {}
        out += ["", f"Roofline of `{r['kernel']}` (the kernel with the largest lone time per proof): {r['achieved']:.0f} GB/s by SURVEY §8(d)'s bytes = **{r['frac']:.3f}** of the 8 TB/s HBM peak over its "
                f"{r['avg_launch_ms']:.3f} ms lone launch; PMC traffic {(r.get('traffic') or 0) / 1e6:.0f} MB per launch ({r.get('traffic_source')}); the kernel is VALU-issue-bound: "
                f"{(iss.get('achieved') or 0) / 1e12:.1f} T lane-instr/s = {f(iss.get('frac'), 3)} of the guide's 78.6 T, {f(iss.get('frac_of_mix_ceiling'), 2)} of its own mix's ceiling "
                f"({iss.get('source')}).  Whole proof: {r['whole_proof']['algorithmic_bytes'] / 1e9:.2f} GB algorithmic / lone latency = {r['whole_proof']['frac']:.3f} of HBM peak, "
                f"{r['whole_proof']['frac_at_throughput']:.3f} at `value`."]
        rs = r.get("steps", {})
        if rs:
            out += ["", "| prover step | lone kernel ms | processed bytes | frac of HBM peak | frac incl. elided columns |", "|---|---|---|---|---|"]
            for s_, v in rs.items():
                out.append(f"| {s_} | {f(v['kernel_ms'], 3)} | {v['processed_bytes'] / 1e9:.3f} GB | {f(v['frac'], 3)} | {f(v['frac_incl_elided'], 3)} |")
        hw = d.get("host_witness") or {}
        out += ["", f"Host-witness detail: pageable {f(hw.get('ms_per_proof'), 2)} ms ({f(hw.get('proofs_per_sec_in_flight'), 1)} proofs/s in flight), "
                f"page-locked {f((hw.get('pinned') or {}).get('ms_per_proof'), 2)} ms, sparse entry {f((hw.get('sparse') or {}).get('ms_per_proof'), 2)} ms; h2d span {f(hw.get('h2d_ms'), 2)} ms."]
        cb = d.get("cpu_baseline") or {}
        st = cb.get("single_thread") or {}
        out += [f"CPU baseline (oracle, same circuit, unscaled): {f(cb.get('seconds'), 2)} s on {cb.get('cores')} threads = {f(cb.get('value'), 3)} proofs/s; one thread {f(st.get('seconds'), 1)} s "
                f"(phases {json.dumps({k: round(v, 2) for k, v in (cb.get('phase_seconds') or {}).items()})})."]
        cp = d.get("cold_process") or {}
        if cp and "error" not in cp:
            out += [f"Cold process (fresh `p2gpu-prove --timing`, best of 5): init {f(cp.get('p2gpu_init_ms'), 1)} + create {f(cp.get('circuit_create_ms'), 1)} + first prove {f(cp.get('first_prove_ms'), 1)} "
                    f"= **{f(cp.get('cold_process_ms'), 1)} ms** (second prove on the warm handle {f(cp.get('second_prove_ms'), 1)}); HIP's own start-up in a process that does nothing else: "
                    f"{f((cp.get('hip_floor') or {}).get('hipGetDeviceCount_ms'), 1)} + {f((cp.get('hip_floor') or {}).get('setdevice_stream_ms'), 1)} ms; the library after that start-up: "
                    f"{f(cp.get('library_ms_after_hip_start_up'), 1)} ms."]
    g = load(os.path.join(here, f"{tag}_bench_gloo2.json"))
    if g:
        sh = g.get("sharded") or {}
        out += ["", "## Multi-rank flow on ONE GPU (functional: two gloo ranks share the device — not a scaling point)\n",
                f"`bench.py --gpus 2 --backend gloo`: replicas {f(g['value'], 1)} proofs/s (`ranks_share_devices` = {g.get('ranks_share_devices')}); one proof over the two ranks, host-callback "
                f"transport {f(g.get('latency_ms_sharded'), 1)} ms; the same from one process as a device group {f(g.get('latency_ms_sharded_group'), 2)} ms, peer_access {g.get('peer_access')}; "
                f"single-process replicas {f(((sh.get('group') or {}).get('one_process_replicas') or {}).get('proofs_per_sec'), 1)} proofs/s.",
                "", "| exchange (device group, µs per exchange from HIP events) | per proof | avg µs |", "|---|---|---|"]
        for k, v in ((sh.get("group") or {}).get("exchanges") or {}).items():
            out.append(f"| {k} | {f(v['per_proof'], 1)} | {f(v['avg_us'], 1)} |")
    gr = load(os.path.join(here, f"{tag}_bench_group2_same_gpu.json"))
    if gr:
        out += ["", f"`bench.py --group 0,0` (every proof sharded over a two-rank device group on one GPU): {f(gr['ms_per_step'], 2)} ms per proof resident, "
                f"{f(gr.get('latency_ms_single_proof_host_witness'), 2)} ms with the host witness."]
    out += instruction_budget(tag)
    open(os.path.join(here, "NUMBERS.md"), "w").write("\n".join(out) + "\n")
    print("\n".join(out))
    # a fraction of a peak above 1 means a numerator or a denominator is not what its column says (a 1.124 was
    # printed): refuse to generate such a table
    bad = []
    if main_line:
        for k, q in ((main_line.get("roofline") or {}).get("kernels") or {}).items():
            if (q.get("hbm_frac") or 0) > 1.0:
                bad.append((k, q["hbm_frac"]))
        for s_, v in ((main_line.get("roofline") or {}).get("steps") or {}).items():
            if max(v.get("frac") or 0, v.get("frac_incl_elided") or 0) > 1.0:
                bad.append((s_, v))
    if bad:
        print("fractions above 1:", bad, file=sys.stderr)
        sys.exit(1)


if __name__ == "__main__":
    main()
