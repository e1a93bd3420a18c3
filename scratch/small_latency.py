"""Lone-proof latency of small circuits (the reference's own example programs are 2^3 .. 2^6 gates): resident and host-witness entry.
usage: python scratch/small_latency.py [mix]      (P2GPU_HOSTPROF=1: host timeline of the last calls on stderr)"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import __graft_entry__ as e
pkg = e.load_package()
mix = sys.argv[1] if len(sys.argv) > 1 else "arith"
for d in (5, 6, 8, 10, 12, 13, 15):
    blob, wires = pkg.make_circuit(d, mix, 1)
    cd = pkg.CircuitData(blob)
    wd = torch.from_numpy(wires.view(np.int64)).cuda()
    for _ in range(20):
        cd.prove(wd)
    N = 200
    t = time.perf_counter()
    for _ in range(N):
        p = cd.prove(wd)
    dev = (time.perf_counter() - t) / N * 1e3
    for _ in range(10):
        cd.prove(wires)
    t = time.perf_counter()
    for _ in range(N):
        cd.prove(wires)
    host = (time.perf_counter() - t) / N * 1e3
    st = None
    cd.set("profile", 1)
    for _ in range(5):
        cd.prove(wd)
    st = cd.kernel_stats()
    cd.set("profile", 0)
    ksum = sum(v["ms"] for v in st.values()) / 5
    nl = sum(v["launches"] for v in st.values()) / 5
    print(f"d={d:2d} ({mix}): resident {dev:.3f} ms, host witness {host:.3f} ms, proof {len(p.to_bytes())} B; kernels {ksum:.3f} ms in {nl:.0f} launches", flush=True)
    cd.close()
