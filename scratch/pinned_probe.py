# CPU-side read speed of p2gpu_host_alloc memory vs pageable numpy memory, and lone host-witness proofs from each
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
blob, wires = pkg.make_circuit(17, "sha", 1)
cd = pkg.CircuitData(blob)
wp = pkg.host_array(wires.shape)
wp[...] = wires
for name, a in (("pageable", wires), ("pinned", wp)):
    for _ in range(2):
        t = time.perf_counter(); s = int(np.bitwise_or.reduce(a.reshape(-1)[-154 * (1 << 17):])); dt = time.perf_counter() - t
    print(name, "CPU or-reduce of the last 154 columns: %.2f ms = %.1f GB/s" % (dt * 1e3, 154 * (1 << 17) * 8 / dt / 1e9))
for rep in range(2):
    for name, a in (("pageable", wires), ("pinned", wp)):
        cd.prove(a)
        t = time.perf_counter()
        tm = [cd.prove(a).timings for _ in range(5)]
        dt = (time.perf_counter() - t) / 5 * 1e3
        print(name, "lone %.3f ms, h2d span %.3f ms, wires_commit %.3f" % (dt, sum(x["h2d_ms"] for x in tm) / 5, sum(x["wires_commit_ms"] for x in tm) / 5))
