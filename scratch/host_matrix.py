"""Host-boundary matrix: {full matrix, sparse entry} x {pageable, page-locked} x {lone, 4 in flight} at 2^20 rows.
usage: python scratch/host_matrix.py [mix] [d] [in_flight]   (P2GPU_HOSTPROF=1 prints the lone calls' host timelines)"""
import os, sys, time, threading
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import __graft_entry__ as e

mix = sys.argv[1] if len(sys.argv) > 1 else "sha"
d = int(sys.argv[2]) if len(sys.argv) > 2 else 17
S = int(sys.argv[3]) if len(sys.argv) > 3 else 4
pkg = e.load_package()
blob, wires = pkg.make_circuit(d, mix, 1)
cds = [pkg.CircuitData(blob) for _ in range(S)]
W = cds[0].num_wires
wm = wires.reshape(W, -1)
nzc = (wm != 0).sum(axis=1)
ncols = int(np.max(np.nonzero(nzc > 1)[0])) + 1
rows = {int(np.nonzero(wm[j])[0][0]) for j in range(ncols, W) if nzc[j] == 1}
row = rows.pop() if len(rows) == 1 else 0
wd = torch.from_numpy(wires.view(np.int64)).cuda()
ref = cds[0].prove(wd).to_bytes()


def run(handles, n, w):
    def work(i):
        for _ in range(i, n, len(handles)):
            if isinstance(w, tuple):
                p = handles[i].prove_sparse(w[0], w[1], w[2], tail=w[3])
            else:
                p = handles[i].prove(w)
        out[i] = p
    out = [None] * len(handles)
    if len(handles) == 1:
        work(0)
    else:
        th = [threading.Thread(target=work, args=(i,)) for i in range(len(handles))]
        [t.start() for t in th]
        [t.join() for t in th]
    return out[0]


def measure(name, w, HW=8):
    assert run(cds[:1], 1, w).to_bytes() == ref, name
    run(cds[:1], 2, w)
    sys.stderr.write(f"--- {name} lone\n")
    sys.stderr.flush()
    t = time.perf_counter()
    run(cds[:1], HW, w)
    lone = (time.perf_counter() - t) / HW * 1e3
    run(cds, 2 * S, w)
    torch.cuda.synchronize()
    best = 0
    for _ in range(3):
        t = time.perf_counter()
        run(cds, HW * S, w)
        torch.cuda.synchronize()
        best = max(best, HW * S / (time.perf_counter() - t))
    print(f"{name:34s} lone {lone:6.3f} ms   {S} in flight {best:6.1f} proofs/s", flush=True)


measure("resident (p2gpu_prove_dev)", wd)
measure("full matrix, pageable", wires)
wp = pkg.host_array(wires.shape)
wp[...] = wires
measure("full matrix, page-locked", wp)
dense = np.ascontiguousarray(wm[:ncols]).reshape(-1)
tail = np.ascontiguousarray(wm[ncols:, row])
measure("sparse entry, pageable", (dense, ncols, row, tail))
dp = pkg.host_array(dense.shape)
dp[...] = dense
measure("sparse entry, page-locked", (dp, ncols, row, tail))
