import sys, time, os, numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT','/root/repo'))
import importlib.util, os
def load_package():
    path = os.path.join(os.environ.get('GRAFT_REPO_ROOT','/root/repo'), "acvm-backend-plonky2_amd", "__init__.py")
    spec = importlib.util.spec_from_file_location("acvm_backend_plonky2_amd", path, submodule_search_locations=[os.path.dirname(path)])
    mod = importlib.util.module_from_spec(spec); sys.modules["acvm_backend_plonky2_amd"] = mod; spec.loader.exec_module(mod); return mod
P = load_package()
from oracle import pyoracle as O
print(P.device_info())
rng = np.random.default_rng(1)
PR = 0xFFFFFFFF00000001
for d in (3, 6, 10, 12, 13, 15):
    v = rng.integers(0, PR, size=(3, 1 << d), dtype=np.uint64)
    got = P.ifft_batch(v)
    exp = np.stack([O.ntt(r, inverse=True) for r in v])
    print('ifft', d, np.array_equal(got, exp))
    lg = P.lde_batch(exp); le = np.stack([O.coset_lde(r) for r in exp])
    print('lde ', d, np.array_equal(lg, le))
rows = rng.integers(0, PR, size=(5, 234), dtype=np.uint64)
print('hash', np.array_equal(P.hash_rows(rows), O.hash_rows(rows)))
for ncol in (2, 3, 17, 20, 33, 34):
    rows = rng.integers(0, PR, size=(4, ncol), dtype=np.uint64)
    print('hash', ncol, np.array_equal(P.hash_rows(rows), O.hash_rows(rows)))
v = rng.integers(0, PR, size=(20, 1 << 8), dtype=np.uint64)
print('commit', P.commit_values(v) == O.commit_values(v))
for d, mix in ((6, 'arith'), (9, 'sha'), (10, 'ecdsa'), (13, 'ecdsa')):
    blob, w = P.make_circuit(d, mix, 1)
    oc = O.OracleCircuit(blob); cd = P.CircuitData(blob)
    print(d, mix, 'cap', oc.cap() == cd.constants_sigmas_cap(), 'digest', oc.digest() == cd.circuit_digest())
    t = time.time(); po, tr = oc.prove(w); to = time.time() - t
    t = time.time(); pg = cd.prove(w); tg = time.time() - t
    pb = pg.to_bytes()
    print('  proof equal', po == pb, len(po), len(pb), 'oracle %.3fs gpu %.3fs' % (to, tg), 'verify', oc.verify(pb), pg.timings)
    if po != pb:
        n = min(len(po), len(pb)); diff = [i for i in range(n) if po[i] != pb[i]]
        print('  first diff at', diff[:5], 'of', n)
