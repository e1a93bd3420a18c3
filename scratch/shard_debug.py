"""world-2 (gloo, one shared GPU) sharded proofs vs the plain path: first differing prover stage per case"""
import os, sys, socket
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests", "golden"))
import numpy as np

def worker(rank, world, port, cases, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch, torch.distributed as dist
    import __graft_entry__ as entry
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = entry.load_package()
    out = []
    for (d, mix, npi) in cases:
        made = pkg.make_circuit(d, mix, 31, num_public_inputs=npi)
        blob, wires = made[0], made[1]
        pis = made[2] if npi else ()
        cd = pkg.CircuitData(blob)
        plain = cd.prove(wires, public_inputs=pis).to_bytes()
        cd.set_shard(rank, world)
        sh = cd.prove(wires, public_inputs=pis).to_bytes()
        out.append((d, mix, npi, plain, sh))
        cd.close()
    dist.barrier(); dist.destroy_process_group()
    q.put((rank, out))

if __name__ == "__main__":
    import torch.multiprocessing as mp
    import proof_stages, __graft_entry__ as entry
    cases = [(8, "ecdsa", 0), (9, "sha", 4), (13, "sha", 0), (13, "sha", 4), (13, "arith", 0), (12, "sha", 0), (10, "sha", 0)]
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn"); q = ctx.Queue()
    ps = [ctx.Process(target=worker, args=(r, 2, port, cases, q)) for r in range(2)]
    for p in ps: p.start()
    res = dict(q.get(timeout=600) for _ in ps)
    for p in ps: p.join()
    pkg = entry.load_package()
    for i, (d, mix, npi) in enumerate(cases):
        blob = pkg.make_circuit(d, mix, 31, num_public_inputs=npi)[0]
        for r in (0, 1):
            _, _, _, plain, sh = res[r][i]
            st_p, st_s = proof_stages.stages(blob, plain), proof_stages.stages(blob, sh)
            bad = [k for k in st_p if st_p[k] != st_s[k]]
            print(d, mix, npi, "rank", r, "ok" if not bad else "DIFF in " + ",".join(bad))
