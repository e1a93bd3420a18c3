import os, sys, time
import numpy as np
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
import torch
print('torch', torch.__version__, torch.cuda.is_available(), flush=True)
P = entry.load_package(); O = entry.load_oracle()
print(P.device_info(), flush=True)
blob, w = P.make_circuit(5, 'arith', 1)
cd = P.CircuitData(blob)
print('created', flush=True)
os.environ['AMD_LOG_LEVEL'] = '0'
pg = cd.prove(w)
print('proved', len(pg), flush=True)
oc = O.OracleCircuit(blob)
print('equal', oc.prove(w)[0] == pg.to_bytes(), flush=True)
