import sys,time,os
sys.path.insert(0,os.getcwd())
import __graft_entry__ as e
p=e.load_package(); o=e.load_oracle()
d=int(sys.argv[1]) if len(sys.argv)>1 else 17
m=p.make_circuit(d,'sha',seed=1)
c=o.OracleCircuit(m[0])
for _ in range(2):
    t=time.perf_counter(); pr,tr=c.prove(m[1]); dt=time.perf_counter()-t
    print(dt, {k:round(getattr(tr,k),3) for k in ['t_wires','t_zs','t_quotient','t_openings','t_fri']}, flush=True)
