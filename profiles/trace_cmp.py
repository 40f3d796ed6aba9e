import sys, numpy as np
sys.path.insert(0,'/root/repo')
import sleipnir_amd as sa
from tests.support import models, oracle
N=int(sys.argv[1]) if len(sys.argv)>1 else 100
sa.lib().slpx_graph_reset(); oracle.lib().orc_reset()
pp=models.cart_pole(N,5.0/N); op=oracle.OracleProblem.cart_pole(N,5.0/N)
rows=[]
def cb(info):
    n=pp.dims[0]; mi=pp.dims[2]
    rows.append([info["iteration"], len(info["x"]), np.linalg.norm(info["x"][:n]), np.linalg.norm(info["s"][:mi]), np.linalg.norm(info["y"]), np.linalg.norm(info["z"]), info["in_restoration"]])
    return False
pp.add_callback(cb)
perm=pp.system().perm()
st,rep=pp.solve()
tr=np.array(rows)
so,to=op.solve_trace(perm=perm)
print("product status",st,"iterations",rep["iterations"],"restorations",rep["restorations"],"records",len(tr))
print("oracle  status",so,"records",len(to), "restoration records", int(np.sum(to[:,1]!=pp.dims[0])))
m=min(len(tr),len(to))
for k in range(m):
    d=np.abs(tr[k,2:6]-to[k,2:6])/np.maximum(1.0,np.abs(to[k,2:6]))
    if k<5 or d.max()>1e-6:
        print(k, tr[k,1], to[k,1], "in_rest", tr[k,6], "rel diff", d)
        if d.max()>1e-6 and k>=5: break
print("first restoration record product:", next((int(r[0]) for r in tr if r[6]), None), " oracle:", next((int(r[0]) for r in to if r[1]!=pp.dims[0]), None))
