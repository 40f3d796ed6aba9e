import sys, numpy as np
sys.path.insert(0,'/root/repo')
import sleipnir_amd as sa
from tests.support import oracle, cases, models
for N in (6,20):
    sa.lib().slpx_graph_reset(); oracle.lib().orc_reset()
    pp=models.cart_pole(N,5.0/N); op=oracle.OracleProblem.cart_pole(N,5.0/N)
    n,me,mi=pp.dims
    scales=op.scaling()
    x,s,y,z,mu=cases.newton_state("interior",op.get_x(),n,me,mi,scales[0])
    so,xo,s_o,yo,zo=op.restoration_steps(x,s,y,z,mu,1)
    sp,xp,s_p,yp,zp=pp.restoration_steps(x,s,y,z,mu,1)
    # matrices at xo from the oracle (scaled as the solver sees them)
    op.newton_step(xo,s_o,yo,zo,mu,False,None)
    def dense(name,rows):
        cp,ri,v=op.csc(name); a=np.zeros((rows,n))
        for c in range(n):
            for q in range(cp[c],cp[c+1]): a[ri[q],c]+=v[q]
        return a
    Ae,Ai=dense("A_e",me),dense("A_i",mi); g=op.vec("g")
    Ahat=np.block([[Ae,np.zeros((me,mi))],[Ai,-np.diag(s_o)]])
    r=np.concatenate([g,-mu*np.ones(mi)])
    w,res,rank,sv=np.linalg.lstsq(Ahat.T,r,rcond=None)
    yt,zt=w[:me],w[me:]
    print("N",N,"rank",rank,"of",me+mi,"cond(Ahat)",sv[0]/sv[-1])
    print("  y: oracle vs lstsq",cases.max_rel(yo,yt)," product vs lstsq",cases.max_rel(yp,yt)," |y|max",np.abs(yt).max())
    print("  z: oracle vs lstsq",cases.max_rel(zo,zt)," product vs lstsq",cases.max_rel(zp,zt))
