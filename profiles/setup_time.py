import sys, time, os
sys.path.insert(0,'/root/repo')
os.environ["SLPX_SETUP_TIMING"]="1"
import sleipnir_amd as sa
for N in (1000,):
    sa.lib().slpx_graph_reset()
    t=time.time(); pp=sa.Problem.cart_pole(N,5.0/N); print("model",time.time()-t)
    t=time.time(); s=sa.System(pp,1,0); print("system",time.time()-t)
    s.close()
    t=time.time(); s=sa.System(pp,1,0); print("system again",time.time()-t)
