# Developer aid: one small null-exact call through the host-pointer API, prints the device status (eigensolver clock,
# weight-basis rank / residual, phase times).  Run on the GPU box: python tools/dbg_status.py
import numpy as np, sys, ctypes as C
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from common import make_data
import bulklmm_jl_amd as B
from bulklmm_jl_amd import api, _lib as L
Y,G,K,Cov=make_data(p=500,m=300,seed=1)
Y=np.asfortranarray(Y); G=np.asfortranarray(G); K=np.asfortranarray(K)
n,m=Y.shape; p=G.shape[1]
ctx=B.default_context(); ctx.set_timing(True)
o=api._opts(L.BLMM_NULL_EXACT)
Lo=np.empty((p,m),order='F'); h2=np.empty(m); st=L.blmm_status()
for rep in range(2):
    rc=ctx.lib.blmm_bulkscan(ctx.h,C.byref(o),api._p(Y),n,m,api._p(G),p,None,0,api._p(K),None,None,0,api._p(Lo),api._p(h2),C.byref(st))
    assert rc==0
print("jacobi cycles", st.jacobi_cycles, "MHz", st.jacobi_cycles/max(st.jacobi_ticks_100mhz,1)*100, "sweeps", st.jacobi_sweeps)
print("lowrank rank", st.lowrank_rank, "resid", st.lowrank_resid)
print("eigen ms", st.t_eigen_ms, "h2 ms", st.t_h2_ms, "scan", st.t_scan_ms)
