import numpy as np, sys, ctypes as C, warnings
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from common import make_data
from oracle import bulklmm_oracle as O
import bulklmm_jl_amd as B
from bulklmm_jl_amd import api, _lib as L
n=int(sys.argv[1]) if len(sys.argv)>1 else 333
Y,G,K,_=make_data(n=n,p=150,m=21,seed=500+n,bxd=False)
Y=np.asfortranarray(Y); G=np.asfortranarray(G); K=np.asfortranarray(K)
m=Y.shape[1]; p=G.shape[1]
ctx=B.default_context()
o=api._opts(L.BLMM_NULL_EXACT)
Lo=np.empty((p,m),order='F'); h2=np.empty(m); st=L.blmm_status()
rc=ctx.lib.blmm_bulkscan(ctx.h,C.byref(o),api._p(Y),n,m,api._p(G),p,None,0,api._p(K),None,None,0,api._p(Lo),api._p(h2),C.byref(st))
print("rc",rc,"rank",st.lowrank_rank,"resid",st.lowrank_resid,"nan",st.n_nan_lod,"zero",st.n_zero_norm)
ref=O.bulkscan_null(Y,G,K,h2_override=h2)
print("max abs diff", np.abs(Lo-ref.L).max(), "Lo range", Lo.min(), Lo.max(), "ref max", ref.L.max())
lam=np.linalg.eigvalsh(K); print("lam range", lam.min(), lam.max())
