import numpy as np, sys
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from common import make_data
import bulklmm_jl_amd as B
from bulklmm_jl_amd import api, _lib as L
Y,G,K,Cov=make_data(p=500,m=300,seed=1)
ctx=B.default_context(); ctx.set_timing(True)
for rep in range(2):
    Lo,h2,st=api._bulkscan_call(L.BLMM_NULL_EXACT,Y,G,K,None,None,True,None,1.0,0.0,False,1,"eigen",0,ctx,return_status=True)
print("jacobi cycles", st.jacobi_cycles, "ticks", st.jacobi_ticks_100mhz, "MHz", st.jacobi_cycles/max(st.jacobi_ticks_100mhz,1)*100)
print("jacobi sweeps", st.jacobi_sweeps, "eigen ms", st.t_eigen_ms, "h2 ms", st.t_h2_ms, "scan", st.t_scan_ms)
print("h2 quantiles", np.quantile(h2,[0,0.1,0.5,0.9,1]))
