import numpy as np, sys
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from common import make_data
from oracle import bulklmm_oracle as O
import bulklmm_jl_amd as B
Y,G,K,Cov=make_data(p=50,m=1,seed=101)
n=79; nperms=5
pidx=np.stack([np.arange(n) for _ in range(nperms)],axis=1).astype(np.int32)
pidx[:,1]=pidx[::-1,1]
got=B.scan(Y[:,0],G,K,permutation_test=True,nperms=nperms,perm_idx=pidx,prior_variance=1.0,prior_sample_size=0.1)
ref=O.scan(Y[:,0],G,K,permutation_test=True,nperms=nperms,perm_idx=pidx,prior_variance=1.0,prior_sample_size=0.1,h2_override=got['h2_null'])
print("lod diff", np.abs(got['lod']-ref['lod']).max())
print("identity perm col0 diff", np.abs(got['L_perms'][:,0]-got['lod']).max(), np.abs(ref['L_perms'][:,0]-ref['lod']).max())
print("reverse perm diff", np.abs(got['L_perms'][:,1]-ref['L_perms'][:,1]).max())
print(got['L_perms'][:3], ref['L_perms'][:3])
