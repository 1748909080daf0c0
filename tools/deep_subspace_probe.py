import sys, numpy as np
sys.path.insert(0,'/root/repo')
from diaglib_amd import capi
from oracle.pyoracle import Oracle
o=Oracle(); ctx=capi.Context()
n,t,m=3000,32,37
o.dense_setup(n); mv,pc=o.fn("orc_dense_matvec"),o.fn("orc_dense_precnd")
g=np.asfortranarray(np.random.default_rng(5).random((n,m))-0.5)
ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE,0)
e,v,ok,info=ctx.davidson_driver(n,t,m,300,1e-9,20,0.0,mv,pc,g)
idx=np.arange(1,n+1.0); a=1/(idx[:,None]+idx[None,:]); np.fill_diagonal(a,idx+1); w=np.linalg.eigvalsh(a)[:t]
print('ok',ok,info,'max eig err',np.abs(e[:t]-w).max())
eo,vo,oko,tr=o.davidson(n,t,m,300,1e-9,20,0.0,mv,pc,g)
print('oracle ok',oko,tr.iters,tr.restarts,np.abs(eo[:t]-w).max())
