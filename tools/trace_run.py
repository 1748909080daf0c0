import sys, numpy as np
sys.path.insert(0,'/root/repo')
from diaglib_amd import capi
ctx=capi.Context(); n=2_000_000; t=8; m=13
ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE,1); ctx.synth_setup(n,0,n)
mv,pc=capi.fn_address("dla_synth_matvec"),capi.fn_address("dla_synth_precnd")
g=np.zeros((n,m),order='F'); g[np.arange(m),np.arange(m)]=1
ev=ctx.panel(g)
eig,_,ok,info=ctx.davidson_driver(n,t,m,200,1e-13,20,0.0,mv,pc,ev,verbose=True)
print(info)
