#!/usr/bin/env python3
"""Generalised and linear-response drivers with 55-column blocks (50 roots) against the reference is a CPU-side check (tests use the host engine); this runs the same calls on the HIP engine and prints the agreement with the oracle-free dense solution."""
import os, sys
os.environ.setdefault("OMP_NUM_THREADS","4")
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np

from diaglib_amd import capi

from oracle.pyoracle import Oracle
import scipy.linalg as sl
o=Oracle(); ctx=capi.Context()
ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE,0)
n,t,m=1500,50,55
o.dense_setup(n); o.metric_setup(n)
mv,pc,bv=o.fn("orc_dense_matvec"),o.fn("orc_dense_precnd"),o.fn("orc_metric_matvec")
g=np.zeros((n,m),order='F'); g[np.arange(m),np.arange(m)]=1
e,v,ok,info=ctx.gen_david_driver(n,t,m,100,1e-9,20,0.0,mv,pc,bv,g)
idx=np.arange(1,n+1.0); a=1/(idx[:,None]+idx[None,:]); np.fill_diagonal(a,idx+1); S=o.metric_setup(n); er=sl.eigh(a,S,eigvals_only=True); okr=True
print('gen_david',ok,okr,info,np.abs(e[:t]-er[:t]).max())
e,v,ok,info=ctx.lobpcg_driver(n,t,m,100,1e-9,0.0,mv,pc,g,bvec=bv)
print('lobpcg_gen',ok,okr,info,np.abs(e[:t]-er[:t]).max())
# LR with wide block
n=600; t=50; m=55
apb,amb,spd,smd=o.lr_setup(n)
fn=[o.fn(k) for k in ("orc_lr_apb","orc_lr_amb","orc_lr_spd","orc_lr_smd","orc_lr_prec")]
g=np.zeros((2*n,m),order='F'); g[np.arange(m),np.arange(m)]=1
e,v,ok,info=ctx.caslr_eff_driver(n,t,m,100,1e-8,20,*fn,g)
A=0.5*(apb+amb); B=0.5*(apb-amb); Sg=0.5*(spd+smd); D=0.5*(spd-smd); w=sl.eigh(np.block([[Sg,D],[-D,-Sg]]),np.block([[A,B],[B,A]]),eigvals_only=True); er=np.sort(1.0/w[w>0])
print('caslr_eff',ok,okr,info,np.abs(e[:t]-er[:t]).max())
