"""Randomised check of the CG inverse of a damped GGN (plain and KFAC-preconditioned) on random small nets: GPU fp32
against CPU float64."""
import sys, os, copy, warnings
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tools')
import numpy as np, torch
from torch import nn
import curvlinops_amd as C
from fuzz_kfac import make_model, rel
dev=torch.device("cuda:0")
warnings.simplefilter("ignore")
fails=[]; worst=0
for seed in (1,2):
    rng=np.random.default_rng(seed)
    for case in range(30):
        torch.manual_seed(seed*100+case)
        m64,shape,out=make_model(rng); m64=m64.double(); m32=copy.deepcopy(m64).float().to(dev)
        loss=nn.MSELoss()
        n=int(rng.integers(2,9)); X=torch.rand(n,*shape,dtype=torch.float64); y=torch.rand(n,out,dtype=torch.float64)
        d64=[(X,y)]; d32=[(X.float().to(dev),y.float().to(dev))]
        G64=C.GGNLinearOperator(m64,loss,dict(m64.named_parameters()),d64,check_deterministic=False)
        G32=C.GGNLinearOperator(m32,loss,dict(m32.named_parameters()),d32,check_deterministic=False)
        D=G64.shape[1]
        V=torch.rand(D,2,dtype=torch.float64)-0.5
        lam=float((G64@V).abs().max()/V.abs().max())
        delta=0.1*lam+1e-3
        try:
            A64=G64+C.DiagonalLinearOperator.identity_like(G64,delta)
            A32=G32+C.DiagonalLinearOperator.identity_like(G32,delta)
            K32=C.KFACLinearOperator(m32,loss,dict(m32.named_parameters()),d32,fisher_type="type-2",check_deterministic=False)
            P32=K32.inverse(damping=delta, use_exact_damping=True)
            ref=C.CGInverseLinearOperator(A64,max_iter=2000,tolerance=1e-12)@V
            got=C.CGInverseLinearOperator(A32,max_iter=2000,tolerance=1e-7)@V.float().to(dev)
            gotp=C.CGInverseLinearOperator(A32,max_iter=2000,tolerance=1e-7,preconditioner=P32.__matmul__)@V.float().to(dev)
            for name,g in (("cg",got),("pcg",gotp)):
                e=rel(g,ref); worst=max(worst,e)
                if not e<2e-3: fails.append((seed,case,name,e))
        except Exception as ex:
            fails.append((seed,case,repr(ex)[:200]))
for f in fails: print(f)
print("done worst",worst,len(fails),"failures")
