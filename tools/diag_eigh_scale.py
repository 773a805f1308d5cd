import sys; sys.path.insert(0,'/root/repo')
import numpy as np, torch, ctypes
from curvlinops_amd import _hip
import _rocsolver  # tools/_rocsolver.py
dev=torch.device("cuda:0")
rng=np.random.default_rng(0)
n=65
Q,_=np.linalg.qr(rng.standard_normal((n,n))); lam=rng.standard_normal(n)
A0=(Q*lam)@Q.T
for p in (0,-3,-5,-6):
    A64=A0*10.0**p; A64=0.5*(A64+A64.T)
    A=torch.as_tensor(A64,dtype=torch.float32,device=dev)
    ld=(n+3)//4*4
    W=torch.zeros(n,ld,device=dev); W[:,:n]=A
    D,E,tau=_hip.sytrd_(W,n)
    T=np.diag(D.double().cpu().numpy())+np.diag(E[:n-1].double().cpu().numpy(),1)+np.diag(E[:n-1].double().cpu().numpy(),-1)
    ref=np.linalg.eigvalsh(A.double().cpu().numpy())
    e_tri=np.abs(np.linalg.eigvalsh(T)-ref).max()/np.abs(ref).max()
    d0,e0=D.clone(),E.clone()
    Z=torch.empty(n,ld,device=dev)
    info=_rocsolver.stedc_(D,E,Z,n)
    e_stedc=np.abs(D.double().cpu().numpy()-np.linalg.eigvalsh(T)).max()/np.abs(ref).max()
    lt=torch.linalg.eigvalsh(A).double().cpu().numpy()
    e_torch=np.abs(lt-ref).max()/np.abs(ref).max()
    print(f"scale 1e{p}: tridiagonal spectrum err {e_tri:.1e} | stedc vs tridiagonal {e_stedc:.1e} info {int(info)} | torch.linalg.eigvalsh err {e_torch:.1e}")
