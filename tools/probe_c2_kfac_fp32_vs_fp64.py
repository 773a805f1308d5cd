import sys, copy, warnings, time
sys.path.insert(0,'/root/repo')
import torch
from torch import nn
import curvlinops_amd as C
warnings.simplefilter("ignore")
dev=torch.device("cuda:0"); torch.manual_seed(0)
m32=nn.Sequential(nn.Linear(1024,2688),nn.ReLU(),nn.Linear(2688,2688),nn.ReLU(),nn.Linear(2688,10)).to(dev); m64=copy.deepcopy(m32).double()
X=torch.rand(64,1024,device=dev); y=torch.rand(64,10,device=dev)
def rel(a,b): return float((a.double()-b.double()).abs().max()/b.double().abs().max())
for ft in ("type-2","empirical"):
  for cls in (C.KFACLinearOperator,C.EKFACLinearOperator):
    kw=dict(fisher_type=ft,separate_weight_and_bias=False,check_deterministic=False)
    K32=cls(m32,nn.MSELoss(),dict(m32.named_parameters()),[(X,y)],**kw); K64=cls(m64,nn.MSELoss(),dict(m64.named_parameters()),[(X.double(),y.double())],**kw)
    v=torch.rand(K32.shape[1],2,device=dev)-0.5
    print(ft,cls.__name__,"@V",rel(K32@v,K64@v.double()),"inv",rel(K32.inverse(damping=1e-3)@v,K64.inverse(damping=1e-3)@v.double()),flush=True)
