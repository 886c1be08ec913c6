import torch, gaot_3d_amd
from gaot_3d_amd import functional as GF
dev="cuda"
torch.manual_seed(0)
import sys
M,K,NS=int(sys.argv[1]),int(sys.argv[2]),[int(v) for v in sys.argv[3].split(',')]
x=torch.randn(M,K,device=dev,requires_grad=True)
ws=[torch.nn.Parameter(torch.randn(n,K,device=dev)*0.05) for n in NS]
ref=torch.cat([x@w.t() for w in ws],1)
out0=GF.multi_linear(x,ws)
print("unfused err",(out0-ref).abs().max().item())
GF.colocate(ws)
print("adjacent",GF._adjacent([w.data for w in ws]))
ref2=torch.cat([x@w.t() for w in ws],1)
print("values preserved",(ref2-ref).abs().max().item())
out1=GF.multi_linear(x,ws)
print("fused err",(out1-ref).abs().max().item())
g=torch.randn_like(out1)
out1.backward(g)
gx=x.grad.clone(); x.grad=None
gw=[w.grad.clone() for w in ws]
for w in ws: w.grad=None
ref2.backward(g)
print("dx err",(gx-x.grad).abs().max().item(), [ (a-w.grad).abs().max().item() for a,w in zip(gw,ws)])
