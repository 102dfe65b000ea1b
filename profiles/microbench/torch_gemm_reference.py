import torch
dev="cuda:0"
R,V,H=6368,20001,1024
def t(f,n=10):
    for _ in range(3): f()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)*1e3/n
O=torch.randn(R,H,device=dev,dtype=torch.bfloat16); W=torch.randn(V,H,device=dev,dtype=torch.bfloat16)
dl=torch.randn(R,V,device=dev,dtype=torch.bfloat16)
GF=2.0*R*V*H
for name,f in [("logits bf16 out", lambda: O@W.t()), ("dO", lambda: dl@W), ("dW", lambda: dl.t()@O)]:
    us=t(f); print("%-16s %8.1f us %7.1f TF"%(name,us,GF/us/1e6))
out32=torch.empty(R,V,device=dev)
try:
    us=t(lambda: torch.mm(O.float(), W.float().t(), out=out32)); print("f32 torch logits %8.1f us %7.1f TF"%(us,GF/us/1e6))
except Exception as e: print(e)
