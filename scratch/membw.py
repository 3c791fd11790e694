import torch, time
dev=torch.device('cuda',0)
n=6*2**30//8
a=torch.empty(n,dtype=torch.float64,device=dev)
b=torch.empty(n//2,dtype=torch.float64,device=dev)
c=torch.empty(n//2,dtype=torch.float64,device=dev)
def t(f,reps=20):
    for _ in range(5): f()
    torch.cuda.synchronize(); s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/reps
ms=t(lambda: a.fill_(1.0)); print("fill 6GiB: %.3f ms  %.2f TB/s write"%(ms, a.numel()*8/ms/1e9))
ms=t(lambda: c.copy_(b)); print("copy 3GiB->3GiB: %.3f ms  %.2f TB/s (r+w)"%(ms, 2*b.numel()*8/ms/1e9))
ms=t(lambda: a.sum()); print("sum 6GiB: %.3f ms  %.2f TB/s read"%(ms, a.numel()*8/ms/1e9))
ms=t(lambda: a.fill_(1.0)); print("fill 6GiB: %.3f ms  %.2f TB/s write"%(ms, a.numel()*8/ms/1e9))
