import torch, sys
sys.path.insert(0, "/root/repo")
from xuance_amd import ops
P=34052
dev="cuda"
def tg(fn, reps=64):
    fn(); torch.cuda.synchronize()
    g=ops.Graph()
    with g:
        for _ in range(reps): fn()
    best=1e9
    for _ in range(5):
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record(); g.launch(); e1.record(); torch.cuda.synchronize()
        best=min(best,e0.elapsed_time(e1)*1e3/reps)
    return best
params=torch.randn(P,device=dev); grad=torch.zeros(P,device=dev); m=torch.zeros(P,device=dev); v=torch.zeros(P,device=dev)
state=ops.adam_state_tensor(4e-4, 1000, device=dev)
sumsq=torch.zeros(256,dtype=torch.float64,device=dev)
sync=torch.zeros(4+(P+255)//256+8,dtype=torch.int32,device=dev)
for S in (1, 4, 8, 32, 128):
    slabs=torch.randn(S,P+640,device=dev)*1e-3
    us=tg(lambda: ops.reduce_adam(slabs,S,P+640,params,grad,m,v,P,state,sumsq,0.5,[],sync))
    print("n_split",S,"reduce_adam in a graph of 64: us",round(us,2))
