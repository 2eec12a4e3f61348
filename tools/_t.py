import torch, sys
sys.path.insert(0, "/root/repo")
from xuance_amd import ops
dev="cuda"
def t(fn, reps=200):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)*1e3/reps
for P in (34052, 142608):
    params=torch.randn(P,device=dev); grad=torch.zeros(P,device=dev); m=torch.zeros(P,device=dev); v=torch.zeros(P,device=dev)
    state=ops.adam_state_tensor(4e-4, 1000, device=dev)
    sumsq=torch.zeros(1024,dtype=torch.float64,device=dev)
    sync=torch.zeros(4+(P+255)//256+8,dtype=torch.int32,device=dev)
    for S in (1, 32, 128, 256):
        slabs=torch.randn(S,P,device=dev)*1e-3
        for clip in (0.5, 0.0):
            us=t(lambda: ops.reduce_adam(slabs,S,P,params,grad,m,v,P,state,sumsq,clip,[],sync))
            print("P",P,"n_split",S,"clip",clip,"us",round(us,2), "GB/s", round(S*P*4/us/1e3,1))
        us=t(lambda: ops.grad_reduce(slabs,S,P,P,grad,sumsq))
        print("   grad_reduce alone us", round(us,2), "GB/s", round(S*P*4/us/1e3,1))
