"""Plan.forward (one launch per stage) vs Plan.forward_chain (xrl_mlp_chain_fwd) on acting-size batches: graph of 64 passes."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xuance_amd import ops
from xuance_amd.nets import ActorCriticNet, Plan

def timed(fn, reps=20, inner=64):
    fn(); torch.cuda.synchronize()
    g = ops.Graph()
    with g:
        for _ in range(inner):
            fn()
    for _ in range(3):
        g.launch()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.launch()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps / inner * 1e6

out = []
for D, A, M in ((6, 3, 512), (6, 3, 64), (4, 2, 512), (6, 3, 2048)):
    net = ActorCriticNet(D, A, "categorical", (128,), (128,), (128,), "leaky_relu")
    X = torch.randn(M, D, device="cuda")
    items = [(net.plan, X, D, M, None)]
    rec = dict(D=D, A=A, M=M, per_stage_us=round(timed(lambda: Plan.forward_many(items)), 2),
               chain_us=round(timed(lambda: Plan.forward_chain(items)), 2))
    from xuance_amd import _lib
    st = torch.zeros(16, dtype=torch.int64, device="cuda")
    Plan.forward_chain(items); torch.cuda.synchronize()
    _lib.call("xrl_debug_mlp_chain_stamps", st.data_ptr())
    Plan.forward_chain(items); torch.cuda.synchronize()
    _lib.call("xrl_debug_mlp_chain_stamps", None)
    d = st.cpu().numpy()
    rec["stamp_cycles"] = [int(x) for x in (d[1:int(d[15])] - d[0:int(d[15]) - 1])]
    print(json.dumps(rec)); out.append(rec)
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
