"""Scale sweep of the HBM-bound buffer kernels and the fp32-MFMA GEMM (SURVEY.md section 8d asks for it next to the
C2 numbers, which are far below one launch's worth of roofline work).  Prints achieved GB/s or TFLOP/s per size,
timed with HIP events over back-to-back launches; algorithmic bytes: GAE 20 B/transition, store 40 B/env-step,
gather 32 B/sample."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xuance_amd import ops


def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps     # us


out = []
for n_envs, T in [(256, 256), (4096, 256), (65536, 256), (65536, 2048)]:
    g = torch.Generator(device="cuda").manual_seed(0)
    rew, val, boot = (torch.randn(T, n_envs, device="cuda", generator=g) for _ in range(3))
    term = (torch.rand(T, n_envs, device="cuda", generator=g) < 0.02).float()
    seg = (term > 0).to(torch.uint8) * 3
    seg[T - 1] = 1
    adv, ret = torch.zeros_like(rew), torch.zeros_like(rew)
    us = timed(lambda: ops.gae_scan(rew, val, term, boot, seg, adv, ret, 0.98, 0.95, True))
    out.append(dict(kernel="gae_scan", n_envs=n_envs, T=T, us=round(us, 2), algorithmic_MB=20 * n_envs * T / 1e6,
                    GBps=round(20 * n_envs * T / us / 1e3, 1)))
    del rew, val, boot, term, seg, adv, ret
for n_envs, D in [(256, 4), (65536, 4), (65536, 17), (64, 28224 // 4)]:
    T = 8
    field = torch.zeros(T, n_envs, D, device="cuda")
    step = torch.randn(n_envs, D, device="cuda")
    us = timed(lambda: ops.soa_store_step([(field, step, D * 4)], n_envs, 3))
    out.append(dict(kernel="soa_store_step", n_envs=n_envs, row_bytes=D * 4, us=round(us, 2),
                    GBps=round(2 * n_envs * D * 4 / us / 1e3, 1)))
for bs, n_envs, T, D in [(8192, 256, 256, 4), (1 << 20, 65536, 64, 4), (1 << 20, 65536, 64, 16)]:
    field = torch.randn(T, n_envs, D, device="cuda")
    idx = torch.randint(0, n_envs * T, (bs,), device="cuda")
    dst = torch.zeros(bs, D, device="cuda")
    us = timed(lambda: ops.soa_gather([(dst, field, D * 4)], idx, n_envs, T))
    out.append(dict(kernel="soa_gather", bs=bs, row_bytes=D * 4, us=round(us, 2), GBps=round(bs * (2 * D * 4 + 8) / us / 1e3, 1)))
for M, N, K in [(8192, 256, 128), (65536, 256, 128), (65536, 256, 256), (262144, 512, 512)]:
    A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda"); b = torch.zeros(N, device="cuda")
    C = torch.zeros(M, N, device="cuda")
    d = [ops.gemm_desc(A.data_ptr(), W.data_ptr(), C.data_ptr(), M, N, K, K, K, N, bias=b.data_ptr(), act="relu")]
    us = timed(lambda: ops.linear_fwd(d))
    out.append(dict(kernel="gemm_f32_kernel<NT>", M=M, N=N, K=K, us=round(us, 2), TFLOPs=round(2.0 * M * N * K / us / 1e6, 2),
                    frac_of_157=round(2.0 * M * N * K / us / 1e6 / 157.3, 3)))
    del A, W, C
for o in out:
    print(json.dumps(o))
