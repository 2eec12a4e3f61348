"""Scratch: where the implicit-GEMM convolution launches of a DQN update (Atari shapes, batch 32: eval rows | target rows) spend
their time: HIP-event time of each launch alone and inside a back-to-back train, per-workgroup real-time start / end (skew,
slowest workgroup, tail) and shader-clock phase marks of every workgroup (median)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from xuance_amd import ops
from xuance_amd.nets import DeepQCNN

M = 32
torch.manual_seed(0)
net = DeepQCNN((84, 84, 4), 4)
X = torch.randint(0, 256, (3 * M, 84 * 84 * 4), dtype=torch.uint8, device="cuda")
q = net.forward_pair(X, M, False)
net.d_out[:M, :4].normal_()
slabs = torch.zeros(4, net.params.P, device="cuda")
net.backward(X, M, slabs, 4)
torch.cuda.synchronize()
cs, ws = net.conv, net._ws
img_e, _ = cs.images(None)
img_t, _ = cs.images(net.target_flat, with_dx=False)


def event_us(fn, n=50):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


launches = []
xe, xt = X, X[M:]
for i, (H, W, C, k, s, p, OH, OW, F) in enumerate(cs.geo):
    groups = [cs._fwd_group(i, xe, M, img_e, None, ws.y[i]), cs._fwd_group(i, xt, M, img_t, net.target_flat, ws.y[i][M * OH * OW:])]
    launches.append((f"fwd layer {i}", groups, cs._k_split(2 * M * OH * OW), 2 * M * OH * OW))
    xe, xt = ws.y[i], ws.y[i][M * OH * OW:]
for i in (2, 1):
    H, W, C, k, s, p, OH, OW, F = cs.geo[i]
    groups = [ops.conv_desc(img=ws.dy[i], w=img_e.data_ptr() + 4 * c["off"], mask=ws.y[i - 1], out=ws.dy[i - 1], B=M, IH=OH, IW=OW, C=F,
                            Th=c["Th"], Tw=c["Tw"], nh=c["nh"], nw=c["nw"], sh=1, off_h=c["off_h"], off_w=c["off_w"], so=s, ph=c["ph"],
                            pw=c["pw"], OHt=H, OWt=W, N=C, act=0, img_u8=0) for c in cs._dx[i]]
    rows = M * sum(c["nh"] * c["nw"] for c in cs._dx[i])
    launches.append((f"dX layer {i}", groups, cs._k_split(rows), rows))

for name, groups, ks0, rows in launches:
    for ks in sorted({ks0, 1, 2, 4, 8}):
        us = event_us(lambda: ops.conv_fwd(groups, ks))
        mark = " <- used" if ks == ks0 else ""
        if ks != ks0:
            print(f"  {name}: k_split {ks}: {us:.2f} us")
            continue
        dbg = torch.zeros(8 * 4096, dtype=torch.int64, device="cuda")
        for _ in range(3):
            dbg.zero_()
            ops.conv_fwd(groups, ks, dbg=dbg)
            torch.cuda.synchronize()
        d = dbg.cpu().numpy().reshape(-1, 8)
        d = d[d[:, 0] > 0]
        t = d[:, :2].astype(np.float64) * 10e-3
        t0 = t[:, 0].min()
        ph = d[:, 3:7] - d[:, 2:3]
        print(f"{name}: k_split {ks}{mark}: {us:.2f} us back to back; {len(d)} workgroups; start skew max {(t[:, 0] - t0).max():.2f} us; "
              f"duration min {(t[:, 1] - t[:, 0]).min():.2f} median {np.median(t[:, 1] - t[:, 0]):.2f} max {(t[:, 1] - t[:, 0]).max():.2f}; "
              f"last end {(t[:, 1] - t0).max():.2f} us; phase marks (shader cycles, median over workgroups): loads issued {np.median(ph[:, 0]):.0f}, "
              f"loop done {np.median(ph[:, 1]):.0f}, reduced {np.median(ph[:, 2]):.0f}, stored {np.median(ph[:, 3]):.0f}")

# weight gradients: the two launches of a backward pass (uint8 first layer; float32 layers 1, 2)
cslab = torch.zeros(32, net.params.P, device="cuda")
P = net.params
for sel, name in ((lambda i: i == 0, "dW layer 0 (uint8)"), (lambda i: i > 0, "dW layers 1, 2")):
    wg = []
    for i, (H, W, C, k, s, p, OH, OW, F) in enumerate(cs.geo):
        if not sel(i):
            continue
        x = ws.y[i - 1] if i > 0 else X
        n = cs.names[i]
        wg.append(ops.conv_desc(img=x, dy=ws.dy[i], out=cslab.data_ptr() + 4 * P.offsets[n + ".weight"],
                                dbias=cslab.data_ptr() + 4 * P.offsets[n + ".bias"], B=M, IH=H, IW=W, C=C, Th=k, Tw=k, nh=OH, nw=OW, sh=s,
                                off_h=-p, off_w=-p, so=1, OHt=OH, OWt=OW, N=F, img_u8=int(x.dtype == torch.uint8),
                                pad=cs._dw_splits(M, 32)[i]))
    us = event_us(lambda: ops.conv_bwd_weight(wg, 32, P.P))
    dbg = torch.zeros(8 * 8192, dtype=torch.int64, device="cuda")
    for _ in range(3):
        dbg.zero_()
        ops.conv_bwd_weight(wg, 32, P.P, dbg=dbg)
        torch.cuda.synchronize()
    d = dbg.cpu().numpy().reshape(-1, 8)
    d = d[d[:, 1] > 0]
    t = d[:, :2].astype(np.float64) * 10e-3
    t0 = t[:, 0].min()
    ph = d[:, 3:6] - d[:, 2:3]
    print(f"{name}: {us:.2f} us back to back; {len(d)} working workgroups; start skew max {(t[:, 0] - t0).max():.2f} us; duration min "
          f"{(t[:, 1] - t[:, 0]).min():.2f} median {np.median(t[:, 1] - t[:, 0]):.2f} max {(t[:, 1] - t[:, 0]).max():.2f}; last end "
          f"{(t[:, 1] - t0).max():.2f} us; marks (cycles, median): first batch requested {np.median(ph[:, 0]):.0f}, loop done "
          f"{np.median(ph[:, 1]):.0f}, stored {np.median(ph[:, 2]):.0f}")
