"""Host side of the implicit-GEMM convolutions (nets.ConvStack._build_image_maps; csrc/conv_mfma.hip): the index maps and the
group geometry the host hands to xrl_conv_fwd are evaluated here with a NumPy statement of the entry point's contract
(include/xrl_hip.h: xrl_conv_t) and compared with torch's conv2d forward / input gradient on the reference's layer shapes
(layers.py:36-65: Conv2d(k, s, pad=(k-s)//2)).  No GPU: the kernels themselves are compared with the im2col path in
tests/test_gpu_dqn_qmix.py."""
import numpy as np
import torch

from xuance_amd.nets import ConvStack


class _Params:
    def __init__(self, specs):
        self.offsets, self.shapes, o = {}, {}, 0
        for n, sh in specs:
            self.offsets[n], self.shapes[n] = o, sh
            o += int(np.prod(sh))
        self.P, self.device = o, "cpu"
        g = torch.Generator().manual_seed(0)
        self.flat = torch.randn(o, generator=g)


def _unfragment(img, N, Kp):
    n, kp = ConvStack._frag_index(N, Kp)
    w = np.zeros((N, Kp), np.float64)
    w[n, kp] = img
    return w


def _contract(img, w, B, IH, IW, C, Th, Tw, nh, nw, sh, off_h, off_w, so, ph, pw, OHt, OWt, N, out):
    """out[b][hh*so+ph][ww*so+pw][n] = sum img[b][hh*sh+off_h+th][ww*sh+off_w+tw][c] * w[n][(th, tw, c)]"""
    for hh in range(nh):
        for ww in range(nw):
            patch = np.zeros((B, Th, Tw, C))
            for th in range(Th):
                for tw in range(Tw):
                    ih, iw = hh * sh + off_h + th, ww * sh + off_w + tw
                    if 0 <= ih < IH and 0 <= iw < IW:
                        patch[:, th, tw] = img[:, ih, iw]
            out[:, hh * so + ph, ww * so + pw] = patch.reshape(B, -1) @ w.T


def test_image_maps_and_class_geometry_vs_torch_conv():
    shapes = [(20, 20, 4), (28, 24, 4)]
    for obs in shapes:
        specs = [("c0.weight", (32, 4, 8, 8)), ("c0.bias", (32,)), ("c1.weight", (64, 32, 4, 4)), ("c1.bias", (64,)),
                 ("c2.weight", (64, 64, 3, 3)), ("c2.bias", (64,))]
        P = _Params(specs)
        cs = ConvStack(P, ["c0", "c1", "c2"], obs, (8, 4, 3), (4, 2, 1), (32, 64, 64))
        assert cs.implicit
        image = P.flat.numpy().astype(np.float64)[cs._map.numpy()]
        B = 2
        x = torch.randn(B, obs[2], obs[0], obs[1], dtype=torch.float64)
        for i, (H, W, C, k, s, p, OH, OW, F) in enumerate(cs.geo):
            wt = P.flat[P.offsets[f"c{i}.weight"]:P.offsets[f"c{i}.weight"] + F * C * k * k].reshape(F, C, k, k).double()
            xin = x.clone().requires_grad_(True)
            y = torch.nn.functional.conv2d(xin, wt, stride=s, padding=p)
            assert y.shape[2:] == (OH, OW)
            # forward
            wf = _unfragment(image[cs._fwd_off[i]:cs._fwd_off[i] + F * k * k * C], F, k * k * C)
            out = np.zeros((B, OH, OW, F))
            _contract(xin.detach().permute(0, 2, 3, 1).numpy(), wf, B, H, W, C, k, k, OH, OW, s, -p, -p, 1, 0, 0, OH, OW, F, out)
            np.testing.assert_allclose(out, y.detach().permute(0, 2, 3, 1).numpy(), rtol=1e-10, atol=1e-10)
            # input gradient, one group per residue class; the classes tile the input exactly once
            if i > 0:
                dy = torch.randn_like(y)
                y.backward(dy)
                dx = np.full((B, H, W, C), np.nan)
                for c in cs._dx[i]:
                    wc = _unfragment(image[c["off"]:c["off"] + C * c["Th"] * c["Tw"] * F], C, c["Th"] * c["Tw"] * F)
                    _contract(dy.permute(0, 2, 3, 1).numpy(), wc, B, OH, OW, F, c["Th"], c["Tw"], c["nh"], c["nw"], 1, c["off_h"],
                              c["off_w"], s, c["ph"], c["pw"], H, W, C, dx)
                assert not np.isnan(dx).any()
                np.testing.assert_allclose(dx, xin.grad.permute(0, 2, 3, 1).numpy(), rtol=1e-10, atol=1e-10)
            x = torch.relu(y.detach())


def test_shapes_outside_the_kernel_limits_take_the_im2col_path():
    P = _Params([("c0.weight", (16, 3, 3, 3)), ("c0.bias", (16,))])
    assert not ConvStack(P, ["c0"], (32, 32, 3), (3,), (1,), (16,)).implicit


def test_inverse_image_maps_invert_the_pack_maps():
    """ConvStack.inverse_maps (what xrl_reduce_adam's mirrors take: parameter -> image position) against the pack map
    (xrl_gather_images: image position -> parameter): every weight of every layer sits exactly once in the forward section, every
    weight of layers 1.. exactly once in the input-gradient section, biases and dense parameters nowhere."""
    from xuance_amd.nets import DeepQCNN
    net = DeepQCNN((84, 84, 4), 4, device="cpu")
    cs = net.conv
    inv_f, inv_d = (t.numpy() for t in cs.inverse_maps())
    m = cs._map.numpy()
    for inv, lo, hi in ((inv_f, 0, cs._n_fwd), (inv_d, cs._n_fwd, cs._n_img)):
        pos = np.nonzero(inv >= 0)[0]
        assert np.array_equal(m[inv[pos]], pos) and ((inv[pos] >= lo) & (inv[pos] < hi)).all()
        assert np.array_equal(np.sort(inv[pos]), np.arange(lo, hi)[m[lo:hi] >= 0])
    P = net.params
    for i, name in enumerate(cs.names):
        o, n = P.offsets[name + ".weight"], int(np.prod(P.shapes[name + ".weight"]))
        assert (inv_f[o:o + n] >= 0).all() and (inv_d[o:o + n] >= 0).all() == (i > 0)
        ob = P.offsets[name + ".bias"]
        assert (inv_f[ob:ob + P.shapes[name + ".bias"][0]] < 0).all()
    dense = P.offsets["eval_Q_head.q_value.0.weight"]
    assert (inv_f[dense:] < 0).all() and (inv_d[dense:] < 0).all()
