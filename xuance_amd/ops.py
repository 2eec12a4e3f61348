"""Thin, allocation-free Python wrappers over the C ABI (include/xrl_hip.h).

Every function takes torch CUDA tensors (device memory owned by PyTorch-ROCm), passes raw device pointers and the
current HIP stream to libxrl_hip.so, and returns nothing: outputs are written into caller-provided tensors, so all
of them may be recorded into a hipGraph (xuance_amd.graph.Graph).
"""
import ctypes as C

import torch

from . import _lib
from ._lib import XrlError  # noqa: F401  (re-exported)
from ._lib import (ActTail, MlpChain, Conv, ImageJob, DqnHeadTd, DqnTailTd, DqnActTail, Classic, Exchange, MarlGate, SynthMarl, SynthFrames, LstmFwd, LstmBwd, EpisodeField, GruFwd, GruBwd, Mirrors, RolloutRun, RolloutWide, SynthCtl, ACT, Field, Gemm, PpoLoss, AdamState, Rms, Sample, CartPole, PostStep, EGreedy, DqnTd, Qmix, RolloutStep, FusedLayer, PpoFused, OptChain, CHAIN_SYNC_WORDS, MarlAct, call, ptr,
                   stream_ptr)


def _chk(t, dtype=torch.float32):
    assert t.is_cuda and t.is_contiguous() and t.dtype == dtype, (t.device, t.is_contiguous(), t.dtype)
    return t


def device_info():
    cu, ws, arch = C.c_int(), C.c_int(), C.create_string_buffer(64)
    call("xrl_device_info", C.byref(cu), C.byref(ws), arch, 64)
    return dict(cu_count=cu.value, wave_size=ws.value, arch=arch.value.decode())


# ------------------------------------------------------------------------------------------ buffer
def _fields(pairs, flags=None):
    arr = (Field * len(pairs))()
    for i, (dst, src, row_bytes) in enumerate(pairs):
        arr[i].dst, arr[i].src, arr[i].row_bytes = ptr(dst), ptr(src), int(row_bytes)
        arr[i].flags = 0 if flags is None else int(flags[i])
    return arr


def soa_store_step(pairs, n_envs, t, size_dev=None, new_size=0):
    """pairs: [(field_tensor [T,n_envs,...], step_tensor [n_envs,...], row_bytes)]; size_dev: int32 [1] tensor that receives
    new_size in the same launch (a replay ring's filled-slot count)."""
    arr = _fields(pairs)
    if size_dev is not None:
        call("xrl_soa_store_step_sized", arr, len(pairs), int(n_envs), int(t), ptr(size_dev), int(new_size), stream_ptr())
    else:
        call("xrl_soa_store_step", arr, len(pairs), int(n_envs), int(t), stream_ptr())


def soa_store_step_ring(pairs, n_envs, n_size, slot_bias, size_bias, counter_dev, offset, size_dev):
    """soa_store_step with the ring slot / filled-slot count derived from a device counter (xrl_soa_store_step_ring): slot =
    (slot_bias + *counter_dev + offset) mod n_size, *size_dev = min(size_bias + *counter_dev + offset + 1, n_size)."""
    _chk(counter_dev, torch.int32)
    arr = _fields(pairs)
    call("xrl_soa_store_step_ring", arr, len(pairs), int(n_envs), int(n_size), int(slot_bias), int(size_bias), ptr(counter_dev),
         int(offset), ptr(size_dev), stream_ptr())


def soa_gather(pairs, idx, n_envs, T, stats=None, flags=None):
    """pairs: [(dst [bs,...], field [T,n_envs,...], row_bytes)]; idx int64 env-major flat indices."""
    _chk(idx, torch.int64)
    arr = _fields(pairs, flags)
    call("xrl_soa_gather", arr, len(pairs), ptr(idx), idx.numel(), int(n_envs), int(T), ptr(stats), stream_ptr())


def soa_gather_sampled(pairs, idx_out, n_envs, n_size, size_dev, seed, counter=0, counter_dev=None):
    """sample_replay_indices + soa_gather in one launch (bs = idx_out.numel() <= 256); idx_out receives the drawn rows."""
    _chk(idx_out, torch.int64)
    arr = _fields(pairs, None)
    call("xrl_soa_gather_sampled", arr, len(pairs), ptr(idx_out), idx_out.numel(), int(n_envs), int(n_size), ptr(size_dev),
         int(seed), int(counter), ptr(counter_dev), stream_ptr())


def adv_stats(adv_field, idx, bs, n_batches, n_envs, T, stats):
    call("xrl_adv_stats", ptr(_chk(adv_field)), ptr(_chk(idx, torch.int64)), int(bs), int(n_batches), int(n_envs),
         int(T), ptr(_chk(stats)), stream_ptr())


def gae_scan(rew, val, term, bootv, seg, adv, ret, gamma, lam, use_gae=True):
    T, n_envs = rew.shape
    for t in (rew, val, term, bootv, adv, ret):
        _chk(t)
    _chk(seg, torch.uint8)
    call("xrl_gae_scan", ptr(rew), ptr(val), ptr(term), ptr(bootv), ptr(seg), ptr(adv), ptr(ret), n_envs, T,
         float(gamma), float(lam), int(bool(use_gae)), stream_ptr())


# ------------------------------------------------------------------------------------------ dense layers
def gemm_desc(A, B, Cm, M, N, K, lda, ldb, ldc, bias=None, dbias=None, aux=None, ldaux=0, act=None):
    g = Gemm()
    g.A, g.B, g.C = A, B, Cm
    g.bias, g.dbias, g.aux = bias, dbias, aux
    g.M, g.N, g.K, g.lda, g.ldb, g.ldc, g.ldaux = M, N, K, lda, ldb, ldc, ldaux
    g.act = ACT[act] if not isinstance(act, int) else act
    return g


def _garr(groups):
    arr = (Gemm * len(groups))()
    for i, g in enumerate(groups):
        arr[i] = g
    return arr


def linear_fwd(groups):
    call("xrl_linear_fwd", _garr(groups), len(groups), stream_ptr())


def linear_fwd_partials(groups):
    """xrl_linear_fwd without the split-K epilogue: every group leaves its raw partial sums in its workspace (aux)."""
    call("xrl_linear_fwd_partials", _garr(groups), len(groups), stream_ptr())


def ppo_act_tail(post=None, **kw):
    """xrl_ppo_act_tail: the hidden layer's split-K epilogue, logits + value, sampling / log-prob / value / bootstrap value, and --
    post: xrl_rollout_poststep's keyword arguments of the PREVIOUS vector step -- that step's bookkeeping, and the copy of the observations
    into their buffer slot (copy_src / copy_dst / copy_bytes), in one launch."""
    from ._lib import PpoActTail
    p = _struct(PpoActTail, kw)
    if post is not None:
        for k, v in post.items():
            if isinstance(v, torch.Tensor):
                v = v.data_ptr()
            setattr(p.post, k, v)
        p.post_n = int(post["n"])
    call("xrl_ppo_act_tail", C.byref(p), stream_ptr())


def linear_bwd_data(groups):
    call("xrl_linear_bwd_data", _garr(groups), len(groups), stream_ptr())


def linear_bwd_weight(groups, n_split, slab_stride):
    call("xrl_linear_bwd_weight", _garr(groups), len(groups), int(n_split), int(slab_stride), stream_ptr())


# ------------------------------------------------------------------------------------------ convolution helpers
def im2col_nhwc(x, col, B, H, W, C, k, s, p):
    call("xrl_im2col_nhwc", ptr(x), int(x.dtype == torch.uint8), ptr(col), B, H, W, C, k, s, p, stream_ptr())


def col2im_nhwc(dcol, xact, dx, B, H, W, C, k, s, p):
    call("xrl_col2im_nhwc", ptr(dcol), ptr(xact), ptr(dx), B, H, W, C, k, s, p, stream_ptr())


def maxpool_hw_fwd(y, feat, argmax, B, P, F, ld_feat):
    call("xrl_maxpool_hw_fwd", ptr(y), ptr(feat), ptr(argmax), B, P, F, ld_feat, stream_ptr())


def maxpool_hw_bwd(dfeat, argmax, y, dy, B, P, F, ld_dfeat):
    call("xrl_maxpool_hw_bwd", ptr(dfeat), ptr(argmax), ptr(y), ptr(dy), B, P, F, ld_dfeat, stream_ptr())


def dqn_head_td(**kw):
    """Q layer + TD rule + the layer's data gradient in one launch (xrl_dqn_head_td)."""
    call("xrl_dqn_head_td", C.byref(_struct(DqnHeadTd, kw)), stream_ptr())


def dqn_tail_td(**kw):
    """Max-pool of the last convolution's output, hidden + Q layer of both networks, TD rule, d_h, d_feat and the pool's
    backward in one launch (xrl_dqn_tail_td)."""
    call("xrl_dqn_tail_td", C.byref(_struct(DqnTailTd, kw)), stream_ptr())


def dqn_act_tail(**kw):
    """Pool + hidden + Q layer + epsilon-greedy action of a DeepQCNN in one launch (xrl_dqn_act_tail)."""
    call("xrl_dqn_act_tail", C.byref(_struct(DqnActTail, kw)), stream_ptr())


def conv_desc(**kw):
    """One group of xrl_conv_fwd / xrl_conv_bwd_weight (xrl_conv_t); tensors or raw addresses for the pointer fields."""
    g = Conv()
    for k, v in kw.items():
        if isinstance(v, torch.Tensor):
            v = v.data_ptr()
        setattr(g, k, v)
    return g


def _carr(groups):
    arr = (Conv * len(groups))()
    for i, g in enumerate(groups):
        arr[i] = g
    return arr


def conv_fwd(groups, k_split, dbg=None):
    if dbg is not None:
        call("xrl_conv_fwd_probe", _carr(groups), len(groups), int(k_split), ptr(dbg), stream_ptr())
    else:
        call("xrl_conv_fwd", _carr(groups), len(groups), int(k_split), stream_ptr())


def conv_bwd_weight(groups, n_split, slab_stride, dbg=None):
    if dbg is not None:
        call("xrl_conv_bwd_weight_probe", _carr(groups), len(groups), int(n_split), int(slab_stride), ptr(dbg), stream_ptr())
    else:
        call("xrl_conv_bwd_weight", _carr(groups), len(groups), int(n_split), int(slab_stride), stream_ptr())


def gather_images(jobs):
    """jobs: [(src, map_int32, dst, n)]: dst[j] = src[map[j]] (map < 0: 0), all jobs in one launch."""
    arr = (ImageJob * len(jobs))()
    for i, (src, mp, dst, n) in enumerate(jobs):
        arr[i].src, arr[i].map, arr[i].dst, arr[i].n = ptr(src), ptr(mp), ptr(dst), int(n)
    call("xrl_gather_images", arr, len(jobs), stream_ptr())


def flatten_chw_fwd(y, feat, B, P, F, ld_feat):
    call("xrl_flatten_chw_fwd", ptr(y), ptr(feat), B, P, F, ld_feat, stream_ptr())


def flatten_chw_bwd(dfeat, y, dy, B, P, F, ld_dfeat):
    call("xrl_flatten_chw_bwd", ptr(dfeat), ptr(y), ptr(dy), B, P, F, ld_dfeat, stream_ptr())


# ------------------------------------------------------------------------------------------ PPO loss
def ppo_loss(dist, **kw):
    p = PpoLoss()
    for k, v in kw.items():
        setattr(p, k, v)
    call("xrl_ppo_loss_categorical" if dist == "categorical" else "xrl_ppo_loss_gaussian", C.byref(p), stream_ptr())


def ppokl_adapt(partials, n_split, count, kl_coef, target_kl, kl_out=None):
    """The kl_coef schedule of PPOKL_Learner.update on the device (xrl_ppokl_adapt)."""
    call("xrl_ppokl_adapt", ptr(partials), int(n_split), float(count), ptr(kl_coef), float(target_kl),
         ptr(kl_out) if kl_out is not None else None, stream_ptr())


def sum_partials(partials, n_rows, width, out):
    call("xrl_sum_partials", ptr(partials), int(n_rows), int(width), ptr(out), stream_ptr())


def sum_partials_batched(partials, n_rows, width, out, n_batches, in_stride, out_stride):
    call("xrl_sum_partials_batched", ptr(partials), int(n_rows), int(width), ptr(out), int(n_batches), int(in_stride),
         int(out_stride), stream_ptr())


# ------------------------------------------------------------------------------------------ optimiser
def adam_state_tensor(lr, total_iters, end_factor=1.0, eps=1e-5, weight_decay=0.0, device="cuda"):
    """Device-resident xrl_adam_state_t, returned as a uint8 tensor (use read_adam_state to inspect)."""
    st = AdamState(0, 0, max(int(total_iters), 1), 0, float(lr), float(end_factor), 0.9, 0.999, float(eps),
                   float(weight_decay), float(lr), 0.0)
    raw = bytes(st)
    return torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(device)


def read_adam_state(t):
    return AdamState.from_buffer_copy(bytes(t.cpu().numpy().tobytes()))


def write_adam_state(t, st):
    t.copy_(torch.frombuffer(bytearray(bytes(st)), dtype=torch.uint8))


def grad_reduce(slabs, n_split, slab_stride, P, grad, sumsq_part, fold=None):
    """fold = (fold_off, fold_len): slab columns [fold_off, fold_off + fold_len) are added onto columns [0, fold_len)."""
    if fold:
        call("xrl_grad_reduce_fold", ptr(slabs), int(n_split), int(slab_stride), int(P), ptr(grad), ptr(sumsq_part),
             sumsq_part.numel(), int(fold[0]), int(fold[1]), stream_ptr())
        return
    call("xrl_grad_reduce", ptr(slabs), int(n_split), int(slab_stride), int(P), ptr(grad), ptr(sumsq_part),
         sumsq_part.numel(), stream_ptr())


def adam_step(params, grad, m, v, P, state, sumsq_part, max_norm):
    call("xrl_adam_step", ptr(params), ptr(grad), ptr(m), ptr(v), int(P), ptr(state), ptr(sumsq_part),
         sumsq_part.numel(), float(max_norm if max_norm else 0.0), stream_ptr())


def adam_step_mirrored(params, grad, m, v, P, state, sumsq_part, max_norm, map_a, dst_a, map_b, dst_b):
    call("xrl_adam_step_mirrored", ptr(params), ptr(grad), ptr(m), ptr(v), int(P), ptr(state), ptr(sumsq_part),
         sumsq_part.numel(), float(max_norm if max_norm else 0.0), ptr(map_a), ptr(dst_a), ptr(map_b), ptr(dst_b),
         stream_ptr())


def fill_mirrors(mir, mirrors):
    """(map, dst) pairs into an xrl_mirrors_t; a triple (map, dst, plane) is a SPLIT mirror (map values <= -2 name 16-bit elements
    of a three-plane bf16 image, planes `plane` elements apart: xrl_mirrors_t.split_plane)."""
    mir.n = len(mirrors)
    for q, mm in enumerate(mirrors):
        mir.map[q] = mm[0].data_ptr(); mir.dst[q] = mm[1].data_ptr()
        if len(mm) > 2:
            assert mir.split_plane in (0, int(mm[2]))
            mir.split_plane = int(mm[2])


def adam_step_mirrors(params, grad, m, v, P, state, sumsq_part, max_norm, mirrors):
    """`mirrors`: up to 4 (int32 map [P], destination tensor) pairs refreshed in the Adam launch."""
    mir = Mirrors()
    fill_mirrors(mir, mirrors)
    call("xrl_adam_step_mirrors", ptr(params), ptr(grad), ptr(m), ptr(v), int(P), ptr(state), ptr(sumsq_part),
         sumsq_part.numel(), float(max_norm if max_norm else 0.0), C.byref(mir), stream_ptr())


def reduce_adam(slabs, n_split, slab_stride, params, grad, m, v, P, state, sumsq_part, max_norm, mirrors, sync, target=None,
                target_every=0, fold=None, exchange=None, target_image=None, tick=None, partials=None, alt=None):
    """grad_reduce + clip + Adam + mirrors (+ the periodic hard target update) in one launch (blocks meet at a counter
    barrier); same numbers.  exchange: a dist.GradientExchange -- the ranks average their gradients inside the launch.
    tick = (counter tensor, increment), partials = (float64 [rows, 8] tensor, rows, float64 [8] out): xrl_counter_add and
    xrl_sum_partials of an update phase done by one block of this launch instead of by launches of their own."""
    mir = Mirrors()
    if tick is not None:
        mir.tick, mir.tick_inc = tick[0].data_ptr(), int(tick[1])
    if partials is not None:
        mir.part, mir.part_rows, mir.part_out = partials[0].data_ptr(), int(partials[1]), partials[2].data_ptr()
    fill_mirrors(mir, mirrors)
    if target is not None and target_every > 0:
        mir.target, mir.target_every = target.data_ptr(), int(target_every)
        if target_image is not None:                       # derived layout of the target, refreshed with it (through map[0])
            mir.target_image = target_image.data_ptr()
    if fold:
        mir.fold_off, mir.fold_len = int(fold[0]), int(fold[1])
    if alt:                                                # (parts, [(lo, hi), (lo, hi)]): ranges summed over `parts` rows only
        mir.alt_split = int(alt[0])
        for i, (lo, hi) in enumerate(alt[1]):
            mir.alt_lo[i], mir.alt_hi[i] = int(lo), int(hi)
    if exchange is not None:
        assert exchange.stride4 * 4 >= P
        call("xrl_reduce_adam_exchange", ptr(slabs), int(n_split), int(slab_stride), ptr(params), ptr(grad), ptr(m), ptr(v),
             int(P), ptr(state), ptr(sumsq_part), sumsq_part.numel(), float(max_norm if max_norm else 0.0), C.byref(mir),
             ptr(sync), C.byref(exchange.struct), stream_ptr())
        return
    call("xrl_reduce_adam", ptr(slabs), int(n_split), int(slab_stride), ptr(params), ptr(grad), ptr(m), ptr(v), int(P),
         ptr(state), ptr(sumsq_part), sumsq_part.numel(), float(max_norm if max_norm else 0.0), C.byref(mir), ptr(sync),
         stream_ptr())


def reduce_adam_fits(P, with_exchange=False):
    """May xrl_reduce_adam run for P parameters here (every block of its spinning barrier resident)?  The learners take the
    two-launch sequence (xrl_grad_reduce + xrl_adam_step) when not."""
    from ._lib import load
    return bool(load().xrl_reduce_adam_fits(int(P), int(bool(with_exchange))))


XC_MAX_RANKS, XC_MAX_GROUPS, IPC_HANDLE_BYTES = 8, 1024, 64


def xc_bytes(stride4):
    return 4 * 2 * XC_MAX_GROUPS + 2 * 16 * int(stride4)


def ipc_alloc(nbytes):
    """-> (device pointer, 64-byte IPC handle) of zero-filled fine-grained device memory."""
    init_device()
    p, h = C.c_void_p(), (C.c_ubyte * IPC_HANDLE_BYTES)()
    call("xrl_ipc_alloc", C.c_size_t(int(nbytes)), C.byref(p), C.cast(h, C.c_void_p))
    return p.value, bytes(h)


def ipc_open(handle):
    h, p = (C.c_ubyte * IPC_HANDLE_BYTES).from_buffer_copy(handle), C.c_void_p()
    call("xrl_ipc_open", C.cast(h, C.c_void_p), C.byref(p))
    return p.value


def ipc_close(p):
    call("xrl_ipc_close", C.c_void_p(p))


def ipc_free(p):
    call("xrl_ipc_free", C.c_void_p(p))


def ipc_clear(p, nbytes):
    call("xrl_ipc_clear", C.c_void_p(p), C.c_size_t(int(nbytes)), stream_ptr())


def pack_mid_frags(plan, params_flat, frag):
    p = PpoFused()
    p.params = params_flat.data_ptr()
    fused_layers_from_plan(plan, p)
    call("xrl_pack_mid_frags", C.byref(p), ptr(frag), frag.numel(), stream_ptr())


FRAG16_PLANE = 2 * 256 * 128            # XRL_FRAG16_PLANE: 16-bit elements per plane of the split fragment image


def split_products_class(plan, obs_dim, action_dim, dist):
    """The class csrc/ppo_trunk_bx.hip is built for: D-128-{128-A | 128-1} with D <= 8, A <= 4 -- the CartPole headline (4, 2) as a
    compile-time instance, Acrobot (6, 3) / LunarLander (8, 4) / MountainCar (2, 3) and the Gaussian Pendulum (3, 1) through the
    any-(D, A) ones: every classic-control PPO yaml of the reference."""
    return 1 <= obs_dim <= 8 and 1 <= action_dim <= 4 and list(plan.widths) == [obs_dim, 128, 256, action_dim + 1]


def pack_mid_frags16(plan, params_flat, image):
    """image (int16 [3 * FRAG16_PLANE]) <- the branch layer as three bf16 planes in matrix-core lane order (xrl_pack_mid_frags16)."""
    p = PpoFused()
    p.params = params_flat.data_ptr()
    fused_layers_from_plan(plan, p)
    call("xrl_pack_mid_frags16", C.byref(p), ptr(image), image.numel(), stream_ptr())


def trunk_forward16(plan, params_flat, frag16, X, M, out, ld, D, A, gaussian, out_act, sample=None):
    """The acting pass of a shared-trunk network (D <= 8, A <= 4) as one launch (xrl_trunk_forward16): rows of X [M][D] -> out[m][0..A]
    (actor output | value); frag16 must hold the current branch layer (pack_mid_frags16 / the optimiser's split mirrors).  sample =
    policy_sample's keyword arguments: the launch samples actions / log-probs / values / bootstrap values itself, no head buffer."""
    p = PpoFused()
    p.params, p.frag16, p.f_obs = params_flat.data_ptr(), frag16.data_ptr(), X.data_ptr()
    p.fwd_out = out.data_ptr() if out is not None else None
    p.M, p.D, p.A, p.fwd_ld, p.dist, p.out_act = int(M), int(D), int(A), int(ld), int(bool(gaussian)), int(out_act)
    fused_layers_from_plan(plan, p)
    smp = C.byref(_struct(Sample, sample)) if sample is not None else None
    call("xrl_trunk_forward16", C.byref(p), smp, stream_ptr())


def frag16_layout_maps(plan, P, device):
    """int32 SPLIT mirror maps [P] (value -(e + 2): 16-bit element e of a plane; -1: not mirrored) of the forward / backward section of
    the split fragment image -- the index formulas of csrc/split3.h (xrl_frag16_fwd_index / _bwd_index), checked against the pack
    kernel by tests/test_gpu_ppo.py."""
    mids = [L for st in plan.stages[1:-1] for L in st]
    w_off = plan.params.offsets[mids[0].w_name]
    i = torch.arange(256 * 128, dtype=torch.int64, device=device)
    n, k = i >> 7, i & 127
    t, qq = n >> 5, k >> 4
    fwd = ((t * 8 + ((qq + t) & 7)) * 64 + (n & 31) + 32 * ((k >> 3) & 1)) * 8 + (k & 7)
    kt, q = k >> 5, n >> 4
    bwd = 256 * 128 + ((kt * 16 + ((q + kt) & 15)) * 64 + (k & 31) + 32 * ((n >> 3) & 1)) * 8 + (n & 7)
    maps = []
    for e in (fwd, bwd):
        m = torch.full((P,), -1, dtype=torch.int32, device=device)
        m[w_off:w_off + 256 * 128] = (-(e + 2)).to(torch.int32)
        maps.append(m)
    return maps


def set_split_product_tr(on):
    """Diagnostics: the split-product kernel's weight-gradient operands through the LDS transpose read (default) or 2-byte reads."""
    call("xrl_set_split_product_tr", int(bool(on)))


def set_rollout_split_products(on):
    """The CartPole class' actor rollout kernel: branch-layer products as exact 3-way bf16 splits (True) or on the float32 instruction."""
    call("xrl_set_rollout_split_products", int(bool(on)))


def set_split_product_ksplit(mode):
    """Diagnostics: the split-product kernel's wave pairs split k instead of rows in 0 no / 1 the backward-data (default) / 2 both
    weight-streamed products."""
    call("xrl_set_split_product_ksplit", int(mode))


def pack_transitions(f_obs, f_act, f_ret, f_adv, f_logp, packed, count):
    call("xrl_pack_transitions", ptr(f_obs), ptr(f_act), ptr(f_ret), ptr(f_adv), ptr(f_logp), ptr(packed), int(count),
         stream_ptr())


def gather_rows(packed, idx, out, count, n_envs, T):
    call("xrl_gather_rows", ptr(packed), ptr(idx), ptr(out), int(count), int(n_envs), int(T), stream_ptr())


def mid_frag_floats(plan):
    """2*N*K of the first middle layer when it has a fragment-ordered form (multiples of 32), else 0."""
    mids = [L for st in plan.stages[1:-1] for L in st]
    if len(mids) < 1 or mids[0].N % 32 or mids[0].K % 32:
        return 0
    return 2 * mids[0].N * mids[0].K


def frag_layout_maps(plan, P, device):
    """int32 maps param index -> index in the forward / backward section of the fragment-ordered copy (ramp inversion)."""
    nf = mid_frag_floats(plan)
    ramp = torch.arange(P, dtype=torch.float32, device=device) + 1.0
    fr = torch.zeros(nf, device=device)
    pack_mid_frags(plan, ramp, fr)
    torch.cuda.synchronize()
    maps = []
    for lo, hi in ((0, nf // 2), (nf // 2, nf)):
        m = torch.full((P,), -1, dtype=torch.int32, device=device)
        sec = fr[lo:hi]
        j = torch.nonzero(sec > 0).flatten()
        m[(sec[j] - 1.0).to(torch.int64)] = (j + lo).to(torch.int32)
        maps.append(m)
    return maps


def derived_layout_maps(plan, P, device):
    """int32 maps param index -> index in (a) the transposed-middle-weights buffer, (b) the packed cache image,
    computed by running the two layout kernels on an index ramp (so they can never disagree with the kernels)."""
    ramp = torch.arange(P, dtype=torch.float32, device=device) + 1.0          # exact in fp32 for P < 2^24
    pt = torch.zeros(P, device=device)
    img = torch.zeros(rollout_cache_floats(plan) + 16, device=device)
    transpose_mid(plan, ramp, pt)
    pack_rollout_cache(plan, ramp, img)
    torch.cuda.synchronize()

    def invert(dst):
        m = torch.full((P,), -1, dtype=torch.int32, device=device)
        j = torch.nonzero(dst > 0).flatten()
        src = (dst[j] - 1.0).to(torch.int64)
        m[src] = j.to(torch.int32)
        return m
    return invert(pt), invert(img)


# ------------------------------------------------------------------------------------------ rollout side
def _struct(cls, kw):
    p = cls()
    for k, v in kw.items():
        if isinstance(v, torch.Tensor):
            v = v.data_ptr()
        setattr(p, k, v)
    return p


def obs_normalize(**kw):
    call("xrl_obs_normalize", C.byref(_struct(Rms, kw)), stream_ptr())


def mlp_chain_desc(jobs):
    """xrl_mlp_chain_t of up to 4 jobs: dict(x, ldx, M, params (pointer), layers [dict(w_off, b_off, K, N, act, in_level, in_off,
    out_level, out_off)], level_width [..], out {level: (pointer, ld)})."""
    q = MlpChain()
    q.n_jobs = len(jobs)
    t = 0
    for j, jb in enumerate(jobs):
        J = q.job[j]
        J.x, J.ldx, J.M, J.params = int(jb["x"]), int(jb["ldx"]), int(jb["M"]), int(jb["params"])
        J.n_layers, J.n_levels = len(jb["layers"]), len(jb["level_width"])
        for l, w in enumerate(jb["level_width"]):
            J.level_width[l] = int(w)
        for l, L in enumerate(jb["layers"]):
            for k, v in L.items():
                setattr(J.layers[l], k, int(v))
        for l, (pt, ld) in jb.get("out", {}).items():
            J.out[l], J.ld_out[l] = int(pt), int(ld)
        q.tile0[j] = t
        t += (int(jb["M"]) + 31) // 32
    q.tile0[len(jobs)] = t
    return q


def mlp_chain_lds_bytes(desc):
    return int(_lib.load().xrl_mlp_chain_lds_bytes(C.byref(desc)))


def mlp_chain_fwd(desc):
    call("xrl_mlp_chain_fwd", C.byref(desc), stream_ptr())


def post_norm(post, rms):
    """rollout_poststep(**post) + obs_normalize(**rms) as one launch (xrl_post_norm)."""
    call("xrl_post_norm", C.byref(_struct(PostStep, post)), C.byref(_struct(Rms, rms)), stream_ptr())


def act_tail(sample, env_kind=0, classic=None, cartpole=None, **kw):
    """Heads + policy_sample(**sample) + the device env's step as one launch (xrl_act_tail); classic / cartpole: the keyword
    arguments of ops.classic_step / ops.cartpole_step."""
    q = _struct(ActTail, kw)
    q.env_kind = int(env_kind)
    q.sample = _struct(Sample, sample)
    if classic is not None:
        q.classic = _struct(Classic, classic)
    if cartpole is not None:
        q.cartpole = _struct(CartPole, cartpole)
    call("xrl_act_tail", C.byref(q), stream_ptr())


def policy_sample(**kw):
    call("xrl_policy_sample", C.byref(_struct(Sample, kw)), stream_ptr())


def cartpole_step(reset=False, **kw):
    call("xrl_cartpole_step", C.byref(_struct(CartPole, kw)), int(bool(reset)), stream_ptr())


def classic_step(reset=False, **kw):
    """Pendulum-v1 / MountainCar-v0 / Acrobot-v1 on the device (xrl_classic_step; kind 1 / 2 / 3)."""
    call("xrl_classic_step", C.byref(_struct(Classic, kw)), int(bool(reset)), stream_ptr())


def synth_control_step(reset=False, **kw):
    call("xrl_synth_control_step", C.byref(_struct(SynthCtl, kw)), int(bool(reset)), stream_ptr())


def synth_frames_step(reset=False, **kw):
    call("xrl_synth_frames_step", C.byref(_struct(SynthFrames, kw)), int(bool(reset)), stream_ptr())


def synth_marl_step(reset=False, **kw):
    call("xrl_synth_marl_step", C.byref(_struct(SynthMarl, kw)), int(bool(reset)), stream_ptr())


def rollout_poststep(**kw):
    call("xrl_rollout_poststep", C.byref(_struct(PostStep, kw)), stream_ptr())


def egreedy(**kw):
    call("xrl_egreedy", C.byref(_struct(EGreedy, kw)), stream_ptr())


_inited = False


def init_device():
    """xrl_init once per process (kernel attributes); must happen outside hipGraph capture."""
    global _inited
    if not _inited:
        call("xrl_init")
        _inited = True
        import os
        if os.environ.get("XRL_ROLLOUT_SPLIT_PRODUCTS"):           # diagnostics: the actor rollout kernel's branch layer on the bf16 instruction
            call("xrl_set_rollout_split_products", int(os.environ["XRL_ROLLOUT_SPLIT_PRODUCTS"]))
        if os.environ.get("XRL_SPLIT_PRODUCT_KSPLIT"):             # diagnostics (tools/, profiles/r06_q_*): see set_split_product_ksplit
            set_split_product_ksplit(int(os.environ["XRL_SPLIT_PRODUCT_KSPLIT"]))


def fused_layers_from_plan(plan, p):
    """Fill p.layers / level widths of an xrl_rollout_step_t from a nets.Plan (stage order = execution order)."""
    li = 0
    for stage in plan.stages:
        for L in stage:
            f = p.layers[li]
            f.w_off, f.b_off = plan.params.offsets[L.w_name], plan.params.offsets[L.b_name]
            f.K, f.N, f.act = L.K, L.N, ACT[L.act]
            f.in_level, f.in_off, f.out_level, f.out_off = L.in_level, L.in_off, L.out_level, L.out_off
            li += 1
    p.n_layers, p.n_levels = li, len(plan.widths)
    p.n_head_layers = len(plan.stages[-1])
    for i, w in enumerate(plan.widths):
        p.level_width[i] = w


def ppo_fused_minibatch(plan, **kw):
    p = _struct(PpoFused, kw)
    fused_layers_from_plan(plan, p)
    call("xrl_ppo_fused_minibatch", C.byref(p), stream_ptr())


def ppo_trunk_chain_fits(M, tile_rows, P):
    """May xrl_ppo_trunk_chained run a minibatch of M rows in tiles of tile_rows for P parameters on this device?"""
    from ._lib import load
    return bool(load().xrl_ppo_trunk_chain_fits(int(M), int(tile_rows), int(P)))


def ppo_trunk_chained(plan, opt, **kw):
    """The shared-trunk minibatch launch whose workgroups first finish the optimiser step of the minibatch before it
    (xrl_ppo_trunk_chained).  opt: dict(slabs, n_split, slab_stride, params, grad, m, v, P, state, sumsq_part, max_norm, mirrors,
    sync, fold) -- what ops.reduce_adam would have been called with for the previous minibatch."""
    p = _struct(PpoFused, kw)
    fused_layers_from_plan(plan, p)
    o = OptChain()
    o.slabs, o.slab_stride, o.n_split = opt["slabs"].data_ptr(), int(opt["slab_stride"]), int(opt["n_split"])
    o.params, o.grad, o.m, o.v = (opt[k].data_ptr() for k in ("params", "grad", "m", "v"))
    o.P, o.state, o.sumsq_part, o.n_part = int(opt["P"]), opt["state"].data_ptr(), opt["sumsq_part"].data_ptr(), opt["sumsq_part"].numel()
    o.max_norm, o.sync = float(opt["max_norm"] or 0.0), opt["sync"].data_ptr()
    assert opt["sync"].numel() >= CHAIN_SYNC_WORDS
    mir = o.mirrors
    fill_mirrors(mir, opt["mirrors"])
    if opt.get("fold"):
        mir.fold_off, mir.fold_len = int(opt["fold"][0]), int(opt["fold"][1])
    call("xrl_ppo_trunk_chained", C.byref(p), C.byref(o), stream_ptr())


class PpoWideState:
    """Host side of xrl_ppo_wide_minibatch for an ActorCriticNet of the class D-256-256-{A | 1} with a Gaussian head (the
    MuJoCo network, configs/ppo/mujoco.yaml:8-13): layer offsets, the fragment-ordered copy of the two middle layers and the
    optimiser mirror map that keeps it current (built by running the layout kernel on an index ramp, so that it cannot
    disagree with it)."""

    @staticmethod
    def eligible(model):
        pl = getattr(model, "plan", None)
        if pl is None or getattr(model, "dist", None) != "gaussian" or not hasattr(model, "head_ld"):
            return False
        D, A = model.obs_dim, model.action_dim
        if list(pl.widths) != [D, 512, 512, A + 1] or not (1 <= D <= 24 and 1 <= A <= 8) or model.head_ld != A + 1:
            return False
        if model.activation not in ("relu", "leaky_relu", "tanh") or model.activation_action not in (None, "tanh"):
            return False
        return "actor.mu.0.weight" in model.params.offsets and model.params.P % 4 == 0

    def __init__(self, model):
        assert self.eligible(model)
        self.model = model
        P, o = model.params, model.params.offsets
        self.D, self.A = model.obs_dim, model.action_dim
        d = _lib.PpoWide()
        for b, key in enumerate(("actor.mu", "critic.values")):
            br = d.br[b]
            br.w0, br.b0 = o[f"{key}.0.weight"], o[f"{key}.0.bias"]
            br.w1, br.b1 = o[f"{key}.2.weight"], o[f"{key}.2.bias"]
            br.w2, br.b2 = o[f"{key}.4.weight"], o[f"{key}.4.bias"]
        d.log_std_off = o[getattr(model, "log_std_name", "actor.log_std")]
        d.D, d.A, d.H = self.D, self.A, 256
        d.act, d.out_act = ACT[model.activation], ACT[model.activation_action]
        self.desc = d
        dev = P.device
        self.frag = torch.zeros(4 * 256 * 256, device=dev)          # [branch][forward | backward section][256 * 256]
        ramp = torch.arange(P.P, dtype=torch.float32, device=dev) + 1.0           # exact in fp32 for P < 2^24
        self.pack(ramp)
        torch.cuda.synchronize()
        src = (self.frag - 1.0).to(torch.int64).view(2, 2, -1)                   # parameter index held by every slot
        dst = torch.arange(self.frag.numel(), dtype=torch.int32, device=dev).view(2, 2, -1)
        self.maps = []
        for sec in range(2):                                                      # every weight has one slot per section
            m = torch.full((P.P,), -1, dtype=torch.int32, device=dev)
            m[src[:, sec].reshape(-1)] = dst[:, sec].reshape(-1)
            self.maps.append(m)
        self.mirrors = [(self.maps[0], self.frag), (self.maps[1], self.frag)]
        self.pack()

    def pack(self, flat=None):
        """frag <- the middle layers of `flat` (default: the model's parameters); one launch, capturable."""
        d = self.desc
        d.params = (self.model.params.flat if flat is None else flat).data_ptr()
        call("xrl_ppo_wide_pack", C.byref(d), ptr(self.frag), stream_ptr())

    def act(self, x, n, seed, step, step_dev, act_out=None, env_action_f=None, logp_out=None, val_out=None, bootv_prev=None,
            raw=None, stats_in=None, stats_out=None, obs_slot=None, update=0, normalize=0, obs_range=0.0, next_raw=None, post=None):
        """xrl_wide_act_step: sample / log-prob / value of rows [0, n) of x (when act_out is given) and the values of rows
        [n, 2n) (when bootv_prev is given), one launch.  raw: rows [0, n) as raw observations, normalised inside the launch
        with the running statistics stats_in = (mean, var, count) -> stats_out (RunningMeanStd.update when `update`).
        next_raw: rows [n, 2n) as the previous step's raw next observations (normalised with stats_in); post: the keyword
        arguments of rollout_poststep for the previous step -- its bookkeeping then rides in this launch."""
        a, d = getattr(self, "_act_desc", None), self.desc
        if a is None:
            a = self._act_desc = _lib.WideAct()
            for b in range(2):
                for f, _ in _lib.WideBranch._fields_:
                    setattr(a.br[b], f, getattr(d.br[b], f))
            a.log_std_off, a.D, a.A, a.H, a.act, a.out_act = d.log_std_off, d.D, d.A, d.H, d.act, d.out_act
        as_ptr = lambda t: None if t is None else (t.data_ptr() if isinstance(t, torch.Tensor) else int(t))
        a.params, a.frag = self.model.params.flat.data_ptr(), self.frag.data_ptr()
        a.n, a.flags = int(n), (1 if act_out is not None else 0) | (2 if bootv_prev is not None else 0)
        a.x, a.act_out, a.env_action_f, a.logp_out = as_ptr(x), as_ptr(act_out), as_ptr(env_action_f), as_ptr(logp_out)
        a.val_out, a.bootv_prev = as_ptr(val_out), as_ptr(bootv_prev)
        a.seed, a.step, a.step_dev = int(seed), int(step), as_ptr(step_dev)
        a.raw, a.obs_slot = as_ptr(raw), as_ptr(obs_slot)
        a.mean_in, a.var_in, a.count_in = [as_ptr(t) for t in (stats_in or (None, None, None))]
        a.mean_out, a.var_out, a.count_out = [as_ptr(t) for t in (stats_out or (None, None, None))]
        a.update, a.normalize, a.range = int(update), int(normalize), float(obs_range)
        a.next_raw, a.has_post = as_ptr(next_raw), int(post is not None)
        a.dbg = as_ptr(getattr(self, "act_dbg", None))          # (diagnostics: tools/probe_wide_phases.py)
        a.post = _struct(PostStep, post) if post is not None else PostStep()
        self.prepare_act(n)
        a.xchg, a.xcnt = self._xchg.data_ptr(), self._xcnt.data_ptr()
        call("xrl_wide_act_step", C.byref(a), stream_ptr())

    def prepare_act(self, n):
        """Scratch of xrl_wide_act_step for n envs (call once outside graph capture; act() allocates on demand otherwise)."""
        pairs = 3 * ((int(n) + 31) // 32)
        if getattr(self, "_xchg", None) is None or self._xcnt.numel() < pairs:
            dev = self.frag.device
            self._xchg = torch.zeros(pairs * 4 * 32 * 8, device=dev)
            self._xcnt = torch.zeros(pairs, dtype=torch.int32, device=dev)

    def prepare_rows(self, M):
        """Row buffers of the split weight gradient (xrl_wide_dw1) for minibatches of up to M rows (outside graph capture)."""
        rows = ((int(M) + 31) // 32) * 32
        if getattr(self, "_rows_ld", 0) < rows:
            dev = self.frag.device
            self._rows_g2 = torch.zeros(2, rows, 256, device=dev)
            self._rows_h1 = torch.zeros(2, rows, 256, device=dev)
            self._rows_ld = rows

    def w1_ranges(self):
        d = self.desc
        return [(d.br[b].w1, d.br[b].w1 + 256 * 256) for b in range(2)]

    def launch(self, M, obs, actions, ret, adv, old_logp, slabs, slab_stride, partials, clip_range, vf_coef, ent_coef,
               stats=None, diag=None, heads=None, dbg=None, dbg_role=0, split_dw1=False):
        """split_dw1: the middle layer's weight gradient as a second launch over all rows (xrl_wide_dw1) -- returns the number of
        slab rows its ranges (w1_ranges()) are split over (reduce_adam's alt=...), else None."""
        d = self.desc
        d.params, d.frag = self.model.params.flat.data_ptr(), self.frag.data_ptr()
        d.M, d.dbg_role = int(M), int(dbg_role)
        as_ptr = lambda t: None if t is None else (t.data_ptr() if isinstance(t, torch.Tensor) else int(t))
        d.obs, d.actions, d.ret, d.adv, d.old_logp = as_ptr(obs), as_ptr(actions), as_ptr(ret), as_ptr(adv), as_ptr(old_logp)
        d.stats, d.slabs, d.slab_stride, d.partials = as_ptr(stats), as_ptr(slabs), int(slab_stride), as_ptr(partials)
        d.diag, d.heads, d.dbg = as_ptr(diag), as_ptr(heads), as_ptr(dbg)
        d.clip_range, d.vf_coef, d.ent_coef = float(clip_range), float(vf_coef), float(ent_coef)
        if split_dw1:
            self.prepare_rows(M)
            d.rows_g2, d.rows_h1, d.rows_ld = self._rows_g2.data_ptr(), self._rows_h1.data_ptr(), self._rows_ld
        else:
            d.rows_g2, d.rows_h1, d.rows_ld = None, None, 0
        call("xrl_ppo_wide_minibatch", C.byref(d), stream_ptr())
        if split_dw1:
            parts = C.c_int32(0)
            call("xrl_wide_dw1", C.byref(d), C.byref(parts), stream_ptr())
            return int(parts.value)
        return None


def transpose_mid(plan, params_flat, params_t):
    p = PpoFused()
    p.params = params_flat.data_ptr()
    fused_layers_from_plan(plan, p)
    call("xrl_transpose_mid", C.byref(p), ptr(params_t), stream_ptr())


def rollout_cache_floats(plan):
    p = RolloutStep()
    fused_layers_from_plan(plan, p)
    return int(_lib.load().xrl_rollout_cache_floats(C.byref(p)))


def pack_rollout_cache(plan, params_flat, image, frag=None):
    p = RolloutStep()
    p.params = params_flat.data_ptr()
    fused_layers_from_plan(plan, p)
    call("xrl_pack_rollout_cache2", C.byref(p), ptr(image), image.numel(), ptr(frag), stream_ptr())


def rollout_step_cartpole(plan, **kw):
    p = _struct(RolloutStep, kw)
    fused_layers_from_plan(plan, p)
    call("xrl_rollout_step_cartpole", C.byref(p), stream_ptr())


class CartPoleRollout:
    """Host side of xrl_rollout_cartpole_run / xrl_rollout_cartpole_values (csrc/rollout_actor.hip): the descriptor of one
    agent's rollout -- layer offsets of the 4-128-{128-2,128-1} network, state and buffer pointers, scratch -- built once;
    run(t0, n_steps) / values(t0, n_steps) enqueue steps [t0, t0 + n_steps) on the current stream."""

    MAX_ENVS = 256

    @staticmethod
    def eligible(plan, n):
        st = plan.stages
        return list(plan.widths) == [4, 128, 256, 3] and n <= CartPoleRollout.max_envs() and len(st) == 3 and len(st[0]) == 1 and \
            len(st[1]) == 1 and len(st[2]) == 2 and st[1][0].act == st[0][0].act and fast_kernels_enabled()

    @staticmethod
    def max_envs():
        """min(MAX_ENVS, what THIS device keeps resident on one XCD): xrl_rollout_cartpole_max_envs (a partitioned or smaller GPU
        takes fewer envs in the one-launch rollout; above it the agent uses the launches per vector step)."""
        if torch.cuda.is_available():
            return min(CartPoleRollout.MAX_ENVS, int(_lib.load().xrl_rollout_cartpole_max_envs()))
        return CartPoleRollout.MAX_ENVS

    def __init__(self, plan, T, **kw):
        q = self.q = RolloutRun()
        off = plan.params.offsets
        L0, L1, (Ha, Hc) = plan.stages[0][0], plan.stages[1][0], plan.stages[2]
        q.w0, q.b0, q.w1, q.b1 = off[L0.w_name], off[L0.b_name], off[L1.w_name], off[L1.b_name]
        q.wa, q.ba, q.wc, q.bc = off[Ha.w_name], off[Ha.b_name], off[Hc.w_name], off[Hc.b_name]
        q.act, q.T = ACT[L0.act], int(T)
        self._keep = []
        for k, v in kw.items():
            if isinstance(v, torch.Tensor):
                self._keep.append(v)
                v = v.data_ptr()
            setattr(q, k, v)

    def run(self, t0, n_steps, flags=0, dbg=None):
        self.q.t0, self.q.n_steps, self.q.flags = int(t0), int(n_steps), int(flags)
        self.q.dbg = None if dbg is None else dbg.data_ptr()
        call("xrl_rollout_cartpole_run", C.byref(self.q), stream_ptr())

    def values(self, t0, n_steps):
        self.q.t0, self.q.n_steps = int(t0), int(n_steps)
        call("xrl_rollout_cartpole_values", C.byref(self.q), stream_ptr())


class WideRollout:
    """Host side of xrl_rollout_wide_run (csrc/rollout_wide.hip): the whole rollout of the D-256-256-{A | 1} Gaussian class on the
    device-resident continuous-control provider as ONE launch with only the actor on the step chain (n_envs <= 256); values and
    bootstrap values are the caller's batched pass afterwards."""

    MAX_ENVS = 256

    @staticmethod
    def eligible(model, n):
        cap = min(WideRollout.MAX_ENVS, int(_lib.load().xrl_rollout_wide_max_envs())) if torch.cuda.is_available() else WideRollout.MAX_ENVS
        return PpoWideState.eligible(model) and n <= cap and model.obs_dim <= 20 and fast_kernels_enabled()

    @staticmethod
    def xchg_words():
        return int(_lib.load().xrl_rollout_wide_words())

    def __init__(self, model, T, **kw):
        q = self.q = RolloutWide()
        o = model.params.offsets
        key = "actor.mu"
        q.w0, q.b0 = o[f"{key}.0.weight"], o[f"{key}.0.bias"]
        q.w1, q.b1 = o[f"{key}.2.weight"], o[f"{key}.2.bias"]
        q.w2, q.b2 = o[f"{key}.4.weight"], o[f"{key}.4.bias"]
        q.log_std_off = o[getattr(model, "log_std_name", "actor.log_std")]
        q.D, q.A, q.H, q.T = model.obs_dim, model.action_dim, 256, int(T)
        q.act, q.out_act = ACT[model.activation], ACT[model.activation_action]
        self._keep = []
        for k, v in kw.items():
            if isinstance(v, torch.Tensor):
                self._keep.append(v)
                v = v.data_ptr()
            setattr(q, k, v)

    def run(self, t0, n_steps, flags=0, dbg=None):
        self.q.t0, self.q.n_steps, self.q.flags = int(t0), int(n_steps), int(flags)
        self.q.dbg = None if dbg is None else dbg.data_ptr()
        call("xrl_rollout_wide_run", C.byref(self.q), stream_ptr())


def copy_column(src, ld, col, dst, n, row0=0):
    """dst[i] = src[row0 + i][col] for i < n (src row-major with `ld` columns)."""
    call("xrl_copy_column", src.data_ptr() + 4 * (int(row0) * int(ld) + int(col)), int(ld), ptr(dst), int(n), stream_ptr())


def dqn_td(**kw):
    call("xrl_dqn_td", C.byref(_struct(DqnTd, kw)), stream_ptr())


def qmix_mix_td(**kw):
    call("xrl_qmix_mix_td", C.byref(_struct(Qmix, kw)), stream_ptr())


def _ep_fields(items):
    """items: [(a, b, c or None, row_bytes, slots, flags)] of device tensors."""
    arr = (EpisodeField * len(items))()
    for i, it in enumerate(items):
        a, b, c, rb, slots, flags = it[:6]
        arr[i].a, arr[i].b, arr[i].c = a.data_ptr(), b.data_ptr(), (c.data_ptr() if c is not None else None)
        arr[i].row_bytes, arr[i].slots, arr[i].flags = int(rb), int(slots), int(flags)
        arr[i].d = it[6].data_ptr() if len(it) > 6 and it[6] is not None else None      # (xrl_episode_store_finish: step data)
    return arr


def episode_store_step(items, steps, n_envs):
    call("xrl_episode_store_step", _ep_fields(items), len(items), ptr(_chk(steps, torch.int32)), int(n_envs), stream_ptr())


def episode_store_finish(items, steps, done, end_step, ptr_size, n_envs, buffer_size, gate=None):
    """episode_store_step + episode_finish(advance=False) as one launch; items: (ring, staging, terminal or None, row_bytes,
    slots, flags, step data or None) per field."""
    call("xrl_episode_store_finish", _ep_fields(items), len(items), ptr(_chk(steps, torch.int32)),
         ptr(gate) if gate is not None else None, ptr(_chk(done)), ptr(_chk(end_step, torch.int32)),
         ptr(_chk(ptr_size, torch.int32)), int(n_envs), int(buffer_size), stream_ptr())


def episode_store_finish_gate(items, steps, done, end_step, ptr_size, n_envs, buffer_size, gate, loop_gate, ticket):
    """episode_store_finish + marl_loop_gate(**loop_gate) as one launch (xrl_episode_store_finish_gate); ticket: [1] int32 zeros."""
    call("xrl_episode_store_finish_gate", _ep_fields(items), len(items), ptr(_chk(steps, torch.int32)),
         ptr(gate) if gate is not None else None, ptr(_chk(done)), ptr(_chk(end_step, torch.int32)),
         ptr(_chk(ptr_size, torch.int32)), int(n_envs), int(buffer_size), C.byref(_struct(MarlGate, loop_gate)), ptr(_chk(ticket, torch.int32)),
         stream_ptr())


def episode_finish(items, done, end_step, ptr_size, n_envs, buffer_size, gate=None, advance=True):
    """gate: None, or a device float scalar -- 0 turns the call into a no-op (xrl_episode_finish_gated).  advance=False:
    the ring's {ptr, size} are left for the caller to advance (xrl_marl_loop_gate does it in its own launch)."""
    call("xrl_episode_finish_gated", _ep_fields(items), len(items), ptr(gate) if gate is not None else None, ptr(_chk(done)),
         ptr(_chk(end_step, torch.int32)), ptr(_chk(ptr_size, torch.int32)), int(n_envs), int(buffer_size), int(bool(advance)),
         stream_ptr())


def host_device_pointer(pinned):
    """Device address (int) of a pinned host tensor, for kernels that publish a word straight to the host."""
    assert pinned.is_pinned()
    out = C.c_void_p()
    call("xrl_host_device_pointer", C.c_void_p(pinned.data_ptr()), C.byref(out))
    return out.value


class QmixFusedState:
    """xrl_qmix_fused_t of a feed-forward MixingQNet with the QMIX mixer + the two weight images the launch reads (eval,
    target: the parameters in the padded LDS layout of csrc/qmix_fused.hip) + the parameter -> image index map that
    xrl_reduce_adam's mirror mechanism keeps them current with (batch pointers are filled in per call)."""

    def __init__(self, model, double_q, gamma, items_per_wg, products=0):
        """items_per_wg: transitions per workgroup; products: 0 = matrix-core tiles when a workgroup's items_per_wg * n_agents
        rows fill half a 16-row tile, VALU loops otherwise; 1 / 2 force either (parity tests)."""
        from ._lib import QmixFused, QfImage
        P = model.params
        layers = [st[0] for st in model.agent_plan.stages]
        assert all(len(st) == 1 for st in model.agent_plan.stages) and 1 <= len(layers) <= 4
        q = QmixFused()
        q.n_layers = len(layers)
        acts = {L.act for L in layers[:-1]}
        assert len(acts) <= 1 and layers[-1].act is None
        q.act = ACT[acts.pop()] if acts else ACT[None]
        q.dims[0] = layers[0].K
        for i, L in enumerate(layers):
            q.dims[i + 1] = L.N
            q.w_off[i], q.b_off[i] = P.offsets[L.w_name], P.offsets[L.b_name]
        m = "eval_Qtot"
        names = [f"{m}.hyper_w_1.0", f"{m}.hyper_b_1", f"{m}.hyper_w_1.2", f"{m}.hyper_w_2.2", f"{m}.hyper_b_2.2"]
        for i, n in enumerate(names):
            q.mix_off[2 * i], q.mix_off[2 * i + 1] = P.offsets[n + ".weight"], P.offsets[n + ".bias"]
        # the three first layers are one stacked matrix: [hyper_w_1.0; hyper_w_2.0; hyper_b_2.0] weights, then their biases
        HH, S, H, N = model.HH, model.state_dim, model.H, model.n_agents
        assert P.offsets[f"{m}.hyper_w_2.0.weight"] == P.offsets[f"{m}.hyper_w_1.0.weight"] + HH * S
        assert P.offsets[f"{m}.hyper_b_2.0.weight"] == P.offsets[f"{m}.hyper_w_1.0.weight"] + 2 * HH * S
        assert P.offsets[f"{m}.hyper_w_2.0.bias"] == P.offsets[f"{m}.hyper_w_1.0.bias"] + HH
        assert P.offsets[f"{m}.hyper_b_2.0.bias"] == P.offsets[f"{m}.hyper_w_1.0.bias"] + 2 * HH
        q.N, q.A, q.S, q.H, q.HH = N, model.n_actions, S, H, HH
        q.items_per_wg, q.double_q, q.gamma, q.products = int(items_per_wg), int(bool(double_q)), float(gamma), int(products)
        im = QfImage()
        call("xrl_qmix_fused_layout", C.byref(q), C.byref(im))
        # parameter index -> image index
        import numpy as np
        mp = np.full(P.P, -1, np.int32)

        def place(off, rows, K, base, ldw):
            r, k = np.divmod(np.arange(rows * K), K)
            mp[off:off + rows * K] = base + r * ldw + k
        for i, L in enumerate(layers):
            place(q.w_off[i], L.N, L.K, im.w[i], im.ldw[i])
            mp[q.b_off[i]:q.b_off[i] + L.N] = im.b[i] + np.arange(L.N)
        mK, mN = [S, S, HH, HH, HH], [3 * HH, H, N * H, H, 1]
        for i in range(5):
            place(q.mix_off[2 * i], mN[i], mK[i], im.agent_floats + im.mw[i], im.mldw[i])
            mp[q.mix_off[2 * i + 1]:q.mix_off[2 * i + 1] + mN[i]] = im.agent_floats + im.mb[i] + np.arange(mN[i])
        used = mp[mp >= 0]                                  # (alignment gaps of the flat buffer map nowhere)
        assert len(used) == sum(int(np.prod(sh)) for sh in P.shapes.values()) and len(np.unique(used)) == len(used)
        dev = P.flat.device
        n_img = im.agent_floats + im.mixer_floats
        self.map = torch.as_tensor(mp, device=dev)          # what the optimiser launch's mirror reads (-1: skip)
        self._map64 = torch.as_tensor(np.where(mp >= 0, mp, n_img), device=dev).to(torch.int64)   # gaps -> a spare slot
        self.img_eval, self.img_target = torch.zeros(n_img + 4, device=dev), torch.zeros(n_img + 4, device=dev)
        q.img_eval, q.img_target = self.img_eval.data_ptr(), self.img_target.data_ptr()
        self.struct, self.model, self.items_per_wg = q, model, int(items_per_wg)
        self.refresh()

    def refresh(self):
        """Rebuild both images from the flat buffers (after anything but xrl_reduce_adam changed the parameters)."""
        self.img_eval.index_copy_(0, self._map64, self.model.params.flat)
        self.img_target.index_copy_(0, self._map64, self.model.target_flat)

    def lds_bytes(self):
        return int(_lib.load().xrl_qmix_fused_lds_bytes(C.byref(self.struct)))

    def n_groups(self, B):
        return (int(B) + self.items_per_wg - 1) // self.items_per_wg


class MarlActGruState:
    """xrl_marl_act_gru_t of a MixingQNet's agent network (GRU agents: mlp blocks -> GRU cell -> Q head; feed-forward
    agents: H = 0, every hidden layer a `pre` layer) + the weight image its launch stages in LDS.  The image follows the eval
    parameters through xrl_reduce_adam's mirror maps (`map`: parameter index -> image index); `refresh()` rebuilds it (two
    launches) after anything else changed them."""

    def __init__(self, model, rows_per_wg=1, lds_staged=False):
        from ._lib import MarlActGru, QaImage
        import numpy as np
        P = model.params
        q = MarlActGru()
        if model.use_rnn:
            pre = [st[0] for st in model.pre_plans[2].stages]      # mlp blocks ..., then the input side of the GRU
            post = [st[0] for st in model.post_plans[2].stages]
            assert not model.lstm and all(len(st) == 1 for st in model.pre_plans[2].stages + model.post_plans[2].stages)
            fc, ih = pre[:-1], pre[-1]
            assert len(fc) <= 3 and 1 <= len(post) <= 3 and ih.N == 3 * model.RH
            q.H = model.RH
            rec = [(ih.w_name, ih.b_name, ih.N, ih.K), (model.w_hh, model.b_hh, 3 * model.RH, model.RH)]
            assert ih.act is None
        else:
            layers = [st[0] for st in model.agent_plan.stages]
            assert all(len(st) == 1 for st in model.agent_plan.stages) and 2 <= len(layers) <= 4
            fc, post, rec = layers[:-1], layers[-1:], []
            q.H = 0
        acts = {L.act for L in fc} | {L.act for L in post[:-1]}
        assert len(acts) <= 1 and post[-1].act is None
        q.O, q.n_pre, q.n_post = model.obs_dim, len(fc), len(post)
        q.act = ACT[acts.pop()] if acts else ACT[None]
        for i, L in enumerate(fc):
            q.pre[i] = L.N
        for i, L in enumerate(post):
            q.post[i] = L.N
        q.rows_per_wg = int(rows_per_wg)
        q.lds_staged = 1 if lds_staged else 0
        im = QaImage()
        call("xrl_marl_act_gru_layout", C.byref(q), C.byref(im))
        self.lds_bytes = int(im.lds_bytes)
        mats = [(L.w_name, L.b_name, L.N, L.K) for L in fc] + rec + [(L.w_name, L.b_name, L.N, L.K) for L in post]
        src, dst = [], []
        for l, (wn, bn, N, K) in enumerate(mats):
            r, k = np.divmod(np.arange(N * K), K)
            src.append(P.offsets[wn] + np.arange(N * K))
            dst.append(im.w[l] + ((k // 4) * im.ldw[l] + r) * 4 + k % 4 if im.interleaved else im.w[l] + r * im.ldw[l] + k)
            src.append(P.offsets[bn] + np.arange(N)); dst.append(im.b[l] + np.arange(N))
        dev = P.flat.device
        src_all, dst_all = np.concatenate(src), np.concatenate(dst)
        mp = np.full(P.P, -1, np.int32)                     # parameter index -> image index (xrl_reduce_adam's mirror map)
        mp[src_all] = dst_all
        self.map = torch.as_tensor(mp, device=dev)
        self._src = torch.as_tensor(src_all, device=dev).to(torch.int64)
        self._dst = torch.as_tensor(dst_all, device=dev).to(torch.int64)
        self.image = torch.zeros(int(im.image_floats), device=dev)
        q.image = self.image.data_ptr()
        self.struct, self.model = q, model
        self._stage = torch.zeros(self._src.numel(), device=dev)
        self.refresh()

    def refresh(self):
        torch.index_select(self.model.params.flat, 0, self._src, out=self._stage)
        self.image.index_copy_(0, self._dst, self._stage)

    def launch(self, obs, R, h, reset, q_out, select=None):
        """select: None, or the keyword arguments of marl_select_actions (action, action_f, avail, eps_dev, seed, step,
        step_dev) -- the selection then happens in the same launch."""
        s = self.struct
        s.R, s.obs, s.q, s.ldq = int(R), ptr(obs), ptr(q_out), int(q_out.shape[1])
        s.h = ptr(h) if h is not None else None
        s.reset = ptr(reset) if reset is not None else None
        if select is None:
            s.action = None
        else:
            s.action = ptr(select["action"])
            s.eps_dev = ptr(select["eps_dev"]) if select.get("eps_dev") is not None else None   # (None: epsilon by value)
            s.eps = float(select.get("eps", 0.0))
            s.action_f = ptr(select["action_f"]) if select.get("action_f") is not None else None
            s.avail = ptr(select["avail"]) if select.get("avail") is not None else None
            s.step_dev = ptr(select["step_dev"]) if select.get("step_dev") is not None else None
            s.seed, s.step = int(select["seed"]), int(select.get("step", 0))
        call("xrl_marl_act_gru", C.byref(s), stream_ptr())
        return q_out


def qmix_fused_update(fs, B, obs, obs_next, state, state_next, actions, rewards, terminals, agent_mask, avail_next, slabs,
                      slab_stride, partials, diag, ring=None):
    """ring: None (the nine tensors are the gathered batch) or dict(n_envs, n_size, size_dev, seed, counter, counter_dev,
    idx_out): the nine tensors are then the replay ring's FIELDS and the launch draws and gathers its own batch."""
    q = fs.struct
    q.B = int(B)
    if ring is None:
        q.ring_n_envs = 0
    else:
        q.ring_n_envs, q.ring_n_size, q.size_dev = int(ring["n_envs"]), int(ring["n_size"]), ptr(ring["size_dev"])
        q.draw_seed, q.draw_counter = int(ring["seed"]), int(ring.get("counter", 0))
        q.counter_dev = ptr(ring["counter_dev"]) if ring.get("counter_dev") is not None else None
        q.idx_out = ptr(ring["idx_out"]) if ring.get("idx_out") is not None else None
    q.obs, q.obs_next, q.state, q.state_next = ptr(obs), ptr(obs_next), ptr(state), ptr(state_next)
    q.actions, q.rewards, q.terminals, q.agent_mask = ptr(actions), ptr(rewards), ptr(terminals), ptr(agent_mask)
    q.avail_next = ptr(avail_next) if avail_next is not None else None
    q.slabs, q.slab_stride, q.partials = ptr(slabs), int(slab_stride), ptr(partials)
    q.diag = ptr(diag) if diag is not None else None
    call("xrl_qmix_fused_update", C.byref(q), stream_ptr())
    return fs.n_groups(B)


def qmix_fused_phase_fits(B, items_per_wg, P):
    return bool(_lib.load().xrl_qmix_fused_phase_fits(int(B), int(items_per_wg), int(P)))


def qmix_fused_phase(fs, B, fields, avail_next, slabs, slab_stride, diag, ring, ph):
    """A whole update phase of the feed-forward QMIX learner as ONE launch (xrl_qmix_fused_phase): `fields` = the replay ring's field
    tensors, `ring` as in qmix_fused_update (update u draws with counter + u), `ph`: dict(n_updates, sync_every, params, grad, m, v, P,
    state, map, target, act_image, act_map, phase_partials, epoch_sums, sumsq_part, scalars, tick, tick_inc, sync)."""
    from ._lib import QmixPhase
    q = fs.struct
    q.B = int(B)
    q.ring_n_envs, q.ring_n_size, q.size_dev = int(ring["n_envs"]), int(ring["n_size"]), ptr(ring["size_dev"])
    q.draw_seed, q.draw_counter = int(ring["seed"]), int(ring.get("counter", 0))
    q.counter_dev = ptr(ring["counter_dev"]) if ring.get("counter_dev") is not None else None
    q.idx_out = ptr(ring["idx_out"]) if ring.get("idx_out") is not None else None
    f = fields
    q.obs, q.obs_next, q.state, q.state_next = ptr(f["obs"]), ptr(f["obs_next"]), ptr(f["state"]), ptr(f["state_next"])
    q.actions, q.rewards, q.terminals, q.agent_mask = ptr(f["actions"]), ptr(f["rewards"]), ptr(f["terminals"]), ptr(f["agent_mask"])
    q.avail_next = ptr(avail_next) if avail_next is not None else None
    q.slabs, q.slab_stride, q.partials = ptr(slabs), int(slab_stride), ptr(ph["phase_partials"])
    q.diag = ptr(diag) if diag is not None else None
    h = QmixPhase()
    h.n_updates, h.sync_every, h.P, h.tick_inc = int(ph["n_updates"]), int(ph["sync_every"]), int(ph["P"]), int(ph.get("tick_inc", 0))
    for k in ("params", "grad", "m", "v", "state", "map", "target", "act_image", "act_map", "phase_partials", "epoch_sums", "sumsq_part",
              "scalars", "tick", "sync"):
        setattr(h, k, ptr(ph.get(k)))
    call("xrl_qmix_fused_phase", C.byref(q), C.byref(h), stream_ptr())
    return fs.n_groups(B)


def marl_stored_state(state, done_prev, out):
    """out <- the state rows the reference's multi-agent loops store for the coming step (xrl_marl_stored_state)."""
    n, S = state.shape
    call("xrl_marl_stored_state", ptr(_chk(state, torch.float32)), ptr(done_prev) if done_prev is not None else None,
         ptr(_chk(out, torch.float32)), int(n), int(S), stream_ptr())


def marl_loop_gate(**kw):
    call("xrl_marl_loop_gate", C.byref(_struct(MarlGate, kw)), stream_ptr())


def episode_gather(items, idx, B):
    call("xrl_episode_gather", _ep_fields(items), len(items), ptr(_chk(idx, torch.int64)), int(B), stream_ptr())


def episode_gather_sampled(items, idx_out, B, n_size, size_dev, seed, counter=0, counter_dev=None):
    """sample_replay_indices(idx_out, 1, n_size, ...) + episode_gather in one launch (xrl_episode_gather_sampled)."""
    call("xrl_episode_gather_sampled", _ep_fields(items), len(items), ptr(_chk(idx_out, torch.int64)), int(B), int(n_size),
         ptr(size_dev), int(seed), int(counter), ptr(counter_dev), stream_ptr())


def per_store(sum_tree, min_tree, max_priority, ptr_, alpha, n_envs, capacity):
    call("xrl_per_store", ptr(sum_tree), ptr(min_tree), ptr(max_priority), int(ptr_), float(alpha), int(n_envs), int(capacity),
         stream_ptr())


def per_sample(sum_tree, min_tree, uniforms, size, beta, n_envs, n_size, capacity, per_env, step_choices, weights, flat_idx=None):
    call("xrl_per_sample", ptr(sum_tree), ptr(min_tree), ptr(uniforms), int(size), float(beta), int(n_envs), int(n_size),
         int(capacity), int(per_env), ptr(step_choices), ptr(weights), ptr(flat_idx), stream_ptr())


def per_update_priorities(sum_tree, min_tree, max_priority, idxes, priorities, alpha, n_envs, capacity, per_env):
    call("xrl_per_update_priorities", ptr(sum_tree), ptr(min_tree), ptr(max_priority), ptr(idxes), ptr(priorities),
         float(alpha), int(n_envs), int(capacity), int(per_env), stream_ptr())


def lstm_forward(**kw):
    call("xrl_lstm_forward", C.byref(_struct(LstmFwd, kw)), stream_ptr())


def lstm_backward(**kw):
    call("xrl_lstm_backward", C.byref(_struct(LstmBwd, kw)), stream_ptr())


def gru_forward(**kw):
    call("xrl_gru_forward", C.byref(_struct(GruFwd, kw)), stream_ptr())


def gru_backward(**kw):
    call("xrl_gru_backward", C.byref(_struct(GruBwd, kw)), stream_ptr())


def sync_target(params, target, P, state, sync_frequency):
    call("xrl_sync_target", ptr(params), ptr(target), int(P), ptr(state), int(sync_frequency), stream_ptr())


def marl_select_actions(**kw):
    call("xrl_marl_select_actions", C.byref(_struct(MarlAct, kw)), stream_ptr())


def counter_add(counter, inc):
    call("xrl_counter_add", ptr(counter), int(inc), stream_ptr())


def random_permutation(out, n_perm, N, take, seed, counter=0, counter_dev=None):
    """out [n_perm][take] int64 <- first `take` entries of n_perm pseudo-random permutations of range(N)."""
    call("xrl_random_permutation", ptr(out), int(n_perm), int(N), int(take), int(seed), int(counter), ptr(counter_dev),
         stream_ptr())


_fast_enabled = True


def sample_replay_indices(out, n_envs, n_size, size_dev, seed, counter=0, counter_dev=None):
    call("xrl_sample_replay_indices", ptr(out), out.numel(), int(n_envs), int(n_size), ptr(size_dev), int(seed), int(counter),
         ptr(counter_dev), stream_ptr())


def set_fast_kernels(enable):
    """Select (default) or bypass the shape-specialised twins of the fused kernels; results are bit-identical."""
    global _fast_enabled
    _fast_enabled = bool(enable)
    call("xrl_set_fast_kernels", int(bool(enable)))


def fast_kernels_enabled():
    return _fast_enabled


# ------------------------------------------------------------------------------------------ graphs
class Branch:
    """Launches that may run BESIDE what the caller enqueues next on the current stream (round 6: independent launches of an update --
    a convolution stack's weight gradients of the upper layers next to the rest of its data-gradient chain): begin() forks ONE per-device
    side stream off the current stream (an event), end() leaves it, join() makes the current stream wait for it.  Inside an ops.Graph
    capture the fork / join become two parallel branches of the graph (the side stream joins the capture through the event)."""
    _streams = {}        # device index -> THE side stream of this process (one more hardware queue, not one per caller: see Graph.__enter__)

    def __init__(self):
        dev = torch.cuda.current_device()
        if dev not in Branch._streams:
            Branch._streams[dev] = torch.cuda.Stream()
        self.side = Branch._streams[dev]
        self._ctx = None

    def begin(self):
        self.main = torch.cuda.current_stream()
        ev = torch.cuda.Event()
        ev.record(self.main)
        self.side.wait_event(ev)
        self._ctx = torch.cuda.stream(self.side)
        self._ctx.__enter__()

    def end(self):
        self._done = torch.cuda.Event()
        self._done.record(self.side)
        self._ctx.__exit__(None, None, None)
        self._ctx = None

    def join(self):
        torch.cuda.current_stream().wait_event(self._done)


class Graph:
    """hipGraph capture / replay of a sequence of xrl ops issued on the current torch stream."""

    def __init__(self):
        self.handle = C.c_void_p()
        self.stream = None

    _capture_streams = {}        # device index -> THE stream every capture of this process runs on

    def __enter__(self):
        init_device()
        # One capture stream per device for the whole process.  A new torch.cuda.Stream() per capture walks through torch's
        # pool of 32 streams = 32 hardware queues of this process; with a few more processes on the same GPU (the
        # several-ranks-on-one-GPU tests after a long pytest session) the queues outnumber the hardware's slots, the driver
        # time-slices them, and kernels of different ranks that wait for each other inside a launch no longer overlap.
        dev = torch.cuda.current_device()
        if dev not in Graph._capture_streams:
            Graph._capture_streams[dev] = torch.cuda.Stream()
        self.stream = Graph._capture_streams[dev]
        self.stream.wait_stream(torch.cuda.current_stream())
        self._ctx = torch.cuda.stream(self.stream)
        self._ctx.__enter__()
        call("xrl_graph_begin", self.stream.cuda_stream)
        return self

    def __exit__(self, et, ev, tb):
        try:
            if et is None:
                call("xrl_graph_end", self.stream.cuda_stream, C.byref(self.handle))
            else:                                          # leave no stream of this thread in capture mode behind
                junk = C.c_void_p()
                try:
                    call("xrl_graph_end", self.stream.cuda_stream, C.byref(junk))
                    if junk:
                        _lib.load().xrl_graph_destroy(junk)
                except Exception:
                    pass
        finally:
            self._ctx.__exit__(et, ev, tb)
        return False

    def launch(self):
        call("xrl_graph_launch", self.handle, stream_ptr())

    def __del__(self):
        try:
            if self.handle:
                _lib.load().xrl_graph_destroy(self.handle)
        except Exception:
            pass
