"""GPU parity of every C-ABI primitive against the oracle (oracle/xrl_oracle.py) and the reference fixtures."""
import numpy as np
import pytest

from conftest import load_golden, sub, assert_close

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from xuance_amd import ops
    info = ops.device_info()
    assert info["wave_size"] == 64
    return ops


def dev(x, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(x))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda().contiguous()


def seg_from_events(term, trunc, boot):
    """(bootv, seg) arrays equivalent to the reference's finish_path call pattern (ppo_agent.py:129-157)."""
    T, n = term.shape
    seg = np.zeros((T, n), np.uint8)
    bootv = np.zeros((T, n), np.float32)
    ended = term | trunc
    seg[ended] = 1
    seg[term] |= 2                       # finish_path(0.0, i): Python float -> float64 carry
    bootv[trunc & ~term] = boot[trunc & ~term]
    seg[T - 1] = 1 | np.where(term[T - 1], 2, 0)
    bootv[T - 1] = np.where(term[T - 1], 0.0, boot[T - 1])
    return bootv, seg


def run_gae(ops, rew, val, term, bootv, seg, gamma, lam, use_gae=True):
    adv = torch.zeros(rew.shape, device="cuda")
    ret = torch.zeros(rew.shape, device="cuda")
    ops.gae_scan(dev(rew), dev(val), dev(term, torch.float32), dev(bootv), dev(seg), adv, ret, gamma, lam, use_gae)
    torch.cuda.synchronize()
    return adv.cpu().numpy(), ret.cpu().numpy()


def test_gae_golden_bit_exact(ops):
    g = load_golden("onpolicy_buffer")
    d = sub(g, "gae")
    gamma, lam = float(g["meta"][3]), float(g["meta"][4])
    bootv, seg = seg_from_events(d["term"], d["trunc"], d["boot"])
    adv, ret = run_gae(ops, d["rew"], d["val"], d["term"].astype(np.float32), bootv, seg, gamma, lam)
    assert np.array_equal(adv.T, d["advantages"])      # fixture is env-major [env][t]
    assert np.array_equal(ret.T, d["returns"])
    dn = sub(g, "nogae")
    bootv, seg = seg_from_events(dn["term"], dn["trunc"], dn["boot"])
    adv, ret = run_gae(ops, dn["rew"], dn["val"], dn["term"].astype(np.float32), bootv, seg, gamma, lam, use_gae=False)
    assert_close(adv.T, dn["advantages"], 1e-6)
    assert_close(ret.T, dn["returns"], 1e-6)


@pytest.mark.parametrize("n_envs,T", [(1, 1), (3, 7), (64, 256), (257, 33), (1000, 64)])
def test_gae_vs_oracle_bit_exact(ops, oracle, n_envs, T):
    rng = np.random.default_rng(n_envs * 1000 + T)
    rew = rng.standard_normal((T, n_envs)).astype(np.float32)
    val = rng.standard_normal((T, n_envs)).astype(np.float32)
    term = rng.random((T, n_envs)) < 0.05
    trunc = (rng.random((T, n_envs)) < 0.03) & ~term
    boot = rng.standard_normal((T, n_envs)).astype(np.float32)
    bootv, seg = seg_from_events(term, trunc, boot)
    if n_envs > 2:          # leave one env without a closing call at the end: its tail must stay untouched
        seg[T - 1, 1] = 0
    adv, ret = run_gae(ops, rew, val, term.astype(np.float32), bootv, seg, 0.98, 0.95)
    exp_adv = np.zeros((T, n_envs), np.float32)
    exp_ret = np.zeros((T, n_envs), np.float32)
    for e in range(n_envs):
        start = 0
        for t in range(T):
            if seg[t, e] & 1:
                v = 0.0 if (seg[t, e] & 2) else np.float32(bootv[t, e])
                r_, a_ = oracle.gae_finish_path(rew[start:t + 1, e], val[start:t + 1, e],
                                                term[start:t + 1, e].astype(np.float32), v, 0.98, 0.95)
                exp_ret[start:t + 1, e], exp_adv[start:t + 1, e] = r_, a_
                start = t + 1
    assert np.array_equal(adv, exp_adv)
    assert np.array_equal(ret, exp_ret)


def test_store_gather_advstats(ops, oracle):
    rng = np.random.default_rng(0)
    n_envs, T, D = 37, 19, 5
    ob = oracle.OnPolicyBufferOracle((D,), (), n_envs, T)
    f_obs = torch.zeros(T, n_envs, D, device="cuda")
    f_adv = torch.zeros(T, n_envs, device="cuda")
    f_u8 = torch.zeros(T, n_envs, 16, dtype=torch.uint8, device="cuda")
    u8_ref = np.zeros((n_envs, T, 16), np.uint8)
    for t in range(T):
        obs = rng.standard_normal((n_envs, D)).astype(np.float32)
        adv = rng.standard_normal(n_envs).astype(np.float32)
        u8 = rng.integers(0, 256, (n_envs, 16)).astype(np.uint8)
        ob.observations[:, t], ob.advantages[:, t], u8_ref[:, t] = obs, adv, u8
        ops.soa_store_step([(f_obs, dev(obs), D * 4), (f_adv, dev(adv), 4), (f_u8, dev(u8), 16)], n_envs, t)
    ob.size = T
    bs, nb = 50, 3
    idx = rng.permutation(n_envs * T)[: bs * nb]
    didx = dev(idx.astype(np.int64))
    stats = torch.zeros(nb, 2, device="cuda")
    ops.adv_stats(f_adv, didx, bs, nb, n_envs, T, stats)
    for b in range(nb):
        sl = idx[b * bs:(b + 1) * bs]
        s = ob.sample(sl)
        o_obs = torch.zeros(bs, D, device="cuda")
        o_adv = torch.zeros(bs, device="cuda")
        o_u8 = torch.zeros(bs, 16, dtype=torch.uint8, device="cuda")
        ops.soa_gather([(o_obs, f_obs, D * 4), (o_adv, f_adv, 4), (o_u8, f_u8, 16)], didx[b * bs:(b + 1) * bs],
                       n_envs, T, stats=stats[b], flags=[0, 1, 0])
        torch.cuda.synchronize()
        assert np.array_equal(o_obs.cpu().numpy(), s["obs"])
        env, step = np.divmod(sl, T)
        assert np.array_equal(o_u8.cpu().numpy(), u8_ref[env, step])
        assert_close(o_adv.cpu().numpy(), s["advantages"], 1e-6, "adv-norm")
        raw = ob.advantages[env, step]
        assert_close(stats[b].cpu().numpy(), [raw.mean(), raw.std()], 1e-6, "stats")


SHAPES = [(1, 1, 1), (5, 3, 2), (100, 2, 17), (64, 64, 32), (130, 70, 33), (256, 128, 128), (513, 9, 30),
          (1000, 256, 4), (77, 257, 65)]


@pytest.mark.parametrize("M,N,K", SHAPES)
@pytest.mark.parametrize("act", [None, "leaky_relu", "tanh"])
def test_linear_fwd(ops, M, N, K, act, oracle):
    rng = np.random.default_rng(M + N + K)
    A = rng.standard_normal((M, K)).astype(np.float32)
    W = rng.standard_normal((N, K)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    dA, dW, db = dev(A), dev(W), dev(b)
    Cm = torch.full((M, N), 7.0, device="cuda")
    ops.linear_fwd([ops.gemm_desc(dA.data_ptr(), dW.data_ptr(), Cm.data_ptr(), M, N, K, K, K, N, bias=db.data_ptr(),
                                  act=act)])
    torch.cuda.synchronize()
    ref = oracle.act_fwd(A.astype(np.float64) @ W.astype(np.float64).T + b, act)
    assert_close(Cm.cpu().numpy(), ref, 1e-6, "fwd", scale=max(1.0, float(np.max(np.abs(A) @ np.abs(W).T))))


@pytest.mark.parametrize("M,N,K", SHAPES)
def test_linear_bwd_data(ops, M, N, K, oracle):
    """C[M,N] = (dY[M,K] . W[K,N]) * act'(aux)."""
    rng = np.random.default_rng(M * 3 + N + K)
    dY = rng.standard_normal((M, K)).astype(np.float32)
    W = rng.standard_normal((K, N)).astype(np.float32)
    aux = np.tanh(rng.standard_normal((M, N))).astype(np.float32)
    ddY, dW, daux = dev(dY), dev(W), dev(aux)
    Cm = torch.full((M, N), 7.0, device="cuda")
    ops.linear_bwd_data([ops.gemm_desc(ddY.data_ptr(), dW.data_ptr(), Cm.data_ptr(), M, N, K, K, N, N,
                                       aux=daux.data_ptr(), ldaux=N, act="tanh")])
    torch.cuda.synchronize()
    ref = (dY.astype(np.float64) @ W.astype(np.float64)) * (1 - aux.astype(np.float64) ** 2)
    assert_close(Cm.cpu().numpy(), ref, 1e-6, "bwd_data", scale=max(1.0, float(np.max(np.abs(dY) @ np.abs(W)))))


@pytest.mark.parametrize("M,N,K", SHAPES)
@pytest.mark.parametrize("n_split", [1, 4])
def test_linear_bwd_weight(ops, M, N, K, n_split):
    """dW[N,K] = dY[M,N]^T X[M,K], db[N] = sum_m dY, as n_split deterministic slabs."""
    rng = np.random.default_rng(M * 5 + N + K)
    dY = rng.standard_normal((M, N)).astype(np.float32)
    X = rng.standard_normal((M, K)).astype(np.float32)
    ddY, dX = dev(dY), dev(X)
    P = N * K + N
    slabs = torch.full((n_split, P), 7.0, device="cuda")
    ops.linear_bwd_weight([ops.gemm_desc(ddY.data_ptr(), dX.data_ptr(), slabs.data_ptr(), M, N, K, N, K, K,
                                         dbias=slabs.data_ptr() + 4 * N * K)], n_split, P)
    torch.cuda.synchronize()
    got = slabs.cpu().numpy().astype(np.float64).sum(0)
    refW = dY.astype(np.float64).T @ X.astype(np.float64)
    refb = dY.astype(np.float64).sum(0)
    sc = max(1.0, float(np.max(np.abs(dY).T @ np.abs(X))))
    assert_close(got[: N * K].reshape(N, K), refW, 1e-6, "dW", scale=sc)
    assert_close(got[N * K:], refb, 1e-6, "db", scale=sc)


def test_linear_grouped(ops, oracle):
    rng = np.random.default_rng(1)
    M = 200
    shapes = [(7, 33), (64, 128), (1, 128)]
    descs, keep, refs = [], [], []
    for (N, K) in shapes:
        A = rng.standard_normal((M, K)).astype(np.float32)
        W = rng.standard_normal((N, K)).astype(np.float32)
        b = rng.standard_normal(N).astype(np.float32)
        dA, dW, db = dev(A), dev(W), dev(b)
        Cm = torch.zeros(M, N, device="cuda")
        keep += [dA, dW, db, Cm]
        descs.append(ops.gemm_desc(dA.data_ptr(), dW.data_ptr(), Cm.data_ptr(), M, N, K, K, K, N, bias=db.data_ptr(),
                                   act="relu"))
        refs.append((Cm, np.maximum(A.astype(np.float64) @ W.T.astype(np.float64) + b, 0), np.abs(A) @ np.abs(W).T))
    ops.linear_fwd(descs)
    torch.cuda.synchronize()
    for Cm, ref, sc in refs:
        assert_close(Cm.cpu().numpy(), ref, 1e-6, "grouped", scale=max(1.0, float(sc.max())))


@pytest.mark.parametrize("dist", ["categorical", "gaussian"])
@pytest.mark.parametrize("M,n_split", [(96, 1), (1000, 7), (8192, 32)])
def test_ppo_loss(ops, oracle, dist, M, n_split):
    rng = np.random.default_rng(M)
    A = 3 if dist == "categorical" else 6
    out = rng.standard_normal((M, A)).astype(np.float32)
    if dist == "gaussian":
        out = np.tanh(out).astype(np.float32)
    v = rng.standard_normal(M).astype(np.float32)
    adv_raw = rng.standard_normal(M).astype(np.float32) * 2 + 0.3
    ret = rng.standard_normal(M).astype(np.float32)
    log_std = (rng.standard_normal(A) * 0.2 - 1).astype(np.float32)
    cfg = dict(clip_range=0.2, vf_coef=0.25, ent_coef=0.01)
    if dist == "categorical":
        act = rng.integers(0, A, M).astype(np.float32)
        lsm = oracle.log_softmax(out)
        base = lsm[np.arange(M), act.astype(int)]
    else:
        act = (out + np.exp(log_std) * rng.standard_normal((M, A))).astype(np.float32)
        base = (-((act - out) ** 2) / (2 * np.exp(log_std) ** 2) - log_std - 0.9189385).sum(-1)
    old_logp = (base + rng.standard_normal(M) * 0.3).astype(np.float32)
    stats = np.array([adv_raw.mean(), adv_raw.std()], np.float32)
    adv = ((adv_raw - stats[0]) / (stats[1] + 1e-8)).astype(np.float32)

    # oracle: a network whose "last layer" is the identity on `out`; we check the head-level gradients only
    lo, hi = np.float32(1 - 0.2), np.float32(1 + 0.2)
    if dist == "categorical":
        p = np.exp(lsm); logp = base.astype(np.float32); ent = -(p * lsm).sum(-1)
    else:
        logp = base.astype(np.float32); ent = np.full(M, (0.5 + 0.9189385 + log_std).sum(), np.float32)
    ratio = np.exp(logp - old_logp)
    s1 = np.clip(ratio, lo, hi) * adv; s2 = adv * ratio
    inside = ((ratio >= lo) & (ratio <= hi)).astype(np.float32)
    w1 = np.where(s1 < s2, 1.0, np.where(s1 == s2, 0.5, 0.0)).astype(np.float32)
    dlogp = -(w1 * inside * adv + (1 - w1) * adv) / M * ratio
    if dist == "categorical":
        onehot = np.zeros_like(out); onehot[np.arange(M), act.astype(int)] = 1
        d_out = dlogp[:, None] * (onehot - p) + (-0.01 / M) * (-p * (lsm + ent[:, None]))
    else:
        var = np.exp(log_std) ** 2
        d_out = dlogp[:, None] * (act - out) / var * (1 - out ** 2)
        d_ls = (dlogp[:, None] * ((act - out) ** 2 / var - 1)).sum(0) - 0.01
    d_v = 0.25 * 2 * (v - ret) / M

    t = {k: dev(x) for k, x in dict(out=out, v=v, act=act, adv=adv_raw, stats=stats, ret=ret, old=old_logp,
                                    ls=log_std).items()}
    g_out = torch.zeros(M, A, device="cuda"); g_v = torch.zeros(M, device="cuda")
    diag = torch.zeros(4, M, device="cuda")
    partials = torch.zeros(n_split, 8, dtype=torch.float64, device="cuda")
    g_ls = torch.zeros(n_split, A, device="cuda")
    ops.ppo_loss(dist, out=t["out"].data_ptr(), value=t["v"].data_ptr(), actions=t["act"].data_ptr(),
                 adv=t["adv"].data_ptr(), stats=t["stats"].data_ptr(), returns=t["ret"].data_ptr(),
                 old_logp=t["old"].data_ptr(), log_std=t["ls"].data_ptr() if dist == "gaussian" else None,
                 d_out=g_out.data_ptr(), d_value=g_v.data_ptr(),
                 d_log_std=g_ls.data_ptr() if dist == "gaussian" else None, diag=diag.data_ptr(),
                 partials=partials.data_ptr(), M=M, A=A, ld_out=A, ld_v=1, out_act=3 if dist == "gaussian" else 0,
                 n_split=n_split, slab_stride=A, **cfg)
    sums = torch.zeros(8, dtype=torch.float64, device="cuda")
    ops.sum_partials(partials, n_split, 8, sums)
    torch.cuda.synchronize()
    lp_scale = max(1.0, float(np.abs(logp).max()))
    dg = diag.cpu().numpy()
    assert_close(dg[0], logp, 1e-6, "logp", scale=lp_scale)
    assert_close(dg[1], ratio, 2e-6, "ratio", scale=lp_scale)
    assert_close(dg[2], s1, 2e-6, "s1", scale=lp_scale * 4)
    assert_close(dg[3], s2, 2e-6, "s2", scale=lp_scale * 4)
    sm = sums.cpu().numpy()
    assert_close(-sm[0] / M, -np.minimum(s1, s2).astype(np.float64).mean(), 1e-5, "a_loss")
    assert_close(sm[1] / M, ((v - ret).astype(np.float64) ** 2).mean(), 1e-5, "c_loss")
    assert_close(sm[2] / M, ent.astype(np.float64).mean(), 1e-5, "e_loss")
    assert_close(sm[3] / M, v.astype(np.float64).mean(), 1e-5, "v mean")
    assert sm[4] == ((ratio < lo).sum() + (ratio > hi).sum())
    gscale = float(np.abs(d_out).max())
    assert_close(g_out.cpu().numpy() / gscale, d_out / gscale, 1e-5, "d_out")
    assert_close(g_v.cpu().numpy() * M, d_v * M, 1e-5, "d_v")
    if dist == "gaussian":
        assert_close(g_ls.cpu().numpy().astype(np.float64).sum(0), d_ls, 1e-5, "d_log_std",
                     scale=max(1.0, float(np.abs(d_ls).max())))


@pytest.mark.parametrize("P", [1, 1000, 34051])
def test_adam_clip(ops, oracle, P):
    rng = np.random.default_rng(P)
    S = 5
    params = rng.standard_normal(P).astype(np.float32)
    sd = {"w": params.copy()}
    opt = oracle.AdamOracle(sd, lr=4e-4, eps=1e-5, end_factor=0.5, total_iters=10)
    d_p = dev(params); d_m = torch.zeros(P, device="cuda"); d_v = torch.zeros(P, device="cuda")
    d_g = torch.zeros(P, device="cuda")
    st = ops.adam_state_tensor(4e-4, 10, end_factor=0.5)
    part = torch.zeros(64, dtype=torch.float64, device="cuda")
    for it in range(4):
        slabs = (rng.standard_normal((S, P)) * (0.01 if it % 2 else 1.0)).astype(np.float32)
        g = {"w": slabs.astype(np.float64).sum(0).astype(np.float32)}
        norm = opt.clip_grad_norm_(g, 0.5)
        opt.step(g)
        ops.grad_reduce(dev(slabs), S, P, P, d_g, part)
        ops.adam_step(d_p, d_g, d_m, d_v, P, st, part, 0.5)
        torch.cuda.synchronize()
        s = ops.read_adam_state(st)
        assert s.step == it + 1 and s.sched_steps == it + 1 and s.ticket == 0
        assert_close(s.last_grad_norm, norm, 1e-6, "norm", scale=norm)
        assert_close(s.last_lr, opt.lr, 1e-12, "lr")
        assert_close(d_g.cpu().numpy(), g["w"], 1e-6, "clipped grad")
        assert_close(d_p.cpu().numpy(), sd["w"], 1e-6, "params")
        assert_close(d_m.cpu().numpy(), opt.m["w"], 1e-6, "m")
        assert_close(d_v.cpu().numpy(), opt.v["w"], 1e-6, "v")


@pytest.mark.parametrize("P", [4096, 1703936 // 16])
def test_adam_step_with_16_byte_accesses_equals_the_element_loop(ops, P):
    """xrl_adam_step_mirrors (round 6): four consecutive elements per thread through 16-byte accesses where every array allows it
    (P % 4 == 0, 16-byte aligned bases) vs the element-by-element loop (taken here by handing the same numbers over at a base that is
    4 bytes off alignment): parameters, moments, clipped gradient, both mirrors and the optimiser state bit-equal over three steps."""
    g = torch.Generator(device="cpu").manual_seed(P)
    mk = lambda: torch.randn(P, generator=g)
    p0, v0 = mk(), mk().abs() * 1e-3
    m0 = mk() * 0.1
    grads = [mk() * (10.0 if k == 1 else 0.01) for k in range(3)]
    perm_a, perm_b = torch.randperm(P, generator=g).to(torch.int32), torch.randperm(P, generator=g).to(torch.int32)
    perm_b[::7] = -1
    res = []
    for off in (0, 1):                                      # off = 1: every array starts 4 bytes past a 16-byte boundary
        def put(x, dtype=torch.float32):
            buf = torch.zeros(P + 4, dtype=dtype, device="cuda")
            buf[off:off + P].copy_(x)
            return buf[off:off + P]
        p, m, v = put(p0), put(m0), put(v0)
        ma, mb = put(perm_a, torch.int32), put(perm_b, torch.int32)
        da, db = torch.zeros(P, device="cuda"), torch.full((P,), -3.0, device="cuda")
        st = ops.adam_state_tensor(3e-4, 10, end_factor=0.5)
        part = torch.zeros(4, dtype=torch.float64, device="cuda")
        gs = []
        for k in range(3):
            gk = put(grads[k])
            part.zero_(); part[0] = float((grads[k].double() ** 2).sum())
            ops.adam_step_mirrors(p, gk, m, v, P, st, part, 0.5, [(ma, da), (mb, db)])
            torch.cuda.synchronize()
            gs.append(gk.clone())
        s = ops.read_adam_state(st)
        res.append((p.clone(), m.clone(), v.clone(), da, db, gs, (s.step, s.sched_steps, s.last_grad_norm, s.last_lr)))
    a, b = res
    for x, y, name in zip(a[:5], b[:5], ("params", "m", "v", "mirror a", "mirror b")):
        assert torch.equal(x, y), name
    assert all(torch.equal(x, y) for x, y in zip(a[5], b[5])) and a[6] == b[6] and a[6][0] == 3
    assert torch.equal(a[3][perm_a.long().cuda()], a[0]) and float((a[4] == -3.0).sum()) == float((perm_b < 0).sum())


@pytest.mark.parametrize("P", [4096, 840004, 77])
def test_grad_reduce_of_one_slab_equals_the_general_path(ops, P):
    """xrl_grad_reduce with ONE slab (round 6: its four waves take four trips of the block's walk at once instead of one wave walking
    alone) against the general path on the same numbers (two slabs, the second all zero: x + 0 = x): the gradient and every block's
    partial sum of squares -- hence the clip coefficient -- bit-equal."""
    g = torch.Generator(device="cpu").manual_seed(P)
    x = torch.randn(P, generator=g).cuda()
    two = torch.zeros(2, P, device="cuda")
    two[0].copy_(x)
    out = []
    for slabs, S in ((x.view(1, P).clone(), 1), (two, 2)):
        grad = torch.full((P,), 7.0, device="cuda")
        part = torch.zeros(256, dtype=torch.float64, device="cuda")
        ops.grad_reduce(slabs, S, P, P, grad, part)
        torch.cuda.synchronize()
        out.append((grad, part))
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][0], x)
    assert torch.equal(out[0][1], out[1][1]) and float(out[0][1].sum()) > 0
    assert abs(float(out[0][1].sum()) - float((x.double() ** 2).sum())) < 1e-9 * float((x.double() ** 2).sum())


@pytest.mark.parametrize("M", [512, 37, 5856])
def test_plans_as_one_launch_equal_the_launch_per_stage(ops, M):
    """xrl_mlp_chain_fwd (round 6): whole feed-forward plans -- the actor-critic of the classic-control yaml files (6-128-{128-3, 128-1},
    leaky_relu; tanh), the fc -> W_ih chain below the recurrent QMIX agents' GRU and their Q head above it, eval and target twins and
    the two mixers' hyper-network plans riding in the same launch -- as ONE launch (a workgroup keeps a 32-row tile of every level in
    LDS) against one grouped xrl_linear_fwd per stage: EVERY level of every plan bit-equal (same MFMA chains), ragged last tiles."""
    from xuance_amd.nets import ActorCriticNet, MixingQNet, Plan
    torch.manual_seed(M)
    g = torch.Generator(device="cpu").manual_seed(M)

    def run(items, chain):
        for plan, *_ in items:
            plan.ensure(M)
            for lvl in plan.acts:
                plan.acts[lvl].fill_(-9.0)
        out = Plan.forward_chain(items) if chain else Plan.forward_many(items)
        assert out is not None
        torch.cuda.synchronize()
        return [{lvl: a[:M].clone() for lvl, a in plan.acts.items()} for plan, *_ in items]

    for act in ("leaky_relu", "tanh"):
        net = ActorCriticNet(6, 3, "categorical", (128,), (128,), (128,), act)
        X = torch.randn(M, 6, generator=g).cuda()
        items = [(net.plan, X, 6, M, None)]
        a, b = run(items, False), run(items, True)
        for pa, pb in zip(a, b):
            for lvl in pa:
                assert torch.equal(pa[lvl], pb[lvl]), (act, lvl)
        assert float(a[0][len(net.plan.widths) - 1].abs().max()) > 0
    q = MixingQNet(3, 30, 9, 48, (), (64,), 32, 32, "relu", use_rnn=True, fc_hidden=(64,), recurrent_hidden=64)
    q.target_flat.add_(0.01 * torch.randn(q.target_flat.shape, generator=g).cuda())
    Xo, Hs, St = torch.randn(M, 30, generator=g).cuda(), torch.randn(M, 64, generator=g).cuda(), torch.randn(M, 48, generator=g).cuda()
    pre = [(q.pre_plans[0], Xo, 30, M, None), (q.pre_plans[1], Xo, 30, M, q.target_flat)]
    post = [(q.post_plans[0], Hs, 64, M, None), (q.post_plans[1], Hs, 64, M, q.target_flat),
            (q.mixer_plan, St, 48, M, None), (q.mixer_target_plan, St, 48, M, q.target_flat)]
    for items in (pre, post):
        a, b = run(items, False), run(items, True)
        for i, (pa, pb) in enumerate(zip(a, b)):
            for lvl in pa:
                assert torch.equal(pa[lvl], pb[lvl]), (i, lvl)
            assert float(pa[max(pa)].abs().max()) > 0


def test_graph_replay(ops):
    P = 1000
    a = torch.ones(P, device="cuda"); g = torch.full((P,), 0.1, device="cuda")
    m = torch.zeros(P, device="cuda"); v = torch.zeros(P, device="cuda")
    slabs = torch.full((2, P), 0.05, device="cuda")
    part = torch.zeros(8, dtype=torch.float64, device="cuda")
    st = ops.adam_state_tensor(1e-3, 100)
    torch.cuda.synchronize()
    gr = ops.Graph()
    with gr:
        ops.grad_reduce(slabs, 2, P, P, g, part)
        ops.adam_step(a, g, m, v, P, st, part, 0.0)
    for _ in range(5):
        gr.launch()
    torch.cuda.synchronize()
    s = ops.read_adam_state(st)
    assert s.step == 5
    assert abs(float(a[0]) - (1 - 5e-3)) < 1e-4


@pytest.mark.parametrize("N,take,n_perm", [(65536, 65536, 8), (1000, 1000, 3), (4097, 4000, 2), (2, 2, 4)])
def test_random_permutation_rows_are_permutations(N, take, n_perm):
    """xrl_random_permutation: every row holds `take` distinct values of range(N); rows and update phases differ."""
    from xuance_amd import ops
    out = torch.full((n_perm, take), -1, dtype=torch.int64, device="cuda")
    ctr = torch.zeros(1, dtype=torch.int32, device="cuda")
    ops.random_permutation(out, n_perm, N, take, seed=7, counter=0, counter_dev=ctr)
    a = out.cpu().numpy()
    for e in range(n_perm):
        assert a[e].min() >= 0 and a[e].max() < N and len(np.unique(a[e])) == take
    if N > 100:
        assert not np.array_equal(a[0], a[1])
        if take == N:                                     # no fixed-point / identity bias worth noticing
            assert (a[0] == np.arange(N)).mean() < 0.01
            assert abs(np.corrcoef(a[0], np.arange(N))[0, 1]) < 0.05
        ops.counter_add(ctr, 1)
        out2 = torch.empty_like(out)
        ops.random_permutation(out2, n_perm, N, take, seed=7, counter=0, counter_dev=ctr)
        assert not np.array_equal(out2.cpu().numpy()[0], a[0])
        out3 = torch.empty_like(out)
        ops.random_permutation(out3, n_perm, N, take, seed=7, counter=1, counter_dev=None)
        assert np.array_equal(out3.cpu().numpy(), out2.cpu().numpy())


def test_c_abi_rejects_bad_arguments_without_launching():
    """Error behaviour of the boundary (include/xrl_hip.h:17): a negative XRL_E* code and a message from xrl_last_error(),
    never an exception across the ABI, and nothing enqueued -- the destination stays untouched."""
    import ctypes as C
    from xuance_amd import _lib, ops
    lib = _lib.load()
    out = torch.full((8,), 7.0, device="cuda")
    part = torch.ones(4, 8, dtype=torch.float64, device="cuda")
    sums = torch.full((8,), 7.0, dtype=torch.float64, device="cuda")
    rc = lib.xrl_sum_partials(None, 4, 8, sums.data_ptr(), None)                   # null input
    assert rc == -1 and len(lib.xrl_last_error()) > 0
    rc = lib.xrl_sum_partials(part.data_ptr(), 4, 65, sums.data_ptr(), None)      # unsupported width
    assert rc < 0 and len(lib.xrl_last_error()) > 0
    rc = lib.xrl_sample_replay_indices(None, 8, 4, 4, None, 1, 0, None, None)
    assert rc < 0
    torch.cuda.synchronize()
    assert float(sums.min()) == 7.0 and float(out.min()) == 7.0
    # the Python wrappers turn the code into XrlError with the library's message
    with pytest.raises(ops.XrlError) as e:
        ops.sum_partials(part, 0, 8, sums)
    assert "xrl_sum_partials" in str(e.value)
    with pytest.raises(ops.XrlError):
        ops.sample_replay_indices(torch.zeros(4, dtype=torch.int64, device="cuda"), 0, 4,
                                  torch.ones(1, dtype=torch.int32, device="cuda"), 1)          # n_envs = 0
    # a valid call still works afterwards
    ops.sum_partials(part, 4, 8, sums)
    assert torch.equal(sums, torch.full((8,), 4.0, dtype=torch.float64, device="cuda"))


@pytest.mark.parametrize("n,A", [(256, 2), (1000, 9), (33, 4)])
def test_policy_sample_categorical_with_supplied_uniforms(ops, oracle, n, A):
    """xrl_policy_sample in its supplied-noise mode (xrl_sample_t.noise): the inverse-CDF draw is an OUTPUT here, checked
    against oracle.categorical_sample_icdf on the same logits and uniforms (incl. u = 0 and u just below 1), with the
    log-prob of the drawn action, the value column and the bootstrap rows."""
    rng = np.random.default_rng(n + A)
    heads = (rng.standard_normal((2 * n, A + 1)) * 2).astype(np.float32)
    u = rng.random(n).astype(np.float32)
    u[0], u[1] = 0.0, np.float32(1.0 - 2.0 ** -24)
    act, val, logp = (torch.zeros(n, device="cuda") for _ in range(3))
    env_a = torch.zeros(n, dtype=torch.int32, device="cuda")
    boot = torch.zeros(n, device="cuda")
    ops.policy_sample(heads=dev(heads), noise=dev(u), act_out=act, val_out=val, logp_out=logp, env_action=env_a,
                      bootv_prev=boot, n=n, A=A, ld=A + 1, gaussian=0, seed=1, step=0, step_dev=None)
    torch.cuda.synchronize()
    logits = heads[:n, :A]
    ref = oracle.categorical_sample_icdf(logits, u)
    got = act.cpu().numpy().astype(int)
    cdf = np.cumsum(np.exp(oracle.log_softmax(logits)), -1, dtype=np.float32)
    tie = np.abs(cdf - u[:, None]).min(-1) < 1e-6                       # float32 cdf vs uniform: knife edges may go either way
    assert np.array_equal(got[~tie], ref[~tie]) and tie.sum() <= 2
    assert np.array_equal(env_a.cpu().numpy(), got)
    assert_close(logp.cpu().numpy(), oracle.log_softmax(logits)[np.arange(n), got], 1e-5, "log_prob of the drawn action")
    assert np.array_equal(val.cpu().numpy(), heads[:n, A]) and np.array_equal(boot.cpu().numpy(), heads[n:, A])
    assert len(set(got.tolist())) == A                                   # every action occurs


def test_policy_sample_gaussian_with_supplied_normals(ops, oracle):
    """Gaussian twin: x = mu + exp(log_std) * z with supplied standard normals z, summed Normal.log_prob
    (distributions.py:172-180) against oracle.gaussian_sample_reparam."""
    rng = np.random.default_rng(6)
    n, A = 300, 6
    heads = np.tanh(rng.standard_normal((2 * n, A + 1))).astype(np.float32)
    log_std = (rng.standard_normal(A) * 0.3 - 1).astype(np.float32)
    z = rng.standard_normal((n, A)).astype(np.float32)
    act, env_a = torch.zeros(n, A, device="cuda"), torch.zeros(n, A, device="cuda")
    val, logp = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    ops.policy_sample(heads=dev(heads), log_std=dev(log_std), noise=dev(z), act_out=act, val_out=val, logp_out=logp,
                      env_action_f=env_a, bootv_prev=None, n=n, A=A, ld=A + 1, gaussian=1, seed=1, step=0, step_dev=None)
    torch.cuda.synchronize()
    x, lp = oracle.gaussian_sample_reparam(heads[:n, :A], log_std, z)
    assert_close(act.cpu().numpy(), x, 1e-6, "sampled action")
    assert np.array_equal(env_a.cpu().numpy(), act.cpu().numpy())
    assert_close(logp.cpu().numpy(), lp, 1e-5, "log_prob", scale=float(np.abs(lp).max()))
    assert np.array_equal(val.cpu().numpy(), heads[:n, A])


def test_device_philox_streams_equal_the_oracle_restatement(ops, oracle):
    """Without supplied noise the draw uses Philox4x32-10 keyed by (seed, env, step, stream) (csrc/rng.h): the oracle's
    restatement (pinned by the Random123 known-answer vector in the CPU suite) must reproduce the device's actions and
    the device CartPole's reset states bit for bit."""
    from xuance_amd.envs import DeviceCartPoleVecEnv
    rng = np.random.default_rng(2)
    n, A, seed, step = 512, 2, 77, 1234
    heads = rng.standard_normal((2 * n, A + 1)).astype(np.float32)
    act, val, logp = (torch.zeros(n, device="cuda") for _ in range(3))
    counter = torch.tensor([1000], dtype=torch.int32, device="cuda")
    ops.policy_sample(heads=dev(heads), act_out=act, val_out=val, logp_out=logp, bootv_prev=None, n=n, A=A, ld=A + 1,
                      gaussian=0, seed=seed, step=step - 1000, step_dev=counter)
    env = DeviceCartPoleVecEnv(n, seed=9)
    env.reset()
    torch.cuda.synchronize()
    u = oracle.action_uniforms(seed, n, step)
    ref = oracle.categorical_sample_icdf(heads[:n, :A], u)
    p0 = np.exp(oracle.log_softmax(heads[:n, :A]))[:, 0]
    tie = np.abs(p0 - u) < 1e-6
    assert np.array_equal(act.cpu().numpy().astype(int)[~tie], ref[~tie]) and tie.sum() <= 1
    assert np.array_equal(env.state.cpu().numpy(), oracle.cartpole_reset_state(9, np.arange(n), 0))


@pytest.mark.parametrize("kind,name", [(1, "DevicePendulumVecEnv"), (2, "DeviceMountainCarVecEnv"), (3, "DeviceAcrobotVecEnv")])
def test_classic_control_device_envs_vs_oracle(oracle, kind, name):
    """xrl_classic_step (csrc/classic_control.hip, csrc/classic.h) against the NumPy statement of the same published equations
    (oracle/xrl_oracle.py: PendulumOracle / MountainCarOracle / AcrobotOracle): 64 envs, random actions, enough steps for cut-offs,
    terminations and auto-resets; observations, rewards, flags per step; the float64 state within a few ulp (the device's sin / cos
    are not NumPy's); initial and reset states bit-equal (same Philox words, same affine map)."""
    import xuance_amd.envs as envs
    n, seed = 64, 11
    steps = {1: 230, 2: 230, 3: 560}[kind]
    env = getattr(envs, name)(n, seed=seed)
    env.reset()
    torch.cuda.synchronize()
    s0 = oracle.classic_reset_state(kind, seed, np.arange(n), np.zeros(n, np.int64))
    assert np.array_equal(env.state.cpu().numpy(), s0)
    O = {1: oracle.PendulumOracle, 2: oracle.MountainCarOracle, 3: oracle.AcrobotOracle}[kind]
    ref = O(s0)
    assert_close(env.buf_obs.cpu().numpy(), O.observe(ref.state), 1e-6, "initial observation")
    rng = np.random.default_rng(5)
    episodes = np.zeros(n, np.int64)
    n_term = n_trunc = 0
    for t in range(steps):
        if kind == 1:
            a = (rng.standard_normal((n, 1)) * 2).astype(np.float32)         # beyond +-2 as well: the env clips
        else:
            a = rng.integers(0, 3, n)
            if kind == 2:
                a = np.where(rng.random(n) < 0.7, 2 * (ref.state[:, 1] >= 0), a)   # mostly pump the car: some envs reach the flag
        obs, rew, term, trunc, infos = env.step(a)
        o_ref, r_ref, term_ref, trunc_ref = ref.step(a)
        assert np.array_equal(term, term_ref) and np.array_equal(trunc, trunc_ref), t
        assert_close(obs, o_ref, 1e-6, f"obs step {t}")
        assert_close(rew, r_ref, 1e-6, f"reward step {t}", scale=max(1.0, float(np.abs(r_ref).max())))
        done = term | trunc
        n_term += int(term.sum()); n_trunc += int((trunc & ~term).sum())
        if done.any():
            idx = np.nonzero(done)[0]
            episodes[idx] += 1
            fresh = oracle.classic_reset_state(kind, seed, idx, episodes[idx])
            ref.state[idx] = fresh[:, :ref.state.shape[1]]
            ref.steps[idx] = 0
            assert np.array_equal(env.state.cpu().numpy()[idx], fresh)           # the reset state: bit-equal
            for i in idx:
                assert_close(infos[i]["reset_obs"], O.observe(ref.state[i:i + 1])[0], 1e-6, "reset_obs")
        live = ~done
        dev_state = env.state.cpu().numpy()[:, :ref.state.shape[1]]
        assert not live.any() or np.abs(dev_state[live] - ref.state[live]).max() <= 1e-9 * max(1.0, np.abs(ref.state).max()), t
        ref.state[live] = dev_state[live]                                         # (follow the device: ulp differences must not pile up)
    assert n_trunc > 0 and (kind == 1 or n_term > 0), (n_term, n_trunc)
    ep, score, length = env.episode_stats()
    assert ep == n_term + n_trunc and length > 0
