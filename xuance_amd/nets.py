"""Network containers: parameters live in ONE flat fp32 device buffer (so gradient slabs, Adam moments and the
RCCL all-reduce are single contiguous ranges) and the forward/backward passes are *plans* of grouped MFMA GEMM
launches (xuance_amd/csrc/gemm.hip).

The classes mirror the reference modules they replace and export ``state_dict()`` with the reference's names, so
checkpoints are interchangeable:
  ActorCriticNet  <-> SharedActorCritic(Basic_MLP | Basic_Identical, CategoricalActorHead | GaussianActorHead,
                      ValueHead)   (rl_models/architectures/single_agent/actor_critic.py:40-72,
                      heads/actor_head.py:14-72, heads/critic_head.py:9-30, representations/mlp.py:10-60)
  SequentialNet   <-> a plain nn.Sequential of mlp_blocks (rl_models/modules/layers.py:16-33), used for the DQN
                      Q-head, the per-agent QMIX Q-network and the mixer hyper-networks.
"""
from collections import OrderedDict
from dataclasses import dataclass
from typing import List, Optional

import math

import numpy as np
import torch

from . import ops


def _align4(n):
    return (n + 3) // 4 * 4


class FlatParams:
    """Named fp32 tensors carved out of one flat buffer; every tensor starts on a 16-byte boundary."""

    def __init__(self, specs, device):
        """specs: (name, shape) -- or (name, shape, "packed"): the tensor starts right behind the previous one, without the
        16-byte alignment (two bias vectors that ONE stacked GEMM reads as a single bias: [b_actor; b_critic] with an odd
        number of actions)."""
        self.names, self.shapes, self.offsets = [], {}, {}
        off = end = 0
        for spec in specs:
            name, shape = spec[0], spec[1]
            if len(spec) > 2 and spec[2] == "packed":
                off = end
            self.names.append(name)
            self.shapes[name] = tuple(shape)
            self.offsets[name] = off
            n = 1
            for s in shape:
                n *= s
            end = off + n
            off = _align4(end)
        self.P = off
        self.device = device
        self.flat = torch.zeros(self.P, dtype=torch.float32, device=device)

    def view(self, name, flat=None):
        flat = self.flat if flat is None else flat
        n = 1
        for s in self.shapes[name]:
            n *= s
        o = self.offsets[name]
        return flat[o:o + n].view(self.shapes[name])

    def ptr(self, name, flat=None):
        flat = self.flat if flat is None else flat
        return flat.data_ptr() + 4 * self.offsets[name]

    def like(self):
        return torch.zeros_like(self.flat)

    def named_views(self, order, flat=None):
        return OrderedDict((n, self.view(n, flat)) for n in order)


@dataclass
class Layer:
    name: str            # reference parameter prefix ("actor.logits.0"); stacked layers carry several names
    K: int
    N: int
    act: Optional[str]
    in_level: int
    in_off: int
    out_level: int
    out_off: int
    w_name: str = ""     # key of the (possibly stacked) weight in FlatParams
    b_name: str = ""


class Plan:
    """A feed-forward DAG of Linear(+act) layers executed level by level with grouped launches."""

    def __init__(self, params: FlatParams, widths: List[int], stages: List[List[Layer]]):
        self.params, self.widths, self.stages = params, widths, stages
        # the data-gradient GEMMs WRITE d(input): two layers of a stage must not read overlapping columns of a hidden
        # level (stack such layers into one, like the first layers of the dueling streams or of the mixer's hyper-networks)
        for stage in stages:
            spans = [(L.in_level, L.in_off, L.in_off + L.K) for L in stage if L.in_level > 0]
            for i, a in enumerate(spans):
                for b in spans[i + 1:]:
                    assert a[0] != b[0] or a[2] <= b[1] or b[2] <= a[1], "layers of one stage share hidden input columns"
        self.cap = 0
        self.acts, self.dacts = {}, {}

    def ensure(self, M):
        if M <= self.cap:
            return
        dev = self.params.device
        self.cap = M
        for lvl in range(1, len(self.widths)):
            self.acts[lvl] = torch.zeros(M, self.widths[lvl], device=dev)
            self.dacts[lvl] = torch.zeros(M, self.widths[lvl], device=dev)

    # -- pointer helpers -------------------------------------------------------------------------
    def _buf(self, store, lvl, off, x, ldx):
        if lvl == 0:
            return x.data_ptr() + 4 * off, ldx
        return store[lvl].data_ptr() + 4 * off, self.widths[lvl]

    def _split_k(self, L, M):
        """(workspace, K ranges) for a layer with a LONG reduction and few output tiles -- the 6 400 -> 512 layer of AC_CNN_Atari on a
        256-frame minibatch is 32 tiles walking 200 slabs each (135 us) -- or (None, 0): xrl_linear_fwd's split-K."""
        if L.K < 2048:
            return None, 0
        tiles, ks = ((M + 63) // 64) * ((L.N + 63) // 64), 1
        # (round 6, measured: K ranges below 128 reduction steps -- 32 ranges for a 16-row acting pass -- are slower: PPO-Atari rollout
        #  7.08 -> 8.27 ms)
        while tiles * ks < 384 and L.K // (ks * 2) >= 128 and ks < 16:
            ks *= 2
        if ks == 1:
            return None, 0
        ws = getattr(self, "_skw", None)
        if ws is None:
            ws = self._skw = {}
        key = (L.w_name, ks)
        if key not in ws or ws[key].numel() < ks * M * L.N:
            ws[key] = torch.empty(ks * max(M, self.cap) * L.N, device=self.params.device)
        return ws[key], ks

    def forward(self, x, ldx, M, flat=None, stages=None):
        """x: device tensor holding the input rows with row stride ldx (floats).  stages: a subset of the plan's layers to run
        instead of all of them (ActorCriticNet.critic_stages: the value branch alone)."""
        self.ensure(M)
        P = self.params
        for stage in (self.stages if stages is None else stages):
            groups = []
            for L in stage:
                a, lda = self._buf(self.acts, L.in_level, L.in_off, x, ldx)
                c, ldc = self._buf(self.acts, L.out_level, L.out_off, x, ldx)
                aux, ks = self._split_k(L, M)
                groups.append(ops.gemm_desc(a, P.ptr(L.w_name, flat), c, M, L.N, L.K, lda, L.K, ldc,
                                            bias=P.ptr(L.b_name, flat), act=L.act, aux=aux.data_ptr() if ks else None, ldaux=ks))
            ops.linear_fwd(groups)
        return self.acts[len(self.widths) - 1]

    # -- the whole plan as ONE launch (xrl_mlp_chain_fwd, csrc/mlp_chain.hip; round 6) ----------------------------------------------
    def chain_job(self, x, ldx, M, flat=None, levels=None):
        """The plan as a job of ops.mlp_chain_desc: layers in stage order, every level written back to self.acts (`levels`: only
        those -- e.g. the output level of a pass that no backward follows).  None if the launch cannot take the plan: more than 8
        layers / 6 levels, a split-K layer, a layer input that does not start on a 16-byte column."""
        self.ensure(M)
        layers = [L for stage in self.stages for L in stage]
        if len(layers) > 8 or len(self.widths) > 6 or any(self._split_k(L, M)[1] for L in layers) or any(L.in_off & 3 for L in layers):
            return None
        P = self.params
        base = (P.flat if flat is None else flat).data_ptr()
        out = {lvl: (self.acts[lvl].data_ptr(), self.widths[lvl]) for lvl in range(1, len(self.widths)) if levels is None or lvl in levels}
        return dict(x=x.data_ptr() if isinstance(x, torch.Tensor) else int(x), ldx=ldx, M=M, params=base, level_width=list(self.widths), out=out,
                    layers=[dict(w_off=P.offsets[L.w_name], b_off=P.offsets[L.b_name], K=L.K, N=L.N, act=ops.ACT[L.act], in_level=L.in_level,
                                 in_off=L.in_off, out_level=L.out_level, out_off=L.out_off) for L in layers])

    @staticmethod
    def forward_chain(items, levels=None):
        """forward_many as ONE launch: items = [(plan, x, ldx, M, flat)], at most 4.  Returns each plan's output level, or None when
        one of the plans does not fit the launch (the caller then takes forward_many).  Same numbers, bit for bit."""
        if not (1 <= len(items) <= 4) or not ops.fast_kernels_enabled():
            return None
        jobs = [plan.chain_job(x, ldx, M, flat, levels) for plan, x, ldx, M, flat in items]
        if any(j is None for j in jobs):
            return None
        desc = ops.mlp_chain_desc(jobs)
        nb = ops.mlp_chain_lds_bytes(desc)
        if nb < 0 or nb > 160 * 1024:
            return None
        ops.mlp_chain_fwd(desc)
        return [it[0].acts[len(it[0].widths) - 1] for it in items]

    @staticmethod
    def forward_many(items, skip_last=False):
        """Several independent plans as ONE grouped launch per stage (an eval network and its target twin; the agent
        networks and the mixer's hyper-networks): items = [(plan, x, ldx, M, flat)]; stage i of every plan that has one
        goes into launch i.  Returns each plan's output level.  skip_last: every plan stops in front of its last stage (the
        caller's own launch computes it: xrl_dqn_head_td)."""
        depth = max(len(it[0].stages) for it in items)
        for plan, x, ldx, M, flat in items:
            plan.ensure(M)
        for si in range(depth):
            groups = []
            for plan, x, ldx, M, flat in items:
                P = plan.params
                if skip_last and si == len(plan.stages) - 1:
                    continue
                for L in (plan.stages[si] if si < len(plan.stages) else ()):
                    a, lda = plan._buf(plan.acts, L.in_level, L.in_off, x, ldx)
                    c, ldc = plan._buf(plan.acts, L.out_level, L.out_off, x, ldx)
                    groups.append(ops.gemm_desc(a, P.ptr(L.w_name, flat), c, M, L.N, L.K, lda, L.K, ldc,
                                                bias=P.ptr(L.b_name, flat), act=L.act))
            if groups:
                ops.linear_fwd(groups)
        return [it[0].acts[len(it[0].widths) - 1] for it in items]

    def backward(self, x, ldx, M, slabs, n_split, flat=None, dx0=None, defer_wgrad=None, skip_last_dg=False, weights_only=False):
        """dacts[last] must hold d loss / d (pre-activation) of the last level. Writes weight/bias gradient
        partials into slabs[s][layout of params].  dx0: optional [M, widths[0]] tensor receiving d loss / d input.
        defer_wgrad: a list -> only the data-gradient chain is launched here and the weight-gradient GEMM descriptors are
        appended to the list; the caller launches them together (ops.linear_bwd_weight) once every plan of the update has
        run its chain: the weight gradients depend on nothing but the dacts, so one grouped launch replaces one per layer."""
        P = self.params
        stride = slabs.shape[1]
        for si in reversed(range(len(self.stages))):
            stage = self.stages[si]
            wg, dg = [], []
            for L in stage:
                dy, lddy = self._buf(self.dacts, L.out_level, L.out_off, x, ldx)
                a, lda = self._buf(self.acts, L.in_level, L.in_off, x, ldx)
                wg.append(ops.gemm_desc(dy, a, slabs.data_ptr() + 4 * P.offsets[L.w_name], M, L.N, L.K, lddy, lda, L.K,
                                        dbias=slabs.data_ptr() + 4 * P.offsets[L.b_name]))
                if L.in_level > 0:
                    dx, lddx = self._buf(self.dacts, L.in_level, L.in_off, x, ldx)
                    aux, ldaux = self._buf(self.acts, L.in_level, L.in_off, x, ldx)
                    prev_act = self._act_of(L.in_level, L.in_off)
                    dg.append(ops.gemm_desc(dy, P.ptr(L.w_name, flat), dx, M, L.K, L.N, lddy, L.K, lddx,
                                            aux=aux, ldaux=ldaux, act=prev_act))
                elif dx0 is not None:
                    dg.append(ops.gemm_desc(dy, P.ptr(L.w_name, flat), dx0.data_ptr() + 4 * L.in_off, M, L.K, L.N, lddy,
                                            L.K, self.widths[0]))
            if defer_wgrad is not None:
                defer_wgrad.extend(wg)
            else:
                ops.linear_bwd_weight(wg, n_split, stride)
            if weights_only:                                                  # (every data gradient came from the caller's launch)
                continue
            if dg and not (skip_last_dg and si == len(self.stages) - 1):     # (skip_last_dg: the caller's launch already wrote
                ops.linear_bwd_data(dg)                                       #  the gradient in front of the last stage)

    @staticmethod
    def backward_many(items, slabs, n_split):
        """The backward passes of several independent plans together: items = [(plan, x, ldx, M)], every plan's dacts[last]
        filled.  Step k launches the data-gradient GEMMs of each plan's k-th stage from the end as one group; all weight
        gradients follow as one grouped launch (chunks of 8 groups)."""
        wg, depth = [], max(len(it[0].stages) for it in items)
        for k in range(depth):
            dg = []
            for plan, x, ldx, M in items:
                si = len(plan.stages) - 1 - k
                if si < 0:
                    continue
                P = plan.params
                for L in plan.stages[si]:
                    dy, lddy = plan._buf(plan.dacts, L.out_level, L.out_off, x, ldx)
                    a, lda = plan._buf(plan.acts, L.in_level, L.in_off, x, ldx)
                    wg.append(ops.gemm_desc(dy, a, slabs.data_ptr() + 4 * P.offsets[L.w_name], M, L.N, L.K, lddy, lda, L.K,
                                            dbias=slabs.data_ptr() + 4 * P.offsets[L.b_name]))
                    if L.in_level > 0:
                        dx, lddx = plan._buf(plan.dacts, L.in_level, L.in_off, x, ldx)
                        aux, ldaux = plan._buf(plan.acts, L.in_level, L.in_off, x, ldx)
                        dg.append(ops.gemm_desc(dy, P.ptr(L.w_name, None), dx, M, L.K, L.N, lddy, L.K, lddx,
                                                aux=aux, ldaux=ldaux, act=plan._act_of(L.in_level, L.in_off)))
            for i in range(0, len(dg), 8):
                ops.linear_bwd_data(dg[i:i + 8])
        for i in range(0, len(wg), 8):
            ops.linear_bwd_weight(wg[i:i + 8], n_split, slabs.shape[1])

    def backward_grouped(self, x, ldx, M, slabs, n_split, flat=None, dx0=None, skip_last_dg=False, weights_only=False):
        """backward() with the data-gradient chain first and ALL weight gradients of the plan as one grouped launch
        (chunks of 8 groups): same kernels per layer, fewer launches."""
        wg = []
        self.backward(x, ldx, M, slabs, n_split, flat=flat, dx0=dx0, defer_wgrad=wg, skip_last_dg=skip_last_dg,
                      weights_only=weights_only)
        for i in range(0, len(wg), 8):
            ops.linear_bwd_weight(wg[i:i + 8], n_split, slabs.shape[1])

    def _act_of(self, lvl, off):
        for stage in self.stages:
            for L in stage:
                if L.out_level == lvl and L.out_off <= off < L.out_off + L.N:
                    return L.act
        raise KeyError((lvl, off))


def _orthogonal(shape, gen_seeded=True):
    w = torch.empty(shape)
    torch.nn.init.orthogonal_(w)      # same initializer call the reference makes (agent.py:133, layers.py:23-26)
    return w


class ActorCriticNet:
    """SharedActorCritic re-laid-out for the device: first actor/critic hidden layers are stacked into one GEMM."""

    def __init__(self, obs_dim, action_dim, dist="categorical", representation_hidden=(128,), actor_hidden=(128,),
                 critic_hidden=(128,), activation="leaky_relu", activation_action=None, device="cuda", init=True, head_rep_layers=0):
        """head_rep_layers = k > 0 (with no shared representation): the first k layers of each head stack are that head's OWN copy of the
        representation -- the reference's ActorCritic model of A2C_Agent (architectures/single_agent/actor_critic.py:75-125) -- and
        state_dict() / load_state_dict() speak that module's key names (actor.representation.model.*, actor.actor_head.logits.*,
        critic.representation.model.*, critic.critic_head.values.*), so its checkpoints load here and ours there."""
        assert dist in ("categorical", "gaussian")
        self.obs_dim, self.action_dim, self.dist = obs_dim, action_dim, dist
        self.activation, self.activation_action = activation, activation_action
        rep, ah, ch = list(representation_hidden or []), list(actor_hidden), list(critic_hidden)
        assert len(ah) == len(ch) and len(ah) >= 1, "actor/critic hidden stacks must have equal depth >= 1"
        akey = "actor.logits" if dist == "categorical" else "actor.mu"
        self.ref_order = []           # reference state_dict order: representation, actor, critic
        specs = []
        feat = obs_dim
        rep_layers = []
        for i, h in enumerate(rep):
            rep_layers.append((f"representation.model.{2 * i}", feat, h))
            feat = h
        a_layers, c_layers = [], []
        fa = fc = feat
        for i, (ha, hc) in enumerate(zip(ah, ch)):
            a_layers.append((f"{akey}.{2 * i}", fa, ha))
            c_layers.append((f"critic.values.{2 * i}", fc, hc))
            fa, fc = ha, hc
        a_out = (f"{akey}.{2 * len(ah)}", fa, action_dim)
        c_out = (f"critic.values.{2 * len(ch)}", fc, 1)
        for n, k, o in rep_layers:
            self.ref_order += [n + ".weight", n + ".bias"]
        if dist == "gaussian":
            pass
        for n, k, o in a_layers + [a_out]:
            self.ref_order += [n + ".weight", n + ".bias"]
        if dist == "gaussian":
            # nn.Module registers parameters before sub-modules' parameters: actor.log_std precedes actor.mu.*
            # (with one representation per head -- head_rep_layers, the reference's ActorCritic -- log_std is a parameter of
            #  actor.actor_head, behind actor.representation.*: in front of the first HEAD layer, not of the branch's first layer)
            idx = self.ref_order.index(f"{akey}.{2 * int(head_rep_layers or 0)}.weight")
            self.ref_order.insert(idx, "actor.log_std")
        for n, k, o in c_layers + [c_out]:
            self.ref_order += [n + ".weight", n + ".bias"]
        self.ext_names = {}                                           # internal name -> the reference module's key (head_rep_layers)
        if head_rep_layers:
            assert not rep and 0 < head_rep_layers <= len(ah)
            for i in range(len(ah) + 1):
                for suf in (".weight", ".bias"):
                    if i < head_rep_layers:
                        self.ext_names[f"{akey}.{2 * i}{suf}"] = f"actor.representation.model.{2 * i}{suf}"
                        self.ext_names[f"critic.values.{2 * i}{suf}"] = f"critic.representation.model.{2 * i}{suf}"
                    else:
                        j = 2 * (i - head_rep_layers)
                        self.ext_names[f"{akey}.{2 * i}{suf}"] = f"actor.actor_head.{akey.split('.')[1]}.{j}{suf}"
                        self.ext_names[f"critic.values.{2 * i}{suf}"] = f"critic.critic_head.values.{j}{suf}"
            if dist == "gaussian":
                self.ext_names["actor.log_std"] = "actor.actor_head.log_std"

        # physical layout: per level, [W_actor; W_critic] adjacent and [b_actor; b_critic] adjacent
        for n, k, o in rep_layers:
            specs += [(n + ".weight", (o, k)), (n + ".bias", (o,))]
        self._stack0 = None
        widths = [obs_dim] + [h for h in rep]
        stages = []
        lvl = 0
        for i, (n, k, o) in enumerate(rep_layers):
            stages.append([Layer(n, k, o, activation, lvl, 0, lvl + 1, 0, n + ".weight", n + ".bias")])
            lvl += 1
        # level of stacked first hidden layers (shared input): weights must be physically adjacent, same K
        (na, ka, oa), (nc, kc, oc) = a_layers[0], c_layers[0]
        specs += [(na + ".weight", (oa, ka)), (nc + ".weight", (oc, kc)), (na + ".bias", (oa,)), (nc + ".bias", (oc,))]
        assert (oa * ka) % 4 == 0 and oa % 4 == 0, "stacked actor/critic layer needs 16-byte aligned halves"
        widths.append(oa + oc)
        stages.append([Layer(na + "+" + nc, ka, oa + oc, activation, lvl, 0, lvl + 1, 0, na + ".weight", na + ".bias")])
        lvl += 1
        prev_a, prev_c = oa, oc
        for (na, ka, oa), (nc, kc, oc) in zip(a_layers[1:], c_layers[1:]):
            specs += [(na + ".weight", (oa, ka)), (na + ".bias", (oa,)), (nc + ".weight", (oc, kc)), (nc + ".bias", (oc,))]
            widths.append(oa + oc)
            stages.append([Layer(na, ka, oa, activation, lvl, 0, lvl + 1, 0, na + ".weight", na + ".bias"),
                           Layer(nc, kc, oc, activation, lvl, prev_a, lvl + 1, oa, nc + ".weight", nc + ".bias")])
            lvl += 1
            prev_a, prev_c = oa, oc
        (na, ka, oa), (nc, kc, oc) = a_out, c_out
        specs += [(na + ".weight", (oa, ka)), (na + ".bias", (oa,)), (nc + ".weight", (oc, kc)), (nc + ".bias", (oc,))]
        widths.append(action_dim + 1)
        stages.append([Layer(na, ka, oa, activation_action if dist == "gaussian" else None, lvl, 0, lvl + 1, 0,
                             na + ".weight", na + ".bias"),
                       Layer(nc, kc, 1, None, lvl, prev_a, lvl + 1, action_dim, nc + ".weight", nc + ".bias")])
        if dist == "gaussian":
            specs.append(("actor.log_std", (action_dim,)))
        self.params = FlatParams(specs, device)
        self.plan = Plan(self.params, widths, stages)
        self.head_ld = action_dim + 1
        # the value branch alone (forward_values): with no shared representation the critic's layers form a chain of their own -- its
        # half of the stacked first level, then its layer of every later stage
        self.critic_stages = None
        if not rep_layers:
            (na0, ka0, oa0), (nc0, kc0, oc0) = a_layers[0], c_layers[0]
            cs = [[Layer(nc0, kc0, oc0, activation, 0, 0, 1, oa0, nc0 + ".weight", nc0 + ".bias")]]
            cs += [[st[1]] for st in stages[1:]]
            self.critic_stages = cs
        if init:
            self.reset_parameters()

    # -- parameters --------------------------------------------------------------------------------
    def reset_parameters(self):
        """orthogonal_(gain=1) weights, zero biases, log_std = -1, drawn in the reference's construction order
        (representation, actor, critic) from torch's global CPU generator."""
        sd = OrderedDict()
        for name in self.ref_order:
            shape = self.params.shapes[name]
            if name == "actor.log_std":
                continue
            if name.endswith(".weight"):
                sd[name] = _orthogonal(shape)
            else:
                sd[name] = torch.zeros(shape)
        if self.dist == "gaussian":
            sd["actor.log_std"] = -torch.ones(self.action_dim)
        self.load_state_dict(sd)

    @property
    def state_keys(self):
        """Keys of state_dict(), in the reference module's order."""
        ext = getattr(self, "ext_names", {})
        return [ext.get(n, n) for n in self.ref_order]

    def state_dict(self):
        ext = getattr(self, "ext_names", {})
        return OrderedDict((ext.get(n, n), self.params.view(n).detach().clone()) for n in self.ref_order)

    def load_state_dict(self, sd):
        ext = getattr(self, "ext_names", {})
        for n in self.ref_order:
            e = ext.get(n, n)
            self.params.view(n).copy_(torch.as_tensor(sd[e if e in sd else n], dtype=torch.float32))

    def parameters(self):
        return [self.params.view(n) for n in self.ref_order]

    # -- compute -------------------------------------------------------------------------------------
    CHAIN_MAX_ROWS = 1024

    def forward(self, x, M, ldx=None):
        """Returns the head buffer [cap, action_dim+1]: columns [0,A) actor output, column A the value.  use_chain_forward = True
        (default False) and at most CHAIN_MAX_ROWS rows: the whole plan as ONE launch (Plan.forward_chain, xrl_mlp_chain_fwd) --
        bit-identical and measured no faster: 21.6 us against 19.6 us for the three launches of an acting pass
        (tools/probe_mlp_chain.py, profiles/r06_p_mlp_chain.json: every phase of the single workgroup per tile is a handful of
        dependent LDS / global round trips, where the per-stage launches overlap theirs across 16-64 workgroups)."""
        ld = self.obs_dim if ldx is None else ldx
        if M <= self.CHAIN_MAX_ROWS and getattr(self, "use_chain_forward", False):
            out = Plan.forward_chain([(self.plan, x, ld, M, None)])
            if out is not None:
                return out[0]
        return self.plan.forward(x, ld, M)

    def forward_values(self, x, M, ldx=None):
        """The head buffer with only column A (the value) computed: the critic branch alone where the network has no shared
        representation (half the launches' work; the bootstrap / value passes of a whole rollout), else forward()."""
        if self.critic_stages is None:
            return self.forward(x, M, ldx)
        return self.plan.forward(x, self.obs_dim if ldx is None else ldx, M, stages=self.critic_stages)

    @property
    def d_heads(self):
        return self.plan.dacts[len(self.plan.widths) - 1]

    def backward(self, x, M, slabs, n_split, ldx=None):
        self.plan.backward_grouped(x, self.obs_dim if ldx is None else ldx, M, slabs, n_split)


class ActorNet:
    """VanillaPolicyGradient(CategoricalActor | GaussianActor) (rl_models/architectures/single_agent/reinforce.py:6-31,
    actors/categorical_actors.py, gaussian_actors.py): representation + actor head, no critic.  Same compute surface as
    ActorCriticNet (forward -> head buffer [cap, action_dim], d_heads, backward), reference parameter names
    ``actor.representation.model.<i>``, ``actor.actor_head.{logits|mu}.<i>``, ``actor.actor_head.log_std``."""

    CHAIN_MAX_ROWS = ActorCriticNet.CHAIN_MAX_ROWS                     # (forward is ActorCriticNet's)

    def __init__(self, obs_dim, action_dim, dist="categorical", representation_hidden=(128,), actor_hidden=(128,),
                 activation="leaky_relu", activation_action=None, device="cuda", init=True):
        assert dist in ("categorical", "gaussian")
        self.obs_dim, self.action_dim, self.dist = obs_dim, action_dim, dist
        self.activation, self.activation_action = activation, activation_action
        specs, order, stages, widths = [], [], [], [obs_dim]
        feat, lvl = _seq_layers("actor.representation.model", obs_dim, list(representation_hidden or []), activation, "same",
                                0, specs, order, stages, widths)
        key = "actor.actor_head.logits" if dist == "categorical" else "actor.actor_head.mu"
        _seq_layers(key, feat, list(actor_hidden) + [action_dim], activation, activation_action if dist == "gaussian" else None,
                    lvl, specs, order, stages, widths)
        self.log_std_name = "actor.actor_head.log_std"
        if dist == "gaussian":
            specs.append((self.log_std_name, (action_dim,)))
            order.insert(order.index(key + ".0.weight"), self.log_std_name)      # nn.Module: own parameters before sub-modules
        self.ref_order = order
        self.params = FlatParams(specs, device)
        self.plan = Plan(self.params, widths, stages)
        self.head_ld = action_dim
        if init:
            for name in order:
                v = self.params.view(name)
                if name == self.log_std_name:
                    v.fill_(-1.0)
                else:
                    v.copy_(_orthogonal(v.shape)) if name.endswith(".weight") else v.zero_()

    state_dict = ActorCriticNet.state_dict
    load_state_dict = ActorCriticNet.load_state_dict
    parameters = ActorCriticNet.parameters
    forward = ActorCriticNet.forward
    d_heads = ActorCriticNet.d_heads
    backward = ActorCriticNet.backward


class SequentialNet:
    """nn.Sequential of mlp_blocks with reference-style parameter names ``<prefix>.<2i>.{weight,bias}``.

    ``segments``: list of (prefix, [hidden...], last_act) chained one after another, e.g. the DQN network is
    [("representation.model", [64], act), ("eval_Q_head.q_value", [64, n_actions], None)]."""

    def __init__(self, in_dim, segments, activation="relu", device="cuda", init=True, params=None):
        self.in_dim, self.activation = in_dim, activation
        specs, stages, widths = [], [], [in_dim]
        self.ref_order = []
        feat, lvl = in_dim, 0
        for prefix, sizes, last_act in segments:
            for i, h in enumerate(sizes):
                n = f"{prefix}.{2 * i}"
                act = activation if (i < len(sizes) - 1 or last_act == "same") else last_act
                specs += [(n + ".weight", (h, feat)), (n + ".bias", (h,))]
                self.ref_order += [n + ".weight", n + ".bias"]
                stages.append([Layer(n, feat, h, act, lvl, 0, lvl + 1, 0, n + ".weight", n + ".bias")])
                widths.append(h)
                feat, lvl = h, lvl + 1
        self.out_dim = feat
        self.params = FlatParams(specs, device) if params is None else params
        self.plan = Plan(self.params, widths, stages)
        if init and params is None:
            self.reset_parameters()

    def reset_parameters(self):
        for name in self.ref_order:
            v = self.params.view(name)
            if name.endswith(".weight"):
                v.copy_(_orthogonal(v.shape))
            else:
                v.zero_()

    def state_dict(self, flat=None):
        return OrderedDict((n, self.params.view(n, flat).detach().clone()) for n in self.ref_order)

    def load_state_dict(self, sd, flat=None):
        for n in self.ref_order:
            self.params.view(n, flat).copy_(torch.as_tensor(sd[n], dtype=torch.float32))

    def forward(self, x, M, ldx=None, flat=None):
        return self.plan.forward(x, self.in_dim if ldx is None else ldx, M, flat)

    @property
    def d_out(self):
        return self.plan.dacts[len(self.plan.widths) - 1]

    def backward(self, x, M, slabs, n_split, ldx=None, flat=None):
        self.plan.backward_grouped(x, self.in_dim if ldx is None else ldx, M, slabs, n_split, flat)


def _seq_layers(prefix, in_dim, sizes, activation, last_act, lvl0, specs, order, stages, widths):
    """Append a chain of mlp_blocks named ``<prefix>.<2i>`` to (specs, order, stages, widths)."""
    feat, lvl = in_dim, lvl0
    for i, h in enumerate(sizes):
        n = f"{prefix}.{2 * i}"
        act = activation if (i < len(sizes) - 1 or last_act == "same") else last_act
        specs += [(n + ".weight", (h, feat)), (n + ".bias", (h,))]
        order += [n + ".weight", n + ".bias"]
        stages.append([Layer(n, feat, h, act, lvl, 0, lvl + 1, 0, n + ".weight", n + ".bias")])
        widths.append(h)
        feat, lvl = h, lvl + 1
    return feat, lvl


def _q_head_layers(feat, q_hidden, n_actions, activation, dueling, lvl, specs, order, stages, widths):
    """BasicQhead (q_head.py:8-39) or DuelingQValueHead (q_head.py:42-80) behind `feat` features at activation level `lvl`: appends to
    specs / order / stages / widths (shared by DeepQNet and DeepQCNN)."""
    if not dueling:
        _seq_layers("eval_Q_head.q_value", feat, list(q_hidden) + [n_actions], activation, None, lvl, specs, order,
                    stages, widths)
    else:
        # DuelingQValueHead (q_head.py:42-80): v_model feat -> h/2 ... -> 1 and a_model feat -> h/2 ... -> A side by
        # side (two groups per launch); the output level is [advantages (A) | value], combined inside xrl_dqn_td
        v_specs, a_specs, v_order, a_order = [], [], [], []
        assert len(list(q_hidden)) >= 1, "dueling head: at least one hidden layer (the two streams' first layers are stacked)"
        vin, ain, k_in = 0, 0, feat
        for i, h in enumerate(list(q_hidden)):
            hh = h // 2
            nv, na = f"eval_Q_head.v_model.{2 * i}", f"eval_Q_head.a_model.{2 * i}"
            v_order += [nv + ".weight", nv + ".bias"]; a_order += [na + ".weight", na + ".bias"]
            if i == 0:
                # both streams read the same features: ONE stacked layer [v; a] (one data-gradient GEMM into the shared
                # input); the flat layout keeps the two weights, then the two biases, adjacent
                assert (hh * k_in) % 4 == 0 and hh % 4 == 0
                specs += [(nv + ".weight", (hh, k_in)), (na + ".weight", (hh, k_in)), (nv + ".bias", (hh,)), (na + ".bias", (hh,))]
                stages.append([Layer(nv + "+" + na, k_in, 2 * hh, activation, lvl, 0, lvl + 1, 0, nv + ".weight", nv + ".bias")])
            else:
                specs += [(nv + ".weight", (hh, k_in)), (nv + ".bias", (hh,)), (na + ".weight", (hh, k_in)), (na + ".bias", (hh,))]
                stages.append([Layer(nv, k_in, hh, activation, lvl, vin, lvl + 1, 0, nv + ".weight", nv + ".bias"),
                               Layer(na, k_in, hh, activation, lvl, ain, lvl + 1, hh, na + ".weight", na + ".bias")])
            widths.append(2 * hh)
            lvl, k_in, vin, ain = lvl + 1, hh, 0, hh
        i = len(list(q_hidden))
        nv, na = f"eval_Q_head.v_model.{2 * i}", f"eval_Q_head.a_model.{2 * i}"
        specs += [(nv + ".weight", (1, k_in)), (nv + ".bias", (1,)), (na + ".weight", (n_actions, k_in)), (na + ".bias", (n_actions,))]
        v_order += [nv + ".weight", nv + ".bias"]; a_order += [na + ".weight", na + ".bias"]
        stages.append([Layer(na, k_in, n_actions, None, lvl, ain, lvl + 1, 0, na + ".weight", na + ".bias"),
                       Layer(nv, k_in, 1, None, lvl, vin, lvl + 1, n_actions, nv + ".weight", nv + ".bias")])
        widths.append(n_actions + 1)
        order += v_order + a_order                          # state_dict order of the reference head: v_model, a_model



class DeepQNet:
    """DeepQNetwork with an MLP / identity representation (rl_models/architectures/single_agent/deep_q_network.py:19-99):
    eval and target networks share one parameter layout; the target lives in a second flat buffer."""

    def __init__(self, obs_dim, n_actions, representation_hidden=(), q_hidden=(64,), activation="relu", device="cuda",
                 init=True, dueling=False):
        self.obs_dim, self.n_actions, self.activation, self.dueling = obs_dim, n_actions, activation, bool(dueling)
        specs, order, stages, widths = [], [], [], [obs_dim]
        feat, lvl = _seq_layers("representation.model", obs_dim, list(representation_hidden or []), activation, "same",
                                0, specs, order, stages, widths)
        _q_head_layers(feat, q_hidden, n_actions, activation, dueling, lvl, specs, order, stages, widths)
        self.eval_order = order
        self.params = FlatParams(specs, device)
        self.target_flat = self.params.like()
        self.plan = Plan(self.params, widths, stages)            # eval network (forward + backward)
        self.target_plan = Plan(self.params, widths, stages)     # target network (forward only, on target_flat)
        rep = [k for k in order if k.startswith("representation.")]
        head = [k for k in order if k.startswith("eval_Q_head.")]
        # reference state_dict order: representation, target_representation, eval_Q_head, target_Q_head
        self.ref_order = rep + ["target_" + k for k in rep] + head + ["target_Q_head." + k[len("eval_Q_head."):] for k in head]
        self.trainable_order = rep + head
        if init:
            for name in order:
                v = self.params.view(name)
                v.copy_(_orthogonal(v.shape)) if name.endswith(".weight") else v.zero_()
            self.copy_target()

    def _target_key(self, k):
        if k.startswith("target_representation."):
            return k[len("target_"):]
        if k.startswith("target_Q_head."):
            return "eval_Q_head." + k[len("target_Q_head."):]
        return None

    def state_dict(self):
        out = OrderedDict()
        for k in self.ref_order:
            tk = self._target_key(k)
            out[k] = (self.params.view(tk, self.target_flat) if tk else self.params.view(k)).detach().clone()
        return out

    def load_state_dict(self, sd):
        for k in self.ref_order:
            tk = self._target_key(k)
            dst = self.params.view(tk, self.target_flat) if tk else self.params.view(k)
            dst.copy_(torch.as_tensor(sd[k], dtype=torch.float32))

    def copy_target(self):                                        # deep_q_network.py:95-99
        self.target_flat.copy_(self.params.flat)

    def forward(self, x, M, ldx=None):
        """x holds x.shape[0] >= M rows; all rows are evaluated, backward() differentiates the first M."""
        return self.plan.forward(x, self.obs_dim if ldx is None else ldx, x.shape[0])

    def target(self, x, M, ldx=None):
        return self.target_plan.forward(x, self.obs_dim if ldx is None else ldx, M, flat=self.target_flat)

    def forward_pair(self, X, M, double_q, skip_last=False):
        """Eval network on X[:M] (+ X[M:2M] under double-Q) and target network on X[M:2M] as grouped launches."""
        return Plan.forward_many([(self.plan, X, self.obs_dim, 2 * M if double_q else M, None),
                                  (self.target_plan, X[M:], self.obs_dim, M, self.target_flat)], skip_last=skip_last)

    def fused_head(self):
        """The last layer as xrl_dqn_head_td wants it, or None: a single Linear(H, n_actions) without activation behind a hidden
        level of its own (BasicQhead with at least one hidden layer; not the dueling streams)."""
        last = self.plan.stages[-1]
        if self.dueling or len(last) != 1 or len(self.plan.stages) < 2:
            return None
        L = last[0]
        if L.act not in (None, "none") or L.in_level < 1 or L.in_off != 0 or L.K != self.plan.widths[L.in_level] or L.N > 64:
            return None
        return L

    def head_td(self, M, double_q, actions, rewards, terminals, diag, partials, gamma, huber_delta=0.0):
        """Q layer + TD + the Q layer's data gradient (after forward_pair(..., skip_last=True)); backward(..., skip_last_dg=True)
        continues from there."""
        L, pl, tp = self.fused_head(), self.plan, self.target_plan
        lvl = L.in_level
        ops.dqn_head_td(h_eval=pl.acts[lvl], h_target=tp.acts[lvl], w_eval=self.params.ptr(L.w_name), b_eval=self.params.ptr(L.b_name),
                        w_target=self.params.ptr(L.w_name, self.target_flat), b_target=self.params.ptr(L.b_name, self.target_flat),
                        actions=actions, rewards=rewards, terminals=terminals, q_eval=pl.acts[L.out_level], q_target=tp.acts[L.out_level],
                        d_q=pl.dacts[L.out_level], d_h=pl.dacts[lvl], diag=diag, partials=partials, M=M, A=L.N, H=L.K,
                        ld_h=pl.widths[lvl], ld_q=pl.widths[L.out_level], double_q=int(double_q),
                        act=ops.ACT[pl._act_of(lvl, 0)], gamma=float(gamma), huber_delta=float(huber_delta))

    @property
    def d_out(self):
        return self.plan.dacts[len(self.plan.widths) - 1]

    def backward(self, x, M, slabs, n_split, skip_last_dg=False):
        self.plan.backward_grouped(x, self.obs_dim, M, slabs, n_split, skip_last_dg=skip_last_dg)


class MixingQNet:
    """MixingQNetwork(ModuleDict{group: DiscreteActionValueCritic(AgentFeatureEncoder(Basic_MLP))}, QMIX_Mixer)
    with parameter sharing (one group) and identity encoding 'none'
    (architectures/multi_agent/value_factorization.py:17-174, critics/base_critics.py:91-132, heads/q_mix_head.py:28-95).
    Agent network and mixer share ONE flat parameter buffer (one optimiser step); the targets are a second buffer."""

    def __init__(self, n_agents, obs_dim, n_actions, state_dim, representation_hidden=(64,), q_hidden=(64,),
                 mixer_hidden=32, hyper_hidden=32, activation="relu", group="shared", device="cuda", init=True,
                 use_rnn=False, fc_hidden=(64,), recurrent_hidden=64, mixer="QMIX", rnn="GRU"):
        self.n_agents, self.obs_dim, self.n_actions, self.state_dim = n_agents, obs_dim, n_actions, state_dim
        self.H, self.HH, self.group = mixer_hidden, hyper_hidden, group
        self.use_rnn, self.RH = bool(use_rnn), int(recurrent_hidden)
        assert rnn in ("GRU", "LSTM")
        self.lstm = bool(use_rnn) and rnn == "LSTM"
        self.G = (4 if self.lstm else 3) * int(recurrent_hidden)       # gate rows of weight_ih_l0 / weight_hh_l0
        N, H, HH, S = n_agents, mixer_hidden, hyper_hidden, state_dim
        specs, a_order, a_stages, a_widths = [], [], [], [obs_dim]
        pe = f"individual_q_networks.{group}"
        if not use_rnn:
            feat, lvl = _seq_layers(f"{pe}.representation.obs_representation.model", obs_dim, list(representation_hidden),
                                    activation, "same", 0, specs, a_order, a_stages, a_widths)
            _seq_layers(f"{pe}.critic_head.q_value", feat, list(q_hidden) + [n_actions], activation, None, lvl, specs,
                        a_order, a_stages, a_widths)
        else:
            # Basic_RNN (rnn.py:38-77): mlp blocks, then nn.GRU; the input-side GRU product is the last layer of the
            # "pre" plan (no activation), the recurrence is xrl_gru_forward, the Q head is the "post" plan.
            assert recurrent_hidden == 64, "xrl_gru_forward keeps one hidden unit per lane: recurrent_hidden_size must be 64"
            rp, G = f"{pe}.representation.obs_representation", self.G
            feat, lvl = _seq_layers(f"{rp}.mlp", obs_dim, list(fc_hidden), activation, "same", 0, specs, a_order,
                                    a_stages, a_widths)
            self.w_ih, self.w_hh, self.b_ih, self.b_hh = (f"{rp}.rnn.weight_ih_l0", f"{rp}.rnn.weight_hh_l0",
                                                          f"{rp}.rnn.bias_ih_l0", f"{rp}.rnn.bias_hh_l0")
            specs += [(self.w_ih, (G, feat)), (self.w_hh, (G, recurrent_hidden)), (self.b_ih, (G,)), (self.b_hh, (G,))]
            a_order += [self.w_ih, self.w_hh, self.b_ih, self.b_hh]               # nn.GRU parameter order
            a_stages.append([Layer(f"{rp}.rnn.ih", feat, G, None, lvl, 0, lvl + 1, 0, self.w_ih, self.b_ih)])
            a_widths.append(G)
            q_stages, q_widths = [], [recurrent_hidden]
            _seq_layers(f"{pe}.critic_head.q_value", recurrent_hidden, list(q_hidden) + [n_actions], activation, None, 0,
                        specs, a_order, q_stages, q_widths)
        # mixer: "QMIX" (QMIX_Mixer, q_mix_head.py:28-95), "VDN" (VDN_Mixer: sum over agents) or "Independent"
        # (IndependentMixer, IQL): the last two have no parameters (vdn_agents.py:71-72, iql_agents.py:71)
        self.mixer = mixer
        mixer_order, m_widths, m_stages = [], None, None
        if mixer == "QMIX":
            # mixer hyper-networks: the three ReLU first layers are stacked into one GEMM ([hyper_w_1.0; hyper_w_2.0;
            # hyper_b_2.0]), hyper_b_1 is a second group of the same launch; second layers are three groups.
            m = "eval_Qtot"
            firsts = [f"{m}.hyper_w_1.0", f"{m}.hyper_w_2.0", f"{m}.hyper_b_2.0"]
            specs += [(n + ".weight", (HH, S)) for n in firsts] + [(n + ".bias", (HH,)) for n in firsts]
            specs += [(f"{m}.hyper_b_1.weight", (H, S)), (f"{m}.hyper_b_1.bias", (H,)),
                      (f"{m}.hyper_w_1.2.weight", (N * H, HH)), (f"{m}.hyper_w_1.2.bias", (N * H,)),
                      (f"{m}.hyper_w_2.2.weight", (H, HH)), (f"{m}.hyper_w_2.2.bias", (H,)),
                      (f"{m}.hyper_b_2.2.weight", (1, HH)), (f"{m}.hyper_b_2.2.bias", (1,))]
            assert (HH * S) % 4 == 0 and HH % 4 == 0
            self.raw_width = N * H + H + 1
            m_widths = [S, 3 * HH + H, (self.raw_width + 3) // 4 * 4]
            m_stages = [[Layer("+".join(firsts), S, 3 * HH, "relu", 0, 0, 1, 0, firsts[0] + ".weight", firsts[0] + ".bias"),
                         Layer(f"{m}.hyper_b_1", S, H, None, 0, 0, 1, 3 * HH, f"{m}.hyper_b_1.weight", f"{m}.hyper_b_1.bias")],
                        [Layer(f"{m}.hyper_w_1.2", HH, N * H, None, 1, 0, 2, 0, f"{m}.hyper_w_1.2.weight", f"{m}.hyper_w_1.2.bias"),
                         Layer(f"{m}.hyper_w_2.2", HH, H, None, 1, HH, 2, N * H, f"{m}.hyper_w_2.2.weight", f"{m}.hyper_w_2.2.bias"),
                         Layer(f"{m}.hyper_b_2.2", HH, 1, None, 1, 2 * HH, 2, N * H + H, f"{m}.hyper_b_2.2.weight", f"{m}.hyper_b_2.2.bias")]]
            for n in (f"{m}.hyper_w_1.0", f"{m}.hyper_w_1.2", f"{m}.hyper_w_2.0", f"{m}.hyper_w_2.2", f"{m}.hyper_b_1",
                      f"{m}.hyper_b_2.0", f"{m}.hyper_b_2.2"):
                mixer_order += [n + ".weight", n + ".bias"]
        self.params = FlatParams(specs, device)
        self.target_flat = self.params.like()
        if not use_rnn:
            self.agent_plan = Plan(self.params, a_widths, a_stages)
            self.agent_target_plan = Plan(self.params, a_widths, a_stages)
        else:
            # [update eval, update target, acting]: separate activation buffers (captured graphs keep their pointers)
            self.pre_plans = [Plan(self.params, a_widths, a_stages) for _ in range(3)]
            self.post_plans = [Plan(self.params, q_widths, q_stages) for _ in range(3)]
            self._seq_ws = {}
        if mixer == "QMIX":
            self.mixer_plan = Plan(self.params, m_widths, m_stages)
            self.mixer_target_plan = Plan(self.params, m_widths, m_stages)
        self.trainable_order = a_order + mixer_order
        # reference order: individual_q_networks, target_individual_q_networks, eval_Qtot, target_Qtot
        self.ref_order = a_order + ["target_" + k for k in a_order] + mixer_order + \
            ["target_Qtot." + k[len("eval_Qtot."):] for k in mixer_order]
        if init:
            for name in self.trainable_order:
                v = self.params.view(name)
                if (name.endswith(".weight") or "rnn.weight_" in name) and name.startswith("individual_q_networks"):
                    v.copy_(_orthogonal(v.shape))               # gru_block: orthogonal on 2-D weights, 0 on biases (layers.py:91-97)
                elif name.startswith("eval_Qtot"):       # nn.Linear default init (q_mix_head.py builds plain nn.Linear)
                    fan_in = self.params.shapes[name[:-5] + ".weight" if name.endswith(".bias") else name][1]
                    bound = 1.0 / fan_in ** 0.5
                    v.copy_((torch.rand(v.shape) * 2 - 1) * bound)
                else:
                    v.zero_()
            self.copy_target()

    # ---------------------------------------------------------------- recurrent agents (time-major sequences)
    def seq_workspace(self, which, R, T1):
        key = (which, R, T1)
        ws = self._seq_ws.get(key)
        if ws is None:
            dev, H = self.params.device, self.RH
            ws = {"hs": torch.zeros((T1 + 1) * R, H, device=dev), "gates": torch.zeros(T1 * R, 4 * H, device=dev)}
            if self.lstm:
                ws["cs"] = torch.zeros((T1 + 1) * R, H, device=dev)
            if which == 0:
                ws["d_hs"] = torch.zeros(T1 * R, H, device=dev)
                if not self.lstm:
                    ws["d_gh"] = torch.zeros(T1 * R, 3 * H, device=dev)
            self._seq_ws[key] = ws
            self.pre_plans[which].ensure(T1 * R)
            self.post_plans[which].ensure(T1 * R)
        return ws

    def _recurrence(self, gi, ws, R, T1, flat, keep, h0=None, c0=None, reset=None, h_last=None, c_last=None, second=None):
        """The serial part between the plan below and the plan above: xrl_gru_forward or xrl_lstm_forward (`rnn: "LSTM"`).
        second = (gi2, ws2, flat2): the target network's sequences in the same launch."""
        P, H, G = self.params, self.RH, self.G
        kw = dict(gi=gi, w_hh=P.ptr(self.w_hh, flat), b_hh=P.ptr(self.b_hh, flat), h0=h0, reset=reset, hs=ws["hs"],
                  gates=ws["gates"] if keep else None, h_last=h_last, R=R, T1=T1, H=H, ld_gi=G)
        if second is not None:
            gi2, ws2, flat2 = second
            kw.update(gi2=gi2, w_hh2=P.ptr(self.w_hh, flat2), b_hh2=P.ptr(self.b_hh, flat2), hs2=ws2["hs"])
        if self.lstm:
            ops.lstm_forward(c0=c0, cs=ws["cs"] if keep else None, c_last=c_last, **kw)
        else:
            ops.gru_forward(**kw)

    def agent_forward_seq(self, X, R, T1, which=0, h0=None, reset=None, h_last=None, c0=None, c_last=None):
        """Q values of R sequences over T1 steps.  X [T1*R, obs_dim] time-major (row t*R + r) -> [T1*R, n_actions].
        which: 0 = eval network (keeps what BPTT needs), 1 = target network, 2 = eval network for acting."""
        flat = self.target_flat if which == 1 else None
        ws = self.seq_workspace(which, R, T1)
        gi = self.pre_plans[which].forward(X, self.obs_dim, T1 * R, flat=flat)
        self._recurrence(gi, ws, R, T1, flat, which == 0, h0=h0, c0=c0, reset=reset, h_last=h_last, c_last=c_last)
        return self.post_plans[which].forward(ws["hs"][R:], self.RH, T1 * R, flat=flat)

    def act_step(self, X, R, h, reset=None, c=None, fused=True, select=None):
        """Q values [R, n_actions] of ONE acting step from observations X [R, obs_dim] and the carried state h [R, H] (and c
        for LSTM agents), which is replaced by the new state; rows with reset != 0 start from zeros.  GRU agents whose
        weights fit LDS take the one-launch path (xrl_marl_act_gru); `act_image().refresh()` must have run since the
        parameters last changed.  select: keyword arguments of ops.marl_select_actions (without q / R / A / ld): the
        epsilon-greedy selection on these Q values, in the same launch on the one-launch path."""
        st = self.act_image() if fused and not self.lstm else None
        if st is None:
            q = self.agent_forward_seq(X, R, 1, which=2, h0=h, reset=reset, h_last=h, c0=c, c_last=c) if self.use_rnn else \
                self.agent_plan.forward(X, self.obs_dim, R)
            if select is not None:
                ops.marl_select_actions(q=q, R=R, A=self.n_actions, ld=self.n_actions, **select)
            return q
        return st.launch(X, R, h, reset, self.act_q_buffer(R), select=select)

    def act_q_buffer(self, R):
        """(allocated outside any graph capture: callers that capture act_step touch it first)"""
        q = self._act_q.get(R)
        if q is None:
            q = self._act_q[R] = torch.zeros(R, self.n_actions, device=self.params.device)
        return q

    def act_image(self):
        """The acting launch's weight image (ops.MarlActGruState), or None when that launch cannot run this network."""
        if not hasattr(self, "_act_state"):
            self._act_state, self._act_q = None, {}
            if not self.lstm:
                try:
                    st = ops.MarlActGruState(self)
                    if st.lds_bytes <= 160 * 1024:
                        self._act_state = st
                except AssertionError:                          # a layer arrangement the launch does not cover
                    self._act_state = None
        return self._act_state

    def agent_forward_seq_pair(self, X, R, T1, ride_along=()):
        """Eval and target networks over the same sequences (iql_learner.py:41-57): the layers below and above the
        recurrence as grouped launches (eval + target in one), the two recurrences as one dual launch.
        ride_along: forward_many items of independent plans (the mixer's hyper-networks) that share the launches of the
        layers above the recurrence.  Returns [Q_eval, Q_target, outputs of the ride-along plans...]."""
        H, M, tf = self.RH, T1 * R, self.target_flat
        ws0, ws1 = self.seq_workspace(0, R, T1), self.seq_workspace(1, R, T1)
        # (round 6, use_chain_forward = True; off by default: measured slower) the layers below the recurrence as ONE launch, the layers
        # above it (+ the ride-along plans) as one: Plan.forward_chain (xrl_mlp_chain_fwd; bit-identical to the launch per stage)
        chain = getattr(self, "use_chain_forward", False) and M <= 1024    # (default False; 5 856 rows: 106 -> 123 us per update)
        pre = [(self.pre_plans[0], X, self.obs_dim, M, None), (self.pre_plans[1], X, self.obs_dim, M, tf)]
        out = Plan.forward_chain(pre) if chain else None
        gi0, gi1 = out if out is not None else Plan.forward_many(pre)
        self._recurrence(gi0, ws0, R, T1, None, True, second=(gi1, ws1, tf))
        post = [(self.post_plans[0], ws0["hs"][R:], H, M, None), (self.post_plans[1], ws1["hs"][R:], H, M, tf)] + list(ride_along)
        out = Plan.forward_chain(post) if chain else None
        return out if out is not None else Plan.forward_many(post)

    def agent_backward_seq(self, X, R, T1, slabs, n_split, defer_wgrad=None):
        """post_plans[0].dacts[last] holds dLoss/dQ [T1*R, A]: data-gradient chain Q head -> BPTT -> layers below the
        recurrence, then every weight gradient of the agent network (Q head, W_hh, W_ih, fc) as ONE grouped launch."""
        P, H, G, M = self.params, self.RH, self.G, T1 * R
        ws, pre, post = self.seq_workspace(0, R, T1), self.pre_plans[0], self.post_plans[0]
        wg = [] if defer_wgrad is None else defer_wgrad
        post.backward(ws["hs"][R:], H, M, slabs, n_split, dx0=ws["d_hs"], defer_wgrad=wg)
        d_gi = pre.dacts[len(pre.widths) - 1]
        if self.lstm:      # one gate-gradient buffer serves the input and the hidden side (csrc/lstm.hip)
            ops.lstm_backward(d_hs=ws["d_hs"], cs=ws["cs"], gates=ws["gates"], w_hh=P.ptr(self.w_hh), d_gates=d_gi, R=R, T1=T1,
                              H=H, ld_dhs=H, ld_dg=G)
            d_gh = d_gi
        else:
            ops.gru_backward(d_hs=ws["d_hs"], hs=ws["hs"], gates=ws["gates"], w_hh=P.ptr(self.w_hh), d_gi=d_gi, d_gh=ws["d_gh"],
                             d_h0=None, R=R, T1=T1, H=H, ld_dhs=H, ld_dgi=G)
            d_gh = ws["d_gh"]
        wg.append(ops.gemm_desc(d_gh.data_ptr(), ws["hs"].data_ptr(), slabs.data_ptr() + 4 * P.offsets[self.w_hh],
                                M, G, H, G, H, H, dbias=slabs.data_ptr() + 4 * P.offsets[self.b_hh]))
        pre.backward(X, self.obs_dim, M, slabs, n_split, defer_wgrad=wg)
        if defer_wgrad is None:
            ops.linear_bwd_weight(wg, n_split, slabs.shape[1])

    def _target_key(self, k):
        if k.startswith("target_individual_q_networks."):
            return k[len("target_"):]
        if k.startswith("target_Qtot."):
            return "eval_Qtot." + k[len("target_Qtot."):]
        return None

    def state_dict(self):
        out = OrderedDict()
        for k in self.ref_order:
            tk = self._target_key(k)
            out[k] = (self.params.view(tk, self.target_flat) if tk else self.params.view(k)).detach().clone()
        return out

    def load_state_dict(self, sd):
        for k in self.ref_order:
            tk = self._target_key(k)
            dst = self.params.view(tk, self.target_flat) if tk else self.params.view(k)
            dst.copy_(torch.as_tensor(sd[k], dtype=torch.float32))
        self.touched()

    def copy_target(self):                                        # value_factorization.py:169-174
        self.target_flat.copy_(self.params.flat)
        self.touched()

    def touched(self):
        """Somebody other than the optimiser launch wrote the parameters: the derived weight images (the one-launch
        update's, the acting launch's) are rebuilt on their next use (their owners compare `version`)."""
        self.version = getattr(self, "version", 0) + 1


class ConvStack:
    """Conv2d(k, s, pad=(k-s)//2) + ReLU layers followed by AdaptiveMaxPool2d((1,1)) (Basic_CNN, cnn.py:11-50) on the
    HIP engine: im2col (column order c, kh, kw == the reference weight layout) + the fp32-MFMA GEMMs, NHWC end to end.
    `Workspace` objects hold the per-pass buffers: one with saved columns for the differentiated pass, scratch ones for
    the no-gradient passes (target network, double-Q, acting)."""

    class Workspace:
        def __init__(self, stack, rows, keep):
            dev = stack.params.device
            self.rows, self.keep = rows, keep
            self.col, self.y, self.dy, self.dcol = [], [], [], []
            self.x_in = None                                   # the frames the first layer read (its weight gradient reads them again)
            for (H, W, C, k, s, p, OH, OW, F) in stack.geo:
                M, K = rows * OH * OW, C * k * k
                self.y.append(torch.empty(M, F, device=dev))
                if keep:
                    self.dy.append(torch.empty(M, F, device=dev))
                if stack.implicit:                             # (implicit GEMMs: no column matrices)
                    continue
                self.col.append(torch.empty(M, K, device=dev))
                if keep:
                    self.dcol.append(torch.empty(M, K, device=dev))
            # split-K workspaces of the forward GEMMs (layers whose row-tile count alone cannot fill the chip)
            self.ksplit = [1 if stack.implicit else stack.ksplit_for(rows * OH * OW, F, C * k * k)
                           for (H, W, C, k, s, p, OH, OW, F) in stack.geo]
            self.skw = [torch.empty(2 * ks * rows * OH * OW * F, device=dev) if ks > 1 else None
                        for ks, (H, W, C, k, s, p, OH, OW, F) in zip(self.ksplit, stack.geo)]
            self.feat = torch.empty(rows, stack.n_feat, device=dev)
            self.arg = torch.zeros(rows, stack.geo[-1][8], dtype=torch.int32, device=dev) if (keep and not stack.flatten) else None

    def __init__(self, params, conv_names, obs_shape, kernels, strides, filters, flatten=False, implicit=True):
        """flatten: the stack ends in nn.Flatten() of the NCHW activation (AC_CNN_Atari, cnn.py:83-96: filters * OH * OW features
        in (c, h, w) order) instead of the global max-pool of Basic_CNN."""
        self.params, self.names, self.flatten = params, list(conv_names), bool(flatten)
        H, W, C = obs_shape
        self.geo = []
        for k, s, F in zip(kernels, strides, filters):
            p = (k - s) // 2                                  # layers.py:46
            OH, OW = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
            self.geo.append((H, W, C, k, s, p, OH, OW, F))
            H, W, C = OH, OW, F
        self.n_feat = filters[-1] * (H * W if self.flatten else 1)
        self._ws = {}
        self.ws_gen = 0
        self._live, self.owner = {}, None
        self.implicit = implicit and self._implicit_eligible()
        if self.implicit:
            self._build_image_maps()

    # ---- implicit GEMMs on the matrix cores (csrc/conv_mfma.hip) ------------------------------------------------------------
    def _implicit_eligible(self):
        pow2 = lambda v: v >= 4 and (v & (v - 1)) == 0
        for i, (H, W, C, k, s, p, OH, OW, F) in enumerate(self.geo):
            if not (pow2(C) and F in (32, 64) and (k * k * C) % 32 == 0 and OW >= 2 and k <= 16 and s <= k):
                return False
            if C < 16 and not (i == 0 and C == 4 and k % 4 == 0):   # (narrow inputs: the 4 stacked uint8 frames of a pixel only)
                return False
            if i > 0:                                         # input-gradient products: N = C, reduction over (taps of a class, F)
                if C not in (32, 64) or not pow2(F) or F < 16 or W < 2 * s:
                    return False
                for rh in range(s):
                    for rw in range(s):
                        if (-(-(k - rh) // s) * -(-(k - rw) // s) * F) % 32:
                            return False
        return True

    @staticmethod
    def _frag_index(N, Kp):
        """(n, k') of every element of a fragment-ordered weight image [Kp/32][4][N/32][64 lanes][4] (include/xrl_hip.h:
        xrl_conv_t.w): sub-step s of group gq feeds lane-half h the indices 32 gq + 16 h + 4 s + (0..3)."""
        j = np.arange(N * Kp, dtype=np.int64)
        j4, lane, rest = j % 4, (j // 4) % 64, j // 256
        nb, q = rest % (N // 32), rest // (N // 32)
        return nb * 32 + lane % 32, 32 * (q // 4) + 16 * (lane // 32) + 4 * (q % 4) + j4

    def _build_image_maps(self):
        """Index maps image -> flat parameter for (a) the forward weights of every layer in (th, tw, c) order and (b) the
        input-gradient weights of layers 1.., one block per residue class (rh, rw) of (h + p, w + p) mod s with that class's
        kernel rows / columns reversed; both in the matrix cores' fragment order.  xrl_gather_images fills the images."""
        P = self.params
        maps, self._fwd_off, self._dx = [], [], []
        pos = 0
        for i, (H, W, C, k, s, p, OH, OW, F) in enumerate(self.geo):
            w_off = P.offsets[self.names[i] + ".weight"]
            n, kp = self._frag_index(F, k * k * C)
            tap, c = kp // C, kp % C
            maps.append(w_off + n * (C * k * k) + c * (k * k) + tap)
            self._fwd_off.append(pos)
            pos += F * k * k * C
        self._n_fwd = pos
        for i, (H, W, C, k, s, p, OH, OW, F) in enumerate(self.geo):
            classes = []
            if i > 0:
                w_off = P.offsets[self.names[i] + ".weight"]
                for rh in range(s):
                    for rw in range(s):
                        Th, Tw = -(-(k - rh) // s), -(-(k - rw) // s)
                        n, kp = self._frag_index(C, Th * Tw * F)
                        tap, f = kp // F, kp % F
                        kh, kw = rh + s * (Th - 1 - tap // Tw), rw + s * (Tw - 1 - tap % Tw)
                        maps.append(w_off + f * (C * k * k) + n * (k * k) + kh * k + kw)
                        ph, pw = (rh - p) % s, (rw - p) % s
                        classes.append(dict(off=pos, Th=Th, Tw=Tw, ph=ph, pw=pw, nh=-(-(H - ph) // s), nw=-(-(W - pw) // s),
                                            off_h=(ph + p - rh) // s - (Th - 1), off_w=(pw + p - rw) // s - (Tw - 1)))
                        pos += C * Th * Tw * F
            self._dx.append(classes)
        self._map = torch.from_numpy(np.concatenate(maps).astype(np.int32)).to(P.device)
        self._n_img = pos
        self._images = {}

    def images(self, flat=None, with_dx=True):
        """(image tensor, pack job) of the parameter set `flat` (None: the stack's own)."""
        flat = self.params.flat if flat is None else flat
        img = self._images.get(flat.data_ptr())
        if img is None:
            img = self._images[flat.data_ptr()] = torch.zeros(self._n_img, device=self.params.device)
        return img, (flat, self._map, img, self._n_img if with_dx else self._n_fwd)

    # -- images kept current by the optimiser launch (xrl_reduce_adam's mirrors) instead of being rebuilt by every pass ---------
    def inverse_maps(self):
        """(parameter -> forward-image position, parameter -> input-gradient-image position), int32 [flat.numel()], -1 = none:
        what xrl_mirrors_t.map wants.  A weight sits once in each of the two image sections."""
        if getattr(self, "_inv", None) is None:
            m = self._map.cpu().numpy().astype(np.int64)
            inv = []
            for lo, hi in ((0, self._n_fwd), (self._n_fwd, self._n_img)):
                a = np.full(self.params.flat.numel(), -1, np.int32)
                sec = m[lo:hi]
                ok = sec >= 0
                assert len(np.unique(sec[ok])) == int(ok.sum()), "a weight sits twice in one image section"
                a[sec[ok]] = (np.arange(lo, hi)[ok]).astype(np.int32)
                inv.append(torch.from_numpy(a).to(self.params.device))
            self._inv = tuple(inv)
        return self._inv

    def _version(self):
        return getattr(getattr(self, "owner", None), "version", 0)

    def is_live(self, flat=None):
        """The image of `flat` is current: the last writer of these parameters was an optimiser launch that mirrored its step
        into the image (mark_live), and nobody has bumped the owning network's `version` since."""
        flat = self.params.flat if flat is None else flat
        return self._live.get(flat.data_ptr(), None) == self._version()

    def mark_live(self, *flats):
        for f in flats:
            self._live[f.data_ptr()] = self._version()

    def invalidate(self):
        self._live.clear()

    def pack_images(self, jobs):
        """xrl_gather_images for the jobs whose image is not live (jobs: [(flat or None, with_dx)]) -> the image tensors."""
        out, todo = [], []
        for flat, with_dx in jobs:
            img, job = self.images(flat, with_dx)
            out.append(img)
            if not self.is_live(flat):
                todo.append(job)
        if todo:
            ops.gather_images(todo)
        return out

    @staticmethod
    def _k_split(rows):
        strips = (rows + 31) // 32
        return 1 if strips >= 2048 else (2 if strips >= 768 else (4 if strips >= 384 else 8))

    def _fwd_group(self, i, x, frames, img, flat, out):
        H, W, C, k, s, p, OH, OW, F = self.geo[i]
        return ops.conv_desc(img=x, w=img.data_ptr() + 4 * self._fwd_off[i], bias=self.params.ptr(self.names[i] + ".bias", flat),
                             out=out, B=frames, IH=H, IW=W, C=C, Th=k, Tw=k, nh=OH, nw=OW, sh=s, off_h=-p, off_w=-p, so=1,
                             ph=0, pw=0, OHt=OH, OWt=OW, N=F, act=ops.ACT["relu"],
                             img_u8=int(isinstance(x, torch.Tensor) and x.dtype == torch.uint8))

    def _forward_implicit(self, x, rows, ws, flat):
        if self.geo[0][2] < 16 and x.dtype != torch.uint8:
            raise ValueError("implicit-GEMM convolutions read 4-channel frames as uint8 (the replay ring's format); build the network "
                             "with implicit_conv=False for float32 frames")
        img, = self.pack_images([(flat, ws.keep)])
        ws.x_in = x
        for i, (H, W, C, k, s, p, OH, OW, F) in enumerate(self.geo):
            ops.conv_fwd([self._fwd_group(i, x, rows, img, flat, ws.y[i])], self._k_split(rows * OH * OW))
            x = ws.y[i]

    def _dw_splits(self, rows, n_max, n_cu=256):
        """Row chunks per layer of the weight-gradient launches.  The uint8 first layer and the float32 layers go out as two
        launches (kernel variants), one workgroup per (32 reduction columns, chunk), 8 waves each -- one workgroup per compute
        unit at these register counts: each launch gets as many chunks as keep its workgroups within ONE round (a 272-workgroup
        launch ran 16 of them after the other 256 and took twice as long), and no more than leave a wave ~30 rows."""
        tiles = [C * k * k // 32 for (H, W, C, k, s, p, OH, OW, F) in self.geo]
        u8 = [i == 0 and self.geo[0][2] == 4 for i in range(len(self.geo))]
        out = []
        for i, (H, W, C, k, s, p, OH, OW, F) in enumerate(self.geo):
            same = sum(t for t, v in zip(tiles, u8) if v == u8[i])
            n = max(1, min(n_max, n_cu // same, (rows * OH * OW) // (8 * 24)))
            out.append(n)
        return out

    def _backward_implicit(self, rows, ws, cs, stride, flat):
        """dy[-1] is set: input gradients down the stack (one launch per layer, a group per residue class), then every
        layer's weight / bias gradient in one grouped launch into the private slab set `cs`."""
        P = self.params
        img, _ = self.images(flat)
        wg, splits = [], self._dw_splits(rows, self.N_SPLIT_IMPLICIT)
        # (round 6) the weight gradients of layers >= 1 need nothing the last data-gradient launch produces: they go out on a side
        # branch as soon as their dy exists and run beside that launch and the first layer's weight gradient (DQN-C3: 16.8 us of the
        # update's chain of ten launches; config.overlap_conv_wgrad / ConvStack.overlap_wgrad: False keeps one grouped launch at the end)
        # MEASURED SLOWER (profiles/r06_g_conv_overlap.json: DQN-C3 update 120.3 -> 131.5 us, PPO-Atari update 7.65 -> 7.98 ms): a fork / join
        # between two hardware queues costs more than the 16.8 us it takes off the chain -- off unless ConvStack.overlap_wgrad = True
        side = ops.Branch() if (getattr(self, "overlap_wgrad", False) and len(self.geo) >= 2 and torch.cuda.is_available()) else None
        for i in reversed(range(len(self.geo))):
            H, W, C, k, s, p, OH, OW, F = self.geo[i]
            n = self.names[i]
            x = ws.y[i - 1] if i > 0 else ws.x_in
            wg.append(ops.conv_desc(img=x, dy=ws.dy[i], out=cs.data_ptr() + 4 * P.offsets[n + ".weight"],
                                    dbias=cs.data_ptr() + 4 * P.offsets[n + ".bias"], B=rows, IH=H, IW=W, C=C, Th=k, Tw=k,
                                    nh=OH, nw=OW, sh=s, off_h=-p, off_w=-p, so=1, OHt=OH, OWt=OW, N=F,
                                    img_u8=int(x.dtype == torch.uint8), pad=splits[i]))
            if side is not None and i == 1:
                side.begin()
                ops.conv_bwd_weight(wg, self.N_SPLIT_IMPLICIT, stride)
                side.end()
                wg = []
            if i > 0:
                groups = [ops.conv_desc(img=ws.dy[i], w=img.data_ptr() + 4 * c["off"], mask=ws.y[i - 1], out=ws.dy[i - 1], B=rows,
                                        IH=OH, IW=OW, C=F, Th=c["Th"], Tw=c["Tw"], nh=c["nh"], nw=c["nw"], sh=1, off_h=c["off_h"],
                                        off_w=c["off_w"], so=s, ph=c["ph"], pw=c["pw"], OHt=H, OWt=W, N=C, act=0, img_u8=0)
                          for c in self._dx[i]]
                ops.conv_fwd(groups, self._k_split(rows * sum(c["nh"] * c["nw"] for c in self._dx[i])))
        ops.conv_bwd_weight(wg, self.N_SPLIT_IMPLICIT, stride)
        if side is not None:
            side.join()


    @staticmethod
    def ksplit_for(M, N, K):
        """K ranges of a conv layer's forward GEMM: ~1.5 workgroups per CU, at least 128 reduction steps each."""
        tiles = ((M + 63) // 64) * ((N + 63) // 64)
        ks = 1
        while tiles * ks < 384 and K // (ks * 2) >= 128 and ks < 8:
            ks *= 2
        return ks

    def workspace(self, tag, rows, keep):
        ws = self._ws.get(tag)
        if ws is None or ws.rows < rows:
            if ws is not None:
                self.ws_gen += 1                       # a workspace was REPLACED: captured graphs that used it hold freed pointers
            ws = self._ws[tag] = ConvStack.Workspace(self, rows, keep)
        return ws

    def forward(self, x, rows, ws, flat=None, pool=True):
        """x: [rows, H*W*C] uint8 (or float32) NHWC frames -> ws.feat[:rows] (n_feat features per frame); pool=False: the caller's
        launch pools ws.y[-1] itself."""
        P = self.params
        for i, (H, W, C, k, s, p, OH, OW, F) in enumerate(self.geo):
            if self.implicit:
                self._forward_implicit(x, rows, ws, flat)
                break
            M, K = rows * OH * OW, C * k * k
            ops.im2col_nhwc(x, ws.col[i], rows, H, W, C, k, s, p)
            n = self.names[i]
            ks = self.ksplit_for(M, F, K)
            ks = ks if (ks > 1 and ws.skw[i] is not None and ws.skw[i].numel() >= ks * M * F) else 1
            ops.linear_fwd([ops.gemm_desc(ws.col[i].data_ptr(), P.ptr(n + ".weight", flat), ws.y[i].data_ptr(), M, F, K, K, K, F,
                                          bias=P.ptr(n + ".bias", flat), act="relu",
                                          aux=ws.skw[i].data_ptr() if ks > 1 else None, ldaux=ks if ks > 1 else 0)])
            x = ws.y[i]
        H, W, C, k, s, p, OH, OW, F = self.geo[-1]
        if not pool:
            return
        if self.flatten:
            ops.flatten_chw_fwd(ws.y[-1], ws.feat, rows, OH * OW, F, self.n_feat)
        else:
            ops.maxpool_hw_fwd(ws.y[-1], ws.feat, ws.arg, rows, OH * OW, F, F)
        return ws.feat

    def forward_dual(self, x, M, Re, ws, flat_t, pool=True):
        """Eval network on frames [0, Re) and target network (parameters `flat_t`) on frames [M, 2M) of x [2M, H*W*C] in
        ONE pass: one im2col per layer over all frames (the first layer's columns of the next_obs frames are shared by the
        target and, under double-Q, the eval network) and one grouped GEMM launch per layer (eval rows | target rows).
        ws rows: frames [0, Re) eval, [Re, Re+M) target; returns ws.feat.  Re = M (DQN) or 2M (double-Q)."""
        P = self.params
        tot = Re + M
        if self.implicit:
            img_e, img_t = self.pack_images([(None, True), (flat_t, False)])
            ws.x_in = x
            xe, xt = x, x[M:]
            for i, (H, W, C, k, s, p, OH, OW, F) in enumerate(self.geo):
                ops.conv_fwd([self._fwd_group(i, xe, Re, img_e, None, ws.y[i]),
                              self._fwd_group(i, xt, M, img_t, flat_t, ws.y[i][Re * OH * OW:])], self._k_split((Re + M) * OH * OW))
                xe, xt = ws.y[i], ws.y[i][Re * OH * OW:]
        for i, (H, W, C, k, s, p, OH, OW, F) in enumerate(self.geo):
            if self.implicit:
                break
            K, ohw = C * k * k, OH * OW
            n = self.names[i]
            if i == 0:
                ops.im2col_nhwc(x, ws.col[0], 2 * M, H, W, C, k, s, p)
                a_e, a_t = ws.col[0].data_ptr(), ws.col[0].data_ptr() + 4 * M * ohw * K
            else:
                ops.im2col_nhwc(ws.y[i - 1], ws.col[i], tot, H, W, C, k, s, p)
                a_e, a_t = ws.col[i].data_ptr(), ws.col[i].data_ptr() + 4 * Re * ohw * K
            y_e, y_t = ws.y[i].data_ptr(), ws.y[i].data_ptr() + 4 * Re * ohw * F
            ks = self.ksplit_for(tot * ohw, F, K)                 # both groups together fill the chip
            ks = ks if (ks > 1 and ws.skw[i] is not None and ws.skw[i].numel() >= ks * tot * ohw * F) else 1
            w_e = ws.skw[i].data_ptr() if ks > 1 else None
            w_t = ws.skw[i].data_ptr() + 4 * ks * Re * ohw * F if ks > 1 else None
            ops.linear_fwd([ops.gemm_desc(a_e, P.ptr(n + ".weight"), y_e, Re * ohw, F, K, K, K, F, bias=P.ptr(n + ".bias"), act="relu",
                                          aux=w_e, ldaux=ks if ks > 1 else 0),
                            ops.gemm_desc(a_t, P.ptr(n + ".weight", flat_t), y_t, M * ohw, F, K, K, K, F,
                                          bias=P.ptr(n + ".bias", flat_t), act="relu", aux=w_t, ldaux=ks if ks > 1 else 0)])
        H, W, C, k, s, p, OH, OW, F = self.geo[-1]
        if pool:                                              # (pool=False: the caller's launch pools -- xrl_dqn_tail_td)
            ops.maxpool_hw_fwd(ws.y[-1], ws.feat, ws.arg, tot, OH * OW, F, F)
        return ws.feat

    N_SPLIT = 64                                              # row chunks of a conv layer's weight gradient (parallelism)
    N_SPLIT_IMPLICIT = 32                                     # (implicit path: a workgroup's four waves split its chunk again)

    def backward(self, dfeat, rows, ws, slabs, n_split, flat=None, direct=False, pool=True):
        """dfeat [rows, n_feat] -> weight / bias gradients of every conv layer, summed into slabs[0] (the conv parameters
        are the first `p_conv` floats of the layout; their regions in slabs[1:] stay zero).  The GEMM rows of a conv
        layer are B*OH*OW (14 112 for the first Atari layer at batch 32), so the weight-gradient GEMM is split over
        N_SPLIT row chunks into a private slab set and reduced in a fixed order."""
        P = self.params
        p_conv = max(P.offsets[n + sfx] + int(np.prod(P.shapes[n + sfx])) for n in self.names for sfx in (".weight", ".bias"))
        if getattr(self, "_cslabs", None) is None:
            self._cslabs = torch.zeros(self.N_SPLIT, (p_conv + 3) // 4 * 4, device=P.device)
            self._csq = torch.zeros(256, dtype=torch.float64, device=P.device)
        cs, stride = self._cslabs, self._cslabs.shape[1]
        H, W, C, k, s, p, OH, OW, F = self.geo[-1]
        if not pool:
            pass                                              # (ws.dy[-1] already holds the last layer's gradient: xrl_dqn_tail_td)
        elif self.flatten:
            ops.flatten_chw_bwd(dfeat, ws.y[-1], ws.dy[-1], rows, OH * OW, F, dfeat.shape[1])
        else:
            ops.maxpool_hw_bwd(dfeat, ws.arg, ws.y[-1], ws.dy[-1], rows, OH * OW, F, dfeat.shape[1])
        if self.implicit:
            if direct and slabs.shape[0] >= self.N_SPLIT_IMPLICIT:
                # the caller's optimiser launch sums N_SPLIT_IMPLICIT slabs: the chunks go straight into the caller's slab set
                # (no private set, no extra reduction launch)
                self._backward_implicit(rows, ws, slabs, slabs.stride(0), flat)
                return self.N_SPLIT_IMPLICIT
            self._backward_implicit(rows, ws, cs, stride, flat)
            ops.grad_reduce(cs, self.N_SPLIT_IMPLICIT, stride, p_conv, slabs[0], self._csq)
            return n_split
        wg = []
        for i in reversed(range(len(self.geo))):
            H, W, C, k, s, p, OH, OW, F = self.geo[i]
            M, K = rows * OH * OW, C * k * k
            n = self.names[i]
            wg.append(ops.gemm_desc(ws.dy[i].data_ptr(), ws.col[i].data_ptr(), cs.data_ptr() + 4 * P.offsets[n + ".weight"],
                                    M, F, K, F, K, K, dbias=cs.data_ptr() + 4 * P.offsets[n + ".bias"]))
            if i > 0:
                ops.linear_bwd_data([ops.gemm_desc(ws.dy[i].data_ptr(), P.ptr(n + ".weight", flat), ws.dcol[i].data_ptr(),
                                                   M, K, F, F, K, K)])
                ops.col2im_nhwc(ws.dcol[i], ws.y[i - 1], ws.dy[i - 1], rows, H, W, C, k, s, p)   # times relu'(y_{i-1})
        # the weight gradients need nothing but dy / col of their layer: all layers in ONE grouped launch after the
        # data-gradient chain (the three launches took 13 + 13 + 33 us one after the other, each on a part of the chip)
        ops.linear_bwd_weight(wg, self.N_SPLIT, stride)
        ops.grad_reduce(cs, self.N_SPLIT, stride, p_conv, slabs[0], self._csq)
        return n_split


class ActorCriticCNN:
    """SharedActorCritic over the AC_CNN_Atari representation of configs/ppo/atari.yaml (rl_models/representations/cnn.py:53-102:
    x / 255, NHWC -> NCHW, Conv2d(k, s, pad=(k-s)//2) + ReLU x3, Flatten, Linear + ReLU per fc_hidden_sizes) with a
    CategoricalActorHead and a ValueHead (actor_hidden_size / critic_hidden_size, both [] in the yaml: the heads sit directly
    on the 512-wide embedding) -- the network DummyOnPolicyBuffer_Atari's uint8 frames train (memory_tools.py:290-328).

    Convolution stack = ConvStack (im2col + fp32-MFMA GEMMs, NHWC) ending in xrl_flatten_chw_fwd, so the dense layer reads its
    6 400 inputs in the reference's (c, h, w) order and `representation.model.7.weight` keeps the reference's layout; dense part
    = a Plan: fc layers, then actor / critic hidden layers of one depth side by side, then [logits | value]."""

    dist = "categorical"
    activation_action = None

    def __init__(self, obs_shape=(84, 84, 4), action_dim=4, kernels=(8, 4, 3), strides=(4, 2, 1), filters=(32, 64, 64),
                 fc_hidden=(512,), actor_hidden=(), critic_hidden=(), activation="relu", device="cuda", init=True,
                 implicit_conv=True):
        assert activation == "relu", "the convolution kernels apply ReLU (configs/ppo/atari.yaml: activation relu)"
        ah, ch = list(actor_hidden or []), list(critic_hidden or [])
        assert len(ah) == len(ch), "actor / critic hidden stacks must have equal depth"
        self.obs_shape, self.action_dim = tuple(obs_shape), int(action_dim)
        self.obs_dim = int(obs_shape[0] * obs_shape[1] * obs_shape[2])
        self.kernels, self.strides, self.filters = tuple(kernels), tuple(strides), tuple(filters)
        self.activation = activation
        specs, self.conv_names = [], []
        C = obs_shape[2]
        for i, (k, f) in enumerate(zip(kernels, filters)):
            n = f"representation.model.{2 * i}"
            specs += [(n + ".weight", (f, C, k, k)), (n + ".bias", (f,))]
            self.conv_names.append(n)
            C = f
        probe = ConvStack(None, self.conv_names, self.obs_shape, self.kernels, self.strides, self.filters, flatten=True, implicit=False)
        feat = probe.n_feat
        self.n_flat = feat
        rep_order = [n + sfx for n in self.conv_names for sfx in (".weight", ".bias")]
        widths, stages, lvl = [feat], [], 0
        base = 2 * len(self.conv_names) + 1                       # index of the first Linear behind nn.Flatten()
        self.fc_names = []
        for j, h in enumerate(fc_hidden):
            n = f"representation.model.{base + 2 * j}"
            specs += [(n + ".weight", (h, feat)), (n + ".bias", (h,))]
            rep_order += [n + ".weight", n + ".bias"]
            self.fc_names.append(n)
            stages.append([Layer(n, feat, h, activation, lvl, 0, lvl + 1, 0, n + ".weight", n + ".bias")])
            widths.append(h)
            feat, lvl = h, lvl + 1
        # Both branches read the whole embedding: their first layers (the heads themselves when the hidden stacks are empty, as
        # in the yaml) are ONE stacked GEMM [W_actor; W_critic] (adjacent weights, adjacent biases) -- the data-gradient GEMM
        # of a Plan writes d(input), so two layers must not read the same columns of a hidden level.
        a_order, c_order = [], []
        sizes_a, sizes_c = ah + [action_dim], ch + [1]
        fa = fc = feat
        for i, (ha, hc) in enumerate(zip(sizes_a, sizes_c)):
            na, nc = f"actor.logits.{2 * i}", f"critic.values.{2 * i}"
            last = i == len(sizes_a) - 1
            act_i = None if last else activation
            a_order += [na + ".weight", na + ".bias"]
            c_order += [nc + ".weight", nc + ".bias"]
            if i == 0:
                assert (ha * fa) % 4 == 0, "stacked actor / critic layer needs a 16-byte aligned critic weight"
                specs += [(na + ".weight", (ha, fa)), (nc + ".weight", (hc, fc)), (na + ".bias", (ha,)), (nc + ".bias", (hc,), "packed")]
                stages.append([Layer(na + "+" + nc, fa, ha + hc, act_i, lvl, 0, lvl + 1, 0, na + ".weight", na + ".bias")])
            else:
                specs += [(na + ".weight", (ha, fa)), (na + ".bias", (ha,)), (nc + ".weight", (hc, fc)), (nc + ".bias", (hc,))]
                stages.append([Layer(na, fa, ha, act_i, lvl, 0, lvl + 1, 0, na + ".weight", na + ".bias"),
                               Layer(nc, fc, hc, act_i, lvl, fa, lvl + 1, ha, nc + ".weight", nc + ".bias")])
            widths.append(ha + hc)
            fa, fc, lvl = ha, hc, lvl + 1
        self.ref_order = rep_order + a_order + c_order            # the reference's state_dict order: representation, actor, critic
        self.params = FlatParams(specs, device)
        self.plan = Plan(self.params, widths, stages)
        self.head_ld = action_dim + 1
        self.conv = ConvStack(self.params, self.conv_names, self.obs_shape, self.kernels, self.strides, self.filters, flatten=True,
                              implicit=implicit_conv)
        if init:
            self.reset_parameters()

    def reset_parameters(self):
        """cnn.py:78-81: orthogonal_(gain sqrt 2), bias 0 for every conv / fc layer of the representation; the heads take the
        agent's initialiser (orthogonal, gain 1; layers.py:23-26)."""
        for name in self.ref_order:
            v = self.params.view(name)
            if name.endswith(".weight"):
                gain = math.sqrt(2.0) if name.startswith("representation.") else 1.0
                v.copy_(_orthogonal(v.shape) * gain)
            else:
                v.zero_()

    state_dict = ActorCriticNet.state_dict
    load_state_dict = ActorCriticNet.load_state_dict
    parameters = ActorCriticNet.parameters

    # -- acting passes without the (c, h, w) flatten launch: a copy of the dense parameters whose first fc layer has its columns in the
    #    convolution output's own (h, w, c) order, so that the layer reads the last convolution's NHWC output as it lies (round 4: the
    #    flatten launch was 10.8 of the ~80 us of a PPO-Atari vector step).  Refreshed once per rollout (refresh_acting_params: two
    #    small copies + one column gather, capturable); products are summed in a different column order than on the training path.
    def acting_params(self):
        if getattr(self, "_act_flat", None) is None:
            Hc, Wc, C, k, s, p, OH, OW, F = self.conv.geo[-1]
            Pp = OH * OW
            assert Pp * F == self.n_flat
            new = torch.arange(Pp * F, dtype=torch.int64)
            self._act_perm = ((new % F) * Pp + new // F).to(self.params.device)      # new column p * F + f <- reference column f * P + p
            self._act_flat = self.params.flat.clone()
            n = self.fc_names[0] + ".weight"
            self._act_w = (self.params.offsets[n], self.params.view(n).numel())
        return self._act_flat

    def refresh_acting_params(self):
        af, (off, cnt) = self.acting_params(), self._act_w
        flat, n = self.params.flat, self.fc_names[0] + ".weight"
        if off > 0:
            af[:off].copy_(flat[:off])
        af[off + cnt:].copy_(flat[off + cnt:])
        W = self.params.view(n)
        torch.index_select(W, 1, self._act_perm, out=af[off:off + cnt].view(W.shape))

    def forward(self, x_u8, M, ldx=None, keep=True, acting=False):
        """x_u8 [rows >= M, H*W*C] uint8 (or float32 in 0..255) frames -> heads [cap, A + 1].  keep: this pass will be
        differentiated (its im2col columns and activations stay in the "grad" workspace).  acting (with keep=False): the dense part
        reads the last convolution's output in place through acting_params() -- the caller has refreshed them since the last
        parameter update (refresh_acting_params)."""
        ws = self.conv.workspace("grad" if keep else "act", M, keep)
        if keep:
            self._ws, self._M = ws, M
        if acting and not keep and self.conv.implicit and getattr(self, "_act_flat", None) is not None:
            self.conv.forward(x_u8[:M].reshape(M, -1), M, ws, pool=False)
            return self.plan.forward(ws.y[-1].view(-1)[:M * self.n_flat].view(M, self.n_flat), self.n_flat, M, flat=self._act_flat)
        feat = self.conv.forward(x_u8[:M].reshape(M, -1), M, ws)
        if keep:
            self._feat_in = feat
        return self.plan.forward(feat, self.n_flat, M)

    def act_tail_eligible(self, M):
        """May an acting pass end in xrl_ppo_act_tail?  The heads sit directly on ONE hidden layer (configs/ppo/atari.yaml: fc_hidden_sizes
        [512], actor / critic hidden []), that layer is a split-K product at M rows, the fast acting copy of the dense parameters exists."""
        st = self.plan.stages
        if len(st) != 2 or len(st[0]) != 1 or len(st[1]) != 1 or not self.conv.implicit or getattr(self, "_act_flat", None) is None:
            return False
        L, Lh = st[0][0], st[1][0]                       # (the heads: ONE stacked layer [W_actor; W_critic], adjacent biases)
        return M <= 64 and L.N % 64 == 0 and L.N <= 1024 and self.action_dim <= 16 and self.plan._split_k(L, M)[1] > 1 and \
            Lh.N == self.action_dim + 1 and Lh.K == L.N and Lh.act is None and Lh.in_off == 0

    def act_tail(self, x_u8, M, n, post=None, copy=None, **sample):
        """An acting pass on M rows of frames ending in ONE tail launch (xrl_ppo_act_tail): convolutions, the hidden layer's split-K
        product WITHOUT its epilogue, then epilogue + heads + sampling (+ the previous step's bookkeeping `post`, + the observation copy
        `copy` = (src, dst, bytes)).  Same numbers as forward(acting=True) + ops.policy_sample (+ ops.rollout_poststep)."""
        ws = self.conv.workspace("act", M, False)
        self.conv.forward(x_u8[:M].reshape(M, -1), M, ws, pool=False)
        plan, P, flat = self.plan, self.params, self._act_flat
        plan.ensure(M)
        L, Lh = plan.stages[0][0], plan.stages[1][0]
        A = self.action_dim
        x = ws.y[-1].view(-1)[:M * self.n_flat].view(M, self.n_flat)
        aux, ks = plan._split_k(L, M)
        c, ldc = plan._buf(plan.acts, L.out_level, L.out_off, x, self.n_flat)
        ops.linear_fwd_partials([ops.gemm_desc(x.data_ptr(), P.ptr(L.w_name, flat), c, M, L.N, L.K, self.n_flat, L.K, ldc,
                                               bias=P.ptr(L.b_name, flat), act=L.act, aux=aux.data_ptr(), ldaux=ks)])
        wa, ba = P.ptr(Lh.w_name, flat), P.ptr(Lh.b_name, flat)        # rows [0, A): the logits, row A: the value
        kw = dict(ws=aux, bias=P.ptr(L.b_name, flat), ks=ks, M=M, H=L.N, act=ops.ACT[L.act], w_actor=wa, b_actor=ba,
                  w_critic=wa + 4 * A * L.N, b_critic=ba + 4 * A, n=n, A=A)
        if copy is not None:
            kw.update(copy_src=copy[0], copy_dst=copy[1], copy_bytes=int(copy[2]))
        kw.update(sample)
        ops.ppo_act_tail(post=post, **kw)

    @property
    def d_heads(self):
        return self.plan.dacts[len(self.plan.widths) - 1]

    def backward(self, x_u8, M, slabs, n_split, ldx=None):
        if getattr(self, "_dfeat", None) is None or self._dfeat.shape[0] < M:
            self._dfeat = torch.zeros(M, self.n_flat, device=self.params.device)
        self.plan.backward_grouped(self._feat_in, self.n_flat, M, slabs, n_split, dx0=self._dfeat)
        self.conv.backward(self._dfeat, M, self._ws, slabs, n_split)


class DeepQCNN:
    """DeepQNetwork with the Basic_CNN representation of configs/dqn/atari.yaml (rl_models/representations/cnn.py:11-50:
    x/255, NHWC->NCHW, Conv2d(k, s, pad=(k-s)//2)+ReLU x3, AdaptiveMaxPool2d(1,1), Flatten) and a QValueHead MLP.

    Fully on the HIP engine: the convolution stack is ConvStack (im2col + fp32-MFMA GEMMs + max-pool kernels over OUR flat
    parameter buffer, csrc/conv.hip), the Q head, TD target, gradient slabs, clip, Adam and target sync as everywhere."""

    def __init__(self, obs_shape=(84, 84, 4), n_actions=4, kernels=(8, 4, 3), strides=(4, 2, 1), filters=(32, 64, 64),
                 q_hidden=(512,), activation="relu", device="cuda", init=True, implicit_conv=True, dueling=False):
        assert activation == "relu"
        self.dueling = bool(dueling)
        self.obs_shape, self.n_actions, self.obs_dim = tuple(obs_shape), n_actions, int(obs_shape[0] * obs_shape[1] * obs_shape[2])
        self.kernels, self.strides, self.filters = tuple(kernels), tuple(strides), tuple(filters)
        specs, order, stages, widths = [], [], [], [filters[-1]]
        C = obs_shape[2]
        self.conv_names = []
        for i, (k, f) in enumerate(zip(kernels, filters)):
            n = f"representation.model.{2 * i}"
            specs += [(n + ".weight", (f, C, k, k)), (n + ".bias", (f,))]
            order += [n + ".weight", n + ".bias"]
            self.conv_names.append(n)
            C = f
        _q_head_layers(filters[-1], q_hidden, n_actions, activation, self.dueling, 0, specs, order, stages, widths)
        self.params = FlatParams(specs, device)
        self.target_flat = self.params.like()
        self.plan = Plan(self.params, widths, stages)
        self.target_plan = Plan(self.params, widths, stages)
        rep = [k for k in order if k.startswith("representation.")]
        head = [k for k in order if k.startswith("eval_Q_head.")]
        self.ref_order = rep + ["target_" + k for k in rep] + head + ["target_Q_head." + k[len("eval_Q_head."):] for k in head]
        self.trainable_order = rep + head
        self.conv = ConvStack(self.params, self.conv_names, self.obs_shape, self.kernels, self.strides, self.filters, implicit=implicit_conv)
        self.conv.owner = self
        if init:
            for name in order:
                v = self.params.view(name)
                v.copy_(_orthogonal(v.shape)) if name.endswith(".weight") else v.zero_()
            self.copy_target()

    _target_key = DeepQNet._target_key
    state_dict = DeepQNet.state_dict

    def load_state_dict(self, sd):
        DeepQNet.load_state_dict(self, sd)
        self.touched()

    def copy_target(self):
        DeepQNet.copy_target(self)
        self.touched()

    def touched(self):
        """Somebody other than a mirroring optimiser launch wrote the parameters: the convolution stack's weight images are
        rebuilt by the next pass (ConvStack.is_live compares `version`)."""
        self.version = getattr(self, "version", 0) + 1

    def forward(self, x_u8, M, ldx=None):
        """Rows [0, M) are differentiated through (eval Q of obs); returns Q [rows, n_actions]."""
        rows = x_u8.shape[0]
        ws = self.conv.workspace("grad", M, True)
        self._ws = ws
        feat = self.conv.forward(x_u8[:M].reshape(M, -1), M, ws)
        if rows > M:                                       # double-Q / acting rows: eval net, no gradient
            ws2 = self.conv.workspace("nograd", rows - M, False)
            f2 = self.conv.forward(x_u8[M:].reshape(rows - M, -1), rows - M, ws2)
            if getattr(self, "_cat", None) is None or self._cat.shape[0] < rows:
                self._cat = torch.empty(rows, self.filters[-1], device=self.params.device)
            self._cat[:M].copy_(feat[:M]); self._cat[M:rows].copy_(f2[:rows - M])
            feat = self._cat
        self._feat_in = feat
        return self.plan.forward(self._feat_in, self.filters[-1], rows)

    def target(self, x_u8, M, ldx=None):
        ws = self.conv.workspace("target", M, False)
        self._tfeat = self.conv.forward(x_u8.reshape(M, -1), M, ws, flat=self.target_flat)
        return self.target_plan.forward(self._tfeat, self.filters[-1], M, flat=self.target_flat)

    fused_head = DeepQNet.fused_head
    head_td = DeepQNet.head_td

    def fused_tail(self):
        """(hidden layer, Q layer) as xrl_dqn_tail_td wants them, or None: Basic_CNN's global max-pool over at most 128 positions of
        64 filters (implicit-GEMM stack: the last layer's gradient is written in place), then exactly Linear(64, H) + activation
        and Linear(H, n_actions), H <= 512 -- configs/dqn/atari.yaml."""
        L2 = self.fused_head()
        st = self.plan.stages
        Hc, Wc, C, k, s, p, OH, OW, F = self.conv.geo[-1]
        if L2 is None or len(st) != 2 or len(st[0]) != 1 or not self.conv.implicit or self.conv.flatten or F != 64 or OH * OW > 128:
            return None
        L1 = st[0][0]
        if L1.in_level != 0 or L1.in_off != 0 or L1.K != 64 or L1.N > 512 or L1.N != self.plan.widths[1] or L2.K != L1.N:
            return None
        return L1, L2

    def act_egreedy(self, x_u8, n, eps_dev, action, action_f, seed, step, step_dev=None, eps=0.0, eps_sched=None):
        """Q(obs) of the eval network + OffPolicyAgent.exploration's choice for n frames (epsilon from eps_dev, or by value when
        eps_dev is None): the convolutions, then pool + hidden + Q
        layers + the epsilon-greedy action in ONE launch (xrl_dqn_act_tail) when fused_tail() covers the network.  Returns the Q
        tensor [n, n_actions] (rows of plan.acts[last])."""
        tail = self.fused_tail()
        if tail is None:
            q = self.forward(x_u8, n)
            ops.egreedy(q=q, eps_dev=eps_dev, eps=float(eps), action=action, action_f=action_f, n=n, A=self.n_actions, ld=q.stride(0),
                        seed=seed, step=step, step_dev=step_dev)
            return q
        L1, L2 = tail
        ws = self.conv.workspace("nograd", n, False)
        self.conv.forward(x_u8[:n].reshape(n, -1), n, ws, pool=False)
        Hc, Wc, C, k, s, p, OH, OW, F = self.conv.geo[-1]
        self.plan.ensure(n)
        q, prm = self.plan.acts[2], self.params
        ops.dqn_act_tail(y=ws.y[-1], w1=prm.ptr(L1.w_name), b1=prm.ptr(L1.b_name), w2=prm.ptr(L2.w_name), b2=prm.ptr(L2.b_name),
                         eps_dev=eps_dev, eps=float(eps), action=action, action_f=action_f, q=q, feat=ws.feat, step_dev=step_dev, seed=int(seed),
                         step=int(step) & 0xffffffff, n=n, A=L2.N, H=L1.N, F=F, P=OH * OW, ld_q=self.plan.widths[2], ld_f=ws.feat.shape[1],
                         act=ops.ACT[L1.act], **(dict(eps_sched=1, eps_n=int(eps_sched[0]), eps_kstar=int(eps_sched[1]),
                                                      eps_start=float(eps_sched[2]), eps_delta=float(eps_sched[3])) if eps_sched else {}))
        return q

    def tail_td(self, M, double_q, actions, rewards, terminals, diag, partials, gamma, slabs=None, huber_delta=0.0):
        """After forward_pair(..., skip_last="tail"): pool + hidden + Q layers + TD + the gradients back to the last convolution's
        output in one launch; backward(..., tail=True) continues with the weight gradients and the convolution stack."""
        (L1, L2), pl, tp, ws = self.fused_tail(), self.plan, self.target_plan, self._ws
        Hc, Wc, C, k, s, p, OH, OW, F = self.conv.geo[-1]
        Re, Pp = (2 * M if double_q else M), OH * OW
        prm, tf = self.params, self.target_flat
        pl.ensure(Re); tp.ensure(M)
        kw = {}
        if slabs is not None:          # each transition's term of the dense layers' gradients straight into its slab
            off = prm.offsets
            kw = dict(slabs=slabs, slab_stride=slabs.stride(0), off_w1=off[L1.w_name], off_b1=off[L1.b_name], off_w2=off[L2.w_name],
                      off_b2=off[L2.b_name])
        self._tail_slabs = slabs is not None
        ops.dqn_tail_td(**kw, y_eval=ws.y[-1], y_target=ws.y[-1][Re * Pp:], feat_eval=ws.feat, feat_target=ws.feat[Re:], arg=ws.arg,
                        w1_eval=prm.ptr(L1.w_name), b1_eval=prm.ptr(L1.b_name), w1_target=prm.ptr(L1.w_name, tf),
                        b1_target=prm.ptr(L1.b_name, tf), w2_eval=prm.ptr(L2.w_name), b2_eval=prm.ptr(L2.b_name),
                        w2_target=prm.ptr(L2.w_name, tf), b2_target=prm.ptr(L2.b_name, tf), actions=actions, rewards=rewards,
                        terminals=terminals, q_eval=pl.acts[2], q_target=tp.acts[2], d_q=pl.dacts[2], h_eval=pl.acts[1],
                        d_h=pl.dacts[1], d_feat=None, dy=ws.dy[-1], diag=diag, partials=partials, M=M, A=L2.N, H=L1.N, F=F, P=Pp,
                        ld_h=pl.widths[1], ld_q=pl.widths[2], ld_f=ws.feat.shape[1], double_q=int(double_q), act=ops.ACT[L1.act],
                        gamma=float(gamma), huber_delta=float(huber_delta))

    def forward_pair(self, X, M, double_q, skip_last=False):
        """One update's three network passes (dqn_learner.py:39-40, ddqn_learner.py:40): eval Q of obs = X[:M] (kept for
        backward), target Q of next_obs = X[M:2M] and, under double-Q, eval Q of next_obs -- as one im2col + one grouped
        GEMM launch per layer.  Returns (Q_eval [Re, A], Q_target [M, A])."""
        Re, F = (2 * M if double_q else M), self.filters[-1]
        ws = self.conv.workspace("dual", 3 * M, True)
        self._ws = ws
        feat = self.conv.forward_dual(X[:2 * M].reshape(2 * M, -1), M, Re, ws, self.target_flat, pool=skip_last != "tail")
        self._feat_in = feat
        if skip_last == "tail":                            # (xrl_dqn_tail_td does the rest of the forward pass)
            return None, None
        q_e, q_t = Plan.forward_many([(self.plan, feat, F, Re, None), (self.target_plan, feat[Re:], F, M, self.target_flat)],
                                     skip_last=skip_last)
        return q_e, q_t

    @property
    def d_out(self):
        return self.plan.dacts[len(self.plan.widths) - 1]

    def backward(self, x_u8, M, slabs, n_split, skip_last_dg=False, tail=False):
        if getattr(self, "_dfeat", None) is None or self._dfeat.shape[0] < M:
            self._dfeat = torch.zeros(M, self.filters[-1], device=self.params.device)
        in_slabs = tail and getattr(self, "_tail_slabs", False)   # (xrl_dqn_tail_td wrote the dense layers' terms, one slab per row)
        if not in_slabs:
            self.plan.backward_grouped(self._feat_in, self.filters[-1], M, slabs, n_split, dx0=self._dfeat, skip_last_dg=skip_last_dg,
                                       weights_only=tail)
        # The dense layers' terms sit in `written` slab rows (one per transition from xrl_dqn_tail_td, n_split from the layered
        # backward); the reduction sums more rows than that over every column, so the dense columns of the rows beyond must read as
        # zero.  They do once zeroed -- until the way of writing changes (another batch size, tail <-> layered): then the stale
        # terms of the old layout are cleared, AFTER this update's writers (zeroing the whole slab block here, as round 3 did,
        # wiped what xrl_dqn_tail_td / backward_grouped had just written: that update stepped the dense layers with zero gradients).
        split = -M if in_slabs else n_split
        if getattr(self, "_head_split", split) != split:
            written = M if in_slabs else n_split
            off = self.params.offsets
            names = [nm for st in self.plan.stages for L in st for nm in (L.w_name, L.b_name)]
            lo = min(off[nm] for nm in names)
            hi = max(off[nm] + (L.N * L.K if nm == L.w_name else L.N) for st in self.plan.stages for L in st for nm in (L.w_name, L.b_name))
            if written < slabs.shape[0]:
                slabs[written:, lo:hi].zero_()
        self._head_split = split
        return self.conv.backward(self._dfeat, M, self._ws, slabs, n_split, direct=True, pool=not tail)   # number of slabs to sum
