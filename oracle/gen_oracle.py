"""CPU oracle of the GENERAL towers - TEST INFRASTRUCTURE (see oracle/ppo_oracle.py for the rules).

Restates, with plain torch-CPU ops on one flat parameter vector in the reference's ``model.parameters()`` order, what
``MLPBase`` / ``MLPLayer`` compute for any ``hidden_size`` / ``layer_N`` / ``activation_id`` /
``use_feature_normalization`` (openrl/modules/networks/utils/mlp.py:8-46,100-180): optional ``feature_norm`` LayerNorm
over the observation, ``fc1`` = Linear-act-LayerNorm, the DEAD ``fc_h`` template (registered by ``MLPLayer.__init__``
whenever ``layer_N > 1`` but never called - its parameters exist and get no gradient), ``layer_N - 1`` deep-copied
``fc2`` clones, ``fc3`` = Linear-LayerNorm, then the head.  ``GenTowerSpec`` plugs into ``ppo_oracle``'s losses / update /
``train_ppo`` through ``ppo_oracle.tower_forward``'s dispatch on ``spec.general``.  Pinned by replaying the goldens
minted from the real reference (tests/test_oracle_cpu.py)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F

from . import ppo_oracle as po

_ACTS = [torch.tanh, F.relu, F.leaky_relu, F.elu]  # activation_id (mlp.py:14)


@dataclass
class GenTowerSpec(po.TowerSpec):
    layer_N: int = 1
    activation_id: int = 1
    feature_norm: bool = False
    general: bool = True

    def sizes(self):
        D, H, K = self.obs_dim, self.hidden, self.n_out
        s = []
        if self.feature_norm:
            s += [("fn_g", (D,)), ("fn_b", (D,))]
        seq = lambda name, n_in: [(name + "_W", (H, n_in)), (name + "_b", (H,)), (name + "_g", (H,)), (name + "_be", (H,))]
        s += seq("fc1", D)
        if self.layer_N > 1:
            s += seq("fc_h", H)  # registered, never run
            for i in range(self.layer_N - 1):
                s += seq("fc2_%d" % i, H)
        s += seq("fc3", H)
        s += [("W3", (K, H)), ("b3", (K,))]
        if self.head == po.HEAD_GAUSSIAN:
            s.append(("logstd", (K,)))
        return s

    def forward(self, theta: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
        p: Dict[str, torch.Tensor] = self.split(theta)
        H, act = self.hidden, _ACTS[self.activation_id]
        if self.feature_norm:
            x = F.layer_norm(x, (self.obs_dim,), p["fn_g"], p["fn_b"], 1e-5)
        seq = lambda name, x: F.layer_norm(act(F.linear(x, p[name + "_W"], p[name + "_b"])), (H,), p[name + "_g"],
                                           p[name + "_be"], 1e-5)
        x = seq("fc1", x)
        for i in range(self.layer_N - 1):
            x = seq("fc2_%d" % i, x)
        x = F.layer_norm(F.linear(x, p["fc3_W"], p["fc3_b"]), (H,), p["fc3_g"], p["fc3_be"], 1e-5)
        return F.linear(x, p["W3"], p["b3"])


def specs_from_cfg(cfg, obs_dim: int, n_act: int, head: int):
    """(policy spec, critic spec) of a golden case's configuration."""
    kw = dict(hidden=int(cfg.hidden_size), layer_N=int(cfg.layer_N), activation_id=int(cfg.activation_id),
              feature_norm=bool(cfg.use_feature_normalization))
    return GenTowerSpec(obs_dim, n_act, head, **kw), GenTowerSpec(obs_dim, 1, po.HEAD_VALUE, **kw)


@dataclass
class GenRnnTowerSpec(GenTowerSpec):
    """General trunk + ``RNNLayer`` (networks/utils/rnn.py:5-99: ``nn.GRU`` / ``nn.LSTM`` with ``recurrent_N`` layers, then
    LayerNorm) + head.  States per sequence: [recurrent_N, H] (GRU) or [recurrent_N, 2 H] = [h | c] (LSTM), flattened."""
    cell: str = "gru"
    recurrent_N: int = 1

    @property
    def G(self):
        return 3 if self.cell == "gru" else 4

    @property
    def state_w(self):
        return self.hidden * (1 if self.cell == "gru" else 2)

    def sizes(self):
        s = GenTowerSpec.sizes(self)
        H, G = self.hidden, self.G
        k = [n for n, _ in s].index("W3")
        rnn = []
        for l in range(self.recurrent_N):  # nn.GRU / nn.LSTM parameter order, layer by layer
            rnn += [("Wih%d" % l, (G * H, H)), ("Whh%d" % l, (G * H, H)), ("bih%d" % l, (G * H,)), ("bhh%d" % l, (G * H,))]
        rnn += [("rg", (H,)), ("rb", (H,))]
        return s[:k] + rnn + s[k:]

    def trunk(self, p, x):
        H, act = self.hidden, _ACTS[self.activation_id]
        if self.feature_norm:
            x = F.layer_norm(x, (self.obs_dim,), p["fn_g"], p["fn_b"], 1e-5)
        seq = lambda name, x: F.layer_norm(act(F.linear(x, p[name + "_W"], p[name + "_b"])), (H,), p[name + "_g"],
                                           p[name + "_be"], 1e-5)
        x = seq("fc1", x)
        for i in range(self.layer_N - 1):
            x = seq("fc2_%d" % i, x)
        return F.layer_norm(F.linear(x, p["fc3_W"], p["fc3_b"]), (H,), p["fc3_g"], p["fc3_be"], 1e-5)

    def rnn_forward(self, theta, x, h0, masks):
        """``x`` [L*N, D] (row = l*N + n), ``h0`` [N, recurrent_N * state_w], ``masks`` [L*N, 1] ->
        (head output [L*N, K], final states [N, recurrent_N * state_w])."""
        p = self.split(theta)
        H, rN, SW = self.hidden, self.recurrent_N, self.state_w
        N = h0.shape[0]
        L = x.shape[0] // N
        feats = self.trunk(p, x).view(L, N, H)
        m = masks.view(L, N, 1)
        st = [h0.view(N, rN, SW)[:, l] for l in range(rN)]
        outs = []
        for t in range(L):
            inp = feats[t]
            for l in range(rN):
                s_in = st[l] * m[t]  # (hxs * masks) on every layer, h and c alike (rnn.py:43-47)
                h_in = s_in[:, :H]
                gi = F.linear(inp, p["Wih%d" % l], p["bih%d" % l])
                gh = F.linear(h_in, p["Whh%d" % l], p["bhh%d" % l])
                if self.cell == "gru":
                    r = torch.sigmoid(gi[:, :H] + gh[:, :H])
                    z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
                    n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
                    h = (1.0 - z) * n + z * h_in
                    st[l] = h
                else:
                    a = gi + gh
                    i_, f_, g_, o_ = (torch.sigmoid(a[:, :H]), torch.sigmoid(a[:, H:2 * H]), torch.tanh(a[:, 2 * H:3 * H]),
                                      torch.sigmoid(a[:, 3 * H:]))
                    c = f_ * s_in[:, H:] + i_ * g_
                    h = o_ * torch.tanh(c)
                    st[l] = torch.cat([h, c], -1)
                inp = h
            outs.append(inp)
        y = F.layer_norm(torch.cat(outs, 0), (H,), p["rg"], p["rb"], 1e-5)
        return F.linear(y, p["W3"], p["b3"]), torch.stack(st, 1).reshape(N, rN * SW)


def rnn_specs_from_cfg(cfg, Dp: int, Dc: int, n_act: int, head: int):
    kw = dict(hidden=int(cfg.hidden_size), layer_N=int(cfg.layer_N), activation_id=int(cfg.activation_id),
              feature_norm=bool(cfg.use_feature_normalization), cell=str(cfg.rnn_type), recurrent_N=int(cfg.recurrent_N))
    return GenRnnTowerSpec(Dp, n_act, head, **kw), GenRnnTowerSpec(Dc, 1, po.HEAD_VALUE, **kw)


# =====================================================================================================
# PolicyValueNetwork (use_share_model; policy_value_network.py:34-172): obs_prep (MLPBase) -> common
# (MLPLayer(H, H, layer_N = 0) = fc1 + fc3) -> [RNNLayer, always a GRU] -> v_out / act heads on ONE parameter vector
# =====================================================================================================
@dataclass
class SharedSpec(GenRnnTowerSpec):
    """One instance per ROLE ("actor": the action head's output, "critic": v_out's) over the same flat vector; ``n_out`` /
    ``head`` describe the role's output, ``n_act`` / ``act_head`` the action head (both heads are always in the vector)."""
    role: str = "actor"
    n_act: int = 1
    act_head: int = po.HEAD_CATEGORICAL
    recurrent: bool = False

    def sizes(self):
        D, H = self.obs_dim, self.hidden
        s = []
        if self.feature_norm:
            s += [("fn_g", (D,)), ("fn_b", (D,))]
        seq = lambda name, n_in: [(name + "_W", (H, n_in)), (name + "_b", (H,)), (name + "_g", (H,)), (name + "_be", (H,))]
        s += seq("fc1", D)
        if self.layer_N > 1:
            s += seq("fc_h", H)
            for i in range(self.layer_N - 1):
                s += seq("fc2_%d" % i, H)
        s += seq("fc3", H) + seq("c1", H) + seq("c3", H)  # obs_prep.mlp, then common.fc1 / common.fc3
        if self.recurrent:
            for l in range(self.recurrent_N):
                s += [("Wih%d" % l, (3 * H, H)), ("Whh%d" % l, (3 * H, H)), ("bih%d" % l, (3 * H,)), ("bhh%d" % l, (3 * H,))]
            s += [("rg", (H,)), ("rb", (H,))]
        s += [("Wv", (1, H)), ("bv", (1,)), ("Wa", (self.n_act, H)), ("ba", (self.n_act,))]
        if self.act_head == po.HEAD_GAUSSIAN:
            s.append(("logstd", (self.n_act,)))
        return s

    def trunk(self, p, x):
        H, act = self.hidden, _ACTS[self.activation_id]
        x = GenRnnTowerSpec.trunk(self, p, x)
        x = F.layer_norm(act(F.linear(x, p["c1_W"], p["c1_b"])), (H,), p["c1_g"], p["c1_be"], 1e-5)
        return F.layer_norm(F.linear(x, p["c3_W"], p["c3_b"]), (H,), p["c3_g"], p["c3_be"], 1e-5)

    def _head(self, p, y):
        return F.linear(y, p["Wa"], p["ba"]) if self.role == "actor" else F.linear(y, p["Wv"], p["bv"])

    def forward(self, theta, x):
        p = self.split(theta)
        return self._head(p, self.trunk(p, x))

    def rnn_forward(self, theta, x, h0, masks):
        p = self.split(theta)
        p["W3"], p["b3"] = (p["Wa"], p["ba"]) if self.role == "actor" else (p["Wv"], p["bv"])
        self_split, self.split = self.split, (lambda th: p)  # reuse the stack code of GenRnnTowerSpec on this layout
        try:
            return GenRnnTowerSpec.rnn_forward(self, theta, x, h0, masks)
        finally:
            self.split = self_split


def shared_specs_from_cfg(cfg, obs_dim: int, n_act: int, act_head: int):
    kw = dict(hidden=int(cfg.hidden_size), layer_N=int(cfg.layer_N), activation_id=int(cfg.activation_id),
              feature_norm=bool(cfg.use_feature_normalization), cell="gru", recurrent_N=int(cfg.recurrent_N), n_act=n_act,
              act_head=act_head, recurrent=bool(cfg.use_recurrent_policy or cfg.use_naive_recurrent_policy))
    return (SharedSpec(obs_dim, n_act, act_head, role="actor", **kw),
            SharedSpec(obs_dim, 1, po.HEAD_VALUE, role="critic", **kw))


def shared_ppo_update(hp, aspec, cspec, theta, adam, vn, sample, recurrent: bool):
    """ppo_update with ``use_share_model`` (ppo.py:46-176): both losses back-propagate into the ONE parameter vector;
    ``get_actor_para()`` and ``get_critic_para()`` both return all of it, so ``clip_grad_norm_`` runs twice -
    actor_grad_norm is the raw norm, critic_grad_norm the once-clipped one - then one Adam step."""
    from . import rnn_oracle as ro

    t = lambda a: None if a is None else torch.as_tensor(a, dtype=torch.float32)
    th = theta.detach().clone().requires_grad_(True)
    if recurrent:
        loss_list, value_loss, policy_loss, dist_entropy, ratio = ro.prepare_loss(hp, aspec, th, cspec, th, vn,
                                                                                  {k: t(v) for k, v in sample.items()})
    else:
        loss_list, value_loss, policy_loss, dist_entropy, ratio = po.prepare_loss(hp, aspec, th, cspec, th, vn,
                                                                                  tuple(t(a) for a in sample))
    for loss in loss_list:
        loss.backward()
    g = th.grad
    if hp.use_max_grad_norm:
        g, an = po.clip_grad_norm(g, hp.max_grad_norm)
        g, cn = po.clip_grad_norm(g, hp.max_grad_norm)
    else:
        an = cn = float(g.norm(2))
    adam.step(theta, g)
    return dict(value_loss=value_loss.item(), policy_loss=policy_loss.item(), dist_entropy=dist_entropy.item(),
                actor_grad_norm=an, critic_grad_norm=cn, ratio=ratio.mean().item())


def train_shared(hp, aspec, cspec, theta, adam, vn, buf, ppo_epoch: int, num_mini_batch: int, data_chunk_length: int = 0):
    """PPOAlgorithm.train_ppo (ppo.py:383-458) on the shared network; ``data_chunk_length > 0`` = recurrent_generator."""
    from . import rnn_oracle as ro

    adv = po.advantages(buf["returns"], buf["value_preds"], buf["active_masks"], vn if hp.use_valuenorm else None,
                        hp.use_adv_normalize)
    keys = ("value_loss", "policy_loss", "dist_entropy", "actor_grad_norm", "critic_grad_norm", "ratio")
    info = {k: 0.0 for k in keys}
    vnn = vn if hp.use_valuenorm else None
    if data_chunk_length:
        rows = ro.buffer_rows(buf, adv)
        M = rows["adv"].shape[0]
    else:
        rows = {"critic_obs": po.flat_rows(buf["critic_obs"][:-1]), "policy_obs": po.flat_rows(buf["policy_obs"][:-1]),
                "actions": po.flat_rows(buf["actions"]), "value_preds": po.flat_rows(buf["value_preds"][:-1]),
                "returns": po.flat_rows(buf["returns"][:-1]), "active_masks": po.flat_rows(buf["active_masks"][:-1]),
                "action_log_probs": po.flat_rows(buf["action_log_probs"]), "adv": adv.reshape(-1, 1),
                "action_masks": po.flat_rows(buf["action_masks"][:-1]) if buf.get("action_masks") is not None else None}
        M = rows["adv"].shape[0]
    for _ in range(ppo_epoch):
        if data_chunk_length:
            for chunks in ro.recurrent_chunk_order(M, data_chunk_length, num_mini_batch):
                step = shared_ppo_update(hp, aspec, cspec, theta, adam, vnn, ro.chunk_sample(rows, chunks, data_chunk_length), True)
                for k in keys:
                    info[k] += step[k]
        else:
            for idx in po.feed_forward_indices(M, num_mini_batch):
                g = lambda k: None if rows[k] is None else rows[k][idx]
                sample = (g("critic_obs"), g("policy_obs"), g("actions"), g("value_preds"), g("returns"), g("active_masks"),
                          g("action_log_probs"), g("adv"), g("action_masks"))
                step = shared_ppo_update(hp, aspec, cspec, theta, adam, vnn, sample, False)
                for k in keys:
                    info[k] += step[k]
    n = ppo_epoch * num_mini_batch
    return {k: v / n for k, v in info.items()}


def multidiscrete_evaluate(x, weights, biases, actions, active_masks=None):
    """ACTLayer.evaluate_actions for a MultiDiscrete space (act.py:136-151): one Categorical per component over the same
    features; log-probs concatenated per component; the entropy is the active-mask weighted (or plain) mean of each
    component's entropy, then the mean over components of DETACHED floats (``torch.tensor(dist_entropy).mean()``: no
    gradient flows into the entropy term for this head)."""
    logps, ents = [], []
    for W, b, a in zip(weights, biases, actions.transpose(0, 1)):
        dist = torch.distributions.Categorical(logits=F.linear(x, W, b))
        logps.append(dist.log_prob(a.long()).unsqueeze(-1))
        if active_masks is not None:
            ents.append((dist.entropy() * active_masks.squeeze(-1)).sum() / active_masks.sum())
        else:
            ents.append(dist.entropy().mean())
    return torch.cat(logps, -1), torch.tensor([float(e) for e in ents]).mean()


def multidiscrete_mode(x, weights, biases):
    """ACTLayer.forward(deterministic=True) for MultiDiscrete (act.py:60-72): per-component argmax + its log-prob."""
    acts, lps = [], []
    for W, b in zip(weights, biases):
        dist = torch.distributions.Categorical(logits=F.linear(x, W, b))
        a = dist.probs.argmax(dim=-1)
        acts.append(a.float().unsqueeze(-1))
        lps.append(dist.log_prob(a).unsqueeze(-1))
    return torch.cat(acts, -1), torch.cat(lps, -1)


def mixed_evaluate(x, Wm, bm, logstd, Wc, bc, actions, active_masks=None):
    """ACTLayer.evaluate_actions for Tuple(Box(cd), Discrete(n)) - the mixed branch (act.py:126-147): a DiagGaussian with
    per-dimension log-probs (distributions.py:30-33, 85-98) and a Categorical on the same features; the log-probs are
    concatenated and SUMMED to one joint column; the entropy is 0.0025 * Gaussian + 0.01 * Categorical, each an
    active-mask weighted mean over rows (the Gaussian one summed over its dimensions) or a plain mean (the Gaussian one
    over rows and dimensions)."""
    cd = Wm.shape[0]
    a, b = actions[:, :cd], actions[:, cd:].long()
    normal = torch.distributions.Normal(F.linear(x, Wm, bm), logstd.reshape(1, -1).exp().expand(x.shape[0], cd))
    cat = torch.distributions.Categorical(logits=F.linear(x, Wc, bc))
    lps = torch.cat([normal.log_prob(a), cat.log_prob(b.squeeze(-1)).unsqueeze(-1)], -1)
    if active_masks is not None:
        e0 = (normal.entropy() * active_masks).sum() / active_masks.sum()
        e1 = (cat.entropy() * active_masks.squeeze(-1)).sum() / active_masks.sum()
    else:
        e0, e1 = normal.entropy().mean(), cat.entropy().mean()
    return lps.sum(-1, keepdim=True), e0 * 0.0025 + e1 * 0.01


def mixed_mode(x, Wm, bm, logstd, Wc, bc):
    """ACTLayer.forward(deterministic=True), mixed branch (act.py:46-63): the Gaussian mean + the Categorical argmax, and
    the joint log-prob."""
    mean = F.linear(x, Wm, bm)
    normal = torch.distributions.Normal(mean, logstd.reshape(1, -1).exp().expand_as(mean))
    cat = torch.distributions.Categorical(logits=F.linear(x, Wc, bc))
    a = cat.probs.argmax(dim=-1, keepdim=True)
    lp = torch.cat([normal.log_prob(mean), cat.log_prob(a.squeeze(-1)).unsqueeze(-1)], -1).sum(-1, keepdim=True)
    return torch.cat([mean, a.float()], -1), lp
