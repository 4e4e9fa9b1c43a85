"""CPU oracle for the RECURRENT (GRU) branch of the hot path - TEST INFRASTRUCTURE (see ppo_oracle.py's header).

Restates, with plain torch-CPU ops + autograd:
  * RNNLayer (openrl/modules/networks/utils/rnn.py:5-99): ``h' = GRU(x, h * mask)`` then LayerNorm, one layer
    (recurrent_N = 1); the has_zeros segmentation of rnn.py:53-90 is an optimisation of per-step masking.
  * PolicyNetwork / ValueNetwork with ``use_recurrent_policy`` (policy_network.py:130-203,
    value_network.py:113-136): MLPBase -> RNNLayer -> ACTLayer / v_out.
  * ReplayData.recurrent_generator (buffers/replay_data.py:1062-1258): chunks of ``data_chunk_length`` rows of the
    [N, A, T]-ordered flat batch (chunks may straddle lanes when T % L != 0), initial state = the stored rnn
    state of the chunk's first row, ``torch.randperm(data_chunks)`` order, ``data_chunks // nmb`` per minibatch.
  * PPOAlgorithm.ppo_update / train_ppo on those samples (algorithms/ppo.py:46-176, 383-458).
Pinned against the real reference classes by ``oracle/gen_golden.py`` (tests/golden/train_recurrent*.npz).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F

from . import ppo_oracle as po

HEAD_VALUE, HEAD_CATEGORICAL, HEAD_GAUSSIAN = po.HEAD_VALUE, po.HEAD_CATEGORICAL, po.HEAD_GAUSSIAN


@dataclass
class RnnTowerSpec(po.TowerSpec):
    """Flat parameter order = ``model.parameters()`` of the reference: base.mlp.fc1 (W,b), LN, fc3 (W,b), LN,
    rnn.rnn (weight_ih_l0, weight_hh_l0, bias_ih_l0, bias_hh_l0), rnn.norm, head."""

    def sizes(self):
        D, H, K = self.obs_dim, self.hidden, self.n_out
        s = [("W1", (H, D)), ("b1", (H,)), ("g1", (H,)), ("be1", (H,)), ("W2", (H, H)), ("b2", (H,)), ("g2", (H,)),
             ("be2", (H,)), ("Wih", (3 * H, H)), ("Whh", (3 * H, H)), ("bih", (3 * H,)), ("bhh", (3 * H,)),
             ("g3", (H,)), ("be3", (H,)), ("W3", (K, H)), ("b3", (K,))]
        if self.head == HEAD_GAUSSIAN:
            s.append(("logstd", (K,)))
        return s


def init_rnn_tower(spec: RnnTowerSpec, gain_head: float, use_orthogonal: bool = True, activation_id: int = 1):
    """Reference RNG consumption order: fc1, fc3 (mlp.py:19-39), nn.GRU default init then orthogonal_ on
    weight_ih_l0, weight_hh_l0 with biases 0 (rnn.py:14-26), head linear."""
    init_method = torch.nn.init.orthogonal_ if use_orthogonal else torch.nn.init.xavier_uniform_
    gain = torch.nn.init.calculate_gain(["tanh", "relu", "leaky_relu", "selu"][activation_id])
    D, H, K = spec.obs_dim, spec.hidden, spec.n_out
    fc1 = torch.nn.Linear(D, H)
    init_method(fc1.weight.data, gain=gain)
    fc3 = torch.nn.Linear(H, H)
    init_method(fc3.weight.data, gain=gain)
    gru = torch.nn.GRU(H, H, num_layers=1)
    init_method(gru.weight_ih_l0.data)
    init_method(gru.weight_hh_l0.data)
    head = torch.nn.Linear(H, K)
    init_method(head.weight.data, gain=gain_head)
    z, o = torch.zeros, torch.ones
    parts = [fc1.weight.data.reshape(-1), z(H), o(H), z(H), fc3.weight.data.reshape(-1), z(H), o(H), z(H),
             gru.weight_ih_l0.data.reshape(-1), gru.weight_hh_l0.data.reshape(-1), z(3 * H), z(3 * H), o(H), z(H),
             head.weight.data.reshape(-1), z(K)]
    if spec.head == HEAD_GAUSSIAN:
        parts.append(z(K))
    return torch.cat(parts).clone()


def gru_cell(p: Dict[str, torch.Tensor], x: torch.Tensor, h: torch.Tensor) -> torch.Tensor:
    """torch.nn.GRU single layer, one step (gate order r, z, n; b_hn inside the r-product)."""
    H = h.shape[-1]
    gi = F.linear(x, p["Wih"], p["bih"])
    gh = F.linear(h, p["Whh"], p["bhh"])
    r = torch.sigmoid(gi[:, :H] + gh[:, :H])
    z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
    n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
    return (1.0 - z) * n + z * h


def rnn_tower_forward(spec: RnnTowerSpec, theta: torch.Tensor, x: torch.Tensor, h0: torch.Tensor,
                      masks: torch.Tensor):
    """``x`` [L*N, D] (row = l*N + n, rnn.py:50-57), ``h0`` [N, H], ``masks`` [L*N, 1].
    Returns (head output [L*N, K], final state [N, H])."""
    if getattr(spec, "general", False):  # general trunks, GRU / LSTM stacks: oracle/gen_oracle.py
        return spec.rnn_forward(theta, x, h0, masks)
    p = spec.split(theta)
    H = spec.hidden
    N = h0.shape[0]
    L = x.shape[0] // N
    feats = po.trunk_forward(p, x).view(L, N, H)
    m = masks.view(L, N, 1)
    h, outs = h0, []
    for l in range(L):
        h = gru_cell(p, feats[l], h * m[l])
        outs.append(h)
    y = F.layer_norm(torch.cat(outs, 0), (H,), p["g3"], p["be3"], 1e-5)
    return F.linear(y, p["W3"], p["b3"]), h


# ------------------------------------------------------------------------------------------ rollout
def get_actions(pspec, ptheta, cspec, ctheta, policy_obs, critic_obs, rnn_states, rnn_states_critic, masks,
                action_masks=None, deterministic=False, forced_u=None):
    """PPOModule.get_actions (ppo_module.py:102-138) with the engine's sampler (see ppo_oracle.get_actions).
    Returns values, actions, logp, new policy state [B,H], new critic state [B,H]."""
    t = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float32)
    with torch.no_grad():
        B = np.asarray(policy_obs).shape[0]
        out, hp_new = rnn_tower_forward(pspec, ptheta, t(policy_obs), t(rnn_states).view(B, -1), t(masks).view(B, 1))
        if pspec.head == HEAD_CATEGORICAL:
            lg = po.masked_logits(out, None if action_masks is None else t(action_masks))
            dist = torch.distributions.Categorical(logits=lg)
            if deterministic:
                a = dist.probs.argmax(dim=-1)
            else:
                a = torch.as_tensor(po.inverse_cdf_sample(lg.numpy(), np.asarray(forced_u, np.float32).reshape(-1)))
            logp = dist.log_prob(a).unsqueeze(-1)
            actions = a.unsqueeze(-1).float()
        else:
            std = pspec.split(ptheta)["logstd"].exp()
            dist = torch.distributions.Normal(out, std.expand_as(out))
            actions = out if deterministic else out + std * t(forced_u)
            logp = dist.log_prob(actions)
        values, hc_new = rnn_tower_forward(cspec, ctheta, t(critic_obs), t(rnn_states_critic).view(B, -1),
                                           t(masks).view(B, 1))
    return values.numpy(), actions.numpy(), logp.numpy(), hp_new.numpy(), hc_new.numpy()


# ------------------------------------------------------------------------------------------ generator
def cast_rows(x: np.ndarray) -> np.ndarray:
    """``_cast`` (buffers/utils/util.py:96-97): [T, N, A, w] -> [N*A*T, w], row = (n*A + a)*T + t."""
    return np.ascontiguousarray(x.transpose(1, 2, 0, 3)).reshape(-1, x.shape[-1])


def recurrent_chunk_order(batch_size: int, data_chunk_length: int, num_mini_batch: int) -> List[np.ndarray]:
    """replay_data.py:1065-1082: ``data_chunks = M // L``, ``randperm(data_chunks)`` split in nmb slices."""
    data_chunks = batch_size // data_chunk_length
    mbs = data_chunks // num_mini_batch
    rand = torch.randperm(data_chunks).numpy()
    return [rand[i * mbs:(i + 1) * mbs] for i in range(num_mini_batch)]


def chunk_sample(rows: Dict[str, Optional[np.ndarray]], chunks: np.ndarray, L: int):
    """One minibatch of replay_data.py:1147-1258: every per-row array becomes [L*Nc, w] with row = l*Nc + i
    (chunk i = rows chunks[i]*L .. +L-1 of the cast order); rnn states are taken at each chunk's first row."""
    first = np.asarray(chunks, dtype=np.int64) * L
    ridx = (first[None, :] + np.arange(L, dtype=np.int64)[:, None]).reshape(-1)
    out = {}
    for k, v in rows.items():
        if v is None:
            out[k] = None
        elif k in ("rnn_states", "rnn_states_critic"):
            out[k] = v[first]
        else:
            out[k] = v[ridx]
    return out


# ------------------------------------------------------------------------------------------ update
def prepare_loss(hp: po.PPOHyper, pspec, ptheta, cspec, ctheta, vn, s: Dict[str, torch.Tensor]):
    """PPOAlgorithm.prepare_loss (ppo.py:238-361) on a recurrent sample (non-joint, no policy v-head)."""
    values, _ = rnn_tower_forward(cspec, ctheta, s["critic_obs"], s["rnn_states_critic"], s["masks"])
    out, _ = rnn_tower_forward(pspec, ptheta, s["policy_obs"], s["rnn_states"], s["masks"])
    active, adv, old_logp, action = s["active_masks"], s["adv"], s["action_log_probs"], s["actions"]
    am = active if hp.use_policy_active_masks else None
    if pspec.head == HEAD_CATEGORICAL:
        dist = torch.distributions.Categorical(logits=po.masked_logits(out, s["action_masks"]))
        logp = dist.log_prob(action.squeeze(-1).long()).view(action.size(0), -1).sum(-1).unsqueeze(-1)
        ent = dist.entropy()
        dist_entropy = (ent * am.squeeze(-1)).sum() / am.sum() if am is not None else ent.mean()
    else:
        std = pspec.split(ptheta)["logstd"].exp()
        dist = torch.distributions.Normal(out, std.expand_as(out))
        logp = dist.log_prob(action)
        ent = dist.entropy()
        dist_entropy = (ent * am).sum() / am.sum() if am is not None else ent.mean()
    ratio = torch.exp(logp - old_logp)
    if hp.dual_clip_ppo:
        ratio = torch.min(ratio, torch.tensor(hp.dual_clip_coeff))
    surr_final = torch.min(ratio * adv, torch.clamp(ratio, 1.0 - hp.clip_param, 1.0 + hp.clip_param) * adv)
    if getattr(hp, "a2c", False):
        surr_final, ratio = adv * logp, torch.zeros(1)
    if hp.use_policy_active_masks:
        policy_loss = (-torch.sum(surr_final, dim=-1, keepdim=True) * active).sum() / active.sum()
    else:
        policy_loss = -torch.sum(surr_final, dim=-1, keepdim=True).mean()
    value_loss = po.cal_value_loss(hp, vn, values, s["value_preds"], s["returns"], active)
    return [policy_loss - dist_entropy * hp.entropy_coef, value_loss * hp.value_loss_coef], value_loss, policy_loss, \
        dist_entropy, ratio


def ppo_update(hp, pspec, ptheta, cspec, ctheta, padam, cadam, vn, sample_np: Dict[str, Optional[np.ndarray]]):
    t = lambda a: None if a is None else torch.as_tensor(a, dtype=torch.float32)
    s = {k: t(v) for k, v in sample_np.items()}
    pth = ptheta.detach().clone().requires_grad_(True)
    cth = ctheta.detach().clone().requires_grad_(True)
    loss_list, value_loss, policy_loss, dist_entropy, ratio = prepare_loss(hp, pspec, pth, cspec, cth, vn, s)
    for loss in loss_list:
        loss.backward()
    if hp.use_max_grad_norm:
        gp, an = po.clip_grad_norm(pth.grad, hp.max_grad_norm)
        gc, cn = po.clip_grad_norm(cth.grad, hp.max_grad_norm)
    else:
        gp, an = pth.grad, float(pth.grad.norm(2))
        gc, cn = cth.grad, float(cth.grad.norm(2))
    raw_p, raw_c = pth.grad.detach().numpy().copy(), cth.grad.detach().numpy().copy()
    padam.step(ptheta, gp)
    cadam.step(ctheta, gc)
    info = dict(value_loss=value_loss.item(), policy_loss=policy_loss.item(), dist_entropy=dist_entropy.item(),
                actor_grad_norm=an, critic_grad_norm=cn, ratio=ratio.mean().item())
    return info, gp.detach().numpy().copy(), gc.detach().numpy().copy(), raw_p, raw_c


def buffer_rows(buf: Dict[str, np.ndarray], adv: np.ndarray) -> Dict[str, Optional[np.ndarray]]:
    """The ``_cast`` views of replay_data.py:1127-1145 (rnn states: [T+1,N,A,1,H] -> [N*A*T, H])."""
    H = buf["rnn_states"].shape[-1] * buf["rnn_states"].shape[-2]  # [recurrent_N, state width] flattened per position
    cs = lambda x: np.ascontiguousarray(x[:-1].transpose(1, 2, 0, 3, 4)).reshape(-1, H)
    return {
        "critic_obs": cast_rows(buf["critic_obs"][:-1]), "policy_obs": cast_rows(buf["policy_obs"][:-1]),
        "actions": cast_rows(buf["actions"]), "value_preds": cast_rows(buf["value_preds"][:-1]),
        "returns": cast_rows(buf["returns"][:-1]), "masks": cast_rows(buf["masks"][:-1]),
        "active_masks": cast_rows(buf["active_masks"][:-1]), "action_log_probs": cast_rows(buf["action_log_probs"]),
        "adv": cast_rows(adv),
        "action_masks": cast_rows(buf["action_masks"][:-1]) if buf.get("action_masks") is not None else None,
        "rnn_states": cs(buf["rnn_states"]), "rnn_states_critic": cs(buf["rnn_states_critic"]),
    }


def train_ppo(hp, pspec, ptheta, cspec, ctheta, padam, cadam, vn, buf: Dict[str, np.ndarray], ppo_epoch: int,
              num_mini_batch: int, data_chunk_length: int, order_fn=None):
    """train_ppo (ppo.py:383-458) with recurrent_generator.  Returns (train_info, advantages, chunk orders)."""
    adv = po.advantages(buf["returns"], buf["value_preds"], buf["active_masks"], vn if hp.use_valuenorm else None,
                        hp.use_adv_normalize)
    rows = buffer_rows(buf, adv)
    M = rows["adv"].shape[0]
    keys = ("value_loss", "policy_loss", "dist_entropy", "actor_grad_norm", "critic_grad_norm", "ratio")
    info = {k: 0.0 for k in keys}
    used = []
    for _ in range(ppo_epoch):
        for chunks in (order_fn or recurrent_chunk_order)(M, data_chunk_length, num_mini_batch):
            used.append(np.asarray(chunks).copy())
            sample = chunk_sample(rows, chunks, data_chunk_length)
            step_info = ppo_update(hp, pspec, ptheta, cspec, ctheta, padam, cadam, vn if hp.use_valuenorm else None,
                                   sample)[0]
            for k in keys:
                info[k] += step_info[k]
    n_upd = ppo_epoch * num_mini_batch
    return {k: v / n_upd for k, v in info.items()}, adv, used


# ------------------------------------------------------------------------------------------ joint-action loss (JRPO)
def cast_v3(x: np.ndarray) -> np.ndarray:
    """``_cast_v3`` (buffers/utils/util.py:100-101): [T, N, A, w] -> [N*T, A, w], position = n*T + t."""
    return np.ascontiguousarray(x.transpose(1, 0, 2, 3)).reshape(-1, *x.shape[2:])


def buffer_rows_v3(buf: Dict[str, np.ndarray], adv: np.ndarray) -> Dict[str, Optional[np.ndarray]]:
    """The views of recurrent_generator_v3 (replay_data.py:446-472): the agent axis is kept."""
    H = buf["rnn_states"].shape[-1]
    cs = lambda x: np.ascontiguousarray(x[:-1].transpose(1, 0, 2, 3, 4)).reshape(-1, x.shape[2], H)
    return {
        "critic_obs": cast_v3(buf["critic_obs"][:-1]), "policy_obs": cast_v3(buf["policy_obs"][:-1]),
        "actions": cast_v3(buf["actions"]), "value_preds": cast_v3(buf["value_preds"][:-1]),
        "returns": cast_v3(buf["returns"][:-1]), "masks": cast_v3(buf["masks"][:-1]),
        "active_masks": cast_v3(buf["active_masks"][:-1]), "action_log_probs": cast_v3(buf["action_log_probs"]),
        "adv": cast_v3(adv),
        "action_masks": cast_v3(buf["action_masks"][:-1]) if buf.get("action_masks") is not None else None,
        "rnn_states": cs(buf["rnn_states"]), "rnn_states_critic": cs(buf["rnn_states_critic"]),
    }


def chunk_sample_v3(rows: Dict[str, Optional[np.ndarray]], chunks: np.ndarray, L: int):
    """One minibatch of replay_data.py:474-551: [L, Nc, A, w] flattened to [L*Nc*A, w] (row = (l*Nc + i)*A + a);
    rnn states [Nc*A, H] from each chunk's first position."""
    first = np.asarray(chunks, dtype=np.int64) * L
    pidx = first[None, :] + np.arange(L, dtype=np.int64)[:, None]  # [L, Nc] positions
    out = {}
    for k, v in rows.items():
        if v is None:
            out[k] = None
        elif k in ("rnn_states", "rnn_states_critic"):
            out[k] = v[first].reshape(-1, v.shape[-1])
        else:
            out[k] = v[pidx].reshape(-1, v.shape[-1])
    return out


def prepare_loss_jrpo(hp: po.PPOHyper, pspec, ptheta, cspec, ctheta, vn, s: Dict[str, torch.Tensor], A: int):
    """PPOAlgorithm.prepare_loss with use_joint_action_loss (ppo.py:254-361): the critic sees agent 0's rows only
    (to_single_np), the ratio is the exponential of the log-ratio summed over agents and action dimensions, the
    advantage and the active mask are agent 0's; the entropy still averages over every agent row."""
    single = lambda x: x.reshape(-1, A, *x.shape[1:])[:, 0, ...]
    critic_obs, h_c, critic_masks = single(s["critic_obs"]), single(s["rnn_states_critic"]), single(s["masks"])
    value_preds, returns = single(s["value_preds"]), single(s["returns"])
    adv = s["adv"].reshape(-1, A, 1)[:, 0, :]
    values, _ = rnn_tower_forward(cspec, ctheta, critic_obs, h_c, critic_masks)
    out, _ = rnn_tower_forward(pspec, ptheta, s["policy_obs"], s["rnn_states"], s["masks"])
    active_all, old_logp, action = s["active_masks"], s["action_log_probs"], s["actions"]
    am = active_all if hp.use_policy_active_masks else None
    if pspec.head == HEAD_CATEGORICAL:
        dist = torch.distributions.Categorical(logits=po.masked_logits(out, s["action_masks"]))
        logp = dist.log_prob(action.squeeze(-1).long()).view(action.size(0), -1).sum(-1).unsqueeze(-1)
        ent = dist.entropy()
        dist_entropy = (ent * am.squeeze(-1)).sum() / am.sum() if am is not None else ent.mean()
    else:
        std = pspec.split(ptheta)["logstd"].exp()
        dist = torch.distributions.Normal(out, std.expand_as(out))
        logp = dist.log_prob(action)
        ent = dist.entropy()
        dist_entropy = (ent * am).sum() / am.sum() if am is not None else ent.mean()
    joint = lambda x: x.reshape(-1, A, x.shape[-1]).sum(dim=(1, -1), keepdim=True).reshape(-1, 1)
    ratio = torch.exp(joint(logp) - joint(old_logp))
    active = active_all.reshape(-1, A, 1)[:, 0, :]
    if hp.dual_clip_ppo:
        ratio = torch.min(ratio, torch.tensor(hp.dual_clip_coeff))
    surr_final = torch.min(ratio * adv, torch.clamp(ratio, 1.0 - hp.clip_param, 1.0 + hp.clip_param) * adv)
    if hp.use_policy_active_masks:
        policy_loss = (-torch.sum(surr_final, dim=-1, keepdim=True) * active).sum() / active.sum()
    else:
        policy_loss = -torch.sum(surr_final, dim=-1, keepdim=True).mean()
    value_loss = po.cal_value_loss(hp, vn, values, value_preds, returns, active)
    return [policy_loss - dist_entropy * hp.entropy_coef, value_loss * hp.value_loss_coef], value_loss, policy_loss, \
        dist_entropy, ratio


def train_ppo_jrpo(hp, pspec, ptheta, cspec, ctheta, padam, cadam, vn, buf: Dict[str, np.ndarray], ppo_epoch: int,
                   num_mini_batch: int, data_chunk_length: int):
    """train_ppo (ppo.py:383-458) with recurrent_generator_v3 + the joint-action loss."""
    adv = po.advantages(buf["returns"], buf["value_preds"], buf["active_masks"], vn if hp.use_valuenorm else None,
                        hp.use_adv_normalize)
    rows = buffer_rows_v3(buf, adv)
    A = buf["actions"].shape[2]
    positions = rows["adv"].shape[0]  # N*T
    keys = ("value_loss", "policy_loss", "dist_entropy", "actor_grad_norm", "critic_grad_norm", "ratio")
    info = {k: 0.0 for k in keys}
    used = []
    t = lambda a: None if a is None else torch.as_tensor(a, dtype=torch.float32)
    for _ in range(ppo_epoch):
        for chunks in recurrent_chunk_order(positions, data_chunk_length, num_mini_batch):
            used.append(np.asarray(chunks).copy())
            s = {k: t(v) for k, v in chunk_sample_v3(rows, chunks, data_chunk_length).items()}
            pth = ptheta.detach().clone().requires_grad_(True)
            cth = ctheta.detach().clone().requires_grad_(True)
            loss_list, value_loss, policy_loss, dist_entropy, ratio = prepare_loss_jrpo(
                hp, pspec, pth, cspec, cth, vn if hp.use_valuenorm else None, s, A)
            for loss in loss_list:
                loss.backward()
            gp, an = po.clip_grad_norm(pth.grad, hp.max_grad_norm) if hp.use_max_grad_norm else (pth.grad, float(pth.grad.norm(2)))
            gc, cn = po.clip_grad_norm(cth.grad, hp.max_grad_norm) if hp.use_max_grad_norm else (cth.grad, float(cth.grad.norm(2)))
            padam.step(ptheta, gp)
            cadam.step(ctheta, gc)
            step = dict(value_loss=value_loss.item(), policy_loss=policy_loss.item(), dist_entropy=dist_entropy.item(),
                        actor_grad_norm=an, critic_grad_norm=cn, ratio=ratio.mean().item())
            for k in keys:
                info[k] += step[k]
    n_upd = ppo_epoch * num_mini_batch
    return {k: v / n_upd for k, v in info.items()}, adv, used
