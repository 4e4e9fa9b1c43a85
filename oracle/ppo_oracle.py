"""CPU oracle for the on-policy rollout + PPO update hot path - TEST INFRASTRUCTURE.

A restatement, in numpy + plain torch-CPU ops, of what the reference (OpenRL v0.2.1) computes on
this path.  Each function cites the reference file:line it follows.  It exists to CHECK the HIP
engine (``tests/``, ``__graft_entry__.smoke()``) and to be timed as the CPU baseline
(``bench.py``'s ``cpu_baseline`` leg, kind "port").  Nothing under ``openrl_amd/`` imports it.

Pinning: the reference ships no golden vectors for this path, so the
oracle is pinned against the REAL reference classes run in the authoring container
(``oracle/gen_golden.py`` -> ``tests/golden/*.npz``; checked by ``tests/test_oracle_cpu.py``),
including the known-answer GAE vector of SURVEY.md section 8c.

Third-party arithmetic of the reference that is not under /root/reference: torch (unpinned in
setup.py:32; 2.10.0 here) for Linear/LayerNorm/Adam/clip_grad_norm_/distributions, numpy for the
buffer maths.  The oracle calls the same torch CPU primitives (F.linear, F.layer_norm, autograd).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from . import philox as px

HEAD_VALUE, HEAD_CATEGORICAL, HEAD_GAUSSIAN = 0, 1, 2


# =====================================================================================================
# ValueNorm  (openrl/modules/utils/valuenorm.py:6-106)
# =====================================================================================================
class ValueNormOracle:
    """State = (running_mean, running_mean_sq, debiasing_term), scalar input_shape=1, norm_axes=1."""

    def __init__(self, beta: float = 0.99999, epsilon: float = 1e-5):
        self.beta = beta
        self.epsilon = epsilon
        self.running_mean = torch.zeros(1)
        self.running_mean_sq = torch.zeros(1)
        self.debiasing_term = torch.tensor(0.0)

    def state(self) -> np.ndarray:
        return np.array([self.running_mean.item(), self.running_mean_sq.item(), self.debiasing_term.item()],
                        dtype=np.float32)

    def set_state(self, s) -> None:
        self.running_mean = torch.tensor([float(s[0])], dtype=torch.float32)
        self.running_mean_sq = torch.tensor([float(s[1])], dtype=torch.float32)
        self.debiasing_term = torch.tensor(float(s[2]), dtype=torch.float32)

    def running_mean_var(self):  # valuenorm.py:45-52
        debiased_mean = self.running_mean / self.debiasing_term.clamp(min=self.epsilon)
        debiased_mean_sq = self.running_mean_sq / self.debiasing_term.clamp(min=self.epsilon)
        debiased_var = (debiased_mean_sq - debiased_mean ** 2).clamp(min=1e-2)
        return debiased_mean, debiased_var

    def update(self, x) -> None:  # valuenorm.py:54-77
        x = torch.as_tensor(np.asarray(x) if not torch.is_tensor(x) else x, dtype=torch.float32)
        batch_mean = x.mean(dim=0)
        batch_sq_mean = (x ** 2).mean(dim=0)
        w = self.beta
        self.running_mean.mul_(w).add_(batch_mean * (1.0 - w))
        self.running_mean_sq.mul_(w).add_(batch_sq_mean * (1.0 - w))
        self.debiasing_term.mul_(w).add_(1.0 * (1.0 - w))

    def normalize(self, x):  # valuenorm.py:79-91
        x = torch.as_tensor(np.asarray(x) if not torch.is_tensor(x) else x, dtype=torch.float32)
        mean, var = self.running_mean_var()
        return (x - mean[None]) / torch.sqrt(var)[None]

    def denormalize(self, x) -> np.ndarray:  # valuenorm.py:93-106 (returns numpy)
        x = torch.as_tensor(np.asarray(x), dtype=torch.float32)
        mean, var = self.running_mean_var()
        return (x * torch.sqrt(var)[None] + mean[None]).numpy()


# =====================================================================================================
# K6  GAE / returns  (openrl/buffers/replay_data.py:320-423)
# =====================================================================================================
def compute_returns(rewards, value_preds, masks, bad_masks, next_value, gamma, gae_lambda, use_gae=True,
                    use_proper_time_limits=False, value_normalizer: Optional[ValueNormOracle] = None):
    """All arrays float32 ``[T(+1), N, A, 1]``; returns (returns[T+1,...], value_preds with slot T filled)."""
    rewards = np.asarray(rewards, dtype=np.float32)
    value_preds = np.array(value_preds, dtype=np.float32, copy=True)
    masks = np.asarray(masks, dtype=np.float32)
    returns = np.zeros_like(value_preds)
    T = rewards.shape[0]
    vn = value_normalizer
    den = (lambda v: vn.denormalize(v)) if vn is not None else (lambda v: v)
    if use_proper_time_limits:
        bad_masks = np.asarray(bad_masks, dtype=np.float32)
        if use_gae:
            value_preds[-1] = next_value
            gae = 0
            for step in reversed(range(T)):
                delta = rewards[step] + gamma * den(value_preds[step + 1]) * masks[step + 1] - den(value_preds[step])
                if vn is not None:
                    gae = delta + gamma * gae_lambda * gae * masks[step + 1]  # :337-340
                else:
                    gae = delta + gamma * gae_lambda * masks[step + 1] * gae  # :353-356
                gae = gae * bad_masks[step + 1]
                returns[step] = gae + den(value_preds[step])
        else:
            returns[-1] = next_value
            for step in reversed(range(T)):
                returns[step] = (returns[step + 1] * gamma * masks[step + 1] + rewards[step]) * bad_masks[step + 1] + (
                    1 - bad_masks[step + 1]) * den(value_preds[step])
    else:
        if use_gae:
            value_preds[-1] = next_value
            gae = 0
            for step in reversed(range(T)):
                delta = rewards[step] + gamma * den(value_preds[step + 1]) * masks[step + 1] - den(value_preds[step])
                gae = delta + gamma * gae_lambda * masks[step + 1] * gae
                returns[step] = gae + den(value_preds[step])
        else:
            returns[-1] = next_value
            for step in reversed(range(T)):
                returns[step] = returns[step + 1] * gamma * masks[step + 1] + rewards[step]
    return returns, value_preds


# =====================================================================================================
# K7  advantages  (openrl/algorithms/ppo.py:384-409)
# =====================================================================================================
def advantages(returns, value_preds, active_masks, value_normalizer: Optional[ValueNormOracle],
               use_adv_normalize=False):
    if value_normalizer is not None:
        adv = returns[:-1] - value_normalizer.denormalize(value_preds[:-1])
    else:
        adv = returns[:-1] - value_preds[:-1]
    if use_adv_normalize:
        adv = (adv - adv.mean()) / (adv.std() + 1e-5)
    adv_copy = adv.copy()
    adv_copy[active_masks[:-1] == 0.0] = np.nan
    mean_adv = np.nanmean(adv_copy)
    std_adv = np.nanstd(adv_copy)
    return ((adv - mean_adv) / (std_adv + 1e-5)).astype(np.float32)


# =====================================================================================================
# K8  minibatch order  (openrl/buffers/replay_data.py:553-646)
# =====================================================================================================
def feed_forward_indices(batch_size: int, num_mini_batch: int) -> List[np.ndarray]:
    """BatchSampler(SubsetRandomSampler(range(M)), M // nmb, drop_last=True) == chunks of torch.randperm(M)
    on the default CPU generator (verified against the reference in gen_golden.py)."""
    mbs = batch_size // num_mini_batch
    perm = torch.randperm(batch_size).numpy()
    return [perm[i * mbs:(i + 1) * mbs] for i in range(batch_size // mbs)]


def transformer_indices(agent_num: int):
    """``feed_forward_generator_transformer`` (replay_data.py:707-804, MATAlgorithm's generator): ``randperm(T * N)``
    over (step, env) pairs, ``mini_batch_size = T * N // nmb`` pairs per minibatch, every pair expanded to the rows of
    all its agents in agent order (``_shuffle_agent_grid`` keeps the agent axis).  Returns an ``index_fn`` for
    ``train_ppo``."""
    def index_fn(batch_rows: int, num_mini_batch: int) -> List[np.ndarray]:
        pairs = batch_rows // agent_num
        mbs = pairs // num_mini_batch
        perm = torch.randperm(pairs).numpy()
        a = np.arange(agent_num)
        return [(perm[i * mbs:(i + 1) * mbs][:, None] * agent_num + a).reshape(-1) for i in range(num_mini_batch)]
    return index_fn


def flat_rows(x: np.ndarray) -> np.ndarray:
    """``x[:-1].reshape(-1, width)`` row order (t*N+n)*A+a (replay_data.py:594-613)."""
    return x.reshape(-1, x.shape[-1])


# =====================================================================================================
# Towers  (modules/networks/utils/mlp.py:8-46, act.py, distributions.py, value_network.py)
# =====================================================================================================
@dataclass
class TowerSpec:
    obs_dim: int
    n_out: int
    head: int
    hidden: int = 64

    def sizes(self):
        D, H, K = self.obs_dim, self.hidden, self.n_out
        s = [("W1", (H, D)), ("b1", (H,)), ("g1", (H,)), ("be1", (H,)), ("W2", (H, H)), ("b2", (H,)), ("g2", (H,)),
             ("be2", (H,)), ("W3", (K, H)), ("b3", (K,))]
        if self.head == HEAD_GAUSSIAN:
            s.append(("logstd", (K,)))
        return s

    def n_params(self):
        return sum(int(np.prod(sh)) for _, sh in self.sizes())

    def split(self, theta: torch.Tensor) -> Dict[str, torch.Tensor]:
        out, o = {}, 0
        for name, sh in self.sizes():
            n = int(np.prod(sh))
            out[name] = theta[o:o + n].view(*sh)
            o += n
        return out


def init_tower(spec: TowerSpec, gain_head: float, use_orthogonal: bool = True, activation_id: int = 1) -> torch.Tensor:
    """Initial flat parameters with the reference's RNG consumption order.

    MLPLayer (mlp.py:8-39): fc1 = init_(nn.Linear(D,H)) [orthogonal, gain=calculate_gain(act)], LayerNorm,
    fc3 = init_(nn.Linear(H,H)), LayerNorm; head: Categorical/DiagGaussian linear with gain=cfg.gain
    (act.py:16, distributions.py:58-66) or v_out with gain 1 (value_network.py:103-109); biases 0.
    nn.Linear's own default init draws from the generator BEFORE the orthogonal init overwrites it."""
    init_method = torch.nn.init.orthogonal_ if use_orthogonal else torch.nn.init.xavier_uniform_
    gain = torch.nn.init.calculate_gain(["tanh", "relu", "leaky_relu", "selu"][activation_id])
    D, H, K = spec.obs_dim, spec.hidden, spec.n_out
    fc1 = torch.nn.Linear(D, H)
    init_method(fc1.weight.data, gain=gain)
    fc3 = torch.nn.Linear(H, H)
    init_method(fc3.weight.data, gain=gain)
    head = torch.nn.Linear(H, K)
    init_method(head.weight.data, gain=gain_head)
    parts = [fc1.weight.data.reshape(-1), torch.zeros(H), torch.ones(H), torch.zeros(H), fc3.weight.data.reshape(-1),
             torch.zeros(H), torch.ones(H), torch.zeros(H), head.weight.data.reshape(-1), torch.zeros(K)]
    if spec.head == HEAD_GAUSSIAN:
        parts.append(torch.zeros(K))
    return torch.cat(parts).clone()


def trunk_forward(p: Dict[str, torch.Tensor], x: torch.Tensor) -> torch.Tensor:
    """MLPBase.forward with layer_N=1, no feature norm (mlp.py:41-46,160-176)."""
    H = p["b1"].numel()
    h = F.layer_norm(F.relu(F.linear(x, p["W1"], p["b1"])), (H,), p["g1"], p["be1"], 1e-5)
    return F.layer_norm(F.linear(h, p["W2"], p["b2"]), (H,), p["g2"], p["be2"], 1e-5)


def tower_forward(spec: TowerSpec, theta: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    if getattr(spec, "general", False):  # general towers restate their own forward (oracle/gen_oracle.py)
        return spec.forward(theta, x)
    p = spec.split(theta)
    return F.linear(trunk_forward(p, x), p["W3"], p["b3"])


def masked_logits(logits: torch.Tensor, action_masks: Optional[torch.Tensor]) -> torch.Tensor:
    if action_masks is not None:  # distributions.py:70-71 (in-place on a fresh linear output)
        logits = logits.clone() if not logits.requires_grad else logits
        logits = torch.where(action_masks == 0, torch.full_like(logits, -6e4), logits)
    return logits


def inverse_cdf_sample(logits: np.ndarray, u: np.ndarray) -> np.ndarray:
    """The ENGINE's categorical sampler (csrc/orl_mlp.h cat_sample) restated in float32 numpy: first c
    with cumsum(p)[c] > u * sum(p); falls back to the last c with p > 0.  (The reference uses
    torch.multinomial, whose stream cannot be reproduced by a HIP kernel - SURVEY.md section 7.)"""
    lg = logits.astype(np.float32)
    mx = lg.max(-1, keepdims=True)
    lse = mx + np.log(np.exp(lg - mx).sum(-1, keepdims=True, dtype=np.float32))
    p = np.exp(lg - lse).astype(np.float32)
    out = np.empty(lg.shape[0], dtype=np.int64)
    for i in range(lg.shape[0]):
        tot = np.float32(0)
        for c in range(lg.shape[1]):
            tot = np.float32(tot + p[i, c])
        ut = np.float32(u[i] * tot)
        cum, a, last = np.float32(0), -1, 0
        for c in range(lg.shape[1]):
            cum = np.float32(cum + p[i, c])
            if p[i, c] > 0:
                last = c
            if a < 0 and cum > ut:
                a = c
        out[i] = last if a < 0 else a
    return out


def get_actions(pspec, ptheta, cspec, ctheta, policy_obs, critic_obs, action_masks=None, deterministic=False,
                forced_u=None):
    """PPOModule.get_actions (ppo_module.py:102-138) with the engine's sampler semantics (forced uniforms /
    normals).  Returns values [B,1], actions [B,a] float32, logp [B,a]."""
    with torch.no_grad():
        x = torch.as_tensor(policy_obs, dtype=torch.float32)
        out = tower_forward(pspec, ptheta, x)
        if pspec.head == HEAD_CATEGORICAL:
            am = None if action_masks is None else torch.as_tensor(action_masks, dtype=torch.float32)
            lg = masked_logits(out, am)
            dist = torch.distributions.Categorical(logits=lg)
            if deterministic:
                a = dist.probs.argmax(dim=-1)
            else:
                a = torch.as_tensor(inverse_cdf_sample(lg.numpy(), np.asarray(forced_u, dtype=np.float32).reshape(-1)))
            logp = dist.log_prob(a).unsqueeze(-1)
            actions = a.unsqueeze(-1).float()
        else:
            p = pspec.split(ptheta)
            std = p["logstd"].exp()
            dist = torch.distributions.Normal(out, std.expand_as(out))
            if deterministic:
                actions = out
            else:
                actions = out + std * torch.as_tensor(forced_u, dtype=torch.float32)
            logp = dist.log_prob(actions)
        values = tower_forward(cspec, ctheta, torch.as_tensor(critic_obs, dtype=torch.float32))
    return values.numpy(), actions.numpy(), logp.numpy()


# =====================================================================================================
# K9-K12  PPO loss  (openrl/algorithms/ppo.py:178-220, 238-361; modules/utils/util.py:20-27)
# =====================================================================================================
@dataclass
class PPOHyper:
    clip_param: float = 0.2
    entropy_coef: float = 0.01
    value_loss_coef: float = 0.5
    huber_delta: float = 10.0
    dual_clip_coeff: float = 3.0
    max_grad_norm: float = 10.0
    use_clipped_value_loss: bool = True
    use_huber_loss: bool = True
    use_value_active_masks: bool = True
    use_policy_active_masks: bool = True
    use_valuenorm: bool = True
    dual_clip_ppo: bool = False
    use_max_grad_norm: bool = True
    use_adv_normalize: bool = False
    a2c: bool = False  # A2CAlgorithm.prepare_loss (algorithms/a2c.py:88-98): -adv * logp, no ratio / clip


def huber_loss(e, d):  # modules/utils/util.py:20-23
    a = (abs(e) <= d).float()
    b = (abs(e) > d).float()
    return a * e ** 2 / 2 + b * d * (abs(e) - d / 2)


def mse_loss(e):  # modules/utils/util.py:26-27
    return e ** 2 / 2


def cal_value_loss(hp: PPOHyper, vn: Optional[ValueNormOracle], values, value_preds_batch, return_batch,
                   active_masks_batch):  # ppo.py:178-220
    value_pred_clipped = value_preds_batch + (values - value_preds_batch).clamp(-hp.clip_param, hp.clip_param)
    if hp.use_valuenorm and vn is not None:
        vn.update(return_batch)
        error_clipped = vn.normalize(return_batch) - value_pred_clipped
        error_original = vn.normalize(return_batch) - values
    else:
        error_clipped = return_batch - value_pred_clipped
        error_original = return_batch - values
    if hp.use_huber_loss:
        vlc, vlo = huber_loss(error_clipped, hp.huber_delta), huber_loss(error_original, hp.huber_delta)
    else:
        vlc, vlo = mse_loss(error_clipped), mse_loss(error_original)
    value_loss = torch.max(vlo, vlc) if hp.use_clipped_value_loss else vlo
    if hp.use_value_active_masks:
        return (value_loss * active_masks_batch).sum() / active_masks_batch.sum()
    return value_loss.mean()


def evaluate_actions(pspec, ptheta, obs, action, action_masks, active_masks, use_policy_active_masks=True):
    """PolicyNetwork.eval_actions + ACTLayer.evaluate_actions (policy_network.py:164-203, act.py:102-172)."""
    out = tower_forward(pspec, ptheta, obs)
    am = active_masks if use_policy_active_masks else None
    if pspec.head == HEAD_CATEGORICAL:
        lg = masked_logits(out, action_masks)
        dist = torch.distributions.Categorical(logits=lg)
        logp = dist.log_prob(action.squeeze(-1).long()).view(action.size(0), -1).sum(-1).unsqueeze(-1)
        ent = dist.entropy()
        dist_entropy = (ent * am.squeeze(-1)).sum() / am.sum() if am is not None else ent.mean()
    else:
        std = pspec.split(ptheta)["logstd"].exp()
        dist = torch.distributions.Normal(out, std.expand_as(out))
        logp = dist.log_prob(action)
        ent = dist.entropy()
        dist_entropy = (ent * am).sum() / am.sum() if am is not None else ent.mean()
    return logp, dist_entropy


def prepare_loss(hp: PPOHyper, pspec, ptheta, cspec, ctheta, vn, sample):
    """PPOAlgorithm.prepare_loss + construct_loss_list (ppo.py:226-361), non-joint, no policy v-head."""
    (critic_obs, obs, actions, value_preds, returns, active, old_logp, adv, action_masks) = sample
    values = tower_forward(cspec, ctheta, critic_obs)
    logp, dist_entropy = evaluate_actions(pspec, ptheta, obs, actions, action_masks, active, hp.use_policy_active_masks)
    ratio = torch.exp(logp - old_logp)
    if hp.dual_clip_ppo:
        ratio = torch.min(ratio, torch.tensor(hp.dual_clip_coeff))
    surr1 = ratio * adv
    surr2 = torch.clamp(ratio, 1.0 - hp.clip_param, 1.0 + hp.clip_param) * adv
    surr_final = torch.min(surr1, surr2)
    if hp.a2c:
        surr_final = adv * logp  # policy_gradient_loss = -adv.detach() * action_log_probs
        ratio = torch.zeros(1)   # a2c.py:139
    if hp.use_policy_active_masks:
        policy_loss = (-torch.sum(surr_final, dim=-1, keepdim=True) * active).sum() / active.sum()
    else:
        policy_loss = -torch.sum(surr_final, dim=-1, keepdim=True).mean()
    value_loss = cal_value_loss(hp, vn, values, value_preds, returns, active)
    loss_list = [policy_loss - dist_entropy * hp.entropy_coef, value_loss * hp.value_loss_coef]
    return loss_list, value_loss, policy_loss, dist_entropy, ratio


def clip_grad_norm(grad: torch.Tensor, max_norm: float) -> Tuple[torch.Tensor, float]:
    """torch.nn.utils.clip_grad_norm_ on one flat tensor (ppo.py:132-145)."""
    total = grad.norm(2)
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    return grad * coef, float(total)


class AdamOracle:
    """torch.optim.Adam single-tensor maths (rl_module.py:80-85: lr, eps=opti_eps, weight_decay)."""

    def __init__(self, n: int, lr: float, eps: float = 1e-5, weight_decay: float = 0.0):
        self.m = torch.zeros(n)
        self.v = torch.zeros(n)
        self.t = 0
        self.lr, self.eps, self.wd = lr, eps, weight_decay

    def step(self, theta: torch.Tensor, grad: torch.Tensor) -> None:
        b1, b2 = 0.9, 0.999
        self.t += 1
        g = grad if self.wd == 0 else grad.add(theta, alpha=self.wd)
        self.m.lerp_(g, 1 - b1)
        self.v.mul_(b2).addcmul_(g, g, value=1 - b2)
        bc1 = 1 - b1 ** self.t
        bc2 = 1 - b2 ** self.t
        step_size = self.lr / bc1
        denom = (self.v.sqrt() / math.sqrt(bc2)).add_(self.eps)
        theta.addcdiv_(self.m, denom, value=-step_size)


def ppo_update(hp: PPOHyper, pspec, ptheta, cspec, ctheta, padam: AdamOracle, cadam: AdamOracle, vn, sample_np):
    """PPOAlgorithm.ppo_update (ppo.py:46-176): losses, two backward passes, clip, two Adam steps."""
    t = lambda a: None if a is None else torch.as_tensor(a, dtype=torch.float32)
    sample = tuple(t(a) for a in sample_np)
    pth = ptheta.detach().clone().requires_grad_(True)
    cth = ctheta.detach().clone().requires_grad_(True)
    loss_list, value_loss, policy_loss, dist_entropy, ratio = prepare_loss(hp, pspec, pth, cspec, cth, vn, sample)
    for loss in loss_list:
        loss.backward()
    if hp.use_max_grad_norm:
        gp, an = clip_grad_norm(pth.grad, hp.max_grad_norm)
        gc, cn = clip_grad_norm(cth.grad, hp.max_grad_norm)
    else:
        gp, an = pth.grad, float(pth.grad.norm(2))
        gc, cn = cth.grad, float(cth.grad.norm(2))
    padam.step(ptheta, gp)
    cadam.step(ctheta, gc)
    info = dict(value_loss=value_loss.item(), policy_loss=policy_loss.item(), dist_entropy=dist_entropy.item(),
                actor_grad_norm=an, critic_grad_norm=cn, ratio=ratio.mean().item())
    return info, gp.detach().numpy().copy(), gc.detach().numpy().copy()


# =====================================================================================================
# Synthetic fixed-step env and CartPole-v1, host restatements of csrc/orl_act.hip's device envs
# =====================================================================================================
class SynthEnvOracle:
    """obs ~ N(0,1) keyed (seed, env, t); reward U(0,1); done every `episode_limit` steps with per-env
    phase (env*7) mod limit; auto-reset semantics; duck-typed like examples/isaac/isaac2openrl.py:28-88."""

    def __init__(self, n_envs, obs_dim, seed, episode_limit=200):
        self.N, self.D, self.seed, self.limit = n_envs, obs_dim, seed, episode_limit
        self.t = 0
        self.steps = (np.arange(n_envs) * 7) % episode_limit

    def _obs(self, t):
        env = np.arange(self.N, dtype=np.uint32)
        cols = []
        for b in range((self.D + 3) // 4):
            x, y, z, w = px.philox4x32_10(self.seed, env, 0x0B5E0000 + b, t & 0xFFFFFFFF, t >> 32)
            n0, n1 = px.box_muller(x, y)
            n2, n3 = px.box_muller(z, w)
            cols += [n0, n1, n2, n3]
        return np.stack(cols[:self.D], axis=-1).astype(np.float32)

    def reset(self):
        self.t = 0
        self.steps = (np.arange(self.N) * 7) % self.limit
        return self._obs(0)[:, None, :]

    def step(self, actions=None):
        env = np.arange(self.N, dtype=np.uint32)
        x, _, _, _ = px.philox4x32_10(self.seed, env, 0x4E3A0000, self.t & 0xFFFFFFFF, self.t >> 32)
        rew = px.u01(x)
        self.steps = self.steps + 1
        done = self.steps >= self.limit
        self.steps = np.where(done, 0, self.steps)
        self.t += 1
        return self._obs(self.t)[:, None, :], rew[:, None, None].astype(np.float32), done[:, None], [{} for _ in range(self.N)]


def cartpole_step_f32(state: np.ndarray, action: np.ndarray):
    """gymnasium CartPole-v1 euler step (classic_control/cartpole.py step()), float32 like the device."""
    f = np.float32
    s = state.astype(np.float32).copy()
    force = np.where(action == 1, f(10.0), f(-10.0)).astype(np.float32)
    costh, sinth = np.cos(s[:, 2]).astype(np.float32), np.sin(s[:, 2]).astype(np.float32)
    temp = ((force + f(0.05) * s[:, 3] * s[:, 3] * sinth) / f(1.1)).astype(np.float32)
    thetaacc = ((f(9.8) * sinth - costh * temp) / (f(0.5) * (f(4.0) / f(3.0) - f(0.1) * costh * costh / f(1.1)))).astype(
        np.float32)
    xacc = (temp - f(0.05) * thetaacc * costh / f(1.1)).astype(np.float32)
    out = np.stack([s[:, 0] + f(0.02) * s[:, 1], s[:, 1] + f(0.02) * xacc, s[:, 2] + f(0.02) * s[:, 3],
                    s[:, 3] + f(0.02) * thetaacc], axis=-1).astype(np.float32)
    th = f(12.0 * 2.0 * math.pi / 360.0)
    term = (out[:, 0] < -2.4) | (out[:, 0] > 2.4) | (out[:, 2] < -th) | (out[:, 2] > th)
    return out, term


def cartpole_reset_state(seed: int, env: np.ndarray, episode: np.ndarray) -> np.ndarray:
    x, y, z, w = px.philox4x32_10(seed, env.astype(np.uint32), 0xCA470000, episode.astype(np.uint32), 0)
    f = np.float32
    return np.stack([px.u01(c) * f(0.1) - f(0.05) for c in (x, y, z, w)], axis=-1).astype(np.float32)


class CartPoleEnvOracle:
    """Vectorised CartPole-v1 on the host with the device env's semantics (csrc/orl_env.h: gymnasium's Euler step in
    float32, termination |x| > 2.4 or |theta| > 12 deg, truncation at ``episode_limit`` = 500, auto-reset to the Philox
    start state of (seed, env, episode) with the observation of the NEW episode returned - sync_venv.py:178-247);
    reward 1 per step.  Duck-typed like SynthEnvOracle; drives oracle/cpu_trainer.py's learning comparison."""

    def __init__(self, n_envs, seed, episode_limit=500):
        self.N, self.seed, self.limit = n_envs, seed, episode_limit
        self.reset()

    def reset(self):
        self.episode = np.zeros(self.N, np.int64)
        self.steps = np.zeros(self.N, np.int64)
        self.state = cartpole_reset_state(self.seed, np.arange(self.N), self.episode)
        return self.state[:, None, :].copy()

    def step(self, actions):
        a = np.asarray(actions).reshape(self.N).astype(np.int64)
        out, term = cartpole_step_f32(self.state, a)
        self.steps = self.steps + 1
        done = term | (self.steps >= self.limit)
        self.episode = np.where(done, self.episode + 1, self.episode)
        fresh = cartpole_reset_state(self.seed, np.arange(self.N), self.episode)
        self.state = np.where(done[:, None], fresh, out).astype(np.float32)
        self.steps = np.where(done, 0, self.steps)
        rew = np.ones((self.N, 1, 1), np.float32)
        return self.state[:, None, :].copy(), rew, done[:, None], [{} for _ in range(self.N)]


# =====================================================================================================
# train_ppo  (openrl/algorithms/ppo.py:383-458) over a dict of buffer arrays [T(+1), N, A, .]
# =====================================================================================================
def train_ppo(hp: PPOHyper, pspec, ptheta, cspec, ctheta, padam, cadam, vn, buf: Dict[str, np.ndarray],
              ppo_epoch: int, num_mini_batch: int, index_fn=None):
    """Returns (train_info averaged like ppo.py:453-456, advantages, list of minibatch index arrays).

    ``index_fn(M, nmb)`` overrides the minibatch order (default: feed_forward_indices = torch.randperm)."""
    adv = advantages(buf["returns"], buf["value_preds"], buf["active_masks"], vn if hp.use_valuenorm else None,
                     hp.use_adv_normalize)
    rows = {
        "critic_obs": flat_rows(buf["critic_obs"][:-1]), "policy_obs": flat_rows(buf["policy_obs"][:-1]),
        "actions": flat_rows(buf["actions"]), "value_preds": flat_rows(buf["value_preds"][:-1]),
        "returns": flat_rows(buf["returns"][:-1]), "active_masks": flat_rows(buf["active_masks"][:-1]),
        "action_log_probs": flat_rows(buf["action_log_probs"]), "adv": adv.reshape(-1, 1),
        "action_masks": flat_rows(buf["action_masks"][:-1]) if buf.get("action_masks") is not None else None,
    }
    M = rows["adv"].shape[0]
    keys = ("value_loss", "policy_loss", "dist_entropy", "actor_grad_norm", "critic_grad_norm", "ratio")
    info = {k: 0.0 for k in keys}
    used = []
    for _ in range(ppo_epoch):
        batches = (index_fn or feed_forward_indices)(M, num_mini_batch)
        for idx in batches:
            used.append(np.asarray(idx).copy())
            g = lambda k: None if rows[k] is None else rows[k][idx]
            sample = (g("critic_obs"), g("policy_obs"), g("actions"), g("value_preds"), g("returns"),
                      g("active_masks"), g("action_log_probs"), g("adv"), g("action_masks"))
            step_info, _, _ = ppo_update(hp, pspec, ptheta, cspec, ctheta, padam, cadam,
                                         vn if hp.use_valuenorm else None, sample)
            for k in keys:
                info[k] += step_info[k]
    n_upd = ppo_epoch * num_mini_batch
    return {k: v / n_upd for k, v in info.items()}, adv, used


def hyper_from_cfg(cfg) -> PPOHyper:
    return PPOHyper(clip_param=cfg.clip_param, entropy_coef=cfg.entropy_coef, value_loss_coef=cfg.value_loss_coef,
                    huber_delta=cfg.huber_delta, dual_clip_coeff=float(cfg.dual_clip_coeff),
                    max_grad_norm=float(cfg.max_grad_norm), use_clipped_value_loss=cfg.use_clipped_value_loss,
                    use_huber_loss=cfg.use_huber_loss, use_value_active_masks=cfg.use_value_active_masks,
                    use_policy_active_masks=cfg.use_policy_active_masks, use_valuenorm=cfg.use_valuenorm,
                    dual_clip_ppo=cfg.dual_clip_ppo, use_max_grad_norm=cfg.use_max_grad_norm,
                    use_adv_normalize=cfg.use_adv_normalize)
