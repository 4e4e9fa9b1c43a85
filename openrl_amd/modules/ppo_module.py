"""``PPOModule`` - policy + critic towers, their Adam state and the ValueNorm state, all as flat
float32 HIP tensors, behind the reference's module interface.

Mirrors ``openrl/modules/ppo_module.py:32-224`` + ``openrl/modules/rl_module.py:65-87``:
``models{"policy","critic"}``, ``optimizers{...}``, ``get_actions / get_values / evaluate_actions / act /
lr_decay / get_critic_value_normalizer / init_rnn_states``.  Forward passes run the fused MFMA tower
kernels (``orl_act_step``); there is no torch.nn module and no CPU path.

Initial weights reproduce the reference bit-for-bit for the same ``cfg.seed``: they are drawn on the
HOST with the same torch CPU-generator consumption order as ``PolicyNetwork`` / ``ValueNetwork``
construction (policy first, then critic; ``nn.Linear`` default init followed by orthogonal init with
gain sqrt(2) for the ReLU trunk, ``cfg.gain`` for the action head, 1 for ``v_out`` - mlp.py:14-39,
act.py:16, value_network.py:103-109) and then uploaded once (policy_network.py:120 does the same).
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Any, Dict, Optional, Union

import numpy as np
import torch

from .. import _native as nat
from .. import ops, ops_rnn, spaces


# ---------------------------------------------------------------------------------------------------------
class ValueNorm:
    """Running-moment value normaliser (openrl/modules/utils/valuenorm.py:6-106); state on the device as
    ``state = [running_mean, running_mean_sq, debiasing_term]`` which the GAE / loss kernels read directly."""

    def __init__(self, input_shape=1, norm_axes=1, beta=0.99999, per_element_update=False, epsilon=1e-5,
                 device="cuda:0"):
        if input_shape != 1 or norm_axes != 1 or per_element_update:
            raise NotImplementedError("ValueNorm: only the scalar-value configuration the PPO path uses is built")
        self.beta, self.epsilon = beta, epsilon
        self.device = nat.require_gpu(device)
        self.state = torch.zeros(3, dtype=torch.float32, device=self.device)
        self._moments = torch.zeros(3, dtype=torch.float64, device=self.device)
        self._scratch = torch.zeros(512, dtype=torch.float64, device=self.device)

    @property
    def running_mean(self):
        return self.state[0:1]

    @property
    def running_mean_sq(self):
        return self.state[1:2]

    @property
    def debiasing_term(self):
        return self.state[2]

    def reset_parameters(self):
        self.state.zero_()

    def running_mean_var(self):
        deb = self.state[2].clamp(min=self.epsilon)
        mean = self.state[0:1] / deb
        var = (self.state[1:2] / deb - mean ** 2).clamp(min=1e-2)
        return mean, var

    def _as_dev(self, x):
        if isinstance(x, np.ndarray):
            x = torch.from_numpy(x)
        return x.to(device=self.device, dtype=torch.float32)

    @torch.no_grad()
    def update(self, input_vector):
        x = self._as_dev(input_vector).contiguous().reshape(-1, 1)
        ops.minibatch_moments(x, 0, None, x.shape[0], self._scratch, self._moments)
        ops.valuenorm_update(self.state, self._moments, self.beta)

    def normalize(self, input_vector):
        x = self._as_dev(input_vector)
        mean, var = self.running_mean_var()
        return (x - mean) / torch.sqrt(var)

    def denormalize(self, input_vector):
        """Returns a numpy array like the reference (valuenorm.py:104-105)."""
        x = self._as_dev(input_vector)
        mean, var = self.running_mean_var()
        return (x * torch.sqrt(var) + mean).cpu().numpy()


# ---------------------------------------------------------------------------------------------------------
def _tower_entries(prefix_head: str, D: int, H: int, K: int, gaussian: bool, recurrent: bool = False):
    """(state_dict key, shape) in the reference's registration order."""
    e = [("base.mlp.fc1.0.weight", (H, D)), ("base.mlp.fc1.0.bias", (H,)), ("base.mlp.fc1.2.weight", (H,)),
         ("base.mlp.fc1.2.bias", (H,)), ("base.mlp.fc3.0.weight", (H, H)), ("base.mlp.fc3.0.bias", (H,)),
         ("base.mlp.fc3.1.weight", (H,)), ("base.mlp.fc3.1.bias", (H,))]
    if recurrent:  # RNNLayer (networks/utils/rnn.py:5-27): one-layer GRU + LayerNorm
        e += [("rnn.rnn.weight_ih_l0", (3 * H, H)), ("rnn.rnn.weight_hh_l0", (3 * H, H)), ("rnn.rnn.bias_ih_l0", (3 * H,)),
              ("rnn.rnn.bias_hh_l0", (3 * H,)), ("rnn.norm.weight", (H,)), ("rnn.norm.bias", (H,))]
    if prefix_head == "critic":
        e += [("v_out.weight", (K, H)), ("v_out.bias", (K,))]
    elif gaussian:
        e += [("act.action_out.fc_mean.weight", (K, H)), ("act.action_out.fc_mean.bias", (K,)),
              ("act.action_out.logstd._bias", (K, 1))]
    else:
        e += [("act.action_out.linear.weight", (K, H)), ("act.action_out.linear.bias", (K,))]
    return e


class Tower:
    """One MLP tower = flat parameter vector ``theta`` + descriptor; quacks enough like a torch module
    (``parameters / state_dict / load_state_dict / train / eval``) for the reference's callers."""

    def __init__(self, role: str, obs_dim: int, n_out: int, head_kind: int, hidden: int, device, theta_host,
                 recurrent: bool = False):
        self.role = role
        self.device = device
        self.recurrent = recurrent
        self.net = ops.net_desc(obs_dim, n_out, head_kind, hidden)
        self.n_params = ops_rnn.rnn_param_count(self.net) if recurrent else ops.param_count(self.net)
        assert theta_host.numel() == self.n_params, (theta_host.numel(), self.n_params)
        self.theta = theta_host.to(device=device, dtype=torch.float32).contiguous()
        self.grad = torch.zeros_like(self.theta)
        self._entries = _tower_entries(role, obs_dim, hidden, n_out, head_kind == ops.HEAD_GAUSSIAN, recurrent)
        self.training = False
        self.value_normalizer: Optional[ValueNorm] = None
        # cfg.use_popart: the reference swaps v_out for a PopArt layer (value_network.py:106-107) but never calls its
        # update / normalize on this path (the algorithm's value_normalizer is ValueNorm or None, base_value_network.py:
        # 29-34), so the flag leaves the arithmetic alone and adds four frozen entries after v_out.{weight,bias}
        self.popart: Optional[OrderedDict] = None

    def enable_popart_entries(self):
        z = lambda v: torch.full((1,), v, dtype=torch.float32, device=self.device)
        self.popart = OrderedDict([("v_out.stddev", z(1.0)), ("v_out.mean", z(0.0)), ("v_out.mean_sq", z(0.0)),
                                   ("v_out.debiasing_term", torch.zeros((), dtype=torch.float32, device=self.device))])

    def parameters(self):
        return [v for _, v in self.named_parameters()]

    def named_parameters(self):
        out, o = [], 0
        for name, shape in self._entries:
            n = int(np.prod(shape))
            out.append((name, self.theta[o:o + n].view(*shape)))
            o += n
        return out

    def state_dict(self):
        sd = OrderedDict()
        if self.value_normalizer is not None:  # ValueNetwork registers them first (base_value_network.py:32)
            sd["value_normalizer.running_mean"] = self.value_normalizer.state[0:1]
            sd["value_normalizer.running_mean_sq"] = self.value_normalizer.state[1:2]
            sd["value_normalizer.debiasing_term"] = self.value_normalizer.state[2]
        for k, v in self.named_parameters():
            sd[k] = v
        if self.popart is not None:
            sd.update(self.popart)
        return sd

    def load_state_dict(self, sd):
        for k, v in self.named_parameters():
            v.copy_(torch.as_tensor(sd[k]).to(self.device, torch.float32).reshape(v.shape))
        if self.popart is not None:
            for k, v in self.popart.items():
                if k in sd:
                    v.copy_(torch.as_tensor(sd[k]).to(self.device, torch.float32).reshape(v.shape))
        if self.value_normalizer is not None and "value_normalizer.running_mean" in sd:
            self.value_normalizer.state[0] = float(sd["value_normalizer.running_mean"].reshape(-1)[0])
            self.value_normalizer.state[1] = float(sd["value_normalizer.running_mean_sq"].reshape(-1)[0])
            self.value_normalizer.state[2] = float(sd["value_normalizer.debiasing_term"])

    def train(self, mode=True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)


class FusedAdam:
    """State of one ``torch.optim.Adam(lr, eps=opti_eps, weight_decay)`` (rl_module.py:80-85); the step
    itself is fused into ``orl_ppo_apply``.  ``param_groups[0]["lr"]`` is honoured (lr_decay)."""

    def __init__(self, tower: Tower, lr: float, eps: float, weight_decay: float):
        self.tower = tower
        self.exp_avg = torch.zeros_like(tower.theta)
        self.exp_avg_sq = torch.zeros_like(tower.theta)
        self.step_count = 0
        self.param_groups = [dict(lr=lr, eps=eps, weight_decay=weight_decay, betas=(0.9, 0.999))]

    def zero_grad(self):
        self.tower.grad.zero_()

    def native_state(self, step: int) -> nat.AdamState:
        g = self.param_groups[0]
        f = nat.fptr
        return nat.AdamState(f(self.tower.theta), f(self.tower.grad), f(self.exp_avg), f(self.exp_avg_sq),
                             float(g["lr"]), float(g["eps"]), float(g["weight_decay"]), int(step))

    def state_dict(self):
        return dict(exp_avg=self.exp_avg, exp_avg_sq=self.exp_avg_sq, step=self.step_count,
                    param_groups=self.param_groups)

    def load_state_dict(self, sd):
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.step_count = int(sd["step"])
        self.param_groups = sd["param_groups"]


def _host_init_tower(D: int, H: int, K: int, gaussian: bool, gain_head: float, use_orthogonal: bool,
                     activation_id: int, recurrent: bool = False) -> torch.Tensor:
    """Host-side initial parameters with the reference's generator consumption (see module docstring).
    Recurrent towers draw the nn.GRU default init and then the orthogonal / xavier init of weight_ih_l0,
    weight_hh_l0 (gain 1, biases 0 - rnn.py:14-26) between the trunk and the head."""
    init_method = torch.nn.init.orthogonal_ if use_orthogonal else torch.nn.init.xavier_uniform_
    gain = torch.nn.init.calculate_gain(["tanh", "relu", "leaky_relu", "selu"][activation_id])
    fc1 = torch.nn.Linear(D, H)
    init_method(fc1.weight.data, gain=gain)
    fc3 = torch.nn.Linear(H, H)
    init_method(fc3.weight.data, gain=gain)
    rnn_parts = []
    if recurrent:
        gru = torch.nn.GRU(H, H, num_layers=1)
        init_method(gru.weight_ih_l0.data)
        init_method(gru.weight_hh_l0.data)
        rnn_parts = [gru.weight_ih_l0.data.reshape(-1), gru.weight_hh_l0.data.reshape(-1), torch.zeros(3 * H),
                     torch.zeros(3 * H), torch.ones(H), torch.zeros(H)]
    head = torch.nn.Linear(H, K)
    init_method(head.weight.data, gain=gain_head)
    zeros, ones = torch.zeros(H), torch.ones(H)
    parts = [fc1.weight.data.reshape(-1), zeros, ones, zeros, fc3.weight.data.reshape(-1), zeros, ones, zeros] + \
        rnn_parts + [head.weight.data.reshape(-1), torch.zeros(K)]
    if gaussian:
        parts.append(torch.zeros(K))
    return torch.cat(parts).clone()


#: model_dict keys -> the reference network class whose construction the built towers reproduce (ppo_module.py:58-89)
BUILT_NETWORKS = {"policy": "PolicyNetwork", "critic": "ValueNetwork", "model": "PolicyValueNetwork"}


def check_model_dict(model_dict) -> None:
    """``model_dict`` (ppo_net.py:57-58 -> ppo_module.py:58-89) injects network CLASSES per role.  The engine runs its
    own HIP towers, so what it can honour is an entry that names the reference's stock class for that role -
    ``PolicyNetwork`` / ``ValueNetwork`` / ``PolicyValueNetwork``, the reference's own or the markers of
    ``openrl_amd.modules.networks`` - whose constructor reads nothing but ``cfg`` and the spaces: the built towers are that
    network.  Anything else (a subclass with its own forward, a foreign ``nn.Module``) cannot run on the kernels and is
    refused loudly rather than silently replaced."""
    for key, net_cls in (model_dict or {}).items():
        if key not in BUILT_NETWORKS:
            raise KeyError("model_dict key %r: the reference reads 'policy', 'critic' and 'model'" % (key,))
        name = getattr(net_cls, "__name__", type(net_cls).__name__)
        module = getattr(net_cls, "__module__", "") or ""
        # identity, not name: the stock class is one DEFINED in the reference's openrl.modules.networks package or the
        # marker of openrl_amd.modules.networks - a user class that merely shares the name has its own forward
        stock = name == BUILT_NETWORKS[key] and (module.startswith("openrl.modules.networks.")
                                                  or module.startswith("openrl_amd.modules.networks"))
        if not stock:
            raise NotImplementedError(
                "model_dict[%r] = %s.%s: only the stock %s (openrl.modules.networks / openrl_amd.modules.networks) is "
                "accepted - the engine's built towers reproduce it from cfg; a custom network class cannot run on the HIP "
                "towers" % (key, module, name, BUILT_NETWORKS[key]))


def check_model_dict_roles(model_dict, share_model: bool) -> None:
    """ppo_module.py:58-89: with ``use_share_model`` the reference reads ONLY model_dict['model'], without it only
    'policy' / 'critic'.  An entry for the other mode would be silently ignored there; here it is refused."""
    keys = set(model_dict or {})
    if share_model and keys - {"model"}:
        raise ValueError("model_dict has %s but use_share_model is on: the shared network is model_dict['model']"
                         % sorted(keys - {"model"}))
    if not share_model and "model" in keys:
        raise ValueError("model_dict['model'] is the shared PolicyValueNetwork: it needs use_share_model=True")


class PPOModule:
    #: False = the fused default-tower kernels; True = the layer-wise general path (modules/generic_net.py)
    generic = False

    def __new__(cls, cfg, policy_input_space=None, critic_input_space=None, act_space=None, share_model: bool = False,
                *args, **kwargs):
        # Configurations outside the fused default tower (hidden_size / layer_N / activation / feature norm / shared
        # model / MultiDiscrete) are served by GenericPPOModule behind the same constructor.
        if cls is PPOModule and act_space is not None:
            from .generic_net import GenericPPOModule, needs_generic

            if needs_generic(cfg, act_space, share_model):  # recurrent or not: GenNet carries the GRU too
                return object.__new__(GenericPPOModule)
        return object.__new__(cls)

    def __init__(self, cfg, policy_input_space, critic_input_space, act_space, share_model: bool = False,
                 device: Union[str, torch.device] = "cuda:0", rank: Optional[int] = None,
                 world_size: Optional[int] = None, model_dict: Optional[Dict[str, Any]] = None):
        # this class = the fused default-tower kernels (feed-forward or GRU); every other feed-forward configuration was
        # routed to GenericPPOModule by __new__, so what is refused here is a RECURRENT policy outside the default tower
        if share_model or cfg.use_share_model:
            raise NotImplementedError("use_share_model with use_recurrent_policy is not built (feed-forward only)")
        check_model_dict(model_dict)
        check_model_dict_roles(model_dict, False)
        for flag in ("use_influence_policy",
                     "use_feature_normalization", "use_policy_vhead", "use_attn", "use_conv1d", "use_amp",
                     "use_deepspeed", "use_single_network"):
            if getattr(cfg, flag, False):
                raise NotImplementedError("cfg.%s=True is not built in the MI355X engine yet" % flag)
        if cfg.layer_N != 1 or cfg.hidden_size != 64 or cfg.activation_id != 1:
            raise NotImplementedError("recurrent towers are built for layer_N=1, hidden_size=64, ReLU (got %d, %d, %d)"
                                      % (cfg.layer_N, cfg.hidden_size, cfg.activation_id))
        # use_naive_recurrent_policy builds the same RNNLayer towers (policy_network.py:82-90); only the update's data
        # generator differs (PPOAlgorithm)
        self.recurrent = bool(cfg.use_recurrent_policy or cfg.use_naive_recurrent_policy)
        if self.recurrent and (cfg.recurrent_N != 1 or getattr(cfg, "rnn_type", "gru") != "gru"):
            raise NotImplementedError("recurrent towers are built for a one-layer GRU (recurrent_N=1, rnn_type=gru)")
        self.cfg = cfg
        self.device = nat.require_gpu(device)
        self.lr, self.critic_lr = cfg.lr, cfg.critic_lr
        self.opti_eps, self.weight_decay = cfg.opti_eps, cfg.weight_decay
        self.act_space = act_space
        self.rank, self.world_size = rank, world_size
        self.share_model = False
        self.policy_input_space, self.critic_input_space = policy_input_space, critic_input_space
        Dp = spaces.obs_dim(spaces.policy_obs_space(policy_input_space))
        Dc = spaces.obs_dim(spaces.critic_obs_space(critic_input_space))
        kind = spaces.kind(act_space)
        if kind == "Discrete":
            head, K = ops.HEAD_CATEGORICAL, int(act_space.n)
        elif kind == "Box":
            head, K = ops.HEAD_GAUSSIAN, int(act_space.shape[0])
        else:
            raise NotImplementedError("action space %s not built (Discrete / Box only)" % kind)
        H = cfg.hidden_size
        # RNG order: policy tower first, then critic (ppo_module.py:58-89 -> rl_module.py:65-87)
        rec = self.recurrent
        tp = _host_init_tower(Dp, H, K, head == ops.HEAD_GAUSSIAN, cfg.gain, cfg.use_orthogonal, cfg.activation_id, rec)
        tc = _host_init_tower(Dc, H, 1, False, 1.0, cfg.use_orthogonal, cfg.activation_id, rec)
        policy = Tower("policy", Dp, K, head, H, self.device, tp, rec)
        critic = Tower("critic", Dc, 1, ops.HEAD_VALUE, H, self.device, tc, rec)
        if cfg.use_valuenorm:
            critic.value_normalizer = ValueNorm(1, device=self.device)
        if cfg.use_popart:
            critic.enable_popart_entries()
        self.models = {"policy": policy, "critic": critic}
        self.optimizers = {"policy": FusedAdam(policy, cfg.lr, cfg.opti_eps, cfg.weight_decay),
                           "critic": FusedAdam(critic, cfg.critic_lr, cfg.opti_eps, cfg.weight_decay)}
        self.act_width = 1 if head == ops.HEAD_CATEGORICAL else K
        self.act_seed = int(cfg.seed)
        self.rng_step = 0  # advances by one per sampled batch; part of the Philox counter
        self.rng_step_dev = None  # optional device-side addend (int64 scalar): set while a rollout hipGraph is captured

    # ------------------------------------------------------------------ helpers
    def _dev(self, x, width: Optional[int] = None) -> Optional[torch.Tensor]:
        if x is None:
            return None
        if isinstance(x, torch.Tensor):
            t = x.to(device=self.device, dtype=torch.float32)
        else:
            t = torch.as_tensor(np.asarray(x), dtype=torch.float32).to(self.device)
        if width is not None:
            t = t.reshape(-1, width)
        return t.contiguous()

    def _forward_rnn(self, critic_obs, obs, h_policy, h_critic, masks, action_masks, deterministic, want_value=True,
                     want_action=True, forced_u=None, out=None, h_out=None):
        """Recurrent get_actions / get_values / act: states [B, (1,) H] in, new states out (``h_out`` = pair of
        destination tensors, e.g. the buffer's next slot; default fresh tensors)."""
        p, c = self.models["policy"], self.models["critic"]
        H = p.net.hidden
        x = self._dev(obs, p.net.obs_dim) if want_action else None
        xc = self._dev(critic_obs, c.net.obs_dim) if want_value else None
        B = (x if x is not None else xc).shape[0]
        mk = self._dev(masks, 1).reshape(B)
        hp_in = self._dev(h_policy, H) if want_action else None
        hc_in = self._dev(h_critic, H) if want_value else None
        am = self._dev(action_masks, p.net.n_out) if (want_action and action_masks is not None and p.net.head_kind ==
                                                       ops.HEAD_CATEGORICAL) else None
        f = lambda *sh: torch.empty(*sh, dtype=torch.float32, device=self.device)
        if out is None:
            values = f(B, 1) if want_value else None
            actions, logp = (f(B, self.act_width), f(B, self.act_width)) if want_action else (None, None)
        else:
            values, actions, logp = out
        hp_out, hc_out = h_out if h_out is not None else (f(B, H) if want_action else None, f(B, H) if want_value else None)
        ops_rnn.rnn_act_step(p.net if want_action else None, p.theta if want_action else None,
                             c.net if want_value else None, c.theta if want_value else None, x, xc, hp_in, hc_in, mk, am,
                             B, deterministic, self.act_seed, 0, self.rng_step, self._dev(forced_u, self.act_width),
                             values, actions, logp, hp_out, hc_out, rng_step_dev=self.rng_step_dev)
        if want_action and not deterministic:
            self.rng_step += 1
        return values, actions, logp, hp_out, hc_out

    def _forward(self, critic_obs, obs, action_masks, deterministic, want_value=True, want_action=True,
                 forced_u=None, out=None):
        p, c = self.models["policy"], self.models["critic"]
        x = self._dev(obs, p.net.obs_dim)
        B = x.shape[0]
        xc = self._dev(critic_obs, c.net.obs_dim) if want_value else None
        am = self._dev(action_masks, p.net.n_out) if (action_masks is not None and p.net.head_kind ==
                                                       ops.HEAD_CATEGORICAL) else None
        if out is None:
            values = torch.empty(B, 1, dtype=torch.float32, device=self.device) if want_value else None
            actions = torch.empty(B, self.act_width, dtype=torch.float32, device=self.device)
            logp = torch.empty(B, self.act_width, dtype=torch.float32, device=self.device)
        else:
            values, actions, logp = out
        ops.act_step(p.net, p.theta, c.net if want_value else None, c.theta if want_value else None, x, xc, am, B,
                     deterministic, self.act_seed, 0, self.rng_step, self._dev(forced_u, self.act_width), values,
                     actions, logp, rng_step_dev=self.rng_step_dev)
        if not deterministic:
            self.rng_step += 1
        return values, actions, logp

    # ------------------------------------------------------------------ reference interface
    def lr_decay(self, episode, episodes):
        """update_linear_schedule (modules/utils/util.py:13-17) on both optimizers (ppo_module.py:91-100)."""
        self.optimizers["policy"].param_groups[0]["lr"] = self.lr - (self.lr * (episode / float(episodes)))
        self.optimizers["critic"].param_groups[0]["lr"] = self.critic_lr - (self.critic_lr * (episode / float(episodes)))

    def get_actions(self, critic_obs, obs, rnn_states_actor, rnn_states_critic, masks, action_masks=None,
                    deterministic=False):
        if self.recurrent:
            v, a, lp, hp, hc = self._forward_rnn(critic_obs, obs, rnn_states_actor, rnn_states_critic, masks,
                                                 action_masks, deterministic)
            return v, a, lp, hp.unsqueeze(1), hc.unsqueeze(1)  # [B, recurrent_N, H] like rnn.py:49
        values, actions, logp = self._forward(critic_obs, obs, action_masks, deterministic)
        return values, actions, logp, rnn_states_actor, rnn_states_critic

    def get_values(self, critic_obs, rnn_states_critic, masks):
        if self.recurrent:
            return self._forward_rnn(critic_obs, None, None, rnn_states_critic, masks, None, True, want_action=False)[0]
        p, c = self.models["policy"], self.models["critic"]
        xc = self._dev(critic_obs, c.net.obs_dim)
        values = torch.empty(xc.shape[0], 1, dtype=torch.float32, device=self.device)
        ops.act_step(p.net, None, c.net, c.theta, None, xc, None, xc.shape[0], True, 0, 0, 0, None, values, None, None)
        return values

    def evaluate_actions(self, critic_obs, obs, rnn_states_actor, rnn_states_critic, action, masks, action_masks=None,
                         active_masks=None, critic_masks_batch=None):
        """Forward-only (no autograd graph: the training path fuses loss + backward in ``orl_ppo_fwd_bwd``).
        Returns ``(values, action_log_probs, dist_entropy, policy_values=None)`` like ppo_module.py:149-193;
        the entropy is the active-mask weighted mean when ``cfg.use_policy_active_masks`` (policy_network.py:199)."""
        if self.recurrent:
            return self._evaluate_actions_rnn(critic_obs, obs, rnn_states_actor, rnn_states_critic, action, masks,
                                              action_masks, active_masks, critic_masks_batch)
        p, c = self.models["policy"], self.models["critic"]
        x = self._dev(obs, p.net.obs_dim)
        xc = self._dev(critic_obs, c.net.obs_dim)
        B = x.shape[0]
        act = self._dev(action, self.act_width)
        am = self._dev(action_masks, p.net.n_out) if (action_masks is not None and p.net.head_kind ==
                                                       ops.HEAD_CATEGORICAL) else None
        active = None
        if active_masks is not None and self.cfg.use_policy_active_masks:
            active = self._dev(active_masks, 1)
        f = lambda *s: torch.empty(*s, dtype=torch.float32, device=self.device)
        values, logp, ent_rows, ent = f(B, 1), f(B, self.act_width), f(B), f(1)
        ops.evaluate_actions(p.net, p.theta, c.net, c.theta, x, xc, act, am, active, B, values, logp, ent_rows, ent)
        return values, logp, ent[0], None

    @torch.no_grad()
    def _evaluate_actions_rnn(self, critic_obs, obs, h_policy, h_critic, action, masks, action_masks, active_masks,
                              critic_masks=None, critic_states_rows=None):
        """Recurrent evaluate_actions (policy_network.py:164-203 + RNNLayer.forward, rnn.py:39-99), forward only:
        the rows are ``L`` steps of ``N`` sequences flattened as [L*N, ...] (the layout of recurrent_generator,
        replay_data.py:1207-1256), the states [N, (1,) H] enter step 0; every step multiplies the carried state by that
        step's mask.  ``critic_*`` may describe FEWER sequences than the policy's (the joint-action loss evaluates the
        critic on agent 0 only): sizes are taken from the state tensors.  One orl_rnn_eval_step per step and tower."""
        p, c = self.models["policy"], self.models["critic"]
        H = p.net.hidden
        x, act = self._dev(obs, p.net.obs_dim), self._dev(action, self.act_width)
        hp = self._dev(h_policy, H)
        Np, Bp = hp.shape[0], x.shape[0]
        assert Bp % Np == 0, "rows must be L steps of the %d policy sequences" % Np
        L = Bp // Np
        mk = self._dev(masks, 1).reshape(-1)
        am = self._dev(action_masks, p.net.n_out) if (action_masks is not None and p.net.head_kind ==
                                                       ops.HEAD_CATEGORICAL) else None
        f = lambda *sh: torch.empty(*sh, dtype=torch.float32, device=self.device)
        logp, ent = f(Bp, self.act_width), f(Bp)
        hp = hp.clone()
        for s in range(L):
            r = slice(s * Np, (s + 1) * Np)
            ops_rnn.rnn_eval_step(p.net, p.theta, None, None, x[r], None, hp, None, mk[r], None if am is None else am[r],
                                  act[r], Np, None, logp[r], ent[r], hp, None)
        values = None
        if critic_obs is not None:
            xc = self._dev(critic_obs, c.net.obs_dim)
            hc = self._dev(h_critic, H).clone()
            Nc = hc.shape[0]
            assert xc.shape[0] == L * Nc, "critic rows must be the same L steps of its %d sequences" % Nc
            mkc = mk if critic_masks is None else self._dev(critic_masks, 1).reshape(-1)
            values = f(L * Nc, 1)
            for s in range(L):
                r = slice(s * Nc, (s + 1) * Nc)
                ops_rnn.rnn_eval_step(None, None, c.net, c.theta, None, xc[r], None, hc, mkc[r], None, None, Nc, values[r],
                                      None, None, None, hc)
        if active_masks is not None and self.cfg.use_policy_active_masks:
            w = self._dev(active_masks, 1).reshape(-1)
            dist_entropy = (ent * w).sum() / w.sum()
        else:
            dist_entropy = ent.mean() / (p.net.n_out if p.net.head_kind == ops.HEAD_GAUSSIAN else 1)
        return values, logp, dist_entropy, None

    def act(self, obs, rnn_states_actor, masks, action_masks=None, deterministic=False):
        if self.recurrent:
            _, actions, _, hp, _ = self._forward_rnn(None, obs, rnn_states_actor, None, masks, action_masks,
                                                     deterministic, want_value=False)
            return actions, hp.unsqueeze(1)
        _, actions, _ = self._forward(None, obs, action_masks, deterministic, want_value=False)
        return actions, rnn_states_actor

    def get_critic_value_normalizer(self):
        return self.models["critic"].value_normalizer

    @staticmethod
    def init_rnn_states(rollout_num: int, agent_num: int, rnn_layers: int, hidden_size: int):
        masks = np.ones((rollout_num * agent_num, 1), dtype=np.float32)
        rnn_state = np.zeros((rollout_num * agent_num, rnn_layers, hidden_size))
        return rnn_state, masks

    # ------------------------------------------------------------------ persistence (rl_module.py:155-192)
    def load_policy(self, model_path: str) -> None:
        sd = torch.load(str(model_path), map_location="cpu")
        self.models["policy"].load_state_dict(sd)

    def restore(self, model_dir: str) -> None:
        for name, m in self.models.items():
            m.load_state_dict(torch.load("%s/%s.pt" % (model_dir, name), map_location="cpu"))

    def save(self, save_dir: str) -> None:
        import os

        os.makedirs(save_dir, exist_ok=True)
        for name, m in self.models.items():
            torch.save({k: v.detach().cpu() for k, v in m.state_dict().items()}, "%s/%s.pt" % (save_dir, name))
