"""General towers: any ``hidden_size`` / ``layer_N`` / ``activation_id`` / ``use_feature_normalization``, the shared
``PolicyValueNetwork`` (``use_share_model``) and MultiDiscrete action heads.

The fused MFMA kernels (``orl_act_step`` / ``orl_ppo_fwd_bwd``) are built for the reference's DEFAULT tower.  Anything
else ``MLPBase`` / ``MLPLayer`` (openrl/modules/networks/utils/mlp.py:8-46,100-180), ``PolicyValueNetwork``
(openrl/modules/networks/policy_value_network.py:34-230) and ``ACTLayer``'s MultiDiscrete branch
(openrl/modules/networks/utils/act.py:26-34,60-72,136-151) describe runs here layer by layer on the HIP primitives of
``ops_gen`` (fp32 MFMA GEMM, fused bias/activation/LayerNorm rows, loss, Adam) with the activations in HBM.  The layer
loop is host code - the place the reference has it (``nn.Module.forward``); there is still no torch.nn module, no
autograd and no CPU path.

Parameters are ONE flat float32 vector per network.  ``GenNet.entries`` lists (state_dict key, shape, offset) in the
reference's registration order (so checkpoints and the golden vectors of ``oracle/gen_golden.py`` map one to one),
including the quirks that matter for parity: ``MLPLayer.fc_h`` is registered next to its clones ``fc2`` and never
used in ``forward`` (its parameters receive no gradient and never move), and the shared model's ``critic_obs_prep``
is the same module object as ``obs_prep``.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, List, Optional

import numpy as np
import torch

from .. import _native as nat
from .. import ops, ops_gen, spaces
from .ppo_module import FusedAdam, PPOModule, ValueNorm

GAINS = ["tanh", "relu", "leaky_relu", "selu"]  # calculate_gain names per activation_id (mlp.py:15-17)


def needs_generic(cfg, act_space, share_model: bool) -> bool:
    """Does this configuration fall outside the fused default-tower kernels?"""
    kind = spaces.kind(act_space)
    stacked = bool(cfg.use_recurrent_policy or cfg.use_naive_recurrent_policy) and (
        cfg.recurrent_N != 1 or getattr(cfg, "rnn_type", "gru") != "gru")  # GRU stacks and LSTMs: the general towers
    return bool(share_model or cfg.use_share_model or cfg.layer_N != 1 or cfg.hidden_size != 64
                or cfg.activation_id != 1 or cfg.use_feature_normalization or kind in ("MultiDiscrete", "Tuple") or stacked)


def _act_head(act_space):
    kind = spaces.kind(act_space)
    if kind == "Discrete":
        return ops_gen.HEAD_CATEGORICAL, [int(act_space.n)]
    if kind == "Box":
        return ops_gen.HEAD_GAUSSIAN, [int(act_space.shape[0])]
    if kind == "MultiDiscrete":
        nvec = [int(h - l + 1) for h, l in zip(np.asarray(act_space.high).reshape(-1), np.asarray(act_space.low).reshape(-1))]
        return ops_gen.HEAD_MULTI_DISCRETE, nvec
    if kind == "Tuple":  # the reference's mixed branch: (Box(cd), Discrete(n)) - act.py:33-43
        if len(act_space) != 2 or spaces.kind(act_space[0]) != "Box" or spaces.kind(act_space[1]) != "Discrete":
            raise NotImplementedError("Tuple action spaces: (Box, Discrete) only - the reference's mixed ACTLayer branch")
        return ops_gen.HEAD_MIXED, [int(act_space[0].shape[0]), int(act_space[1].n)]
    raise NotImplementedError("action space %s not built (Discrete / Box / MultiDiscrete / Tuple(Box, Discrete))" % kind)


class GenNet:
    """One reference network (``PolicyNetwork`` / ``ValueNetwork`` / ``PolicyValueNetwork``) as a flat vector."""

    def __init__(self, role: str, cfg, obs_dim: int, act_space, device, recurrent: bool = False) -> None:
        assert role in ("policy", "critic", "model")
        self.role, self.device, self.recurrent = role, device, bool(recurrent)
        self.H = H = int(cfg.hidden_size)
        if H % 4 or H > 512:
            raise NotImplementedError("hidden_size %d: the general towers take multiples of 4 up to 512" % H)
        self.D = int(obs_dim)
        self.act_id = int(cfg.activation_id)
        if not 0 <= self.act_id <= 3:
            raise NotImplementedError("activation_id %d" % self.act_id)
        layer_N = 0 if getattr(cfg, "use_single_network", False) else int(cfg.layer_N)
        self.entries: List = []   # (key, shape, offset) in the reference's registration order
        self.layers: List[Dict] = []
        self.ctor: List = []      # construction order of the Linear layers: ("layer", L) / ("dead", fc_h, [clones])
        self.fn: Optional[Dict] = None
        self._o = 0
        base = "obs_prep" if role == "model" else "base"
        if cfg.use_feature_normalization:
            self.fn = dict(g=self._add(base + ".feature_norm.weight", (self.D,)),
                           be=self._add(base + ".feature_norm.bias", (self.D,)), dim=self.D)
        self._mlp_layer(base + ".mlp", self.D, H, layer_N)
        if role == "model":  # self.common = MLPLayer(H, H, layer_N=0, ...)  (policy_value_network.py:78-84)
            self._mlp_layer("common", H, H, 0)
        self.rnn: Optional[Dict] = None
        self.recurrent_N, self.cell, self.G, self.state_w = 1, "gru", 3, H
        if self.recurrent:  # RNNLayer (networks/utils/rnn.py:5-27): nn.GRU(H, H, num_layers=1) + LayerNorm(H)
            self.recurrent_N = rN = int(cfg.recurrent_N)
            # PolicyValueNetwork builds its RNNLayer without rnn_type: always a GRU (policy_value_network.py:85-91)
            self.cell = "gru" if role == "model" else str(getattr(cfg, "rnn_type", "gru"))
            assert self.cell in ("gru", "lstm")
            self.G = G = 3 if self.cell == "gru" else 4          # gate blocks of the projections
            self.state_w = H if self.cell == "gru" else 2 * H   # stored state per layer: h, or [h | c] (rnn.py:31-37)
            layers = []
            for k in range(rN):  # nn.GRU / nn.LSTM register weight_ih, weight_hh, bias_ih, bias_hh layer by layer
                layers.append(dict(Wih=self._add("rnn.rnn.weight_ih_l%d" % k, (G * H, H)),
                                   Whh=self._add("rnn.rnn.weight_hh_l%d" % k, (G * H, H)),
                                   bih=self._add("rnn.rnn.bias_ih_l%d" % k, (G * H,)),
                                   bhh=self._add("rnn.rnn.bias_hh_l%d" % k, (G * H,))))
            self.rnn = dict(layers=layers, g=self._add("rnn.norm.weight", (H,)), be=self._add("rnn.norm.bias", (H,)))
        self.heads: Dict[str, Dict] = OrderedDict()
        if role in ("critic", "model"):
            self.heads["v_out"] = dict(W=self._add("v_out.weight", (1, H)), b=self._add("v_out.bias", (1,)), n=1,
                                       gain=1.0, parts=[1])
        if role in ("policy", "model"):
            kind, nvec = _act_head(act_space)
            n = sum(nvec)
            oW, ob = self._o, self._o + n * H
            self._o += n * H + n
            if kind == ops_gen.HEAD_MULTI_DISCRETE:
                r = 0
                for i, k in enumerate(nvec):  # act.action_outs.{i}.linear.{weight,bias}: interleaved in the reference
                    self.entries.append(("act.action_outs.%d.linear.weight" % i, (k, H), oW + r * H))
                    self.entries.append(("act.action_outs.%d.linear.bias" % i, (k,), ob + r))
                    r += k
            elif kind == ops_gen.HEAD_GAUSSIAN:
                self.entries.append(("act.action_out.fc_mean.weight", (n, H), oW))
                self.entries.append(("act.action_out.fc_mean.bias", (n,), ob))
            elif kind == ops_gen.HEAD_MIXED:  # ModuleList([DiagGaussian, Categorical]) (act.py:37-42), registration order
                cd, nc = nvec
                self.entries.append(("act.action_outs.0.fc_mean.weight", (cd, H), oW))
                self.entries.append(("act.action_outs.0.fc_mean.bias", (cd,), ob))
                ols = self._add("act.action_outs.0.logstd._bias", (cd, 1))
                self.entries.append(("act.action_outs.1.linear.weight", (nc, H), oW + cd * H))
                self.entries.append(("act.action_outs.1.linear.bias", (nc,), ob + cd))
            else:
                self.entries.append(("act.action_out.linear.weight", (n, H), oW))
                self.entries.append(("act.action_out.linear.bias", (n,), ob))
            self.heads["act"] = dict(W=oW, b=ob, n=n, gain=float(cfg.gain), parts=nvec, kind=kind)
            if kind == ops_gen.HEAD_GAUSSIAN:
                self.heads["act"]["logstd"] = self._add("act.action_out.logstd._bias", (n, 1))
                self.heads["act"]["n_ls"] = n
            elif kind == ops_gen.HEAD_MIXED:
                self.heads["act"]["logstd"], self.heads["act"]["n_ls"] = ols, nvec[0]
            self.head_desc = ops_gen.head_desc(kind, n, nvec if kind in (ops_gen.HEAD_MULTI_DISCRETE, ops_gen.HEAD_MIXED)
                                               else None)
            self.act_kind, self.nvec = kind, nvec
            self.act_width = (1 if kind == ops_gen.HEAD_CATEGORICAL else len(nvec) if kind == ops_gen.HEAD_MULTI_DISCRETE
                              else nvec[0] + 1 if kind == ops_gen.HEAD_MIXED else n)
        self.n_params = self._o
        self.theta = torch.zeros(self.n_params, dtype=torch.float32, device=device)
        self.grad = torch.zeros_like(self.theta)
        self.training = False
        self.value_normalizer: Optional[ValueNorm] = None
        self.max_width = max([self.D, H] + [h["n"] for h in self.heads.values()] + ([self.G * H] if self.recurrent else []))

    # ------------------------------------------------------------------ layout
    def _add(self, key: str, shape) -> int:
        off = self._o
        self.entries.append((key, tuple(shape), off))
        self._o += int(np.prod(shape))
        return off

    def _seq(self, prefix: str, n_in: int, n_out: int, act: int, dead: bool = False) -> Dict:
        ln = 2 if act != ops_gen.ACT_NONE else 1  # nn.Sequential(Linear, act, LayerNorm) / (Linear, LayerNorm)
        d = dict(n_in=n_in, n_out=n_out, act=act, dead=dead,
                 W=self._add(prefix + ".0.weight", (n_out, n_in)), b=self._add(prefix + ".0.bias", (n_out,)),
                 g=self._add(prefix + ".%d.weight" % ln, (n_out,)), be=self._add(prefix + ".%d.bias" % ln, (n_out,)))
        return d

    def _mlp_layer(self, prefix: str, n_in: int, H: int, layer_N: int) -> None:
        """MLPLayer.__init__ (mlp.py:8-39): fc1, [fc_h + (layer_N-1) clones in fc2], fc3."""
        fc1 = self._seq(prefix + ".fc1", n_in, H, self.act_id)
        self.layers.append(fc1)
        self.ctor.append(("layer", fc1))
        if layer_N > 1:
            fc_h = self._seq(prefix + ".fc_h", H, H, self.act_id, dead=True)  # registered, never run in forward
            clones = [self._seq(prefix + ".fc2.%d" % i, H, H, self.act_id) for i in range(layer_N - 1)]
            self.layers.extend(clones)
            self.ctor.append(("dead", fc_h, clones))
        fc3 = self._seq(prefix + ".fc3", H, H, ops_gen.ACT_NONE)
        self.layers.append(fc3)
        self.ctor.append(("layer", fc3))

    def mlp_desc(self, head_names):
        """``orl_gen_mlp_desc`` of this network with the given heads (rollout side: the whole tower in one launch), or
        None when a width is outside what ``orl_gen_mlp_fwd`` takes.  Cached per parameter vector."""
        key = (self.theta.data_ptr(), tuple(head_names))
        hit = getattr(self, "_mlp_desc", None)
        if hit is not None and hit[0] == key:
            return hit[1]
        from .. import _native as nat

        ok = (len(self.layers) + len(head_names) <= nat.ORL_GEN_MLP_MAX_LAYERS and 0 <= len(head_names) <= 2
              and all(L["n_out"] <= 256 and L["n_out"] % 4 == 0 for L in self.layers)
              and all(self.heads[h]["n"] <= 256 for h in head_names) and self.D <= 1024)
        d = None
        if ok:
            base = self.theta.data_ptr()
            at = lambda off: base + 4 * off
            d = nat.GenMlpDesc()
            d.n_layers, d.n_heads = len(self.layers), len(head_names)
            if self.fn is not None:
                d.fn_gamma, d.fn_beta = at(self.fn["g"]), at(self.fn["be"])
            for k, L in enumerate(self.layers):
                e = d.layer[k]
                e.W, e.bias, e.gamma, e.beta = at(L["W"]), at(L["b"]), at(L["g"]), at(L["be"])
                e.n_in, e.n_out, e.act = L["n_in"], L["n_out"], L["act"]
            for k, name in enumerate(head_names):
                h, e = self.heads[name], d.layer[len(self.layers) + k]
                e.W, e.bias, e.n_in, e.n_out, e.act = at(h["W"]), at(h["b"]), self.H, h["n"], ops_gen.ACT_NONE
        self._mlp_desc = (key, d)
        return d

    def gt(self, head_names):
        """The cross-layer fused update tower (``FusedTower``: ``orl_gt_prep / _fwd / _bwd``) of this network with the
        given heads, or None when the fused kernels do not take it (recurrent towers, widths other than 64 / 128, too
        many layers or LDS).  Cached per parameter vector."""
        key = (self.theta.data_ptr(), self.grad.data_ptr(), tuple(head_names))
        hit = getattr(self, "_gt", None)
        if hit is None:
            hit = self._gt = {}
        if key not in hit:
            hit[key] = FusedTower.build(self, head_names)
        return hit[key]

    def v(self, off: int, *shape, grad: bool = False) -> torch.Tensor:
        t = self.grad if grad else self.theta
        return t[off:off + int(np.prod(shape))].view(*shape)

    # ------------------------------------------------------------------ reference-order views
    def named_parameters(self):
        return [(k, self.v(o, *s)) for k, s, o in self.entries]

    def parameters(self):
        return [t for _, t in self.named_parameters()]

    def reference_flat(self) -> torch.Tensor:
        """``torch.cat([p.reshape(-1) for p in model.parameters() if p.requires_grad])`` of the reference model."""
        return torch.cat([t.reshape(-1) for _, t in self.named_parameters()])

    def load_reference_flat(self, flat) -> None:
        flat = torch.as_tensor(flat, dtype=torch.float32).reshape(-1)
        assert flat.numel() == self.n_params, (flat.numel(), self.n_params)
        o = 0
        for _, t in self.named_parameters():
            t.copy_(flat[o:o + t.numel()].reshape(t.shape).to(self.device))
            o += t.numel()

    def reference_grad_flat(self) -> torch.Tensor:
        return torch.cat([self.v(o, *s, grad=True).reshape(-1) for _, s, o in self.entries])

    def enable_popart_entries(self):
        """cfg.use_popart: four frozen PopArt entries after v_out.{weight,bias}; the arithmetic is untouched (see
        ``ppo_module.Tower.enable_popart_entries``)."""
        z = lambda v: torch.full((1,), v, dtype=torch.float32, device=self.device)
        self.popart = OrderedDict([("v_out.stddev", z(1.0)), ("v_out.mean", z(0.0)), ("v_out.mean_sq", z(0.0)),
                                   ("v_out.debiasing_term", torch.zeros((), dtype=torch.float32, device=self.device))])

    def state_dict(self):
        sd = OrderedDict()
        if self.value_normalizer is not None:
            sd["value_normalizer.running_mean"] = self.value_normalizer.state[0:1]
            sd["value_normalizer.running_mean_sq"] = self.value_normalizer.state[1:2]
            sd["value_normalizer.debiasing_term"] = self.value_normalizer.state[2]
        named = self.named_parameters()
        for k, t in named:
            if k.startswith("obs_prep."):
                sd[k] = t
        for k, t in named:  # the same module object registered under a second name (policy_value_network.py:75)
            if k.startswith("obs_prep."):
                sd["critic_" + k] = t
        for k, t in named:
            if not k.startswith("obs_prep."):
                sd[k] = t
                if k == "v_out.bias" and getattr(self, "popart", None) is not None:
                    sd.update(self.popart)
        return sd

    def load_state_dict(self, sd):
        for k, t in self.named_parameters():
            t.copy_(torch.as_tensor(sd[k]).to(self.device, torch.float32).reshape(t.shape))
        if getattr(self, "popart", None) is not None:
            for k, t in self.popart.items():
                if k in sd:
                    t.copy_(torch.as_tensor(sd[k]).to(self.device, torch.float32).reshape(t.shape))
        if self.value_normalizer is not None and "value_normalizer.running_mean" in sd:
            self.value_normalizer.state[0] = float(torch.as_tensor(sd["value_normalizer.running_mean"]).reshape(-1)[0])
            self.value_normalizer.state[1] = float(torch.as_tensor(sd["value_normalizer.running_mean_sq"]).reshape(-1)[0])
            self.value_normalizer.state[2] = float(torch.as_tensor(sd["value_normalizer.debiasing_term"]))

    def train(self, mode=True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    # ------------------------------------------------------------------ initial weights (reference RNG order)
    def host_init(self, cfg) -> None:
        """Draw the initial parameters on the HOST with the generator consumption of the reference constructors:
        per Linear the nn.Linear default init, then orthogonal_/xavier_uniform_ with gain (mlp.py:14-39, act.py,
        distributions.py:58-98, value_network.py:103-109, policy_value_network.py:78-113); LayerNorm draws nothing;
        ``fc2`` clones are deepcopies of ``fc_h``."""
        init_method = torch.nn.init.orthogonal_ if cfg.use_orthogonal else torch.nn.init.xavier_uniform_
        gain = torch.nn.init.calculate_gain(GAINS[self.act_id])

        def linear(n_in, n_out, g):
            m = torch.nn.Linear(n_in, n_out)
            init_method(m.weight.data, gain=g)
            return m.weight.data.reshape(-1)

        def fill(layer, w):
            self.v(layer["W"], layer["n_out"] * layer["n_in"]).copy_(w)
            self.v(layer["b"], layer["n_out"]).zero_()
            self.v(layer["g"], layer["n_out"]).fill_(1.0)
            self.v(layer["be"], layer["n_out"]).zero_()

        if self.fn is not None:
            self.v(self.fn["g"], self.D).fill_(1.0)
            self.v(self.fn["be"], self.D).zero_()
        for item in self.ctor:
            if item[0] == "layer":
                fill(item[1], linear(item[1]["n_in"], item[1]["n_out"], gain))
            else:  # fc_h is constructed and initialised once; get_clones deep-copies it (no generator use)
                w = linear(item[1]["n_in"], item[1]["n_out"], gain)
                for L in [item[1]] + item[2]:
                    fill(L, w)
        if self.rnn is not None:  # nn.GRU's default init, then orthogonal_/xavier on both weights, biases 0 (rnn.py:14-26)
            H = self.H
            G = self.G
            gru = (torch.nn.GRU if self.cell == "gru" else torch.nn.LSTM)(H, H, num_layers=self.recurrent_N)
            for k, ly in enumerate(self.rnn["layers"]):  # named_parameters order: w_ih, w_hh (biases draw nothing)
                wih, whh = getattr(gru, "weight_ih_l%d" % k), getattr(gru, "weight_hh_l%d" % k)
                init_method(wih.data)
                init_method(whh.data)
                self.v(ly["Wih"], G * H * H).copy_(wih.data.reshape(-1))
                self.v(ly["Whh"], G * H * H).copy_(whh.data.reshape(-1))
                self.v(ly["bih"], G * H).zero_()
                self.v(ly["bhh"], G * H).zero_()
            self.v(self.rnn["g"], H).fill_(1.0)
            self.v(self.rnn["be"], H).zero_()
        for name, h in self.heads.items():
            if name == "v_out":
                self.v(h["W"], self.H).copy_(linear(self.H, 1, h["gain"]))
                self.v(h["b"], 1).zero_()
            else:
                r = 0
                for k in h["parts"]:
                    self.v(h["W"] + r * self.H, k * self.H).copy_(linear(self.H, k, h["gain"]))
                    r += k
                self.v(h["b"], h["n"]).zero_()
                if "logstd" in h:
                    self.v(h["logstd"], h["n_ls"]).zero_()


class FusedTower:
    """A feed-forward ``GenNet`` as the cross-layer fused kernels see it (csrc/orl_gen_tower.h): the descriptor, the folded /
    split weight image ``orl_gt_prep`` rebuilds after every optimiser step, and the scratch of the backward's per-workgroup
    gradient sums.  ``forward`` = the trunk and head(s) of the rows in one launch, ``backward`` = recomputed forward +
    every parameter gradient written into ``net.grad``."""

    @staticmethod
    def build(net: "GenNet", head_names):
        if net.recurrent or not 1 <= len(head_names) <= 2 or len(net.layers) > nat.ORL_GT_MAX_LAYERS:
            return None
        if any(L["n_out"] != net.H or (k > 0 and L["n_in"] != net.H) for k, L in enumerate(net.layers)):
            return None
        d = nat.GtDesc()
        d.theta = net.theta.data_ptr()
        d.D, d.H, d.n_layers, d.n_heads = net.D, net.H, len(net.layers), len(head_names)
        d.o_fn_g, d.o_fn_be = (net.fn["g"], net.fn["be"]) if net.fn is not None else (-1, -1)
        for k, L in enumerate(net.layers):
            d.oW[k], d.ob[k], d.og[k], d.obe[k], d.act[k] = L["W"], L["b"], L["g"], L["be"], L["act"]
        for k, name in enumerate(head_names):
            h = net.heads[name]
            d.head_oW[k], d.head_ob[k], d.head_n[k] = h["W"], h["b"], h["n"]
        if not ops_gen.gt_supported(d):
            return None
        return FusedTower(net, d, head_names)

    def __init__(self, net: "GenNet", desc, head_names) -> None:
        self.net, self.desc, self.head_names = net, desc, tuple(head_names)
        n_img, n_raw = ops_gen.gt_sizes(desc)
        dev = net.device
        self.image = torch.zeros(n_img, dtype=torch.float32, device=dev)
        self.raw = torch.empty(n_raw, dtype=torch.float32, device=dev)
        self.partials = torch.empty(256 * (n_raw + 24), dtype=torch.float32, device=dev)  # one row per workgroup
        self.sums = torch.zeros(24, dtype=torch.float32, device=dev)  # loss / logging sums of ``train``

    def prep(self) -> None:
        ops_gen.gt_prep(self.desc, self.image)

    def zero_grad_once(self) -> None:
        """``orl_gt_bwd`` / ``orl_gt_train`` WRITE every gradient the reference trains; what they never touch (the dead
        ``fc_h`` block) only has to be zero once per gradient vector."""
        key = self.net.grad.data_ptr()
        if getattr(self, "_zeroed", None) != key:
            self.net.grad.zero_()
            self._zeroed = key

    def forward(self, x: torch.Tensor, col0: int, idx, mb: int, out0: torch.Tensor, out1=None) -> None:
        ops_gen.gt_fwd(self.desc, self.image, x, col0, idx, mb, out0, out1)

    def backward(self, x: torch.Tensor, col0: int, idx, mb: int, dh0: torch.Tensor, dh1=None) -> None:
        ops_gen.gt_bwd(self.desc, self.image, x, col0, idx, mb, dh0, dh1, self.partials, self.raw, self.net.grad)

    def train(self, records: torch.Tensor, col0: int, idx, mb: int, head_desc, logstd, Dp: int, Dc: int, a_w: int, K: int,
              den: torch.Tensor, vn_state, hp, policy_grad: bool = True) -> torch.Tensor:
        """The minibatch's whole update of this tower in one launch (``orl_gt_train``): forward, PPO policy / value loss on
        the head outputs, backward into ``net.grad``.  Returns the 24 loss / logging sums: [0:20] what
        ``orl_gen_policy_loss``'s partials reduce to, [20] the value-loss sum."""
        L = nat.GtLoss()
        names = self.head_names
        L.policy_head = names.index("act") if "act" in names else -1
        L.value_head = names.index("v_out") if "v_out" in names else -1
        L.policy_grad = 1 if policy_grad else 0
        if L.policy_head >= 0:
            L.head = head_desc
            L.logstd = logstd.data_ptr() if logstd is not None else None
        L.den = den.data_ptr()
        L.vn_state = vn_state.data_ptr() if vn_state is not None else None
        L.hp = hp
        L.Dp, L.Dc, L.a_w, L.K = Dp, Dc, a_w, K
        ops_gen.gt_train(self.desc, self.image, records, col0, idx, mb, L, self.partials, self.raw, self.net.grad, self.sums)
        return self.sums


class GenWorkspace:
    """Activation / gradient buffers of one trunk pass over up to ``rows`` rows (allocated once, reused).  Per layer the
    forward keeps ``a`` (post-activation, pre-LayerNorm), the row statistics (mean, rstd) and ``y`` - what
    ``orl_gen_layer_bwd`` and the next layer's weight gradient read."""

    def __init__(self, net: GenNet, rows: int, training: bool, heads_only: bool = False) -> None:
        """``heads_only``: the fused towers' workspace - head outputs / gradients and the loss scratch, no per-layer arrays."""
        dev, f = net.device, torch.float32
        e = lambda *s: torch.empty(*s, dtype=f, device=dev)
        self.rows = rows
        self.heads_only = heads_only
        W = net.max_width
        self.head_out = {k: e(rows, h["n"]) for k, h in net.heads.items()}
        if not heads_only:
            self.x0 = e(rows, net.D)
            self.fn = dict(xhat=e(rows, net.D), rstd=e(rows), y=e(rows, net.D)) if net.fn is not None else None
            self.layers = []
            for L in net.layers:
                self.layers.append(dict(a=e(rows, L["n_out"]) if training else None,
                                        stats=e(rows, 2) if training else None, y=e(rows, L["n_out"])))
        if training:
            self.dhead = {k: e(rows, h["n"]) for k, h in net.heads.items()}
            if not heads_only:
                self.da, self.db2, self.dz, self.dfeat = e(rows, W), e(rows, W), e(rows, W), e(rows, W)
                self.col_partials = e(ops_gen.MAX_BLOCKS * 3 * W)
                self.wgrad_partials = e(max(4 * W * W, min(512 * W * W, 1 << 24)))
            self.loss_partials = e(ops_gen.MAX_BLOCKS * 20)
            self.loss_sums = {"act": e(20), "v_out": e(20)}

    def v(self, t: torch.Tensor, rows: int, width: int) -> torch.Tensor:
        return t.view(-1)[:rows * width].view(rows, width)


def trunk_forward(net: GenNet, ws: GenWorkspace, x: torch.Tensor, save: bool) -> torch.Tensor:
    """MLPBase.forward (+ the shared model's ``common`` layers): features [B, H].  One launch per layer
    (``orl_gen_layer_fwd``: Linear + bias + activation + LayerNorm on the MFMA accumulators)."""
    B = x.shape[0]
    if net.fn is not None:
        f = ws.fn
        y = ws.v(f["y"], B, net.D)
        ops_gen.row_fwd(x, None, ops_gen.ACT_NONE, net.v(net.fn["g"], net.D), net.v(net.fn["be"], net.D), None,
                        ws.v(f["xhat"], B, net.D) if save else None, f["rstd"][:B] if save else None, y)
        x = y
    for L, w in zip(net.layers, ws.layers):
        n_in, n_out = L["n_in"], L["n_out"]
        y = ws.v(w["y"], B, n_out)
        ops_gen.layer_fwd(x, net.v(L["W"], n_out, n_in), net.v(L["b"], n_out), L["act"], net.v(L["g"], n_out),
                          net.v(L["be"], n_out), ws.v(w["a"], B, n_out) if save else None,
                          ws.v(w["stats"], B, 2) if save else None, y)
        w["x_in"] = x
        x = y
    return x


def head_forward(net: GenNet, ws: GenWorkspace, name: str, feats: torch.Tensor) -> torch.Tensor:
    h = net.heads[name]
    B, n = feats.shape[0], h["n"]
    out = ws.v(ws.head_out[name], B, n)
    ops_gen.layer_fwd(feats, net.v(h["W"], n, net.H), net.v(h["b"], n), ops_gen.ACT_NONE, None, None, None, None, out)
    return out


def head_backward(net: GenNet, ws: GenWorkspace, name: str, feats: torch.Tensor, dout: torch.Tensor,
                  dfeat: torch.Tensor, accumulate: bool) -> None:
    """nn.Linear backward of one head: dW, db into ``net.grad``; dfeat (+)= dout @ W."""
    h = net.heads[name]
    B, n, H = feats.shape[0], h["n"], net.H
    ops_gen.wgrad(dout, feats, net.v(h["W"], n, H, grad=True), ws.wgrad_partials)
    nb = ops_gen.layer_bwd(dout, None, None, None, ops_gen.ACT_NONE, None, None, None, ws.col_partials)
    ops_gen.colsum(ws.col_partials, nb, [(None, n), (None, n), (net.v(h["b"], n, grad=True), n)])
    if accumulate:
        tmp = ws.v(ws.dz, B, H)
        ops_gen.linear_dgrad(dout, net.v(h["W"], n, H), tmp)
        ops_gen.vec_add(dfeat, tmp)
    else:
        ops_gen.linear_dgrad(dout, net.v(h["W"], n, H), dfeat)


def trunk_backward(net: GenNet, ws: GenWorkspace, dfeat: torch.Tensor) -> None:
    """Backward of ``trunk_forward``: every trunk parameter's gradient is WRITTEN into ``net.grad``.  Per layer:
    ``orl_gen_layer_bwd`` (LayerNorm / activation backward + the input gradient of square layers), one column-sum
    launch for d gamma / d beta / d bias, ``orl_gen_wgrad``."""
    B = dfeat.shape[0]
    dy = dfeat
    bufs = [ws.da, ws.db2]
    for k in range(len(net.layers) - 1, -1, -1):
        L, w = net.layers[k], ws.layers[k]
        n_in, n_out = L["n_in"], L["n_out"]
        dz = ws.v(ws.dz, B, n_out)
        want_dx = k > 0 or net.fn is not None
        fused_dx = want_dx and n_in == n_out
        dx = ws.v(bufs[k & 1], B, n_in) if want_dx else None
        Wm = net.v(L["W"], n_out, n_in)
        nb = ops_gen.layer_bwd(dy, ws.v(w["a"], B, n_out), ws.v(w["stats"], B, 2), net.v(L["g"], n_out), L["act"],
                               Wm if fused_dx else None, dz, dx if fused_dx else None, ws.col_partials)
        ops_gen.colsum(ws.col_partials, nb, [(net.v(L["g"], n_out, grad=True), n_out),
                                             (net.v(L["be"], n_out, grad=True), n_out),
                                             (net.v(L["b"], n_out, grad=True), n_out)])
        ops_gen.wgrad(dz, w["x_in"], net.v(L["W"], n_out, n_in, grad=True), ws.wgrad_partials)
        if want_dx:
            if not fused_dx:
                ops_gen.linear_dgrad(dz, Wm, dx)
            dy = dx
    if net.fn is not None:
        f = ws.fn
        nb = ops_gen.row_bwd(dy, net.v(net.fn["g"], net.D), ws.v(f["xhat"], B, net.D), f["rstd"][:B], None,
                             ops_gen.ACT_NONE, None, ws.col_partials)
        ops_gen.colsum(ws.col_partials, nb, [(net.v(net.fn["g"], net.D, grad=True), net.D),
                                             (net.v(net.fn["be"], net.D, grad=True), net.D), (None, net.D)])


class GruWorkspace:
    """Buffers of the GRU / LSTM stack between trunk and head over ``L`` steps of ``N`` sequences (rows ordered [L, N], the
    layout of ``recurrent_generator``): per layer the projections, masked inputs and gate values; the LayerNorm after the
    last layer."""

    def __init__(self, net: GenNet, L: int, N: int, training: bool) -> None:
        H, G, dev = net.H, net.G, net.device
        lstm = net.cell == "lstm"
        e = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        self.L, self.N = L, N
        self.layers = []
        for _ in net.rnn["layers"]:
            d = dict(gi=e(L * N, G * H), gh=e(L * N, G * H), h_in=e(L * N, H), h=e(L * N, H))
            if lstm:
                d.update(c_in=e(L * N, H), c=e(L * N, H))
            if training:
                d.update(save=e(L * N, (5 if lstm else 4) * H), dgi=e(L * N, G * H), dx=e(L * N, H))
                d["dgh"] = d["dgi"] if lstm else e(L * N, G * H)  # an LSTM's two projections share one gradient
            self.layers.append(d)
        self.h = self.layers[-1]["h"]      # output states of the last layer, every step
        self.y = e(L * N, H)                # LayerNorm(h): what the head reads
        self.h_last = e(N, len(self.layers), net.state_w)  # states after the last step, [N, recurrent_N, H or 2H]
        self.h0 = e(N, len(self.layers) * net.state_w)     # gathered initial states / masks of an update minibatch
        self.mrows = e(L * N)
        if training:
            self.xhat, self.rstd = e(L * N, H), e(L * N)
            self.dh, self.dh_dir, self.tmp = e(L * N, H), e(N, H), e(N, H)
            if lstm:
                self.dc = [e(N, H), e(N, H)]
            self.col_partials = e(ops_gen.MAX_BLOCKS * 3 * G * H)
            self.wgrad_partials = e(max(4 * G * H * H, min(256 * G * H * H, 1 << 24)))


def gru_forward(net: GenNet, gw: GruWorkspace, feats: torch.Tensor, h0: torch.Tensor, masks: torch.Tensor, L: int, N: int,
                save: bool) -> torch.Tensor:
    """RNNLayer.forward (rnn.py:39-99) on [L * N, H] features: per layer h_t = cell(x_t, state_{t-1} * mask_t) with the
    layer below's output sequence as input, y = LayerNorm(h of the last layer).  ``h0``: [N, recurrent_N, state_w]
    (state_w = H for a GRU, 2 H = [h | c] for an LSTM).  Per layer ONE projection GEMM for all steps' inputs; per step one
    recurrent GEMM + one gate launch.  Returns y [L * N, H]; ``gw.h_last`` holds the states after the last step."""
    H, G, r = net.H, net.G, net.rnn
    lstm = net.cell == "lstm"
    B = L * N
    h0 = h0.reshape(N, len(r["layers"]), net.state_w)
    x = feats
    for k, (ly, w) in enumerate(zip(r["layers"], gw.layers)):
        Wih, Whh = net.v(ly["Wih"], G * H, H), net.v(ly["Whh"], G * H, H)
        bih, bhh = net.v(ly["bih"], G * H), net.v(ly["bhh"], G * H)
        ops_gen.layer_fwd(x, Wih, bih, ops_gen.ACT_NONE, None, None, None, None, w["gi"][:B])
        ops_gen.row_affine(h0[:, k, :H].contiguous(), None, masks[:N], None, w["h_in"][:N])
        if lstm:
            ops_gen.row_affine(h0[:, k, H:].contiguous(), None, masks[:N], None, w["c_in"][:N])
        for t in range(L):
            s, nx = slice(t * N, (t + 1) * N), slice((t + 1) * N, (t + 2) * N)
            ops_gen.layer_fwd(w["h_in"][s], Whh, bhh, ops_gen.ACT_NONE, None, None, None, None, w["gh"][s])
            last = t == L - 1
            if lstm:
                ops_gen.lstm_gate_fwd(w["gi"][s], w["gh"][s], w["c_in"][s], None if last else masks[nx], w["h"][s], w["c"][s],
                                      None if last else w["h_in"][nx], None if last else w["c_in"][nx],
                                      w["save"][s] if save else None)
            else:
                ops_gen.gru_gate_fwd(w["gi"][s], w["gh"][s], w["h_in"][s], None if last else masks[nx], w["h"][s],
                                     None if last else w["h_in"][nx], w["save"][s] if save else None)
        gw.h_last[:N, k, :H].copy_(w["h"][(L - 1) * N:B])
        if lstm:
            gw.h_last[:N, k, H:].copy_(w["c"][(L - 1) * N:B])
        x = w["h"][:B]
    ops_gen.row_fwd(x, None, ops_gen.ACT_NONE, net.v(r["g"], H), net.v(r["be"], H), None,
                    gw.xhat[:B] if save else None, gw.rstd[:B] if save else None, gw.y[:B])
    return gw.y[:B]


def gru_backward(net: GenNet, gw: GruWorkspace, feats: torch.Tensor, masks: torch.Tensor, dy: torch.Tensor, L: int,
                 N: int) -> torch.Tensor:
    """Back-propagation through ``gru_forward`` (time and depth): writes the stack's and its LayerNorm's gradients into
    ``net.grad`` and returns d feats [L * N, H].  The chunk's initial states are data: no gradient leaves."""
    H, G, r = net.H, net.G, net.rnn
    lstm = net.cell == "lstm"
    B = L * N
    # LayerNorm after the last layer: dy -> dh of that layer (all steps), d gamma / d beta
    nb = ops_gen.row_bwd(dy, net.v(r["g"], H), gw.xhat[:B], gw.rstd[:B], None, ops_gen.ACT_NONE, gw.dh[:B], gw.col_partials)
    ops_gen.colsum(gw.col_partials, nb, [(net.v(r["g"], H, grad=True), H), (net.v(r["be"], H, grad=True), H), (None, H)])
    dh = gw.dh
    for k in range(len(r["layers"]) - 1, -1, -1):
        ly, w = r["layers"][k], gw.layers[k]
        Wih, Whh = net.v(ly["Wih"], G * H, H), net.v(ly["Whh"], G * H, H)
        x_in = feats if k == 0 else gw.layers[k - 1]["h"][:B]
        dc = None  # LSTM: gradient at c_t carried from step t + 1
        for t in range(L - 1, -1, -1):
            s = slice(t * N, (t + 1) * N)
            # dh[s] holds the total gradient at h_t: from above (LayerNorm or the next layer's input gradient) + the carry
            if lstm:
                dc_in = gw.dc[t & 1][:N]
                ops_gen.lstm_gate_bwd(dh[s], dc, w["save"][s], w["c_in"][s], w["dgi"][s], dc_in)
            else:
                ops_gen.gru_gate_bwd(dh[s], w["save"][s], w["h_in"][s], w["dgi"][s], w["dgh"][s], gw.dh_dir)
            if t > 0:  # d h_{t-1} += (dgh W_hh [+ z * dh]) * mask_t;  LSTM: d c_{t-1} = dc_in * mask_t
                p = slice((t - 1) * N, t * N)
                ops_gen.linear_dgrad(w["dgh"][s], Whh, gw.tmp[:N])
                ops_gen.row_affine(gw.tmp[:N], None if lstm else gw.dh_dir[:N], masks[s], dh[p], dh[p])
                if lstm:
                    ops_gen.row_affine(dc_in, None, masks[s], None, dc_in)
                    dc = dc_in
        ops_gen.wgrad(w["dgi"][:B], x_in, net.v(ly["Wih"], G * H, H, grad=True), gw.wgrad_partials)
        ops_gen.wgrad(w["dgh"][:B], w["h_in"][:B], net.v(ly["Whh"], G * H, H, grad=True), gw.wgrad_partials)
        ops_gen.colsum_rows(w["dgi"][:B], net.v(ly["bih"], G * H, grad=True), gw.col_partials)
        ops_gen.colsum_rows(w["dgh"][:B], net.v(ly["bhh"], G * H, grad=True), gw.col_partials)
        ops_gen.linear_dgrad(w["dgi"][:B], Wih, w["dx"][:B])  # gradient at this layer's input sequence
        dh = w["dx"]
    return gw.layers[0]["dx"][:B]


class GenAdam(FusedAdam):
    """Adam state of one general network; the step is ``orl_gen_adam`` (norm, clip, Adam in two launches)."""


class GenericPPOModule(PPOModule):
    """``PPOModule`` for everything outside the fused default tower (see the module docstring)."""

    generic = True

    def __init__(self, cfg, policy_input_space, critic_input_space, act_space, share_model: bool = False,
                 device="cuda:0", rank=None, world_size=None, model_dict=None):
        from .ppo_module import check_model_dict, check_model_dict_roles

        check_model_dict(model_dict)
        check_model_dict_roles(model_dict, bool(share_model or cfg.use_share_model))
        for flag in ("use_influence_policy", "use_policy_vhead", "use_attn", "use_conv1d", "use_amp", "use_deepspeed"):
            if getattr(cfg, flag, False):
                raise NotImplementedError("cfg.%s=True is not built for the general towers" % flag)
        self.cfg = cfg
        self.device = nat.require_gpu(device)
        # use_naive_recurrent_policy builds the same RNNLayer towers (policy_network.py:82-90)
        self.recurrent = bool(cfg.use_recurrent_policy or cfg.use_naive_recurrent_policy)
        if self.recurrent:
            if cfg.recurrent_N < 1 or getattr(cfg, "rnn_type", "gru") not in ("gru", "lstm"):
                raise NotImplementedError("recurrent towers: rnn_type gru or lstm, recurrent_N >= 1")
            if (share_model or cfg.use_share_model) and getattr(cfg, "rnn_type", "gru") != "gru":
                raise NotImplementedError("use_share_model with rnn_type lstm: the reference's shared network is a GRU "
                                          "while PPONet sizes the stored states for an LSTM")
        self.lr, self.critic_lr = cfg.lr, cfg.critic_lr
        self.opti_eps, self.weight_decay = cfg.opti_eps, cfg.weight_decay
        self.act_space = act_space
        self.rank, self.world_size = rank, world_size
        self.share_model = bool(share_model or cfg.use_share_model)
        self.policy_input_space, self.critic_input_space = policy_input_space, critic_input_space
        self.Dp = spaces.obs_dim(spaces.policy_obs_space(policy_input_space))
        self.Dc = spaces.obs_dim(spaces.critic_obs_space(critic_input_space))
        if self.share_model:
            if self.Dc != self.Dp:
                raise NotImplementedError("use_share_model: critic observations go through the policy's obs_prep "
                                          "(policy_value_network.py:75) and must have the same width")
            model = GenNet("model", cfg, self.Dp, act_space, self.device, recurrent=self.recurrent)
            model.host_init(cfg)
            if cfg.use_valuenorm:
                model.value_normalizer = ValueNorm(1, device=self.device)
            if cfg.use_popart:
                model.enable_popart_entries()
            self.models = {"model": model}
            self.optimizers = {"model": GenAdam(model, cfg.lr, cfg.opti_eps, cfg.weight_decay)}
            pol = model
        else:
            policy = GenNet("policy", cfg, self.Dp, act_space, self.device, recurrent=self.recurrent)
            policy.host_init(cfg)  # RNG order: policy first, then critic (rl_module.py:65-87)
            critic = GenNet("critic", cfg, self.Dc, act_space, self.device, recurrent=self.recurrent)
            critic.host_init(cfg)
            if cfg.use_valuenorm:
                critic.value_normalizer = ValueNorm(1, device=self.device)
            if cfg.use_popart:
                critic.enable_popart_entries()
            self.models = {"policy": policy, "critic": critic}
            self.optimizers = {"policy": GenAdam(policy, cfg.lr, cfg.opti_eps, cfg.weight_decay),
                               "critic": GenAdam(critic, cfg.critic_lr, cfg.opti_eps, cfg.weight_decay)}
            pol = policy
        self.act_width = pol.act_width
        self.act_kind = pol.act_kind
        self.n_logits = pol.heads["act"]["n"]
        self.K = self.n_logits if pol.act_kind == ops_gen.HEAD_CATEGORICAL else 0
        self.act_seed = int(cfg.seed)
        self.rng_step = 0
        self.rng_step_dev = None
        self._ws: Dict = {}
        self._ws_retired: List = []  # outgrown workspaces, kept alive for hipGraphs captured while they were current
        self._side_stream = None  # the critic's chain of a recurrent rollout step (_forward_rnn) / of an update
        self.two_stream = True    # False: everything on the caller's stream (the equality test of the two routes)

    # ------------------------------------------------------------------ nets
    @property
    def policy_net(self) -> GenNet:
        return self.models["model" if self.share_model else "policy"]

    @property
    def critic_net(self) -> GenNet:
        return self.models["model" if self.share_model else "critic"]

    def side_stream(self):
        """The second stream the critic's chain runs on beside the policy's (rollout steps of recurrent towers, updates of
        separate networks)."""
        if self._side_stream is None:
            self._side_stream = torch.cuda.Stream(self.device)
        return self._side_stream

    def workspace(self, net: GenNet, rows: int, training: bool, tag: str = "", heads_only: bool = False) -> GenWorkspace:
        key = (id(net), training, tag, heads_only)
        ws = self._ws.get(key)
        if ws is None or ws.rows < rows:
            if ws is not None:  # a captured rollout hipGraph may still point into the old buffers: never free them
                self._ws_retired.append(ws)
            ws = self._ws[key] = GenWorkspace(net, rows, training, heads_only)
        return ws

    def fused_towers(self, one_pass: bool):
        """(policy tower, critic tower) of the cross-layer fused update (``cfg.amd_gen_update = fused``), or None when a
        network is outside what the fused kernels take.  The shared network is ONE tower with both heads and needs the
        one-pass case (critic observations alias the policy's)."""
        if getattr(self.cfg, "amd_gen_update", "fused") != "fused" or self.recurrent:
            return None
        if self.share_model:
            if not one_pass:
                return None
            ft = self.policy_net.gt(("act", "v_out"))
            return None if ft is None else (ft, ft)
        fp, fc = self.policy_net.gt(("act",)), self.critic_net.gt(("v_out",))
        return None if fp is None or fc is None else (fp, fc)

    def gru_workspace(self, net: GenNet, L: int, N: int, training: bool, tag: str = "") -> "GruWorkspace":
        key = ("gru", id(net), training, tag)
        gw = self._ws.get(key)
        if gw is None or gw.L * gw.N < L * N or gw.N < N:
            if gw is not None:
                self._ws_retired.append(gw)
                L, N = max(L, gw.L), max(N, gw.N)
            gw = self._ws[key] = GruWorkspace(net, L, N, training)
        return gw

    # ------------------------------------------------------------------ fused rollout (device envs, feed-forward)
    def fused_rollout_ready(self, data) -> bool:
        """True when ``orl_gen_rollout_fused`` can run this module's rollouts on ``data``: feed-forward towers whose widths
        the one-launch tower takes, single-agent buffers whose critic observations alias the policy's."""
        if self.recurrent or data.num_agents != 1 or data.critic_obs is not data.policy_obs:
            return False
        pn, cn = self.policy_net, self.critic_net
        if pn.mlp_desc(("act", "v_out") if self.share_model else ("act",)) is None:
            return False
        return self.share_model or cn.mlp_desc(("v_out",)) is not None

    @torch.no_grad()
    def rollout_fused(self, data, env, next_value_out) -> None:
        """One whole rollout (onpolicy_driver.py:154-203) in two launches: ``orl_gen_rollout_fused`` (policy tower, sampling,
        env.step, insert for all episode_length steps) and, for separate networks, ONE critic forward over all T + 1
        observation slots - slot T's value is the bootstrap value of ``compute_returns``."""
        pn, cn = self.policy_net, self.critic_net
        T, N = data.episode_length, data.n_rollout_threads
        desc = pn.mlp_desc(("act", "v_out") if self.share_model else ("act",))
        ops_gen.rollout_fused(desc, pn.head_desc, self._logstd(), data.buffer_ptrs(),
                              data.value_preds if self.share_model else None, data.actions, data.action_log_probs,
                              env.env_state, env.ep_stats, env.env_kind, env.episode_limit, env.seed, env.global_step,
                              self.act_seed, self.rng_step, self.act_width, self.device)
        self.rng_step += T
        if not self.share_model:
            rows = (T + 1) * N
            ft = cn.gt(("v_out",)) if getattr(self.cfg, "amd_gen_update", "fused") == "fused" else None
            if ft is not None:  # the update's forward kernel: activations on chip, two workgroups per CU
                ft.prep()
                ft.forward(data.critic_obs.view(rows, self.Dc), 0, None, rows, data.value_preds.view(rows, 1))
            else:
                ops_gen.mlp_fwd(cn.mlp_desc(("v_out",)), data.critic_obs.view(rows, self.Dc), data.value_preds.view(rows, 1),
                                None)
            next_value_out.copy_(data.value_preds[T])

    def _logstd(self):
        h = self.policy_net.heads["act"]
        return self.policy_net.v(h["logstd"], h["n_ls"]) if "logstd" in h else None

    # ------------------------------------------------------------------ rollout side
    @torch.no_grad()
    def _forward(self, critic_obs, obs, action_masks, deterministic, want_value=True, want_action=True, forced_u=None,
                 out=None):
        pn, cn = self.policy_net, self.critic_net
        x = self._dev(obs, self.Dp) if want_action else None
        xc = self._dev(critic_obs, self.Dc) if want_value else None
        B = (x if x is not None else xc).shape[0]
        f = lambda *sh: torch.empty(*sh, dtype=torch.float32, device=self.device)
        if out is None:
            values = f(B, 1) if want_value else None
            actions, logp = (f(B, self.act_width), f(B, self.act_width)) if want_action else (None, None)
        else:
            values, actions, logp = out
        feats_p = None
        shared_once = self.share_model and want_action and want_value and xc.data_ptr() == x.data_ptr()
        if want_action and want_value and values.is_contiguous() and (shared_once or not self.share_model):
            # the whole step - both towers and the sampling - in one launch
            dp = pn.mlp_desc(("act", "v_out") if shared_once else ("act",))
            dc = None if shared_once else cn.mlp_desc(("v_out",))
            if dp is not None and (shared_once or dc is not None):
                am = self._dev(action_masks, self.n_logits) if (action_masks is not None and self.K) else None
                ops_gen.act_step(dp, x, dc, xc, values, pn.head_desc, self._logstd(), am, deterministic, self.act_seed, 0,
                                 self.rng_step, self.rng_step_dev, self._dev(forced_u, self.act_width), self.act_width,
                                 actions, logp)
                if not deterministic:
                    self.rng_step += 1
                return values, actions, logp
        if want_action:
            ws = self.workspace(pn, B, False, "p")
            logits = ws.v(ws.head_out["act"], B, pn.heads["act"]["n"])
            desc = pn.mlp_desc(("act", "v_out") if shared_once else ("act",))
            if desc is not None:  # the whole tower (+ the value head of a shared network) in one launch
                vdst = None
                if shared_once:
                    vdst = values if values.is_contiguous() else ws.v(ws.head_out["v_out"], B, 1)
                ops_gen.mlp_fwd(desc, x, logits, vdst)
                if shared_once and vdst is not values:
                    values.copy_(vdst)
            else:
                feats_p = trunk_forward(pn, ws, x, False)
                logits = head_forward(pn, ws, "act", feats_p)
            am = self._dev(action_masks, self.n_logits) if (action_masks is not None and self.K) else None
            ops_gen.sample(pn.head_desc, logits, self._logstd(), am, B, deterministic, self.act_seed, 0, self.rng_step,
                           self.rng_step_dev, self._dev(forced_u, self.act_width), self.act_width, actions, logp)
            if not deterministic:
                self.rng_step += 1
        if want_value and not (shared_once and feats_p is None):
            if shared_once:
                ws, feats_c = self.workspace(pn, B, False, "p"), feats_p
                v = head_forward(cn, ws, "v_out", feats_c)
            else:
                ws = self.workspace(cn, B, False, "c")
                desc = cn.mlp_desc(("v_out",))
                if desc is not None:
                    v = values if values.is_contiguous() else ws.v(ws.head_out["v_out"], B, 1)
                    ops_gen.mlp_fwd(desc, xc, v)
                else:
                    v = head_forward(cn, ws, "v_out", trunk_forward(cn, ws, xc, False))
            if v is not values:
                values.copy_(v)
        return values, actions, logp

    def _tower_step(self, net: GenNet, tag: str, x, h_in, mk, B: int):
        """Trunk + one GRU step of ``B`` rows: (LayerNorm(h_new) [B, H], h_new [B, H])."""
        ws = self.workspace(net, B, False, tag)
        desc = net.mlp_desc(())
        if desc is not None:  # the whole trunk in one launch
            feats = ws.v(ws.layers[-1]["y"], B, net.H)
            ops_gen.mlp_fwd(desc, x, None, None, feats)
        else:
            feats = trunk_forward(net, ws, x, False)
        gw = self.gru_workspace(net, 1, B, False, tag)
        y = gru_forward(net, gw, feats, h_in, mk, 1, B, False)
        return ws, y, gw.h_last[:B]

    @torch.no_grad()
    def _forward_rnn(self, critic_obs, obs, h_policy, h_critic, masks, action_masks, deterministic, want_value=True,
                     want_action=True, forced_u=None, out=None, h_out=None):
        """Recurrent get_actions / get_values / act (same contract as ``PPOModule._forward_rnn``): states [B, (1,) H] in,
        new states out (``h_out`` = pair of destination tensors, e.g. the buffer's next slot)."""
        pn, cn = self.policy_net, self.critic_net
        H = pn.state_w * pn.recurrent_N  # a row of states: [recurrent_N, H (GRU) or 2 H (LSTM: h | c)] flattened
        x = self._dev(obs, self.Dp) if want_action else None
        xc = self._dev(critic_obs, self.Dc) if want_value else None
        B = (x if x is not None else xc).shape[0]
        mk = self._dev(masks, 1).reshape(B)
        f = lambda *sh: torch.empty(*sh, dtype=torch.float32, device=self.device)
        if out is None:
            values = f(B, 1) if want_value else None
            actions, logp = (f(B, self.act_width), f(B, self.act_width)) if want_action else (None, None)
        else:
            values, actions, logp = out
        hp_out, hc_out = h_out if h_out is not None else (f(B, H) if want_action else None, f(B, H) if want_value else None)
        hp_in = self._dev(h_policy, H) if want_action else None
        hc_in = self._dev(h_critic, H) if want_value else None

        def critic_step():
            ws, y, h_new = self._tower_step(cn, "c", xc, hc_in, mk, B)
            values.copy_(head_forward(cn, ws, "v_out", y))
            hc_out.copy_(h_new.view_as(hc_out))

        # A recurrent step is a chain of ~10 small launches per tower: with separate networks and caller-owned outputs
        # (the driver's buffer slots - nothing is allocated below) the critic's chain runs on a second stream beside
        # the policy's, forked and joined here, also inside a captured rollout graph.
        fork = (want_action and want_value and not self.share_model and out is not None and h_out is not None
                and self.two_stream)
        if fork:
            main = torch.cuda.current_stream(self.device)
            self.side_stream().wait_stream(main)
            with torch.cuda.stream(self._side_stream):
                critic_step()
        if want_action:
            ws, y, h_new = self._tower_step(pn, "p", x, hp_in, mk, B)
            logits = head_forward(pn, ws, "act", y)
            am = self._dev(action_masks, self.n_logits) if (action_masks is not None and self.K) else None
            ops_gen.sample(pn.head_desc, logits, self._logstd(), am, B, deterministic, self.act_seed, 0, self.rng_step,
                           self.rng_step_dev, self._dev(forced_u, self.act_width), self.act_width, actions, logp)
            if not deterministic:
                self.rng_step += 1
            hp_out.copy_(h_new.view_as(hp_out))
        if fork:
            main.wait_stream(self._side_stream)
        elif want_value:
            critic_step()
        return values, actions, logp, hp_out, hc_out

    def get_actions(self, critic_obs, obs, rnn_states_actor, rnn_states_critic, masks, action_masks=None,
                    deterministic=False):
        if self.recurrent:  # states come back as [B, recurrent_N, H] (rnn.py:49)
            v, a, lp, hp, hc = self._forward_rnn(critic_obs, obs, rnn_states_actor, rnn_states_critic, masks, action_masks,
                                                 deterministic)
            rN, Hh = self.policy_net.recurrent_N, self.policy_net.state_w
            return v, a, lp, hp.reshape(-1, rN, Hh), hc.reshape(-1, rN, Hh)
        values, actions, logp = self._forward(critic_obs, obs, action_masks, deterministic)
        return values, actions, logp, rnn_states_actor, rnn_states_critic

    def get_values(self, critic_obs, rnn_states_critic, masks):
        if self.recurrent:
            return self._forward_rnn(critic_obs, None, None, rnn_states_critic, masks, None, True, want_action=False)[0]
        return self._forward(critic_obs, None, None, True, want_action=False)[0]

    def act(self, obs, rnn_states_actor, masks, action_masks=None, deterministic=False):
        if self.recurrent:
            _, actions, _, hp, _ = self._forward_rnn(None, obs, rnn_states_actor, None, masks, action_masks, deterministic,
                                                     want_value=False)
            return actions, hp.reshape(-1, self.policy_net.recurrent_N, self.policy_net.state_w)
        _, actions, _ = self._forward(None, obs, action_masks, deterministic, want_value=False)
        return actions, rnn_states_actor

    def _logp_entropy(self, logits, action, action_masks, active_masks, B: int):
        """ACTLayer.evaluate_actions on dense rows: log-probs [B, a_w] and the (masked) mean entropy."""
        pn = self.policy_net
        a_w, K = self.act_width, self.K
        R = ops.record_width(self.Dp, self.Dc, a_w, K)
        rec = torch.zeros(B, R, dtype=torch.float32, device=self.device)
        o_act = self.Dp + self.Dc
        rec[:, o_act:o_act + a_w] = self._dev(action, a_w)
        if K:
            o_mk = o_act + 2 * a_w + 4
            rec[:, o_mk:o_mk + K] = 1.0 if action_masks is None else self._dev(action_masks, K)
        logp = torch.empty(B, a_w, dtype=torch.float32, device=self.device)
        ent = torch.empty(B, dtype=torch.float32, device=self.device)
        ops_gen.policy_eval(pn.head_desc, logits, self._logstd(), rec, self.Dp, self.Dc, a_w, K, B,
                            ops.make_hparams(self.cfg), logp, ent)
        if active_masks is not None and self.cfg.use_policy_active_masks:  # policy_network.py:196-201
            am = self._dev(active_masks, 1).reshape(-1)
            dist_entropy = (ent * am).sum() / am.sum()
        else:
            dist_entropy = ent.mean() / (self.n_logits if self.act_kind == ops_gen.HEAD_GAUSSIAN else 1)
        return logp, dist_entropy

    @torch.no_grad()
    def evaluate_actions(self, critic_obs, obs, rnn_states_actor, rnn_states_critic, action, masks, action_masks=None,
                         active_masks=None, critic_masks_batch=None):
        """Forward-only ``PPOModule.evaluate_actions`` (ppo_module.py:149-193): values, log-probs, dist_entropy."""
        if self.recurrent:
            return self._evaluate_actions_rnn(critic_obs, obs, rnn_states_actor, rnn_states_critic, action, masks,
                                              action_masks, active_masks, critic_masks_batch)
        pn = self.policy_net
        x = self._dev(obs, self.Dp)
        B = x.shape[0]
        values = self.get_values(critic_obs, None, None)
        ws = self.workspace(pn, B, False, "p")
        logits = head_forward(pn, ws, "act", trunk_forward(pn, ws, x, False))
        logp, dist_entropy = self._logp_entropy(logits, action, action_masks, active_masks, B)
        return values, logp, dist_entropy, None

    @torch.no_grad()
    def _evaluate_actions_rnn(self, critic_obs, obs, h_policy, h_critic, action, masks, action_masks, active_masks,
                              critic_masks=None, critic_states_rows=None):
        """Recurrent evaluate_actions (policy_network.py:164-203 + RNNLayer.forward): the rows are L steps of N sequences
        flattened [L * N, ...] (recurrent_generator's layout), the states [N, (1,) H] enter step 0."""
        pn, cn = self.policy_net, self.critic_net
        H = pn.state_w * pn.recurrent_N
        x = self._dev(obs, self.Dp)
        hp = self._dev(h_policy, H)
        Np, Bp = hp.shape[0], x.shape[0]
        assert Bp % Np == 0, "rows must be L steps of the %d policy sequences" % Np
        L = Bp // Np
        mk = self._dev(masks, 1).reshape(-1)
        ws = self.workspace(pn, Bp, False, "p")
        feats = trunk_forward(pn, ws, x, False)
        y = gru_forward(pn, self.gru_workspace(pn, L, Np, False, "pe"), feats, hp, mk, L, Np, False)
        logits = head_forward(pn, ws, "act", y)
        logp, dist_entropy = self._logp_entropy(logits, action, action_masks, active_masks, Bp)
        values = None
        if critic_obs is not None:
            xc = self._dev(critic_obs, self.Dc)
            hc = self._dev(h_critic, H)
            Nc = hc.shape[0]
            assert xc.shape[0] == L * Nc, "critic rows must be the same L steps of its %d sequences" % Nc
            mkc = mk if critic_masks is None else self._dev(critic_masks, 1).reshape(-1)
            wc = self.workspace(cn, L * Nc, False, "c")
            fc = trunk_forward(cn, wc, xc, False)
            yc = gru_forward(cn, self.gru_workspace(cn, L, Nc, False, "ce"), fc, hc, mkc, L, Nc, False)
            values = head_forward(cn, wc, "v_out", yc).clone()
        return values, logp, dist_entropy, None

    def lr_decay(self, episode, episodes):
        if self.share_model:
            self.optimizers["model"].param_groups[0]["lr"] = self.lr - (self.lr * (episode / float(episodes)))
        else:
            super().lr_decay(episode, episodes)

    def get_critic_value_normalizer(self):
        return self.critic_net.value_normalizer
