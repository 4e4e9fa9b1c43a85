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
