"""Cross-layer fused general towers (csrc/orl_gen_tower.{h,hip}: orl_gt_prep / orl_gt_fwd / orl_gt_bwd) against

* a float64 torch.autograd restatement of MLPBase.forward + the Linear heads built from the same flat parameter vector
  (openrl/modules/networks/utils/mlp.py:8-48,100-180; feature LayerNorm, Linear / activation / LayerNorm layers, heads), and
* the layer-wise route (orl_gen_layer_fwd / _bwd / orl_gen_wgrad), which the reference goldens pin,

on rows gathered through a permutation out of wide records, ragged batch sizes included.  The reference-minted goldens
of the general towers (tests/test_generic_gpu.py) run through these kernels whenever the shape is eligible; here the
shapes are chosen to cover every template instance.  Needs a MI355X."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

ACTS = {0: torch.tanh, 1: torch.relu, 2: lambda z: torch.nn.functional.leaky_relu(z, 0.01), 3: torch.nn.functional.elu}


def make_net(role, H, layer_N, act_id, fn, D, act_space):
    from openrl_amd.configs.config import default_cfg
    from openrl_amd.modules.generic_net import GenNet

    argv = ["--hidden_size", str(H), "--layer_N", str(layer_N), "--activation_id", str(act_id),
            "--use_feature_normalization", "true" if fn else "false"]
    cfg = default_cfg(argv)
    net = GenNet(role, cfg, D, act_space, DEV)
    torch.manual_seed(H + 7 * layer_N + act_id + D)
    net.host_init(cfg)
    # move every LayerNorm affine and bias off its initial 1 / 0 so the folds are exercised
    g = torch.Generator().manual_seed(1000 + D)
    for key, shape, off in net.entries:
        n = int(np.prod(shape))
        t = net.theta[off:off + n]
        if key.endswith(".weight") and len(shape) == 1:
            t.copy_((1.0 + 0.3 * torch.randn(n, generator=g)).to(DEV))
        elif key.endswith(".bias") or key.endswith("._bias"):
            t.copy_((0.2 * torch.randn(n, generator=g)).to(DEV))
    return cfg, net


def torch_reference(net, head_names, x64, dheads64):
    """float64 forward / backward from the flat parameter vector: (head outputs, flat gradient)."""
    theta = net.theta.detach().double().cpu().requires_grad_(True)
    v = lambda off, *shape: theta[off:off + int(np.prod(shape))].view(*shape)
    h = x64
    if net.fn is not None:
        h = torch.nn.functional.layer_norm(h, (net.D,), v(net.fn["g"], net.D), v(net.fn["be"], net.D), 1e-5)
    for L in net.layers:
        if L.get("dead"):
            continue
        z = h @ v(L["W"], L["n_out"], L["n_in"]).t() + v(L["b"], L["n_out"])
        if L["act"] >= 0:
            z = ACTS[L["act"]](z)
        h = torch.nn.functional.layer_norm(z, (L["n_out"],), v(L["g"], L["n_out"]), v(L["be"], L["n_out"]), 1e-5)
    outs = []
    for name in head_names:
        hd = net.heads[name]
        outs.append(h @ v(hd["W"], hd["n"], net.H).t() + v(hd["b"], hd["n"]))
    loss = sum((o * d).sum() for o, d in zip(outs, dheads64))
    loss.backward()
    return [o.detach() for o in outs], theta.grad.detach()


CASES = [
    # H, layer_N, act, fn, D, space / role, B
    (64, 1, 1, False, 4, ("policy", "disc", 2), 1000),
    (128, 1, 1, False, 4, ("policy", "disc", 2), 5000),
    (128, 1, 0, True, 17, ("policy", "box", 6), 777),
    (128, 2, 3, False, 4, ("critic", None, 1), 300),
    (64, 3, 2, True, 54, ("policy", "disc", 5), 2051),
    (64, 2, 0, False, 18, ("policy", "disc", 9), 16),
    (128, 1, 2, False, 30, ("critic", None, 1), 5),
    (64, 1, 3, True, 8, ("model", "box", 3), 1300),
    (128, 1, 1, False, 4, ("policy", "disc", 2), 128 * 300 + 9),
    (128, 3, 0, True, 12, ("policy", "box", 2), 900),     # four 128-wide layers (the spilling build)
    (128, 1, 1, False, 6, ("model", "disc", 4), 1111),     # the shared network at 128: base + common = four layers
]


@pytest.mark.parametrize("H,layer_N,act_id,fn,D,spec,B", CASES)
def test_fused_tower_forward_backward_against_torch_fp64(H, layer_N, act_id, fn, D, spec, B):
    from openrl_amd import spaces

    role, kind, n = spec
    act_space = spaces.Discrete(n) if kind == "disc" else spaces.Box(-1, 1, (n,)) if kind == "box" else spaces.Discrete(2)
    cfg, net = make_net(role, H, layer_N, act_id, fn, D, act_space)
    head_names = {"policy": ("act",), "critic": ("v_out",), "model": ("act", "v_out")}[role]
    ft = net.gt(head_names)
    assert ft is not None, "this shape must take the fused kernels"
    g = torch.Generator().manual_seed(B + D)
    R, col0 = D + 11, 3  # wide records, the observation in the middle
    n_rows = B + 37
    rec = torch.randn(n_rows, R, generator=g)
    idx = torch.randperm(n_rows, generator=g)[:B].contiguous()
    x64 = rec[idx][:, col0:col0 + D].double()
    dheads = [torch.randn(B, net.heads[h]["n"], generator=g) / B for h in head_names]
    outs_ref, grad_ref = torch_reference(net, head_names, x64, [d.double() for d in dheads])

    rec_d, idx_d = rec.to(DEV), idx.to(DEV)
    outs = [torch.full((B, net.heads[h]["n"]), float("nan"), device=DEV) for h in head_names]
    ft.prep()
    ft.forward(rec_d, col0, idx_d, B, outs[0], outs[1] if len(outs) > 1 else None)
    torch.cuda.synchronize()
    for o, r, name in zip(outs, outs_ref, head_names):
        o = o.cpu().double()
        assert torch.isfinite(o).all(), name
        err = (o - r).abs().max().item()
        assert err <= 2e-5 + 1e-4 * r.abs().max().item(), (name, err, r.abs().max().item())

    net.grad.fill_(float("nan"))
    dh = [d.to(DEV).contiguous() for d in dheads]
    ft.backward(rec_d, col0, idx_d, B, dh[0], dh[1] if len(dh) > 1 else None)
    torch.cuda.synchronize()
    got = net.grad.cpu().double()
    # every parameter the reference trains is written; the never-run fc_h block and logstd are not the kernels' to write
    bad = []
    for key, shape, off in net.entries:
        nel = int(np.prod(shape))
        a, b = got[off:off + nel], grad_ref[off:off + nel]
        if ".fc_h." in key or "logstd" in key:
            continue
        if not torch.isfinite(a).all():
            bad.append((key, "non-finite"))
            continue
        scale = max(b.abs().max().item(), 1e-6)
        err = (a - b).abs().max().item()
        if err > 3e-4 * scale + 1e-7:
            bad.append((key, err / scale))
    assert not bad, bad


@pytest.mark.parametrize("H,layer_N,act_id,fn,D", [(128, 1, 1, False, 4), (64, 2, 0, True, 17), (128, 2, 2, False, 6)])
def test_fused_tower_equals_the_layerwise_route(H, layer_N, act_id, fn, D):
    """Same rows, same parameters: head outputs and gradients of the fused kernels against orl_gen_layer_fwd / _bwd /
    orl_gen_wgrad (the route the reference goldens pin)."""
    from openrl_amd import ops_gen, spaces
    from openrl_amd.modules import generic_net as gn

    cfg, net = make_net("policy", H, layer_N, act_id, fn, D, spaces.Discrete(3))
    B = 3000
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, D, generator=g).to(DEV)
    dlog = (torch.randn(B, 3, generator=g) / B).to(DEV)
    ws = gn.GenWorkspace(net, B, True)
    feats = gn.trunk_forward(net, ws, x, True)
    logits_lw = gn.head_forward(net, ws, "act", feats).clone()
    net.grad.zero_()
    dfeat = ws.v(ws.dfeat, B, H)
    gn.head_backward(net, ws, "act", feats, dlog, dfeat, False)
    gn.trunk_backward(net, ws, dfeat)
    grad_lw = net.grad.clone()

    ft = net.gt(("act",))
    assert ft is not None
    logits = torch.empty(B, 3, device=DEV)
    ft.prep()
    ft.forward(x, 0, None, B, logits)
    net.grad.zero_()
    ft.backward(x, 0, None, B, dlog)
    torch.cuda.synchronize()
    assert torch.allclose(logits, logits_lw, rtol=1e-4, atol=2e-5)
    scale = grad_lw.abs().max().item()
    assert (net.grad - grad_lw).abs().max().item() <= 3e-4 * scale


def test_unsupported_shapes_fall_back():
    from openrl_amd import spaces

    _, net = make_net("policy", 96, 1, 1, False, 4, spaces.Discrete(2))
    assert net.gt(("act",)) is None
    _, net = make_net("policy", 128, 4, 1, False, 4, spaces.Discrete(2))  # five layers
    assert net.gt(("act",)) is None
    _, net = make_net("policy", 128, 1, 1, False, 64, spaces.Discrete(2))  # obs 64 at 128: LDS
    assert net.gt(("act",)) is None


@pytest.mark.parametrize("argv,kw", [
    (["--hidden_size", "128"], dict(obs_dim=4, episode_limit=7)),
    (["--hidden_size", "64", "--layer_N", "2", "--activation_id", "0", "--use_feature_normalization", "true"],
     dict(obs_dim=17, episode_limit=9, action_space="box6")),
    (["--use_share_model", "true", "--hidden_size", "64"], dict(obs_dim=5, episode_limit=6)),
])
def test_training_with_fused_towers_equals_the_layerwise_route(argv, kw):
    """One whole iteration (rollout, returns, 3 PPO epochs x 2 minibatches) with cfg.amd_gen_update = fused against
    layerwise: same seeds, same rollout, the weights after the update within the update path's fp32 tolerance."""
    from openrl_amd import spaces
    from openrl_amd.algorithms.ppo import PPOAlgorithm
    from openrl_amd.buffers import NormalReplayBuffer
    from openrl_amd.configs.config import default_cfg
    from openrl_amd.drivers.onpolicy_driver import OnPolicyDriver
    from openrl_amd.envs.common import make
    from openrl_amd.modules.common import PPONet
    from openrl_amd.utils.util import set_seed

    if isinstance(kw.get("action_space"), str):
        kw = dict(kw, action_space=spaces.Box(-1.0, 1.0, (int(kw["action_space"][3:]),)))
    N, T = 96, 20
    thetas, infos = [], []
    for mode in ("fused", "layerwise"):
        cfg = default_cfg(["--seed", "5", "--episode_length", str(T), "--ppo_epoch", "3", "--num_mini_batch", "2",
                           "--amd_gen_update", mode] + argv)
        env = make("SyntheticFixedStep-v0", env_num=N, device=DEV, seed=5, **kw)
        set_seed(5)
        net = PPONet(env, cfg=cfg, device=DEV, n_rollout_threads=N)
        assert net.module.generic
        used = net.module.fused_towers(one_pass=True)
        assert (used is not None) == (mode == "fused")

        class _Agent:
            num_time_steps = 0

        cfg.num_env_steps = N * T
        trainer = PPOAlgorithm(cfg, net.module, agent_num=1, device=DEV)
        buf = NormalReplayBuffer(cfg, 1, env.observation_space, env.action_space, device=DEV)
        drv = OnPolicyDriver({"cfg": cfg, "num_agents": 1, "run_dir": None, "envs": env, "device": DEV}, trainer, buf, _Agent())
        drv.reset_and_buffer_init()
        drv.episode = 0
        drv._inner_loop()
        torch.cuda.synchronize()
        thetas.append({k: m.theta.cpu().numpy().copy() for k, m in net.module.models.items()})
    for k in thetas[0]:
        np.testing.assert_allclose(thetas[0][k], thetas[1][k], rtol=2e-3, atol=3e-5, err_msg=k)


@pytest.mark.parametrize("argv", [["--hidden_size", "128"], ["--use_share_model", "true", "--hidden_size", "64"]])
def test_turn_on_false_on_fused_general_towers(argv):
    """PPOAlgorithm.train(buffer, turn_on=False) (ppo.py:226-236: the policy loss is not in the loss list) through the
    one-launch update: separate networks - the policy's parameters do not move and the critic gets bit for bit the update it
    gets with turn_on=True; shared network - the update equals the layer-wise route's (value loss only)."""
    from openrl_amd.algorithms.ppo import PPOAlgorithm
    from openrl_amd.buffers import NormalReplayBuffer
    from openrl_amd.configs.config import default_cfg
    from openrl_amd.drivers.onpolicy_driver import OnPolicyDriver
    from openrl_amd.envs.common import make
    from openrl_amd.modules.common import PPONet
    from openrl_amd.utils.util import set_seed

    N, T = 64, 16
    shared = "--use_share_model" in argv

    def run(turn_on, mode="fused"):
        cfg = default_cfg(["--seed", "7", "--episode_length", str(T), "--ppo_epoch", "2", "--amd_gen_update", mode] + argv)
        env = make("SyntheticFixedStep-v0", env_num=N, device=DEV, seed=7, obs_dim=6, episode_limit=5)
        set_seed(7)
        net = PPONet(env, cfg=cfg, device=DEV, n_rollout_threads=N)

        class _Agent:
            num_time_steps = 0

        cfg.num_env_steps = N * T
        trainer = PPOAlgorithm(cfg, net.module, agent_num=1, device=DEV)
        buf = NormalReplayBuffer(cfg, 1, env.observation_space, env.action_space, device=DEV)
        drv = OnPolicyDriver({"cfg": cfg, "num_agents": 1, "run_dir": None, "envs": env, "device": DEV}, trainer, buf, _Agent())
        drv.reset_and_buffer_init()
        drv.episode = 0
        drv.actor_rollout()
        drv.compute_returns()
        before = {k: m.theta.clone() for k, m in net.module.models.items()}
        trainer.prep_training()
        trainer.train(buf.data, turn_on=turn_on)
        torch.cuda.synchronize()
        return before, {k: m.theta.clone() for k, m in net.module.models.items()}

    if shared:
        _, a = run(False, "fused")
        _, b = run(False, "layerwise")
        np.testing.assert_allclose(a["model"].cpu().numpy(), b["model"].cpu().numpy(), rtol=2e-3, atol=3e-5)
    else:
        b0, off = run(False)
        _, on = run(True)
        assert torch.equal(off["policy"], b0["policy"]), "the policy moved with turn_on=False"
        assert not torch.equal(on["policy"], b0["policy"])
        assert torch.equal(off["critic"], on["critic"])
