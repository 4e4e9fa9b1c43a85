"""Host-side logic that needs no GPU: cfg flags/defaults, spaces, shard math, and the linearity that
the multi-GPU design rests on (sum of per-shard raw gradient sums / global denominators == full batch)."""
import numpy as np
import pytest
import torch

from oracle import ppo_oracle as po
from tests import helpers as H


def test_hot_path_flag_defaults_match_the_reference():
    """SURVEY.md section 5.6 (each verified against openrl/configs/config.py)."""
    from openrl_amd.configs.config import default_cfg

    cfg = default_cfg([])
    want = dict(seed=0, n_rollout_threads=32, episode_length=200, hidden_size=64, layer_N=1, activation_id=1,
                use_valuenorm=True, use_popart=False, use_feature_normalization=False, use_orthogonal=True, gain=0.01,
                use_recurrent_policy=False, recurrent_N=1, data_chunk_length=2, lr=5e-4, critic_lr=5e-4, opti_eps=1e-5,
                weight_decay=0, ppo_epoch=10, num_mini_batch=1, mini_batch_size=None, clip_param=0.2,
                entropy_coef=0.01, value_loss_coef=0.5, use_max_grad_norm=True, max_grad_norm=10, use_gae=True,
                gamma=0.99, gae_lambda=0.95, use_proper_time_limits=False, use_huber_loss=True, huber_delta=10.0,
                use_clipped_value_loss=True, use_value_active_masks=True, use_policy_active_masks=True,
                dual_clip_ppo=False, dual_clip_coeff=3, use_adv_normalize=False, use_share_model=False,
                use_linear_lr_decay=False, use_joint_action_loss=False, use_amp=False, use_deepspeed=False)
    for k, v in want.items():
        assert getattr(cfg, k) == v, k
    # store_true / store_false toggles and typed flags behave like the reference parser
    cfg = default_cfg(["--use_huber_loss", "--use_valuenorm", "false", "--ppo_epoch", "3", "--lr", "7e-4"])
    assert cfg.use_huber_loss is False and cfg.use_valuenorm is False and cfg.ppo_epoch == 3 and cfg.lr == 7e-4
    assert cfg.amd_perm_mode == "reference"


def test_yaml_config_file(tmp_path):
    from openrl_amd.configs.config import default_cfg

    p = tmp_path / "ppo.yaml"
    p.write_text("seed: 5\nlr: 7e-4\nepisode_length: 25\nuse_adv_normalize: true\n")
    cfg = default_cfg(["--config", str(p), "--seed", "9"])
    assert cfg.lr == 7e-4 and cfg.episode_length == 25 and cfg.use_adv_normalize is True and cfg.seed == 9


def test_spaces_shapes():
    from openrl_amd import spaces

    assert spaces.obs_dim(spaces.Box(-1, 1, (17,))) == 17
    assert spaces.act_shape(spaces.Discrete(5)) == 1 and spaces.act_shape(spaces.Box(-1, 1, (6,))) == 6
    d = spaces.Dict({"policy": spaces.Box(-1, 1, (18,)), "critic": spaces.Box(-1, 1, (54,))})
    assert spaces.obs_dim(spaces.policy_obs_space(d)) == 18 and spaces.obs_dim(spaces.critic_obs_space(d)) == 54
    with pytest.raises(NotImplementedError):
        spaces.obs_dim(spaces.Box(0, 1, (3, 8, 8)))


def test_env_shards_cover_all_envs_contiguously():
    from openrl_amd.distributed import shard_range

    for n, g in ((4096, 8), (4096, 3), (7, 8), (10, 4)):
        spans = [shard_range(n, r, g) for r in range(g)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert max(hi - lo for lo, hi in spans) - min(hi - lo for lo, hi in spans) <= 1


def _raw_sums(hp, pspec, cspec, ptheta, ctheta, sample, vn):
    """Unnormalised loss numerators + denominators of one shard (what orl_ppo_fwd_bwd accumulates)."""
    t = lambda a: None if a is None else torch.as_tensor(a, dtype=torch.float32)
    (critic_obs, obs, actions, value_preds, returns, active, old_logp, adv, amask) = tuple(t(a) for a in sample)
    pth, cth = ptheta.clone().requires_grad_(True), ctheta.clone().requires_grad_(True)
    values = po.tower_forward(cspec, cth, critic_obs)
    out = po.tower_forward(pspec, pth, obs)
    lg = po.masked_logits(out, amask)
    dist = torch.distributions.Categorical(logits=lg)
    logp = dist.log_prob(actions.squeeze(-1).long()).unsqueeze(-1)
    ratio = torch.exp(logp - old_logp)
    surr = torch.min(ratio * adv, torch.clamp(ratio, 1 - hp.clip_param, 1 + hp.clip_param) * adv)
    p_num = (-surr * active).sum() - hp.entropy_coef * (dist.entropy() * active.squeeze(-1)).sum()
    vclip = value_preds + (values - value_preds).clamp(-hp.clip_param, hp.clip_param)
    rn = vn.normalize(returns) if vn is not None else returns
    vl = torch.max(po.huber_loss(rn - values, hp.huber_delta), po.huber_loss(rn - vclip, hp.huber_delta))
    v_num = (vl * active).sum() * hp.value_loss_coef
    p_num.backward()
    v_num.backward()
    return pth.grad.clone(), cth.grad.clone(), active.sum()


def test_shard_sums_reproduce_the_full_batch_gradient():
    """G ranks == 1 rank with the concatenated batch: gradients of the per-shard loss
    NUMERATORS are summed and divided by the summed denominators."""
    g = H.load_golden("train_discrete")
    cfg = H.case_cfg(g)
    hp = po.hyper_from_cfg(cfg)
    pspec, cspec = H.case_specs(g)
    ptheta, ctheta = torch.tensor(g["theta_p0"]), torch.tensor(g["theta_c0"])
    b = H.case_buffer(g)
    vn = po.ValueNormOracle()
    adv = po.advantages(b["returns"], b["value_preds"], b["active_masks"], vn, False)
    fr = po.flat_rows
    full = (fr(b["critic_obs"][:-1]), fr(b["policy_obs"][:-1]), fr(b["actions"]), fr(b["value_preds"][:-1]),
            fr(b["returns"][:-1]), fr(b["active_masks"][:-1]), fr(b["action_log_probs"]), adv.reshape(-1, 1),
            fr(b["action_masks"][:-1]))
    vn.update(full[4])  # every rank applies the same (all-reduced) batch moments
    # reference semantics on the full batch
    pth, cth = ptheta.clone().requires_grad_(True), ctheta.clone().requires_grad_(True)
    vn2 = po.ValueNormOracle()
    losses, *_ = po.prepare_loss(hp, pspec, pth, cspec, cth, vn2, tuple(torch.as_tensor(a) for a in full))
    for loss in losses:
        loss.backward()
    # two shards (env halves), summed like the all-reduce does
    M = full[0].shape[0]
    halves = [np.arange(0, M // 2), np.arange(M // 2, M)]
    gp, gc, den = 0, 0, 0
    for idx in halves:
        a, c, d = _raw_sums(hp, pspec, cspec, ptheta, ctheta, tuple(x[idx] for x in full), vn)
        gp, gc, den = gp + a, gc + c, den + d
    np.testing.assert_allclose((gp / den).numpy(), pth.grad.numpy(), rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose((gc / den).numpy(), cth.grad.numpy(), rtol=1e-4, atol=1e-7)


def test_model_dict_accepts_the_stock_network_classes_and_refuses_foreign_ones():
    """ppo_net.py:57-58 / ppo_module.py:58-89: model_dict names network classes per role.  The stock classes (the
    reference's or this package's markers of the same names) select the built towers; anything else is refused."""
    import pytest

    from openrl_amd.modules.networks import PolicyNetwork, PolicyValueNetwork, ValueNetwork
    from openrl_amd.modules.ppo_module import check_model_dict

    check_model_dict(None)
    check_model_dict({"policy": PolicyNetwork, "critic": ValueNetwork})
    check_model_dict({"model": PolicyValueNetwork})
    # the reference's own class: identified by the module that DEFINES it, not by its name alone
    ref_like = type("PolicyNetwork", (), {"__module__": "openrl.modules.networks.policy_network"})
    check_model_dict({"policy": ref_like})
    impostor = type("PolicyNetwork", (), {"__module__": "my_project.nets", "forward": lambda self, x: x})
    with pytest.raises(NotImplementedError, match="only the stock PolicyNetwork"):
        check_model_dict({"policy": impostor})  # same NAME, its own forward: refused, not silently replaced
    from openrl_amd.modules.ppo_module import check_model_dict_roles

    check_model_dict_roles({"policy": PolicyNetwork, "critic": ValueNetwork}, share_model=False)
    check_model_dict_roles({"model": PolicyValueNetwork}, share_model=True)
    with pytest.raises(ValueError):
        check_model_dict_roles({"model": PolicyValueNetwork}, share_model=False)
    with pytest.raises(ValueError):
        check_model_dict_roles({"policy": PolicyNetwork}, share_model=True)

    class MyNet(PolicyNetwork):  # a custom forward cannot run on the HIP towers
        pass

    with pytest.raises(NotImplementedError, match="only the stock PolicyNetwork"):
        check_model_dict({"policy": MyNet})
    with pytest.raises(NotImplementedError):
        check_model_dict({"critic": PolicyNetwork})
    with pytest.raises(KeyError):
        check_model_dict({"actor": PolicyNetwork})
    with pytest.raises(TypeError):
        PolicyNetwork()


def test_committed_bench_line_carries_the_contract_fields():
    """profiles/r05_bench_line.json is what `python bench.py` printed on an MI355X for the final tree: the fields the
    driver's contract names are all there, typed, and consistent with each other (a schema check - no GPU needed)."""
    import json
    import os

    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r05_bench_line.json")
    line = json.load(open(path))
    for k, t in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                 ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                 ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert isinstance(line[k], t), k
    assert "vs_baseline" in line and line["n_gpus"] == 1 and line["higher_is_better"] is True
    assert "workload" in line["config"] and "model" not in line["config"]
    # value = env-steps of the timed steps / their wall time
    steps_total = line["config"]["global_envs"] * line["config"]["rollout_len"] * line["steps"]
    assert abs(line["value"] - steps_total / (line["ms_per_step"] * 1e-3 * line["steps"])) <= 1e-3 * line["value"]
    r = line["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert abs(r["achieved"] - r["flops_per_launch"] / (r["launch_ms"] * 1e-3) / 1e12) < 0.05  # TFLOP/s from the live launch time
    assert r["launches_timed"] > 0 and (r["traffic"] is None or r["traffic"] > 0)
    c = line["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["value"] > 0 and c["cores"] >= 1 and c["unit"] == line["unit"]
    assert isinstance(c["sample"], str) and c["sample"]
    # round 5: every other_configs entry prices its dominant launch (flops per launch, fraction of the fp32 MFMA peak)
    for o in line.get("other_configs", []):
        if "error" in o:
            continue
        assert "flops_fwd_per_row_both_towers" in o and "flops_per_launch" in o and "frac" in o, o.get("workload")
        if o["flops_per_launch"] is not None:
            assert abs(o["frac"] - o["flops_per_launch"] / (o["dominant_kernel_ms"] * 1e-3) / 157.3e12) < 2e-3


def test_tape_layout_is_a_bijection_and_its_operand_reads_are_conflict_free():
    """The recurrent update's tape is streamed HBM -> LDS verbatim, so its HBM layout IS the LDS access pattern of the weight-gradient
    kernel (csrc/orl_rnn.h: tape_rot / tape_off / tape_krow).  The model of tools/lds_bank_model.py (lane groups and bank modulus of
    ds_read_b32 from the MI355X guide) must say: every (group, row, element) has its own slot, the k-step row map covers the 16 rows,
    and both operand reads take the ideal 2 cycles - round 4's rotation took 8 / 4 (measured: SQ_LDS_BANK_CONFLICT 28.1 M -> 0)."""
    import importlib.util
    import os
    import re

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("lds_bank_model", os.path.join(root, "tools", "lds_bank_model.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    assert m.tape_report(True) == (True, 2, 2, 2)
    old = m.tape_report(False)
    assert old[0] and old[1] > 2  # the old rotation was a valid layout with conflicting bf16 operand reads
    # the header states the same two maps (and ships with the new rotation on)
    src = open(os.path.join(root, "openrl_amd", "csrc", "orl_rnn.h")).read()
    assert re.search(r"#define ORL_TAPE_ROT8 1\b", src)
    assert "ORL_TAPE_ROT8 ? (g & 7) : 4 * (g & 3)" in src and "ORL_TAPE_ROT8 ? s + 4 * q : 4 * s + q" in src
