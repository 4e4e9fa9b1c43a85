"""General towers, host side (no GPU): parameter layout and initial weights of ``GenNet`` against the reference
goldens (oracle/gen_golden.py), the dispatch rule, and the state_dict key set of every network kind."""
import numpy as np
import pytest
import torch

from openrl_amd import spaces
from openrl_amd.modules import generic_net as gn
from openrl_amd.utils.util import set_seed
from tests import helpers as H

CASES = ["train_gen_h128_l2_tanh_fn", "train_gen_elu_box", "train_gen_leaky_l3", "train_share", "train_share_box_fn",
         "train_share_h128", "train_gen_h128_l3_elu_fn"]


def _spaces(g):
    D = g["buf_policy_obs"].shape[-1]
    if "buf_action_masks" in g:
        return D, spaces.Discrete(g["buf_action_masks"].shape[-1])
    return D, spaces.Box(-1, 1, (g["buf_actions"].shape[-1],))


@pytest.mark.parametrize("case", CASES)
def test_layout_and_initial_weights_match_the_reference(case):
    """Parameter count, registration order and initial values (generator consumption order of the reference
    constructors, incl. the never-run fc_h block and its deep-copied clones) of policy / critic / shared networks."""
    g = H.load_golden(case)
    cfg = H.case_cfg(g)
    cfg.seed = int(g["perm_seed"]) - 1234
    D, act = _spaces(g)
    set_seed(cfg.seed)
    roles = [("model", "theta_m0")] if "theta_m0" in g else [("policy", "theta_p0"), ("critic", "theta_c0")]
    for role, key in roles:  # policy first, then critic: the reference's construction order
        net = gn.GenNet(role, cfg, D, act, "cpu")
        net.host_init(cfg)
        got = net.reference_flat().numpy()
        assert got.shape == g[key].shape, (role, got.shape, g[key].shape)
        np.testing.assert_allclose(got, g[key], rtol=0, atol=1e-5, err_msg=role)
        # round trip through the reference-order flat vector
        net2 = gn.GenNet(role, cfg, D, act, "cpu")
        net2.load_reference_flat(g[key])
        assert np.array_equal(net2.reference_flat().numpy(), g[key])


def test_dispatch_rule_and_state_dict_keys():
    from openrl_amd.configs.config import default_cfg

    base = default_cfg([])
    assert not gn.needs_generic(base, spaces.Discrete(2), False)
    assert not gn.needs_generic(base, spaces.Box(-1, 1, (3,)), False)
    assert gn.needs_generic(base, spaces.MultiDiscrete([3, 2]), False)
    assert gn.needs_generic(base, spaces.Discrete(2), True)
    for argv in (["--hidden_size", "128"], ["--layer_N", "2"], ["--activation_id", "0"],
                 ["--use_feature_normalization", "true"], ["--use_share_model", "true"]):
        assert gn.needs_generic(default_cfg(argv), spaces.Discrete(2), False), argv
    cfg = default_cfg(["--layer_N", "3", "--use_feature_normalization", "true"])
    net = gn.GenNet("model", cfg, 5, spaces.MultiDiscrete([3, 2]), "cpu")
    keys = [k for k, _, _ in net.entries]
    assert keys[:2] == ["obs_prep.feature_norm.weight", "obs_prep.feature_norm.bias"]
    assert "obs_prep.mlp.fc_h.0.weight" in keys and "obs_prep.mlp.fc2.1.2.bias" in keys
    assert keys.index("common.fc3.1.bias") < keys.index("v_out.weight") < keys.index("act.action_outs.0.linear.weight")
    assert keys[-2:] == ["act.action_outs.1.linear.weight", "act.action_outs.1.linear.bias"]
    sd = net.state_dict()
    assert "critic_obs_prep.mlp.fc1.0.weight" in sd
    assert sd["critic_obs_prep.mlp.fc1.0.weight"].data_ptr() == sd["obs_prep.mlp.fc1.0.weight"].data_ptr()
    # every parameter is covered exactly once by the internal layout
    cover = np.zeros(net.n_params, np.int32)
    for _, shape, off in net.entries:
        cover[off:off + int(np.prod(shape))] += 1
    assert (cover == 1).all()


@pytest.mark.parametrize("tag", ["default", "general", "shared"])
def test_state_dict_keys_and_shapes_equal_the_reference_networks(tag):
    """Keys (in order) and shapes of ``state_dict()`` against state dicts SAVED BY THE REFERENCE's own PolicyNetwork /
    ValueNetwork / PolicyValueNetwork (tests/golden/state_dicts.npz, oracle/gen_golden.py::state_dict_case), and a
    load -> state_dict round trip of those tensors."""
    from openrl_amd.modules import ppo_module as pm
    from openrl_amd import ops

    g = H.load_golden("state_dicts")
    cfg = H.case_cfg({"argv": g[tag + "/argv"]})
    D = g[tag + "/probe_obs"].shape[1]
    names = sorted({k.split("/")[1] for k in g if k.startswith(tag + "/") and k.count("/") == 2})
    for name in names:
        ref = OrderedDictFrom(g, tag + "/" + name + "/")
        if tag == "default":
            K, head = (2, ops.HEAD_CATEGORICAL) if name == "policy" else (1, ops.HEAD_VALUE)
            net = pm.Tower(name, D, K, head, 64, "cpu", torch.zeros(4866 if name == "policy" else 4801))
        else:
            act = spaces.Box(-1, 1, (3,)) if tag == "general" else spaces.Discrete(3)
            net = gn.GenNet(name, cfg, D, act, "cpu")
        if cfg.use_valuenorm and name in ("critic", "model"):
            net.value_normalizer = _HostValueNorm()
        ours = net.state_dict()
        assert list(ours.keys()) == list(ref.keys()), (name, list(ours.keys()), list(ref.keys()))
        for k in ref:
            assert tuple(ours[k].shape) == tuple(ref[k].shape), (k, ours[k].shape, ref[k].shape)
        net.load_state_dict({k: torch.tensor(v) for k, v in ref.items()})
        for k, v in net.state_dict().items():
            assert np.array_equal(np.asarray(v), ref[k]), k


def OrderedDictFrom(g, prefix):
    from collections import OrderedDict

    return OrderedDict((k[len(prefix):], g[k]) for k in g.files if k.startswith(prefix)) if hasattr(g, "files") else \
        OrderedDict((k[len(prefix):], v) for k, v in g.items() if k.startswith(prefix))


class _HostValueNorm:
    """state holder with the device ValueNorm's interface (a CPU stand-in for the key / shape checks)."""

    def __init__(self):
        self.state = torch.zeros(3)
