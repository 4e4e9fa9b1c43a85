"""Import paths of the reference's network classes (``openrl/modules/networks/{policy,value,policy_value}_network.py``).

In this engine a network is a flat parameter vector driven by HIP kernels (``modules/ppo_module.py::Tower``,
``modules/generic_net.py::GenNet``), not a ``torch.nn.Module``; the three names exist so that code written against the
reference - ``PPONet(env, cfg, model_dict={"policy": PolicyNetwork, "critic": ValueNetwork})`` (ppo_net.py:57-58,
ppo_module.py:58-89) - keeps working: ``model_dict`` entries naming these classes, or the reference's own classes of
the same names, select the built towers (the constructor configuration they would read - hidden_size, layer_N,
activation_id, use_feature_normalization, use_recurrent_policy ... - is read from ``cfg`` exactly as they do)."""
from .policy_network import PolicyNetwork  # noqa: F401
from .policy_value_network import PolicyValueNetwork  # noqa: F401
from .value_network import ValueNetwork  # noqa: F401
