"""``PolicyNetwork`` (openrl/modules/networks/policy_network.py:36-128): a NAME for ``model_dict`` - see the package docstring."""


class PolicyNetwork:
    """Selects the engine's built tower of the same role in ``model_dict``; never instantiated."""

    def __init__(self, *args, **kwargs):
        raise TypeError("PolicyNetwork is a model_dict marker: the MI355X engine builds its towers itself (PPOModule)")
