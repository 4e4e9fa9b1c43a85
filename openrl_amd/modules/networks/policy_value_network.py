"""``PolicyValueNetwork`` (openrl/modules/networks/policy_value_network.py:34-110): a NAME for ``model_dict`` - see the package docstring."""


class PolicyValueNetwork:
    """Selects the engine's built tower of the same role in ``model_dict``; never instantiated."""

    def __init__(self, *args, **kwargs):
        raise TypeError("PolicyValueNetwork is a model_dict marker: the MI355X engine builds its towers itself (PPOModule)")
