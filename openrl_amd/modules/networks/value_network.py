"""``ValueNetwork`` (openrl/modules/networks/value_network.py:37-111): a NAME for ``model_dict`` - see the package docstring."""


class ValueNetwork:
    """Selects the engine's built tower of the same role in ``model_dict``; never instantiated."""

    def __init__(self, *args, **kwargs):
        raise TypeError("ValueNetwork is a model_dict marker: the MI355X engine builds its towers itself (PPOModule)")
