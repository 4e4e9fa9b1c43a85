"""Import path of the reference's ``openrl/utils/callbacks/callbacks.py``."""
from . import BaseCallback, CallbackList, ConvertCallback, EventCallback, EveryNTimesteps  # noqa: F401
