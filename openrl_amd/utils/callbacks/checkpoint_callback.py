"""Import path of the reference's ``openrl/utils/callbacks/checkpoint_callback.py``."""
from . import CheckpointCallback  # noqa: F401
