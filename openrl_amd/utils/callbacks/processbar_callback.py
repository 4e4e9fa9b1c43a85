"""Import path of the reference's ``openrl/utils/callbacks/processbar_callback.py``."""
from . import ProgressBarCallback  # noqa: F401
