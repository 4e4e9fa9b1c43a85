"""Import path of the reference's ``openrl/utils/callbacks/eval_callback.py``."""
from . import EvalCallback  # noqa: F401
