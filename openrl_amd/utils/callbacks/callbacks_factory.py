"""Import path of the reference's ``openrl/utils/callbacks/callbacks_factory.py``."""
from . import CallbackFactory, callbacks_dict  # noqa: F401
