"""Import path of the reference's ``openrl/utils/callbacks/stop_callback.py``."""
from . import StopTrainingOnMaxEpisodes, StopTrainingOnNoModelImprovement, StopTrainingOnRewardThreshold  # noqa: F401
