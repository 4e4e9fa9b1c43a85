"""Minimal ``Logger`` with the reference's hook points (``openrl/utils/logger.py:31-207``):
``log_info(infos: dict, step: int)`` every ``log_interval`` episodes, ``info(msg)``, ``close()``.
wandb / tensorboardX back-ends are host tooling outside the hot path and are not rebuilt; the scalars
are kept in ``history`` and optionally printed."""
from __future__ import annotations

import logging
from typing import Any, Dict, List, Optional, Tuple


class Logger:
    def __init__(self, cfg=None, project_name: str = "openrl_amd", scenario_name: str = "", wandb_entity=None,
                 exp_name: Optional[str] = None, log_path=None, use_wandb: bool = False, use_tensorboard: bool = False,
                 log_level: int = logging.INFO, log_to_terminal: bool = True, verbose: bool = False):
        if use_wandb or use_tensorboard:
            raise NotImplementedError("wandb / tensorboard back-ends are not part of the MI355X engine")
        self.project_name, self.scenario_name, self.exp_name = project_name, scenario_name, exp_name
        self.history: List[Tuple[int, Dict[str, Any]]] = []
        self.verbose = verbose
        self._log = logging.getLogger("openrl_amd")

    def info(self, msg: str):
        if self.verbose:
            self._log.info(msg)

    def log_info(self, infos: Dict[str, Any], step: int) -> None:
        if not infos:
            return
        self.history.append((step, dict(infos)))
        if self.verbose:
            print("[step %d] " % step + " ".join("%s=%.5g" % (k, float(v)) for k, v in infos.items()))

    def close(self):
        pass
