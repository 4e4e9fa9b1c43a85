"""``Logger`` with the reference's interface and on-disk layout (``openrl/utils/logger.py:31-207``).

* run directory ``<log_path>/<project_name>/<scenario_name>/<exp_name>/run<k>`` with ``k`` = 1 + the largest existing
  run number (logger.py:76-104), ``log.txt`` inside it, ``cfg.render_save_path`` set when the config has ``render``;
* ``log_info(infos, step)`` formats every key exactly like logger.py:185-207 (tensors -> ``.item()``, sequences ->
  ``np.mean``) and forwards each ``(key, value, step)`` to the scalar back-end; ``log_learner_info`` prefixes the keys
  with ``Learner_<id>/`` (logger.py:167-183);
* back-ends: ``use_tensorboard`` writes through tensorboardX when that package is importable and otherwise to
  ``<run_dir>/logs/scalars.jsonl`` + ``scalars.csv`` (same keys / steps, readable without any dependency - there is no
  network to install tensorboardX in the build image); ``use_wandb`` needs the ``wandb`` package and raises
  ``ImportError`` without it.

Every logged record is also kept in ``history`` (tests, notebooks).  Logging reads the train_info scalars, which is
the one device -> host synchronisation of an iteration; the driver only calls it every ``log_interval`` iterations.
"""
from __future__ import annotations

import csv
import json
import logging
import os
import socket
from pathlib import Path
from typing import Any, Dict, List, Optional, Tuple

import numpy as np
import torch

RUNNING_PROGRAMS = ("learner", "server_learner", "local", "whole", "local_evaluator")


class _FileScalarWriter:
    """tensorboard-free scalar sink: one JSON line and one CSV row per (key, value, step)."""

    def __init__(self, log_dir: str) -> None:
        os.makedirs(log_dir, exist_ok=True)
        self._jsonl = open(os.path.join(log_dir, "scalars.jsonl"), "a")
        self._csv_file = open(os.path.join(log_dir, "scalars.csv"), "a", newline="")
        self._csv = csv.writer(self._csv_file)
        if self._csv_file.tell() == 0:
            self._csv.writerow(["step", "key", "value"])

    def add_scalars(self, main_tag: str, tag_scalar_dict: Dict[str, float], global_step: int) -> None:
        for k, v in tag_scalar_dict.items():
            self._jsonl.write(json.dumps({"step": int(global_step), "key": k, "value": float(v)}) + "\n")
            self._csv.writerow([int(global_step), k, float(v)])
        self._jsonl.flush()
        self._csv_file.flush()

    def close(self) -> None:
        self._jsonl.close()
        self._csv_file.close()


class Logger:
    def __init__(self, cfg=None, project_name: str = "openrl", scenario_name: str = "openrl", wandb_entity: str = "openrl",
                 exp_name: Optional[str] = None, log_path: Optional[str] = None, use_wandb: bool = False,
                 use_tensorboard: bool = False, log_level: int = logging.DEBUG, log_to_terminal: bool = True,
                 verbose: bool = False) -> None:
        self.use_wandb, self.use_tensorboard = bool(use_wandb), bool(use_tensorboard)
        self.skip_logging = bool(cfg is not None and getattr(cfg, "use_deepspeed", False)
                                 and getattr(cfg, "local_rank", 0) != 0)
        self.log_level, self.log_path = log_level, log_path
        self.project_name, self.scenario_name, self.wandb_entity = project_name, scenario_name, wandb_entity
        self.log_to_terminal = log_to_terminal
        self.exp_name = exp_name if exp_name is not None else (getattr(cfg, "experiment_name", "") if cfg is not None else "")
        self.cfg = cfg
        self.verbose = verbose
        self.history: List[Tuple[int, Dict[str, Any]]] = []
        self.run_dir: Optional[Path] = None
        self.writter = None  # (sic) the reference's attribute name
        self._wandb = None
        self._log = logging.getLogger("openrl_amd")
        self._init()

    # ------------------------------------------------------------------ logger.py:64-151
    def _init(self) -> None:
        program_type = getattr(self.cfg, "program_type", "local") if self.cfg is not None else "local"
        if program_type not in RUNNING_PROGRAMS:
            return
        if self.log_path is None:
            assert not self.use_wandb and not self.use_tensorboard, "log_path must be set when using wandb or tensorboard"
            run_dir = None
        else:
            run_dir = Path(self.log_path) / self.project_name / self.scenario_name / (self.exp_name or "rl")
            os.makedirs(str(run_dir), exist_ok=True)
            if not self.use_wandb:
                nums = [int(str(f.name).split("run")[1]) for f in run_dir.iterdir()
                        if str(f.name).startswith("run") and str(f.name)[3:].isdigit()]
                run_dir = run_dir / ("run%i" % (max(nums) + 1 if nums else 1))
                os.makedirs(str(run_dir), exist_ok=True)
        if self.cfg is not None and hasattr(self.cfg, "render") and run_dir is not None:
            self.cfg.render_save_path = run_dir / "render.png"
        self._log.setLevel(self.log_level)
        for h in list(self._log.handlers):
            self._log.removeHandler(h)
        fmt = logging.Formatter("%(asctime)s [%(levelname)s] %(message)s")
        if run_dir is not None:
            fh = logging.FileHandler(os.path.join(run_dir, "log.txt"))
            fh.setFormatter(fmt)
            self._log.addHandler(fh)
        if self.verbose:
            sh = logging.StreamHandler()
            sh.setFormatter(fmt)
            self._log.addHandler(sh)
        self._log.propagate = False
        if self.use_wandb and not self.skip_logging:
            try:
                import wandb
            except ImportError as e:
                raise ImportError("use_wandb=True needs the wandb package (not installable in this image)") from e
            self._wandb = wandb
            wandb.init(config=self.cfg, project=self.project_name, entity=self.wandb_entity, notes=socket.gethostname(),
                       name=self.scenario_name + "_" + str(self.exp_name) + "_seed" + str(getattr(self.cfg, "seed", 0)),
                       dir=str(run_dir), job_type="training", reinit=True)
        elif self.use_tensorboard:
            self.log_dir = str(run_dir / "logs")
            os.makedirs(self.log_dir, exist_ok=True)
            try:
                from tensorboardX import SummaryWriter

                self.writter = SummaryWriter(self.log_dir)
            except ImportError:
                self.writter = _FileScalarWriter(self.log_dir)
        self.run_dir = run_dir

    def close(self) -> None:
        if self._wandb is not None and not self.skip_logging:
            self._wandb.finish()
        if self.writter is not None and hasattr(self.writter, "close"):
            self.writter.close()
        for h in list(self._log.handlers):
            h.close()
            self._log.removeHandler(h)

    def info(self, msg: str) -> None:
        self._log.info(msg)

    @staticmethod
    def _scalar(v) -> float:
        if isinstance(v, torch.Tensor):
            v = v.item()
        if not isinstance(v, (int, float)):
            v = np.mean(v)
        return float(v)

    # ------------------------------------------------------------------ logger.py:167-183
    def log_learner_info(self, leaner_id: int, infos: Dict[str, Any], step: int) -> None:
        if not (self.use_wandb or self.use_tensorboard):
            return
        for k, v in infos.items():
            key = "Learner_{}/{}".format(leaner_id, k)
            if self._wandb is not None:
                if not self.skip_logging:
                    self._wandb.log({key: self._scalar(v)}, step=step)
            elif self.writter is not None:
                self.writter.add_scalars(key, {key: self._scalar(v)}, step)

    # ------------------------------------------------------------------ logger.py:185-207
    def log_info(self, infos: Dict[str, Any], step: int) -> None:
        if not infos:
            return
        rec = {k: self._scalar(v) for k, v in infos.items()}
        self.history.append((step, rec))
        if not (self.use_wandb or self.use_tensorboard or self.log_to_terminal):
            return
        text = "\n"
        for k, v in rec.items():
            text += f"\t{k}: {v}\n"
            if self._wandb is not None:
                if not self.skip_logging:
                    self._wandb.log({k: v}, step=step)
            elif self.writter is not None:
                self.writter.add_scalars(k, {k: v}, step)
        if self.log_to_terminal:
            self._log.info(text)
