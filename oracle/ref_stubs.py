"""Import shims that let the REAL reference (``/root/reference/openrl``) run in this container.

TEST INFRASTRUCTURE ONLY (oracle side).  Nothing under ``openrl_amd/`` imports this file.

The reference needs ``gymnasium``, ``gym``, ``treevalue`` and ``jsonargparse`` which are not
installed here and cannot be installed (no network).  The hot path only touches a sliver of
them:

* ``gymnasium.spaces.{Space,Box,Discrete,Dict,MultiDiscrete,MultiBinary,Tuple}`` - the
  reference dispatches on ``space.__class__.__name__`` (buffers/utils/util.py:59-71,
  replay_data.py:148, act.py:14-25) so the class NAMES must be exactly these;
* ``treevalue.{TreeValue,reduce_}`` - base class of ``ObsData`` (buffers/utils/obs_data.py:23);
* ``jsonargparse.{ArgumentParser,ActionConfigFile}`` - flag parser (configs/config.py:19).

``install()`` registers the shims in ``sys.modules`` and puts ``/root/reference`` on
``sys.path``.  It only works where ``/root/reference`` exists (the authoring container); the
GPU box never calls it - golden vectors generated through it are committed under
``tests/golden/``.
"""
from __future__ import annotations

import argparse
import os
import sys
import types

import numpy as np

REFERENCE_ROOT = "/root/reference"


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "openrl"))


# ----------------------------------------------------------------------------- gymnasium.spaces
class Space:
    def __init__(self, shape=None, dtype=None):
        self._shape = None if shape is None else tuple(shape)
        self.dtype = dtype

    @property
    def shape(self):
        return self._shape


class Box(Space):
    def __init__(self, low=-np.inf, high=np.inf, shape=None, dtype=np.float32):
        if shape is None:
            shape = np.shape(low)
        super().__init__(shape, dtype)
        self.low = np.full(self._shape, low, dtype=dtype) if np.isscalar(low) else np.asarray(low, dtype=dtype)
        self.high = np.full(self._shape, high, dtype=dtype) if np.isscalar(high) else np.asarray(high, dtype=dtype)


class Discrete(Space):
    def __init__(self, n, start=0):
        super().__init__((), np.int64)
        self.n = int(n)
        self.start = int(start)


class MultiDiscrete(Space):
    def __init__(self, nvec):
        self.nvec = np.asarray(nvec, dtype=np.int64)
        super().__init__(self.nvec.shape, np.int64)
        self.low = np.zeros_like(self.nvec)
        self.high = self.nvec - 1


class MultiBinary(Space):
    def __init__(self, n):
        super().__init__((int(n),), np.int8)
        self.n = int(n)


class Dict(Space):  # noqa: A001 - name must match the reference's class-name dispatch
    def __init__(self, spaces=None, **kw):
        super().__init__(None, None)
        self.spaces = dict(spaces or {})
        self.spaces.update(kw)

    def __getitem__(self, k):
        return self.spaces[k]

    def __iter__(self):
        return iter(self.spaces)

    def keys(self):
        return self.spaces.keys()


class Tuple(Space):  # noqa: A001
    def __init__(self, spaces):
        super().__init__(None, None)
        self.spaces = tuple(spaces)

    def __getitem__(self, i):
        return self.spaces[i]

    def __len__(self):
        return len(self.spaces)


def _make_gym_module(name: str) -> types.ModuleType:
    m = types.ModuleType(name)
    m.__version__ = "0.29.1-stub"
    sp = types.ModuleType(name + ".spaces")
    for cls in (Space, Box, Discrete, MultiDiscrete, MultiBinary, Dict, Tuple):
        setattr(sp, cls.__name__, cls)
    m.spaces = sp

    class Env:  # minimal placeholders used only in type annotations
        pass

    class Wrapper(Env):
        pass

    m.Env = Env
    m.Wrapper = Wrapper
    sys.modules[name] = m
    sys.modules[name + ".spaces"] = sp
    return m


# ----------------------------------------------------------------------------- treevalue
class TreeValue(dict):
    def __init__(self, data=None):
        super().__init__(data or {})


def reduce_(tree, fn):
    return fn(**dict(tree))


# ----------------------------------------------------------------------------- jsonargparse
def _str2bool(v):
    if isinstance(v, bool):
        return v
    s = str(v).lower()
    if s in ("1", "true", "yes", "y", "t"):
        return True
    if s in ("0", "false", "no", "n", "f"):
        return False
    raise argparse.ArgumentTypeError("bool expected, got %r" % (v,))


class ActionConfigFile(argparse.Action):
    def __call__(self, parser, namespace, values, option_string=None):  # pragma: no cover
        raise NotImplementedError("--config is not supported by the oracle shim")


class ArgumentParser(argparse.ArgumentParser):
    """argparse with the three jsonargparse behaviours config.py relies on."""

    def __init__(self, *a, **kw):
        kw.pop("env_prefix", None)
        kw.pop("default_env", None)
        super().__init__(*a, **kw)

    def add_argument(self, *names, **kw):
        t = kw.get("type")
        if t is bool:
            kw["type"] = _str2bool
        elif t is not None and t not in (int, float, str) and not callable(getattr(t, "__call__", None)):
            kw.pop("type")
        elif t is not None and t not in (int, float, str, _str2bool) and not isinstance(t, type):
            kw.pop("type")  # typing generics such as List[dict]
        elif t in (dict, list):
            kw.pop("type")
        if names and not names[0].startswith("-"):
            kw.setdefault("nargs", "?")  # the four positional args with defaults (config.py:1044-1063)
        if kw.get("action") is ActionConfigFile:
            kw.pop("action")
        return super().add_argument(*names, **kw)


def install() -> None:
    """Register the shims and make ``import openrl`` resolve to the reference."""
    if not reference_available():
        raise RuntimeError("reference tree %s is not present on this machine" % REFERENCE_ROOT)
    sys.dont_write_bytecode = True
    if "gymnasium" not in sys.modules:
        _make_gym_module("gymnasium")
    if "gym" not in sys.modules:
        _make_gym_module("gym")
    if "treevalue" not in sys.modules:
        tv = types.ModuleType("treevalue")
        tv.TreeValue = TreeValue
        tv.reduce_ = reduce_
        sys.modules["treevalue"] = tv
    if "jsonargparse" not in sys.modules:
        ja = types.ModuleType("jsonargparse")
        ja.ArgumentParser = ArgumentParser
        ja.ActionConfigFile = ActionConfigFile
        sys.modules["jsonargparse"] = ja
    if REFERENCE_ROOT not in sys.path:
        sys.path.append(REFERENCE_ROOT)


def reference_cfg(argv=None):
    """``create_config_parser().parse_args(argv)`` of the reference (configs/config.py:24)."""
    install()
    from openrl.configs.config import create_config_parser

    return create_config_parser().parse_args(list(argv or []))
