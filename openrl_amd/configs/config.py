"""``create_config_parser()`` - the flat ``cfg`` namespace of the drop-in API.

Mirrors the reference entry point ``openrl/configs/config.py:24`` (``create_config_parser``):
same flag names, same defaults, ``--config file.yaml`` support (plain YAML with an optional
``globals:`` Jinja pre-pass, ``openrl/configs/utils.py:28-101``).  Built on ``argparse`` +
``yaml`` only (``jsonargparse`` is not available offline).

MI355X-engine extensions live under the ``amd_`` prefix so they can never collide with a
reference flag.
"""
from __future__ import annotations

import argparse
from typing import Any, List, Optional

from ._flag_table import FLAGS

# Engine-side extensions (not in the reference).
AMD_FLAGS = [
    # minibatch permutation source: "reference" = host torch.randperm on the CPU generator, bit-exact
    # with replay_data.py:578-580; "device" = keyed Feistel bijection generated on the GPU.
    ("amd_perm_mode", "str", "reference", "opt", ["reference", "device", "identity"]),
    # rollout engine: "auto" picks the fused persistent kernel for device-resident envs.
    ("amd_rollout_mode", "str", "auto", "opt", ["auto", "fused", "stepwise"]),
    # fused rollout of the single-agent device envs (synthetic, CartPole): "chain" (round 6, default) = the policy-only dependent
    # chain of csrc/orl_rollout2.h + one batched critic sweep over the stored observations; "lockstep" = the round-5 kernel
    # (policy and critic towers in the step loop) - the comparison switch.  Tic-tac-toe always runs the round-5 kernel.
    ("amd_rollout_kernel", "str", "chain", "opt", ["chain", "lockstep"]),
    # capture the PPO update epoch in a hipGraph.
    ("amd_use_graph", "bool", True, "opt", None),
    # multi-GPU gradient exchange: "p2p" = the one-shot xGMI push all-reduce fused into the optimiser-step launches
    # (orl_ppo_reduce_pair_comm / orl_ppo_apply_comm, falls back to "rccl" if peer memory cannot be mapped);
    # "rccl" = one torch.distributed all-reduce (RCCL) per optimiser step.
    ("amd_collective", "str", "p2p", "opt", ["p2p", "rccl"]),
    # fused recurrent rollout (orl_rnn_rollout_fused): critic workgroups in the same launch, one step behind their
    # policy workgroups (true), or a second launch after the policy's (false).
    ("amd_rnn_rollout_chase", "bool", True, "opt", None),
    # GEMMs of the fused tower update: "split" = exact three-term bf16 splits on the bf16 MFMA (default; error <= the fp32
    # MFMA's), "fp32" = v_mfma_f32_16x16x4_f32 (comparison / measurement).
    # "split_two_image" = round 3's variants (no transposing-read full split for wide observations): comparison switch.
    ("amd_tower_gemm", "str", "split", "opt", ["split", "fp32", "split_two_image"]),
    # GEMMs of the RECURRENT row kernel: "fp32" (default) = v_mfma_f32_16x16x4_f32 out of resident LDS images (the faster one:
    # 0.80 ms per epoch at the cfg4 shape); "split" = bf16x3 splits over images STREAMED through an LDS ring
    # (csrc/orl_rnn_stream.h; 0.83 ms - the stream's waits cost more than the MFMA time it saves),
    # "split_w4" = the same with 4 waves per workgroup / 512 registers per wave (1.07 ms).  All three are parity-tested.
    # Round 5: with "fp32" chunks of data_chunk_length == 2 (the reference default) run the register-resident row kernel
    # (csrc/orl_rnn_l2.h: both steps of a chunk in one wave's registers, no forward recompute, no state tape); other lengths,
    # and every length under "fp32_recompute" (comparison switch), the forward-sweep + recompute kernel of rounds 1 - 4.
    ("amd_rnn_gemm", "str", "fp32", "opt", ["fp32", "fp32_recompute", "split", "split_w4"]),
    # optimiser step of an MLP-tower minibatch: "two_launch" (default) = orl_ppo_reduce_pair then orl_ppo_apply(_perm);
    # "fused" = column sums of the towers' partials + clip + Adam (+ the next epoch's permutation) in ONE launch
    # (orl_ppo_reduce_apply: ticketed workgroups, the last one of a tower steps it) - same results bit for bit, but the
    # device-scope release / acquire around the ticket (L2 write-back + invalidate across the 8 XCDs) costs more than the
    # kernel boundary it removes: 15.7 us against 4.4 + 9.3 us, iteration + 1 %.
    # Round 6: "step" = orl_ppo_step: ONE launch with two DESIGNATED optimiser workgroups that fetch parameters and moments
    # while the column sums are formed and read them - published write-through - behind a ticket word; no cache fence.  Same
    # results bit for bit, and the same time: 14.3 us against 4.7 + 9.7 us (profiles/r06_experiments.md) - the step is a chain of
    # memory round trips (partials -> sums -> published -> fetched -> clip -> Adam -> stores), the kernel boundary was never the
    # cost.  The two-launch form stays the default.
    ("amd_optim_step", "str", "two_launch", "opt", ["two_launch", "step", "fused"]),
    # general (non-default) feed-forward towers: "fused" = the cross-layer kernels of csrc/orl_gen_tower.h where they take
    # the shape (hidden_size 64 / 128), "layerwise" = one launch per layer and direction everywhere.
    ("amd_gen_update", "str", "fused", "opt", ["fused", "layerwise"]),
]


def _str2bool(v: Any) -> bool:
    if isinstance(v, bool):
        return v
    s = str(v).strip().lower()
    if s in ("1", "true", "t", "yes", "y"):
        return True
    if s in ("0", "false", "f", "no", "n"):
        return False
    raise argparse.ArgumentTypeError("expected a boolean, got %r" % (v,))


def _identity(v):
    return v


_TYPES = {"int": int, "float": float, "str": str, "bool": _str2bool, "any": _identity}


class _YamlConfigAction(argparse.Action):
    """``--config x.yaml``: YAML keys become defaults; later CLI flags still override."""

    def __call__(self, parser, namespace, values, option_string=None):
        import yaml

        with open(values, "r") as fh:
            text = fh.read()
        # optional jinja ``globals:`` block (configs/utils.py:28-101)
        try:
            head = yaml.safe_load(text.split("\n\n")[0]) if text.lstrip().startswith("globals:") else None
        except Exception:
            head = None
        if isinstance(head, dict) and "globals" in head:
            import jinja2

            text = jinja2.Template(text).render(**head["globals"])
        data = yaml.safe_load(text) or {}
        data.pop("globals", None)
        types = {a.dest: a.type for a in parser._actions if a.type is not None}
        for k, v in _flatten(data).items():
            conv = types.get(k)
            if conv is not None and isinstance(v, (str, int, float, bool)) and v is not None:
                v = conv(v)  # e.g. YAML 1.1 reads "7e-4" as a string
            setattr(namespace, k, v)
        setattr(namespace, self.dest, values)


def _flatten(d, prefix=""):
    out = {}
    for k, v in d.items():
        key = prefix + str(k)
        if isinstance(v, dict) and key in ("selfplay_api", "reward_class", "vec_info_class", "env"):
            out.update(_flatten(v, key + "."))
        else:
            out[key] = v
    return out


def create_config_parser() -> argparse.ArgumentParser:
    parser = argparse.ArgumentParser(description="openrl_amd", formatter_class=argparse.RawDescriptionHelpFormatter)
    parser.add_argument("--config", action=_YamlConfigAction, default=None)
    for name, tname, default, kind, choices in list(FLAGS) + AMD_FLAGS:
        if kind == "flag":
            parser.add_argument("--" + name, dest=name, action="store_false" if default is True else "store_true",
                                default=default)
        elif kind == "pos":
            parser.add_argument(name, type=_TYPES[tname], default=default, nargs="?")
        else:
            kw = dict(dest=name, type=_TYPES[tname], default=default)
            if choices:
                kw["choices"] = choices
            parser.add_argument("--" + name, **kw)
    return parser


def default_cfg(argv: Optional[List[str]] = None):
    """Convenience: ``create_config_parser().parse_args(argv or [])``."""
    return create_config_parser().parse_args(list(argv or []))
