"""Tensor-level wrappers over the C ABI (``include/orl_hip.h``).

Thin and mechanical on purpose: shape/dtype checks, raw pointers, current stream.  Every function
launches HIP kernels asynchronously on ``torch.cuda.current_stream()`` and raises ``NativeError``
on any failure.  No function here computes anything in Python.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import torch

from . import _native as nat
from ._native import (AdamState, BufferPtrs, GatherDesc, NetDesc, PackSrc, PPOHParams, RolloutArgs, fptr, ptr,
                      stream_ptr)

HEAD_VALUE, HEAD_CATEGORICAL, HEAD_GAUSSIAN = nat.ORL_HEAD_VALUE, nat.ORL_HEAD_CATEGORICAL, nat.ORL_HEAD_GAUSSIAN
ENV_SYNTH, ENV_CARTPOLE = nat.ORL_ENV_SYNTH, nat.ORL_ENV_CARTPOLE
N_STATS = nat.ORL_N_STATS


def _lib():
    return nat.load()


def net_desc(obs_dim: int, n_out: int, head_kind: int, hidden: int = 64) -> NetDesc:
    return NetDesc(int(obs_dim), int(hidden), int(n_out), int(head_kind))


def param_count(net: NetDesc) -> int:
    n = _lib().orl_param_count(C.byref(net))
    if n < 0:
        nat.check(n, "orl_param_count")
    return n


def raw_grad_count(net: NetDesc) -> int:
    n = _lib().orl_raw_grad_count(C.byref(net))
    if n < 0:
        nat.check(n, "orl_raw_grad_count")
    return n


def record_width(Dp: int, Dc: int, a: int, K: int) -> int:
    return _lib().orl_record_width(Dp, Dc, a, K)


def ppo_max_blocks() -> int:
    return _lib().orl_ppo_max_blocks()


def _dev(t: torch.Tensor):
    nat.require_gpu(t.device)
    return t.device


# ------------------------------------------------------------------------------------------------ K6/K7
def gae_max_partials(T: int, L: int) -> int:
    return _lib().orl_gae_max_partials(T, L)


def gae_scan(rewards, value_preds, masks, bad_masks, next_value, vn_state, returns, gamma: float, gae_lambda: float,
             use_gae: bool, use_proper_time_limits: bool, active_masks=None, adv_raw=None, stat_partials=None) -> int:
    """In-place on ``value_preds`` (slot T) and ``returns``; returns the number of stat partial rows."""
    dev = _dev(rewards)
    T = rewards.shape[0]
    L = rewards[0].numel()
    flags = (1 if use_gae else 0) | (2 if use_proper_time_limits else 0)
    n_part = C.c_int(0)
    rc = _lib().orl_gae_scan(fptr(rewards), fptr(value_preds), fptr(masks), fptr(bad_masks), fptr(next_value),
                             fptr(vn_state), fptr(returns), T, L, float(gamma), float(gae_lambda), flags,
                             fptr(active_masks), fptr(adv_raw), ptr(stat_partials), C.byref(n_part), stream_ptr(dev))
    nat.check(rc, "orl_gae_scan")
    return n_part.value


def adv_stats(returns, value_preds, active_masks, vn_state, T: int, L: int, adv_raw, stat_partials) -> int:
    dev = _dev(returns)
    n_part = C.c_int(0)
    rc = _lib().orl_adv_stats(fptr(returns), fptr(value_preds), fptr(active_masks), fptr(vn_state), T, L,
                              fptr(adv_raw), ptr(stat_partials), C.byref(n_part), stream_ptr(dev))
    nat.check(rc, "orl_adv_stats")
    return n_part.value


def adv_normalize_pack(adv, stat_partials, n_partials: int, T: int, L: int, use_adv_normalize: bool, stats_out=None,
                       src: Optional[PackSrc] = None, records=None) -> None:
    dev = _dev(adv)
    rc = _lib().orl_adv_normalize_pack(fptr(adv), ptr(stat_partials), n_partials, T, L, int(bool(use_adv_normalize)),
                                       ptr(stats_out), C.byref(src) if src is not None else None, fptr(records),
                                       stream_ptr(dev))
    nat.check(rc, "orl_adv_normalize_pack")


# ------------------------------------------------------------------------------------------------ K5 / K8
def buffer_insert(buf: BufferPtrs, step: int, next_policy_obs, next_critic_obs, rewards, dones, bad_transition=None,
                  next_action_masks=None, h_policy_next=None, h_critic_next=None) -> None:
    """``h_*_next`` (recurrent): rnn_states[step+1] views, multiplied in place by masks[step+1] in the same launch."""
    dev = _dev(rewards)
    if dones.dtype != torch.uint8:
        raise nat.NativeError("dones must be uint8")
    if h_policy_next is None:
        rc = _lib().orl_buffer_insert(C.byref(buf), step, fptr(next_policy_obs), fptr(next_critic_obs), fptr(rewards),
                                      ptr(dones), ptr(bad_transition), fptr(next_action_masks), stream_ptr(dev))
    else:
        rc = _lib().orl_buffer_insert_rnn(C.byref(buf), step, fptr(next_policy_obs), fptr(next_critic_obs),
                                          fptr(rewards), ptr(dones), ptr(bad_transition), fptr(next_action_masks),
                                          fptr(h_policy_next), fptr(h_critic_next),
                                          int(h_policy_next.shape[-1] * (h_policy_next.shape[-2] if h_policy_next.dim() >= 4 else 1)),
                                          stream_ptr(dev))
    nat.check(rc, "orl_buffer_insert")


def multi_copy(pairs) -> None:
    """dst.copy_(src) for up to 8 (dst, src) pairs of contiguous float32 tensors in ONE launch (orl_multi_copy)."""
    d = nat.CopyDesc()
    d.count = len(pairs)
    for k, (dst, src) in enumerate(pairs):
        d.dst[k], d.src[k], d.n[k] = dst.data_ptr(), src.data_ptr(), src.numel()
    rc = _lib().orl_multi_copy(C.byref(d), stream_ptr(_dev(pairs[0][0])))
    nat.check(rc, "orl_multi_copy")


def gather_minibatch(srcs: Sequence[torch.Tensor], idx: torch.Tensor) -> List[torch.Tensor]:
    """dst[k][i, :] = srcs[k][idx[i], :] for up to 12 row-major 2-D float32 arrays in one launch."""
    dev = _dev(idx)
    if idx.dtype != torch.int64:
        raise nat.NativeError("minibatch indices must be int64")
    n = idx.numel()
    d = GatherDesc()
    outs = []
    if len(srcs) > nat.ORL_GATHER_MAX:
        raise nat.NativeError("too many arrays for one gather")
    for k, s in enumerate(srcs):
        w = s.shape[1]
        o = torch.empty((n, w), dtype=torch.float32, device=dev)
        d.src[k], d.dst[k], d.width[k] = fptr(s), fptr(o), w
        outs.append(o)
    d.count = len(srcs)
    rc = _lib().orl_gather_minibatch(C.byref(d), ptr(idx), n, stream_ptr(dev))
    nat.check(rc, "orl_gather_minibatch")
    return outs


def perm_feistel(n: int, seed: int, stream_id: int, device, vn=None) -> torch.Tensor:
    """``vn`` = (state, moments, beta): also run ValueNorm.update in the same launch (orl_perm_feistel_vn)."""
    dev = nat.require_gpu(device)
    idx = torch.empty(n, dtype=torch.int64, device=dev)
    if vn is None:
        rc = _lib().orl_perm_feistel(ptr(idx), n, seed & (2 ** 64 - 1), stream_id & (2 ** 64 - 1), stream_ptr(dev))
    else:
        rc = _lib().orl_perm_feistel_vn(ptr(idx), n, seed & (2 ** 64 - 1), stream_id & (2 ** 64 - 1), fptr(vn[0]),
                                        ptr(vn[1]), float(vn[2]), stream_ptr(dev))
    nat.check(rc, "orl_perm_feistel")
    return idx


# ------------------------------------------------------------------------------------------------ K1-K4
def act_step(pnet: NetDesc, ptheta, cnet: Optional[NetDesc], ctheta, policy_obs, critic_obs, action_masks, B: int,
             deterministic: bool, seed: int, row0: int, rng_step: int, forced_u, values, actions, logp,
             rng_step_dev=None) -> None:
    """``rng_step_dev``: optional int64 device scalar added to ``rng_step`` by the kernel (hipGraph replays)."""
    dev = _dev(policy_obs if policy_obs is not None else critic_obs)
    rc = _lib().orl_act_step(C.byref(pnet), fptr(ptheta), C.byref(cnet) if cnet is not None else None, fptr(ctheta),
                             fptr(policy_obs), fptr(critic_obs), fptr(action_masks), B, int(bool(deterministic)),
                             seed & (2 ** 64 - 1), row0, rng_step, ptr(rng_step_dev), fptr(forced_u), fptr(values),
                             fptr(actions), fptr(logp), stream_ptr(dev))
    nat.check(rc, "orl_act_step")


def act_step_grouped(pnet: NetDesc, pthetas, rows_per_group: int, policy_obs, action_masks, B: int, deterministic: bool,
                     seed: int, row0: int, rng_step: int, actions, logp, rng_step_dev=None) -> None:
    """One launch for a pool of policies: row group g is evaluated with ``pthetas[g]`` (orl_act_step_grouped)."""
    dev = _dev(policy_obs)
    rc = _lib().orl_act_step_grouped(C.byref(pnet), fptr(pthetas), pthetas.stride(0), rows_per_group, fptr(policy_obs),
                                     fptr(action_masks), B, int(bool(deterministic)), seed & (2 ** 64 - 1), row0,
                                     rng_step, ptr(rng_step_dev), fptr(actions), fptr(logp), stream_ptr(dev))
    nat.check(rc, "orl_act_step_grouped")


def act_step_pool(pnet: NetDesc, pthetas, opp_index, policy_obs, action_masks, B: int, deterministic: bool, seed: int,
                  row0: int, rng_step: int, actions, logp, rng_step_dev=None) -> None:
    """Row i is evaluated with policy ``pthetas[opp_index[i]]`` (orl_act_step_pool: exact per-row assignment)."""
    dev = _dev(policy_obs)
    assert opp_index.dtype == torch.int32
    rc = _lib().orl_act_step_pool(C.byref(pnet), fptr(pthetas), pthetas.stride(0), pthetas.shape[0], ptr(opp_index),
                                  fptr(policy_obs), fptr(action_masks), B, int(bool(deterministic)),
                                  seed & (2 ** 64 - 1), row0, rng_step, ptr(rng_step_dev), fptr(actions), fptr(logp),
                                  stream_ptr(dev))
    nat.check(rc, "orl_act_step_pool")


def opponent_sample(opp_index, dones, n_filled: int, last_slot: int, strategy: int, per_tile: bool, seed: int,
                    draw_id: int, draw_id_dev=None) -> None:
    """opp_index[n] <- a pool slot for every env with dones[n] (None: all): strategy 0 uniform over the filled slots,
    1 the newest (openrl/selfplay/sample_strategy/{random,last}_opponent.py)."""
    dev = _dev(opp_index)
    rc = _lib().orl_opponent_sample(ptr(opp_index), ptr(dones), opp_index.numel(), n_filled, last_slot, strategy,
                                    int(bool(per_tile)), seed & (2 ** 64 - 1), draw_id, ptr(draw_id_dev),
                                    stream_ptr(dev))
    nat.check(rc, "orl_opponent_sample")


def critic_values(cnet: NetDesc, ctheta, critic_obs, values) -> None:
    """values[rows] = V(critic_obs[rows, D]) in one persistent launch (orl_critic_values)."""
    dev = _dev(critic_obs)
    rows = critic_obs.numel() // cnet.obs_dim
    rc = _lib().orl_critic_values(C.byref(cnet), fptr(ctheta), fptr(critic_obs), rows, fptr(values), stream_ptr(dev))
    nat.check(rc, "orl_critic_values")


def evaluate_actions(pnet, ptheta, cnet, ctheta, policy_obs, critic_obs, actions, action_masks, active_masks, B: int,
                     values, logp, ent_rows, dist_entropy) -> None:
    dev = _dev(policy_obs)
    rc = _lib().orl_evaluate_actions(C.byref(pnet), fptr(ptheta), C.byref(cnet) if cnet is not None else None,
                                     fptr(ctheta), fptr(policy_obs), fptr(critic_obs), fptr(actions),
                                     fptr(action_masks), fptr(active_masks), B, fptr(values), fptr(logp),
                                     fptr(ent_rows), fptr(dist_entropy), stream_ptr(dev))
    nat.check(rc, "orl_evaluate_actions")


# ------------------------------------------------------------------------------------------------ K9-K14
def ppo_fwd_bwd(pnet, ptheta, cnet, ctheta, records, idx, mb: int, vn_state, hp: PPOHParams, partials):
    """Returns (policy workgroups, critic workgroups) = rows of the two partial regions."""
    dev = _dev(records)
    nb = (C.c_int * 2)(0, 0)
    rc = _lib().orl_ppo_fwd_bwd(C.byref(pnet), fptr(ptheta), C.byref(cnet), fptr(ctheta), fptr(records),
                                records.shape[1], ptr(idx), mb, fptr(vn_state), C.byref(hp), fptr(partials),
                                nb, stream_ptr(dev))
    nat.check(rc, "orl_ppo_fwd_bwd")
    return nb[0], nb[1]


def ppo_reduce(partials_ptr: int, n_blocks: int, width: int, sums_ptr: int, dev) -> None:
    rc = _lib().orl_ppo_reduce(partials_ptr, n_blocks, width, sums_ptr, stream_ptr(dev))
    nat.check(rc, "orl_ppo_reduce")


def ppo_reduce_pair(partials, nb_p: int, width_p: int, nb_c: int, width_c: int, sums, comm=None) -> None:
    """``comm`` (a ``distributed.SmallAllreduce``): also push every column sum to the peers (opens the collective
    that ``ppo_apply(..., comm=comm)`` closes)."""
    dev = _dev(sums)
    if comm is None:
        rc = _lib().orl_ppo_reduce_pair(fptr(partials), nb_p, width_p, nb_c, width_c, fptr(sums), stream_ptr(dev))
    else:
        rc = _lib().orl_ppo_reduce_pair_comm(comm.handle, fptr(partials), nb_p, width_p, nb_c, width_c, fptr(sums),
                                             stream_ptr(dev))
    nat.check(rc, "orl_ppo_reduce_pair")


def ppo_apply(pnet, cnet, sums, hp: PPOHParams, padam: AdamState, cadam: AdamState, train_info_accum,
              next_perm=None, comm=None):
    """``next_perm`` = (n, seed, stream_id, vn | None): also produce the next epoch's permutation (returned) and, with
    vn = (state, moments, beta), its ValueNorm.update, on idle workgroups of the same launch (orl_ppo_apply_perm)."""
    dev = _dev(sums)
    if comm is not None:  # multi-GPU: sum the ranks' contributions (pushed by ppo_reduce_pair) while staging them
        idx, n, seed, stream_id, vn = None, 0, 0, 0, None
        if next_perm is not None:
            n, seed, stream_id, vn = next_perm
            idx = torch.empty(n, dtype=torch.int64, device=dev)
        rc = _lib().orl_ppo_apply_comm(comm.handle, C.byref(pnet), C.byref(cnet), fptr(sums), C.byref(hp),
                                       C.byref(padam), C.byref(cadam), fptr(train_info_accum), ptr(idx), n,
                                       seed & (2 ** 64 - 1), stream_id & (2 ** 64 - 1), fptr(vn[0]) if vn else None,
                                       ptr(vn[1]) if vn else None, float(vn[2]) if vn else 0.0, stream_ptr(dev))
        nat.check(rc, "orl_ppo_apply_comm")
        return idx
    if next_perm is None:
        rc = _lib().orl_ppo_apply(C.byref(pnet), C.byref(cnet), fptr(sums), C.byref(hp), C.byref(padam),
                                  C.byref(cadam), fptr(train_info_accum), stream_ptr(dev))
        nat.check(rc, "orl_ppo_apply")
        return None
    n, seed, stream_id, vn = next_perm
    idx = torch.empty(n, dtype=torch.int64, device=dev)
    rc = _lib().orl_ppo_apply_perm(C.byref(pnet), C.byref(cnet), fptr(sums), C.byref(hp), C.byref(padam),
                                   C.byref(cadam), fptr(train_info_accum), ptr(idx), n, seed & (2 ** 64 - 1),
                                   stream_id & (2 ** 64 - 1), fptr(vn[0]) if vn else None, ptr(vn[1]) if vn else None,
                                   float(vn[2]) if vn else 0.0, stream_ptr(dev))
    nat.check(rc, "orl_ppo_apply_perm")
    return idx


def ppo_reduce_apply(partials, nb_p: int, width_p: int, nb_c: int, width_c: int, sums, pnet, cnet, hp: PPOHParams,
                     padam: AdamState, cadam: AdamState, train_info_accum, sync_ctr, next_perm=None, comm=None,
                     entry: str = "orl_ppo_reduce_apply"):
    """``ppo_reduce_pair`` + ``ppo_apply`` in ONE launch: same arguments, same results.  ``entry``: ``orl_ppo_step`` (round 6:
    designated optimiser workgroups behind write-through sums) or ``orl_ppo_reduce_apply`` (round 5's ticketed form, only in
    an ORL_BUILD_EXPERIMENTS library).  ``sync_ctr``: 4 x int32 device tensor, zero before the first call (every launch
    leaves it zero)."""
    dev = _dev(sums)
    idx, n, seed, stream_id, vn = None, 0, 0, 0, None
    if next_perm is not None:
        n, seed, stream_id, vn = next_perm
        idx = torch.empty(n, dtype=torch.int64, device=dev)
    rc = getattr(_lib(), entry)(comm.handle if comm is not None else None, fptr(partials), nb_p, width_p, nb_c,
                                width_c, fptr(sums), C.byref(pnet), C.byref(cnet), C.byref(hp), C.byref(padam),
                                C.byref(cadam), fptr(train_info_accum), ptr(idx), n, seed & (2 ** 64 - 1),
                                stream_id & (2 ** 64 - 1), fptr(vn[0]) if vn else None, ptr(vn[1]) if vn else None,
                                float(vn[2]) if vn else 0.0, ptr(sync_ctr), stream_ptr(dev))
    nat.check(rc, entry)
    return idx


def valuenorm_update(vn_state, moments, beta: float = 0.99999) -> None:
    dev = _dev(vn_state)
    rc = _lib().orl_valuenorm_update(fptr(vn_state), ptr(moments), float(beta), stream_ptr(dev))
    nat.check(rc, "orl_valuenorm_update")


def minibatch_moments(records, ret_col: int, idx, mb: int, scratch, moments) -> None:
    dev = _dev(records)
    rc = _lib().orl_minibatch_moments(fptr(records), records.shape[1], ret_col, ptr(idx), mb, ptr(scratch),
                                      ptr(moments), stream_ptr(dev))
    nat.check(rc, "orl_minibatch_moments")


# ------------------------------------------------------------------------------------------------ device envs
def env_state_width(env_kind: int) -> int:
    n = _lib().orl_env_state_width(env_kind)
    if n < 0:
        nat.check(n, "orl_env_state_width")
    return n


def env_reset(env_kind: int, env_state, ep_stats, obs0, N: int, obs_dim: int, env_seed: int, episode_limit: int) -> None:
    dev = _dev(env_state)
    rc = _lib().orl_env_reset(env_kind, fptr(env_state), fptr(ep_stats), fptr(obs0), N, obs_dim,
                              env_seed & (2 ** 64 - 1), episode_limit, stream_ptr(dev))
    nat.check(rc, "orl_env_reset")


def env_step(env_kind: int, env_state, ep_stats, actions, obs, rewards, dones, N: int, obs_dim: int, env_seed: int,
             episode_limit: int, global_step: int, global_step_dev=None) -> None:
    """``global_step_dev``: optional int64 device scalar added to ``global_step`` (mod 2^64) - see ``orl_env_step_dev``."""
    dev = _dev(env_state)
    a_w = 0 if actions is None else actions.shape[-1]
    rc = _lib().orl_env_step_dev(env_kind, fptr(env_state), fptr(ep_stats), fptr(actions), a_w, fptr(obs), fptr(rewards),
                                 ptr(dones), N, obs_dim, env_seed & (2 ** 64 - 1), episode_limit,
                                 global_step & (2 ** 64 - 1), ptr(global_step_dev), stream_ptr(dev))
    nat.check(rc, "orl_env_step")


def rollout_fused(pnet, ptheta, cnet, ctheta, args: RolloutArgs, next_value) -> None:
    dev = _dev(ptheta)
    rc = _lib().orl_rollout_fused(C.byref(pnet), fptr(ptheta), C.byref(cnet), fptr(ctheta), C.byref(args),
                                  fptr(next_value), stream_ptr(dev))
    nat.check(rc, "orl_rollout_fused")


def make_hparams(cfg, recurrent: bool = False) -> PPOHParams:
    """cfg flags -> orl_ppo_hparams (defaults: SURVEY.md section 5.6).  ``recurrent``: the GEMM-path bits of ``reserved``
    come from ``amd_rnn_gemm`` (the recurrent row kernel) instead of ``amd_tower_gemm`` (the feed-forward tower pair)."""
    tower = str(getattr(cfg, "amd_tower_gemm", "split"))
    if recurrent:
        # fp32 (default): chunks of 2 steps -> the register-resident row kernel of round 5, other lengths -> the recompute
        # kernel; fp32_recompute: the recompute kernel whatever the length; split / split_w4: round 4's streamed kernel
        rnn = "fp32" if tower == "fp32" else str(getattr(cfg, "amd_rnn_gemm", "fp32"))
        gemm_bits = {"fp32": 0, "fp32_recompute": 4, "split": 8, "split_w4": 8 | 16}[rnn]
    else:
        gemm_bits = {"fp32": 4, "split_two_image": 8}.get(tower, 0)
    return PPOHParams(clip_param=float(cfg.clip_param), entropy_coef=float(cfg.entropy_coef),
                      value_loss_coef=float(cfg.value_loss_coef), huber_delta=float(cfg.huber_delta),
                      dual_clip_coeff=float(cfg.dual_clip_coeff), max_grad_norm=float(cfg.max_grad_norm),
                      use_clipped_value_loss=int(bool(cfg.use_clipped_value_loss)),
                      use_huber_loss=int(bool(cfg.use_huber_loss)),
                      use_value_active_masks=int(bool(cfg.use_value_active_masks)),
                      use_policy_active_masks=int(bool(cfg.use_policy_active_masks)),
                      use_valuenorm=int(bool(cfg.use_valuenorm or cfg.use_popart)),
                      dual_clip_ppo=int(bool(cfg.dual_clip_ppo)), use_max_grad_norm=int(bool(cfg.use_max_grad_norm)),
                      reserved=gemm_bits)
