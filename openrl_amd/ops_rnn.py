"""Tensor-level wrappers over the recurrent (GRU) part of the C ABI (``include/orl_hip.h``, ``orl_rnn_*``).
Same contract as ``ops.py``: checks, raw pointers, current stream; nothing is computed in Python."""
from __future__ import annotations

import ctypes as C

import torch

from . import _native as nat
from ._native import AdamState, NetDesc, PPOHParams, RnnBatch, fptr, ptr, stream_ptr


def _lib():
    return nat.load()


def _dev(t: torch.Tensor):
    nat.require_gpu(t.device)
    return t.device


def rnn_param_count(net: NetDesc) -> int:
    n = _lib().orl_rnn_param_count(C.byref(net))
    if n < 0:
        nat.check(n, "orl_rnn_param_count")
    return n


def rnn_raw_grad_count(net: NetDesc) -> int:
    n = _lib().orl_rnn_raw_grad_count(C.byref(net))
    if n < 0:
        nat.check(n, "orl_rnn_raw_grad_count")
    return n


def rnn_act_step(pnet, ptheta, cnet, ctheta, policy_obs, critic_obs, h_policy_in, h_critic_in, masks, action_masks,
                 B: int, deterministic: bool, seed: int, row0: int, rng_step: int, forced_u, values, actions, logp,
                 h_policy_out, h_critic_out, rng_step_dev=None) -> None:
    dev = _dev(masks)
    rc = _lib().orl_rnn_act_step(C.byref(pnet) if pnet is not None else None, fptr(ptheta),
                                 C.byref(cnet) if cnet is not None else None, fptr(ctheta), fptr(policy_obs),
                                 fptr(critic_obs), fptr(h_policy_in), fptr(h_critic_in), fptr(masks),
                                 fptr(action_masks), B, int(bool(deterministic)), seed & (2 ** 64 - 1), row0, rng_step,
                                 nat.ptr(rng_step_dev), fptr(forced_u), fptr(values), fptr(actions), fptr(logp), fptr(h_policy_out),
                                 fptr(h_critic_out), stream_ptr(dev))
    nat.check(rc, "orl_rnn_act_step")


def rnn_eval_step(pnet, ptheta, cnet, ctheta, policy_obs, critic_obs, h_policy_in, h_critic_in, masks, action_masks,
                  actions, B: int, values, logp, entropy, h_policy_out, h_critic_out) -> None:
    """One recurrent step of evaluate_actions: log-probs / entropy of the GIVEN actions, values, new hidden states."""
    dev = _dev(masks)
    rc = _lib().orl_rnn_eval_step(C.byref(pnet) if pnet is not None else None, fptr(ptheta),
                                  C.byref(cnet) if cnet is not None else None, fptr(ctheta), fptr(policy_obs),
                                  fptr(critic_obs), fptr(h_policy_in), fptr(h_critic_in), fptr(masks),
                                  fptr(action_masks), fptr(actions), B, fptr(values), fptr(logp), fptr(entropy),
                                  fptr(h_policy_out), fptr(h_critic_out), stream_ptr(dev))
    nat.check(rc, "orl_rnn_eval_step")


def rnn_chunk_rows(chunk_idx, n_chunks: int, L: int, T: int, lanes: int, rows: torch.Tensor) -> None:
    dev = _dev(rows)
    assert rows.dtype == torch.int64 and rows.numel() >= n_chunks * L
    rc = _lib().orl_rnn_chunk_rows(ptr(chunk_idx), n_chunks, L, T, lanes, ptr(rows), stream_ptr(dev))
    nat.check(rc, "orl_rnn_chunk_rows")


def rnn_workspace_floats(pnet, cnet, n_chunks: int, L: int) -> int:
    n = _lib().orl_rnn_workspace_floats(C.byref(pnet), C.byref(cnet), n_chunks, L)
    if n < 0:
        nat.check(int(n), "orl_rnn_workspace_floats")
    return int(n)


def rnn_chunk_rows_v3(chunk_idx, n_chunks: int, L: int, T: int, n_envs: int, n_agents: int, agent0_only: bool,
                      rows: torch.Tensor) -> None:
    """recurrent_generator_v3 rows (agent axis kept): rows[l, i*A + a] (or rows[l, i] for agent 0 only)."""
    dev = _dev(rows)
    assert rows.dtype == torch.int64 and rows.numel() >= n_chunks * L * (1 if agent0_only else n_agents)
    rc = _lib().orl_rnn_chunk_rows_v3(ptr(chunk_idx), n_chunks, L, T, n_envs, n_agents, int(bool(agent0_only)), ptr(rows),
                                      stream_ptr(dev))
    nat.check(rc, "orl_rnn_chunk_rows_v3")


def rnn_jrpo_records(records, records_out, Dp: int, Dc: int, a_w: int, rows, n_chunks: int, L: int, n_agents: int,
                     logp_new) -> None:
    dev = _dev(records)
    rc = _lib().orl_rnn_jrpo_records(fptr(records), fptr(records_out), records.shape[1], Dp, Dc, a_w, ptr(rows), n_chunks,
                                     L, n_agents, fptr(logp_new), stream_ptr(dev))
    nat.check(rc, "orl_rnn_jrpo_records")


def rnn_ppo_fwd_bwd(pnet, ptheta, cnet, ctheta, records, rows, masks, h_policy, h_critic, n_chunks: int, L: int,
                    vn_state, hp: PPOHParams, workspace, sums, rows_critic=None, n_chunks_critic: int = 0) -> None:
    """``rows_critic`` / ``n_chunks_critic``: the critic tower's own sequences (joint-action loss: agent 0 only)."""
    dev = _dev(records)
    b = RnnBatch(fptr(records), ptr(rows), fptr(masks), fptr(h_policy), fptr(h_critic), records.shape[1], n_chunks, L,
                 n_chunks_critic, ptr(rows_critic))
    rc = _lib().orl_rnn_ppo_fwd_bwd(C.byref(pnet), fptr(ptheta), C.byref(cnet), fptr(ctheta), C.byref(b),
                                    fptr(vn_state), C.byref(hp), fptr(workspace), fptr(sums), stream_ptr(dev))
    nat.check(rc, "orl_rnn_ppo_fwd_bwd")


def rnn_ppo_apply(pnet, cnet, sums, hp: PPOHParams, padam: AdamState, cadam: AdamState, train_info_accum,
                  scratch) -> None:
    dev = _dev(sums)
    rc = _lib().orl_rnn_ppo_apply(C.byref(pnet), C.byref(cnet), fptr(sums), C.byref(hp), C.byref(padam),
                                  C.byref(cadam), fptr(train_info_accum), fptr(scratch), stream_ptr(dev))
    nat.check(rc, "orl_rnn_ppo_apply")


# ------------------------------------------------------------------------------------------------ MPE device env
def mpe_state_width() -> int:
    return _lib().orl_mpe_state_width()


def mpe_reset(env_state, ep_stats, obs_policy, obs_critic, N: int, seed: int) -> None:
    dev = _dev(env_state)
    rc = _lib().orl_mpe_reset(fptr(env_state), fptr(ep_stats), fptr(obs_policy), fptr(obs_critic), N,
                              seed & (2 ** 64 - 1), stream_ptr(dev))
    nat.check(rc, "orl_mpe_reset")


def mpe_step(env_state, ep_stats, actions, obs_policy, obs_critic, rewards, dones, N: int, seed: int,
             world_length: int) -> None:
    dev = _dev(env_state)
    rc = _lib().orl_mpe_step(fptr(env_state), fptr(ep_stats), fptr(actions), fptr(obs_policy), fptr(obs_critic),
                             fptr(rewards), ptr(dones), N, seed & (2 ** 64 - 1), world_length, stream_ptr(dev))
    nat.check(rc, "orl_mpe_step")


def rnn_rollout_fused(pnet, ptheta, cnet, ctheta, args: "nat.RnnRolloutArgs", device) -> None:
    """The whole recurrent actor_rollout on the device MPE env: one policy + env launch, one critic sweep."""
    rc = _lib().orl_rnn_rollout_fused(C.byref(pnet), fptr(ptheta), C.byref(cnet), fptr(ctheta), C.byref(args),
                                      stream_ptr(nat.require_gpu(device)))
    nat.check(rc, "orl_rnn_rollout_fused")


# ------------------------------------------------------------------------------------------------ device tic-tac-toe
def ttt_state_width() -> int:
    return int(_lib().orl_ttt_state_width())


def ttt_reset(env_state, ep_stats, obs, action_masks, N: int, seed: int) -> None:
    dev = _dev(env_state)
    rc = _lib().orl_ttt_reset(fptr(env_state), fptr(ep_stats), fptr(obs), fptr(action_masks), N, seed & (2 ** 64 - 1),
                              stream_ptr(dev))
    nat.check(rc, "orl_ttt_reset")


def ttt_step(env_state, ep_stats, actions, obs, action_masks, rewards, dones, N: int, seed: int) -> None:
    dev = _dev(env_state)
    rc = _lib().orl_ttt_step(fptr(env_state), fptr(ep_stats), fptr(actions), fptr(obs), fptr(action_masks),
                             fptr(rewards), ptr(dones), N, seed & (2 ** 64 - 1), stream_ptr(dev))
    nat.check(rc, "orl_ttt_step")


def ttt_agent_move(env_state, actions, opp_obs, opp_masks, rewards, dones, N: int) -> None:
    dev = _dev(env_state)
    rc = _lib().orl_ttt_agent_move(fptr(env_state), fptr(actions), fptr(opp_obs), fptr(opp_masks), fptr(rewards),
                                   ptr(dones), N, stream_ptr(dev))
    nat.check(rc, "orl_ttt_agent_move")


def ttt_opponent_move(env_state, ep_stats, opp_actions, obs, action_masks, rewards, dones, N: int, seed: int) -> None:
    dev = _dev(env_state)
    rc = _lib().orl_ttt_opponent_move(fptr(env_state), fptr(ep_stats), fptr(opp_actions), fptr(obs), fptr(action_masks),
                                      fptr(rewards), ptr(dones), N, seed & (2 ** 64 - 1), stream_ptr(dev))
    nat.check(rc, "orl_ttt_opponent_move")
