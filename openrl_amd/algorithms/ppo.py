"""``PPOAlgorithm`` - the PPO update (``openrl/algorithms/ppo.py:32-469``) as a short chain of fused
HIP launches per minibatch:

    train_ppo (ppo.py:383-458)
      advantages + nan-stats normalisation (:384-409)      -> orl_gae_scan's fused stats / orl_adv_stats
                                                              + orl_adv_normalize_pack (also packs records)
      for epoch, for minibatch (:424-451):
        minibatch order (replay_data.py:553-580)            -> host torch.randperm (bit-exact, default)
                                                              or orl_perm_feistel (device, amd_perm_mode)
        ValueNorm.update(return_batch) (:190-191)           -> orl_minibatch_moments + orl_valuenorm_update
        prepare_loss + 2x backward (:98-124)                -> orl_ppo_fwd_bwd   (one launch per tower)
        [multi-GPU] sum over ranks                          -> orl_ppo_reduce + ONE all-reduce (RCCL)
        clip_grad_norm_ x2 + Adam x2 + .item() x4 (:132-164, :445-451) -> orl_ppo_apply (stats stay on device)

The six ``train_info`` scalars are accumulated on the device and read back ONCE per ``train`` call.
"""
from __future__ import annotations

from typing import Dict, Union

import numpy as np
import torch

from .. import _native as nat
from .. import distributed as dist_utils
from .. import ops, ops_rnn
from .base_algorithm import BaseAlgorithm

INFO_KEYS = ("value_loss", "policy_loss", "dist_entropy", "actor_grad_norm", "critic_grad_norm", "ratio")


class DeviceTrainInfo(dict):
    """The train_info dict of ``PPOAlgorithm.train`` (ppo.py:445-458) whose values are still on the device: the one
    device->host copy happens on first access (any read of the dict), not at the end of every update."""

    def __init__(self, keys, device_values: torch.Tensor, after_sync=None, scale=None):
        """``scale``: per-key fp32 factors applied on the HOST at read time (the 1 / num_updates of ppo.py:453-456) - the
        device tensor holds the raw sums and belongs to this dict alone (every train() call accumulates into a fresh
        one), so no multiply launch is needed to snapshot it."""
        super().__init__()
        self._keys, self._dev, self._after_sync, self._scale = tuple(keys), device_values, after_sync, scale

    def _materialize(self) -> None:
        if self._dev is not None:
            raw, self._dev = self._dev.cpu().numpy(), None
            if self._scale is not None:  # fp32 product, bit-identical to the device multiply it replaces
                raw = raw * np.asarray(self._scale, dtype=np.float32)
            vals = raw.tolist()
            if self._after_sync is not None:  # e.g. the collective's error word, posted behind this update's launches
                cb, self._after_sync = self._after_sync, None
                cb()
            for k, v in zip(self._keys, vals):
                super().__setitem__(k, v)

    def __getitem__(self, k):
        self._materialize()
        return super().__getitem__(k)

    def __iter__(self):
        self._materialize()
        return super().__iter__()

    def __len__(self):
        self._materialize()
        return super().__len__()

    def __contains__(self, k):
        self._materialize()
        return super().__contains__(k)

    def __repr__(self):
        self._materialize()
        return super().__repr__()

    def __eq__(self, other):
        self._materialize()
        return super().__eq__(other)

    def get(self, k, default=None):
        self._materialize()
        return super().get(k, default)

    def keys(self):
        self._materialize()
        return super().keys()

    def values(self):
        self._materialize()
        return super().values()

    def items(self):
        self._materialize()
        return super().items()

    def copy(self):
        self._materialize()
        return dict(self)

    def pop(self, *a):
        self._materialize()
        return super().pop(*a)

    def update(self, *a, **k):
        self._materialize()
        return super().update(*a, **k)

    def __setitem__(self, k, v):
        self._materialize()
        super().__setitem__(k, v)

    def __delitem__(self, k):
        self._materialize()
        super().__delitem__(k)


class PPOAlgorithm(BaseAlgorithm):
    #: train_info keys in the order of the device accumulator (a prefix of INFO_KEYS; A2C drops the trailing "ratio")
    info_keys = INFO_KEYS

    def __init__(self, cfg, init_module, agent_num: int = 1, device: Union[str, torch.device] = "cuda:0") -> None:
        if cfg.use_deepspeed or cfg.use_amp:
            raise NotImplementedError("deepspeed / amp are not built in this engine")
        if cfg.use_joint_action_loss and not cfg.use_recurrent_policy:
            raise NotImplementedError("use_joint_action_loss is built for recurrent policies (recurrent_generator_v3, "
                                      "the only generator the reference pairs it with, ppo.py:363-372)")
        super().__init__(cfg, init_module, agent_num, device)
        # use_naive_recurrent_policy (get_data_generator, ppo.py:365-372): naive_recurrent_generator samples whole
        # trajectories per (env, agent) lane with randperm(lanes) - exactly recurrent_generator's chunks of length
        # episode_length (one chunk per lane, the same permutation stream, the same (L, Nc) row order), so it runs
        # through the chunked update with L = T (set per update from the buffer)
        self.naive_recurrent = bool(cfg.use_naive_recurrent_policy) and not bool(cfg.use_recurrent_policy)
        self.train_list = [self.train_ppo]
        self.hp = ops.make_hparams(cfg, recurrent=bool(getattr(self.algo_module, "recurrent", False)))
        self.generic = bool(getattr(self.algo_module, "generic", False))
        self.use_joint_action_loss = bool(cfg.use_joint_action_loss)
        self.perm_mode = getattr(cfg, "amd_perm_mode", "reference")
        self._perm_counter = 0
        self._vn_in_perm = False
        self.last_indices = None
        self._info = torch.zeros(8, dtype=torch.float32, device=self.device)
        self._moments = torch.zeros(3, dtype=torch.float64, device=self.device)
        self._mom_scratch = torch.zeros(512, dtype=torch.float64, device=self.device)
        self._adv_stats = torch.zeros(11, dtype=torch.float64, device=self.device)
        self._moments_mb = self._moments
        self._full_batch_moments = False
        if self.generic:  # general towers (modules/generic_net.py): layer-wise update, algorithms/ppo_generic.py
            self.recurrent = bool(getattr(self.algo_module, "recurrent", False))
            self._rnn_rows = None
            self.fuse_next_perm = False
            self._comm = None
            self._gen_den4 = torch.zeros(4, dtype=torch.float32, device=self.device)  # [policy den | value den (JRPO)]
            self._gen_den = self._gen_den4[:2]
            self._gen_den_c = self._gen_den4[2:]
            self._gen_den_scratch = torch.zeros(257, dtype=torch.float32, device=self.device)
            self._gen_scratch = torch.zeros(256, dtype=torch.float32, device=self.device)
            self._gen_flat = self._gen_sums = None
            self._gen_den_stamp = None
            self._comm_watch = None
            if self.world_size > 1:
                # multi-GPU: ONE flat vector per optimiser step - every network's gradient and the
                # two 20-float loss / logging sum rows live in one allocation (the networks' ``grad`` become views of it),
                # summed over ranks by one collective after the backward passes
                nets = [m for m in self.algo_module.models.values()]
                n = sum(int(m.n_params) for m in nets) + 40
                self._gen_flat = torch.zeros(n, dtype=torch.float32, device=self.device)
                o = 0
                for m in nets:
                    m.grad = self._gen_flat[o:o + int(m.n_params)]
                    o += int(m.n_params)
                self._gen_sums = (self._gen_flat[o:o + 20], self._gen_flat[o + 20:o + 40])
                # the one-shot xGMI push is built for the latency regime (every rank pushes its whole vector to every
                # peer, 8-byte tagged granules, <= 32 blocks): above P2P_MAX_FLOATS the vector goes through RCCL, whose
                # ring is bandwidth-optimal there
                self._comm = dist_utils.make_small_allreduce(n, self.device, getattr(cfg, "amd_collective", "p2p")) \
                    if n <= self.P2P_MAX_FLOATS else None
                if self._comm is not None:
                    self._comm_watch = nat.DeviceErrorWatch("orl_comm: a peer's contribution did not arrive within 10 s - "
                                                            "the optimiser step ran on a partial gradient sum")
            return
        p, c = self.algo_module.models["policy"], self.algo_module.models["critic"]
        self.recurrent = bool(getattr(self.algo_module, "recurrent", False))
        raw = ops_rnn.rnn_raw_grad_count if self.recurrent else ops.raw_grad_count
        self._raw_p = raw(p.net) + ops.N_STATS
        self._raw_c = raw(c.net) + ops.N_STATS
        mb = ops.ppo_max_blocks()
        dev = self.device
        self._partials = None if self.recurrent else torch.empty(mb * (self._raw_p + self._raw_c), dtype=torch.float32,
                                                                 device=dev)
        self._rnn_ws = None      # activation tapes + partials of the recurrent update, sized on first use
        self._rnn_rows = None    # [L, n_chunks] record rows of the current minibatch
        self._rnn_scratch = torch.zeros(512, dtype=torch.float32, device=dev) if self.recurrent else None
        self._sums = torch.zeros(self._raw_p + self._raw_c, dtype=torch.float32, device=dev)
        # reduce + optimiser step of an MLP-tower minibatch in two launches (default) or in one (orl_ppo_reduce_apply:
        # built, bit-identical, 1 % slower - the comparison switch amd_optim_step)
        self._optim_step = str(getattr(cfg, "amd_optim_step", "two_launch"))
        self._fused_step = self._optim_step in ("fused", "step")
        self._sync_ctr = torch.zeros(4, dtype=torch.int32, device=dev)  # its tickets; every launch leaves them zero
        # multi-GPU: the one-shot xGMI all-reduce of the sums vector, fused into the optimiser-step launches (MLP towers)
        # or as its own launch (recurrent); None = single process or amd_collective=rccl -> torch.distributed
        self._comm = dist_utils.make_small_allreduce(self._sums.numel(), dev, getattr(cfg, "amd_collective", "p2p")) \
            if self.world_size > 1 else None
        self.fuse_next_perm = True  # device permutation of epoch e+1 rides in epoch e's optimiser-step launch
        self._comm_watch = nat.DeviceErrorWatch("orl_comm: a peer's contribution did not arrive within 10 s - the "
                                                "optimiser step ran on a partial gradient sum") \
            if self._comm is not None else None

    #: largest flat gradient vector (floats) the one-shot P2P all-reduce carries; wider general towers use RCCL
    P2P_MAX_FLOATS = 1 << 16

    def _allreduce_vec(self, t: torch.Tensor) -> torch.Tensor:
        """In-place SUM of a small fp32 vector over the ranks: the one-shot xGMI push when the comm is up, else
        torch.distributed (RCCL)."""
        if self.world_size > 1:
            if self._comm is not None and t.dtype == torch.float32 and t.numel() <= self._comm.capacity:
                self._comm.allreduce_(t)
            else:
                dist_utils.allreduce_(t)
        return t

    # ------------------------------------------------------------------------------------------ advantages
    def _advantages_and_records(self, buffer) -> None:
        T, N, A = buffer.episode_length, buffer.n_rollout_threads, buffer.num_agents
        L = N * A
        vn = self.algo_module.get_critic_value_normalizer() if (self._use_popart or self._use_valuenorm) else None
        vn_state = vn.state if vn is not None else None
        if not (buffer._adv_fresh and (getattr(buffer, "_adv_vn", None) is vn_state)):
            buffer.n_partials = ops.adv_stats(buffer.returns, buffer.value_preds, buffer.active_masks, vn_state, T, L,
                                              buffer.advantages, buffer.stat_partials)
        partials, n_part = buffer.stat_partials, buffer.n_partials
        if self.world_size > 1:
            # global statistics (ppo.py:405-409 is over the WHOLE batch): sum the 8 doubles over ranks
            row = dist_utils.allreduce_stat_rows(partials[:n_part])
            partials, n_part = row, 1
        ops.adv_normalize_pack(buffer.advantages, partials, n_part, T, L, self._use_adv_normalize, self._adv_stats,
                               buffer.pack_src(), buffer.ensure_records())
        buffer._adv_fresh = False  # advantages are now normalised in place

    # ------------------------------------------------------------------------------------------ one minibatch
    def _update_minibatch(self, buffer, idx, mb: int, turn_on: bool, next_perm=None):
        if self.generic:
            from .ppo_generic import update_minibatch_generic

            return update_minibatch_generic(self, buffer, idx, mb, turn_on)
        mod = self.algo_module
        p, c = mod.models["policy"], mod.models["critic"]
        po, co = mod.optimizers["policy"], mod.optimizers["critic"]
        rec = buffer.records
        vn = mod.get_critic_value_normalizer() if self._use_valuenorm else None
        vn_state = None
        if vn is not None:
            if self._full_batch_moments:
                # one minibatch == the whole batch: sum(ret), sum(ret^2), count are the (already globally
                # reduced) return statistics the GAE kernel produced - no extra pass over the rows
                pass
            else:
                ret_col = buffer.Dp + buffer.Dc + 2 * buffer.act_shape + 2
                ops.minibatch_moments(rec, ret_col, idx, mb, self._mom_scratch, self._moments)
                if self.world_size > 1:
                    dist_utils.allreduce_(self._moments)
            if self._vn_in_perm:
                self._vn_in_perm = False  # this epoch's update already ran inside the permutation launch
            else:
                ops.valuenorm_update(vn.state, self._moments, vn.beta)  # BEFORE normalize (ppo.py:190-195)
            vn_state = vn.state
        ev = getattr(self, "profile_events", None)
        if ev is not None:  # bench.py: HIP events on the launch stream around the dominant kernel pair
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        nb_p, nb_c = ops.ppo_fwd_bwd(p.net, p.theta, c.net, c.theta, rec, idx, mb, vn_state, self.hp, self._partials)
        if ev is not None:
            e1.record()
            ev.append((e0, e1))
        # ONE small collective per optimiser step: pushed by the reduce launch, summed by the apply
        # launch (orl_comm), or one RCCL all-reduce between them
        comm = self._comm
        fused = self._fused_step and (self.world_size == 1 or comm is not None)  # an RCCL all-reduce sits between the two
        if not fused:
            ops.ppo_reduce_pair(self._partials, nb_p, self._raw_p, nb_c, self._raw_c, self._sums, comm=comm)
            if self.world_size > 1 and comm is None:
                dist_utils.allreduce_(self._sums)
        po.step_count += 1 if turn_on else 0
        co.step_count += 1
        hp = self.hp
        first = getattr(self, "_info_first", False)
        if not turn_on or first:
            hp = nat.PPOHParams.from_buffer_copy(self.hp)
            if not turn_on:
                hp.reserved |= 1  # critic-only update (construct_loss_list, ppo.py:226-236)
            if first:
                hp.reserved |= 32  # first optimiser step of this train() call: the apply launch starts the averages
                self._info_first = False
        if fused:
            return ops.ppo_reduce_apply(self._partials, nb_p, self._raw_p, nb_c, self._raw_c, self._sums, p.net, c.net, hp,
                                        po.native_state(max(po.step_count, 1)), co.native_state(co.step_count),
                                        self._info, self._sync_ctr, next_perm, comm=comm,
                                        entry="orl_ppo_reduce_apply" if self._optim_step == "fused" else "orl_ppo_step")
        return ops.ppo_apply(p.net, c.net, self._sums, hp, po.native_state(max(po.step_count, 1)),
                             co.native_state(co.step_count), self._info, next_perm, comm=comm)

    # ------------------------------------------------------------------------------------------ recurrent
    def _update_minibatch_jrpo(self, buffer, chunks, n_chunks: int, turn_on: bool) -> None:
        """One recurrent_generator_v3 minibatch with the joint-action loss (JRPO: ppo.py:254-300,
        replay_data.py:425-551).  The joint ratio over agents needs every agent's CURRENT log-prob before any row's
        loss can be evaluated, so the step is: (1) forward-only evaluation of the minibatch's policy sequences
        (``orl_rnn_eval_step`` per chunk step), (2) ``orl_rnn_jrpo_records`` folds the joint log-ratio, agent 0's
        advantage (x A) and agent 0's active mask into a copy of the records, (3) the ordinary fused recurrent update on
        that copy - policy tower over all (chunk, agent) sequences, critic tower over agent 0's only."""
        if self.generic:
            return self._update_minibatch_jrpo_generic(buffer, chunks, n_chunks, turn_on)
        mod = self.algo_module
        p, c = mod.models["policy"], mod.models["critic"]
        po, co = mod.optimizers["policy"], mod.optimizers["critic"]
        T, N, A, L = buffer.episode_length, buffer.n_rollout_threads, buffer.num_agents, self.data_chunk_length
        lanes, H = N * A, p.net.hidden
        rec = buffer.records
        Dp, Dc, a_w, K = buffer.Dp, buffer.Dc, buffer.act_shape, buffer.K
        ns = n_chunks * A
        dev = self.device
        if getattr(self, "_jr", None) is None or self._jr["ns"] < ns or self._jr["L"] != L:
            f = lambda *sh: torch.empty(*sh, dtype=torch.float32, device=dev)
            self._jr = dict(ns=ns, L=L, rows_p=torch.empty(L * ns, dtype=torch.int64, device=dev),
                            rows_c=torch.empty(L * n_chunks, dtype=torch.int64, device=dev), x=f(ns, Dp), act=f(ns, a_w),
                            mk=f(ns), am=f(ns, K) if K else None, h=f(ns, H), logp=f(L * ns, a_w), rec2=torch.empty_like(rec))
        j = self._jr
        rows_p, rows_c = j["rows_p"][:L * ns], j["rows_c"][:L * n_chunks]
        ops_rnn.rnn_chunk_rows_v3(chunks, n_chunks, L, T, N, A, False, rows_p)
        ops_rnn.rnn_chunk_rows_v3(chunks, n_chunks, L, T, N, A, True, rows_c)
        vn = mod.get_critic_value_normalizer() if self._use_valuenorm else None
        vn_state = None
        if vn is not None:  # ValueNorm.update(return_batch): the returns of agent 0's rows (to_single_np, ppo.py:258)
            ret_col = Dp + Dc + 2 * a_w + 2
            ops.minibatch_moments(rec, ret_col, rows_c, n_chunks * L, self._mom_scratch, self._moments_mb)
            if self.world_size > 1:
                dist_utils.allreduce_(self._moments_mb)
            ops.valuenorm_update(vn.state, self._moments_mb, vn.beta)
            self._vn_in_perm = False
            vn_state = vn.state
        # (1) current log-probs of every (step, chunk, agent) row
        from .. import ops_gen

        x, act, mk, am, h = j["x"][:ns], j["act"][:ns], j["mk"][:ns], (j["am"][:ns] if K else None), j["h"][:ns]
        o_act = Dp + Dc
        masks_flat = buffer.masks.view(-1, 1)
        hflat = buffer.rnn_states.view(-1, H)
        ops_gen.gather_cols(hflat, 0, H, rows_p[:ns], ns, h)  # the state entering each sequence (replay_data.py:519)
        for s in range(L):
            r = rows_p[s * ns:(s + 1) * ns]
            ops_gen.gather_cols(rec, 0, Dp, r, ns, x)
            ops_gen.gather_cols(rec, o_act, a_w, r, ns, act)
            ops_gen.gather_cols(masks_flat, 0, 1, r, ns, mk.view(ns, 1))
            if K:
                ops_gen.gather_cols(rec, o_act + 2 * a_w + 4, K, r, ns, am)
            ops_rnn.rnn_eval_step(p.net, p.theta, None, None, x, None, h, None, mk, am, act, ns, None,
                                  j["logp"][s * ns:(s + 1) * ns], None, h, None)
        # (2) joint-ratio records
        rec2 = j["rec2"]
        if self._jr.get("stamp") is not buffer.records or self._jr.get("fresh") != self._jrpo_epoch_id:
            rec2.copy_(rec)
            self._jr["stamp"], self._jr["fresh"] = buffer.records, self._jrpo_epoch_id
        ops_rnn.rnn_jrpo_records(rec, rec2, Dp, Dc, a_w, rows_p, n_chunks, L, A, j["logp"])
        # (3) fused recurrent update on the adjusted records
        need = ops_rnn.rnn_workspace_floats(p.net, c.net, ns, L)
        if self._rnn_ws is None or self._rnn_ws.numel() < need:
            self._rnn_ws = torch.empty(need, dtype=torch.float32, device=dev)
        ops_rnn.rnn_ppo_fwd_bwd(p.net, p.theta, c.net, c.theta, rec2, rows_p, buffer.masks, buffer.rnn_states,
                                buffer.rnn_states_critic, ns, L, vn_state, self.hp, self._rnn_ws, self._sums,
                                rows_critic=rows_c, n_chunks_critic=n_chunks)
        if self.world_size > 1:
            if self._comm is not None:
                self._comm.allreduce_(self._sums)
            else:
                dist_utils.allreduce_(self._sums)
        po.step_count += 1 if turn_on else 0
        co.step_count += 1
        hp = self.hp
        if not turn_on:
            hp = nat.PPOHParams.from_buffer_copy(self.hp)
            hp.reserved |= 1
        ops_rnn.rnn_ppo_apply(p.net, c.net, self._sums, hp, po.native_state(max(po.step_count, 1)),
                              co.native_state(co.step_count), self._info, self._rnn_scratch)

    def _update_minibatch_jrpo_generic(self, buffer, chunks, n_chunks: int, turn_on: bool) -> None:
        """The joint-action loss on recurrent GENERAL towers: same three steps as ``_update_minibatch_jrpo`` - current
        log-probs of every (step, chunk, agent) row, ``orl_rnn_jrpo_records``, then the layer-wise update with the critic
        on agent 0's rows."""
        from .. import ops_gen
        from .ppo_generic import update_minibatch_generic

        mod = self.algo_module
        T, N, A, L = buffer.episode_length, buffer.n_rollout_threads, buffer.num_agents, self.data_chunk_length
        H = mod.policy_net.state_w * mod.policy_net.recurrent_N
        rec = buffer.records
        Dp, Dc, a_w, K = buffer.Dp, buffer.Dc, buffer.act_shape, buffer.K
        ns, dev = n_chunks * A, self.device
        if getattr(self, "_jr", None) is None or self._jr["ns"] < ns or self._jr["L"] != L:
            f = lambda *sh: torch.empty(*sh, dtype=torch.float32, device=dev)
            self._jr = dict(ns=ns, L=L, rows_p=torch.empty(L * ns, dtype=torch.int64, device=dev),
                            rows_c=torch.empty(L * n_chunks, dtype=torch.int64, device=dev), x=f(L * ns, Dp),
                            act=f(L * ns, a_w), am=f(L * ns, K) if K else None, rec2=torch.empty_like(rec))
        j = self._jr
        rows_p, rows_c = j["rows_p"][:L * ns], j["rows_c"][:L * n_chunks]
        ops_rnn.rnn_chunk_rows_v3(chunks, n_chunks, L, T, N, A, False, rows_p)
        ops_rnn.rnn_chunk_rows_v3(chunks, n_chunks, L, T, N, A, True, rows_c)
        vn = mod.get_critic_value_normalizer() if self._use_valuenorm else None
        if vn is not None:  # ValueNorm.update(return_batch): the returns of agent 0's rows (to_single_np, ppo.py:258)
            ret_col = Dp + Dc + 2 * a_w + 2
            ops.minibatch_moments(rec, ret_col, rows_c, n_chunks * L, self._mom_scratch, self._moments_mb)
            if self.world_size > 1:
                dist_utils.allreduce_(self._moments_mb)
            ops.valuenorm_update(vn.state, self._moments_mb, vn.beta)
            self._vn_in_perm = False
        # (1) current log-probs of every (step, chunk, agent) row
        B = L * ns
        x, act, am = j["x"][:B], j["act"][:B], (j["am"][:B] if K else None)
        o_act = Dp + Dc
        ops_gen.gather_cols(rec, 0, Dp, rows_p, B, x)
        ops_gen.gather_cols(rec, o_act, a_w, rows_p, B, act)
        if K:
            ops_gen.gather_cols(rec, o_act + 2 * a_w + 4, K, rows_p, B, am)
        if j.get("h0") is None or j["h0"].shape != (ns, H):
            j["h0"], j["mk"] = torch.empty(ns, H, dtype=torch.float32, device=dev), torch.empty(L * ns, dtype=torch.float32, device=dev)
        h0, mk = j["h0"], j["mk"][:B]
        ops_gen.gather_cols(buffer.rnn_states.view(-1, H), 0, H, rows_p[:ns], ns, h0)
        ops_gen.gather_cols(buffer.masks.view(-1, 1), 0, 1, rows_p, B, mk.view(B, 1))
        _, logp, _, _ = mod._evaluate_actions_rnn(None, x, h0, None, act, mk, am, None)
        # (2) joint-ratio records
        rec2 = j["rec2"]
        if j.get("stamp") is not buffer.records or j.get("fresh") != self._jrpo_epoch_id:
            rec2.copy_(rec)
            j["stamp"], j["fresh"] = buffer.records, self._jrpo_epoch_id
        ops_rnn.rnn_jrpo_records(rec, rec2, Dp, Dc, a_w, rows_p, n_chunks, L, A, logp)
        # (3) the layer-wise recurrent update on the adjusted records; the critic sees agent 0's sequences
        update_minibatch_generic(self, buffer, rows_p, B, turn_on, rnn=(L, ns), jrpo=(rec2, rows_c, n_chunks))

    def _update_minibatch_rnn(self, buffer, chunks, n_chunks: int, turn_on: bool) -> None:
        """One recurrent_generator minibatch (replay_data.py:1062-1258): ``chunks`` = chunk ids (device int64)."""
        if self.use_joint_action_loss:
            return self._update_minibatch_jrpo(buffer, chunks, n_chunks, turn_on)
        if self.generic:  # general towers with a GRU: the layer-wise update over the chunks' rows in [L, n_chunks] order
            from .ppo_generic import update_minibatch_generic

            T, lanes, L = buffer.episode_length, buffer.n_rollout_threads * buffer.num_agents, self.data_chunk_length
            if self._rnn_rows is None or self._rnn_rows.numel() < n_chunks * L:
                self._rnn_rows = torch.empty(n_chunks * L, dtype=torch.int64, device=self.device)
            rows = self._rnn_rows[:n_chunks * L]
            ops_rnn.rnn_chunk_rows(chunks, n_chunks, L, T, lanes, rows)
            return update_minibatch_generic(self, buffer, rows, n_chunks * L, turn_on, rnn=(L, n_chunks))
        mod = self.algo_module
        p, c = mod.models["policy"], mod.models["critic"]
        po, co = mod.optimizers["policy"], mod.optimizers["critic"]
        T, lanes, L = buffer.episode_length, buffer.n_rollout_threads * buffer.num_agents, self.data_chunk_length
        rec = buffer.records
        if self._rnn_rows is None or self._rnn_rows.numel() < n_chunks * L:
            self._rnn_rows = torch.empty(n_chunks * L, dtype=torch.int64, device=self.device)
        rows = self._rnn_rows
        ops_rnn.rnn_chunk_rows(chunks, n_chunks, L, T, lanes, rows)
        vn = mod.get_critic_value_normalizer() if self._use_valuenorm else None
        vn_state = None
        if vn is not None:
            if self._full_batch_moments:
                pass  # every chunk is in the minibatch: the GAE pass already summed these returns (train_ppo)
            else:
                ret_col = buffer.Dp + buffer.Dc + 2 * buffer.act_shape + 2
                ops.minibatch_moments(rec, ret_col, rows, n_chunks * L, self._mom_scratch, self._moments)
                if self.world_size > 1:
                    dist_utils.allreduce_(self._moments)
            if self._vn_in_perm:
                self._vn_in_perm = False  # ran inside the permutation launch
            else:
                ops.valuenorm_update(vn.state, self._moments, vn.beta)
            vn_state = vn.state
        need = ops_rnn.rnn_workspace_floats(p.net, c.net, n_chunks, L)
        if self._rnn_ws is None or self._rnn_ws.numel() < need:
            self._rnn_ws = torch.empty(need, dtype=torch.float32, device=self.device)
        ev = getattr(self, "profile_events", None)
        if ev is not None:  # benchmarks: HIP events on the launch stream around the recurrent forward + backward
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        ops_rnn.rnn_ppo_fwd_bwd(p.net, p.theta, c.net, c.theta, rec, rows, buffer.masks, buffer.rnn_states,
                                buffer.rnn_states_critic, n_chunks, L, vn_state, self.hp, self._rnn_ws, self._sums)
        if ev is not None:
            e1.record()
            ev.append((e0, e1))
        if self.world_size > 1:
            if self._comm is not None:
                self._comm.allreduce_(self._sums)
            else:
                dist_utils.allreduce_(self._sums)
        po.step_count += 1 if turn_on else 0
        co.step_count += 1
        hp = self.hp
        if not turn_on:
            hp = nat.PPOHParams.from_buffer_copy(self.hp)
            hp.reserved |= 1
        ops_rnn.rnn_ppo_apply(p.net, c.net, self._sums, hp, po.native_state(max(po.step_count, 1)),
                              co.native_state(co.step_count), self._info, self._rnn_scratch)

    def _chunk_batches(self, M: int):
        L = self.data_chunk_length
        if self.use_joint_action_loss:  # recurrent_generator_v3 chunks (env, time) positions, the agent axis is kept
            M = M // self.agent_num
        assert M >= L, "PPO requires n_rollout_threads * num_agents * episode_length >= data_chunk_length"
        data_chunks = M // L
        mbs = data_chunks // self.num_mini_batch
        if self.perm_mode == "device":
            n, seed, sid, vn = self._perm_job(data_chunks)
            self._vn_in_perm = vn is not None
            rand = ops.perm_feistel(n, seed, sid, self.device, vn)
        elif self.perm_mode == "identity":  # chunks in buffer order (parity debugging / determinism)
            rand = torch.arange(data_chunks, dtype=torch.int64, device=self.device)
        else:
            rand = torch.randperm(data_chunks).to(self.device, non_blocking=True)  # replay_data.py:1078
        return [rand[i * mbs:(i + 1) * mbs] for i in range(self.num_mini_batch)], mbs

    def _perm_job(self, M: int):
        """Arguments of the next device permutation (and the ValueNorm.update that rides with it)."""
        self._perm_counter += 1
        vn = None
        if self._full_batch_moments and self._use_valuenorm:  # one launch: permutation + this epoch's ValueNorm.update
            v = self.algo_module.get_critic_value_normalizer()
            vn = (v.state, self._moments, v.beta)
        return M, int(self.cfg.seed), self._perm_counter, vn

    def _minibatch_indices(self, M: int, perm=None):
        """``perm``: this epoch's permutation if the previous epoch's optimiser-step launch already produced it."""
        mbs = M // self.num_mini_batch
        n_batches = M // mbs  # drop_last=True (replay_data.py:578-580)
        if self.perm_mode == "identity":  # buffer order: the whole batch, or contiguous slices of it
            if self.num_mini_batch == 1:
                return [None], mbs
            order = torch.arange(M, dtype=torch.int64, device=self.device)
            return [order[b * mbs:(b + 1) * mbs] for b in range(n_batches)], mbs
        if perm is not None:
            self._vn_in_perm = self._full_batch_moments and self._use_valuenorm
        elif self.perm_mode == "device":
            n, seed, sid, vn = self._perm_job(M)
            self._vn_in_perm = vn is not None
            perm = ops.perm_feistel(n, seed, sid, self.device, vn)
        else:
            perm = torch.randperm(M).to(self.device, non_blocking=True)  # CPU generator, like the reference
        return [perm[b * mbs:(b + 1) * mbs] for b in range(n_batches)], mbs

    # ------------------------------------------------------------------------------------------ reference API
    def train_ppo(self, buffer, turn_on: bool = True) -> Dict[str, float]:
        if self.naive_recurrent:
            self.data_chunk_length = buffer.episode_length
        if getattr(self, "_comm_watch", None) is not None:
            self._comm_watch.poll()  # the previous update's collectives (no sync)
        self._advantages_and_records(buffer)
        M = buffer.episode_length * buffer.n_rollout_threads * buffer.num_agents
        # one minibatch == every sample (recurrent: every chunk, when the chunks tile the batch exactly)
        self._full_batch_moments = self.num_mini_batch == 1 and (not self.recurrent or M % self.data_chunk_length == 0)
        # full batch: {sum ret, sum ret^2, count} are already in the statistics row the pack kernel wrote (a view, no copy)
        self._moments = self._adv_stats[8:11] if self._full_batch_moments else self._moments_mb
        # a FRESH accumulator per call (torch.empty: no launch): the returned DeviceTrainInfo keeps it, so nothing has to be
        # copied or scaled on the device afterwards.  Default MLP towers: the first apply launch overwrites its slots
        # (hparams.reserved & 32) - no zero fill either; the other update paths accumulate from zero.
        self._info = torch.empty(8, dtype=torch.float32, device=self.device)
        num_updates = self.ppo_epoch * self.num_mini_batch
        if num_updates <= 0:
            raise ValueError("train_ppo: ppo_epoch * num_mini_batch must be positive (got %d x %d)"
                             % (self.ppo_epoch, self.num_mini_batch))
        self._info_first = not (self.generic or self.recurrent)
        if not self._info_first:  # (slots 6 and 7 are no train_info keys: only [:len(info_keys)] is ever returned)
            self._info.zero_()
        self.last_indices = []
        next_perm = None
        self._jrpo_epoch_id = getattr(self, "_jrpo_epoch_id", 0) + 1  # a fresh records copy per train() call
        if self.use_joint_action_loss:
            self._full_batch_moments = False  # ValueNorm sees agent 0's returns only (to_single_np)
            self._moments = self._moments_mb
        for epoch in range(self.ppo_epoch):
            if self.recurrent:  # get_data_generator (ppo.py:363-372)
                batches, mbs = self._chunk_batches(M)
                for chunks in batches:
                    self.last_indices.append(chunks)
                    self._update_minibatch_rnn(buffer, chunks, mbs, turn_on)
                continue
            batches, mbs = self._minibatch_indices(M, next_perm)
            next_perm = None
            for k, idx in enumerate(batches):
                self.last_indices.append(idx)
                # the last optimiser step of an epoch also produces the next epoch's permutation (same launch)
                job = None
                if (self.fuse_next_perm and k == len(batches) - 1 and epoch + 1 < self.ppo_epoch
                        and self.perm_mode == "device"):
                    job = self._perm_job(M)
                next_perm = self._update_minibatch(buffer, idx, mbs, turn_on, job)
        self._info_first = False  # armed for this call's first apply launch only
        # no device->host sync here: the averages stay on the device until somebody reads the dict (logging every
        # log_interval iterations, tests), so the host can enqueue the next rollout while this update still runs
        keys = self.info_keys
        scale = np.full(len(keys), np.float32(1.0) / np.float32(num_updates), dtype=np.float32)
        if self.use_joint_action_loss:  # every agent row (and action dim) of a (step, chunk) carried the joint term
            scale[1] /= np.float32(buffer.num_agents * buffer.act_shape)
        watch = getattr(self, "_comm_watch", None)
        if watch is None:
            return DeviceTrainInfo(keys, self._info[:len(keys)], scale=scale)
        # multi-GPU: the comm's error word rides behind this update's launches; it is looked at when train_info is read
        # and (without a sync) when the next update starts - a timed-out peer never stays silent
        watch.post(self._comm.error_flag())
        return DeviceTrainInfo(keys, self._info[:len(keys)], after_sync=lambda: watch.poll(wait=True), scale=scale)

    def train(self, buffer, turn_on: bool = True) -> Dict[str, float]:
        if len(self.train_list) == 1:
            return self.train_list[0](buffer, turn_on)  # keeps a DeviceTrainInfo lazy
        train_info = {}
        for train_func in self.train_list:
            train_info.update(train_func(buffer, turn_on))
        return train_info
