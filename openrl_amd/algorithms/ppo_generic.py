"""One PPO minibatch on the GENERAL towers (``modules/generic_net.py``): ``PPOAlgorithm.ppo_update``
(openrl/algorithms/ppo.py:46-176) = prepare_loss (:238-361) -> backward of every loss in the loss list (:112-124)
-> clip_grad_norm_ per parameter group (:127-145) -> optimizer.step() for every optimizer (:160-164), as a chain of
HIP launches per layer (``ops_gen``).  Separate policy / critic networks take one trunk pass each; the shared
``PolicyValueNetwork`` takes ONE pass when policy and critic observations are the same array (the reference runs
``obs_prep`` + ``common`` twice on the same data and sums the gradients - identical up to fp32 summation order),
two passes otherwise, and is clipped twice exactly as the reference clips ``get_actor_para()`` and then
``get_critic_para()`` (both = all parameters, base_value_policy_network.py:58-62).
"""
from __future__ import annotations

import torch

from .. import _native as nat
from .. import distributed as dist_utils
from .. import ops, ops_gen
from ..modules import generic_net as gn


def _loss_sums(ws, name: str, nb: int, width: int, out=None):
    out = ws.loss_sums[name] if out is None else out
    ops_gen.colsum(ws.loss_partials, nb, [(out, width)])
    return out


def update_minibatch_generic(algo, buffer, idx, mb: int, turn_on: bool, rnn=None, jrpo=None) -> None:
    """``rnn = (L, n_chunks)``: recurrent towers - ``idx`` holds the chunks' record rows in [L, n_chunks] order
    (``orl_rnn_chunk_rows``), the GRU between trunk and head runs over the L steps from the stored states at the chunk
    starts (recurrent_generator, replay_data.py:1062-1258; RNNLayer, rnn.py:39-99).
    ``jrpo = (records', critic_rows, critic_chunks)``: the joint-action loss (ppo.py:254-300) - the policy's rows read the
    adjusted records of ``orl_rnn_jrpo_records`` (joint ratio, agent 0's advantage / active mask), the critic runs on agent
    0's rows only; the caller has already updated ValueNorm from those rows."""
    mod = algo.algo_module
    rec = buffer.records
    Dp, Dc, a_w, K = buffer.Dp, buffer.Dc, buffer.act_shape, buffer.K
    hp = algo.hp
    vn = mod.get_critic_value_normalizer() if algo._use_valuenorm else None
    vn_state = None
    idx_c, mb_c = idx, mb
    if jrpo is not None:
        rec, idx_c, nc_c = jrpo
        mb_c = rnn[0] * nc_c
        vn_state = vn.state if vn is not None else None
    elif vn is not None:
        if not algo._full_batch_moments:
            ret_col = Dp + Dc + 2 * a_w + 2
            ops.minibatch_moments(rec, ret_col, idx, mb, algo._mom_scratch, algo._moments)
            if algo.world_size > 1:
                dist_utils.allreduce_(algo._moments)
        if algo._vn_in_perm:
            algo._vn_in_perm = False  # this epoch's update already ran inside the permutation launch
        else:
            ops.valuenorm_update(vn.state, algo._moments, vn.beta)  # BEFORE normalize (ppo.py:190-195)
        vn_state = vn.state
    den = den_c = algo._gen_den
    # one minibatch == the whole batch: the masked-mean denominators (counts of 0 / 1 floats - exact in fp32) are the
    # same for every epoch of this train() call, so they are computed and summed over ranks once per call
    # keyed on the train() call, the batch size AND the record tensor (a direct caller handing another buffer of the same
    # size must not reuse - or, multi-GPU, skip the collective of - stale denominators)
    stamp = ((getattr(algo, "_jrpo_epoch_id", 0), mb, int(rec.data_ptr()))
             if (jrpo is None and algo.num_mini_batch == 1) else None)
    if stamp is None or stamp != algo._gen_den_stamp:
        ops_gen.denoms(rec, Dp, Dc, a_w, idx, mb, den, algo._gen_den_scratch)
        if jrpo is not None:  # the value loss averages over agent 0's rows
            den_c = algo._gen_den_c
            ops_gen.denoms(rec, Dp, Dc, a_w, idx_c, mb_c, den_c, algo._gen_den_scratch)
        if algo.world_size > 1:  # global denominators: ONE tiny collective ahead of the losses (2 or 4 floats)
            algo._allreduce_vec(algo._gen_den4 if jrpo is not None else den)
        algo._gen_den_stamp = stamp
    elif jrpo is not None:
        den_c = algo._gen_den_c
    pn, cn = mod.policy_net, mod.critic_net
    shared = mod.share_model
    # (recurrent: the actor and the critic carry their own stored states through the shared GRU - two passes)
    one_pass = shared and buffer.critic_obs is buffer.policy_obs and rnn is None

    head = pn.head_desc
    gsums = algo._gen_sums  # multi-GPU: the sum rows live in the flat vector the one collective reduces

    def reduce_over_ranks():
        # ONE collective per optimiser step - every network's gradient + the loss / logging sums
        if algo.world_size > 1:
            algo._allreduce_vec(algo._gen_flat)

    def step(net, opt, n_clips, slot_first, slot_second):
        opt.step_count += 1
        ops_gen.adam(opt.native_state(opt.step_count), net.n_params, hp.max_grad_norm, bool(hp.use_max_grad_norm), n_clips,
                     algo._gen_scratch, algo._info, slot_first, slot_second, algo.device)

    def finish(psums, vsums):
        # train_info (device-side accumulation; the sums are global in a multi-GPU run: they rode in the flat vector)
        gauss = head.kind == ops_gen.HEAD_GAUSSIAN
        ent_div = float(head.n_out) if (gauss and not hp.use_policy_active_masks) else 1.0
        ratio_div = float(a_w) if head.kind != ops_gen.HEAD_CATEGORICAL else 1.0
        if jrpo is None:
            ops_gen.info(psums, vsums, den, hp, ent_div, ratio_div, algo._info)
        else:  # the value loss is a mean over agent 0's rows
            ops_gen.info(psums, None, den, hp, ent_div, ratio_div, algo._info)
            ops_gen.info(None, vsums, den_c, hp, ent_div, ratio_div, algo._info)

    # ---- cross-layer fused towers (csrc/orl_gen_tower.h): ONE launch per tower and minibatch straight from the records -
    # forward, the losses of prepare_loss on the head outputs, backward (+ the fixed-order reduction of its per-workgroup
    # sums); no activation, head output or head gradient crosses HBM
    fused = mod.fused_towers(one_pass) if (rnn is None and jrpo is None) else None
    if fused is not None:
        ftp, ftc = fused
        def logstd_grad_f(net, psums):
            h = net.heads["act"]
            if "logstd" in h:
                ops.multi_copy([(net.v(h["logstd"], h["n_ls"], grad=True), psums[4:4 + h["n_ls"]])])

        def take(sums, lo, hi, dst):
            # multi-GPU: the sums ride in the flat vector the one collective reduces
            if dst is None:
                return sums[lo:hi]
            dst[:hi - lo].copy_(sums[lo:hi])
            return dst

        if shared:  # one tower, two heads: forward, both losses and backward in ONE launch
            model, opt = mod.models["model"], mod.optimizers["model"]
            ftp.prep()
            ftp.zero_grad_once()
            sums = ftp.train(rec, 0, idx, mb, head, mod._logstd(), Dp, Dc, a_w, K, den, vn_state, hp, policy_grad=turn_on)
            psums = take(sums, 0, 20, None if gsums is None else gsums[0])
            vsums = take(sums, 20, 21, None if gsums is None else gsums[1])
            if turn_on:
                logstd_grad_f(model, psums)
            reduce_over_ranks()
            step(model, opt, 2, 3, 4)
        else:
            popt, copt = mod.optimizers["policy"], mod.optimizers["critic"]

            def critic_chain():
                ftc.prep()
                ftc.zero_grad_once()
                sums = ftc.train(rec, Dp, idx, mb, None, None, Dp, Dc, a_w, K, den, vn_state, hp)
                return take(sums, 20, 21, None if gsums is None else gsums[1])

            fork = turn_on and mod.two_stream
            if fork:  # the critic's chain beside the policy's, as on the layer-wise route
                main = torch.cuda.current_stream(algo.device)
                side = mod.side_stream()
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    vsums = critic_chain()
            ftp.prep()
            ftp.zero_grad_once()
            ev = getattr(algo, "profile_events", None)
            if ev is not None:  # benchmarks: HIP events on the launch stream around the policy tower's launch
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            # (turn_on = False: the launch still runs for the logging sums; the policy's gradient is dropped in the kernel
            # and the optimiser step below is skipped - ppo.py:226-236)
            sums = ftp.train(rec, 0, idx, mb, head, mod._logstd(), Dp, Dc, a_w, K, den, None, hp, policy_grad=turn_on)
            if ev is not None:
                e1.record()
                ev.append((e0, e1))
            psums = take(sums, 0, 20, None if gsums is None else gsums[0])
            if turn_on:
                logstd_grad_f(pn, psums)
            if fork:
                main.wait_stream(side)
            else:
                vsums = critic_chain()
            reduce_over_ranks()
            if turn_on:
                step(pn, popt, 1, 3, -1)
            step(cn, copt, 1, 4, -1)
        finish(psums, vsums)
        return

    # ---- forward
    wp = mod.workspace(pn, mb, True, "p")
    xp = wp.v(wp.x0, mb, Dp)
    ops_gen.gather_cols(rec, 0, Dp, idx, mb, xp)
    trunk_p = gp = mrows = None
    if rnn is not None:
        L, Nc = rnn
        H = pn.state_w * pn.recurrent_N
        gp = mod.gru_workspace(pn, L, Nc, True, "p")
        mrows, h0p = gp.mrows[:mb], gp.h0[:Nc]
        ops_gen.gather_cols(buffer.masks.view(-1, 1), 0, 1, idx, mb, mrows.view(mb, 1))   # mask of step l of chunk i
        ops_gen.gather_cols(buffer.rnn_states.view(-1, H), 0, H, idx[:Nc], Nc, h0p)        # states at the chunk starts

    def policy_forward():
        feats = gn.trunk_forward(pn, wp, xp, True)
        trunk = None
        if rnn is not None:
            trunk, feats = feats, gn.gru_forward(pn, gp, feats, h0p, mrows, L, Nc, True)
        return trunk, feats, gn.head_forward(pn, wp, "act", feats)

    def critic_forward():
        """The critic's trunk (+ recurrent stack) + value head on its rows -> (workspace, features, trunk features, GRU
        workspace, mask rows, values)."""
        if one_pass:  # (after policy_forward)
            return wp, feats_p, None, None, None, gn.head_forward(cn, wp, "v_out", feats_p)
        wc = mod.workspace(cn, mb_c, True, "c")
        xc = wc.v(wc.x0, mb_c, Dc)
        ops_gen.gather_cols(rec, Dp, Dc, idx_c, mb_c, xc)
        feats_c = gn.trunk_forward(cn, wc, xc, True)
        trunk_c = gc = mrows_c = None
        if rnn is not None:
            Ncc, Hc = mb_c // L, cn.state_w * cn.recurrent_N
            gc = mod.gru_workspace(cn, L, Ncc, True, "c")
            mrows_c, h0c = mrows, gc.h0[:Ncc]
            if idx_c is not idx:
                mrows_c = gc.mrows[:mb_c]
                ops_gen.gather_cols(buffer.masks.view(-1, 1), 0, 1, idx_c, mb_c, mrows_c.view(mb_c, 1))
            ops_gen.gather_cols(buffer.rnn_states_critic.view(-1, Hc), 0, Hc, idx_c[:Ncc], Ncc, h0c)
            trunk_c, feats_c = feats_c, gn.gru_forward(cn, gc, feats_c, h0c, mrows_c, L, Ncc, True)
        return wc, feats_c, trunk_c, gc, mrows_c, gn.head_forward(cn, wc, "v_out", feats_c)

    def critic_loss(wc, values):
        dvalues = wc.v(wc.dhead["v_out"], mb_c, 1)
        nb = ops_gen.value_loss(values.view(-1), rec, Dp, Dc, a_w, K, idx_c, mb_c, vn_state, den_c, hp, dvalues.view(-1),
                                wc.loss_partials)
        return dvalues, _loss_sums(wc, "v_out", nb, 1, None if algo._gen_sums is None else algo._gen_sums[1])

    def critic_backward(wc, feats_c, trunk_c, gc, mrows_c, dvalues):
        cn.grad.zero_()
        dfeat = wc.v(wc.dfeat, mb_c, cn.H)
        gn.head_backward(cn, wc, "v_out", feats_c, dvalues, dfeat, False)
        if rnn is not None:
            dfeat = gn.gru_backward(cn, gc, trunk_c, mrows_c, dfeat, L, mb_c // L)
        gn.trunk_backward(cn, wc, dfeat)

    # Separate networks: the critic's whole chain (forward, value loss, backward) is independent of the policy's - it runs
    # on a second stream beside it, so one tower's K loops fill the other's store phases; the optimiser steps (shared
    # norm scratch, the gradient all-reduce of a multi-GPU run) follow the join on the main stream.
    fork = not shared and turn_on and mod.two_stream
    side = main = None
    if fork:
        main = torch.cuda.current_stream(algo.device)
        side = mod.side_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            wc, feats_c, trunk_c, gc, mrows_c, values = critic_forward()
            dvalues, vsums = critic_loss(wc, values)
            critic_backward(wc, feats_c, trunk_c, gc, mrows_c, dvalues)
    trunk_p, feats_p, logits = policy_forward()
    if not fork:
        wc, feats_c, trunk_c, gc, mrows_c, values = critic_forward()

    # ---- losses: d loss / d head outputs (already divided by the denominators) + statistics
    dlogits = wp.v(wp.dhead["act"], mb, head.n_out)
    nb = ops_gen.policy_loss(head, logits, mod._logstd(), rec, Dp, Dc, a_w, K, idx, mb, den, hp, dlogits,
                             wp.loss_partials)
    psums = _loss_sums(wp, "act", nb, 20, None if gsums is None else gsums[0])
    if not fork:
        dvalues, vsums = critic_loss(wc, values)

    # ---- backward + optimiser
    def logstd_grad(net):
        h = net.heads["act"]
        if "logstd" in h:
            ops.multi_copy([(net.v(h["logstd"], h["n_ls"], grad=True), psums[4:4 + h["n_ls"]])])

    if shared:
        model, opt = mod.models["model"], mod.optimizers["model"]
        model.grad.zero_()
        if one_pass:
            dfeat = wp.v(wp.dfeat, mb, model.H)
            gn.head_backward(model, wp, "v_out", feats_c, dvalues, dfeat, False)
            if turn_on:
                gn.head_backward(model, wp, "act", feats_p, dlogits, dfeat, True)
                logstd_grad(model)
            gn.trunk_backward(model, wp, dfeat)
        else:
            dfeat = wc.v(wc.dfeat, mb_c, model.H)
            gn.head_backward(model, wc, "v_out", feats_c, dvalues, dfeat, False)
            if rnn is not None:
                dfeat = gn.gru_backward(model, gc, trunk_c, mrows_c, dfeat, L, mb_c // L)
            gn.trunk_backward(model, wc, dfeat)
            if turn_on:
                g1 = model.grad.clone()
                dfeat = wp.v(wp.dfeat, mb, model.H)
                gn.head_backward(model, wp, "act", feats_p, dlogits, dfeat, False)
                logstd_grad(model)
                if rnn is not None:  # rewrites the GRU's slots like the trunk's; g1 holds the critic pass's
                    dfeat = gn.gru_backward(model, gp, trunk_p, mrows, dfeat, L, Nc)
                gn.trunk_backward(model, wp, dfeat)  # rewrites the trunk slots; head slots of the other pass stay
                hv = model.heads["v_out"]
                lo, hi = hv["W"], hv["b"] + 1
                g1[lo:hi] = 0.0  # v_out's gradient is already in model.grad (the second pass did not touch it)
                ops_gen.vec_add(model.grad, g1)
        reduce_over_ranks()
        step(model, opt, 2, 3, 4)  # actor_grad_norm, then critic_grad_norm of the once-clipped gradient
    else:
        popt, copt = mod.optimizers["policy"], mod.optimizers["critic"]
        if turn_on:
            pn.grad.zero_()
            dfeat = wp.v(wp.dfeat, mb, pn.H)
            gn.head_backward(pn, wp, "act", feats_p, dlogits, dfeat, False)
            logstd_grad(pn)
            if rnn is not None:
                dfeat = gn.gru_backward(pn, gp, trunk_p, mrows, dfeat, L, Nc)
            gn.trunk_backward(pn, wp, dfeat)
        if fork:
            main.wait_stream(side)
        else:
            critic_backward(wc, feats_c, trunk_c, gc, mrows_c, dvalues)
        reduce_over_ranks()
        if turn_on:
            step(pn, popt, 1, 3, -1)
        step(cn, copt, 1, 4, -1)

    finish(psums, vsums)
