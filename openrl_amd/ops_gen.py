"""Tensor-level wrappers over the general tower path of the C ABI (``include/orl_hip.h``, csrc/orl_gen.hip).

Same contract as ``ops.py``: shape/dtype checks, raw pointers, current stream; nothing is computed in Python.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import torch

from . import _native as nat
from ._native import AdamState, HeadDesc, PPOHParams, fptr, ptr, stream_ptr

ACT_NONE, ACT_TANH, ACT_RELU, ACT_LEAKY_RELU, ACT_ELU = (nat.ORL_ACT_NONE, nat.ORL_ACT_TANH, nat.ORL_ACT_RELU,
                                                         nat.ORL_ACT_LEAKY_RELU, nat.ORL_ACT_ELU)
HEAD_CATEGORICAL, HEAD_GAUSSIAN, HEAD_MULTI_DISCRETE = nat.ORL_HEAD_CATEGORICAL, nat.ORL_HEAD_GAUSSIAN, nat.ORL_HEAD_MULTI_DISCRETE
HEAD_MIXED = nat.ORL_HEAD_MIXED
MAX_BLOCKS = 1024  # rows of every per-workgroup partial buffer of this path


def _lib():
    return nat.load()


def head_desc(kind: int, n_out: int, nvec: Optional[Sequence[int]] = None) -> HeadDesc:
    h = HeadDesc()
    h.kind, h.n_out, h.n_heads = int(kind), int(n_out), 1 if nvec is None else len(nvec)
    for k, v in enumerate(nvec or []):
        h.nvec[k] = int(v)
    return h


def gemm(A: torch.Tensor, sam: int, sak: int, B: torch.Tensor, sbk: int, sbn: int, Cmat: torch.Tensor, ldc: int, M: int,
         N: int, K: int, n_split: int = 1, partials: Optional[torch.Tensor] = None) -> None:
    """Cmat[m*ldc + n] = sum_k A[m*sam + k*sak] * B[k*sbk + n*sbn] (fp32 MFMA); split-K through ``partials``."""
    dev = nat.require_gpu(Cmat.device)
    if n_split > 1:
        assert partials is not None and partials.numel() >= n_split * M * N
    rc = _lib().orl_gemm(fptr(A), sam, sak, fptr(B), sbk, sbn, fptr(Cmat), ldc, M, N, K, n_split, fptr(partials),
                         stream_ptr(dev))
    nat.check(rc, "orl_gemm")


def linear_fwd(x: torch.Tensor, W: torch.Tensor, z: torch.Tensor) -> None:
    """z[B, out] = x[B, in] @ W[out, in]^T (nn.Linear without the bias)."""
    Bn, K = x.shape
    N = W.shape[0]
    gemm(x, K, 1, W, 1, K, z, N, Bn, N, K)


def linear_dgrad(dz: torch.Tensor, W: torch.Tensor, dx: torch.Tensor) -> None:
    """dx[B, in] = dz[B, out] @ W[out, in] (``orl_gen_matmul``: the tiled kernel with W read k-major)."""
    Bn, K = dz.shape
    N = W.shape[1]
    assert dz.is_contiguous() and W.is_contiguous() and dx.is_contiguous()
    rc = _lib().orl_gen_matmul(fptr(dz), Bn, K, fptr(W), N, fptr(dx), stream_ptr(nat.require_gpu(dz.device)))
    nat.check(rc, "orl_gen_matmul")


def linear_wgrad(dz: torch.Tensor, x: torch.Tensor, dW: torch.Tensor, partials: torch.Tensor) -> None:
    """dW[out, in] = dz[B, out]^T @ x[B, in]; the batch rows are the K dimension (deterministic split-K)."""
    Bn, M = dz.shape
    N = x.shape[1]
    tiles = ((M + 63) // 64) * ((N + 63) // 64)
    n_split = max(1, min(Bn // 512, max(1, 1024 // tiles), partials.numel() // (M * N)))
    gemm(dz, 1, M, x, N, 1, dW, N, M, N, Bn, n_split, partials)


def row_fwd(z, bias, act: int, gamma, beta, a_out, xhat_out, rstd_out, y_out) -> None:
    dev = nat.require_gpu(z.device)
    Bn, H = z.shape
    rc = _lib().orl_row_fwd(fptr(z), fptr(bias), act, fptr(gamma), fptr(beta), Bn, H, fptr(a_out), fptr(xhat_out),
                            fptr(rstd_out), fptr(y_out), stream_ptr(dev))
    nat.check(rc, "orl_row_fwd")


def row_bwd(dy, gamma, xhat, rstd, a, act: int, dz_out, col_partials) -> int:
    """Returns the number of partial rows written to ``col_partials`` ([rows][3H])."""
    dev = nat.require_gpu(dy.device)
    Bn, H = dy.shape
    nb = C.c_int(0)
    max_blocks = min(MAX_BLOCKS, col_partials.numel() // (3 * H))
    rc = _lib().orl_row_bwd(fptr(dy), fptr(gamma), fptr(xhat), fptr(rstd), fptr(a), act, Bn, H, fptr(dz_out),
                            fptr(col_partials), max_blocks, C.byref(nb), stream_ptr(dev))
    nat.check(rc, "orl_row_bwd")
    return nb.value


def layer_fwd(x, W, bias, act: int, gamma, beta, a_out, stats_out, y_out) -> None:
    """y = LN(act(x @ W^T + bias)) in one launch (``orl_gen_layer_fwd``); ``a_out`` [B, out] / ``stats_out`` [B, 2] feed
    ``layer_bwd``."""
    dev = nat.require_gpu(x.device)
    Bn, n_in = x.shape
    n_out = W.shape[0]
    assert x.is_contiguous() and W.shape[1] == n_in
    rc = _lib().orl_gen_layer_fwd(fptr(x), Bn, n_in, fptr(W), fptr(bias), act, fptr(gamma), fptr(beta), n_out,
                                  fptr(a_out), fptr(stats_out), fptr(y_out), stream_ptr(dev))
    nat.check(rc, "orl_gen_layer_fwd")


def mlp_fwd(desc, x, head_out0, head_out1=None, feats_out=None) -> None:
    """The whole tower of a rollout step in one launch (``orl_gen_mlp_fwd``); ``desc``: ``_native.GenMlpDesc``;
    ``feats_out`` [B, H]: the trunk's features (recurrent towers run their GRU cell on them)."""
    dev = nat.require_gpu(x.device)
    assert x.is_contiguous() and x.shape[1] == desc.layer[0].n_in
    rc = _lib().orl_gen_mlp_fwd(C.byref(desc), fptr(x), x.shape[0], fptr(head_out0), fptr(head_out1), fptr(feats_out),
                                stream_ptr(dev))
    nat.check(rc, "orl_gen_mlp_fwd")


def gt_supported(desc) -> bool:
    return bool(_lib().orl_gt_supported(C.byref(desc)))


def gt_sizes(desc):
    """(image floats, raw gradient-sum floats) of a fused general tower (``orl_gt_image_floats`` / ``orl_gt_raw_floats``)."""
    a, b = _lib().orl_gt_image_floats(C.byref(desc)), _lib().orl_gt_raw_floats(C.byref(desc))
    if a < 0 or b < 0:
        msg = _lib().orl_last_error_string()
        raise nat.NativeError("orl_gt_image_floats / orl_gt_raw_floats: %s" % (msg.decode() if msg else "unsupported tower"))
    return int(a), int(b)


def gt_prep(desc, image) -> None:
    """theta -> the folded / split image the fused tower kernels read (``orl_gt_prep``); after every optimiser step."""
    dev = nat.require_gpu(image.device)
    nat.check(_lib().orl_gt_prep(C.byref(desc), fptr(image), stream_ptr(dev)), "orl_gt_prep")


def gt_fwd(desc, image, x, col0: int, idx, mb: int, out0, out1=None) -> None:
    """Head outputs of rows ``x[idx[i] or i, col0 : col0 + D]`` in one launch (``orl_gt_fwd``)."""
    dev = nat.require_gpu(x.device)
    assert x.dim() == 2 and x.is_contiguous() and x.dtype == torch.float32 and col0 + desc.D <= x.shape[1]
    assert out0.is_contiguous() and out0.numel() >= mb * desc.head_n[0]
    assert idx is None or (idx.dtype == torch.int64 and idx.numel() >= mb)
    rc = _lib().orl_gt_fwd(C.byref(desc), fptr(image), fptr(x), x.shape[1], col0, ptr(idx), mb, fptr(out0), fptr(out1),
                           stream_ptr(dev))
    nat.check(rc, "orl_gt_fwd")


def gt_bwd(desc, image, x, col0: int, idx, mb: int, dh0, dh1, partials, raw, grad) -> None:
    """Backward of the same rows: every gradient the descriptor names is written into ``grad`` (``orl_gt_bwd``)."""
    dev = nat.require_gpu(x.device)
    assert x.dim() == 2 and x.is_contiguous() and col0 + desc.D <= x.shape[1]
    assert dh0.is_contiguous() and dh0.numel() >= mb * desc.head_n[0] and (dh1 is None or dh1.is_contiguous())
    rc = _lib().orl_gt_bwd(C.byref(desc), fptr(image), fptr(x), x.shape[1], col0, ptr(idx), mb, fptr(dh0), fptr(dh1),
                           fptr(partials), partials.numel(), fptr(raw), fptr(grad), stream_ptr(dev))
    nat.check(rc, "orl_gt_bwd")


def gt_train(desc, image, records, col0: int, idx, mb: int, loss, partials, raw, grad, sums_out) -> None:
    """Forward + losses + backward of one tower's minibatch in one launch (``orl_gt_train``); ``loss``: ``_native.GtLoss``;
    ``sums_out`` [24] receives the loss / logging sums the two loss kernels' partials would reduce to."""
    dev = nat.require_gpu(records.device)
    assert records.dim() == 2 and records.is_contiguous() and sums_out.numel() >= 24 and sums_out.is_contiguous()
    rc = _lib().orl_gt_train(C.byref(desc), fptr(image), fptr(records), records.shape[1], col0, ptr(idx), mb, C.byref(loss),
                             fptr(partials), partials.numel(), fptr(raw), fptr(grad), fptr(sums_out), stream_ptr(dev))
    nat.check(rc, "orl_gt_train")


def act_step(policy_desc, obs, critic_desc, critic_obs, values, head: HeadDesc, logstd, action_masks, deterministic: bool,
        seed: int, row0: int, rng_step: int, rng_step_dev, forced_u, a_w: int, actions, logp, logits_out=None) -> None:
    """A rollout step in one launch (``orl_gen_act``): the policy tower, ``sample`` on its logits, and the critic tower
    (``critic_desc``) or the shared network's value head (``policy_desc`` with two heads); ``values`` [B] contiguous."""
    dev = nat.require_gpu(obs.device)
    assert obs.is_contiguous() and obs.shape[1] == policy_desc.layer[0].n_in
    assert critic_desc is None or (critic_obs.is_contiguous() and critic_obs.shape[0] == obs.shape[0] and
                                   critic_obs.shape[1] == critic_desc.layer[0].n_in)
    assert values is None or values.is_contiguous()
    rc = _lib().orl_gen_act(C.byref(policy_desc), fptr(obs), C.byref(critic_desc) if critic_desc is not None else None,
                            fptr(critic_obs) if critic_desc is not None else None, obs.shape[0], fptr(logits_out),
                            fptr(values), C.byref(head), fptr(logstd), fptr(action_masks), int(bool(deterministic)),
                            seed & (2 ** 64 - 1), row0, rng_step, ptr(rng_step_dev), fptr(forced_u), a_w, fptr(actions),
                            fptr(logp), stream_ptr(dev))
    nat.check(rc, "orl_gen_act")


def rollout_fused(policy_desc, head: HeadDesc, logstd, buf_ptrs, value_preds, actions, logp, env_state, ep_stats,
                  env_kind: int, episode_limit: int, env_seed: int, env_step0: int, act_seed: int, rng_step0: int,
                  a_w: int, device) -> None:
    """All episode_length rollout steps of a general policy tower on a device-resident single-agent env in one launch
    (``orl_gen_rollout_fused``); ``value_preds`` only for a shared network (policy descriptor with two heads)."""
    dev = nat.require_gpu(device)
    rc = _lib().orl_gen_rollout_fused(C.byref(policy_desc), C.byref(head), fptr(logstd), C.byref(buf_ptrs),
                                      fptr(value_preds), fptr(actions), fptr(logp), fptr(env_state), fptr(ep_stats),
                                      env_kind, episode_limit, env_seed & (2 ** 64 - 1), env_step0 & (2 ** 64 - 1),
                                      act_seed & (2 ** 64 - 1), rng_step0 & (2 ** 64 - 1), a_w, stream_ptr(dev))
    nat.check(rc, "orl_gen_rollout_fused")


def layer_bwd(dy, a, stats, gamma, act: int, W, dz_out, dx_out, col_partials) -> int:
    """dy -> dz (+ dx = dz @ W for a square layer when ``dx_out`` is given) and the [d gamma | d beta | d bias] partial
    rows; returns their count."""
    dev = nat.require_gpu(dy.device)
    Bn, n_out = dy.shape
    n_in = W.shape[1] if W is not None else n_out
    nb = C.c_int(0)
    max_blocks = min(MAX_BLOCKS, col_partials.numel() // (3 * n_out))
    rc = _lib().orl_gen_layer_bwd(fptr(dy), fptr(a), fptr(stats), fptr(gamma), act, Bn, n_out, fptr(W), n_in,
                                  fptr(dz_out), fptr(dx_out), fptr(col_partials), max_blocks, C.byref(nb),
                                  stream_ptr(dev))
    nat.check(rc, "orl_gen_layer_bwd")
    return nb.value


def wgrad(dz, x, dW, partials) -> None:
    """dW[out, in] = dz[B, out]^T @ x[B, in] (``orl_gen_wgrad``: persistent split-K, fixed summation order)."""
    dev = nat.require_gpu(dz.device)
    Bn, n_out = dz.shape
    n_in = x.shape[1]
    rc = _lib().orl_gen_wgrad(fptr(dz), fptr(x), Bn, n_out, n_in, fptr(dW), fptr(partials), partials.numel(),
                              stream_ptr(dev))
    nat.check(rc, "orl_gen_wgrad")


def colsum(partials, n_rows: int, dsts) -> None:
    """Column sums of ``partials[n_rows][sum(widths)]``; ``dsts`` = up to three (tensor or None, width) segments."""
    dev = nat.require_gpu(partials.device)
    dsts = list(dsts) + [(None, 0)] * (3 - len(dsts))
    width = sum(w for _, w in dsts)
    rc = _lib().orl_gen_colsum(fptr(partials), n_rows, width, fptr(dsts[0][0]), dsts[0][1], fptr(dsts[1][0]), dsts[1][1],
                               fptr(dsts[2][0]), dsts[2][1], stream_ptr(dev))
    nat.check(rc, "orl_gen_colsum")


def gather_cols(records, col0: int, width: int, idx, mb: int, out) -> None:
    dev = nat.require_gpu(records.device)
    rc = _lib().orl_gather_cols(fptr(records), records.shape[1], col0, width, ptr(idx), mb, fptr(out), stream_ptr(dev))
    nat.check(rc, "orl_gather_cols")


def denoms(records, Dp: int, Dc: int, a_w: int, idx, mb: int, den, scratch) -> None:
    """``scratch``: 257 floats zeroed once by the caller (left zeroed by every call)."""
    dev = nat.require_gpu(records.device)
    assert scratch.numel() >= 257
    rc = _lib().orl_gen_denoms(fptr(records), records.shape[1], Dp, Dc, a_w, ptr(idx), mb, fptr(den), fptr(scratch),
                               stream_ptr(dev))
    nat.check(rc, "orl_gen_denoms")


def policy_loss(head: HeadDesc, logits, logstd, records, Dp: int, Dc: int, a_w: int, K: int, idx, mb: int, den,
                hp: PPOHParams, dlogits, partials) -> int:
    dev = nat.require_gpu(records.device)
    nb = C.c_int(0)
    rc = _lib().orl_gen_policy_loss(C.byref(head), fptr(logits), fptr(logstd), fptr(records), records.shape[1], Dp, Dc,
                                    a_w, K, ptr(idx), mb, fptr(den), C.byref(hp), fptr(dlogits), fptr(partials),
                                    min(MAX_BLOCKS, partials.numel() // 20), C.byref(nb), None, None, stream_ptr(dev))
    nat.check(rc, "orl_gen_policy_loss")
    return nb.value


def policy_eval(head: HeadDesc, logits, logstd, records, Dp: int, Dc: int, a_w: int, K: int, mb: int, hp: PPOHParams,
                logp_out, ent_out) -> None:
    """ACTLayer.evaluate_actions on dense rows (records in identity order): log-probs [mb, a_w], entropy [mb]."""
    dev = nat.require_gpu(records.device)
    rc = _lib().orl_gen_policy_loss(C.byref(head), fptr(logits), fptr(logstd), fptr(records), records.shape[1], Dp, Dc,
                                    a_w, K, None, mb, None, C.byref(hp), None, None, 0, None, fptr(logp_out),
                                    fptr(ent_out), stream_ptr(dev))
    nat.check(rc, "orl_gen_policy_loss(eval)")


def value_loss(values, records, Dp: int, Dc: int, a_w: int, K: int, idx, mb: int, vn_state, den, hp: PPOHParams, dvalues,
               partials) -> int:
    dev = nat.require_gpu(records.device)
    nb = C.c_int(0)
    rc = _lib().orl_gen_value_loss(fptr(values), fptr(records), records.shape[1], Dp, Dc, a_w, K, ptr(idx), mb,
                                   fptr(vn_state), fptr(den), C.byref(hp), fptr(dvalues), fptr(partials),
                                   min(MAX_BLOCKS, partials.numel()), C.byref(nb), stream_ptr(dev))
    nat.check(rc, "orl_gen_value_loss")
    return nb.value


def sample(head: HeadDesc, logits, logstd, action_masks, Bn: int, deterministic: bool, seed: int, row0: int,
           rng_step: int, rng_step_dev, forced_u, a_w: int, actions, logp) -> None:
    dev = nat.require_gpu(logits.device)
    rc = _lib().orl_gen_sample(C.byref(head), fptr(logits), fptr(logstd), fptr(action_masks), Bn,
                               int(bool(deterministic)), seed & (2 ** 64 - 1), row0, rng_step, ptr(rng_step_dev),
                               fptr(forced_u), a_w, fptr(actions), fptr(logp), stream_ptr(dev))
    nat.check(rc, "orl_gen_sample")


def adam(state: AdamState, n: int, max_grad_norm: float, use_max_grad_norm: bool, n_clips: int, scratch, info,
         slot_first: int, slot_second: int, device) -> None:
    rc = _lib().orl_gen_adam(C.byref(state), n, float(max_grad_norm), int(bool(use_max_grad_norm)), n_clips,
                             fptr(scratch), fptr(info), slot_first, slot_second, stream_ptr(device))
    nat.check(rc, "orl_gen_adam")


def colsum_rows(x, dst, partials) -> None:
    """dst[c] = sum_r x[r, c] of a tall matrix (``orl_gen_colsum_rows``, fixed summation order)."""
    n_rows, width = x.shape
    rc = _lib().orl_gen_colsum_rows(fptr(x), n_rows, width, fptr(dst), fptr(partials), partials.numel(),
                                    stream_ptr(nat.require_gpu(x.device)))
    nat.check(rc, "orl_gen_colsum_rows")


def gru_gate_fwd(gi, gh, h_in, mask_next, h_out, h_in_next, save) -> None:
    """GRU gates of one step after the two projections (``orl_gen_gru_gate_fwd``)."""
    N, H = h_in.shape
    rc = _lib().orl_gen_gru_gate_fwd(fptr(gi), fptr(gh), fptr(h_in), fptr(mask_next), N, H, fptr(h_out), fptr(h_in_next),
                                     fptr(save), stream_ptr(nat.require_gpu(h_in.device)))
    nat.check(rc, "orl_gen_gru_gate_fwd")


def gru_gate_bwd(dh, save, h_in, dgi, dgh, dh_in) -> None:
    N, H = h_in.shape
    rc = _lib().orl_gen_gru_gate_bwd(fptr(dh), fptr(save), fptr(h_in), N, H, fptr(dgi), fptr(dgh), fptr(dh_in),
                                     stream_ptr(nat.require_gpu(h_in.device)))
    nat.check(rc, "orl_gen_gru_gate_bwd")


def lstm_gate_fwd(gi, gh, c_in, mask_next, h_out, c_out, h_in_next, c_in_next, save) -> None:
    N, H = c_in.shape
    rc = _lib().orl_gen_lstm_gate_fwd(fptr(gi), fptr(gh), fptr(c_in), fptr(mask_next), N, H, fptr(h_out), fptr(c_out),
                                      fptr(h_in_next), fptr(c_in_next), fptr(save), stream_ptr(nat.require_gpu(c_in.device)))
    nat.check(rc, "orl_gen_lstm_gate_fwd")


def lstm_gate_bwd(dh, dc, save, c_in, dgates, dc_in) -> None:
    N, H = c_in.shape
    rc = _lib().orl_gen_lstm_gate_bwd(fptr(dh), fptr(dc), fptr(save), fptr(c_in), N, H, fptr(dgates), fptr(dc_in),
                                      stream_ptr(nat.require_gpu(c_in.device)))
    nat.check(rc, "orl_gen_lstm_gate_bwd")


def row_affine(a, b, row_scale, add, out) -> None:
    """out = (a + b) * row_scale[:, None] + add; ``b`` / ``row_scale`` / ``add`` may be None."""
    N, H = a.shape
    rc = _lib().orl_gen_row_affine(fptr(a), fptr(b), fptr(row_scale), fptr(add), N, H, fptr(out),
                                   stream_ptr(nat.require_gpu(a.device)))
    nat.check(rc, "orl_gen_row_affine")


def vec_add(dst: torch.Tensor, src: torch.Tensor) -> None:
    assert dst.numel() == src.numel()
    rc = _lib().orl_vec_add(fptr(dst), fptr(src), dst.numel(), stream_ptr(nat.require_gpu(dst.device)))
    nat.check(rc, "orl_vec_add")


def info(policy_sums, value_sums, den, hp: PPOHParams, entropy_div: float, ratio_div: float, info_accum) -> None:
    dev = nat.require_gpu(info_accum.device)
    rc = _lib().orl_gen_info(fptr(policy_sums), fptr(value_sums), fptr(den), C.byref(hp), float(entropy_div),
                             float(ratio_div), fptr(info_accum), stream_ptr(dev))
    nat.check(rc, "orl_gen_info")
