// orl_gen_loss.h - the per-row loss terms of the general tower path, shared by the stand-alone loss kernels
// (orl_gen.hip: orl_gen_policy_loss / orl_gen_value_loss) and the fused tower update (orl_gen_tower.h), so that the two
// routes evaluate PPOAlgorithm.prepare_loss (openrl/algorithms/ppo.py:178-361) with the same arithmetic.
#pragma once
#include "orl_common.h"
#include "orl_gen_sample.h"

namespace orl {

// record columns (orl_adv_normalize_pack's row: [pobs | cobs | action | old_logp | adv | vpred | return | active | mask])
struct GenCols { int o_act, o_lp, o_adv, o_vp, o_rt, o_am, o_mk, a_w, K; };
__host__ __device__ inline GenCols gen_cols(int Dp, int Dc, int a_w, int K) {
  GenCols c;
  c.a_w = a_w; c.K = K;
  c.o_act = Dp + Dc; c.o_lp = c.o_act + a_w; c.o_adv = c.o_lp + a_w; c.o_vp = c.o_adv + 1; c.o_rt = c.o_adv + 2;
  c.o_am = c.o_adv + 3; c.o_mk = c.o_adv + 4;
  return c;
}

// PPO surrogate pieces shared by every head: given logp and the old logp of one action component
struct Surr { float surr, gl, ratio; };
__device__ inline Surr ppo_surr(float lp, float old_lp, float adv, const orl_ppo_hparams& hp) {
  float ratio = __expf(lp - old_lp);
  const float ratio_raw = ratio;
  float dr_eff = 1.f;
  if (hp.dual_clip_ppo && ratio > hp.dual_clip_coeff) { ratio = hp.dual_clip_coeff; dr_eff = 0.f; }
  const float s1 = ratio * adv;
  const float s2 = fminf(fmaxf(ratio, 1.f - hp.clip_param), 1.f + hp.clip_param) * adv;
  Surr o;
  o.surr = fminf(s1, s2);
  o.gl = -((s1 <= s2) ? adv : 0.f) * dr_eff * ratio_raw;  // d(-surr)/d logp
  o.ratio = ratio;
  if (hp.reserved & 2) {  // A2C (algorithms/a2c.py:88-98)
    o.surr = adv * lp; o.gl = -adv; o.ratio = 0.f;
  }
  return o;
}


// The policy part of prepare_loss + ACTLayer.evaluate_actions for ONE row: lg[0..n_out) = the head's logits (masked in
// place), r = the row's record.  mode 0 (training): dl[0..n_out) = d(policy_loss - entropy_coef * dist_entropy) / d logits
// already divided by the denominators, the row's contributions added to st_loss / st_ent / st_ratio / dls[16]
// (d logstd); mode 1 (evaluate_actions): logp_row[0..a_w), *ent_row_out.
__device__ inline void gen_policy_loss_row(const orl_head_desc& hd, float* lg, const float* __restrict__ logstd,
                                           const float* __restrict__ r, const GenCols& c, float inv_den, float inv_ent_den,
                                           const orl_ppo_hparams& hp, int mode, float* dl, float* logp_row,
                                           float* ent_row_out, float& st_loss, float& st_ent, float& st_ratio,
                                           float (&dls)[16]) {
  const int NT = hd.n_out;
  const float active = r[c.o_am];
  const float w = hp.use_policy_active_masks ? active : 1.f;
  const float adv = r[c.o_adv];
  if (hd.kind == ORL_HEAD_CATEGORICAL || hd.kind == ORL_HEAD_MULTI_DISCRETE) {
    const int nh = hd.kind == ORL_HEAD_CATEGORICAL ? 1 : hd.n_heads;
    int off = 0;
    float surr_sum = 0.f, ent_sum = 0.f, ratio_sum = 0.f;
    for (int h = 0; h < nh; ++h) {
      const int n = hd.kind == ORL_HEAD_CATEGORICAL ? NT : hd.nvec[h];
      float mx = -3.0e38f;
      for (int k = 0; k < n; ++k) {
        if (hd.kind == ORL_HEAD_CATEGORICAL && c.K > 0 && r[c.o_mk + k] == 0.f) lg[off + k] = -6e4f;
        mx = fmaxf(mx, lg[off + k]);
      }
      float se = 0.f;
      for (int k = 0; k < n; ++k) se += __expf(lg[off + k] - mx);
      const float lse = mx + __logf(se);
      const int a = (int)r[c.o_act + h];
      float ent = 0.f;
      for (int k = 0; k < n; ++k) {
        const float ell = lg[off + k] - lse;
        ent -= __expf(ell) * ell;
      }
      const float lp = lg[off + a] - lse;
      if (mode == 1) {
        logp_row[h] = lp;
        ent_sum += ent;
      } else {
        const Surr s = ppo_surr(lp, r[c.o_lp + h], adv, hp);
        surr_sum += s.surr; ratio_sum += s.ratio; ent_sum += ent;
        // MultiDiscrete: ACTLayer.evaluate_actions builds dist_entropy with torch.tensor([...]).mean()
        // (act.py:150-151), which DETACHES it - the entropy bonus has no gradient there
        const float ec = hd.kind == ORL_HEAD_CATEGORICAL ? hp.entropy_coef : 0.f;
        for (int k = 0; k < n; ++k) {
          const float ell = lg[off + k] - lse, p = __expf(ell);
          float d = s.gl * ((k == a ? 1.f : 0.f) - p) * inv_den + ec * p * (ell + ent) * inv_ent_den;
          const bool masked = hd.kind == ORL_HEAD_CATEGORICAL && c.K > 0 && r[c.o_mk + k] == 0.f;
          dl[off + k] = masked ? 0.f : w * d;
        }
      }
      off += n;
    }
    if (mode == 1) *ent_row_out = hd.kind == ORL_HEAD_CATEGORICAL ? ent_sum : ent_sum / (float)nh;
    else {
      st_loss += -surr_sum * w;
      st_ent += (hd.kind == ORL_HEAD_CATEGORICAL ? ent_sum : ent_sum / (float)nh) * w;
      st_ratio += ratio_sum;
    }
  } else if (hd.kind == ORL_HEAD_MIXED) {
    // Tuple(Box(cd), Discrete(n)) - ACTLayer.evaluate_actions' mixed branch (act.py:126-147) under prepare_loss
    // (ppo.py:302-361): ONE joint log-prob lp = sum_k lp_gauss_k + lp_cat against cd + 1 stored (identical) old
    // log-probs -> cd + 1 ratio / surrogate columns, summed (dim=-1); dist_entropy = 0.0025 * (Gaussian entropy
    // summed over dims, masked mean over rows - or the mean over rows AND dims without active masks) + 0.01 *
    // (Categorical entropy, masked mean), both differentiable.
    const int cd = hd.nvec[0], n = hd.nvec[1];
    const float ent_scale = hp.use_policy_active_masks ? 1.f : 1.f / (float)cd;
    float lp = 0.f, ent_g = 0.f;
    for (int k = 0; k < cd; ++k) {
      const float ls = logstd[k], sd = expf(ls), var = sd * sd;
      const float dmu = r[c.o_act + k] - lg[k];
      lp += -(dmu * dmu) / (2.f * var) - ls - 0.91893853320467274178f;
      ent_g += 1.41893853320467274178f + ls;
    }
    float mx = -3.0e38f;
    for (int k = 0; k < n; ++k) mx = fmaxf(mx, lg[cd + k]);
    float se = 0.f;
    for (int k = 0; k < n; ++k) se += __expf(lg[cd + k] - mx);
    const float lse = mx + __logf(se);
    const int a = (int)r[c.o_act + cd];
    float ent_c = 0.f;
    for (int k = 0; k < n; ++k) {
      const float ell = lg[cd + k] - lse;
      ent_c -= __expf(ell) * ell;
    }
    lp += lg[cd + a] - lse;
    const float ent_row = 0.0025f * ent_scale * ent_g + 0.01f * ent_c;
    if (mode == 1) {
      for (int k = 0; k <= cd; ++k) logp_row[k] = lp;
      *ent_row_out = ent_row;
    } else {
      float surr_sum = 0.f, ratio_sum = 0.f, gl = 0.f;
      for (int k = 0; k <= cd; ++k) {
        const Surr s = ppo_surr(lp, r[c.o_lp + k], adv, hp);
        surr_sum += s.surr; ratio_sum += s.ratio; gl += s.gl;
      }
      for (int k = 0; k < cd; ++k) {
        const float ls = logstd[k], sd = expf(ls), var = sd * sd;
        const float dmu = r[c.o_act + k] - lg[k];
        dl[k] = w * gl * dmu / var * inv_den;
        dls[k & 15] += w * (gl * (dmu * dmu / var - 1.f) * inv_den
                            - hp.entropy_coef * 0.0025f * ent_scale * inv_ent_den);
      }
      for (int k = 0; k < n; ++k) {
        const float ell = lg[cd + k] - lse, p = __expf(ell);
        dl[cd + k] = w * (gl * ((k == a ? 1.f : 0.f) - p) * inv_den
                          + hp.entropy_coef * 0.01f * p * (ell + ent_c) * inv_ent_den);
      }
      st_loss += -surr_sum * w; st_ent += ent_row * w; st_ratio += ratio_sum;
    }
  } else if (hd.kind == ORL_HEAD_GAUSSIAN) {
    // DiagGaussian, everything per action dimension (distributions.py:34-43, ppo.py:302-317)
    const float ent_scale = hp.use_policy_active_masks ? 1.f : 1.f / (float)NT;
    float surr_sum = 0.f, ent_sum = 0.f, ratio_sum = 0.f;
    for (int k = 0; k < NT; ++k) {
      const float ls = logstd[k], sd = expf(ls), var = sd * sd;
      const float dmu = r[c.o_act + k] - lg[k];
      const float lp = -(dmu * dmu) / (2.f * var) - ls - 0.91893853320467274178f;
      ent_sum += 1.41893853320467274178f + ls;
      if (mode == 1) { logp_row[k] = lp; continue; }
      const Surr s = ppo_surr(lp, r[c.o_lp + k], adv, hp);
      surr_sum += s.surr; ratio_sum += s.ratio;
      dl[k] = w * s.gl * dmu / var * inv_den;
      dls[k & 15] += w * (s.gl * (dmu * dmu / var - 1.f) * inv_den - hp.entropy_coef * ent_scale * inv_ent_den);
    }
    if (mode == 1) *ent_row_out = ent_sum;
    else { st_loss += -surr_sum * w; st_ent += ent_sum * w; st_ratio += ratio_sum; }
  }
}

// ValueNorm.normalize coefficients (valuenorm.py:79-91) from the running state
__device__ inline void gen_vn_coeffs(const float* __restrict__ vn_state, const orl_ppo_hparams& hp, float& vn_mean,
                                     float& vn_sd) {
  vn_mean = 0.f; vn_sd = 1.f;
  if (hp.use_valuenorm && vn_state != nullptr) {
    const float deb = fmaxf(vn_state[2], 1e-5f);
    vn_mean = vn_state[0] / deb;
    const float msq = vn_state[1] / deb;
    vn_sd = sqrtf(fmaxf(msq - vn_mean * vn_mean, 1e-2f));
  }
}

// cal_value_loss (ppo.py:178-220) for ONE row: returns d loss / d value (already * value_loss_coef / denominator), adds the
// row's loss to st.
__device__ inline float gen_value_loss_row(float v, const float* __restrict__ r, const GenCols& c, float vn_mean, float vn_sd,
                                           float inv_den, const orl_ppo_hparams& hp, float& st) {
  const float w = hp.use_value_active_masks ? r[c.o_am] : 1.f;
  const float vp = r[c.o_vp];
  float rt = r[c.o_rt];
  if (hp.use_valuenorm) rt = (rt - vn_mean) / vn_sd;
  const float dv = v - vp;
  const float dvc = fminf(fmaxf(dv, -hp.clip_param), hp.clip_param);
  const bool inside = (dv >= -hp.clip_param) && (dv <= hp.clip_param);
  const float e_c = rt - (vp + dvc), e_o = rt - v;
  auto hub = [&](float e, float& de) -> float {
    if (hp.use_huber_loss) {
      const float ae = fabsf(e), d = hp.huber_delta;
      if (ae <= d) { de = e; return e * e * 0.5f; }
      de = e > 0.f ? d : -d;
      return d * (ae - d * 0.5f);
    }
    de = e;
    return e * e * 0.5f;
  };
  float de_c, de_o;
  const float l_c = hub(e_c, de_c), l_o = hub(e_o, de_o);
  float vl, g;
  if (hp.use_clipped_value_loss) {
    vl = fmaxf(l_o, l_c);
    if (l_o > l_c) g = -de_o;
    else if (l_o < l_c) g = inside ? -de_c : 0.f;
    else g = -0.5f * de_o + (inside ? -0.5f * de_c : 0.f);
  } else {
    vl = l_o;
    g = -de_o;
  }
  st += vl * w;
  return w * g * hp.value_loss_coef * inv_den;
}

}  // namespace orl
