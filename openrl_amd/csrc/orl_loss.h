// orl_loss.h - per-row PPO losses and their head gradients (K9-K11), shared form for the recurrent kernels.
//   policy: clipped surrogate (+ dual clip) and entropy bonus        openrl/algorithms/ppo.py:302-330
//   value : clipped huber / mse value loss with ValueNorm targets    openrl/algorithms/ppo.py:178-220
// Same arithmetic as the in-line code of orl_ppo_tower.h (the feed-forward kernel keeps its own copy so that its
// register allocation is undisturbed).
#pragma once
#include "orl_common.h"
#include "orl_mlp.h"

namespace orl {

struct LossCols {
  int o_act, o_lp, o_adv, o_vp, o_rt, o_am, o_mk;  // record columns
  int K;                                           // action-mask width (0 = none)
};

struct LossStats {
  float active, rows, loss, ent, ratio;
};

__device__ inline float huber_or_mse_l(float e, float d, int use_huber, float& de) {
  if (use_huber) {
    const float ae = fabsf(e);
    if (ae <= d) { de = e; return e * e * 0.5f; }
    de = e > 0.f ? d : -d;
    return d * (ae - d * 0.5f);
  }
  de = e;
  return e * e * 0.5f;
}

// hd: head outputs of this lane's batch row; rec(col): record field of that row; logstd: [n_out] (Gaussian).
// dh[c] = d(total loss numerator)/d hd[c] (unnormalised: multiplied by the row weight, not yet divided by the
// masked-mean denominator); dls[c] = the same for logstd (Gaussian only).  `count` selects the one lane per row
// that adds the row to the statistics.
template <int HEAD, int NO, class REC>
__device__ inline void ppo_row_loss(float (&hd)[NO], int n_out, bool valid, REC rec, const LossCols& C,
                                    const orl_ppo_hparams& hp, float vn_mean, float vn_sd,
                                    const float* __restrict__ logstd, float (&dh)[NO], float (&dls)[NO],
                                    LossStats& st, bool count) {
#pragma unroll
  for (int c = 0; c < NO; ++c) { dh[c] = 0.f; dls[c] = 0.f; }
  const float active = valid ? rec(C.o_am) : 0.f;
  if (HEAD == ORL_HEAD_VALUE) {
    const float w = valid ? (hp.use_value_active_masks ? active : 1.f) : 0.f;
    const float v = hd[0];
    const float vp = valid ? rec(C.o_vp) : 0.f;
    float rt = valid ? rec(C.o_rt) : 0.f;
    if (hp.use_valuenorm) rt = (rt - vn_mean) / vn_sd;
    const float dv = v - vp;
    const float dvc = fminf(fmaxf(dv, -hp.clip_param), hp.clip_param);
    const bool inside = (dv >= -hp.clip_param) && (dv <= hp.clip_param);
    const float e_c = rt - (vp + dvc);
    const float e_o = rt - v;
    float de_c, de_o;
    const float l_c = huber_or_mse_l(e_c, hp.huber_delta, hp.use_huber_loss, de_c);
    const float l_o = huber_or_mse_l(e_o, hp.huber_delta, hp.use_huber_loss, de_o);
    float vl, g;
    if (hp.use_clipped_value_loss) {
      vl = fmaxf(l_o, l_c);
      if (l_o > l_c) g = -de_o;
      else if (l_o < l_c) g = inside ? -de_c : 0.f;
      else g = -0.5f * de_o + (inside ? -0.5f * de_c : 0.f);  // torch.max splits ties
    } else {
      vl = l_o;
      g = -de_o;
    }
    dh[0] = w * g * hp.value_loss_coef;
    if (count && valid) { st.active += active; st.rows += 1.f; st.loss += vl * w; }
  } else {
    const float w = valid ? (hp.use_policy_active_masks ? active : 1.f) : 0.f;
    const float adv = valid ? rec(C.o_adv) : 0.f;
    if (HEAD == ORL_HEAD_CATEGORICAL) {
      float mk[NO];
#pragma unroll
      for (int c = 0; c < NO; ++c) mk[c] = (C.K > 0 && valid && c < n_out) ? rec(C.o_mk + c) : 1.f;
      const float lse = cat_lse<NO>(hd, n_out, mk);
      const int act = valid ? (int)rec(C.o_act) : 0;
      const float old_lp = valid ? rec(C.o_lp) : 0.f;
      const float lp = pick<NO>(hd, act) - lse;
      float ent = 0.f;
      float p[NO];
#pragma unroll
      for (int c = 0; c < NO; ++c) {
        p[c] = 0.f;
        if (c < n_out) {
          const float ell = hd[c] - lse;
          p[c] = __expf(ell);
          ent -= p[c] * ell;
        }
      }
      float ratio = __expf(lp - old_lp);
      const float ratio_raw = ratio;
      float dr_eff = 1.f;
      if (hp.dual_clip_ppo && ratio > hp.dual_clip_coeff) { ratio = hp.dual_clip_coeff; dr_eff = 0.f; }
      const float s1 = ratio * adv;
      const float s2 = fminf(fmaxf(ratio, 1.f - hp.clip_param), 1.f + hp.clip_param) * adv;
      float surr = fminf(s1, s2);
      float gl = -((s1 <= s2) ? adv : 0.f) * dr_eff * ratio_raw;  // d(-surr)/d logp
      if (hp.reserved & 2) {  // A2C (algorithms/a2c.py:88-98): loss = -adv * logp, train_info ratio = 0
        surr = adv * lp;
        gl = -adv;
        ratio = 0.f;
      }
#pragma unroll
      for (int c = 0; c < NO; ++c) {
        if (c < n_out) {
          const float ell = hd[c] - lse;
          const float d = gl * ((c == act ? 1.f : 0.f) - p[c]) + hp.entropy_coef * p[c] * (ell + ent);
          dh[c] = (mk[c] == 0.f) ? 0.f : w * d;
        }
      }
      if (count && valid) {
        st.active += active; st.rows += 1.f; st.loss += -surr * w; st.ent += ent * w; st.ratio += ratio;
      }
    } else {
      const float ent_scale = hp.use_policy_active_masks ? 1.f : 1.f / (float)n_out;
      float surr_sum = 0.f, ent_sum = 0.f, ratio_sum = 0.f;
#pragma unroll
      for (int c = 0; c < NO; ++c) {
        if (c < n_out) {
          const float ls = logstd[c];
          const float sd = expf(ls);
          const float av = valid ? rec(C.o_act + c) : 0.f;
          const float old_lp = valid ? rec(C.o_lp + c) : 0.f;
          const float dmu = av - hd[c];
          const float var = sd * sd;
          const float lp = -(dmu * dmu) / (2.f * var) - ls - 0.91893853320467274178f;
          float ratio = __expf(lp - old_lp);
          const float ratio_raw = ratio;
          float dr_eff = 1.f;
          if (hp.dual_clip_ppo && ratio > hp.dual_clip_coeff) { ratio = hp.dual_clip_coeff; dr_eff = 0.f; }
          const float s1 = ratio * adv;
          const float s2 = fminf(fmaxf(ratio, 1.f - hp.clip_param), 1.f + hp.clip_param) * adv;
          float surr_c = fminf(s1, s2);
          float gl = -((s1 <= s2) ? adv : 0.f) * dr_eff * ratio_raw;
          if (hp.reserved & 2) {  // A2C
            surr_c = adv * lp;
            gl = -adv;
            ratio = 0.f;
          }
          surr_sum += surr_c;
          dh[c] = w * gl * dmu / var;
          dls[c] = w * (gl * (dmu * dmu / var - 1.f) - hp.entropy_coef * ent_scale);
          ent_sum += 1.41893853320467274178f + ls;
          ratio_sum += ratio;
        }
      }
      if (count && valid) {
        st.active += active; st.rows += 1.f; st.loss += -surr_sum * w; st.ent += ent_sum * w; st.ratio += ratio_sum;
      }
    }
  }
}

// Categorical PPO loss on logits DISTRIBUTED over the 4 lanes of a batch row: lane q holds classes 4q..4q+3 (the
// layout the MFMA head GEMM leaves them in); row-wide maxima / sums by permlane swaps.  Same formulas as
// ppo_row_loss<ORL_HEAD_CATEGORICAL>; dhv[r] = d(loss numerator)/d logit[4q+r].
template <class REC>
__device__ inline void ppo_cat_loss_dist(const f32x4& hv, int n_out, int q, bool valid, REC rec, const LossCols& C,
                                         const orl_ppo_hparams& hp, f32x4& dhv, LossStats& st, bool count) {
  dhv = f32x4{0.f, 0.f, 0.f, 0.f};
  const float active = valid ? rec(C.o_am) : 0.f;
  const float w = valid ? (hp.use_policy_active_masks ? active : 1.f) : 0.f;
  const float adv = valid ? rec(C.o_adv) : 0.f;
  float lg[4], mk[4];
  float mx = -3.0e38f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int c = 4 * q + r;
    mk[r] = (C.K > 0 && valid && c < n_out) ? rec(C.o_mk + c) : 1.f;
    lg[r] = (mk[r] == 0.f) ? -6e4f : hv[r];
    if (c < n_out) mx = fmaxf(mx, lg[r]);
  }
  mx = row_allmax(mx);
  float se = 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r)
    if (4 * q + r < n_out) se += __expf(lg[r] - mx);
  se = row_allsum(se);
  const float lse = mx + __logf(se);
  const int act = valid ? (int)rec(C.o_act) : 0;
  const float old_lp = valid ? rec(C.o_lp) : 0.f;
  float pk = 0.f, entp = 0.f;
  float p[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int c = 4 * q + r;
    p[r] = 0.f;
    if (c < n_out) {
      const float ell = lg[r] - lse;
      p[r] = __expf(ell);
      entp -= p[r] * ell;
      if (c == act) pk = lg[r];
    }
  }
  const float lp = row_allsum(pk) - lse;
  const float ent = row_allsum(entp);
  float ratio = __expf(lp - old_lp);
  const float ratio_raw = ratio;
  float dr_eff = 1.f;
  if (hp.dual_clip_ppo && ratio > hp.dual_clip_coeff) { ratio = hp.dual_clip_coeff; dr_eff = 0.f; }
  const float s1 = ratio * adv;
  const float s2 = fminf(fmaxf(ratio, 1.f - hp.clip_param), 1.f + hp.clip_param) * adv;
  float surr = fminf(s1, s2);
  float gl = -((s1 <= s2) ? adv : 0.f) * dr_eff * ratio_raw;  // d(-surr)/d logp
  if (hp.reserved & 2) {  // A2C
    surr = adv * lp;
    gl = -adv;
    ratio = 0.f;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int c = 4 * q + r;
    if (c < n_out) {
      const float ell = lg[r] - lse;
      const float d = gl * ((c == act ? 1.f : 0.f) - p[r]) + hp.entropy_coef * p[r] * (ell + ent);
      dhv[r] = (mk[r] == 0.f) ? 0.f : w * d;
    }
  }
  if (count && valid) {
    st.active += active; st.rows += 1.f; st.loss += -surr * w; st.ent += ent * w; st.ratio += ratio;
  }
}

}  // namespace orl
