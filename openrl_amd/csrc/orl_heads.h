// orl_heads.h - action-head sampling shared by the rollout kernels (orl_act.hip, orl_rnn.hip).
#pragma once
#include "orl_common.h"
#include "orl_mlp.h"

namespace orl {

// Sample / evaluate the action head for batch row `row` of this lane (all 4 lanes of a row agree).
// Returns through act_out/logp_out (arrays of NO; categorical uses slot 0).
template <int NO, int HEAD>
__device__ inline void sample_head(float (&hd)[NO], int n_out, const float* lds_logstd, const float* amask_row,
                                   const float* forced_row, int deterministic, uint64_t seed, uint64_t grow,
                                   uint64_t rng_step, float (&act_out)[NO], float (&logp_out)[NO]) {
  if (HEAD == ORL_HEAD_CATEGORICAL) {
    const float lse = cat_lse<NO>(hd, n_out, amask_row);
    int a;
    if (deterministic) {
      a = cat_mode<NO>(hd, n_out);
    } else {
      float u;
      if (forced_row != nullptr) u = forced_row[0];
      else {
        const u4 r = philox4x32_10(seed, (uint32_t)grow, (uint32_t)(grow >> 32), (uint32_t)rng_step,
                                   (uint32_t)(rng_step >> 32) << 8);
        u = u01(r.x);
      }
      a = cat_sample<NO>(hd, n_out, lse, u);
    }
    act_out[0] = (float)a;
    logp_out[0] = pick<NO>(hd, a) - lse;
  } else {
    // DiagGaussian: std = exp(logstd), per-dimension log-prob (distributions.py:34-43, 75-98)
#pragma unroll
    for (int b = 0; b < (NO + 3) / 4; ++b) {
      float e[4] = {0.f, 0.f, 0.f, 0.f};
      if (!deterministic && forced_row == nullptr && 4 * b < n_out) {
        const u4 r = philox4x32_10(seed, (uint32_t)grow, (uint32_t)(grow >> 32), (uint32_t)rng_step,
                                   ((uint32_t)(rng_step >> 32) << 8) | (uint32_t)b);
        box_muller(r.x, r.y, e[0], e[1]);
        box_muller(r.z, r.w, e[2], e[3]);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int c = 4 * b + k;
        if (c < NO) {
          float av = 0.f, lp = 0.f;
          if (c < n_out) {
            const float ls = lds_logstd[c];
            const float sd = expf(ls);
            float eps = e[k];
            if (forced_row != nullptr && !deterministic) eps = forced_row[c];
            av = deterministic ? hd[c] : hd[c] + sd * eps;
            const float d = av - hd[c];
            // Normal.log_prob: -(x-mu)^2/(2 var) - log(sd) - log(sqrt(2 pi))
            lp = -(d * d) / (2.f * (sd * sd)) - ls - 0.91893853320467274178f;
          }
          act_out[c] = av;
          logp_out[c] = lp;
        }
      }
    }
  }
}

// Categorical sampling on the MFMA fragment of a wide head (head_mfma_T's accumulator BEFORE its trip through LDS):
// lane (j, q) holds the logits of classes 4q..4q+3 of row j and works on those four only; row-wide maxima / sums / the
// class that crosses the uniform are combined over the row's 4 lanes with permlane swaps.  Same semantics as
// cat_lse + cat_sample + pick (masked classes at -6e4, inverse CDF in class order, fall back to the last class of
// non-zero probability); the sums associate per lane first, so results agree with sample_head to fp32 round-off.
// Every lane of the row returns the action and its log-probability.
// (the mask of this lane's four classes as a value: the chain rollout derives it from the board word in registers)
__device__ inline void sample_cat_frag_v(const f32x4& lgv, int n_out, int q, const f32x4 mk, float u, float& act, float& logp);
__device__ inline void sample_cat_frag(const f32x4& lgv, int n_out, int q, const float* __restrict__ mask_row, float u,
                                       float& act, float& logp) {
  // mask_row = 16 floats, 16-byte aligned
  const f32x4 mk = mask_row != nullptr ? *(const f32x4*)(mask_row + 4 * q) : f32x4{1.f, 1.f, 1.f, 1.f};
  sample_cat_frag_v(lgv, n_out, q, mk, u, act, logp);
}
__device__ inline void sample_cat_frag_v(const f32x4& lgv, int n_out, int q, const f32x4 mk, float u, float& act, float& logp) {
  // Branch-free: the four classes of a lane go through selects (the short-circuit form compiled to ~20 exec-mask
  // branches on the step's serial chain).  Classes >= n_out drop out through val[] exactly as before (their terms are +0
  // in every sum).
  float lg[4];
  bool val[4];
  float mx = -3.0e38f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    val[r] = 4 * q + r < n_out;
    lg[r] = mk[r] == 0.f ? -6e4f : lgv[r];
    mx = val[r] ? fmaxf(mx, lg[r]) : mx;
  }
  mx = row_allmax(mx);
  float se = 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) se += val[r] ? __expf(lg[r] - mx) : 0.f;
  se = row_allsum(se);
  const float lse = mx + __logf(se);
  float p[4], ps = 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    p[r] = val[r] ? __expf(lg[r] - lse) : 0.f;
    ps += p[r];
  }
  float s4[4];
  row_gather4(ps, s4);
  const float ut = u * (((s4[0] + s4[1]) + s4[2]) + s4[3]);
  float cum = (q > 0 ? s4[0] : 0.f) + (q > 1 ? s4[1] : 0.f) + (q > 2 ? s4[2] : 0.f);
  float cand = 999.f, last = 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    cum += p[r];  // + 0 for a class that is not there
    last = p[r] > 0.f ? (float)(4 * q + r) : last;
    cand = (val[r] & (cand == 999.f) & (cum > ut)) ? (float)(4 * q + r) : cand;
  }
  float first = -cand;  // the lowest class whose cumulative probability exceeds u * total
  row_allmax2(first, last);
  first = -first;
  const float a = first == 999.f ? last : first;
  float pk = 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) pk = (float)(4 * q + r) == a ? lg[r] : pk;
  act = a;
  logp = row_allsum(pk) - lse;
}

// Evaluate GIVEN actions on the head outputs of batch row `row` (ACTLayer.evaluate_actions, utils/act.py:153-172):
// log-probs (categorical: slot 0; Gaussian: per dimension) and the row's entropy (Gaussian: summed over dimensions).
template <int NO, int HEAD>
__device__ inline void eval_head(float (&hd)[NO], int n_out, const float* lds_logstd, const float* amask_row,
                                 const float* given_row, float (&logp_out)[NO], float& ent_out) {
#pragma unroll
  for (int c = 0; c < NO; ++c) logp_out[c] = 0.f;
  if (HEAD == ORL_HEAD_CATEGORICAL) {
    const float lse = cat_lse<NO>(hd, n_out, amask_row);
    const int a = (int)given_row[0];
    logp_out[0] = pick<NO>(hd, a) - lse;
    float ent = 0.f;
#pragma unroll
    for (int c = 0; c < NO; ++c) {
      if (c < n_out) {
        const float ell = hd[c] - lse;
        ent -= __expf(ell) * ell;
      }
    }
    ent_out = ent;
  } else {
    float ent = 0.f;
#pragma unroll
    for (int c = 0; c < NO; ++c) {
      if (c < n_out) {
        const float ls = lds_logstd[c], sd = expf(ls);
        const float d = given_row[c] - hd[c];
        logp_out[c] = -(d * d) / (2.f * (sd * sd)) - ls - 0.91893853320467274178f;
        ent += 1.41893853320467274178f + ls;
      }
    }
    ent_out = ent;
  }
}

}  // namespace orl
