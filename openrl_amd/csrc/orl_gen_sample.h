// orl_gen_sample.h - action sampling of the general tower path as a per-row device function, shared by gen_sample_kernel
// (orl_gen.hip) and the one-launch act kernel (orl_gen_fused.hip): identical arithmetic, identical Philox counters.
#pragma once
#include "orl_common.h"
#include "orl_mlp.h"

namespace orl {

constexpr int GEN_MAX_OUT = 64;  // logits per row the loss / sampling code keeps in registers

// One row: logits -> (actions, log-probs).  ``lg`` is a WORKING copy of the row's hd.n_out logits (masked entries are
// overwritten): a local array in gen_sample_kernel, the row of the LDS logits tile in the act kernel.  ``amask`` /
// ``forced`` / ``actions`` / ``logp`` point at THIS row; ``grow`` is the row's index in the Philox counter.
template <typename LgPtr>
__device__ inline void gen_sample_row(const orl_head_desc& hd, LgPtr lg,
                                      const float* __restrict__ logstd, const float* __restrict__ amask_row,
                                      int deterministic, uint64_t seed, uint64_t grow, uint64_t step,
                                      const float* __restrict__ forced_row, float* __restrict__ actions_row,
                                      float* __restrict__ logp_row) {
  const int NT = hd.n_out;
  auto draw = [&](int b) -> u4 {  // block b of 4 words: the counter layout of orl_heads.h::sample_head
    return philox4x32_10(seed, (uint32_t)grow, (uint32_t)(grow >> 32), (uint32_t)step,
                         ((uint32_t)(step >> 32) << 8) | (uint32_t)b);
  };
  auto word = [&](const u4& r, int k) -> uint32_t { return k == 0 ? r.x : k == 1 ? r.y : k == 2 ? r.z : r.w; };
  if (hd.kind == ORL_HEAD_CATEGORICAL || hd.kind == ORL_HEAD_MULTI_DISCRETE) {
    const int nh = hd.kind == ORL_HEAD_CATEGORICAL ? 1 : hd.n_heads;
    int off = 0;
    u4 r4 = draw(0);
    for (int h = 0; h < nh; ++h) {
      const int n = hd.kind == ORL_HEAD_CATEGORICAL ? NT : hd.nvec[h];
      float mx = -3.0e38f;
      for (int k = 0; k < n; ++k) {
        if (hd.kind == ORL_HEAD_CATEGORICAL && amask_row && amask_row[k] == 0.f) lg[off + k] = -6e4f;
        mx = fmaxf(mx, lg[off + k]);
      }
      float se = 0.f;
      for (int k = 0; k < n; ++k) se += __expf(lg[off + k] - mx);
      const float lse = mx + __logf(se);
      int a = 0;
      if (deterministic) {
        for (int k = 1; k < n; ++k) if (lg[off + k] > lg[off + a]) a = k;
      } else {
        if (h > 0 && (h & 3) == 0) r4 = draw(h >> 2);
        const float u = forced_row ? forced_row[h] : u01(word(r4, h & 3));
        // inverse CDF over softmax probabilities (cat_sample of orl_mlp.h)
        float tot = 0.f;
        for (int k = 0; k < n; ++k) tot += __expf(lg[off + k] - lse);
        const float ut = u * tot;
        float cum = 0.f;
        int last = 0;
        a = -1;
        for (int k = 0; k < n; ++k) {
          const float p = __expf(lg[off + k] - lse);
          cum += p;
          if (p > 0.f) last = k;
          if (a < 0 && cum > ut) a = k;
        }
        if (a < 0) a = last;
      }
      actions_row[h] = (float)a;
      logp_row[h] = lg[off + a] - lse;
      off += n;
    }
  } else if (hd.kind == ORL_HEAD_MIXED) {
    // Tuple(Box(cd), Discrete(n)): ACTLayer.forward's mixed branch (act.py:46-63) - the Gaussian part, then the
    // Categorical (no action masks there), ONE joint log-prob (the sum) in every stored column.  forced_row = cd normal
    // deviates + the categorical's uniform.  Philox blocks 0 .. ceil(cd/4)-1 feed the Gaussian, the next one the class.
    const int cd = hd.nvec[0], n = hd.nvec[1];
    float joint = 0.f;
    for (int b = 0; 4 * b < cd; ++b) {
      float e[4] = {0.f, 0.f, 0.f, 0.f};
      if (!deterministic && forced_row == nullptr) {
        const u4 r = draw(b);
        box_muller(r.x, r.y, e[0], e[1]);
        box_muller(r.z, r.w, e[2], e[3]);
      }
      for (int k = 0; k < 4 && 4 * b + k < cd; ++k) {
        const int cdim = 4 * b + k;
        const float ls = logstd[cdim], sd = expf(ls);
        float eps = e[k];
        if (forced_row != nullptr && !deterministic) eps = forced_row[cdim];
        const float av = deterministic ? lg[cdim] : lg[cdim] + sd * eps;
        const float d = av - lg[cdim];
        actions_row[cdim] = av;
        joint += -(d * d) / (2.f * (sd * sd)) - ls - 0.91893853320467274178f;
      }
    }
    float mx = -3.0e38f;
    for (int k = 0; k < n; ++k) mx = fmaxf(mx, lg[cd + k]);
    float se = 0.f;
    for (int k = 0; k < n; ++k) se += __expf(lg[cd + k] - mx);
    const float lse = mx + __logf(se);
    int a = 0;
    if (deterministic) {
      for (int k = 1; k < n; ++k) if (lg[cd + k] > lg[cd + a]) a = k;
    } else {
      const float u = forced_row ? forced_row[cd] : u01(draw((cd + 3) >> 2).x);
      float tot = 0.f;
      for (int k = 0; k < n; ++k) tot += __expf(lg[cd + k] - lse);
      const float ut = u * tot;
      float cum = 0.f;
      int last = 0;
      a = -1;
      for (int k = 0; k < n; ++k) {
        const float p = __expf(lg[cd + k] - lse);
        cum += p;
        if (p > 0.f) last = k;
        if (a < 0 && cum > ut) a = k;
      }
      if (a < 0) a = last;
    }
    actions_row[cd] = (float)a;
    joint += lg[cd + a] - lse;
    for (int k = 0; k <= cd; ++k) logp_row[k] = joint;
  } else if (hd.kind == ORL_HEAD_GAUSSIAN) {
    for (int b = 0; 4 * b < NT; ++b) {
      float e[4] = {0.f, 0.f, 0.f, 0.f};
      if (!deterministic && forced_row == nullptr) {
        const u4 r = draw(b);
        box_muller(r.x, r.y, e[0], e[1]);
        box_muller(r.z, r.w, e[2], e[3]);
      }
      for (int k = 0; k < 4 && 4 * b + k < NT; ++k) {
        const int cdim = 4 * b + k;
        const float ls = logstd[cdim], sd = expf(ls);
        float eps = e[k];
        if (forced_row != nullptr && !deterministic) eps = forced_row[cdim];
        const float av = deterministic ? lg[cdim] : lg[cdim] + sd * eps;
        const float d = av - lg[cdim];
        actions_row[cdim] = av;
        logp_row[cdim] = -(d * d) / (2.f * (sd * sd)) - ls - 0.91893853320467274178f;
      }
    }
  }
}

}  // namespace orl
