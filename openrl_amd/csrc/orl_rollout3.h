// orl_rollout3.h - round 6: the chain rollout (orl_rollout2.h) with the policy tower SPECULATED over the two actions of a
// Discrete(2) env (CartPole-v1 and the synthetic env of its shape - BASELINE.json's configs[1]).  Included by orl_act.hip
// after orl_rollout2.h, inside namespace orl.
//
// In orl_rollout2.h a step is one chain: tower(obs_t) -> head + sampler -> env -> obs_{t+1} -> tower(obs_{t+1}) ...: 3 300 cycles,
// of which the tower is 1 480 and the rest of the step 1 800.  With two actions the next observation has only two candidates,
// obs_{t+1}^(b) = env(s_t, b), and both are known as soon as s_t is - one step EARLIER than a_t.  So:
//   * waves 0-3 / 4-7 ("trunk groups" A / B) run the policy tower on candidate 0 / 1 of step t while
//   * wave 8 ("T") is still finishing step t - 1: it adds the four partials of the candidate that step t - 2's action selected
//     (per row), LayerNorm 2 + head from partials, samples a_{t-1}, and
//   * wave 9 ("E") turns a_{t-1} into the real state s_t (a select between its two candidate states), stages reward / done,
//     publishes the real observation, and computes the two candidates of step t + 1 (the action-dependent half of the env step,
//     twice; the action-independent half - cos / sin - was prepared for both candidates off the chain).
// The towers and the tail overlap: a step costs max(tower on two groups sharing the four SIMDs, tail) + one hand-over instead of
// their sum (measured beforehand with a shadow group: two trunks on the same SIMDs cost 1 910 cycles instead of 1 480,
// profiles/r06_experiments.md section 7).  Same arithmetic per value as orl_rollout2.h - the candidate that is selected is
// computed exactly as the chain kernel computes it - so actions, log-probabilities, observations and env state are bit-identical
// to it (tests/test_rollout_gpu.py::test_speculative_rollout_kernel_equals_the_chain_kernel).  Nothing is guessed: both
// candidates are evaluated, the sampled action picks.  For the synthetic env (whose observations ignore the action) the two
// candidates coincide and both are still evaluated: one code path, the same dependency structure as CartPole.
//
// Other roles as in orl_rollout2.h: wave 10 noise (two steps ahead), wave 11 the buffer stores, waves 12-14 the critic on the
// REAL observations, wave 15 a generator (synthetic env: reward + next observation; CartPole: the reset states of the next two
// episodes of every row).  All hand-overs: single-writer LDS words, bounded polls.
#pragma once

constexpr int RO3_THREADS = 1024;
enum { R3_CAND = 0, R3_ACT = 1, R3_REAL = 2, R3_STAGE = 3, R3_NOISE = 4, R3_GEN = 5, R3_STORED = 6, R3_ERR = 7, R3_PART = 8,
       R3_CRIT = 16, R3_EP = 19, R3_WORDS = 20 };
constexpr int R3_PSLOT = 4 * TILE_B * 4;  // floats of one wave's partial record: [q][row] x {sum z, sum z^2, W3g[0] . z, W3g[1] . z}

struct Ro3Lds {
  int critic, obs, cand, part, z2, noise, gen, stage, act, ep, ctr, total;  // float offsets
};
__host__ __device__ inline Ro3Lds ro3_lds(int policy_total, int critic_total) {
  Ro3Lds L;
  int o = policy_total;
  L.critic = o; o += critic_total;
  L.obs = o; o += RO2_ORING * TILE_B * 4;          // real observations (critic, stores)
  L.cand = o; o += 2 * 2 * TILE_B * 4;             // [step parity][action] candidate observations
  L.part = o; o += 2 * 8 * R3_PSLOT;               // [step parity][group * 4 + wave]
  L.z2 = o; o += 2 * 2 * TILE_B * GS;              // [step parity][group] z tiles (guarded LayerNorm 2 path)
  L.noise = o; o += RO2_RING * TILE_B;
  L.gen = o; o += RO2_RING * TILE_B * 8;
  L.stage = o; o += RO2_RING * TILE_B * 4;         // {action, log-prob, reward, done}
  L.act = o; o += RO2_RING * TILE_B;
  L.ep = o; o += TILE_B;
  L.ctr = o; o += R3_WORDS;
  L.total = o;
  return L;
}

template <int ENV>
__global__ __launch_bounds__(RO3_THREADS) void rollout3_kernel(RolloutArgs A) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NO = 2;
  const orl_buffer_ptrs& b = A.r.buf;
  const int N = b.N, T = b.T;
  const TowerLayout tlp(A.pnet), tlc(A.cnet);
  const TowerLds twp(4, 2, false, false, false);
  const TowerLds twc(4, 1, false, false, false, true);  // the critic's W2 as bf16 split images
  stage_tower(smem, A.ptheta, tlp, twp, false, threadIdx.x, blockDim.x, false, false, true);
  stage_tower(smem + twp.total, A.ctheta, tlc, twc, false, threadIdx.x, blockDim.x, false, true, true);
  const Ro3Lds L = ro3_lds(twp.total, twc.total);
  float* s_obs = smem + L.obs;
  float* s_cand = smem + L.cand;
  unsigned* ctr = (unsigned*)(smem + L.ctr);
  unsigned* err = ctr + R3_ERR;
  const int wave = threadIdx.x >> 6, l = threadIdx.x & 63, j = l & 15, q = l >> 4;
  const int n0 = blockIdx.x * TILE_B;
  const int n = n0 + j;
  const bool ok = n < N;
  const int nrow = (N - n0) < TILE_B ? (N - n0) : TILE_B;
  const bool sep_c = b.critic_obs != b.policy_obs;
  const int D = A.pnet.obs_dim;  // <= 4 (columns beyond D are zero)
  // observation 0 comes from the buffer: real slot 0, and both candidates of step 0
  for (int e = threadIdx.x; e < TILE_B * 4; e += blockDim.x) {
    const int jj = e >> 2, k = e & 3;
    const float v = (n0 + jj < N && k < D) ? b.policy_obs[(size_t)(n0 + jj) * D + k] : 0.f;
    s_obs[e] = v;
    s_cand[e] = v;
    s_cand[TILE_B * 4 + e] = v;
  }
  if (threadIdx.x < R3_WORDS) ctr[threadIdx.x] = (threadIdx.x == R3_CAND || threadIdx.x == R3_REAL) ? 1u : 0u;
  __syncthreads();
  const uint64_t tg0 = A.r.rng_step0;

  if (wave < 8) {
    // ================================================================ trunk groups: the policy tower on candidate g ===========
    __builtin_amdgcn_s_setprio(3);
    const int g = wave >> 2, gw = wave & 3;
    CoopRegs creg;
    coop_load(smem, twp, gw, j, q, creg);
    f32x4 w3s[NO];
#pragma unroll
    for (int c = 0; c < NO; ++c) w3s[c] = *(const f32x4*)(smem + twp.W3 + c * HID + 16 * gw + 4 * q);
    for (int t = 0; t < T; ++t) {
      const float* cand = s_cand + ((t & 1) * 2 + g) * TILE_B * 4;
      float xb;
      for (unsigned spins = 0;; ++spins) {
        const unsigned c = ro2_ld(ctr + R3_CAND);
        xb = ro2_ldf(cand + j * 4 + q);
        if ((int)c >= t + 1) break;
        if (spins > RO2_MAX_SPINS) { ro2_fail(err); break; }
      }
      asm volatile("" ::: "memory");
      f32x4 x[4];
      float rstd;
#pragma unroll
      for (int m = 0; m < 4; ++m) x[m] = ORL_MFMA(creg.w1[m][0], xb, creg.b1a[m]);
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) x[m][r] = fmaxf(x[m][r], 0.f);
      ln_normalize_T(x, rstd);
      f32x4 z = creg.b2, z2b = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          z = ORL_MFMA(creg.w2[mi][r], x[mi][r], z);
          z2b = ORL_MFMA(creg.w2[mi + 2][r], x[mi + 2][r], z2b);
        }
      }
      z = z + z2b;
      f32x4 pv;
      pv[0] = (z[0] + z[1]) + (z[2] + z[3]);
      pv[1] = (z[0] * z[0] + z[1] * z[1]) + (z[2] * z[2] + z[3] * z[3]);
#pragma unroll
      for (int c = 0; c < NO; ++c)
        pv[2 + c] = (w3s[c][0] * z[0] + w3s[c][1] * z[1]) + (w3s[c][2] * z[2] + w3s[c][3] * z[3]);
      *(f32x4*)(smem + L.part + ((t & 1) * 8 + g * 4 + gw) * R3_PSLOT + (q * TILE_B + j) * 4) = pv;
      *(f32x4*)(smem + L.z2 + ((t & 1) * 2 + g) * TILE_B * GS + j * GS + 16 * gw + 4 * q) = z;
      ro2_post(ctr + R3_PART + g * 4 + gw, t + 1);
    }
    return;
  }
  if (wave == 8) {
    // ================================================================ T: head + sampler of the selected candidate ============
    __builtin_amdgcn_s_setprio(3);
    float sw3[NO], b3v[NO];
#pragma unroll
    for (int c = 0; c < NO; ++c) {
      sw3[c] = wave_sum(smem[twp.W3 + c * HID + l]);
      b3v[c] = smem[twp.b3 + c];
    }
    float a_prev = 0.f;  // the action that selected this step's candidate (step 0: both candidates are observation 0)
    for (int t = 0; t < T; ++t) {
      const float* pb = smem + L.part + (t & 1) * 8 * R3_PSLOT;
      f32x4 pa[4], pbb[4];
      float u;
      for (unsigned spins = 0;; ++spins) {
        const u32x4 fa = ro2_ld4u(ctr + R3_PART), fb = ro2_ld4u(ctr + R3_PART + 4), fs = ro2_ld4u(ctr + R3_NOISE);
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          pa[w] = ro2_ld4f(pb + w * R3_PSLOT + (q * TILE_B + j) * 4);
          pbb[w] = ro2_ld4f(pb + (4 + w) * R3_PSLOT + (q * TILE_B + j) * 4);
        }
        u = ro2_ldf(smem + L.noise + (t & 3) * TILE_B + j);
        const bool ready = (int)fa[0] > t && (int)fa[1] > t && (int)fa[2] > t && (int)fa[3] > t && (int)fb[0] > t &&
                           (int)fb[1] > t && (int)fb[2] > t && (int)fb[3] > t && (int)fs[0] > t && (int)fs[2] >= t - 3;
        if (ready) break;
        if (spins > RO2_MAX_SPINS) { ro2_fail(err); break; }
      }
      asm volatile("" ::: "memory");
      const bool sel = a_prev != 0.f;
      f32x4 a4[4];
#pragma unroll
      for (int w = 0; w < 4; ++w)
#pragma unroll
        for (int r = 0; r < 4; ++r) a4[w][r] = sel ? pbb[w][r] : pa[w][r];
      const f32x4 sv = (a4[0] + a4[1]) + (a4[2] + a4[3]);
      float s1 = sv[0], s2 = sv[1], d0 = sv[2], d1 = sv[3];
      row_allsum2(s1, s2);
      row_allsum2(d0, d1);
      float hd[NO] = {d0, d1};
      const float mean2 = s1 * (1.0f / 64.0f), ex2 = s2 * (1.0f / 64.0f);
      const float var2 = ex2 - mean2 * mean2;
      if (__builtin_expect(__builtin_amdgcn_ballot_w64(var2 * ORL_LN_GUARD < ex2) != 0ull, 0)) {
        f32x4 zz[4];
        float r2;
        const float* zt = smem + L.z2 + ((t & 1) * 2 + (sel ? 1 : 0)) * TILE_B * GS;
#pragma unroll
        for (int m = 0; m < 4; ++m) zz[m] = *(const f32x4*)(zt + j * GS + 16 * m + 4 * q);
        ln_normalize_T(zz, r2);
        float h2[NO];
        head_T<NO>(smem + twp.W3, smem + twp.b3, 2, zz, q, h2);
        hd[0] = h2[0]; hd[1] = h2[1];
      } else {
        const float rstd2 = __builtin_amdgcn_rsqf(fmaxf(var2, 0.f) + 1e-5f);
        const float mr = mean2 * rstd2;
#pragma unroll
        for (int c = 0; c < NO; ++c) hd[c] = (hd[c] * rstd2 - mr * sw3[c]) + b3v[c];
      }
      float act, lp;
      ro2_sample_cat2(hd[0], hd[1], u, act, lp);
      a_prev = act;
      if (q == 0) {
        smem[L.act + (t & 3) * TILE_B + j] = act;
        *(f32x2*)(smem + L.stage + ((t & 3) * TILE_B + j) * 4) = f32x2{act, lp};
      }
      ro2_post(ctr + R3_ACT, t + 1);
    }
    return;
  }
  if (wave == 9) {
    // ================================================================ E: the env ================================================
    __builtin_amdgcn_s_setprio(2);
    constexpr int SW = ENV == ORL_ENV_SYNTH ? SYNTH_STATE_W : CARTPOLE_STATE_W;
    float est[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float ep_ret = 0.f, ep_len = 0.f, fin_ret = 0.f, fin_cnt = 0.f;
    if (ok) {
#pragma unroll
      for (int k = 0; k < 8; ++k) est[k] = k < SW ? A.r.env_state[(size_t)n * SW + k] : 0.f;
      ep_ret = A.r.ep_stats[n * 4 + 0]; ep_len = A.r.ep_stats[n * 4 + 1];
      fin_ret = A.r.ep_stats[n * 4 + 2]; fin_cnt = A.r.ep_stats[n * 4 + 3];
    }
    // the two candidates of the NEXT step: state, whether that step ends the episode, and (CartPole) their cos / sin half
    float c0[4], c1[4];
    bool dn0 = false, dn1 = false;
    CartPolePre pr0, pr1, prs;
    float steps = ENV == ORL_ENV_SYNTH ? est[0] : est[4];
    float ep = ENV == ORL_ENV_CARTPOLE ? est[5] : 0.f;
    float s[4] = {est[0], est[1], est[2], est[3]};  // the real state s_t (CartPole); the synthetic env's "state" is its step counter
    const float limit = (float)A.r.episode_limit;
    auto candidates = [&](const float (&rs)[4], int slot) {
      // env(s, b) for b = 0, 1 from the real state s with its prepared half prs; rs = the first observation of the next episode
      if (ENV == ORL_ENV_CARTPOLE) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { c0[k] = s[k]; c1[k] = s[k]; }
        const bool t0 = cartpole_post(c0, prs, 0), t1 = cartpole_post(c1, prs, 1);
        const bool trunc = steps + 1.f >= limit;
        dn0 = t0 || trunc; dn1 = t1 || trunc;
#pragma unroll
        for (int k = 0; k < 4; ++k) { c0[k] = dn0 ? rs[k] : c0[k]; c1[k] = dn1 ? rs[k] : c1[k]; }
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) { c0[k] = rs[k]; c1[k] = rs[k]; }  // (rs = the generated next observation)
        dn0 = dn1 = steps + 1.f >= limit;
      }
      if (q == 0) {
        *(f32x4*)(s_cand + (slot * 2 + 0) * TILE_B * 4 + j * 4) = ok ? f32x4{c0[0], c0[1], c0[2], c0[3]} : f32x4{0.f, 0.f, 0.f, 0.f};
        *(f32x4*)(s_cand + (slot * 2 + 1) * TILE_B * 4 + j * 4) = ok ? f32x4{c1[0], c1[1], c1[2], c1[3]} : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    };
    // prologue: the candidates of step 1 = env(s_0, b)
    {
      float rs[4];
      if (ENV == ORL_ENV_CARTPOLE) {
        prs = cartpole_pre(s);
        cartpole_reset(A.r.env_seed, (uint32_t)n, (uint32_t)(ep + 1.f), rs);
      } else {
        ro2_wait(ctr + R3_GEN, 1, err);
        const f32x4 o = *(const f32x4*)(smem + L.gen + j * 8);
        rs[0] = o[0]; rs[1] = o[1]; rs[2] = o[2]; rs[3] = o[3];
      }
      candidates(rs, 1);
      if (q == 0) smem[L.ep + j] = ep;
      ro2_post(ctr + R3_EP, 1);
      ro2_post(ctr + R3_CAND, 2);
      if (ENV == ORL_ENV_CARTPOLE) { pr0 = cartpole_pre(c0); pr1 = cartpole_pre(c1); }
    }
    for (int t = 0; t < T; ++t) {
      // ---- a_t arrives: the real step t
      float a;
      for (unsigned spins = 0;; ++spins) {
        const u32x4 f0 = ro2_ld4u(ctr), fs = ro2_ld4u(ctr + R3_NOISE), fc = ro2_ld4u(ctr + R3_CRIT);
        a = ro2_ldf(smem + L.act + (t & 3) * TILE_B + j);
        const bool ready = (int)f0[R3_ACT] > t && (int)fs[2] >= t - 7 && (int)fc[0] >= t - 6 && (int)fc[1] >= t - 6 &&
                           (int)fc[2] >= t - 6;
        if (ready) break;
        if (spins > RO2_MAX_SPINS) { ro2_fail(err); break; }
      }
      asm volatile("" ::: "memory");
      const bool sel = a != 0.f;
      const bool done = sel ? dn1 : dn0;
#pragma unroll
      for (int k = 0; k < 4; ++k) s[k] = sel ? c1[k] : c0[k];
      if (ENV == ORL_ENV_CARTPOLE) {
        prs.costh = sel ? pr1.costh : pr0.costh; prs.sinth = sel ? pr1.sinth : pr0.sinth;
        prs.t1 = sel ? pr1.t1 : pr0.t1; prs.den = sel ? pr1.den : pr0.den;
      }
      steps = done ? 0.f : steps + 1.f;
      ep += done ? 1.f : 0.f;
      float rew = 1.0f;
      if (ENV == ORL_ENV_SYNTH) {
        ro2_wait<false>(ctr + R3_GEN, t + 1, err);
        rew = smem[L.gen + ((t & 3) * TILE_B + j) * 8 + 4];
      }
      ep_ret += rew; ep_len += 1.f;
      if (done) { fin_ret += ep_ret; fin_cnt += 1.f; ep_ret = 0.f; ep_len = 0.f; }
      if (q == 0) {
        *(f32x2*)(smem + L.stage + ((t & 3) * TILE_B + j) * 4 + 2) = f32x2{rew, done ? 1.f : 0.f};
        *(f32x4*)(s_obs + ((t + 1) & (RO2_ORING - 1)) * TILE_B * 4 + j * 4) =
            ok ? f32x4{s[0], s[1], s[2], s[3]} : f32x4{0.f, 0.f, 0.f, 0.f};
      }
      ro2_post(ctr + R3_STAGE, t + 1);
      ro2_post(ctr + R3_REAL, t + 2);
      // ---- the candidates of step t + 2 = env(s_{t+1}, b)
      float rs[4];
      if (ENV == ORL_ENV_CARTPOLE) {
        ro2_wait<false>(ctr + R3_GEN, t + 1, err);  // the generator's record t: reset states of episodes ep_t + 1, ep_t + 2
        const float* gr = smem + L.gen + ((t & 3) * TILE_B + j) * 8 + (done ? 4 : 0);
        const f32x4 o = *(const f32x4*)gr;
        rs[0] = o[0]; rs[1] = o[1]; rs[2] = o[2]; rs[3] = o[3];
      } else {
        ro2_wait<false>(ctr + R3_GEN, t + 2, err);  // observation t + 2 = the generator's record t + 1
        const f32x4 o = *(const f32x4*)(smem + L.gen + (((t + 1) & 3) * TILE_B + j) * 8);
        rs[0] = o[0]; rs[1] = o[1]; rs[2] = o[2]; rs[3] = o[3];
      }
      candidates(rs, t & 1);  // slot (t + 2) & 1
      if (q == 0) smem[L.ep + j] = ep;
      ro2_post(ctr + R3_EP, t + 2);
      ro2_post(ctr + R3_CAND, t + 3);
      if (ENV == ORL_ENV_CARTPOLE) { pr0 = cartpole_pre(c0); pr1 = cartpole_pre(c1); }  // off the chain: the next selection's half
    }
    const bool poisoned = ro2_ld(err) != 0u;
    if (q == 0 && ok) {
      if (ENV == ORL_ENV_CARTPOLE) {
#pragma unroll
        for (int k = 0; k < 4; ++k) A.r.env_state[(size_t)n * SW + k] = s[k];
        A.r.env_state[(size_t)n * SW + 4] = steps;
        A.r.env_state[(size_t)n * SW + 5] = ep;
      } else {
        A.r.env_state[(size_t)n * SW + 0] = steps;
      }
      A.r.ep_stats[n * 4 + 0] = ep_ret; A.r.ep_stats[n * 4 + 1] = ep_len;
      A.r.ep_stats[n * 4 + 2] = poisoned ? u2f(0x7fc00000u) : fin_ret; A.r.ep_stats[n * 4 + 3] = fin_cnt;
    }
    return;
  }
  __builtin_amdgcn_s_setprio(0);
  if (wave == 10) {
    // ================================================================ sampling noise, ahead of the chain ====================
    for (int t = 0; t < T; ++t) {
      ro2_wait(ctr + R3_ACT, t - 3, err);
      const uint64_t tg = tg0 + (uint64_t)t;
      const u4 r = philox4x32_10(A.r.act_seed, (uint32_t)n, (uint32_t)((uint64_t)n >> 32), (uint32_t)tg, (uint32_t)(tg >> 32) << 8);
      if (q == 0) smem[L.noise + (t & 3) * TILE_B + j] = u01(r.x);
      ro2_post(ctr + R3_NOISE, t + 1);
    }
  } else if (wave == 11) {
    // ================================================================ the step's rows of the rollout buffer ==================
    for (int t = 0; t < T; ++t) {
      ro2_wait(ctr + R3_ACT, t + 1, err);
      ro2_wait(ctr + R3_STAGE, t + 1, err);
      ro2_wait(ctr + R3_REAL, t + 2, err);
      const float* stg = smem + L.stage + (t & 3) * TILE_B * 4;
      const float* nxt = s_obs + ((t + 1) & (RO2_ORING - 1)) * TILE_B * 4;
      const size_t r0 = (size_t)t * N + n0, r1 = (size_t)(t + 1) * N + n0;
      if (l < nrow) {
        const f32x4 sr = *(const f32x4*)(stg + l * 4);
        A.r.actions[r0 + l] = sr[0];
        A.r.action_log_probs[r0 + l] = sr[1];
        b.rewards[r0 + l] = sr[2];
        b.masks[r1 + l] = sr[3] != 0.f ? 0.f : 1.f;
        b.active_masks[r1 + l] = 1.f;
        b.bad_masks[r1 + l] = 1.f;
      }
      for (int e = l; e < nrow * D; e += 64) {
        const int rr = e / D, d = e - rr * D;
        const float v = nxt[rr * 4 + d];
        b.policy_obs[r1 * D + e] = v;
        if (sep_c) b.critic_obs[r1 * D + e] = v;
      }
      if (b.action_masks != nullptr)
        for (int e = l; e < nrow * b.K; e += 64) b.action_masks[r1 * b.K + e] = 1.f;
      ro2_post(ctr + R3_STORED, t + 1);
    }
    if (ro2_ld(err) != 0u && l < nrow) b.rewards[(size_t)n0 + l] = u2f(0x7fc00000u);  // a poll timed out: the tile's data are void
  } else if (wave >= 12 && wave <= 14) {
    // ================================================================ the critic on the REAL observations =====================
    const int c = wave - 12;
    const float* lc = smem + L.critic;
    f32x4 w3[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) w3[m] = *(const f32x4*)(lc + twc.W3 + 16 * m + 4 * q);
    const float b3 = lc[twc.b3];
    for (int t = c; t <= T; t += 3) {
      ro2_wait(ctr + R3_REAL, t + 1, err);
      const float xb = s_obs[(t & (RO2_ORING - 1)) * TILE_B * 4 + j * 4 + q];
      ro2_post(ctr + R3_CRIT + c, t + 1);
      f32x4 z[4];
      load_vec_T(lc + twc.b1, q, z);
#pragma unroll
      for (int m = 0; m < 4; ++m) z[m] = ORL_MFMA(lc[twc.W1 + (16 * m + j) * 4 + q], xb, z[m]);
      relu_T(z);
      float rstd;
      ln_normalize_T(z, rstd);
      f32x4 acc[4];
      load_vec_T(lc + twc.b2, q, acc);
      {
        u32x4 xs[2][3];
        split_T(z, xs);
        mm64_T_split((const unsigned short*)(lc + twc.W2), xs, acc, j, q);
      }
      ln_normalize_T(acc, rstd);
      float p = 0.f;
#pragma unroll
      for (int m = 0; m < 4; ++m)
        p += (w3[m][0] * acc[m][0] + w3[m][1] * acc[m][1]) + (w3[m][2] * acc[m][2] + w3[m][3] * acc[m][3]);
      p = row_allsum(p) + b3;
      if (ok && q == 0) {
        A.r.value_preds[(size_t)t * N + n] = p;
        if (t == T && A.next_value != nullptr) A.next_value[n] = p;
      }
    }
  } else if (wave == 15) {
    // ================================================================ generator ===============================================
    for (int t = 0; t < T; ++t) {
      float* gr = smem + L.gen + ((t & 3) * TILE_B + j) * 8;
      if (ENV == ORL_ENV_SYNTH) {
        // reward t and observation t + 1 (record t is read by E in its iterations t - 1 and t)
        ro2_wait(ctr + R3_STAGE, t - 3, err);
        const uint64_t tg = tg0 + (uint64_t)t;
        float o[4] = {0.f, 0.f, 0.f, 0.f};
        if (q == 0) synth_obs_block(A.r.env_seed, (uint32_t)n, tg + 1, 0u, o);
        const float rew = synth_reward(A.r.env_seed, (uint32_t)n, tg);
        if (q == 0) {
          *(f32x4*)gr = ok ? f32x4{D > 0 ? o[0] : 0.f, D > 1 ? o[1] : 0.f, D > 2 ? o[2] : 0.f, D > 3 ? o[3] : 0.f}
                           : f32x4{0.f, 0.f, 0.f, 0.f};
          gr[4] = ok ? rew : 0.f;
        }
      } else {
        // the first observations of episodes ep + 1 and ep + 2, ep = the count E posted after its iteration t - 1
        ro2_wait(ctr + R3_EP, t + 1, err);
        const float ep = ro2_ldf(smem + L.ep + j);
        float r1[4], r2[4];
        cartpole_reset(A.r.env_seed, (uint32_t)n, (uint32_t)(ep + 1.f), r1);
        cartpole_reset(A.r.env_seed, (uint32_t)n, (uint32_t)(ep + 2.f), r2);
        if (q == 0) {
          *(f32x4*)gr = f32x4{r1[0], r1[1], r1[2], r1[3]};
          *(f32x4*)(gr + 4) = f32x4{r2[0], r2[1], r2[2], r2[3]};
        }
      }
      ro2_post(ctr + R3_GEN, t + 1);
    }
  }
}
