// orl_rnn_rollout_coop.h - round 5: the fused recurrent MPE rollout with every 16-row tile split over FOUR cooperating waves
// (included by orl_rnn_rollout.hip, inside namespace orl, after the one-wave-per-tile bodies it replaces as the default).
//
// The one-wave bodies run ~470 fp32 MFMAs (32 cycles each, VALU-blocking) per step and tile on ONE wave while three SIMDs of the
// CU idle: 15 us per step, 375 - 395 us per 25-step rollout.  Here wave c of a tile computes the 16 output features 16c .. 16c+15
// of every layer (5 + 16 + 96 = 117 MFMAs; the gates of its own 16 features), and the three 64-wide vectors a step needs as the
// next layer's B operand (fc1 output, fc2 output, h') are exchanged through a double-buffered 4 KB LDS tile with ONE workgroup
// barrier each.  LayerNorm statistics, the head and the sampler are computed redundantly by all four waves from the exchanged
// vector (they are short; every wave needs the normalised vector anyway).
//
// Arithmetic: per output element exactly the chain of rnn_tower_fwd_lds (bias first, k-steps ascending, x-part before h-part of
// the r / z gates; LayerNorm on the full vector by the same function), so a cooperative rollout reproduces the stepwise one as the
// one-wave rollout did (tests/test_mpe_gpu.py::test_fused_recurrent_rollout_equals_stepwise_rollout).
//
// Reference: openrl/drivers/onpolicy_driver.py:154-233 (actor_rollout), openrl/modules/networks/utils/rnn.py:39-99 (GRU + mask),
// openrl/modules/networks/utils/mlp.py:8-46 (base), openrl/envs/mpe/core.py:216-323 (world step).
#pragma once

constexpr int XCH = TILE_B * HID;  // floats of one exchange buffer: [16 rows][16 groups of 4 features], groups rotated by the row

// this wave's 4 features (group 4c + q) of row j -> the tile's exchange buffer; ds_write_b128, the 16 rows of a quad on 16 slots
__device__ __forceinline__ void xch_put(float* __restrict__ buf, const f32x4 v, int c, int j, int q) {
  *(f32x4*)(buf + j * HID + 4 * ((4 * c + q + j) & 15)) = v;
}
// the full 64-wide vector back in T layout (x[m] = features 16m + 4q .. + 3 of row j)
__device__ __forceinline__ void xch_get(const float* __restrict__ buf, f32x4 (&x)[4], int j, int q) {
#pragma unroll
  for (int m = 0; m < 4; ++m) x[m] = *(const f32x4*)(buf + j * HID + 4 * ((4 * m + q + j) & 15));
}
__device__ __forceinline__ f32x4 pick4(const f32x4 (&x)[4], int c) {  // x[c] for a wave-uniform c without a register index
  return c == 0 ? x[0] : c == 1 ? x[1] : c == 2 ? x[2] : x[3];
}

// acc += W[rows 16c + j of a 64 x 64 matrix, row stride S] * in : 16 MFMAs in the k order of mm64_T / mm64_S
template <int S>
__device__ __forceinline__ void mm64_blk(const float* __restrict__ Wc, const f32x4 (&in)[4], f32x4& acc, int j, int q) {
  f32x4 a[4];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) a[mi] = *(const f32x4*)(Wc + j * S + 16 * mi + 4 * q);
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc = ORL_MFMA(a[mi][r], in[mi][r], acc);
}

// three independent chains (the r, z and n / gh_n blocks of one GRU operand) interleaved so that no MFMA waits for its own
// predecessor: acc_g += W[g][rows 16c + j] * in for g = 0, 1, 2 (blocks HID * S apart)
template <int S>
__device__ __forceinline__ void mm64_blk3(const float* __restrict__ Wc, const f32x4 (&in)[4], f32x4& a0, f32x4& a1, f32x4& a2,
                                          int j, int q) {
  f32x4 w[2][3];
#pragma unroll
  for (int g = 0; g < 3; ++g) w[0][g] = *(const f32x4*)(Wc + g * HID * S + j * S + 4 * q);
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    if (mi < 3) {
#pragma unroll
      for (int g = 0; g < 3; ++g) w[(mi + 1) & 1][g] = *(const f32x4*)(Wc + g * HID * S + j * S + 16 * (mi + 1) + 4 * q);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      a0 = ORL_MFMA(w[mi & 1][0][r], in[mi][r], a0);
      a1 = ORL_MFMA(w[mi & 1][1][r], in[mi][r], a1);
      a2 = ORL_MFMA(w[mi & 1][2][r], in[mi][r], a2);
    }
  }
}

#ifndef ORL_COOP_TILE_SYNC
#define ORL_COOP_TILE_SYNC 1  // 1: the four waves of a tile meet at an LDS counter (the tiles of a workgroup drift apart: one tile's
#endif                        // LayerNorm / gate VALU under another's MFMAs); 0: every exchange is a workgroup barrier
#ifndef ORL_COOP_PRIO
#define ORL_COOP_PRIO 0  // tile-sync only: the tiles of a workgroup at different wave priorities, so that they do NOT share the MFMA
#endif                   // pipe evenly and arrive at their VALU phases together (measured: no effect - fp32 MFMA and VALU share a datapath)
// Exchange k of a tile is complete when its counter reaches 4k: a wave adds 1 AFTER its ds_write (LDS instructions of one wave
// execute in order, and the release orders the compiler), then polls.  cnt lives in LDS, zeroed before the first step.
__device__ __forceinline__ void tile_sync(unsigned* __restrict__ cnt, unsigned& seq, int* err) {
#if ORL_COOP_TILE_SYNC
  seq += 4;
  if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  // bounded like the chase wait: a wave that never arrives (a lost workgroup-mate) must not hang the GPU; ~0.5 s of polls, far
  // beyond any exchange (< 10 us), after which the rollout's results are void (on the MPE env that includes critic_obs[1..T],
  // which the critic workgroups store): err = the chase protocol's sticky error word (sync_flags[n_groups], polled by the
  // host's DeviceErrorWatch).  Without sync_flags (the two-launch mode) there is no word to raise: the wave TRAPS - the launch
  // fails and the next HIP call on the stream reports it (a timeout must never let the rollout continue silently)
  for (unsigned spins = 0; __hip_atomic_load(cnt, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < seq; ++spins) {
    __builtin_amdgcn_s_sleep(1);
    if (spins > (1u << 22)) {
      if (err != nullptr) *err = 1;
      else __builtin_trap();
      break;
    }
  }
#else
  __syncthreads();
#endif
}

template <int S>
__device__ __forceinline__ void mm64_blk2(const float* __restrict__ Wc, const f32x4 (&in)[4], f32x4& a0, f32x4& a1, int j, int q) {
  f32x4 w[2][2];
#pragma unroll
  for (int g = 0; g < 2; ++g) w[0][g] = *(const f32x4*)(Wc + g * HID * S + j * S + 4 * q);
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    if (mi < 3) {
#pragma unroll
      for (int g = 0; g < 2; ++g) w[(mi + 1) & 1][g] = *(const f32x4*)(Wc + g * HID * S + j * S + 16 * (mi + 1) + 4 * q);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      a0 = ORL_MFMA(w[mi & 1][0][r], in[mi][r], a0);
      a1 = ORL_MFMA(w[mi & 1][1][r], in[mi][r], a1);
    }
  }
}

// base -> GRU -> LayerNorm of one 16-row tile by four waves (this one = block c).  wa(s) = this lane's fc1 A operand
// W1[16c + j][4s + q], xb(s) = obs column 4s + q of row j (both 0 beyond D), KS = fc1's k-steps.  xbuf = the tile's two exchange
// buffers, par = which one the next exchange writes (toggled here: exchange k writes buffer k & 1 after barrier k - 1, which
// every wave reaches only after its reads of exchange k - 2).  EVERY wave of the workgroup must call this (3 barriers).
// gh_n block of this wave for the hidden state hin: b_hn + W_hn[rows 16c + j] * hin - its OWN accumulator in gru_fwd_T, so it
// can be computed apart from (and before) the rest of a step without changing any sum
__device__ __forceinline__ f32x4 gru_ghn_blk(const float* __restrict__ lw, const RnnLds& tw, const f32x4 (&hin)[4], int c,
                                             int j, int q) {
  f32x4 ag = *(const f32x4*)(lw + tw.bhh + 2 * HID + 16 * c + 4 * q);
  mm64_blk<W2S>(lw + tw.Whh + 2 * HID * W2S + 16 * c * W2S, hin, ag, j, q);
  return ag;
}

// PRE: the caller supplies the gh_n block (gru_ghn_blk of the same hin) - the policy computes it in the idle windows of the
// previous step
template <int KS, bool PRE, class WA, class XB>
__device__ __forceinline__ void rnn_tower_fwd_coop(const float* __restrict__ lw, const RnnLds& tw, WA wa, XB xb,
                                                   const f32x4 (&hin)[4], const f32x4 ghn_pre, f32x4 (&hnew)[4],
                                                   f32x4 (&n3)[4],
                                                   float* __restrict__ xbuf, int& par, unsigned* __restrict__ cnt,
                                                   unsigned& seq, int* err, int c, int j, int q) {
  const int co = 16 * c + 4 * q;
  f32x4 z[4], n1[4];
  float rstd;
  {
    f32x4 acc = *(const f32x4*)(lw + tw.b1 + co);
#pragma unroll
    for (int s = 0; s < KS; ++s) acc = ORL_MFMA(wa(s), xb(s), acc);
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = fmaxf(acc[r], 0.f);
    xch_put(xbuf + par * XCH, acc, c, j, q);
    tile_sync(cnt, seq, err);
    xch_get(xbuf + par * XCH, z, j, q);
    par ^= 1;
  }
  ln_normalize_T(z, rstd);
  ln_affine_T(z, lw + tw.g1, lw + tw.be1, q, n1);
  {
    f32x4 acc = *(const f32x4*)(lw + tw.b2 + co);
    mm64_blk<W2S>(lw + tw.W2 + 16 * c * W2S, n1, acc, j, q);
    xch_put(xbuf + par * XCH, acc, c, j, q);
    tile_sync(cnt, seq, err);
    xch_get(xbuf + par * XCH, z, j, q);
    par ^= 1;
  }
  ln_normalize_T(z, rstd);
  ln_affine_T(z, lw + tw.g2, lw + tw.be2, q, n1);  // n1 = the GRU's input from here on
  {
    f32x4 ar = *(const f32x4*)(lw + tw.bih + co) + *(const f32x4*)(lw + tw.bhh + co);
    f32x4 az = *(const f32x4*)(lw + tw.bih + HID + co) + *(const f32x4*)(lw + tw.bhh + HID + co);
    f32x4 an = *(const f32x4*)(lw + tw.bih + 2 * HID + co);
    f32x4 ag = PRE ? ghn_pre : *(const f32x4*)(lw + tw.bhh + 2 * HID + co);
    mm64_blk3<W2S>(lw + tw.Wih + 16 * c * W2S, n1, ar, az, an, j, q);
    if (PRE) mm64_blk2<W2S>(lw + tw.Whh + 16 * c * W2S, hin, ar, az, j, q);
    else mm64_blk3<W2S>(lw + tw.Whh + 16 * c * W2S, hin, ar, az, ag, j, q);
    const f32x4 hc = pick4(hin, c);
    f32x4 hb;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float rr = sigmoid_f(ar[k]);
      const float zz = sigmoid_f(az[k]);
      const float nn = tanh_f(an[k] + rr * ag[k]);
      hb[k] = (1.0f - zz) * nn + zz * hc[k];
    }
    xch_put(xbuf + par * XCH, hb, c, j, q);
    tile_sync(cnt, seq, err);
    xch_get(xbuf + par * XCH, hnew, j, q);
    par ^= 1;
  }
#pragma unroll
  for (int m = 0; m < 4; ++m) z[m] = hnew[m];
  ln_normalize_T(z, rstd);
  ln_affine_T(z, lw + tw.g3, lw + tw.be3, q, n3);
}

#ifndef ORL_COOP_GHN_EARLY
#define ORL_COOP_GHN_EARLY 1  // policy: the next step's gh_n GEMM block in the idle windows of this step (head / env phases)
#endif
#ifndef ORL_COOP_CRITIC_SHARE
#define ORL_COOP_CRITIC_SHARE 1  // the critic workgroups build share_obs (= the world's three observations, contiguous in policy_obs)
#endif                           // and store it; the policy only stores its own observations
#ifndef ORL_COOP_LATE_STORES
#define ORL_COOP_LATE_STORES 1  // chase: hidden-state / reward / constant-mask stores behind the publication instead of in front of its vmcnt(0)
#endif
#ifndef ORL_COOP_DBG
#define ORL_COOP_DBG 0  // timing builds only (results are wrong): 1 no env step, 2 no tower, 4 no observation copies, 8 no head /
#endif                  // sampler, 16 no critic
constexpr int COOP_THREADS = 768;  // 3 tiles x 4 waves
constexpr int MPE_KS_P = (MPE_OBS + 3) / 4, MPE_KS_C = (MPE_COBS + 3) / 4;
constexpr int MPE_NACT = 5;  // Discrete(5): no-op + 4 moves (envs/mpe/environment.py:85-93)

// Policy workgroup: 16 worlds; waves 4a .. 4a + 3 = agent a of those worlds.  Everything outside the tower is the one-wave body's
// (same LDS tiles, same stores, same publication protocol) on four times the threads.
template <int NO, bool CHASE>
__device__ __forceinline__ void rnn_rollout_policy_body_coop(const RnnRolloutArgs& A, const int bid) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const orl_buffer_ptrs& b = A.buf;
  const RnnLayout tl(A.pnet);
  const RnnLds tw(MPE_OBS, A.pnet.n_out, false);
  stage_rnn_tower(smem, A.ptheta, tl, tw, threadIdx.x, blockDim.x);
  const float* lw = smem;
  float* s_obs = smem + tw.total;                    // [3 agents][16 worlds][OBS_S]
  float* s_act = s_obs + MPE_A * TILE_B * OBS_S;     // [3][16]
  float* s_rew = s_act + MPE_A * TILE_B;             // [16]
  float* s_done = s_rew + TILE_B;                    // [16]
  float* s_x = s_done + TILE_B;                      // [3 agents][2][XCH] exchange buffers
  unsigned* s_cnt = (unsigned*)(s_x + MPE_A * 2 * XCH);  // [3] exchange counters (+ 1 pad)
  if (threadIdx.x < 4) s_cnt[threadIdx.x] = 0u;
  const int wv = threadIdx.x >> 6, ag = wv >> 2, c = wv & 3, l = threadIdx.x & 63, j = l & 15, q = l >> 4;
  const int N = b.N, T = b.T;
  const int LA = N * MPE_A;
  const int e0 = bid * TILE_B;
  const int env = e0 + j;
  const bool ok = env < N;
  const int row = (ok ? env : 0) * MPE_A + ag;  // this lane's buffer row (world, agent)
  const int n_here = (N - e0) < TILE_B ? (N - e0) : TILE_B;
  const uint64_t rng0 = A.rng_step0 + (A.rng_dev ? *A.rng_dev : 0ull);

  for (int e = threadIdx.x; e < MPE_A * TILE_B * OBS_S; e += blockDim.x) {
    const int i = e / (TILE_B * OBS_S), r = e - i * (TILE_B * OBS_S), jj = r / OBS_S, k = r - jj * OBS_S;
    s_obs[e] = (jj < n_here && k < MPE_OBS) ? b.policy_obs[((size_t)(e0 + jj) * MPE_A + i) * MPE_OBS + k] : 0.f;
  }
  f32x4 h[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) h[m] = *(const f32x4*)(A.hp + (size_t)row * HID + 16 * m + 4 * q);
  float mk = b.masks[row];
  // the worlds live on wave 3 (agent 0, block 3): SIMD 3 carries no head wave (those are waves 0, 5, 10 on SIMDs 0, 1, 2)
  constexpr int ENV_WAVE = 3;
  const bool world_lane = wv == ENV_WAVE && q == 0 && ok;
  MpeWorld w;
  float est[4] = {0.f, 0.f, 0.f, 0.f};
  if (world_lane) {
    mpe_load(A.env_state + (size_t)env * MPE_STATE_W, w);
    if (A.ep_stats != nullptr) {
#pragma unroll
      for (int k = 0; k < 4; ++k) est[k] = A.ep_stats[(size_t)env * 4 + k];
    }
  }
  __syncthreads();
  float* xbuf = s_x + ag * 2 * XCH;
  int par = 0;
  unsigned seq = 0u;
  int* err = A.flags != nullptr ? A.flags + (N + TILE_B - 1) / TILE_B : nullptr;
#if ORL_COOP_TILE_SYNC && ORL_COOP_PRIO
  if (ag == 0) __builtin_amdgcn_s_setprio(2);
  else if (ag == 1) __builtin_amdgcn_s_setprio(1);
#endif
  const float* w1row = lw + tw.W1 + (16 * c + j) * tw.DP + q;

#if ORL_COOP_GHN_EARLY
  f32x4 ghn;
  {
    f32x4 hin0[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) hin0[m] = h[m] * mk;
    ghn = gru_ghn_blk(lw, tw, hin0, c, j, q);
  }
#else
  const f32x4 ghn = {0.f, 0.f, 0.f, 0.f};
#endif
  for (int t = 0; t < T; ++t) {
    f32x4 hin[4], hnew[4], n3[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) hin[m] = h[m] * mk;
    const float* xrow = s_obs + (ag * TILE_B + j) * OBS_S + q;
#if ORL_COOP_DBG & 2
#pragma unroll
    for (int m = 0; m < 4; ++m) hnew[m] = hin[m] + xrow[0], n3[m] = hin[m];
#else
    rnn_tower_fwd_coop<MPE_KS_P, ORL_COOP_GHN_EARLY != 0>(
        lw, tw, [&](int s) -> float { return w1row[4 * s]; }, [&](int s) -> float { return xrow[4 * s]; }, hin, ghn, hnew,
        n3, xbuf, par, s_cnt + ag, seq, err, c, j, q);
#endif
#if ORL_COOP_GHN_EARLY
    // the NEXT step's gh_n block on the unmasked h' (rows whose world ends get the bias below: W * 0 adds exact zeros), in this
    // step's idle windows: now for the waves without a head to compute, behind the actions barrier for the three head waves
    if (c != ag) ghn = gru_ghn_blk(lw, tw, hnew, c, j, q);
#endif
    float hd[NO], act_o[NO], lp_o[NO];
#if ORL_COOP_DBG & 8
    act_o[0] = (float)((int)(n3[0][0] * 3.f) & 3), lp_o[0] = n3[1][0];
#else
    // simple_spread's Discrete(5) as a LITERAL (the launcher requires it): with the run-time n_out every class of head_T and of
    // the sampler sat behind its own scalar branch and the logits went through scratch - 2.4 us of the step
    // the pieces of sample_head<NO, CATEGORICAL> called directly: its pick<NO>(hd, a) is compiled to an INDEXED SCRATCH LOAD
    // (the select chain is recognised as a table lookup) followed by s_waitcnt vmcnt(0) on the step's serial chain
    // ... by ONE of the tile's four waves, wave c == agent: the three agents' heads then run on three different SIMDs instead of
    // twelve waves doing the same ~350 VALU side by side, three to a SIMD
    if (c == ag) {
      head_T<NO>(lw + tw.W3, lw + tw.b3, MPE_NACT, n3, q, hd);
      const float lse = cat_lse<NO>(hd, MPE_NACT, nullptr);
      int a;
      if (A.deterministic) a = cat_mode<NO>(hd, MPE_NACT);
      else {
        const uint64_t rs = rng0 + (uint64_t)t;
        const u4 r = philox4x32_10(A.act_seed, (uint32_t)row, 0u, (uint32_t)rs, (uint32_t)(rs >> 32) << 8);
        a = cat_sample<NO>(hd, MPE_NACT, lse, u01(r.x));
      }
      float pk = hd[0];
#pragma unroll
      for (int k = 1; k < MPE_NACT; ++k) {
        float x = hd[k];
        asm volatile("" : "+v"(x));  // keeps the selects selects
        pk = a == k ? x : pk;
      }
      act_o[0] = (float)a;
      lp_o[0] = pk - lse;
    }
#endif
    if (q == 0 && c == ag) {
      s_act[ag * TILE_B + j] = act_o[0];
      if (ok) {
        A.actions[(size_t)t * LA + row] = act_o[0];
        A.logp[(size_t)t * LA + row] = lp_o[0];
      }
    }
    __syncthreads();  // actions of the 3 agents visible; every wave is done reading this step's observations
#if ORL_COOP_GHN_EARLY
    if (c == ag) ghn = gru_ghn_blk(lw, tw, hnew, c, j, q);
#endif
    if (world_lane) {
      int act[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) act[i] = (int)s_act[i * TILE_B + j];
      float rew;
      bool done;
#if ORL_COOP_DBG & 1
      rew = (float)act[0], done = act[1] + act[2] > 100;
      s_rew[j] = rew;
      s_done[j] = done ? 1.f : 0.f;
#else
      mpe_advance(w, act, A.env_seed, (uint32_t)env, A.world_length, rew, done);
      mpe_ep_stats(est, rew, done);
      s_rew[j] = rew;
      s_done[j] = done ? 1.f : 0.f;
      float o[3][MPE_OBS];
      mpe_obs(w, o);
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int k = 0; k < MPE_OBS; ++k) s_obs[(i * TILE_B + j) * OBS_S + k] = o[i][k];
#endif
    }
    __syncthreads();  // next observations, rewards and done flags visible
    mk = s_done[j] != 0.f ? 0.f : 1.f;
#if ORL_COOP_GHN_EARLY
    if (mk == 0.f) ghn = *(const f32x4*)(lw + tw.bhh + 2 * HID + 16 * c + 4 * q);
#endif
#pragma unroll
    for (int m = 0; m < 4; ++m) h[m] = hnew[m] * mk;
    // What the critic workgroup reads (masks, the observations below) is stored FIRST and alone in front of the publication's
    // vmcnt(0); the stores nobody waits for inside this launch (hidden states, rewards, the constant masks) follow behind it.
    auto late_stores = [&]() {
      if (ok) *(f32x4*)(A.hp + ((size_t)(t + 1) * LA + row) * HID + 16 * c + 4 * q) = pick4(h, c);  // this wave's quarter
      if ((int)threadIdx.x < MPE_A * n_here) {
        const int jj = threadIdx.x / MPE_A;
        const size_t r1 = (size_t)(t + 1) * LA + (size_t)e0 * MPE_A + threadIdx.x;
        b.rewards[(size_t)t * LA + (size_t)e0 * MPE_A + threadIdx.x] = s_rew[jj];
        b.active_masks[r1] = 1.f;
        b.bad_masks[r1] = 1.f;
      }
    };
    if constexpr (CHASE && ORL_COOP_LATE_STORES) {
      if ((int)threadIdx.x < MPE_A * n_here)
        st_agent(b.masks + (size_t)(t + 1) * LA + (size_t)e0 * MPE_A + threadIdx.x, s_done[threadIdx.x / MPE_A] != 0.f ? 0.f : 1.f);
    } else {
      if (ok) *(f32x4*)(A.hp + ((size_t)(t + 1) * LA + row) * HID + 16 * c + 4 * q) = pick4(h, c);
      if ((int)threadIdx.x < MPE_A * n_here) {
        const int jj = threadIdx.x / MPE_A;
        const size_t r1 = (size_t)(t + 1) * LA + (size_t)e0 * MPE_A + threadIdx.x;
        b.rewards[(size_t)t * LA + (size_t)e0 * MPE_A + threadIdx.x] = s_rew[jj];
        if (CHASE) st_agent(b.masks + r1, s_done[jj] != 0.f ? 0.f : 1.f);
        else b.masks[r1] = s_done[jj] != 0.f ? 0.f : 1.f;
        b.active_masks[r1] = 1.f;
        b.bad_masks[r1] = 1.f;
      }
    }
    if (!(ORL_COOP_DBG & 4)) {
      float* dp = b.policy_obs + ((size_t)(t + 1) * LA + (size_t)e0 * MPE_A) * MPE_OBS;
      for (int e = threadIdx.x; e < n_here * MPE_A * MPE_OBS; e += blockDim.x) {
        const int jj = e / (MPE_A * MPE_OBS), r = e - jj * (MPE_A * MPE_OBS), i = r / MPE_OBS, k = r - i * MPE_OBS;
        const float v = s_obs[(i * TILE_B + jj) * OBS_S + k];
#if ORL_COOP_CRITIC_SHARE
        if (CHASE) st_agent(dp + e, v);  // read by the critic workgroup of this group
        else dp[e] = v;
#else
        dp[e] = v;
#endif
        if (t == T - 1 && A.obs_p_out != nullptr) A.obs_p_out[(size_t)e0 * MPE_A * MPE_OBS + e] = v;
      }
      float* dc = b.critic_obs + ((size_t)(t + 1) * LA + (size_t)e0 * MPE_A) * MPE_COBS;
#if ORL_COOP_CRITIC_SHARE
      if (t == T - 1 && A.obs_c_out != nullptr)  // only the env's final share_obs is still written here
        for (int e = threadIdx.x; e < n_here * MPE_A * MPE_COBS; e += blockDim.x) {
          const int jj = e / (MPE_A * MPE_COBS), r = e - jj * (MPE_A * MPE_COBS), cc = r % MPE_COBS;
          const int i2 = cc / MPE_OBS, k = cc - i2 * MPE_OBS;
          A.obs_c_out[(size_t)e0 * MPE_A * MPE_COBS + e] = s_obs[(i2 * TILE_B + jj) * OBS_S + k];
        }
      (void)dc;
#else
      for (int e = threadIdx.x; e < n_here * MPE_A * MPE_COBS; e += blockDim.x) {
        const int jj = e / (MPE_A * MPE_COBS), r = e - jj * (MPE_A * MPE_COBS), cc = r % MPE_COBS;
        const int i2 = cc / MPE_OBS, k = cc - i2 * MPE_OBS;
        const float v = s_obs[(i2 * TILE_B + jj) * OBS_S + k];
        if (CHASE) st_agent(dc + e, v);
        else dc[e] = v;
        if (t == T - 1 && A.obs_c_out != nullptr) A.obs_c_out[(size_t)e0 * MPE_A * MPE_COBS + e] = v;
      }
#endif
    }
    if (CHASE) {  // slot t+1 (share_obs, masks) is complete: publish it to the critic workgroup of this group
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (threadIdx.x == 0) __hip_atomic_store(A.flags + bid, t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if constexpr (ORL_COOP_LATE_STORES != 0) late_stores();
    }
  }
  if (world_lane) {
    mpe_store(A.env_state + (size_t)env * MPE_STATE_W, w);
    if (A.ep_stats != nullptr) {
#pragma unroll
      for (int k = 0; k < 4; ++k) A.ep_stats[(size_t)env * 4 + k] = est[k];
    }
  }
}

// Critic workgroup: three 16-row tiles (tiles 3 * bid' .. + 2 per trip) of the stored share_obs, four waves per tile; W1 (the
// only matrix whose size depends on the 54-wide observation) as 14 registers per lane instead of 14 KB of LDS, the observation
// row as 14 more - no slab.  Every wave of the workgroup runs every trip (tiles beyond the end compute on row 0 and store nothing):
// the exchanges are workgroup barriers.
template <bool CHASE>
__device__ __forceinline__ void rnn_rollout_critic_body_coop(const RnnRolloutArgs& A, const int bid, const int nblk) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const orl_buffer_ptrs& b = A.buf;
  const RnnLayout tl(A.cnet);
  const RnnLds tw(MPE_COBS, 1, false, false, false, true);
  stage_rnn_tower(smem, A.ctheta, tl, tw, threadIdx.x, blockDim.x, false, false, true);
  const float* lw = smem;
  const int wv = threadIdx.x >> 6, ti = wv >> 2, c = wv & 3, l = threadIdx.x & 63, j = l & 15, q = l >> 4;
  constexpr int KS = MPE_KS_C, D = MPE_COBS;
  float w1r[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) w1r[s] = (4 * s + q < D) ? A.ctheta[tl.oW1 + (16 * c + j) * D + 4 * s + q] : 0.f;
  if (threadIdx.x < 4) ((unsigned*)(smem + tw.total + 3 * 2 * XCH))[threadIdx.x] = 0u;
  __syncthreads();
  const int T = b.T;
  const int LA = b.N * b.A;
  const int n_tiles = (LA + TILE_B - 1) / TILE_B;
  float* xbuf = smem + tw.total + ti * 2 * XCH;
  unsigned* s_cnt = (unsigned*)(smem + tw.total + 3 * 2 * XCH);
  int par = 0;
  unsigned seq = 0u;
  int* err = A.flags != nullptr ? A.flags + (b.N + TILE_B - 1) / TILE_B : nullptr;
#if ORL_COOP_TILE_SYNC && ORL_COOP_PRIO
  if (ti == 0) __builtin_amdgcn_s_setprio(2);
  else if (ti == 1) __builtin_amdgcn_s_setprio(1);
#endif
  for (int tile0 = bid * 3; tile0 < n_tiles; tile0 += nblk * 3) {
    const int row = (tile0 + ti) * TILE_B + j;
    const bool ok = row < LA;
    const int rr = ok ? row : 0;
    f32x4 h[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) h[m] = *(const f32x4*)(A.hc + (size_t)rr * HID + 16 * m + 4 * q);
    float xv[KS], mk = b.masks[rr];
    auto load_x = [&](int t, float (&x)[KS]) {
#if ORL_COOP_CRITIC_SHARE
      // slot 0 is the caller's; later slots are read where the policy workgroup wrote them: the three observations of a world
      // are contiguous in policy_obs ([world][agent][18]) = the share_obs row of each of its agents
      const float* xr = t == 0 ? b.critic_obs + (size_t)rr * D + q
                               : b.policy_obs + ((size_t)t * LA + (size_t)(rr / MPE_A) * MPE_A) * MPE_OBS + q;
#else
      const float* xr = b.critic_obs + ((size_t)t * LA + rr) * D + q;
#endif
#pragma unroll
      for (int s = 0; s < KS; ++s) x[s] = (4 * s + q < D) ? (CHASE && t > 0 ? ld_agent(xr + 4 * s) : xr[4 * s]) : 0.f;
    };
    auto wait_for = [&](int need) {
      if (__hip_atomic_load(A.flags + bid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= need) return;
      const unsigned long long t0 = wall_clock64();
      while (__hip_atomic_load(A.flags + bid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
        __builtin_amdgcn_s_sleep(8);
        if (wall_clock64() - t0 > CHASE_TIMEOUT_TICKS) {
          A.flags[(b.N + TILE_B - 1) / TILE_B] = 1;  // error word: the rollout's critic outputs are not valid
          break;
        }
      }
    };
    load_x(0, xv);
    for (int t = 0; t <= T; ++t) {
      float xn[KS], mkn = 0.f;
      if (!CHASE && t < T) {  // next slot's inputs in flight behind this step's GEMMs
        load_x(t + 1, xn);
        mkn = b.masks[(size_t)(t + 1) * LA + rr];
      }
      f32x4 hin[4], hnew[4], n3[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) hin[m] = h[m] * mk;
      rnn_tower_fwd_coop<KS, false>(
          lw, tw, [&](int s) -> float { return w1r[s]; }, [&](int s) -> float { return xv[s]; }, hin,
          f32x4{0.f, 0.f, 0.f, 0.f}, hnew, n3, xbuf, par, s_cnt + ti, seq, err, c, j, q);
      float v[1];
      head_T<1>(lw + tw.W3, lw + tw.b3, 1, n3, q, v);
      if (t == T) {
        if (ok && q == 0 && c == 0 && A.next_value != nullptr) A.next_value[row] = v[0];
        break;
      }
      if (ok && q == 0 && c == 0) A.value_preds[(size_t)t * LA + row] = v[0];
      if (CHASE) {
        wait_for(t + 1);
        load_x(t + 1, xn);
        mkn = ld_agent(b.masks + (size_t)(t + 1) * LA + rr);
      }
#if ORL_COOP_CRITIC_SHARE
      if (ok) {  // share_obs of slot t + 1 for the update: k-steps s = c, c + 4, .. from this wave
        float* dc = b.critic_obs + ((size_t)(t + 1) * LA + row) * D + q;
#pragma unroll
        for (int s = 0; s < KS; ++s)
          if ((s & 3) == c && 4 * s + q < D) dc[4 * s] = xn[s];
      }
#endif
#pragma unroll
      for (int m = 0; m < 4; ++m) h[m] = hnew[m] * mkn;  // rnn_states_critic[dones_env] = 0
      if (ok) *(f32x4*)(A.hc + ((size_t)(t + 1) * LA + row) * HID + 16 * c + 4 * q) = pick4(h, c);
      mk = mkn;
#pragma unroll
      for (int s = 0; s < KS; ++s) xv[s] = xn[s];
    }
  }
}

template <int NO>
__global__ __launch_bounds__(COOP_THREADS) void rnn_rollout_mpe_policy_coop_kernel(RnnRolloutArgs A) {
  rnn_rollout_policy_body_coop<NO, false>(A, (int)blockIdx.x);
}
__global__ __launch_bounds__(COOP_THREADS) void rnn_rollout_critic_coop_kernel(RnnRolloutArgs A) {
  rnn_rollout_critic_body_coop<false>(A, (int)blockIdx.x, (int)gridDim.x);
}
// policy and critic workgroups interleaved, the critic one step behind its policy group
template <int NO>
__global__ __launch_bounds__(COOP_THREADS) void rnn_rollout_mpe_chase_coop_kernel(RnnRolloutArgs A) {
  const int bid = (int)blockIdx.x >> 1;
  if ((blockIdx.x & 1) == 0) rnn_rollout_policy_body_coop<NO, true>(A, bid);
  else if (!(ORL_COOP_DBG & 16)) rnn_rollout_critic_body_coop<true>(A, bid, (int)gridDim.x >> 1);
}
