// orl_rollout2.h - round 6: the fused rollout of the single-agent device envs (synthetic fixed-step env, CartPole-v1) as ONE
// dependent chain per 16-env tile with everything else taken off it (included by orl_act.hip, inside namespace orl).
//
// What the round-5 kernel (rollout_kernel, still built: tic-tac-toe runs on it, args.opp_reserved = 1 selects it for A/B) did
// per step: policy AND critic tower in lock step on the same four SIMDs (an fp32 MFMA occupies the SIMD's VALU datapath, so the
// critic's 20 MFMAs per wave and step are 640 cycles the policy wave cannot issue in), three workgroup barriers, the full
// 64-wide LayerNorm 2 + head on one wave after an all-gather, the Philox draws / env trigonometry / ten global stores of a step
// in the waves of the chain: 5 270 cycles per step (profiles/r05_rollout_phase_prof.txt), 250 us per 128 steps.
//
// A wave issues at most one instruction every 4 - 5.5 cycles whatever its kind (tools/mfma_valu_overlap.hip: 21.8 cycles per 4
// independent v_fma on one wave, the same per wave with 4 waves on the SIMD), so the length of a step is the NUMBER OF
// INSTRUCTIONS on the waves of its dependent chain plus the LDS round trips between them.  Here (reference
// loop: openrl/drivers/onpolicy_driver.py:154-203):
//   * waves 0-3 ("P") = the policy tower, 16 output features each; wave 0 also finishes the step (head, sampler, env, hand-over).
//     NOTHING else runs on their chain: no critic, no Philox, no global store, no workgroup barrier;
//   * LayerNorm 2 and the head are computed from PARTIALS: with z = fc2's output (pre-LayerNorm) the head is
//         logit_c = rstd (W3g[c] . z  -  mean sum_f W3g[c][f]) + b3'[c]        (W3g = W3 diag(g2), b3' = b3 + W3 be2: stage_tower(fold)),
//     so each P wave reduces ITS 16 features to {sum z, sum z^2, W3g[c] . z} (narrow heads: 4 floats per lane; wide heads: 4 MFMAs
//     give the 16-class partial) and wave 0 adds four partials - no all-gather of the tile, no second LayerNorm pass, no 16-MFMA
//     head chain.  Exact algebra; the cancellation in (W3g . z - mean sum W3g) is the one-pass LayerNorm's, guarded the same way:
//     a tile holding a row with mean^2 > 16 var takes the round-5 path (the z tile is always left in LDS for it);
//   * the CRITIC runs in the same launch but off the chain and off wave 0's SIMD: waves 9-11 take the observation slots t = c, c + 3,
//     ... (all T + 1 of them, slot T = the bootstrap value) out of the LDS ring as they are published and evaluate the value
//     tower the way the update does - fc2 as two-term fp16 splits on the 16-bit MFMA (orl_mlp.h, ORL_TOWER_F16: 3 products over the scaled
//     image, fp32 accuracy; rounds 3 - 5: three-term bf16 splits; a 16-bit MFMA does not occupy the VALU datapath), LayerNorm affines folded - at priority 0, three steps of time per
//     tile.  Values are not needed to act;
//   * waves 5-7 ("services") run AHEAD of or BEHIND the chain through small LDS rings: wave 5 draws the sampling noise two steps
//     ahead, wave 6 the action-independent half of the env step (synthetic: reward + next observation; CartPole: cos / sin of
//     the pole angle and the reset state of the next episode - cartpole_pre / cartpole_reset), wave 7 writes the step's rows of
//     the rollout buffer (coalesced: the 16 rows of a field are contiguous) from an LDS staging ring.  Waves 4 and 8 exit: they
//     would share wave 0's SIMD;
//   * all hand-overs are single-writer LDS words (monotonic step counters) polled by their readers - LDS operations of a CU
//     execute in order, a writer stores data then counter, a reader loads counter then data.  The chain reads counters AND
//     payload in one batch and repeats the batch until the counters say the payload is valid: one LDS round trip per
//     hand-over.  Every poll is bounded; a timeout poisons the tile's rewards with NaN (it cannot happen short of a hardware
//     fault: the waves of a workgroup are co-resident by construction).
//
// Same per-element arithmetic as the round-5 kernel up to the head (fc1, guarded one-pass LayerNorm 1, fc2 in two chains of 8):
// tests/test_rollout_gpu.py::test_fused_rollout_equals_stepwise_rollout and the new-vs-old kernel test pin it.
#pragma once

constexpr int RO2_RING = 4;   // noise / env-service / staging rings (steps)
constexpr int RO2_ORING = 8;  // observation ring (slots): the critic waves may lag the chain by a few steps
constexpr int RO2_STG = 36;   // floats per staged row: actions[16] | log-probs[16] | reward | done | 2 pad
constexpr int RO2_ENVW = 8;   // floats per env-service row
constexpr int RO2_THREADS = 768;
// counter words (unsigned, 16-byte groups)
enum { RC_OBS = 0, RC_STAGE = 1, RC_NOISE = 4, RC_ENV = 5, RC_STORED = 6, RC_ERR = 7, RC_PART = 8, RC_XG = 12, RC_CRIT = 16,
       RC_WORDS = 20 };

struct Ro2Lds {
  int critic, obs, word, xg, part, z2, noise, env, stage, gtab, ctr, total;  // float offsets
};
__host__ __device__ inline Ro2Lds ro2_lds(int policy_total, int critic_total, int DP) {
  Ro2Lds L;
  int o = policy_total;
  L.critic = o; o += critic_total;
  L.obs = o; o += RO2_ORING * TILE_B * DP;
  L.word = o; o += RO2_ORING * TILE_B * 3;  // tic-tac-toe: [slot]{board word, opponent moves, episode}[row]
  L.xg = o; o += 2 * TILE_B * GS;
  L.part = o; o += 2 * 4 * (TILE_B * 16 + 4 * TILE_B * 2);  // [slot][wave]{16 x 16 logits | [q][row] 2 stats}; narrow heads use the head of it
  L.z2 = o; o += 2 * TILE_B * GS;
  L.noise = o; o += RO2_RING * TILE_B * 16;
  L.env = o; o += RO2_RING * TILE_B * RO2_ENVW;
  L.stage = o; o += RO2_RING * TILE_B * RO2_STG;
  L.gtab = o; o += 3 * 16;  // Gaussian heads: {std, 1 / (2 std^2), log std + log sqrt(2 pi)} per action dimension
  L.ctr = o; o += RC_WORDS;
  L.total = o;
  return L;
}

// LDS words and payload through explicit LDS (address space 3) pointers: through a generic `volatile unsigned*` hipcc emits FLAT
// loads / stores with system scope (flat_load_dword ... sc0 sc1 + s_waitcnt vmcnt(0)) - every poll a trip through the vector
// memory path, 600 - 1 200 cycles per hand-over in the first build's phase profile.
typedef __attribute__((address_space(3))) unsigned ro2_lds_u32;
typedef __attribute__((address_space(3))) u32x4 ro2_lds_u32x4;
typedef __attribute__((address_space(3))) f32x4 ro2_lds_f32x4;
typedef __attribute__((address_space(3))) float ro2_lds_f32;
__device__ __forceinline__ unsigned ro2_ld(const unsigned* w) { return *(volatile ro2_lds_u32*)(ro2_lds_u32*)w; }
__device__ __forceinline__ u32x4 ro2_ld4u(const unsigned* w) { return *(volatile ro2_lds_u32x4*)(ro2_lds_u32x4*)w; }
__device__ __forceinline__ f32x4 ro2_ld4f(const float* w) { return *(volatile ro2_lds_f32x4*)(ro2_lds_f32x4*)w; }
__device__ __forceinline__ float ro2_ldf(const float* w) { return *(volatile ro2_lds_f32*)(ro2_lds_f32*)w; }
__device__ __forceinline__ void ro2_fail(unsigned* err) { *(volatile ro2_lds_u32*)(ro2_lds_u32*)err = 1u; }
constexpr unsigned RO2_MAX_SPINS = 1u << 24;
template <bool SLEEP = true>
__device__ __forceinline__ void ro2_wait(const unsigned* __restrict__ w, int target, unsigned* __restrict__ err) {
  if (target <= 0) return;
  unsigned spins = 0;
#pragma unroll 1
  while ((int)ro2_ld(w) < target) {
    if (SLEEP) __builtin_amdgcn_s_sleep(1);
    if (++spins > RO2_MAX_SPINS) { ro2_fail(err); break; }
  }
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ void ro2_post(unsigned* __restrict__ w, int v) {
  asm volatile("" ::: "memory");  // the data stores above stay above (LDS executes a wave's operations in order)
  *(volatile ro2_lds_u32*)(ro2_lds_u32*)w = (unsigned)v;
  asm volatile("" ::: "memory");
}

// Discrete(2) without masks: cat_lse + cat_sample + pick of orl_mlp.h with the class loops and the `c < n_out` tests resolved -
// the same operations in the same order (bit-identical action and log-probability), a third of the instructions: the generic
// form is ~70 issue slots on the step's serial chain.
__device__ __forceinline__ void ro2_sample_cat2(float l0, float l1, float u, float& act, float& logp) {
  const float mx = fmaxf(fmaxf(-3.0e38f, l0), l1);
  float se = 0.f;
  se += __expf(l0 - mx);
  se += __expf(l1 - mx);
  const float lse = mx + __logf(se);
  const float p0 = __expf(l0 - lse), p1 = __expf(l1 - lse);
  float tot = 0.f;
  tot += p0;
  tot += p1;
  const float ut = u * tot;
  float cum = 0.f;
  cum += p0;
  int a = -1, last = 0;
  if (cum > ut) a = 0;
  cum += p1;
  if (p1 > 0.f) last = 1;
  if (a < 0 && cum > ut) a = 1;
  a = a < 0 ? last : a;
  act = (float)a;
  logp = (a == 1 ? l1 : l0) - lse;
}

#ifdef ORL_PROF
// timing build (tools/rollout2_phase_prof.py): wave 0 of workgroup 0 stamps the shader clock after each phase of a step
#define R2_T(k)                                                                          \
  do {                                                                                   \
    if (prof_on) {                                                                       \
      const unsigned long long t_now = __builtin_readcyclecounter();                     \
      if (l == 0) atomicAdd(&prof_lds[k], t_now - t_last);                               \
      t_last = t_now;                                                                    \
    }                                                                                    \
  } while (0)
#else
#define R2_T(k) do {} while (0)
#endif

// KS: fc1 k-steps known at compile time (1 = observations of <= 4 columns: configuration 2 / CartPole), 0 = run-time DP
template <int NO, int HEAD, int ENV, int KS>
__global__ __launch_bounds__(RO2_THREADS) void rollout2_kernel(RolloutArgs A) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
#ifdef ORL_PROF
  __shared__ unsigned long long prof_lds[16];
  if (threadIdx.x < 16) prof_lds[threadIdx.x] = 0ull;
  const bool prof_on = blockIdx.x == 0 && threadIdx.x < 64;
  unsigned long long t_last = 0ull;
#endif
  const orl_buffer_ptrs& b = A.r.buf;
  const int D = A.pnet.obs_dim;
  const int N = b.N, T = b.T;
  const TowerLayout tlp(A.pnet), tlc(A.cnet);
  constexpr bool HMM = NO > 4;
  // both towers' W2 diag(g1) as the scaled two-term fp16 images of orl_mlp.h (ORL_TOWER_F16; the policy's fc2 - 16 fp32 MFMAs of
  // 32 cycles on every step's chain - is 6 fp16 MFMAs + 32 VALU of splitting: 710 -> ~300 cycles of the step)
  constexpr bool PSPLIT = ORL_TOWER_F16 != 0;
  const TowerLds twp(D, A.pnet.n_out, HEAD == ORL_HEAD_GAUSSIAN, false, HMM, PSPLIT);
  const TowerLds twc(D, 1, false, false, false, true);  // the critic's W2 as split images
  stage_tower(smem, A.ptheta, tlp, twp, false, threadIdx.x, blockDim.x, HMM, PSPLIT, true);
  stage_tower(smem + twp.total, A.ctheta, tlc, twc, false, threadIdx.x, blockDim.x, false, true, true);
  const int DP = KS > 0 ? 4 * KS : twp.DP;
  const Ro2Lds L = ro2_lds(twp.total, twc.total, DP);
  float* s_obs = smem + L.obs;
  unsigned* ctr = (unsigned*)(smem + L.ctr);
  unsigned* err = ctr + RC_ERR;
  const int wave = threadIdx.x >> 6, l = threadIdx.x & 63, j = l & 15, q = l >> 4;
  const int n0 = blockIdx.x * TILE_B;
  const int n = n0 + j;
  const bool ok = n < N;
  const int nrow = (N - n0) < TILE_B ? (N - n0) : TILE_B;
  const int n_out = A.pnet.n_out;
  const int a_w = (HEAD == ORL_HEAD_CATEGORICAL) ? 1 : n_out;
  const bool sep_c = b.critic_obs != b.policy_obs;
  // Tic-tac-toe vs the random opponent (ORL_ENV_TTT, round 6): the observation of a row travels as ONE board word - agent bits
  // | opponent bits << 9 | valid << 18 (the round-5 kernel's form) - from which the trunk waves expand their fc1 operands
  // (feature d = cell d >> 1 of player d & 1), wave 0 the legal-move mask of its four classes, the store wave the 18 + 9
  // floats of the buffer rows; the env service draws the step's three Philox words (opponent reply, who opens the next game,
  // its opening move) from the row's (episode, opponent moves) while the policy works.
  constexpr bool IS_TTT = ENV == ORL_ENV_TTT;
  int* s_word = (int*)(smem + L.word);               // [slot][16]
  int* s_meta = s_word + RO2_ORING * TILE_B;        // [slot][2][16]: opponent moves this game, episode
  auto ttt_operand_bits = [&](int w) -> int { return (w >> ((q & 1) * 9 + (q >> 1))) & (q < 2 ? 0x155 : 0x55); };
  if (IS_TTT && threadIdx.x < TILE_B) {
    const int jj = threadIdx.x, nn = n0 + jj;
    int ta = 0, to = 0, mv = 0, epi = 0;
    if (nn < N) {
      const float* st = A.r.env_state + (size_t)nn * TTT_STATE_W;
      for (int c = 0; c < 9; ++c) {
        ta |= (st[c] == 1.f ? 1 : 0) << c;
        to |= (st[c] == 2.f ? 1 : 0) << c;
      }
      mv = (int)st[9]; epi = (int)st[10];
    }
    s_word[jj] = nn < N ? (ta | (to << 9) | (1 << 18)) : 0;
    s_meta[jj] = mv; s_meta[TILE_B + jj] = epi;
  }
  // slot 0 of the observation ring comes from the buffer (after_update / init_buffer put it there)
  for (int e = threadIdx.x; e < TILE_B * DP; e += blockDim.x) {
    const int jj = e / DP, k = e - jj * DP;
    s_obs[e] = (n0 + jj < N && k < D) ? b.policy_obs[(size_t)(n0 + jj) * D + k] : 0.f;
  }
  if (threadIdx.x < RC_WORDS) ctr[threadIdx.x] = threadIdx.x == RC_OBS ? 1u : 0u;
  __syncthreads();
  const uint64_t tg0 = A.r.rng_step0;

  if (wave < 4) {
    // ================================================================ the policy chain ==========================
    __builtin_amdgcn_s_setprio(3);
    const int gw = wave;
    CoopRegs creg;
    coop_load(smem, twp, gw, j, q, creg);
    // fp16 images: this wave's 16 output rows of fc2 - A fragments (K-step h, part p) of row block gw, in registers for the launch
    u32x4 wA[2][2];
    float ln2_eps = 1e-5f;
    if constexpr (PSPLIT) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) wA[h][pp] = *(const u32x4*)((const unsigned short*)(smem + twp.W2) + wb_off(pp, 16 * gw + j, h, q));
      ln2_eps = smem[twp.wsc + 3];  // 1e-5 x 4^kw: the accumulators below are 2^kw z2
    }
    // wide observations: this wave's fc1 A operands W1[16 gw + j][4 s + q] of the first 8 k-steps in registers (observations
    // of <= 32 columns: cfg3's 17, cfg5's 18); k-steps beyond come from the LDS image every step
    float w1w[KS > 0 ? 1 : 8];
    if constexpr (KS == 0) {
#pragma unroll
      for (int s = 0; s < 8; ++s)
        w1w[s] = (DP > COOP_SMALL_DP && 4 * s < DP && (!IS_TTT || s < 5)) ? smem[twp.W1 + (16 * gw + j) * DP + 4 * s + q] : 0.f;
    }
    // head operands of this wave's 16 features: narrow W3g[c][16 gw + 4 q ..], wide the MFMA fragment W3P[j][16 gw + 4 q ..]
    static_assert(HMM || NO <= 2, "narrow heads: two classes ride in the 4-float partial");
    f32x4 w3s[HMM ? 1 : NO];
    f32x4 hfr = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (HMM) {
      hfr = *(const f32x4*)(smem + twp.W3P + j * W2S + 16 * gw + 4 * q);
      w3s[0] = hfr;
    } else {
#pragma unroll
      for (int c = 0; c < NO; ++c)
        w3s[c] = c < n_out ? *(const f32x4*)(smem + twp.W3 + c * HID + 16 * gw + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // wave 0: row sums of W3g and the folded bias of the classes this lane finishes (narrow: all NO; wide: 4 q .. 4 q + 3)
    float sw3[HMM ? 4 : NO], b3v[HMM ? 4 : NO];
    if (gw == 0) {
      if constexpr (HMM) {
        const int no4 = (n_out + 3) & ~3;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const f32x4 w = *(const f32x4*)(smem + twp.W3P + (4 * q + r) * W2S + 4 * j);  // 4 of the row's 64 entries per lane j
          float s = (w[0] + w[1]) + (w[2] + w[3]);
          s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4); s += __shfl_xor(s, 8);
          sw3[r] = s;
          b3v[r] = 4 * q < no4 ? smem[twp.b3 + 4 * q + r] : 0.f;
        }
      } else {
#pragma unroll
        for (int c = 0; c < NO; ++c) {
          sw3[c] = c < n_out ? wave_sum(smem[twp.W3 + c * HID + l]) : 0.f;
          b3v[c] = c < n_out ? smem[twp.b3 + c] : 0.f;
        }
      }
    }
    // wide Gaussian heads: the per-dimension constants of Normal.log_prob, once per launch (wave 0 writes and reads them: a
    // wave's LDS operations execute in order) - per step they were an exp, a reciprocal and ~10 more instructions per dimension
    if (HMM && HEAD == ORL_HEAD_GAUSSIAN && gw == 0 && l < 16) {
      const float ls = l < n_out ? smem[twp.logstd + l] : 0.f;
      const float sd = expf(ls);
      smem[L.gtab + l] = sd;
      smem[L.gtab + 16 + l] = 1.f / (2.f * (sd * sd));
      smem[L.gtab + 32 + l] = ls + 0.91893853320467274178f;
    }
    // env state of the tile's rows (every lane of a row keeps a copy; lanes q == 0 write)
    constexpr int SW = ENV == ORL_ENV_SYNTH ? SYNTH_STATE_W : IS_TTT ? 0 : CARTPOLE_STATE_W;  // (tic-tac-toe: bitboards below)
    float est[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float ep_ret = 0.f, ep_len = 0.f, fin_ret = 0.f, fin_cnt = 0.f;
    if (gw == 0 && ok) {
#pragma unroll
      for (int k = 0; k < 8; ++k) est[k] = k < SW ? A.r.env_state[(size_t)n * SW + k] : 0.f;
      ep_ret = A.r.ep_stats[n * 4 + 0]; ep_len = A.r.ep_stats[n * 4 + 1];
      fin_ret = A.r.ep_stats[n * 4 + 2]; fin_cnt = A.r.ep_stats[n * 4 + 3];
    }
    int tA = 0, tO = 0, tmoves = 0, tep = 0;  // tic-tac-toe: this row's bitboards and counters (wave 0)
    if (IS_TTT && gw == 0) {
      const int w0 = s_word[j];
      tA = w0 & 0x1FF; tO = (w0 >> 9) & 0x1FF;
      tmoves = s_meta[j]; tep = s_meta[TILE_B + j];
    }
    asm volatile("" : "+v"(ep_ret), "+v"(ep_len), "+v"(fin_ret), "+v"(fin_cnt));
#pragma unroll
    for (int k = 0; k < 8; ++k) asm volatile("" : "+v"(est[k]));
    float* s_xg = smem + L.xg;
    float* s_part = smem + L.part;
    float* s_z2 = smem + L.z2;
    constexpr int PSLOT = TILE_B * 16 + 4 * TILE_B * 2;  // floats of one wave's partial record

#ifdef ORL_PROF
    t_last = __builtin_readcyclecounter();
#endif
    for (int t = 0; t < T; ++t) {
      const float* cur = s_obs + (t & (RO2_ORING - 1)) * TILE_B * DP;
      // ---- observation t: counter and operand in one batch, repeated until the counter covers the operand
      f32x4 x[4];
      float rstd;
      if constexpr (KS == 1) {
        float xb;
        for (unsigned spins = 0;; ++spins) {
          const unsigned c = ro2_ld(ctr + RC_OBS);
          xb = ro2_ldf(cur + j * 4 + q);
          if ((int)c >= t + 1) break;
          if (spins > RO2_MAX_SPINS) { ro2_fail(err); break; }
        }
        asm volatile("" ::: "memory");
        R2_T(0);
#pragma unroll
        for (int m = 0; m < 4; ++m) x[m] = ORL_MFMA(creg.w1[m][0], xb, creg.b1a[m]);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int r = 0; r < 4; ++r) x[m][r] = fmaxf(x[m][r], 0.f);
      } else {
        ro2_wait<false>(ctr + RC_OBS, t + 1, err);
        R2_T(0);
        const int wq = IS_TTT ? ttt_operand_bits(s_word[(t & (RO2_ORING - 1)) * TILE_B + j]) : 0;
        auto xop = [&](int s) -> float {
          if constexpr (IS_TTT) return (float)((wq >> (2 * s)) & 1);
          else return cur[j * DP + 4 * s + q];
        };
        // ---- fc1 + relu (small observations: all four M-tiles in every wave; wide: own M-tile, all-gather through LDS)
        if (DP <= COOP_SMALL_DP) {
#pragma unroll
          for (int m = 0; m < 4; ++m) x[m] = creg.b1a[m];
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            if (4 * s < DP) {
              const float xb = xop(s);
#pragma unroll
              for (int m = 0; m < 4; ++m) x[m] = ORL_MFMA(creg.w1[m][s], xb, x[m]);
            }
          }
#pragma unroll
          for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) x[m][r] = fmaxf(x[m][r], 0.f);
        } else {
          f32x4 acc = creg.b1;
#pragma unroll
          for (int s = 0; s < 8; ++s)
            if (4 * s < DP && (!IS_TTT || s < 5)) acc = ORL_MFMA(w1w[KS > 0 ? 0 : s], xop(s), acc);  // (tic-tac-toe: DP = 20)
          for (int s = 8; !IS_TTT && 4 * s < DP; ++s)
            acc = ORL_MFMA(smem[twp.W1 + (16 * gw + j) * DP + 4 * s + q], xop(s), acc);
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[r] = fmaxf(acc[r], 0.f);
          float* xg = s_xg + (t & 1) * TILE_B * GS;
          *(f32x4*)(xg + j * GS + 16 * gw + 4 * q) = acc;
          ro2_post(ctr + RC_XG + gw, t + 1);
          for (unsigned spins = 0;; ++spins) {  // the four flags in one read
            const u32x4 f = ro2_ld4u(ctr + RC_XG);
            if ((int)f[0] > t && (int)f[1] > t && (int)f[2] > t && (int)f[3] > t) break;
            if (spins > RO2_MAX_SPINS) { ro2_fail(err); break; }
          }
          asm volatile("" ::: "memory");
#pragma unroll
          for (int m = 0; m < 4; ++m) x[m] = *(const f32x4*)(xg + j * GS + 16 * m + 4 * q);
        }
      }
      R2_T(1);
      ln_normalize_T(x, rstd);  // xhat1 (the W2 image carries diag(g1), the bias slice W2 be1)
      R2_T(2);
      // ---- fc2: this wave's 16 output features, two chains of 8 MFMAs (the round-5 kernel's order)
      f32x4 z = creg.b2, z2b = f32x4{0.f, 0.f, 0.f, 0.f};
      if constexpr (PSPLIT) {
        u32x4 xs[2][2];
        split_Th(x, xs);
#pragma unroll
        for (int h = 0; h < 2; ++h) {  // mm64_T_h2's order for one row block
          z = mfma_f16_16(wA[h][1], xs[h][0], z);
          z = mfma_f16_16(wA[h][0], xs[h][1], z);
          z = mfma_f16_16(wA[h][0], xs[h][0], z);
        }
      } else {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            z = ORL_MFMA(creg.w2[mi][r], x[mi][r], z);
            z2b = ORL_MFMA(creg.w2[mi + 2][r], x[mi + 2][r], z2b);
          }
        }
        z = z + z2b;
      }
      R2_T(3);
      // ---- this wave's partials of LayerNorm 2 + head, and the z slice for the guarded path
      float* pw = s_part + ((t & 1) * 4 + gw) * PSLOT;
      const float p1 = (z[0] + z[1]) + (z[2] + z[3]);
      const float p2 = (z[0] * z[0] + z[1] * z[1]) + (z[2] * z[2] + z[3] * z[3]);
      if constexpr (HMM) {
        f32x4 pl = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) pl = ORL_MFMA(hfr[r], z[r], pl);
        *(f32x4*)(pw + j * 16 + 4 * q) = pl;
        *(f32x2*)(pw + TILE_B * 16 + (q * TILE_B + j) * 2) = f32x2{p1, p2};
      } else {
        f32x4 pv = f32x4{p1, p2, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < NO; ++c)
          pv[2 + c] = (w3s[c][0] * z[0] + w3s[c][1] * z[1]) + (w3s[c][2] * z[2] + w3s[c][3] * z[3]);
        *(f32x4*)(pw + (q * TILE_B + j) * 4) = pv;
      }
      *(f32x4*)(s_z2 + (t & 1) * TILE_B * GS + j * GS + 16 * gw + 4 * q) = z;
      ro2_post(ctr + RC_PART + gw, t + 1);
      R2_T(4);
      if (gw != 0) continue;

      // ================================ wave 0: finish the step ================================
      const uint64_t tg = tg0 + (uint64_t)t;
      // One batch: the partial flags, the service counters, the critic's consumption flags, and the payload they cover (the four
      // partials, the step's noise, the env service's record) - repeated until every counter covers its payload.  The service
      // counters are ahead of the chain almost always; the partial flags are what this loop really waits for.
      const float* pb = s_part + (t & 1) * 4 * PSLOT;
      const float* noise = smem + L.noise + (t & 3) * TILE_B * 16 + j * 16;
      const float* er = smem + L.env + ((t & 3) * TILE_B + j) * RO2_ENVW;
      f32x4 a4[4], envr0, envr1 = f32x4{0.f, 0.f, 0.f, 0.f}, nz4 = f32x4{0.f, 0.f, 0.f, 0.f};
      f32x2 st[4];
      for (unsigned spins = 0;; ++spins) {
        const u32x4 fp = ro2_ld4u(ctr + RC_PART), fs = ro2_ld4u(ctr + RC_NOISE), fc = ro2_ld4u(ctr + RC_CRIT);
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          if constexpr (HMM) {
            a4[w] = ro2_ld4f(pb + w * PSLOT + j * 16 + 4 * q);
            const f32x4 s4 = ro2_ld4f(pb + w * PSLOT + TILE_B * 16 + ((q * TILE_B + j) & ~1) * 2);  // (8-byte pair inside a 16-byte read)
            st[w] = (j & 1) ? f32x2{s4[2], s4[3]} : f32x2{s4[0], s4[1]};
          } else {
            a4[w] = ro2_ld4f(pb + w * PSLOT + (q * TILE_B + j) * 4);
          }
        }
        if constexpr (HMM && HEAD == ORL_HEAD_GAUSSIAN) nz4 = ro2_ld4f(noise + 4 * q);
        else nz4 = ro2_ld4f(noise);
        envr0 = ro2_ld4f(er);
        if (ENV == ORL_ENV_CARTPOLE) envr1 = ro2_ld4f(er + 4);
        // slot (t + 1) & 7 of the observation ring held observation t - 7: the critic wave of step t - 7 has read it when its
        // flag says so; the staging / env rings (4 deep) are free once the store wave has finished step t - 4
        const int need_c = t - 6;
        const bool ready = (int)fp[1] > t && (int)fp[2] > t && (int)fp[3] > t && (int)fs[0] > t && (int)fs[1] > t &&
                           (int)fs[2] >= t - 3 && (int)fc[0] >= need_c && (int)fc[1] >= need_c && (int)fc[2] >= need_c;
        if (ready) break;
        if (spins > RO2_MAX_SPINS) { ro2_fail(err); break; }
      }
      asm volatile("" ::: "memory");
      R2_T(5);
      float s1, s2;
      float hd[HMM ? 1 : NO];
      f32x4 lgv = f32x4{0.f, 0.f, 0.f, 0.f};
      if constexpr (HMM) {
        lgv = (a4[0] + a4[1]) + (a4[2] + a4[3]);
        const f32x2 ss = (st[0] + st[1]) + (st[2] + st[3]);
        s1 = ss[0]; s2 = ss[1];
        row_allsum2(s1, s2);
      } else {
        const f32x4 sv = (a4[0] + a4[1]) + (a4[2] + a4[3]);
        s1 = sv[0]; s2 = sv[1];
        float d0 = sv[2], d1 = sv[3];
        row_allsum2(s1, s2);
        row_allsum2(d0, d1);
        hd[0] = d0;
        if constexpr (NO > 1) hd[NO > 1 ? 1 : 0] = d1;
      }
      const float mean2 = s1 * (1.0f / 64.0f), ex2 = s2 * (1.0f / 64.0f);
      const float var2 = ex2 - mean2 * mean2;
      if (__builtin_expect(__builtin_amdgcn_ballot_w64(var2 * ORL_LN_GUARD < ex2) != 0ull, 0)) {
        // ill-conditioned row in the tile: the full 64-wide LayerNorm 2 (itself guarded: two-pass) and the head on xhat2
        f32x4 zz[4];
        float r2;
        const float* zt = s_z2 + (t & 1) * TILE_B * GS;
#pragma unroll
        for (int m = 0; m < 4; ++m) zz[m] = *(const f32x4*)(zt + j * GS + 16 * m + 4 * q);
        ln_normalize_T(zz, r2, ln2_eps);
        if constexpr (HMM) {
          const int no4 = (n_out + 3) & ~3;
          lgv = f32x4{0.f, 0.f, 0.f, 0.f};
          if (4 * q < no4) lgv = *(const f32x4*)(smem + twp.b3 + 4 * q);
#pragma unroll
          for (int mi = 0; mi < 4; ++mi) {
            const f32x4 wf = *(const f32x4*)(smem + twp.W3P + j * W2S + 16 * mi + 4 * q);
#pragma unroll
            for (int r = 0; r < 4; ++r) lgv = ORL_MFMA(wf[r], zz[mi][r], lgv);
          }
        } else {
          float h2[NO];
          head_T<NO>(smem + twp.W3, smem + twp.b3, n_out, zz, q, h2);
#pragma unroll
          for (int c = 0; c < NO; ++c) hd[c] = h2[c];
        }
      } else {
        const float rstd2 = __builtin_amdgcn_rsqf(fmaxf(var2, 0.f) + ln2_eps);
        const float mr = mean2 * rstd2;
        if constexpr (HMM) {
#pragma unroll
          for (int r = 0; r < 4; ++r) lgv[r] = (lgv[r] * rstd2 - mr * sw3[r]) + b3v[r];
        } else {
#pragma unroll
          for (int c = 0; c < NO; ++c) hd[c] = (hd[c] * rstd2 - mr * sw3[c]) + b3v[c];
        }
      }
      R2_T(6);
      // ---- sample (the noise of step t was drawn two steps ago by wave 5)
      float* stg = smem + L.stage + ((t & 3) * TILE_B + j) * RO2_STG;
      float act0 = 0.f;
      if constexpr (HMM && HEAD == ORL_HEAD_GAUSSIAN) {
        f32x4 av4 = f32x4{0.f, 0.f, 0.f, 0.f}, lp4 = f32x4{0.f, 0.f, 0.f, 0.f};
        // (the constants live in LDS: hoisting them into registers costs 12 this build does not have at 768 threads - it spilled)
        const f32x4 sd4 = *(const f32x4*)(smem + L.gtab + 4 * q), iv4 = *(const f32x4*)(smem + L.gtab + 16 + 4 * q),
                    lc4 = *(const f32x4*)(smem + L.gtab + 32 + 4 * q);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = 4 * q + r;
          if (c < n_out) {
            const float av = lgv[r] + sd4[r] * nz4[r];
            const float d = av - lgv[r];
            av4[r] = av;
            lp4[r] = -(d * d) * iv4[r] - lc4[r];
          }
        }
        *(f32x4*)(stg + 4 * q) = av4;
        *(f32x4*)(stg + 16 + 4 * q) = lp4;
      } else if constexpr (HMM) {
        float av, lp;
        f32x4 mk = f32x4{1.f, 1.f, 1.f, 1.f};
        if constexpr (IS_TTT) {  // classes 4 q .. 4 q + 3 of the row's legal moves; an invalid row has class 0 legal
          const int w0 = tA | (tO << 9);
          const int emp = (ok ? (~(w0 | (w0 >> 9)) & 0x1FF) : 1) >> (4 * q);
          mk = f32x4{(float)(emp & 1), (float)((emp >> 1) & 1), (float)((emp >> 2) & 1), (float)((emp >> 3) & 1)};
        }
        sample_cat_frag_v(lgv, n_out, q, mk, nz4[0], av, lp);
        act0 = av;
        if (q == 0) { stg[0] = av; stg[16] = lp; }
      } else {
        float act_o[NO], lp_o[NO];
        const float nzv[4] = {nz4[0], nz4[1], nz4[2], nz4[3]};
        if (HEAD == ORL_HEAD_CATEGORICAL && NO == 2 && n_out == 2) {
          ro2_sample_cat2(hd[0], hd[NO > 1 ? 1 : 0], nzv[0], act_o[0], lp_o[0]);
        } else {
          sample_head<NO, HEAD>(hd, n_out, smem + twp.logstd, nullptr, nzv, 0, A.r.act_seed, (uint64_t)n, tg, act_o, lp_o);
        }
        act0 = act_o[0];
        if (q == 0) {
#pragma unroll
          for (int c = 0; c < NO; ++c) {
            if (c < a_w) { stg[c] = act_o[c]; stg[16 + c] = lp_o[c]; }
          }
        }
      }
      R2_T(7);
      // ---- env step: the action-dependent half (wave 6 prepared the rest)
      float rew;
      bool done;
      if (ENV == ORL_ENV_SYNTH) {
        rew = envr0[0];
        const float c = est[0] + 1.f;
        done = c >= (float)A.r.episode_limit;
        est[0] = done ? 0.f : c;
        // (observation t + 1 is in the ring already: it does not depend on the action)
      } else if constexpr (IS_TTT) {
        // orl_ttt_step's game logic on bitboards (every lane of the row runs it: no divergence); the step's draws are wave 6's
        const uint32_t rx = f2u(envr0[0]), r_begin = f2u(envr0[1]), r_open = f2u(envr0[2]);
        const int a = (int)act0;
        const int e0 = ~(tA | tO) & 0x1FF;
        rew = 0.f; done = false;
        const bool legal = (unsigned)a < 9u && ((e0 >> a) & 1) != 0;
        if (!legal) { rew = -1.f; done = true; }
        else {
          tA |= 1 << a;
          if (ttt_wins_bits(tA)) { rew = 1.f; done = true; }
          else if ((tA | tO) == 0x1FF) done = true;
          else {
            tO |= ttt_pick_empty_bits(~(tA | tO) & 0x1FF, rx);
            ++tmoves;
            if (ttt_wins_bits(tO)) { rew = -1.f; done = true; }
            else if ((tA | tO) == 0x1FF) done = true;
          }
        }
        if (done) {
          ++tep;
          tA = 0; tO = 0; tmoves = 0;
          if (r_begin & 1u) { tO = ttt_pick_empty_bits(0x1FF, r_open); tmoves = 1; }
        }
        if (q == 0) {
          const int sl = ((t + 1) & (RO2_ORING - 1));
          s_word[sl * TILE_B + j] = ok ? (tA | (tO << 9) | (1 << 18)) : 0;
          s_meta[sl * 2 * TILE_B + j] = tmoves; s_meta[sl * 2 * TILE_B + TILE_B + j] = tep;
        }
      } else {
        CartPolePre cp;
        cp.costh = envr0[0]; cp.sinth = envr0[1]; cp.t1 = envr0[2]; cp.den = envr0[3];
        float s[4] = {est[0], est[1], est[2], est[3]};
        const bool term = cartpole_post(s, cp, (int)act0);
        const float steps = est[4] + 1.f;
        done = term || steps >= (float)A.r.episode_limit;
        rew = 1.0f;
        est[4] = done ? 0.f : steps;
        est[5] += done ? 1.f : 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) est[k] = done ? envr1[k] : s[k];  // auto-reset: the first observation of the next episode
        float* nxt = s_obs + ((t + 1) & (RO2_ORING - 1)) * TILE_B * DP;
        if (q == 0) *(f32x4*)(nxt + j * DP) = ok ? f32x4{est[0], est[1], est[2], est[3]} : f32x4{0.f, 0.f, 0.f, 0.f};
      }
      ep_ret += rew; ep_len += 1.f;
      if (done) { fin_ret += ep_ret; fin_cnt += 1.f; ep_ret = 0.f; ep_len = 0.f; }
      if (q == 0) *(f32x2*)(stg + 32) = f32x2{rew, done ? 1.f : 0.f};
      ro2_post(ctr + RC_STAGE, t + 1);
      ro2_post(ctr + RC_OBS, t + 2);
      R2_T(8);
    }
#ifdef ORL_PROF
    if (prof_on && l < 16) atomicAdd(&g_roll_prof[l], prof_lds[l]);
#endif
    if (gw == 0) {
      const bool poisoned = ro2_ld(err) != 0u;
      if (q == 0 && ok) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
          if (k < SW) A.r.env_state[(size_t)n * SW + k] = est[k];
        if constexpr (IS_TTT) {
          float* st = A.r.env_state + (size_t)n * TTT_STATE_W;
#pragma unroll
          for (int c = 0; c < 9; ++c) st[c] = ((tA >> c) & 1) ? 1.f : (((tO >> c) & 1) ? 2.f : 0.f);
          st[9] = (float)tmoves; st[10] = (float)tep;
        }
        A.r.ep_stats[n * 4 + 0] = ep_ret; A.r.ep_stats[n * 4 + 1] = ep_len;
        A.r.ep_stats[n * 4 + 2] = poisoned ? u2f(0x7fc00000u) : fin_ret; A.r.ep_stats[n * 4 + 3] = fin_cnt;
      }
    }
    return;
  }
  __builtin_amdgcn_s_setprio(0);
  if (wave == 5) {
    // ================================================================ sampling noise, two steps ahead ============
    for (int t = 0; t < T; ++t) {
      ro2_wait(ctr + RC_OBS, t - 2, err);  // slot t & 3 held step t - 4's noise: consumed before observation t - 3 was published
      float* dst = smem + L.noise + (t & 3) * TILE_B * 16;
      const uint64_t tg = tg0 + (uint64_t)t;
      if (HEAD == ORL_HEAD_CATEGORICAL) {
        const u4 r = philox4x32_10(A.r.act_seed, (uint32_t)n, (uint32_t)((uint64_t)n >> 32), (uint32_t)tg, (uint32_t)(tg >> 32) << 8);
        if (q == 0) dst[j * 16] = u01(r.x);
      } else if (4 * q < n_out) {
        const u4 r = philox4x32_10(A.r.act_seed, (uint32_t)n, (uint32_t)((uint64_t)n >> 32), (uint32_t)tg,
                                   ((uint32_t)(tg >> 32) << 8) | (uint32_t)q);
        float e[4];
        box_muller(r.x, r.y, e[0], e[1]);
        box_muller(r.z, r.w, e[2], e[3]);
        *(f32x4*)(dst + j * 16 + 4 * q) = f32x4{e[0], e[1], e[2], e[3]};
      }
      ro2_post(ctr + RC_NOISE, t + 1);
    }
  } else if (wave == 6) {
    // ================================================================ the action-independent half of env.step =====
    float ep = (ENV == ORL_ENV_CARTPOLE && ok) ? A.r.env_state[(size_t)n * CARTPOLE_STATE_W + 5] : 0.f;
    for (int t = 0; t < T; ++t) {
      float* er = smem + L.env + ((t & 3) * TILE_B + j) * RO2_ENVW;
      const uint64_t tg = tg0 + (uint64_t)t;
      if (ENV == ORL_ENV_SYNTH) {
        // reward t and observation t + 1 (ring slot (t + 1) & 7: held observation t - 7 - read by the chain long ago, by the store
        // wave in its iteration t - 8 and by the critic wave of step t - 7; the env record slot t & 3 was read in step t - 4)
        ro2_wait(ctr + RC_OBS, t - 1, err);
        ro2_wait(ctr + RC_STORED, t - 3, err);
#pragma unroll
        for (int c = 0; c < 3; ++c) ro2_wait(ctr + RC_CRIT + c, t - 6, err);
        float* nxt = s_obs + ((t + 1) & (RO2_ORING - 1)) * TILE_B * DP;
        if (q == 0) er[0] = ok ? synth_reward(A.r.env_seed, (uint32_t)n, tg) : 0.f;
        for (int bb = q; bb < (D + 3) / 4; bb += 4) {
          float o[4];
          synth_obs_block(A.r.env_seed, (uint32_t)n, tg + 1, (uint32_t)bb, o);
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int d = 4 * bb + k;
            if (d < DP) nxt[j * DP + d] = (ok && d < D) ? o[k] : 0.f;
          }
        }
      } else if constexpr (IS_TTT) {
        // the step's three Philox words (ttt_draw: lane group q = 0 the opponent's reply, 1 = who opens the next game, 2 = that
        // game's opening move) from the (episode, opponent moves) wave 0 published with observation t
        ro2_wait(ctr + RC_OBS, t + 1, err);
        const int* mt = s_meta + (t & (RO2_ORING - 1)) * 2 * TILE_B;
        const int mv = mt[j], epi = mt[TILE_B + j];
        if (q < 3) er[q] = u2f(ttt_draw(A.r.env_seed, (uint32_t)n, q, (uint32_t)(epi + (q != 0 ? 1 : 0)), (uint32_t)mv));
      } else {
        ro2_wait(ctr + RC_OBS, t + 1, err);
        if (t >= 1) ep += smem[L.stage + (((t - 1) & 3) * TILE_B + j) * RO2_STG + 33];  // done of step t - 1 (staged before obs t)
        const f32x4 sv = *(const f32x4*)(s_obs + (t & (RO2_ORING - 1)) * TILE_B * DP + j * DP);
        const float s[4] = {sv[0], sv[1], sv[2], sv[3]};
        const CartPolePre cp = cartpole_pre(s);
        float rs[4];
        cartpole_reset(A.r.env_seed, (uint32_t)n, (uint32_t)(ep + 1.f), rs);
        if (q == 0) {
          *(f32x4*)er = f32x4{cp.costh, cp.sinth, cp.t1, cp.den};
          *(f32x4*)(er + 4) = f32x4{rs[0], rs[1], rs[2], rs[3]};
        }
      }
      ro2_post(ctr + RC_ENV, t + 1);
    }
  } else if (wave == 7) {
    // ================================================================ the step's rows of the rollout buffer ========
    for (int t = 0; t < T; ++t) {
      ro2_wait(ctr + RC_STAGE, t + 1, err);
      ro2_wait(ctr + RC_OBS, t + 2, err);
      const float* stg = smem + L.stage + (t & 3) * TILE_B * RO2_STG;
      const float* nxt = s_obs + ((t + 1) & (RO2_ORING - 1)) * TILE_B * DP;
      const size_t r0 = (size_t)t * N + n0, r1 = (size_t)(t + 1) * N + n0;
      for (int e = l; e < nrow * a_w; e += 64) {  // actions / log-probs: the tile's rows are contiguous
        const int rr = e / a_w, c = e - rr * a_w;
        A.r.actions[r0 * a_w + e] = stg[rr * RO2_STG + c];
        A.r.action_log_probs[r0 * a_w + e] = stg[rr * RO2_STG + 16 + c];
      }
      if (l < nrow) {
        const float dn = stg[l * RO2_STG + 33];
        b.rewards[r0 + l] = stg[l * RO2_STG + 32];
        b.masks[r1 + l] = dn != 0.f ? 0.f : 1.f;
        b.active_masks[r1 + l] = 1.f;
        b.bad_masks[r1 + l] = 1.f;
      }
      if constexpr (IS_TTT) {
        // the 18 observation floats and 9 legal-move floats of every row, expanded from the tile's board words
        const int* words = s_word + ((t + 1) & (RO2_ORING - 1)) * TILE_B;
        for (unsigned e = l; e < (unsigned)nrow * 27u; e += 64u) {
          if (e < (unsigned)nrow * 18u) {
            const unsigned rr = e / 18u, d = e - rr * 18u;
            const float v = (float)((words[rr] >> ((d & 1u) * 9u + (d >> 1))) & 1);
            b.policy_obs[r1 * 18 + e] = v;
            if (sep_c) b.critic_obs[r1 * 18 + e] = v;
          } else {
            const unsigned e2 = e - (unsigned)nrow * 18u, rr = e2 / 9u, c = e2 - rr * 9u;
            const int w = words[rr];
            b.action_masks[r1 * 9 + e2] = (float)((~(w | (w >> 9)) >> c) & 1);
          }
        }
      } else {
        for (int e = l; e < nrow * D; e += 64) {
          const int rr = e / D, d = e - rr * D;
          const float v = nxt[rr * DP + d];
          b.policy_obs[r1 * D + e] = v;
          if (sep_c) b.critic_obs[r1 * D + e] = v;
        }
        if (b.action_masks != nullptr)
          for (int e = l; e < nrow * b.K; e += 64) b.action_masks[r1 * b.K + e] = 1.f;
      }
      ro2_post(ctr + RC_STORED, t + 1);
    }
    if (ro2_ld(err) != 0u && l < nrow) b.rewards[(size_t)n0 + l] = u2f(0x7fc00000u);  // a poll timed out: the tile's data are void
  } else if (wave >= 9 && wave <= 11) {
    // ================================================================ the critic, off the chain ==================
    // wave c evaluates V(observation t) for t = c, c + 3, ... <= T: the whole value tower for one 16-row tile per trip, fc2 on
    // the 16-bit MFMA (mm64_T_h2 over the scaled fp16 image of W2 diag(g1)), everything else as critic_sweep_kernel / the update tower
    const int c = wave - 9;
    const float* lc = smem + L.critic;
    f32x4 w3[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) w3[m] = *(const f32x4*)(lc + twc.W3 + 16 * m + 4 * q);
    const float b3 = lc[twc.b3];
    const int nk = DP >> 2;
    for (int t = c; t <= T; t += 3) {
      ro2_wait(ctr + RC_OBS, t + 1, err);
      const float* cur = s_obs + (t & (RO2_ORING - 1)) * TILE_B * DP;
      f32x4 z[4];
      load_vec_T(lc + twc.b1, q, z);
      if constexpr (KS == 1) {
        const float xb = cur[j * 4 + q];
        ro2_post(ctr + RC_CRIT + c, t + 1);  // the ring slot is free again (LDS reads of a wave complete in order: the read above is first)
#pragma unroll
        for (int m = 0; m < 4; ++m) z[m] = ORL_MFMA(lc[twc.W1 + (16 * m + j) * 4 + q], xb, z[m]);
      } else {
        float xr[16];
        if constexpr (IS_TTT) {
          const int wq = ttt_operand_bits(s_word[(t & (RO2_ORING - 1)) * TILE_B + j]);
#pragma unroll
          for (int s = 0; s < 16; ++s) xr[s] = (float)((wq >> (2 * s)) & 1);
        } else {
#pragma unroll
          for (int s = 0; s < 16; ++s) xr[s] = s < nk ? cur[j * DP + 4 * s + q] : 0.f;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the operands are in registers before the slot is released
        ro2_post(ctr + RC_CRIT + c, t + 1);
#pragma unroll
        for (int s = 0; s < 16; ++s) {
          if (s < nk) {
#pragma unroll
            for (int m = 0; m < 4; ++m) z[m] = ORL_MFMA(lc[twc.W1 + (16 * m + j) * DP + 4 * s + q], xr[s], z[m]);
          }
        }
      }
      relu_T(z);
      float rstd;
      ln_normalize_T(z, rstd);
      f32x4 acc[4];
      load_vec_T(lc + twc.b2, q, acc);
#if ORL_TOWER_F16
      {  // two-term fp16 split over the scaled image (orl_mlp.h): acc = 2^kw z2, LayerNorm 2 with eps x 4^kw returns xhat2 itself
        u32x4 xs[2][2];
        split_Th(z, xs);
        mm64_T_h2((const unsigned short*)(lc + twc.W2), xs, acc, j, q);
      }
      ln_normalize_T(acc, rstd, lc[twc.wsc + 3]);
#else
      {
        u32x4 xs[2][3];
        split_T(z, xs);
        mm64_T_split((const unsigned short*)(lc + twc.W2), xs, acc, j, q);
      }
      ln_normalize_T(acc, rstd);
#endif
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
  }
}
