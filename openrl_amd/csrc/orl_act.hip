// orl_act.hip - rollout side of the hot path for gfx950:
//   orl_act_step      : K1-K4 fused policy + value forward and action sampling for B rows
//   orl_rollout_fused : T steps of {forward, sample, env.step, buffer insert} in ONE launch for
//                       device-resident batched envs (synthetic fixed-step env, CartPole-v1)
//   orl_env_reset     : initial env state + first observation
//
// Geometry: one workgroup owns one 16-row tile (fp32 MFMA 16x16x4, see orl_mlp.h).  orl_act_step uses 2
// waves (policy tower, critic tower).  Env lanes are independent, so the fused rollout needs no
// inter-workgroup communication: each 8-wave workgroup owns 16 envs for all T steps, keeps both towers'
// weights in LDS for the whole rollout and splits each tower's GEMMs over 4 waves (trunk_fwd_coop) because a
// rollout step is a dependent-latency chain, not a throughput problem.
#include "orl_common.h"
#include "orl_mlp.h"
#include "orl_heads.h"
#include "orl_env.h"
#include "orl_ttt.h"

namespace orl {

struct ActArgs {
  orl_net_desc pnet, cnet;
  const float* ptheta;
  const float* ctheta;  // may be NULL
  const float* pobs;
  const float* cobs;
  const float* amask;   // [B, n_out] or NULL
  const float* forced;  // [B, a] or NULL
  float* values;
  float* actions;
  float* logp;
  int B;
  int deterministic;
  uint64_t seed, row0, rng_step;
  const unsigned long long* rng_dev;  // optional device-side addend of rng_step (rng_step_dev argument)
  int group_rows;          // > 0: rows [g*group_rows, (g+1)*group_rows) use the policy parameters ptheta + g*theta_stride
  long long theta_stride;  //      (orl_act_step_grouped: one launch for a pool of policies); a multiple of 16 rows
  const int* opp_index;    // orl_act_step_pool: per-row policy index (NULL otherwise)
  int n_policies;
};

template <int NO, int HEAD>
__global__ __launch_bounds__(128) void act_step_kernel(ActArgs A) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const TowerLayout tlp(A.pnet);
  constexpr bool HMM = NO > 4;  // wide heads on MFMA (head_mfma_T): same logits in the rollout, act and update kernels
  const TowerLds twp(A.pnet.obs_dim, A.pnet.n_out, HEAD == ORL_HEAD_GAUSSIAN, false, HMM);
  TowerLayout tlc;
  TowerLds twc;
  const bool has_c = A.ctheta != nullptr;
  const bool has_p = A.ptheta != nullptr;
  const float* ptheta = A.ptheta;
  if (has_p && A.group_rows > 0) ptheta += (size_t)((blockIdx.x * TILE_B) / A.group_rows) * A.theta_stride;
  if (has_p && A.opp_index == nullptr) stage_tower(smem, ptheta, tlp, twp, false, threadIdx.x, blockDim.x, HMM);
  if (has_c) {
    tlc = TowerLayout(A.cnet);
    twc = TowerLds(A.cnet.obs_dim, 1, false, false);
    stage_tower(smem + twp.total, A.ctheta, tlc, twc, false, threadIdx.x, blockDim.x);
  }
  __syncthreads();

  const int wave = threadIdx.x >> 6, l = threadIdx.x & 63, j = l & 15, q = l >> 4;
  const int row = blockIdx.x * TILE_B + j;
  const bool row_ok = row < A.B;
  // orl_act_step_pool: every row names its own policy.  The tile runs the tower once per distinct index it holds
  // (workgroup-uniform loop: stage policy k, evaluate, rows whose index is k keep the result).
  const int my_k = (A.opp_index != nullptr && row_ok) ? A.opp_index[row] : -1;
  const int n_pass = A.opp_index != nullptr ? A.n_policies : 1;
  for (int pass = 0; pass < n_pass; ++pass) {
  if (A.opp_index != nullptr) {
    if (!__syncthreads_or(my_k == pass)) continue;  // also orders the previous pass's LDS reads before restaging
    stage_tower(smem, A.ptheta + (size_t)pass * A.theta_stride, tlp, twp, false, threadIdx.x, blockDim.x, HMM);
    __syncthreads();
  }
  const bool keep = A.opp_index == nullptr || my_k == pass;
  if (wave == 0) {
    if (!has_p) return;
    const int D = A.pnet.obs_dim;
    const float* xrow = A.pobs + (size_t)row * D;
    auto xb = [&](int s) -> float {
      const int k = 4 * s + q;
      return (row_ok && k < D) ? xrow[k] : 0.f;
    };
    f32x4 n2[4];
    trunk_fwd_T(smem, twp, xb, j, q, n2);
    float hd[NO];
    if constexpr (HMM) {
      float* tile = smem + twp.total + (has_c ? TowerLds(A.cnet.obs_dim, 1, false, false).total : 0);  // [16][16]
      head_mfma_T<NO>(smem + twp.W3P, smem + twp.b3, A.pnet.n_out, n2, tile, j, q, hd);
    } else {
      head_T<NO>(smem + twp.W3, smem + twp.b3, A.pnet.n_out, n2, q, hd);
    }
    const int n_out = A.pnet.n_out;
    const int a_w = (HEAD == ORL_HEAD_CATEGORICAL) ? 1 : n_out;
    float act_o[NO], lp_o[NO];
    const float* am = (A.amask != nullptr && row_ok) ? A.amask + (size_t)row * n_out : nullptr;
    const float* fr = (A.forced != nullptr && row_ok) ? A.forced + (size_t)row * a_w : nullptr;
    sample_head<NO, HEAD>(hd, n_out, smem + twp.logstd, am, fr, A.deterministic, A.seed, A.row0 + (uint64_t)row,
                          A.rng_step + (A.rng_dev ? *A.rng_dev : 0ull), act_o, lp_o);
    if (row_ok && q == 0 && keep) {
#pragma unroll
      for (int c = 0; c < NO; ++c) {
        if (c < a_w) {
          A.actions[(size_t)row * a_w + c] = act_o[c];
          A.logp[(size_t)row * a_w + c] = lp_o[c];
        }
      }
    }
  } else if (has_c) {
    const int D = A.cnet.obs_dim;
    const float* xrow = A.cobs + (size_t)row * D;
    auto xb = [&](int s) -> float {
      const int k = 4 * s + q;
      return (row_ok && k < D) ? xrow[k] : 0.f;
    };
    f32x4 n2[4];
    const float* lc = smem + twp.total;
    trunk_fwd_T(lc, twc, xb, j, q, n2);
    float v[1];
    head_T<1>(lc + twc.W3, lc + twc.b3, 1, n2, q, v);
    if (row_ok && q == 0) A.values[row] = v[0];
  }
  }  // pass
}

// Opponent sampling for the self-play pool (see orl_opponent_sample in include/orl_hip.h)
__global__ void opponent_sample_kernel(int* __restrict__ opp_index, const uint8_t* __restrict__ dones, int N,
                                       int n_filled, int last_slot, int strategy, int per_tile, uint64_t seed,
                                       uint64_t draw_id, const unsigned long long* __restrict__ draw_id_dev) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  if (dones != nullptr && dones[n] == 0) return;
  const uint64_t id = draw_id + (draw_id_dev ? *draw_id_dev : 0ull);
  const uint32_t key = per_tile ? (uint32_t)(n / TILE_B) : (uint32_t)n;
  int k = last_slot;
  if (strategy == 0) {
    const u4 r = philox4x32_10(seed, key, 0x0FF05A3Fu, (uint32_t)id, (uint32_t)(id >> 32));
    k = (int)(u01(r.x) * (float)n_filled);
    k = k < n_filled - 1 ? k : n_filled - 1;
  }
  opp_index[n] = k < 0 ? 0 : k;
}

template <int ENV>
__global__ __launch_bounds__(256) void env_reset_kernel(float* __restrict__ env_state, float* __restrict__ ep_stats,
                                                        float* __restrict__ obs0, int N, int D, uint64_t seed,
                                                        int episode_limit) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  if (ep_stats != nullptr) {
    ep_stats[n * 4 + 0] = 0.f; ep_stats[n * 4 + 1] = 0.f; ep_stats[n * 4 + 2] = 0.f; ep_stats[n * 4 + 3] = 0.f;
  }
  if (ENV == ORL_ENV_SYNTH) {
    float* st = env_state + (size_t)n * SYNTH_STATE_W;
    st[0] = (float)((n * 7) % episode_limit);  // per-env phase offset so that dones are staggered
    st[1] = st[2] = st[3] = 0.f;
    for (int b = 0; b < (D + 3) / 4; ++b) {
      float o[4];
      synth_obs_block(seed, (uint32_t)n, 0, (uint32_t)b, o);
      for (int k = 0; k < 4; ++k)
        if (4 * b + k < D) obs0[(size_t)n * D + 4 * b + k] = o[k];
    }
  } else {
    float* st = env_state + (size_t)n * CARTPOLE_STATE_W;
    float s[4];
    cartpole_reset(seed, (uint32_t)n, 0u, s);
    for (int k = 0; k < 4; ++k) { st[k] = s[k]; obs0[(size_t)n * 4 + k] = s[k]; }
    st[4] = 0.f; st[5] = 0.f; st[6] = 0.f; st[7] = 0.f;
  }
}

// --------------------------------------------------------------------------------------------------
// Fused T-step rollout.  LDS: [policy tower | critic tower | obs tile 2 x 16 x DP].
// --------------------------------------------------------------------------------------------------
struct RolloutArgs {
  orl_net_desc pnet, cnet;
  const float* ptheta;
  const float* ctheta;
  orl_rollout_args r;
  float* next_value;  // [N] bootstrap value of slot T (may be NULL)
};

// Trunk forward of ONE 16-row tile split over the 4 waves of a tower group: wave `gw` owns M-tile gw (16 of the
// 64 hidden features) of both GEMMs, the pre-LayerNorm activations are all-gathered through LDS (one
// workgroup barrier each), and every wave then normalises the full tile itself, so the result in each wave is
// bit-identical to trunk_fwd_T (same k order per output element).  Per step this shortens the dependent MFMA
// chain from 68 to 17 instructions: the rollout is a latency problem (one 16-env tile per 8-wave workgroup),
// not a throughput one.  Every wave of the workgroup must call this the same number of times (barriers).
constexpr int GS = 68;  // gather-slab row stride (floats)
#ifdef ORL_PROF
// Phase timing build (tools/rollout_phase_prof.py): waves 0 and 1 of workgroup 0 stamp the shader clock after each
// phase of a rollout step; slots 0-7 = wave 0 (policy leader), 8-15 = wave 1 (env wave of the synthetic env).
__device__ unsigned long long g_roll_prof[16];
struct RollProf {
  unsigned long long* lds;
  unsigned long long t_last;
  bool on;
  int base;
};
#define RO_T(rp, k)                                                             \
  do {                                                                          \
    if ((rp).on) {                                                              \
      const unsigned long long t_now = __builtin_readcyclecounter();            \
      if ((threadIdx.x & 63) == 0) atomicAdd(&(rp).lds[(rp).base + (k)], t_now - (rp).t_last); \
      (rp).t_last = t_now;                                                      \
    }                                                                           \
  } while (0)
#else
struct RollProf {};
#define RO_T(rp, k) do {} while (0)
#endif
// Round 4: three cuts of the per-step chain:
//  * the towers' LDS images are staged with the LayerNorm affines FOLDED into the next Linear (stage_tower(fold), exact
//    algebra, the update kernel's form): both affine passes of a step (8 LDS reads + 16 FMAs each, in the dependent
//    chain) are gone - the GEMMs run on xhat;
//  * every operand that does not change over the T steps - this wave's 16 rows of W2 (4 fragments), its bias slices, and
//    on the small-observation path W1 - is read from LDS ONCE into registers (CoopRegs); hipcc cannot hoist those reads
//    itself (LDS stores inside the loop may alias);
//  * small observations (DP <= 8: 1 - 2 k-steps): every wave computes ALL four M-tiles of fc1 itself (4 - 8 independent
//    MFMAs instead of 1 - 2) and skips the first all-gather - one workgroup barrier and one LDS round trip per step less.
//    Same k order per output element: bit-identical to the gathered form.
// trunk_barriers(DP) = workgroup barriers one call executes (callers that only keep the barrier count need it).
#ifndef ORL_COOP_FC2_CHAINS
#define ORL_COOP_FC2_CHAINS 2
#endif
constexpr int COOP_SMALL_DP = 8;
__host__ __device__ inline int trunk_barriers(int DP) { return DP <= COOP_SMALL_DP ? 1 : 2; }

struct CoopRegs {
  f32x4 w2[4];     // W2 (folded) rows 16 gw + j, columns 16 mi + 4 q .. + 3
  f32x4 b2;        // folded bias slice 16 gw + 4 q ..
  f32x4 b1;        // gathered path: bias slice of this wave's M-tile
  f32x4 b1a[4];    // small-observation path: all four M-tiles' bias slices
  float w1[4][2];  // small-observation path: W1[16 m + j][4 s + q], s < DP / 4 <= 2
};
__device__ inline void coop_load(const float* __restrict__ lds, const TowerLds& tw, int gw, int j, int q, CoopRegs& R) {
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) R.w2[mi] = *(const f32x4*)(lds + tw.W2 + (16 * gw + j) * W2S + 16 * mi + 4 * q);
  R.b2 = *(const f32x4*)(lds + tw.b2 + 16 * gw + 4 * q);
  R.b1 = *(const f32x4*)(lds + tw.b1 + 16 * gw + 4 * q);
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    R.b1a[m] = *(const f32x4*)(lds + tw.b1 + 16 * m + 4 * q);
#pragma unroll
    for (int s = 0; s < 2; ++s)
      R.w1[m][s] = (tw.DP <= COOP_SMALL_DP && 4 * s < tw.DP) ? lds[tw.W1 + (16 * m + j) * tw.DP + 4 * s + q] : 0.f;
  }
}

// returns xhat2 (post-LN2, pre-affine: the head images carry diag(g2) and the folded bias)
template <class XB>
__device__ inline void trunk_fwd_coop(const float* __restrict__ lds, const TowerLds& tw, const CoopRegs& R, XB xb,
                                      float* __restrict__ gA, float* __restrict__ gB, int gw, int j, int q,
                                      f32x4 (&n2)[4], RollProf& rp) {
  f32x4 x[4];
  float rstd;
  if (tw.DP <= COOP_SMALL_DP) {
#pragma unroll
    for (int m = 0; m < 4; ++m) x[m] = R.b1a[m];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      if (4 * s < tw.DP) {
        const float b = xb(s);
#pragma unroll
        for (int m = 0; m < 4; ++m) x[m] = ORL_MFMA(R.w1[m][s], b, x[m]);
      }
    }
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) x[m][r] = fmaxf(x[m][r], 0.f);
    RO_T(rp, 0);
    RO_T(rp, 1);
  } else {
    f32x4 acc = R.b1;
    for (int s = 0; s < (tw.DP >> 2); ++s) {
      const float a = lds[tw.W1 + (16 * gw + j) * tw.DP + 4 * s + q];
      acc = ORL_MFMA(a, xb(s), acc);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = fmaxf(acc[r], 0.f);
    *(f32x4*)(gA + j * GS + 16 * gw + 4 * q) = acc;
    RO_T(rp, 0);
    __syncthreads();
    RO_T(rp, 1);
#pragma unroll
    for (int m = 0; m < 4; ++m) x[m] = *(const f32x4*)(gA + j * GS + 16 * m + 4 * q);
  }
  ln_normalize_T(x, rstd);  // xhat1: the GEMM runs on it (W2 diag(g1) image, folded bias)
  f32x4 acc = R.b2;
#if ORL_COOP_FC2_CHAINS == 2
  // build-time switch (round 5): the 16 dependent fp32 MFMAs of this wave's fc2 M-tile as TWO chains of 8 (k-blocks 0 - 1 and
  // 2 - 3) that interleave in the MFMA pipe - the step's dependent chain is 8 x 32 cycles + one add instead of 16 x 32.  A
  // different fp32 summation order than the stepwise kernel (the fused-vs-stepwise tests compare at rtol 1e-5 already).
  f32x4 acc2 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      acc = ORL_MFMA(R.w2[mi][r], x[mi][r], acc);
      acc2 = ORL_MFMA(R.w2[mi + 2][r], x[mi + 2][r], acc2);
    }
  }
  acc = acc + acc2;
#else
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
    for (int r = 0; r < 4; ++r) acc = ORL_MFMA(R.w2[mi][r], x[mi][r], acc);
  }
#endif
  *(f32x4*)(gB + j * GS + 16 * gw + 4 * q) = acc;
  RO_T(rp, 2);
  __syncthreads();
  RO_T(rp, 3);
#pragma unroll
  for (int m = 0; m < 4; ++m) n2[m] = *(const f32x4*)(gB + j * GS + 16 * m + 4 * q);
  ln_normalize_T(n2, rstd);
  RO_T(rp, 4);
}

// WC = false: policy-only rollout with 4 waves (one per SIMD); the values of all T+1 slots are then computed by ONE
// batched launch of critic_values_kernel over the stored observations - same parameters, the same arithmetic up to
// the few-ulp contraction differences between the cooperative and the per-wave trunk, but throughput-bound instead of sharing the SIMDs of a
// latency-bound step loop with the policy.
// internal variant of ORL_ENV_TTT_POOL (orl_rollout_args.opp_per_reset): every env names its own pool slot, up to 4
// snapshot images stay resident in LDS (the critic then runs after the fact: WC = false), the opponent's tower is
// walked once per distinct slot a tile holds, and a finished game draws its next opponent in-kernel with
// orl_opponent_sample's Philox stream
#define ORL_ENV_TTT_POOLK 100
template <int NO, int HEAD, int ENV, bool WC>
__global__ __launch_bounds__(512) void rollout_kernel(RolloutArgs A) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const orl_buffer_ptrs& b = A.r.buf;
  const int D = A.pnet.obs_dim;
  const int N = b.N, T = b.T;
  const TowerLayout tlp(A.pnet), tlc(A.cnet);
  constexpr bool HMM = NO > 4;  // wide heads on MFMA (head_mfma_T)
  const TowerLds twp(D, A.pnet.n_out, HEAD == ORL_HEAD_GAUSSIAN, false, HMM);
  const TowerLds twc(D, 1, false, false);
  // LayerNorm affines folded into W2 / W3 / the biases (see trunk_fwd_coop)
  stage_tower(smem, A.ptheta, tlp, twp, false, threadIdx.x, blockDim.x, HMM, false, true);
  if (WC) stage_tower(smem + twp.total, A.ctheta, tlc, twc, false, threadIdx.x, blockDim.x, false, false, true);
  const int DP = twp.DP;
  float* s_obs = smem + twp.total + (WC ? twc.total : 0);  // [2][16][DP]
  float* s_gather = s_obs + 2 * TILE_B * DP;    // [2 towers][2 slabs][16][GS]
  float* s_noise = s_gather + (WC ? 4 : 2) * TILE_B * GS;  // [2][16][16]: sampling noise of step t (parity), drawn one step ahead
  float* s_logits = s_noise + 2 * TILE_B * 16;  // [16][16] logits tile of head_mfma_T (wide heads only)

  // waves 0-3: policy tower (wave 0 also samples and steps the env); waves 4-7: critic tower
  const int wave = threadIdx.x >> 6, l = threadIdx.x & 63, j = l & 15, q = l >> 4;
  const int grp = wave >> 2, gw = wave & 3;
  const float* tlds = grp == 0 ? smem : smem + twp.total;
  const TowerLds& tww = grp == 0 ? twp : twc;
  float* gA = s_gather + grp * 2 * TILE_B * GS;
  float* gB = gA + TILE_B * GS;
  const int n = blockIdx.x * TILE_B + j;
  const bool ok = n < N;
  const int n_out = A.pnet.n_out;
  const int a_w = (HEAD == ORL_HEAD_CATEGORICAL) ? 1 : n_out;
  const bool sep_c = b.critic_obs != b.policy_obs;

  // obs tile for slot 0 comes from the buffer (after_update / init_buffer put it there)
  for (int e = threadIdx.x; e < TILE_B * DP; e += blockDim.x) {
    const int jj = e / DP, k = e - jj * DP;
    const int nn = blockIdx.x * TILE_B + jj;
    s_obs[e] = (nn < N && k < D) ? b.policy_obs[(size_t)nn * D + k] : 0.f;
  }
  // env state in registers of the owning lanes (q == 0 of the env wave).  The synthetic env does not depend on
  // the action, so its step runs on wave 1 concurrently with wave 0's head + sampling; CartPole needs the action
  // and stays on wave 0.
  constexpr int ENV_WAVE = (ENV == ORL_ENV_SYNTH) ? 1 : 0;
  constexpr bool PERK = ENV == ORL_ENV_TTT_POOLK;                         // per-env opponents (see above)
  constexpr bool IS_POOL = ENV == ORL_ENV_TTT_POOL || PERK;
  constexpr bool IS_TTT = ENV == ORL_ENV_TTT || IS_POOL;
  constexpr int ESW = IS_TTT ? TTT_STATE_W : 8;
  float est[ESW] = {};
  const int SW = (ENV == ORL_ENV_SYNTH) ? SYNTH_STATE_W : IS_TTT ? TTT_STATE_W : CARTPOLE_STATE_W;
  float ep_ret = 0.f, ep_len = 0.f, fin_ret = 0.f, fin_cnt = 0.f;
  // tic-tac-toe: the legal-move mask of the current step lives in LDS (written with the board one step earlier)
  float* s_mask = s_logits + TILE_B * 16;  // [16][16]
  // self-play (ORL_ENV_TTT_POOL): the opponent's tower image, the opponent-side boards and masks of the current step
  float* s_opp = s_mask + TILE_B * 16;                         // TowerLds image of this tile's opponent policy
  const int n_img = PERK ? A.r.opp_n_policies : (IS_POOL ? 1 : 0);
  float* s_oobs = s_opp + n_img * twp.total;                   // [16][DP] (unused: the opponent's fc1 reads s_oword)
  float* s_omask = s_oobs + TILE_B * DP;                       // [16][16]
  float* s_oppk = s_omask + TILE_B * 16;                       // [16] PERK: slot of the row's opponent this step, -1 = none
  int my_opp = 0;                                              // PERK: this env's pool slot (owning lanes)
  if constexpr (PERK) {
    for (int k = 0; k < n_img; ++k)
      stage_tower(s_opp + k * twp.total, A.r.opp_thetas + (size_t)k * A.r.opp_theta_stride, tlp, twp, false, threadIdx.x,
                  blockDim.x, HMM, false, true);
    my_opp = ok ? A.r.opp_index[n] : 0;
  } else if constexpr (IS_POOL) {
    const int first = blockIdx.x * TILE_B;  // opp_index: uniform over the tile (orl_opponent_sample per_tile)
    const int og = A.r.opp_index != nullptr ? A.r.opp_index[first < N ? first : N - 1] : first / A.r.opp_group_rows;
    const float* oth = A.r.opp_thetas + (size_t)og * A.r.opp_theta_stride;
    stage_tower(s_opp, oth, tlp, twp, false, threadIdx.x, blockDim.x, HMM, false, true);
  }
  if (IS_TTT && wave == 0 && q == 0) {
    for (int c = 0; c < 16; ++c) s_mask[j * 16 + c] = (ok && c < 9) ? b.action_masks[(size_t)n * 9 + c] : (c == 0 ? 1.f : 0.f);
  }
  if (wave == ENV_WAVE && q == 0 && ok) {
#pragma unroll
    for (int k = 0; k < ESW; ++k) est[k] = (k < SW) ? A.r.env_state[(size_t)n * SW + k] : 0.f;
    ep_ret = A.r.ep_stats[n * 4 + 0]; ep_len = A.r.ep_stats[n * 4 + 1];
    fin_ret = A.r.ep_stats[n * 4 + 2]; fin_cnt = A.r.ep_stats[n * 4 + 3];
  }
  // tic-tac-toe: the game state of the owning lanes as bitboards (cell c = bit c per player; env_state keeps its
  // float[12] rows in HBM), and one word per row - agent bits | opponent bits << 9 | valid << 18 - in LDS, from which every
  // consumer expands what it needs: fc1 its operands, wave 0 the legal-move mask, all threads the rows of the rollout
  // buffer.  The logits tile is free here (fragment-sampled categorical head).
  int* s_word = (int*)s_logits;  // [16] the learner's view; [16] more behind it: the opponent's view (self-play)
  int* s_oword = s_word + TILE_B;
  int* s_meta = s_oword + TILE_B;                      // [2][16] opponent moves this game, episode (for the draws)
  uint32_t* s_draw = (uint32_t*)(s_meta + 2 * TILE_B);  // [4][16] the step's Philox words, from wave 1
  int tA = 0, tO = 0, tmoves = 0, tep = 0;
  if (IS_TTT && wave == 0 && q == 0) {
    if (ok) {
#pragma unroll
      for (int c = 0; c < 9; ++c) {
        tA |= (est[c] == 1.f ? 1 : 0) << c;
        tO |= (est[c] == 2.f ? 1 : 0) << c;
      }
      tmoves = (int)est[9]; tep = (int)est[10];
    }
    s_word[j] = ok ? (tA | (tO << 9) | (1 << 18)) : 0;  // = slot 0's observation row (the env keeps them in step)
    s_meta[j] = tmoves; s_meta[TILE_B + j] = tep;
  }
  // the loads above must have landed before the step loop: a pending load into a loop-carried register makes the
  // compiler wait for ALL but the newest stores of the wave at its first use in every iteration (s_waitcnt vmcnt(2)
  // in the middle of the env step)
  asm volatile("" : "+v"(ep_ret), "+v"(ep_len), "+v"(fin_ret), "+v"(fin_cnt));
#pragma unroll
  for (int k = 0; k < ESW; ++k) asm volatile("" : "+v"(est[k]));
  // Observations never exist as an LDS tile here: lane (j, q)'s fc1 operands of row j - features 4 s + q = cell
  // (4 s + q) >> 1 of player q & 1 - are bits 2 s of ttt_operand_bits(word, q).
  auto ttt_operand_bits = [&](int w) -> int {
    return (w >> ((q & 1) * 9 + (q >> 1))) & (q < 2 ? 0x155 : 0x55);
  };
  // classes 4 q .. 4 q + 3 of row j's legal-move mask; an invalid row has class 0 legal
  auto ttt_mask_to_lds = [&](int w, float* mask_tile) {
    const int emp = (((w >> 18) & 1) ? (~(w | (w >> 9)) & 0x1FF) : 1) >> (4 * q);
    *(f32x4*)(mask_tile + j * 16 + 4 * q) = f32x4{(float)(emp & 1), (float)((emp >> 1) & 1), (float)((emp >> 2) & 1),
                                                  (float)((emp >> 3) & 1)};
  };
  // slot t + 1 of the [T+1, N, 18] observation and [T+1, N, 9] mask arrays: coalesced rows expanded from the tile's board
  // words by ALL threads of the workgroup (one pass of 512), after the step's last barrier - off wave 0's serial chain
  auto ttt_rows_to_buffer = [&](const int* words, int t) {
    const int row0 = blockIdx.x * TILE_B;
    const unsigned nrow = (unsigned)((N - row0) < TILE_B ? (N - row0) : TILE_B);
    const size_t base = (size_t)(t + 1) * N + row0;
    for (unsigned e = threadIdx.x; e < nrow * 27u; e += blockDim.x) {
      if (e < nrow * 18u) {
        const unsigned rr = e / 18u, d = e - rr * 18u;
        const float v = (float)((words[rr] >> ((d & 1u) * 9u + (d >> 1))) & 1);
        b.policy_obs[base * 18 + e] = v;
        if (sep_c) b.critic_obs[base * 18 + e] = v;
      } else {
        const unsigned e2 = e - nrow * 18u, rr = e2 / 9u, c = e2 - rr * 9u;
        const int w = words[rr];
        b.action_masks[base * 9 + e2] = (float)((~(w | (w >> 9)) >> c) & 1);
      }
    }
  };
  // The sampling noise does not depend on the network output: wave 3 (idle while wave 0 runs the head) draws the
  // NEXT step's uniforms / normals into LDS with the same Philox counters sample_head would use, and wave 0 consumes
  // them through sample_head's forced-noise argument - bit-identical actions, Philox off the critical path.
  constexpr int NOISE_WAVE = 3, OBS_WAVE = (ENV == ORL_ENV_SYNTH) ? 2 : 0;
  auto draw_noise = [&](uint64_t tg, float* dst) {
    if (HEAD == ORL_HEAD_CATEGORICAL) {
      const u4 r = philox4x32_10(A.r.act_seed, (uint32_t)n, (uint32_t)((uint64_t)n >> 32), (uint32_t)tg,
                                 (uint32_t)(tg >> 32) << 8);
      if (q == 0) dst[j * 16] = u01(r.x);
    } else if (4 * q < n_out) {
      const u4 r = philox4x32_10(A.r.act_seed, (uint32_t)n, (uint32_t)((uint64_t)n >> 32), (uint32_t)tg,
                                 ((uint32_t)(tg >> 32) << 8) | (uint32_t)q);
      float e[4];
      box_muller(r.x, r.y, e[0], e[1]);
      box_muller(r.z, r.w, e[2], e[3]);
      *(f32x4*)(dst + j * 16 + 4 * q) = f32x4{e[0], e[1], e[2], e[3]};
    }
  };
  if (wave == NOISE_WAVE) draw_noise(A.r.rng_step0, s_noise);
  __syncthreads();
  CoopRegs creg;  // this wave's loop-invariant trunk operands, out of LDS once
  coop_load(tlds, tww, gw, j, q, creg);
  // ... and the head's: the policy head on wave 0 (narrow heads: W3 rows; wide heads: the 16 x 64 MFMA image's fragments and
  // the bias quad), the value head on wave 4
  HeadRegs<HMM ? 1 : NO> hreg_p;
  HeadRegs<1> hreg_c;
  f32x4 hfrag[4], hbias = f32x4{0.f, 0.f, 0.f, 0.f};
  if constexpr (HMM) {
    const int no4 = (A.pnet.n_out + 3) & ~3;
    if (4 * q < no4) hbias = *(const f32x4*)(smem + twp.b3 + 4 * q);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) hfrag[mi] = *(const f32x4*)(smem + twp.W3P + j * W2S + 16 * mi + 4 * q);
  } else {
    head_regs_load<NO>(smem + twp.W3, smem + twp.b3, A.pnet.n_out, q, hreg_p);
  }
  if (WC) head_regs_load<1>(smem + twp.total + twc.W3, smem + twp.total + twc.b3, 1, q, hreg_c);
  RollProf rp;
#ifdef ORL_PROF
  __shared__ unsigned long long prof_lds[16];
  if (threadIdx.x < 16) prof_lds[threadIdx.x] = 0ull;
  __syncthreads();
  rp.lds = prof_lds; rp.on = blockIdx.x == 0 && wave < 2; rp.base = 8 * wave;
  rp.t_last = __builtin_readcyclecounter();
#endif

  for (int t = 0; t < T; ++t) {
    const float* cur = s_obs + (t & 1) * TILE_B * DP;
    float* nxt = s_obs + ((t + 1) & 1) * TILE_B * DP;
    const int wq = IS_TTT ? ttt_operand_bits(s_word[j]) : 0;
    auto xb = [&](int s) -> float {
      if constexpr (IS_TTT) return (float)((wq >> (2 * s)) & 1);
      else return cur[j * DP + 4 * s + q];
    };
    const uint64_t tg = A.r.rng_step0 + (uint64_t)t;
    f32x4 n2[4];
    trunk_fwd_coop(tlds, tww, creg, xb, gA, gB, gw, j, q, n2, rp);
    float act_o[NO];
    if constexpr (HMM && HEAD == ORL_HEAD_GAUSSIAN) {
      // Wide Gaussian heads: the MFMA leaves means 4q..4q+3 of row j in lane (j, q) - each lane samples exactly those
      // dimensions (same expressions as sample_head) and stores them, instead of every lane of a row walking all NO
      // dimensions after a trip of the tile through LDS: this phase was 3 300 of the step's 7 900 cycles at Box(6).
      if (wave == 0) {
        f32x4 mu = hbias;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
          for (int r = 0; r < 4; ++r) mu = ORL_MFMA(hfrag[mi][r], n2[mi][r], mu);
        }
        const float* noise = s_noise + (t & 1) * TILE_B * 16 + j * 16;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = 4 * q + r;
          if (ok && c < n_out) {
            const float ls = smem[twp.logstd + c];
            const float sd = expf(ls);
            const float av = mu[r] + sd * noise[c];
            const float d = av - mu[r];
            const float lp = -(d * d) / (2.f * (sd * sd)) - ls - 0.91893853320467274178f;
            A.r.actions[((size_t)t * N + n) * a_w + c] = av;
            A.r.action_log_probs[((size_t)t * N + n) * a_w + c] = lp;
          }
        }
      }
    } else if constexpr (HMM && HEAD == ORL_HEAD_CATEGORICAL) {
      // Wide categorical heads: sampled on the MFMA fragment (sample_cat_frag) - no trip of the logits tile through
      // LDS, 4 classes per lane instead of NO per lane: this phase was 4 200 of the step's 8 900 cycles at Discrete(9).
      if (wave == 0) {
        f32x4 lgv = hbias;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
          for (int r = 0; r < 4; ++r) lgv = ORL_MFMA(hfrag[mi][r], n2[mi][r], lgv);
        }
        float av, lp;
        sample_cat_frag(lgv, n_out, q, IS_TTT ? s_mask + j * 16 : nullptr, s_noise[(t & 1) * TILE_B * 16 + j * 16], av, lp);
        act_o[0] = av;
        if (ok && q == 0) {
          A.r.actions[(size_t)t * N + n] = av;
          A.r.action_log_probs[(size_t)t * N + n] = lp;
        }
      }
    } else if (wave == 0) {
      float hd[NO], lp_o[NO];
      if constexpr (HMM) head_mfma_T<NO>(smem + twp.W3P, smem + twp.b3, n_out, n2, s_logits, j, q, hd);
      else head_T_regs<NO>(hreg_p, n2, hd);
      // the built-in device envs never mask actions (their action_masks rows stay all-ones), so the mask is
      // not re-read from HBM on the per-step critical path
      sample_head<NO, HEAD>(hd, n_out, smem + twp.logstd, IS_TTT ? s_mask + j * 16 : nullptr,
                            s_noise + (t & 1) * TILE_B * 16 + j * 16, 0, A.r.act_seed, (uint64_t)n, tg, act_o, lp_o);
      if (ok && q == 0) {
#pragma unroll
        for (int c = 0; c < NO; ++c) {
          if (c < a_w) {
            A.r.actions[((size_t)t * N + n) * a_w + c] = act_o[c];
            A.r.action_log_probs[((size_t)t * N + n) * a_w + c] = lp_o[c];
          }
        }
      }
    }
    if constexpr (IS_POOL) {
      // The step's Philox draws, one pass over the four lanes of a row ON WAVE 1 (idle while wave 0 runs the head; they
      // were up to four passes on wave 0 behind divergent branches), handed over by phase A's barrier: q = 0 the
      // opponent's uniform (its own stream, sample_head's counter layout), 1 = who opens the next game, 2 = that game's
      // opening move, 3 (PERK) = the next game's opponent (opponent_sample_kernel's draw).
      if (wave == 1) {
        const int ep = s_meta[TILE_B + j];
        const uint64_t og = A.r.opp_rng_step0 + (uint64_t)t, id = A.r.opp_draw_id0 + (uint64_t)t;
        const uint64_t key = q == 0 ? A.r.opp_seed : (q == 3 ? A.r.opp_sample_seed : A.r.env_seed);
        const uint32_t c1 = q == 0 ? (uint32_t)((uint64_t)n >> 32) : q == 1 ? 0x77C7FFFFu : q == 2 ? 0x77C70000u : 0x0FF05A3Fu;
        const uint32_t c2 = q == 0 ? (uint32_t)og : q == 3 ? (uint32_t)id : (uint32_t)(ep + 1);
        const uint32_t c3 = q == 0 ? (uint32_t)(og >> 32) << 8 : q == 3 ? (uint32_t)(id >> 32) : 0u;
        s_draw[q * TILE_B + j] = philox4x32_10(key, (uint32_t)n, c1, c2, c3).x;
      }
      // ---- phase A (orl_ttt_agent_move): the agent's move on the owning lanes of wave 0; boards from the opponent's
      // side go to LDS for the opponent's policy
      int ph = 2;
      float rew_a = 0.f;
      if (wave == 0) {
        if (q == 0) {
          if (ok) {
            const int a = (int)act_o[0];
            const int e0 = ~(tA | tO) & 0x1FF;
            const bool legal = (unsigned)a < 9u && ((e0 >> a) & 1) != 0;
            ph = 1;
            if (!legal) { rew_a = -1.f; ph = 2; }
            else {
              tA |= 1 << a;
              if (ttt_wins_bits(tA)) { rew_a = 1.f; ph = 2; }
              else if ((tA | tO) == 0x1FF) ph = 2;
            }
          }
          s_oword[j] = ph == 1 ? (tO | (tA << 9) | (1 << 18)) : 0;  // the board as the opponent sees it
          if (PERK) s_oppk[j] = (ok && ph == 1) ? (float)my_opp : -1.f;
        }
        wave_lds_fence();
        ttt_mask_to_lds(s_oword[j], s_omask);
      }
      __syncthreads();
      const int wqo = ttt_operand_bits(s_oword[j]);
      float ou = 0.f;
      uint32_t r_begin = 0u, r_open = 0u, r_samp = 0u;
      if (wave == 0) {
        ou = u01(s_draw[j]);
        r_begin = s_draw[TILE_B + j]; r_open = s_draw[2 * TILE_B + j]; r_samp = s_draw[3 * TILE_B + j];
      }
      // ---- the opponent's policy: cooperative trunk of the policy group on the opponent-side boards (the critic
      // group only keeps the barrier count)
      // One walk of the opponent tower per image: PERK walks image k only if some row of the tile plays slot k this
      // step (workgroup-uniform test), rows keep the action sampled from their own slot's logits.
      float oact[NO], olp = 0.f;
      oact[0] = 0.f;
      for (int k = 0; k < (PERK ? n_img : 1); ++k) {
        const float* img = s_opp + k * twp.total;
        if (PERK) {
          if (!__syncthreads_or(s_oppk[j] == (float)k)) continue;
        }
        f32x4 o2[4];
        if (grp == 0) {
          auto xo = [&](int s) -> float { return (float)((wqo >> (2 * s)) & 1); };
          CoopRegs oreg;  // the opponent's image (a pool slot): read per walk
          coop_load(img, twp, gw, j, q, oreg);
          trunk_fwd_coop(img, twp, oreg, xo, gA, gB, gw, j, q, o2, rp);
        } else {
          for (int nb = 0; nb < trunk_barriers(twp.DP); ++nb) __syncthreads();
        }
        if (wave == 0) {  // sampled on the MFMA fragment like the learner's action
          const int no4 = (n_out + 3) & ~3;
          f32x4 lgv = f32x4{0.f, 0.f, 0.f, 0.f};
          if (4 * q < no4) lgv = *(const f32x4*)(img + twp.b3 + 4 * q);
#pragma unroll
          for (int mi = 0; mi < 4; ++mi) {
            const f32x4 a4 = *(const f32x4*)(img + twp.W3P + j * W2S + 16 * mi + 4 * q);
#pragma unroll
            for (int r = 0; r < 4; ++r) lgv = ORL_MFMA(a4[r], o2[mi][r], lgv);
          }
          float ak, lk;
          sample_cat_frag(lgv, n_out, q, s_omask + j * 16, ou, ak, lk);
          if (!PERK || s_oppk[j] == (float)k) { oact[0] = ak; olp = lk; }
        }
      }
      if (wave == 0) {
        // ---- phase B (orl_ttt_opponent_move) ----
        if (q == 0) {
          const size_t s1 = (size_t)(t + 1) * N + n;
          if (ok) {
            float rew = rew_a;
            bool done = ph == 2;
            if (ph == 1) {
              const int e0 = ~(tA | tO) & 0x1FF;
              const int a = (int)oact[0];
              const bool legal = (unsigned)a < 9u && ((e0 >> a) & 1) != 0;
              tO |= legal ? (1 << a) : (e0 & -e0);  // an illegal sample falls back to the lowest empty cell
              ++tmoves;
              if (ttt_wins_bits(tO)) { rew = -1.f; done = true; }
              else if ((tA | tO) == 0x1FF) done = true;
            }
            b.rewards[(size_t)t * N + n] = rew;
            b.masks[s1] = done ? 0.f : 1.f;
            b.active_masks[s1] = 1.f;
            b.bad_masks[s1] = 1.f;
            ep_ret += rew; ep_len += 1.f;
            if (done) {
              fin_ret += ep_ret; fin_cnt += 1.f; ep_ret = 0.f; ep_len = 0.f;
              ++tep;
              tA = 0; tO = 0; tmoves = 0;
              if (r_begin & 1u) { tO = ttt_pick_empty_bits(0x1FF, r_open); tmoves = 1; }
              if (PERK) {  // a finished game -> a fresh opponent for the next one (opponent_sample_kernel's draw)
                int k = A.r.opp_last_slot;
                if (A.r.opp_strategy == 0) {
                  k = (int)(u01(r_samp) * (float)A.r.opp_n_filled);
                  k = k < A.r.opp_n_filled - 1 ? k : A.r.opp_n_filled - 1;
                }
                my_opp = k < 0 ? 0 : k;
              }
            }
          }
          s_word[j] = ok ? (tA | (tO << 9) | (1 << 18)) : 0;
          s_meta[j] = tmoves; s_meta[TILE_B + j] = tep;
        }
        wave_lds_fence();
        ttt_mask_to_lds(s_word[j], s_mask);
      }
    }
    if constexpr (ENV == ORL_ENV_TTT) {
      // the step's three Philox draws (ttt_draw: q = 0 the opponent's reply, 1 = who opens the next game, 2 = that game's
      // opening move) in one pass over the lanes of wave 1 while wave 0 runs the head - they were three passes on wave 0,
      // behind divergent branches; one more barrier hands them over
      if (wave == 1 && q < 3)
        s_draw[q * TILE_B + j] = ttt_draw(A.r.env_seed, (uint32_t)n, q, (uint32_t)(s_meta[TILE_B + j] + (q != 0 ? 1 : 0)),
                                          (uint32_t)s_meta[j]);
      __syncthreads();
    }
    RO_T(rp, 5);
    if (wave == NOISE_WAVE) draw_noise(tg + 1, s_noise + ((t + 1) & 1) * TILE_B * 16);
    if (ENV == ORL_ENV_SYNTH && wave == OBS_WAVE) {
      // synthetic observations of slot t+1: every lane of the row generates obs blocks b = q, q+4, ...
      const size_t s1 = (size_t)(t + 1) * N + n;
      for (int bb = q; bb < (D + 3) / 4; bb += 4) {
        float o[4];
        synth_obs_block(A.r.env_seed, (uint32_t)n, tg + 1, (uint32_t)bb, o);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int d = 4 * bb + k;
          if (d < DP) nxt[j * DP + d] = (d < D) ? o[k] : 0.f;
          if (ok && d < D) {
            b.policy_obs[s1 * D + d] = o[k];
            if (sep_c) b.critic_obs[s1 * D + d] = o[k];
          }
        }
      }
    }
    if (wave == ENV_WAVE) {
      // ---- env.step + insert (slot t+1) ----
      const size_t s1 = (size_t)(t + 1) * N + n;
      if (ENV == ORL_ENV_SYNTH) {
        bool done = false;
        if (q == 0 && ok) {
          const float rew = synth_reward(A.r.env_seed, (uint32_t)n, tg);
          est[0] += 1.f;
          done = est[0] >= (float)A.r.episode_limit;
          if (done) est[0] = 0.f;
          b.rewards[(size_t)t * N + n] = rew;
          b.masks[s1] = done ? 0.f : 1.f;
          b.active_masks[s1] = 1.f;
          b.bad_masks[s1] = 1.f;
          ep_ret += rew; ep_len += 1.f;
          if (done) { fin_ret += ep_ret; fin_cnt += 1.f; ep_ret = 0.f; ep_len = 0.f; }
        }
      } else if (ENV == ORL_ENV_TTT) {
        // orl_ttt_step's game logic on the owning lane, on bitboards; the step's draws come from wave 1 (above), the new
        // board leaves as one word per row.
        const uint32_t rx = s_draw[j], r_begin = s_draw[TILE_B + j], r_open = s_draw[2 * TILE_B + j];
        if (q == 0) {
          if (ok) {
            const int a = (int)act_o[0];
            const int e0 = ~(tA | tO) & 0x1FF;
            float rew = 0.f;
            bool done = false;
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
            b.rewards[(size_t)t * N + n] = rew;
            b.masks[s1] = done ? 0.f : 1.f;
            b.active_masks[s1] = 1.f;
            b.bad_masks[s1] = 1.f;
            ep_ret += rew; ep_len += 1.f;
            if (done) {
              fin_ret += ep_ret; fin_cnt += 1.f; ep_ret = 0.f; ep_len = 0.f;
              ++tep;
              tA = 0; tO = 0; tmoves = 0;
              if (r_begin & 1u) { tO = ttt_pick_empty_bits(0x1FF, r_open); tmoves = 1; }
            }
          }
          s_word[j] = ok ? (tA | (tO << 9) | (1 << 18)) : 0;
          s_meta[j] = tmoves; s_meta[TILE_B + j] = tep;
        }
        wave_lds_fence();
        ttt_mask_to_lds(s_word[j], s_mask);
      } else if (ENV == ORL_ENV_CARTPOLE) {
        if (q == 0 && ok) {
          float s[4] = {est[0], est[1], est[2], est[3]};
          const int action = (int)act_o[0];
          const bool term = cartpole_step(s, action);
          est[4] += 1.f;
          const bool trunc = est[4] >= (float)A.r.episode_limit;
          const bool done = term || trunc;
          const float rew = 1.0f;
          b.rewards[(size_t)t * N + n] = rew;
          b.masks[s1] = done ? 0.f : 1.f;
          b.active_masks[s1] = 1.f;
          b.bad_masks[s1] = 1.f;
          ep_ret += rew; ep_len += 1.f;
          if (done) {
            fin_ret += ep_ret; fin_cnt += 1.f; ep_ret = 0.f; ep_len = 0.f;
            est[5] += 1.f;
            est[4] = 0.f;
            cartpole_reset(A.r.env_seed, (uint32_t)n, (uint32_t)est[5], s);  // auto-reset: obs = first obs of new episode
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            est[k] = s[k];
            nxt[j * DP + k] = s[k];
            b.policy_obs[s1 * 4 + k] = s[k];
            if (sep_c) b.critic_obs[s1 * 4 + k] = s[k];
          }
        } else if (q == 0) {
#pragma unroll
          for (int k = 0; k < 4; ++k) nxt[j * DP + k] = 0.f;
        }
      }
      if (!IS_TTT && b.action_masks != nullptr && ok && q == 0) {
        for (int c = 0; c < b.K; ++c) b.action_masks[s1 * b.K + c] = 1.f;
      }
    }
    if (WC && wave == 4) {
      float v[1];
      head_T_regs<1>(hreg_c, n2, v);
      if (ok && q == 0) A.r.value_preds[(size_t)t * N + n] = v[0];
    }
    RO_T(rp, 6);
    __syncthreads();
    RO_T(rp, 7);
    if constexpr (IS_TTT) ttt_rows_to_buffer(s_word, t);
  }
#ifdef ORL_PROF
  __syncthreads();
  if (blockIdx.x == 0 && threadIdx.x < 16) atomicAdd(&g_roll_prof[threadIdx.x], prof_lds[threadIdx.x]);
#endif
  // bootstrap value of the last observation (OnPolicyDriver.compute_returns, onpolicy_driver.py:205-233);
  // all waves walk the cooperative trunk once more (barriers), only the critic leader uses the result
  if (WC && A.next_value != nullptr) {
    const float* cur = s_obs + (T & 1) * TILE_B * DP;
    const int wq = IS_TTT ? ttt_operand_bits(s_word[j]) : 0;
    auto xb = [&](int s) -> float {
      if constexpr (IS_TTT) return (float)((wq >> (2 * s)) & 1);
      else return cur[j * DP + 4 * s + q];
    };
    f32x4 n2[4];
    trunk_fwd_coop(tlds, tww, creg, xb, gA, gB, gw, j, q, n2, rp);
    if (wave == 4) {
      float v[1];
      head_T_regs<1>(hreg_c, n2, v);
      if (ok && q == 0) A.next_value[n] = v[0];
    }
  }
  if (wave == ENV_WAVE && q == 0 && ok) {
    if constexpr (IS_TTT) {
#pragma unroll
      for (int c = 0; c < 9; ++c) est[c] = ((tA >> c) & 1) ? 1.f : (((tO >> c) & 1) ? 2.f : 0.f);
      est[9] = (float)tmoves; est[10] = (float)tep;
    }
#pragma unroll
    for (int k = 0; k < ESW; ++k)
      if (k < SW) A.r.env_state[(size_t)n * SW + k] = est[k];
    A.r.ep_stats[n * 4 + 0] = ep_ret; A.r.ep_stats[n * 4 + 1] = ep_len;
    A.r.ep_stats[n * 4 + 2] = fin_ret; A.r.ep_stats[n * 4 + 3] = fin_cnt;
    if (PERK) const_cast<int32_t*>(A.r.opp_index)[n] = my_opp;  // the slots the next rollout / stepwise step starts from
  }
}

// V(obs) for a large batch of stored observations (all T+1 slots of a rollout in one launch): persistent workgroups
// of 8 waves, the critic tower staged once per workgroup, one 16-row tile per wave and trip, the next tile's
// observation fragments prefetched into registers.  Per-row arithmetic = act_step_kernel's critic wave (trunk_fwd_T).
__global__ __launch_bounds__(512) void critic_values_kernel(orl_net_desc cnet, const float* __restrict__ ctheta,
                                                            const float* __restrict__ obs, long long rows,
                                                            float* __restrict__ values, float* __restrict__ tail_out,
                                                            long long tail_row0) {
  // tail_out (optional): rows >= tail_row0 are ALSO written to tail_out[row - tail_row0] - the bootstrap value of a rollout's
  // slot T lands in the driver's next_value array without a copy launch
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const TowerLayout tl(cnet);
  const TowerLds tw(cnet.obs_dim, 1, false, false);
  stage_tower(smem, ctheta, tl, tw, false, threadIdx.x, blockDim.x);
  __syncthreads();
  const int wave = threadIdx.x >> 6, l = threadIdx.x & 63, j = l & 15, q = l >> 4;
  const int D = cnet.obs_dim, nk = tw.DP >> 2, nwv = blockDim.x >> 6;
  const long long n_tiles = (rows + TILE_B - 1) / TILE_B;
  const long long stride = (long long)gridDim.x * nwv;
  float xr[16];
  auto fetch = [&](long long tile) {
    const long long row = tile * TILE_B + j;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int k = 4 * s + q;
      xr[s] = (s < nk && tile < n_tiles && row < rows && k < D) ? obs[row * D + k] : 0.f;
    }
  };
  long long tile = (long long)blockIdx.x * nwv + wave;
  fetch(tile);
  for (; tile < n_tiles; tile += stride) {
    float xc[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) xc[s] = xr[s];
    fetch(tile + stride);
    auto xb = [&](int s) -> float {
      float v = xc[0];
#pragma unroll
      for (int t = 1; t < 16; ++t) v = (s == t) ? xc[t] : v;
      return v;
    };
    f32x4 n2[4];
    trunk_fwd_T(smem, tw, xb, j, q, n2);
    float v[1];
    head_T<1>(smem + tw.W3, smem + tw.b3, 1, n2, q, v);
    const long long row = tile * TILE_B + j;
    if (row < rows && q == 0) {
      values[row] = v[0];
      if (tail_out != nullptr && row >= tail_row0) tail_out[row - tail_row0] = v[0];
    }
  }
}

#include "orl_rollout2.h"

// Stand-alone env.step for device envs (evaluation loops, the stepwise driver): same dynamics and RNG
// streams as the fused rollout; one thread per env.
template <int ENV>
__global__ __launch_bounds__(256) void env_step_kernel(float* __restrict__ env_state, float* __restrict__ ep_stats,
                                                       const float* __restrict__ actions, int a_w,
                                                       float* __restrict__ obs, float* __restrict__ rew,
                                                       uint8_t* __restrict__ done, int N, int D, uint64_t seed,
                                                       int episode_limit, uint64_t tg,
                                                       const long long* __restrict__ tg_dev) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  if (tg_dev) tg += (uint64_t)*tg_dev;  // hipGraph replays: the host part is frozen, the device part advances
  float r;
  bool d;
  env_step_one<ENV>(env_state, ep_stats, n, D, seed, episode_limit, tg, ENV == ORL_ENV_SYNTH ? 0 : (int)actions[(size_t)n * a_w],
                    obs + (size_t)n * D, r, d);
  rew[n] = r;
  done[n] = d ? 1 : 0;
}

// --------------------------------------------------------------------------------------------------
// evaluate_actions (forward only): log-prob of GIVEN actions, per-row entropy, values.
// --------------------------------------------------------------------------------------------------
struct EvalArgs {
  orl_net_desc pnet, cnet;
  const float* ptheta;
  const float* ctheta;
  const float* pobs;
  const float* cobs;
  const float* actions;  // [B, a]
  const float* amask;    // [B, n_out] or NULL
  const float* active;   // [B] or NULL
  float* values;         // [B]
  float* logp;           // [B, a]
  float* ent_w;          // [B]: entropy(row) * weight(row)
  int B;
};

template <int NO, int HEAD>
__global__ __launch_bounds__(128) void eval_actions_kernel(EvalArgs A) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const TowerLayout tlp(A.pnet);
  const TowerLds twp(A.pnet.obs_dim, A.pnet.n_out, HEAD == ORL_HEAD_GAUSSIAN, false);
  TowerLds twc;
  const bool has_c = A.ctheta != nullptr;
  stage_tower(smem, A.ptheta, tlp, twp, false, threadIdx.x, blockDim.x);
  if (has_c) {
    const TowerLayout tlc(A.cnet);
    twc = TowerLds(A.cnet.obs_dim, 1, false, false);
    stage_tower(smem + twp.total, A.ctheta, tlc, twc, false, threadIdx.x, blockDim.x);
  }
  __syncthreads();
  const int wave = threadIdx.x >> 6, l = threadIdx.x & 63, j = l & 15, q = l >> 4;
  const int row = blockIdx.x * TILE_B + j;
  const bool ok = row < A.B;
  if (wave == 0) {
    const int D = A.pnet.obs_dim, n_out = A.pnet.n_out;
    const float* xrow = A.pobs + (size_t)row * D;
    auto xb = [&](int s) -> float {
      const int k = 4 * s + q;
      return (ok && k < D) ? xrow[k] : 0.f;
    };
    f32x4 n2[4];
    trunk_fwd_T(smem, twp, xb, j, q, n2);
    float hd[NO];
    head_T<NO>(smem + twp.W3, smem + twp.b3, n_out, n2, q, hd);
    const float w = (A.active != nullptr && ok) ? A.active[row] : 1.f;
    if (HEAD == ORL_HEAD_CATEGORICAL) {
      const float* am = (A.amask != nullptr && ok) ? A.amask + (size_t)row * n_out : nullptr;
      const float lse = cat_lse<NO>(hd, n_out, am);
      const int act = ok ? (int)A.actions[row] : 0;
      float ent = 0.f;
#pragma unroll
      for (int c = 0; c < NO; ++c) {
        if (c < n_out) {
          const float ell = hd[c] - lse;
          ent -= expf(ell) * ell;
        }
      }
      if (ok && q == 0) {
        A.logp[row] = pick<NO>(hd, act) - lse;
        A.ent_w[row] = ent * w;
      }
    } else {
      float ent = 0.f;
#pragma unroll
      for (int c = 0; c < NO; ++c) {
        if (c < n_out) {
          const float ls = smem[twp.logstd + c];
          const float sd = expf(ls);
          const float d = (ok ? A.actions[(size_t)row * n_out + c] : 0.f) - hd[c];
          if (ok && q == 0) A.logp[(size_t)row * n_out + c] = -(d * d) / (2.f * (sd * sd)) - ls - 0.91893853320467274178f;
          ent += 1.41893853320467274178f + ls;
        }
      }
      if (ok && q == 0) A.ent_w[row] = ent * w;
    }
  } else if (has_c) {
    const int D = A.cnet.obs_dim;
    const float* xrow = A.cobs + (size_t)row * D;
    auto xb = [&](int s) -> float {
      const int k = 4 * s + q;
      return (ok && k < D) ? xrow[k] : 0.f;
    };
    f32x4 n2[4];
    const float* lc = smem + twp.total;
    trunk_fwd_T(lc, twc, xb, j, q, n2);
    float v[1];
    head_T<1>(lc + twc.W3, lc + twc.b3, 1, n2, q, v);
    if (ok && q == 0) A.values[row] = v[0];
  }
}

// dist_entropy = sum(ent_w) / (sum(active) | B | B*a)   (act.py:151-168)
__global__ __launch_bounds__(256) void eval_entropy_kernel(const float* __restrict__ ent_w,
                                                           const float* __restrict__ active, int B, float den_scale,
                                                           float* __restrict__ out) {
  __shared__ double sh[4][2];
  double s = 0, d = 0;
  for (int i = threadIdx.x; i < B; i += blockDim.x) {
    s += (double)ent_w[i];
    d += active != nullptr ? (double)active[i] : 1.0;
  }
  s = wave_sum(s);
  d = wave_sum(d);
  if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6][0] = s; sh[threadIdx.x >> 6][1] = d; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double ss = 0, dd = 0;
    for (int k = 0; k < 4; ++k) { ss += sh[k][0]; dd += sh[k][1]; }
    out[0] = (float)(ss / (dd * (double)den_scale));
  }
}

static int check_net(const orl_net_desc* n, const char* who, bool is_critic) {
  if (!n) return fail(ORL_E_INVALID, "%s: null net descriptor", who);
  if (n->hidden != HID) return fail(ORL_E_UNSUPPORTED, "%s: hidden_size %d not built (only 64)", who, n->hidden);
  if (n->obs_dim < 1 || n->obs_dim > 256) return fail(ORL_E_INVALID, "%s: obs_dim %d outside [1,256]", who, n->obs_dim);
  if (is_critic) {
    if (n->head_kind != ORL_HEAD_VALUE || n->n_out != 1) return fail(ORL_E_INVALID, "%s: critic must be a value head", who);
  } else {
    if (n->head_kind != ORL_HEAD_CATEGORICAL && n->head_kind != ORL_HEAD_GAUSSIAN)
      return fail(ORL_E_UNSUPPORTED, "%s: head kind %d not built (Discrete / Box only)", who, n->head_kind);
    if (n->n_out < 1 || n->n_out > 16) return fail(ORL_E_UNSUPPORTED, "%s: n_out %d outside [1,16]", who, n->n_out);
  }
  return 0;
}

}  // namespace orl

using namespace orl;

#define ORL_DISPATCH_HEAD(KERNEL_MACRO)                                        \
  do {                                                                         \
    const int no = pnet->n_out;                                                \
    if (pnet->head_kind == ORL_HEAD_CATEGORICAL) {                             \
      if (no <= 2) KERNEL_MACRO(2, ORL_HEAD_CATEGORICAL);                      \
      else if (no <= 8) KERNEL_MACRO(8, ORL_HEAD_CATEGORICAL);                 \
      else KERNEL_MACRO(16, ORL_HEAD_CATEGORICAL);                             \
    } else {                                                                   \
      if (no <= 2) KERNEL_MACRO(2, ORL_HEAD_GAUSSIAN);                         \
      else if (no <= 8) KERNEL_MACRO(8, ORL_HEAD_GAUSSIAN);                    \
      else KERNEL_MACRO(16, ORL_HEAD_GAUSSIAN);                                \
    }                                                                          \
  } while (0)

extern "C" {

static int act_step_impl(const char* what, int group_rows, long long theta_stride, const int* opp_index, int n_policies,
                         const orl_net_desc* pnet, const float* ptheta, const orl_net_desc* cnet, const float* ctheta,
                 const float* policy_obs, const float* critic_obs, const float* action_masks, int B,
                 int deterministic, uint64_t seed, uint64_t row0, uint64_t rng_step, const uint64_t* rng_step_dev,
                 const float* forced_u, float* values, float* actions, float* logp, void* stream) {
  int rc = check_net(pnet, "orl_act_step(policy)", false);
  if (rc) return rc;
  ORL_REQUIRE(ptheta || ctheta, "orl_act_step: neither tower given");
  if (ptheta) ORL_REQUIRE(policy_obs && actions && logp, "orl_act_step: null policy pointer");
  ORL_REQUIRE(B > 0, "orl_act_step: B=%d", B);
  const bool wide = pnet->n_out > 2;  // = the NO > 4 instantiations of ORL_DISPATCH_HEAD: padded W3 image + logits tile
  size_t lds = TowerLds(pnet->obs_dim, pnet->n_out, pnet->head_kind == ORL_HEAD_GAUSSIAN, false, wide).total;
  if (ctheta) {
    rc = check_net(cnet, "orl_act_step(critic)", true);
    if (rc) return rc;
    ORL_REQUIRE(critic_obs && values, "orl_act_step: critic given without critic_obs/values");
    lds += TowerLds(cnet->obs_dim, 1, false, false).total;
  }
  if (wide) lds += TILE_B * 16;
  lds *= sizeof(float);
  ORL_REQUIRE(lds <= 160 * 1024, "orl_act_step: towers need %zu B of LDS (> 160 KiB)", lds);
  ActArgs A;
  A.pnet = *pnet;
  A.cnet = ctheta ? *cnet : *pnet;
  A.ptheta = ptheta; A.ctheta = ctheta; A.pobs = policy_obs; A.cobs = critic_obs; A.amask = action_masks;
  A.forced = forced_u; A.values = values; A.actions = actions; A.logp = logp; A.B = B;
  A.deterministic = deterministic; A.seed = seed; A.row0 = row0; A.rng_step = rng_step; A.rng_dev = (const unsigned long long*)rng_step_dev;
  A.group_rows = group_rows; A.theta_stride = theta_stride;
  A.opp_index = opp_index; A.n_policies = n_policies;
  const int grid = (B + TILE_B - 1) / TILE_B;
#define ORL_ACT_LAUNCH(NO, HD)                                                                                  \
  do {                                                                                                          \
    if (lds > 48 * 1024)                                                                                        \
      (void)hipFuncSetAttribute((const void*)act_step_kernel<NO, HD>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                (int)lds);                                                                      \
    hipLaunchKernelGGL((act_step_kernel<NO, HD>), dim3(grid), dim3(128), lds, (hipStream_t)stream, A);          \
  } while (0)
  ORL_DISPATCH_HEAD(ORL_ACT_LAUNCH);
#undef ORL_ACT_LAUNCH
  return launch_status(what);
}

int orl_act_step(const orl_net_desc* pnet, const float* ptheta, const orl_net_desc* cnet, const float* ctheta,
                 const float* policy_obs, const float* critic_obs, const float* action_masks, int B, int deterministic,
                 uint64_t seed, uint64_t row0, uint64_t rng_step, const uint64_t* rng_step_dev, const float* forced_u,
                 float* values, float* actions, float* logp, void* stream) {
  return act_step_impl("orl_act_step", 0, 0, nullptr, 0, pnet, ptheta, cnet, ctheta, policy_obs, critic_obs, action_masks, B,
                       deterministic, seed, row0, rng_step, rng_step_dev, forced_u, values, actions, logp, stream);
}

int orl_act_step_grouped(const orl_net_desc* pnet, const float* pthetas, int64_t theta_stride, int rows_per_group,
                         const float* policy_obs, const float* action_masks, int B, int deterministic, uint64_t seed,
                         uint64_t row0, uint64_t rng_step, const uint64_t* rng_step_dev, float* actions, float* logp,
                         void* stream) {
  ORL_REQUIRE(pthetas && rows_per_group > 0 && rows_per_group % TILE_B == 0 && theta_stride >= 0,
              "orl_act_step_grouped: rows_per_group must be a positive multiple of %d", TILE_B);
  return act_step_impl("orl_act_step_grouped", rows_per_group, theta_stride, nullptr, 0, pnet, pthetas, nullptr, nullptr, policy_obs,
                       nullptr, action_masks, B, deterministic, seed, row0, rng_step, rng_step_dev, nullptr, nullptr, actions,
                       logp, stream);
}

int orl_act_step_pool(const orl_net_desc* pnet, const float* pthetas, int64_t theta_stride, int n_policies,
                      const int32_t* opp_index, const float* policy_obs, const float* action_masks, int B,
                      int deterministic, uint64_t seed, uint64_t row0, uint64_t rng_step, const uint64_t* rng_step_dev,
                      float* actions, float* logp, void* stream) {
  ORL_REQUIRE(pthetas && opp_index && n_policies >= 1 && theta_stride >= 0, "orl_act_step_pool: bad pool arguments");
  return act_step_impl("orl_act_step_pool", 0, theta_stride, (const int*)opp_index, n_policies, pnet, pthetas, nullptr,
                       nullptr, policy_obs, nullptr, action_masks, B, deterministic, seed, row0, rng_step, rng_step_dev,
                       nullptr, nullptr, actions, logp, stream);
}

int orl_opponent_sample(int32_t* opp_index, const uint8_t* dones, int N, int n_filled, int last_slot, int strategy,
                        int per_tile, uint64_t seed, uint64_t draw_id, const uint64_t* draw_id_dev, void* stream) {
  ORL_REQUIRE(opp_index && N > 0 && n_filled >= 1 && (strategy == 0 || strategy == 1),
              "orl_opponent_sample: bad arguments (n_filled=%d, strategy=%d)", n_filled, strategy);
  hipLaunchKernelGGL(opponent_sample_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, (int*)opp_index,
                     dones, N, n_filled, last_slot, strategy, per_tile, seed, draw_id,
                     (const unsigned long long*)draw_id_dev);
  return launch_status("orl_opponent_sample");
}

int orl_evaluate_actions(const orl_net_desc* pnet, const float* ptheta, const orl_net_desc* cnet, const float* ctheta,
                         const float* policy_obs, const float* critic_obs, const float* actions,
                         const float* action_masks, const float* active_masks, int B, float* values,
                         float* action_log_probs, float* entropy_rows, float* dist_entropy, void* stream) {
  int rc = check_net(pnet, "orl_evaluate_actions(policy)", false);
  if (rc) return rc;
  ORL_REQUIRE(ptheta && policy_obs && actions && action_log_probs && entropy_rows && dist_entropy && B > 0,
              "orl_evaluate_actions: bad arguments");
  size_t lds = TowerLds(pnet->obs_dim, pnet->n_out, pnet->head_kind == ORL_HEAD_GAUSSIAN, false).total;
  if (ctheta) {
    rc = check_net(cnet, "orl_evaluate_actions(critic)", true);
    if (rc) return rc;
    ORL_REQUIRE(critic_obs && values, "orl_evaluate_actions: critic given without critic_obs/values");
    lds += TowerLds(cnet->obs_dim, 1, false, false).total;
  }
  lds *= sizeof(float);
  ORL_REQUIRE(lds <= 160 * 1024, "orl_evaluate_actions: towers need %zu B of LDS", lds);
  EvalArgs A;
  A.pnet = *pnet; A.cnet = ctheta ? *cnet : *pnet; A.ptheta = ptheta; A.ctheta = ctheta; A.pobs = policy_obs;
  A.cobs = critic_obs; A.actions = actions; A.amask = action_masks; A.active = active_masks; A.values = values;
  A.logp = action_log_probs; A.ent_w = entropy_rows; A.B = B;
  const int grid = (B + TILE_B - 1) / TILE_B;
#define ORL_EVAL_LAUNCH(NO, HD)                                                                                     \
  do {                                                                                                              \
    if (lds > 48 * 1024)                                                                                            \
      (void)hipFuncSetAttribute((const void*)eval_actions_kernel<NO, HD>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                (int)lds);                                                                          \
    hipLaunchKernelGGL((eval_actions_kernel<NO, HD>), dim3(grid), dim3(128), lds, (hipStream_t)stream, A);          \
  } while (0)
  ORL_DISPATCH_HEAD(ORL_EVAL_LAUNCH);
#undef ORL_EVAL_LAUNCH
  // masked mean: sum(ent*active)/sum(active); unmasked: mean over rows (categorical) or rows*dims (Gaussian)
  const float den_scale = (active_masks == nullptr && pnet->head_kind == ORL_HEAD_GAUSSIAN) ? (float)pnet->n_out : 1.f;
  hipLaunchKernelGGL(eval_entropy_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, entropy_rows, active_masks, B,
                     den_scale, dist_entropy);
  return launch_status("orl_evaluate_actions");
}

int orl_env_state_width(int env_kind) {
  if (env_kind == ORL_ENV_SYNTH) return SYNTH_STATE_W;
  if (env_kind == ORL_ENV_CARTPOLE) return CARTPOLE_STATE_W;
  return fail(ORL_E_INVALID, "orl_env_state_width: unknown env kind %d", env_kind);
}

int orl_env_reset(int env_kind, float* env_state, float* ep_stats, float* obs0, int N, int obs_dim,
                  uint64_t env_seed, int episode_limit, void* stream) {
  ORL_REQUIRE(env_state && obs0 && N > 0 && obs_dim > 0 && episode_limit > 0, "orl_env_reset: bad arguments");
  const int grid = (N + 255) / 256;
  if (env_kind == ORL_ENV_SYNTH) {
    hipLaunchKernelGGL((env_reset_kernel<ORL_ENV_SYNTH>), dim3(grid), dim3(256), 0, (hipStream_t)stream, env_state,
                       ep_stats, obs0, N, obs_dim, env_seed, episode_limit);
  } else if (env_kind == ORL_ENV_CARTPOLE) {
    ORL_REQUIRE(obs_dim == 4, "orl_env_reset: CartPole obs_dim must be 4");
    hipLaunchKernelGGL((env_reset_kernel<ORL_ENV_CARTPOLE>), dim3(grid), dim3(256), 0, (hipStream_t)stream,
                       env_state, ep_stats, obs0, N, obs_dim, env_seed, episode_limit);
  } else {
    return fail(ORL_E_INVALID, "orl_env_reset: unknown env kind %d", env_kind);
  }
  return launch_status("orl_env_reset");
}

int orl_env_step(int env_kind, float* env_state, float* ep_stats, const float* actions, int action_width, float* obs,
                 float* rewards, uint8_t* dones, int N, int obs_dim, uint64_t env_seed, int episode_limit,
                 uint64_t global_step, void* stream) {
  return orl_env_step_dev(env_kind, env_state, ep_stats, actions, action_width, obs, rewards, dones, N, obs_dim, env_seed,
                          episode_limit, global_step, nullptr, stream);
}

int orl_env_step_dev(int env_kind, float* env_state, float* ep_stats, const float* actions, int action_width, float* obs,
                     float* rewards, uint8_t* dones, int N, int obs_dim, uint64_t env_seed, int episode_limit,
                     uint64_t global_step, const int64_t* global_step_dev, void* stream) {
  ORL_REQUIRE(env_state && obs && rewards && dones && N > 0 && obs_dim > 0 && episode_limit > 0,
              "orl_env_step: bad arguments");
  const int grid = (N + 255) / 256;
  if (env_kind == ORL_ENV_SYNTH) {
    hipLaunchKernelGGL((env_step_kernel<ORL_ENV_SYNTH>), dim3(grid), dim3(256), 0, (hipStream_t)stream, env_state,
                       ep_stats, actions, action_width, obs, rewards, dones, N, obs_dim, env_seed, episode_limit,
                       global_step, (const long long*)global_step_dev);
  } else if (env_kind == ORL_ENV_CARTPOLE) {
    ORL_REQUIRE(actions && action_width >= 1 && obs_dim == 4, "orl_env_step: CartPole needs actions and 4-d obs");
    hipLaunchKernelGGL((env_step_kernel<ORL_ENV_CARTPOLE>), dim3(grid), dim3(256), 0, (hipStream_t)stream, env_state,
                       ep_stats, actions, action_width, obs, rewards, dones, N, obs_dim, env_seed, episode_limit,
                       global_step, (const long long*)global_step_dev);
  } else {
    return fail(ORL_E_INVALID, "orl_env_step: unknown env kind %d", env_kind);
  }
  return launch_status("orl_env_step");
}

#ifdef ORL_PROF
// debug build only: cumulative per-phase cycle counts of waves 0/1 of workgroup 0 of every fused rollout; reset on read
int orl_debug_rollout_prof(unsigned long long* out16) {
  unsigned long long zero[16] = {0};
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_roll_prof), sizeof(zero));
  hipMemcpyToSymbol(HIP_SYMBOL(g_roll_prof), zero, sizeof(zero));
  return 0;
}
#endif

static int critic_values_launch(const orl_net_desc* cnet, const float* ctheta, const float* critic_obs, int64_t rows,
                                float* values, float* tail_out, int64_t tail_row0, void* stream) {
  int rc = check_net(cnet, "orl_critic_values", true);
  if (rc) return rc;
  ORL_REQUIRE(ctheta && critic_obs && values && rows > 0, "orl_critic_values: bad arguments");
  const size_t lds = (size_t)TowerLds(cnet->obs_dim, 1, false, false).total * sizeof(float);
  const long long n_tiles = (rows + TILE_B - 1) / TILE_B;
  int grid = (int)((n_tiles + 7) / 8);
  if (grid > 512) grid = 512;  // two 8-wave workgroups per CU
  if (lds > 48 * 1024)
    (void)hipFuncSetAttribute((const void*)critic_values_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(critic_values_kernel, dim3(grid), dim3(512), lds, (hipStream_t)stream, *cnet, ctheta, critic_obs,
                     (long long)rows, values, tail_out, (long long)tail_row0);
  return launch_status("orl_critic_values");
}

int orl_critic_values(const orl_net_desc* cnet, const float* ctheta, const float* critic_obs, int64_t rows,
                      float* values, void* stream) {
  return critic_values_launch(cnet, ctheta, critic_obs, rows, values, nullptr, 0, stream);
}

int orl_rollout_fused(const orl_net_desc* pnet, const float* ptheta, const orl_net_desc* cnet, const float* ctheta,
                      const orl_rollout_args* args, float* next_value, void* stream) {
  int rc = check_net(pnet, "orl_rollout_fused(policy)", false);
  if (rc) return rc;
  rc = check_net(cnet, "orl_rollout_fused(critic)", true);
  if (rc) return rc;
  ORL_REQUIRE(ptheta && ctheta && args, "orl_rollout_fused: null pointer");
  const orl_buffer_ptrs& b = args->buf;
  ORL_REQUIRE(b.A == 1, "orl_rollout_fused: device envs are single-agent (A=%d)", b.A);
  ORL_REQUIRE(b.Dp == pnet->obs_dim && b.Dc == cnet->obs_dim && b.Dp == b.Dc,
              "orl_rollout_fused: obs dims buffer(%d,%d) nets(%d,%d)", b.Dp, b.Dc, pnet->obs_dim, cnet->obs_dim);
  ORL_REQUIRE(b.policy_obs && b.critic_obs && b.rewards && b.masks && b.bad_masks && b.active_masks &&
                  args->value_preds && args->actions && args->action_log_probs && args->env_state && args->ep_stats,
              "orl_rollout_fused: null buffer array");
  ORL_REQUIRE(args->episode_limit > 0, "orl_rollout_fused: episode_limit");
  if (args->env_kind == ORL_ENV_CARTPOLE)
    ORL_REQUIRE(pnet->head_kind == ORL_HEAD_CATEGORICAL && pnet->n_out == 2 && b.Dp == 4,
                "orl_rollout_fused: CartPole needs Discrete(2) and 4-d obs");
  const bool pool = args->env_kind == ORL_ENV_TTT_POOL;
  const bool perk = pool && args->opp_per_reset != 0;  // per-env opponents, re-drawn in-kernel at every auto-reset
  if (pool)
    ORL_REQUIRE(args->opp_thetas && args->opp_group_rows > 0 && args->opp_group_rows % TILE_B == 0 &&
                    args->opp_theta_stride >= 0,
                "orl_rollout_fused: the opponent pool needs opp_thetas and opp_group_rows (a multiple of %d)", TILE_B);
  if (perk)
    ORL_REQUIRE(args->opp_index && args->opp_n_policies >= 1 && args->opp_n_filled >= 1 &&
                    args->opp_n_filled <= args->opp_n_policies && args->opp_last_slot >= 0 &&
                    args->opp_last_slot < args->opp_n_policies,
                "orl_rollout_fused: opp_per_reset needs opp_index and 1 <= opp_n_filled <= opp_n_policies (got %d, %d)",
                args->opp_n_filled, args->opp_n_policies);
  if (args->env_kind == ORL_ENV_TTT || pool)
    ORL_REQUIRE(pnet->head_kind == ORL_HEAD_CATEGORICAL && pnet->n_out == 9 && b.Dp == 18 && b.K == 9 && b.action_masks,
                "orl_rollout_fused: tic-tac-toe needs Discrete(9), 18-d obs and the action-mask array");
  const TowerLds twp(pnet->obs_dim, pnet->n_out, pnet->head_kind == ORL_HEAD_GAUSSIAN, false, pnet->n_out > 2);
  const TowerLds twc(cnet->obs_dim, 1, false, false);
  // The critic stays inside the step loop: a policy-only loop + one batched orl_critic_values launch over all T+1
  // slots was measured slower at config 2 (271 -> 237 us step loop, +60 us value launch).
  // Per-env opponents keep every pool image resident instead of the critic (which then runs as ONE batched launch over
  // the stored observations afterwards): 1 + K policy images fit the 160 KiB for K <= 4.
  const size_t lds =
      perk ? (size_t)(twp.total + 2 * TILE_B * twp.DP + 2 * TILE_B * GS + 4 * TILE_B * 16 +
                      (size_t)args->opp_n_policies * twp.total + TILE_B * twp.DP + TILE_B * 16 + TILE_B) * sizeof(float)
           : (size_t)(twp.total + twc.total + 2 * TILE_B * twp.DP + 4 * TILE_B * GS + 4 * TILE_B * 16 +
                      (pool ? twp.total + TILE_B * twp.DP + TILE_B * 16 + TILE_B : 0)) * sizeof(float);
  if (perk && lds > 160 * 1024)
    return fail(ORL_E_UNSUPPORTED, "orl_rollout_fused: a pool of %d snapshots does not fit the LDS (%zu B; 4 do)",
                args->opp_n_policies, lds);
  ORL_REQUIRE(lds <= 160 * 1024, "orl_rollout_fused: needs %zu B of LDS", lds);
  RolloutArgs A;
  A.pnet = *pnet; A.cnet = *cnet; A.ptheta = ptheta; A.ctheta = ctheta; A.r = *args; A.next_value = next_value;
  const int grid = (b.N + TILE_B - 1) / TILE_B;
  // Round 6: the synthetic env and CartPole roll out on the chain kernel (orl_rollout2.h: policy-only step chain, the critic on
  // background waves of the same launch).  args.opp_reserved = 1 keeps the round-5 lock-step kernel (policy + critic in the
  // step loop) - the A/B switch of tests and benchmarks.
  const size_t lds2 = (size_t)ro2_lds(TowerLds(pnet->obs_dim, pnet->n_out, pnet->head_kind == ORL_HEAD_GAUSSIAN, false, pnet->n_out > 2,
                                               ORL_TOWER_F16 != 0).total,
                                      TowerLds(cnet->obs_dim, 1, false, false, false, true).total, twp.DP).total *
                      sizeof(float);
  // (the widest towers - observations of ~60 columns with 16 outputs - do not fit the chain kernel's rings beside both tower
  // images: they keep the round-5 kernel)
  if ((args->env_kind == ORL_ENV_SYNTH || args->env_kind == ORL_ENV_CARTPOLE || args->env_kind == ORL_ENV_TTT) &&
      args->opp_reserved != 1 && lds2 <= 160 * 1024) {
#define ORL_RO2_LAUNCH3(NO, HD, EV, KS)                                                                               \
  do {                                                                                                               \
    if (lds2 > 48 * 1024)                                                                                            \
      (void)hipFuncSetAttribute((const void*)rollout2_kernel<NO, HD, EV, KS>,                                        \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);                              \
    hipLaunchKernelGGL((rollout2_kernel<NO, HD, EV, KS>), dim3(grid), dim3(RO2_THREADS), lds2, (hipStream_t)stream, A); \
  } while (0)
    if (args->env_kind == ORL_ENV_CARTPOLE) {
      ORL_RO2_LAUNCH3(2, ORL_HEAD_CATEGORICAL, ORL_ENV_CARTPOLE, 1);
    } else if (args->env_kind == ORL_ENV_TTT) {
      ORL_RO2_LAUNCH3(16, ORL_HEAD_CATEGORICAL, ORL_ENV_TTT, 0);  // the random opponent (the pool variants keep the round-5 kernel)
    } else if (twp.DP == 4 && pnet->head_kind == ORL_HEAD_CATEGORICAL && pnet->n_out <= 2) {
      ORL_RO2_LAUNCH3(2, ORL_HEAD_CATEGORICAL, ORL_ENV_SYNTH, 1);  // configuration 2's shape: fc1's single k-step at compile time
    } else {
#define ORL_RO2_LAUNCH(NO, HD) ORL_RO2_LAUNCH3(NO, HD, ORL_ENV_SYNTH, 0)
      ORL_DISPATCH_HEAD(ORL_RO2_LAUNCH);
#undef ORL_RO2_LAUNCH
    }
#undef ORL_RO2_LAUNCH3
    return launch_status("orl_rollout_fused");
  }
#define ORL_RO_LAUNCH3(NO, HD, EV, WC, THREADS)                                                                      \
  do {                                                                                                              \
    if (lds > 48 * 1024)                                                                                            \
      (void)hipFuncSetAttribute((const void*)rollout_kernel<NO, HD, EV, WC>,                                        \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                              \
    hipLaunchKernelGGL((rollout_kernel<NO, HD, EV, WC>), dim3(grid), dim3(THREADS), lds, (hipStream_t)stream, A);   \
  } while (0)
#define ORL_RO_LAUNCH2(NO, HD, EV) ORL_RO_LAUNCH3(NO, HD, EV, true, 512)
  if (args->env_kind == ORL_ENV_SYNTH) {
#define ORL_RO_LAUNCH(NO, HD) ORL_RO_LAUNCH2(NO, HD, ORL_ENV_SYNTH)
    ORL_DISPATCH_HEAD(ORL_RO_LAUNCH);
#undef ORL_RO_LAUNCH
  } else if (args->env_kind == ORL_ENV_CARTPOLE) {
    ORL_RO_LAUNCH2(2, ORL_HEAD_CATEGORICAL, ORL_ENV_CARTPOLE);
  } else if (args->env_kind == ORL_ENV_TTT) {
    ORL_RO_LAUNCH2(16, ORL_HEAD_CATEGORICAL, ORL_ENV_TTT);
  } else if (perk) {
    ORL_RO_LAUNCH3(16, ORL_HEAD_CATEGORICAL, ORL_ENV_TTT_POOLK, false, 256);
  } else if (pool) {
    ORL_RO_LAUNCH2(16, ORL_HEAD_CATEGORICAL, ORL_ENV_TTT_POOL);
  } else {
    return fail(ORL_E_INVALID, "orl_rollout_fused: unknown env kind %d", args->env_kind);
  }
#undef ORL_RO_LAUNCH2
#undef ORL_RO_LAUNCH3
  rc = launch_status("orl_rollout_fused");
  if (rc || !perk) return rc;
  // per-env opponents: values of every slot 0..T from the stored critic observations (the same per-row arithmetic as
  // orl_act_step's critic); slot T doubles as the bootstrap value
  rc = orl_critic_values(cnet, ctheta, b.critic_obs, (int64_t)(b.T + 1) * b.N, args->value_preds, stream);
  if (rc) return rc;
  if (next_value != nullptr &&
      hipMemcpyAsync(next_value, args->value_preds + (size_t)b.T * b.N, sizeof(float) * b.N, hipMemcpyDeviceToDevice,
                     (hipStream_t)stream) != hipSuccess)
    return fail(ORL_E_INVALID, "orl_rollout_fused: copying the bootstrap values failed");
  return launch_status("orl_rollout_fused(values)");
}

}  // extern "C"
