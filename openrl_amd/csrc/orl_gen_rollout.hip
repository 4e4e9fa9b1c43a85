// orl_gen_rollout.hip - the fused rollout of the GENERAL tower path for gfx950: all episode_length steps of
//   {policy tower (any hidden_size / layer_N / activation / feature norm, also the shared network's value head),
//    ACTLayer sampling, env.step of a device-resident single-agent env, ReplayData.insert}
// in ONE launch (orl_gen_rollout_fused), the loop of openrl/drivers/onpolicy_driver.py:154-203 for towers outside the
// default 64-wide one.  A workgroup owns 16 envs for the whole rollout: per step the 16 rows go through mlp_tile
// (orl_gen_mlp.h - the arithmetic of the one-launch step orl_gen_act), lanes 0-15 then step their env with the shared
// device functions of orl_env.h and write observation / reward / masks straight into slot t + 1 of the buffer.  A
// SEPARATE critic is not walked step by step: its values are not needed before the GAE, so the host computes all
// T + 1 slots in ONE orl_gen_mlp_fwd launch over (T + 1) N rows afterwards (throughput-bound instead of sharing a
// latency chain).  Built with the env kernels' arithmetic flags (-ffast-math, like orl_act.hip) so that observations,
// rewards and dones are bit-identical to the stepwise route's given the same actions.
#include <string.h>
#include "orl_common.h"
#include "orl_mlp.h"
#include "orl_env.h"
#include "orl_gen_mlp.h"

namespace orl {

struct GenRollArgs {
  orl_buffer_ptrs buf;
  float* value_preds;  // [T + 1][N] (written per step only by a shared network's value head)
  float* actions;      // [T][N][a_w]
  float* logp;         // [T][N][a_w]
  float* env_state;
  float* ep_stats;
  int env_kind, episode_limit;
  uint64_t env_seed, tg0;  // env's global step at the first rollout step
};

// WL: the tower's parameters fit LDS behind the slab (MlpLds): staged once, read from there by all T steps
template <int NBW, int WAVES, int ENV, bool WL>
__global__ __launch_bounds__(64 * WAVES) void gen_rollout_kernel(MlpArgs A, ActArgs S, GenRollArgs R, int SLD, MlpLds WLY) {
  extern __shared__ __attribute__((aligned(16))) float slab[];  // [16][SLD] + [16][LGS_LD] (+ the parameters' copy)
  const orl_buffer_ptrs& b = R.buf;
  const int N = b.N, T = b.T, D = b.Dp, K = b.K;
  const int tid = threadIdx.x;
  const long long m0 = (long long)blockIdx.x * 16;
  float* wbase = slab + ((16 * SLD + 16 * LGS_LD + 3) & ~3);
  float* act_lds = wbase + (WL ? WLY.total : 0);  // [16][2 a_w]: the step's sampled actions / log-probs
  if constexpr (WL) {
    mlp_lds_stage(A.d, WLY, wbase, tid, 64 * WAVES);
    __syncthreads();
  }
  // lanes 0-15 own an env each: its state row and episode statistics stay in registers for the whole rollout, the next
  // observation goes straight into the slab (and to the buffer), the action comes out of LDS - no global round trip
  // on the step's dependent chain (state load -> store -> load, action store -> load, observation store -> load)
  constexpr int SW = ENV == ORL_ENV_SYNTH ? SYNTH_STATE_W : CARTPOLE_STATE_W;
  const int n = (int)(m0 + tid);
  const bool own = tid < 16 && n < N;
  const bool has_stats = R.ep_stats != nullptr;
  float est[SW], eps[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < SW; ++k) est[k] = own ? R.env_state[(size_t)n * SW + k] : 0.f;
  if (own && has_stats) {
#pragma unroll
    for (int k = 0; k < 4; ++k) eps[k] = R.ep_stats[(size_t)n * 4 + k];
  }
  const int DP = (D + 15) & ~15, a_w = S.a_w;
  for (int t = 0; t < T; ++t) {
    // A.x = slot 0 of the observations, A.head_out / S.actions / S.logp / S.amask = slot 0 of their arrays (set by the
    // host): the row offset t * N selects slot t, the step counter advances by t
    mlp_tile<NBW, WAVES, WL>(A, SLD, slab, &S, (long long)t * N, (unsigned long long)t, &WLY, wbase, t > 0, act_lds);
    // lanes 0-15 sampled row m0 + tid just above: the same lanes step that env and insert (onpolicy_driver.py:80-152)
    if (own) {
      const float* al = act_lds + tid * 2 * a_w;
      for (int k = 0; k < a_w; ++k) {
        S.actions[((size_t)t * N + n) * a_w + k] = al[k];
        S.logp[((size_t)t * N + n) * a_w + k] = al[a_w + k];
      }
      const int action = ENV == ORL_ENV_SYNTH ? 0 : (int)al[0];
      float r;
      bool d;
      const size_t s1 = (size_t)(t + 1) * N + n;
      float* orow = slab + tid * SLD;  // the next step's input row
      env_step_state<ENV>(est, has_stats ? eps : nullptr, n, D, R.env_seed, R.episode_limit, R.tg0 + (uint64_t)t, action,
                          orow, r, d);
      for (int c = D; c < DP; ++c) orow[c] = 0.f;  // (the layers' outputs went through these columns)
      for (int c = 0; c < D; ++c) b.policy_obs[s1 * D + c] = orow[c];
      b.rewards[(size_t)t * N + n] = r;
      b.masks[s1] = d ? 0.f : 1.f;
      b.active_masks[s1] = 1.f;
      b.bad_masks[s1] = 1.f;
      if (K > 0 && b.action_masks != nullptr)
        for (int k = 0; k < K; ++k) b.action_masks[s1 * K + k] = 1.f;  // these envs have no illegal actions
    } else if (tid < 16) {
      float* orow = slab + tid * SLD;  // rows beyond N: keep the tile's input finite
      for (int c = 0; c < DP; ++c) orow[c] = 0.f;
    }
    __syncthreads();  // the next step's input rows are in the slab
  }
  if (own) {
#pragma unroll
    for (int k = 0; k < SW; ++k) R.env_state[(size_t)n * SW + k] = est[k];
    if (has_stats) {
#pragma unroll
      for (int k = 0; k < 4; ++k) R.ep_stats[(size_t)n * 4 + k] = eps[k];
    }
  }
}

}  // namespace orl

using namespace orl;

extern "C" {

int orl_gen_rollout_fused(const orl_gen_mlp_desc* policy, const orl_head_desc* head, const float* logstd,
                          const orl_buffer_ptrs* buf, float* value_preds, float* actions, float* action_log_probs,
                          float* env_state, float* ep_stats, int env_kind, int episode_limit, uint64_t env_seed,
                          uint64_t env_step0, uint64_t act_seed, uint64_t rng_step0, int a_w, void* stream) {
  ORL_REQUIRE(policy && head && buf && actions && action_log_probs && env_state, "orl_gen_rollout_fused: null pointer");
  ORL_REQUIRE(buf->A == 1 && buf->N > 0 && buf->T > 0 && buf->policy_obs && buf->rewards && buf->masks && buf->bad_masks &&
                  buf->active_masks && (buf->critic_obs == buf->policy_obs || buf->critic_obs == nullptr),
              "orl_gen_rollout_fused: single-agent buffers whose critic observations alias the policy's");
  ORL_REQUIRE(env_kind == ORL_ENV_SYNTH || env_kind == ORL_ENV_CARTPOLE, "orl_gen_rollout_fused: env kind %d (synthetic / CartPole)", env_kind);
  ORL_REQUIRE(env_kind != ORL_ENV_CARTPOLE || (buf->Dp == 4 && head->kind == ORL_HEAD_CATEGORICAL && head->n_out == 2),
              "orl_gen_rollout_fused: CartPole needs 4-d observations and a Discrete(2) head");
  ORL_REQUIRE(policy->n_layers >= 1 && policy->n_heads >= 1 && policy->n_heads <= 2 &&
                  policy->n_layers + policy->n_heads <= ORL_GEN_MLP_MAX_LAYERS && policy->layer[0].n_in == buf->Dp,
              "orl_gen_rollout_fused: the policy tower reads %d columns, the buffer holds %d", policy->layer[0].n_in, buf->Dp);
  ORL_REQUIRE(policy->n_heads == 1 || value_preds, "orl_gen_rollout_fused: a value head needs value_preds");
  ORL_REQUIRE(head->n_out == policy->layer[policy->n_layers].n_out && head->n_out >= 1 && head->n_out <= GEN_MAX_OUT,
              "orl_gen_rollout_fused: the head descriptor has %d logits, the tower's first head %d", head->n_out,
              policy->layer[policy->n_layers].n_out);
  ORL_REQUIRE((head->kind != ORL_HEAD_GAUSSIAN && head->kind != ORL_HEAD_MIXED) || logstd, "orl_gen_rollout_fused: Gaussian head without logstd");
  const int want_aw = head->kind == ORL_HEAD_CATEGORICAL ? 1 : head->kind == ORL_HEAD_GAUSSIAN ? head->n_out
                      : head->kind == ORL_HEAD_MULTI_DISCRETE ? head->n_heads : head->nvec[0] + 1;
  ORL_REQUIRE(a_w == want_aw, "orl_gen_rollout_fused: a_w %d for head kind %d", a_w, head->kind);
  ORL_REQUIRE(buf->K == 0 || head->kind == ORL_HEAD_CATEGORICAL, "orl_gen_rollout_fused: action masks with a non-categorical head");
  int wmax = 0, width = (policy->layer[0].n_in + 15) & ~15;
  for (int L = 0; L < policy->n_layers + policy->n_heads; ++L) {
    const orl_gen_mlp_layer& ly = policy->layer[L];
    ORL_REQUIRE(ly.W && ly.n_in > 0 && ly.n_out > 0 && ly.n_out <= 256, "orl_gen_rollout_fused: entry %d: n_in %d, n_out %d", L, ly.n_in, ly.n_out);
    ORL_REQUIRE(L >= policy->n_layers || (ly.gamma && ly.beta && (ly.n_out & 3) == 0), "orl_gen_rollout_fused: layer %d needs LayerNorm parameters and a width that is a multiple of 4", L);
    if (ly.n_out > wmax) wmax = ly.n_out;
    const int w16 = (ly.n_out + 15) & ~15;
    if (w16 > width) width = w16;
  }
  ORL_REQUIRE(width <= 1024, "orl_gen_rollout_fused: %d columns do not fit the wave's LDS slab", width);
  MlpArgs A;
  memset(&A, 0, sizeof(A));
  A.d = *policy; A.B = buf->N; A.x = buf->policy_obs;
  A.head_out[0] = nullptr; A.head_out[1] = policy->n_heads == 2 ? value_preds : nullptr;
  ActArgs S;
  memset(&S, 0, sizeof(S));
  S.hd = *head; S.logstd = logstd; S.deterministic = 0; S.seed = act_seed; S.row0 = 0; S.rng_step = rng_step0; S.a_w = a_w;
  S.actions = actions; S.logp = action_log_probs;
  S.amask = (buf->K > 0 && buf->action_masks != nullptr) ? buf->action_masks : nullptr;
  GenRollArgs R;
  R.buf = *buf; R.value_preds = value_preds; R.actions = actions; R.logp = action_log_probs; R.env_state = env_state;
  R.ep_stats = ep_stats; R.env_kind = env_kind; R.episode_limit = episode_limit; R.env_seed = env_seed; R.tg0 = env_step0;
  const int NBW = wmax <= 128 ? 1 : 2, WV = wmax <= 64 ? 4 : 8;  // as orl_gen_act picks them
  if (width < 16 * WV * NBW) width = 16 * WV * NBW;
  const int SLD = width + 4;
  // the tower's parameters behind the slab when everything fits the CU's 160 KB (a workgroup per CU at 4096 envs)
  const MlpLds wly = mlp_lds_layout(*policy);
  const size_t lds_slab = ((size_t)((16 * SLD + 16 * LGS_LD + 3) & ~3)) * sizeof(float);
  const size_t lds_act = (size_t)16 * 2 * a_w * sizeof(float);  // the step's sampled actions / log-probs
  const bool wl = lds_slab + lds_act + (size_t)wly.total * sizeof(float) <= 160 * 1024;
  const size_t lds = lds_slab + lds_act + (wl ? (size_t)wly.total * sizeof(float) : 0);
  const dim3 grid((unsigned)((buf->N + 15) / 16));
#define ORL_ROLL_LAUNCH2(NBX, WVX, ENVX, WLX)                                                                             \
  do {                                                                                                                    \
    (void)hipFuncSetAttribute((const void*)gen_rollout_kernel<NBX, WVX, ENVX, WLX>,                                       \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                      \
    hipLaunchKernelGGL((gen_rollout_kernel<NBX, WVX, ENVX, WLX>), grid, dim3(64 * WVX), lds, (hipStream_t)stream, A, S, R, \
                       SLD, wly);                                                                                         \
  } while (0)
#define ORL_ROLL_LAUNCH(NBX, WVX, ENVX)                 \
  do {                                                  \
    if (wl) ORL_ROLL_LAUNCH2(NBX, WVX, ENVX, true);     \
    else ORL_ROLL_LAUNCH2(NBX, WVX, ENVX, false);       \
  } while (0)
#define ORL_ROLL_ENV(NBX, WVX)                                         \
  do {                                                                 \
    if (env_kind == ORL_ENV_SYNTH) ORL_ROLL_LAUNCH(NBX, WVX, ORL_ENV_SYNTH); \
    else ORL_ROLL_LAUNCH(NBX, WVX, ORL_ENV_CARTPOLE);                  \
  } while (0)
  if (WV == 4) ORL_ROLL_ENV(1, 4);
  else if (NBW == 1) ORL_ROLL_ENV(1, 8);
  else ORL_ROLL_ENV(2, 8);
#undef ORL_ROLL_ENV
#undef ORL_ROLL_LAUNCH
#undef ORL_ROLL_LAUNCH2
  return launch_status("orl_gen_rollout_fused");
}

}  // extern "C"
