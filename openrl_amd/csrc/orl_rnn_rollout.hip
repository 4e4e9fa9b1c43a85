// orl_rnn_rollout.hip - fused rollout of a recurrent (GRU) MAPPO policy on the device-resident MPE `simple_spread`
// (BASELINE config 4): the whole episode_length-step actor_rollout
// (openrl/drivers/onpolicy_driver.py:154-233 with use_recurrent_policy) in ONE launch instead of
// T x {orl_rnn_act_step, orl_mpe_step, orl_buffer_insert}.
//
//   policy workgroups : 16 worlds each; wave a = agent a of those worlds (a 16-row tile of the policy tower, weights
//       staged into LDS once).  Hidden states stay in registers for all T steps, the worlds in the registers of 16
//       lanes; actions reach the world lanes and the new observations reach the towers through LDS.  Every per-step
//       buffer field (obs / share_obs, actions, log-probs, rewards, masks, rnn_states) is written from here, the
//       observations as coalesced copies of the LDS tile.
//   critic workgroups : the critic has no influence on the trajectory, so it only consumes the stored share_obs:
//       one wave per 16-row tile, T + 1 steps with its hidden state in registers (the last step is the bootstrap
//       value of compute_returns).
// The two kinds alternate in the grid (block 2b = policy of worlds 16b.., block 2b+1 = critic of the same 48 rows) and
// the critic CHASES its policy workgroup one step behind through a per-pair step counter in global memory (release
// store after slot t+1 is written, acquire spin before it is read; the spin is bounded by the wall clock and a
// critic only ever waits for a LOWER-numbered block, which the dispatcher has already placed).  Without the counter
// array (sync_flags == NULL) the two kinds run as two launches, critic after policy.
//
// Both use the per-tile arithmetic of the stepwise kernels (rnn_tower_fwd_lds, sample_head, mpe_advance) and the same
// Philox counters (act_seed, row, rng_step0 + t), so a fused rollout reproduces the stepwise one.
#include "orl_common.h"
#include "orl_mlp.h"
#include "orl_heads.h"
#include "orl_rnn.h"
#include "orl_mpe.h"
#include "orl_env.h"

namespace orl {

struct RnnRolloutArgs {
  orl_net_desc pnet, cnet;
  const float *ptheta, *ctheta;
  orl_buffer_ptrs buf;
  float *value_preds, *actions, *logp, *hp, *hc;
  float *env_state, *ep_stats, *obs_p_out, *obs_c_out, *next_value;
  int world_length, deterministic;
  uint64_t env_seed, act_seed, rng_step0;
  const unsigned long long* rng_dev;
  int* flags;  // [n_pairs + 1]: steps published by policy workgroup b; [n_pairs] = error word (chase timed out)
  uint64_t env_step0;  // single-agent envs: the env's global step at the first rollout step
};

constexpr unsigned long long CHASE_TIMEOUT_TICKS = 200000000ull;  // 2 s of the 100 MHz wall clock

// Device-coherent accesses for the data a critic workgroup reads while its policy workgroup is still running (the two
// may sit on different XCDs, whose L2s are not coherent for ordinary cached accesses): relaxed agent-scope atomics are
// written through / read around the non-coherent levels, so no L2 write-back (an agent-scope RELEASE fence flushes the
// whole L2: measured 2.3x on the policy's step) is needed - only program order plus the step counter.
__device__ inline void st_agent(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline float ld_agent(const float* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

constexpr int OBS_S = 20;  // padded LDS row of one agent's observation (18 -> 20 floats: fc1's k-steps of 4)

// bid = index of the 16-world group; CHASE: publish the step counter for the critic workgroup of the same group
template <int NO, bool CHASE>
__device__ __forceinline__ void rnn_rollout_policy_body(const RnnRolloutArgs& A, const int bid) {
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
  const int wave = threadIdx.x >> 6, l = threadIdx.x & 63, j = l & 15, q = l >> 4;
  const int N = b.N, T = b.T;
  const int LA = N * MPE_A;
  const int e0 = bid * TILE_B;
  const int env = e0 + j;
  const bool ok = env < N;
  const int row = (ok ? env : 0) * MPE_A + wave;  // this lane's buffer row (world, agent = wave)
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
  const bool world_lane = wave == 0 && q == 0 && ok;
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

  for (int t = 0; t < T; ++t) {
    f32x4 hin[4], hnew[4], n3[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) hin[m] = h[m] * mk;
    const float* xrow = s_obs + (wave * TILE_B + j) * OBS_S;
    rnn_tower_fwd_lds(lw, tw, [&](int s) -> float { return xrow[4 * s + q]; }, hin, hnew, n3, j, q);
    float hd[NO], act_o[NO], lp_o[NO];
    head_T<NO>(lw + tw.W3, lw + tw.b3, tl.n_out, n3, q, hd);
    sample_head<NO, ORL_HEAD_CATEGORICAL>(hd, tl.n_out, nullptr, nullptr, nullptr, A.deterministic, A.act_seed,
                                          (uint64_t)row, rng0 + (uint64_t)t, act_o, lp_o);
    if (q == 0) {
      s_act[wave * TILE_B + j] = act_o[0];
      if (ok) {
        A.actions[(size_t)t * LA + row] = act_o[0];
        A.logp[(size_t)t * LA + row] = lp_o[0];
      }
    }
    __syncthreads();  // actions of the 3 agents visible; every wave is done reading this step's observations
    if (world_lane) {
      int act[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) act[i] = (int)s_act[i * TILE_B + j];
      float rew;
      bool done;
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
    }
    __syncthreads();  // next observations, rewards and done flags visible
    // masks[t+1] = 0 where the world finished (every agent of a world finishes together), and
    // rnn_states[dones_env] = 0 (onpolicy_driver.py:100-113) applied to the registers and the stored slot
    mk = s_done[j] != 0.f ? 0.f : 1.f;
#pragma unroll
    for (int m = 0; m < 4; ++m) h[m] = hnew[m] * mk;
    if (ok) {
#pragma unroll
      for (int m = 0; m < 4; ++m) *(f32x4*)(A.hp + ((size_t)(t + 1) * LA + row) * HID + 16 * m + 4 * q) = h[m];
    }
    // per-step scalars of the 16 worlds x 3 agents: rows e0*3 .. e0*3 + 3*n_here - 1 are contiguous
    if ((int)threadIdx.x < MPE_A * n_here) {
      const int jj = threadIdx.x / MPE_A;
      const size_t r1 = (size_t)(t + 1) * LA + (size_t)e0 * MPE_A + threadIdx.x;
      b.rewards[(size_t)t * LA + (size_t)e0 * MPE_A + threadIdx.x] = s_rew[jj];
      if (CHASE) st_agent(b.masks + r1, s_done[jj] != 0.f ? 0.f : 1.f);
      else b.masks[r1] = s_done[jj] != 0.f ? 0.f : 1.f;
      b.active_masks[r1] = 1.f;  // (done && !all_done) never holds: the agents of a world share the done flag
      b.bad_masks[r1] = 1.f;
    }
    // observations of slot t+1 as coalesced copies of the LDS tile: policy [world][agent][18], critic
    // [world][agent][3 x 18] (every agent's share_obs is the concatenation of all three observations)
    {
      float* dp = b.policy_obs + ((size_t)(t + 1) * LA + (size_t)e0 * MPE_A) * MPE_OBS;
      for (int e = threadIdx.x; e < n_here * MPE_A * MPE_OBS; e += blockDim.x) {
        const int jj = e / (MPE_A * MPE_OBS), r = e - jj * (MPE_A * MPE_OBS), i = r / MPE_OBS, k = r - i * MPE_OBS;
        const float v = s_obs[(i * TILE_B + jj) * OBS_S + k];
        dp[e] = v;
        if (t == T - 1 && A.obs_p_out != nullptr) A.obs_p_out[(size_t)e0 * MPE_A * MPE_OBS + e] = v;
      }
      float* dc = b.critic_obs + ((size_t)(t + 1) * LA + (size_t)e0 * MPE_A) * MPE_COBS;
      for (int e = threadIdx.x; e < n_here * MPE_A * MPE_COBS; e += blockDim.x) {
        const int jj = e / (MPE_A * MPE_COBS), r = e - jj * (MPE_A * MPE_COBS), c = r % MPE_COBS;
        const int i2 = c / MPE_OBS, k = c - i2 * MPE_OBS;
        const float v = s_obs[(i2 * TILE_B + jj) * OBS_S + k];
        if (CHASE) st_agent(dc + e, v);
        else dc[e] = v;
        if (t == T - 1 && A.obs_c_out != nullptr) A.obs_c_out[(size_t)e0 * MPE_A * MPE_COBS + e] = v;
      }
    }
    if (CHASE) {  // slot t+1 (share_obs, masks) is complete: publish it to the critic workgroup of this group
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this thread's write-through stores have been acknowledged
      __syncthreads();
      if (threadIdx.x == 0) __hip_atomic_store(A.flags + bid, t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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

// Critic sweep over the stored observations, rows [0, N*A) of every slot; one 16-row tile per wave.  CHASE: tile =
// 3 * bid + wave (the rows of policy group bid) and slot t is read only after that group has published step t.
template <bool CHASE>
__device__ __forceinline__ void rnn_rollout_critic_body(const RnnRolloutArgs& A, const int bid, const int nblk) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const orl_buffer_ptrs& b = A.buf;
  const RnnLayout tl(A.cnet);
  const RnnLds tw(A.cnet.obs_dim, 1, false);
  stage_rnn_tower(smem, A.ctheta, tl, tw, threadIdx.x, blockDim.x);
  __syncthreads();
  const float* lw = smem;
  const int wave = threadIdx.x >> 6, l = threadIdx.x & 63, j = l & 15, q = l >> 4;
  const int D = tl.D, T = b.T;
  const int LA = b.N * b.A;
  const int n_tiles = (LA + TILE_B - 1) / TILE_B;
  const int nwv = blockDim.x >> 6;
  constexpr int XK = 16;  // k-steps of 4 observation columns held per lane (D <= 64)
  float* slab = smem + tw.total + wave * TILE_B * 64;  // this wave's observation tile [16][DP]
  for (int tile = bid * nwv + wave; tile < n_tiles; tile += nblk * nwv) {
    const int row = tile * TILE_B + j;
    const bool ok = row < LA;
    const int rr = ok ? row : 0;
    f32x4 h[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) h[m] = *(const f32x4*)(A.hc + (size_t)rr * HID + 16 * m + 4 * q);
    float xv[XK], mk = b.masks[rr];
    auto load_x = [&](int t, float (&x)[XK]) {
      const float* xr = b.critic_obs + ((size_t)t * LA + rr) * D;
#pragma unroll
      for (int s = 0; s < XK; ++s) {
        const int c = 4 * s + q;
        x[s] = (4 * s < D && c < D) ? (CHASE && t > 0 ? ld_agent(xr + c) : xr[c]) : 0.f;
      }
    };
    // chase: bounded wait until the policy group has published `need` steps (all lanes of the wave spin together)
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
      float xn[XK], mkn = 0.f;
      if (!CHASE && t < T) {  // next slot's inputs in flight behind this step's GEMMs
        load_x(t + 1, xn);
        mkn = b.masks[(size_t)(t + 1) * LA + rr];
      }
      f32x4 hin[4], hnew[4], n3[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) hin[m] = h[m] * mk;
      // fc1's k-loop has a run-time trip count: the prefetched registers go through this lane's own slots of the
      // wave's LDS slab (written and read by the same lane, no cross-lane ordering involved)
#pragma unroll
      for (int s = 0; s < XK; ++s)
        if (4 * s < tw.DP) slab[j * tw.DP + 4 * s + q] = xv[s];
      rnn_tower_fwd_lds(lw, tw, [&](int s) -> float { return slab[j * tw.DP + 4 * s + q]; }, hin, hnew, n3, j, q);
      float v[1];
      head_T<1>(lw + tw.W3, lw + tw.b3, 1, n3, q, v);
      if (t == T) {
        if (ok && q == 0 && A.next_value != nullptr) A.next_value[row] = v[0];
        break;
      }
      if (ok && q == 0) A.value_preds[(size_t)t * LA + row] = v[0];
      if (CHASE) {  // slot t+1 exists once the policy group has finished step t; by now it usually has
        wait_for(t + 1);
        load_x(t + 1, xn);
        mkn = ld_agent(b.masks + (size_t)(t + 1) * LA + rr);
      }
#pragma unroll
      for (int m = 0; m < 4; ++m) h[m] = hnew[m] * mkn;  // rnn_states_critic[dones_env] = 0
      if (ok) {
#pragma unroll
        for (int m = 0; m < 4; ++m) *(f32x4*)(A.hc + ((size_t)(t + 1) * LA + row) * HID + 16 * m + 4 * q) = h[m];
      }
      mk = mkn;
#pragma unroll
      for (int s = 0; s < XK; ++s) xv[s] = xn[s];
    }
  }
}

// Single-agent device envs (synthetic fixed-step env, CartPole): one wave = one 16-env tile for the whole rollout - hidden
// state in registers, the tile's observations in a wave-private LDS slab, the env stepped by lane (j, q == 0) with the
// shared device functions of orl_env.h.  Waves are independent (no workgroup barrier after the tower image is staged);
// the critic follows in a second launch (rnn_rollout_critic_kernel) over the stored observations.
template <int NO, int HEAD, int ENV>
__global__ __launch_bounds__(256, 1) void rnn_rollout_single_policy_kernel(RnnRolloutArgs A) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const orl_buffer_ptrs& b = A.buf;
  const RnnLayout tl(A.pnet);
  const int D = tl.D;
  const RnnLds tw(D, A.pnet.n_out, HEAD == ORL_HEAD_GAUSSIAN);
  stage_rnn_tower(smem, A.ptheta, tl, tw, threadIdx.x, blockDim.x);
  __syncthreads();
  const float* lw = smem;
  const int wave = threadIdx.x >> 6, l = threadIdx.x & 63, j = l & 15, q = l >> 4, nwv = blockDim.x >> 6;
  float* slab = smem + tw.total + wave * (TILE_B * 64 + TILE_B);  // [16][DP] observations + [16] done flags
  float* s_done = slab + TILE_B * 64;
  const int N = b.N, T = b.T, DP = tw.DP;
  const int n_out = tl.n_out;
  const int a_w = (HEAD == ORL_HEAD_CATEGORICAL) ? 1 : n_out;
  const int tile = blockIdx.x * nwv + wave;
  const int env = tile * TILE_B + j;
  if (tile * TILE_B >= N) return;
  const bool ok = env < N;
  const int row = ok ? env : 0;
  const uint64_t rng0 = A.rng_step0 + (A.rng_dev ? *A.rng_dev : 0ull);
  for (int e = l; e < TILE_B * DP; e += 64) {
    const int jj = e / DP, k = e - jj * DP, nn = tile * TILE_B + jj;
    slab[e] = (nn < N && k < D) ? b.policy_obs[(size_t)nn * D + k] : 0.f;
  }
  f32x4 h[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) h[m] = *(const f32x4*)(A.hp + (size_t)row * HID + 16 * m + 4 * q);
  float mk = b.masks[row];
  wave_lds_fence();
  for (int t = 0; t < T; ++t) {
    f32x4 hin[4], hnew[4], n3[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) hin[m] = h[m] * mk;
    const float* xrow = slab + j * DP;
    rnn_tower_fwd_lds(lw, tw, [&](int s) -> float { return xrow[4 * s + q]; }, hin, hnew, n3, j, q);
    float hd[NO], act_o[NO], lp_o[NO];
    head_T<NO>(lw + tw.W3, lw + tw.b3, n_out, n3, q, hd);
    sample_head<NO, HEAD>(hd, n_out, lw + tw.logstd, nullptr, nullptr, A.deterministic, A.act_seed, (uint64_t)row,
                          rng0 + (uint64_t)t, act_o, lp_o);
    wave_lds_fence();  // every lane of the wave is done reading this step's observations
    if (q == 0 && ok) {
#pragma unroll
      for (int c = 0; c < NO; ++c)
        if (c < a_w) {
          A.actions[((size_t)t * N + row) * a_w + c] = act_o[c];
          A.logp[((size_t)t * N + row) * a_w + c] = lp_o[c];
        }
      float r;
      bool d;
      float* orow = slab + j * DP;  // the next observation lands in the slab (next step's fc1) and in slot t + 1
      env_step_one<ENV>(A.env_state, A.ep_stats, env, D, A.env_seed, A.world_length, A.env_step0 + (uint64_t)t,
                        ENV == ORL_ENV_SYNTH ? 0 : (int)act_o[0], orow, r, d);
      const size_t s1 = (size_t)(t + 1) * N + row;
      for (int k = 0; k < D; ++k) b.policy_obs[s1 * D + k] = orow[k];
      b.rewards[(size_t)t * N + row] = r;
      b.masks[s1] = d ? 0.f : 1.f;
      b.active_masks[s1] = 1.f;
      b.bad_masks[s1] = 1.f;
      if (b.K > 0 && b.action_masks != nullptr)
        for (int k = 0; k < b.K; ++k) b.action_masks[s1 * b.K + k] = 1.f;
      s_done[j] = d ? 1.f : 0.f;
    }
    wave_lds_fence();
    // masks[t + 1] = 0 and rnn_states = 0 where the env finished (onpolicy_driver.py:100-113)
    mk = (ok && s_done[j] != 0.f) ? 0.f : 1.f;
#pragma unroll
    for (int m = 0; m < 4; ++m) h[m] = hnew[m] * mk;
    if (ok) {
#pragma unroll
      for (int m = 0; m < 4; ++m) *(f32x4*)(A.hp + ((size_t)(t + 1) * N + row) * HID + 16 * m + 4 * q) = h[m];
    }
  }
}

template <int NO>
__global__ __launch_bounds__(192, 1) void rnn_rollout_mpe_policy_kernel(RnnRolloutArgs A) {
  rnn_rollout_policy_body<NO, false>(A, (int)blockIdx.x);
}

__global__ __launch_bounds__(128, 1) void rnn_rollout_critic_kernel(RnnRolloutArgs A) {
  rnn_rollout_critic_body<false>(A, (int)blockIdx.x, (int)gridDim.x);
}

// policy and critic workgroups interleaved, the critic one step behind its policy group
template <int NO>
__global__ __launch_bounds__(192, 1) void rnn_rollout_mpe_chase_kernel(RnnRolloutArgs A) {
  const int bid = (int)blockIdx.x >> 1;
  if ((blockIdx.x & 1) == 0) rnn_rollout_policy_body<NO, true>(A, bid);
  else rnn_rollout_critic_body<true>(A, bid, (int)gridDim.x >> 1);
}

#ifndef ORL_RNN_ROLLOUT_COOP
#define ORL_RNN_ROLLOUT_COOP 1  // 1: four cooperating waves per tile (round 5, default); 0: one wave per tile (rounds 3 - 4)
#endif
#include "orl_rnn_rollout_coop.h"

}  // namespace orl

using namespace orl;

extern "C" {

int orl_rnn_rollout_fused(const orl_net_desc* pnet, const float* ptheta, const orl_net_desc* cnet, const float* ctheta,
                          const orl_rnn_rollout_args* a, void* stream) {
  ORL_REQUIRE(pnet && ptheta && cnet && ctheta && a, "orl_rnn_rollout_fused: null argument");
  const orl_buffer_ptrs& b = a->buf;
  if (a->env_kind == ORL_ENV_SYNTH || a->env_kind == ORL_ENV_CARTPOLE) {
    // single-agent device envs: policy launch (one wave per 16-env tile) + the generic critic sweep
    ORL_REQUIRE(b.A == 1 && b.N > 0 && b.T > 0 && b.Dp >= 1 && b.Dp <= 64 && b.Dc == b.Dp && b.policy_obs &&
                    (b.critic_obs == b.policy_obs) && b.rewards && b.masks && b.bad_masks && b.active_masks,
                "orl_rnn_rollout_fused: single-agent envs need one shared observation array of width <= 64");
    ORL_REQUIRE(pnet->hidden == HID && cnet->hidden == HID && pnet->obs_dim == b.Dp && cnet->obs_dim == b.Dp &&
                    cnet->n_out == 1 && pnet->n_out >= 1 && pnet->n_out <= 16 &&
                    (pnet->head_kind == ORL_HEAD_CATEGORICAL || pnet->head_kind == ORL_HEAD_GAUSSIAN),
                "orl_rnn_rollout_fused: towers do not match the buffer (obs %d, hidden 64)", b.Dp);
    ORL_REQUIRE(a->env_kind != ORL_ENV_CARTPOLE || (b.Dp == 4 && pnet->head_kind == ORL_HEAD_CATEGORICAL && pnet->n_out == 2),
                "orl_rnn_rollout_fused: CartPole needs 4-d observations and a Discrete(2) head");
    ORL_REQUIRE(a->value_preds && a->actions && a->action_log_probs && a->rnn_states && a->rnn_states_critic &&
                    a->env_state && a->world_length > 0, "orl_rnn_rollout_fused: null buffer / env pointer");
    RnnRolloutArgs A;
    A.pnet = *pnet; A.cnet = *cnet; A.ptheta = ptheta; A.ctheta = ctheta; A.buf = b;
    A.value_preds = a->value_preds; A.actions = a->actions; A.logp = a->action_log_probs; A.hp = a->rnn_states;
    A.hc = a->rnn_states_critic; A.env_state = a->env_state; A.ep_stats = a->ep_stats; A.obs_p_out = nullptr;
    A.obs_c_out = nullptr; A.next_value = a->next_value; A.world_length = a->world_length;
    A.deterministic = a->deterministic; A.env_seed = a->env_seed; A.act_seed = a->act_seed; A.rng_step0 = a->rng_step0;
    A.rng_dev = (const unsigned long long*)a->rng_step_dev; A.flags = nullptr; A.env_step0 = a->env_step0;
    hipStream_t s = (hipStream_t)stream;
    const bool gauss = pnet->head_kind == ORL_HEAD_GAUSSIAN;
    const RnnLds twp(b.Dp, pnet->n_out, gauss), twc(b.Dp, 1, false);
    const int nwv = 4;
    const size_t lds_p = (size_t)(twp.total + nwv * (TILE_B * 64 + TILE_B)) * sizeof(float);
    const size_t lds_c = (size_t)(twc.total + 3 * TILE_B * 64) * sizeof(float);
    ORL_REQUIRE(lds_p <= 160 * 1024 && lds_c <= 160 * 1024, "orl_rnn_rollout_fused: tower image exceeds the LDS");
    const int n_tiles = (b.N + TILE_B - 1) / TILE_B;
    const dim3 grid((unsigned)((n_tiles + nwv - 1) / nwv));
    const int no = pnet->n_out;
#define ORL_RS(NOX, HDX, ENVX)                                                                                              \
  do {                                                                                                                      \
    (void)hipFuncSetAttribute((const void*)rnn_rollout_single_policy_kernel<NOX, HDX, ENVX>,                                \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_p);                                      \
    hipLaunchKernelGGL((rnn_rollout_single_policy_kernel<NOX, HDX, ENVX>), grid, dim3(64 * nwv), lds_p, s, A);             \
  } while (0)
    if (a->env_kind == ORL_ENV_CARTPOLE) ORL_RS(2, ORL_HEAD_CATEGORICAL, ORL_ENV_CARTPOLE);
    else if (!gauss) {
      if (no <= 2) ORL_RS(2, ORL_HEAD_CATEGORICAL, ORL_ENV_SYNTH);
      else if (no <= 8) ORL_RS(8, ORL_HEAD_CATEGORICAL, ORL_ENV_SYNTH);
      else ORL_RS(16, ORL_HEAD_CATEGORICAL, ORL_ENV_SYNTH);
    } else {
      if (no <= 2) ORL_RS(2, ORL_HEAD_GAUSSIAN, ORL_ENV_SYNTH);
      else if (no <= 8) ORL_RS(8, ORL_HEAD_GAUSSIAN, ORL_ENV_SYNTH);
      else ORL_RS(16, ORL_HEAD_GAUSSIAN, ORL_ENV_SYNTH);
    }
#undef ORL_RS
    int rc = launch_status("orl_rnn_rollout_fused(policy, single-agent)");
    if (rc) return rc;
    (void)hipFuncSetAttribute((const void*)rnn_rollout_critic_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_c);
    int cg = (n_tiles + 1) / 2;
    if (cg > 1024) cg = 1024;
    hipLaunchKernelGGL(rnn_rollout_critic_kernel, dim3(cg), dim3(128), lds_c, s, A);
    return launch_status("orl_rnn_rollout_fused(critic)");
  }
  ORL_REQUIRE(a->env_kind == ORL_ENV_MPE_SPREAD, "orl_rnn_rollout_fused: env kind %d not built (MPE simple_spread, synthetic, CartPole)",
              a->env_kind);
  ORL_REQUIRE(b.A == MPE_A && b.Dp == MPE_OBS && b.Dc == MPE_COBS && b.N > 0 && b.T > 0,
              "orl_rnn_rollout_fused: buffer (A %d, Dp %d, Dc %d) is not simple_spread's (3, 18, 54)", b.A, b.Dp, b.Dc);
  ORL_REQUIRE(pnet->hidden == HID && cnet->hidden == HID && pnet->obs_dim == MPE_OBS && cnet->obs_dim == MPE_COBS &&
                  pnet->head_kind == ORL_HEAD_CATEGORICAL && pnet->n_out == 5 && cnet->n_out == 1,
              "orl_rnn_rollout_fused: towers do not match simple_spread (obs 18 / 54, Discrete(5), hidden 64)");
  ORL_REQUIRE(b.policy_obs && b.critic_obs && b.critic_obs != b.policy_obs && b.rewards && b.masks && b.bad_masks &&
                  b.active_masks && a->value_preds && a->actions && a->action_log_probs && a->rnn_states &&
                  a->rnn_states_critic && a->env_state && a->world_length > 0,
              "orl_rnn_rollout_fused: null buffer / env pointer");
  RnnRolloutArgs A;
  A.pnet = *pnet; A.cnet = *cnet; A.ptheta = ptheta; A.ctheta = ctheta; A.buf = b;
  A.value_preds = a->value_preds; A.actions = a->actions; A.logp = a->action_log_probs; A.hp = a->rnn_states;
  A.hc = a->rnn_states_critic; A.env_state = a->env_state; A.ep_stats = a->ep_stats; A.obs_p_out = a->obs_policy_out;
  A.obs_c_out = a->obs_critic_out; A.next_value = a->next_value; A.world_length = a->world_length;
  A.deterministic = a->deterministic; A.env_seed = a->env_seed; A.act_seed = a->act_seed; A.rng_step0 = a->rng_step0;
  A.rng_dev = (const unsigned long long*)a->rng_step_dev;
  A.flags = a->sync_flags; A.env_step0 = 0;
  hipStream_t s = (hipStream_t)stream;
  const RnnLds twp(MPE_OBS, 5, false), twc(MPE_COBS, 1, false);
  const size_t lds_p = (size_t)(twp.total + MPE_A * TILE_B * OBS_S + MPE_A * TILE_B + 2 * TILE_B) * sizeof(float);
  const size_t lds_c = (size_t)(twc.total + 3 * TILE_B * 64) * sizeof(float);
  ORL_REQUIRE(lds_p <= 160 * 1024 && lds_c <= 160 * 1024, "orl_rnn_rollout_fused: tower image exceeds the LDS");
  const int n_groups = (b.N + TILE_B - 1) / TILE_B;
#if ORL_RNN_ROLLOUT_COOP
  {
    const RnnLds twc2(MPE_COBS, 1, false, false, false, true);
    const size_t lp = lds_p + (size_t)(MPE_A * 2 * XCH + 4) * sizeof(float);  // + exchange buffers and counters
    const size_t lc = (size_t)(twc2.total + 3 * 2 * XCH + 4) * sizeof(float);
    ORL_REQUIRE(lp <= 160 * 1024 && lc <= 160 * 1024, "orl_rnn_rollout_fused: tower image exceeds the LDS");
    if (a->sync_flags != nullptr) {
      if (hipMemsetAsync(a->sync_flags, 0, (size_t)n_groups * sizeof(int), s) != hipSuccess)
        return fail(ORL_E_INVALID, "orl_rnn_rollout_fused: clearing the step counters failed");
      const size_t lds = lp > lc ? lp : lc;
      (void)hipFuncSetAttribute((const void*)rnn_rollout_mpe_chase_coop_kernel<8>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL((rnn_rollout_mpe_chase_coop_kernel<8>), dim3(2 * n_groups), dim3(COOP_THREADS), lds, s, A);
      return launch_status("orl_rnn_rollout_fused(chase)");
    }
    (void)hipFuncSetAttribute((const void*)rnn_rollout_mpe_policy_coop_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lp);
    hipLaunchKernelGGL((rnn_rollout_mpe_policy_coop_kernel<8>), dim3(n_groups), dim3(COOP_THREADS), lp, s, A);
    const int rc = launch_status("orl_rnn_rollout_fused(policy)");
    if (rc) return rc;
    (void)hipFuncSetAttribute((const void*)rnn_rollout_critic_coop_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lc);
    const int n_tiles = (b.N * b.A + TILE_B - 1) / TILE_B;
    int grid = (n_tiles + 2) / 3;
    if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(rnn_rollout_critic_coop_kernel, dim3(grid), dim3(COOP_THREADS), lc, s, A);
    return launch_status("orl_rnn_rollout_fused(critic)");
  }
#endif
  if (a->sync_flags != nullptr) {  // one launch, critic workgroups chase their policy workgroups
    // only the n_groups step counters are cleared: the error word flags[n_groups] is STICKY (caller-zeroed once, like
    // orl_comm's) - a timeout in rollout k must still be visible after rollout k + 1 when the host polls late
    if (hipMemsetAsync(a->sync_flags, 0, (size_t)n_groups * sizeof(int), s) != hipSuccess)
      return fail(ORL_E_INVALID, "orl_rnn_rollout_fused: clearing the step counters failed");
    const size_t lds = lds_p > lds_c ? lds_p : lds_c;
    (void)hipFuncSetAttribute((const void*)rnn_rollout_mpe_chase_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
    hipLaunchKernelGGL((rnn_rollout_mpe_chase_kernel<8>), dim3(2 * n_groups), dim3(192), lds, s, A);
    return launch_status("orl_rnn_rollout_fused(chase)");
  }
  {
    (void)hipFuncSetAttribute((const void*)rnn_rollout_mpe_policy_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds_p);
    hipLaunchKernelGGL((rnn_rollout_mpe_policy_kernel<8>), dim3(n_groups), dim3(192), lds_p, s, A);
    const int rc = launch_status("orl_rnn_rollout_fused(policy)");
    if (rc) return rc;
  }
  {
    (void)hipFuncSetAttribute((const void*)rnn_rollout_critic_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds_c);
    const int n_tiles = (b.N * b.A + TILE_B - 1) / TILE_B;
    int grid = (n_tiles + 1) / 2;
    if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(rnn_rollout_critic_kernel, dim3(grid), dim3(128), lds_c, s, A);
  }
  return launch_status("orl_rnn_rollout_fused(critic)");
}

}  // extern "C"
