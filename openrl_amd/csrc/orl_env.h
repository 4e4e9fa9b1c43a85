// orl_env.h - the device-resident single-agent envs (synthetic fixed-step env of SURVEY.md section 8d, CartPole-v1) as
// per-env device functions, shared by the fused rollout kernels (orl_act.hip: default towers, orl_gen_fused.hip: general
// towers) and the stand-alone env kernels: one definition, so every route steps an env with the same arithmetic and the
// same Philox streams.  Not part of the C ABI.
#pragma once
#include "orl_common.h"
#include "orl_mlp.h"

namespace orl {

// --------------------------------------------------------------------------------------------------
// Device-resident envs.  State lives in `env_state[N][W]`; one lane (q == 0) owns one env.
// --------------------------------------------------------------------------------------------------
constexpr int SYNTH_STATE_W = 4;     // {steps_in_episode, -, -, -}
constexpr int CARTPOLE_STATE_W = 8;  // {x, x_dot, theta, theta_dot, steps_in_episode, episodes, -, -}

// synthetic obs component block b (4 normals) for (env, global time t)
__device__ inline void synth_obs_block(uint64_t seed, uint32_t env, uint64_t t, uint32_t b, float (&o)[4]) {
  const u4 r = philox4x32_10(seed, env, 0x0B5E0000u + b, (uint32_t)t, (uint32_t)(t >> 32));
  box_muller(r.x, r.y, o[0], o[1]);
  box_muller(r.z, r.w, o[2], o[3]);
}
__device__ inline float synth_reward(uint64_t seed, uint32_t env, uint64_t t) {
  const u4 r = philox4x32_10(seed, env, 0x4E3A0000u, (uint32_t)t, (uint32_t)(t >> 32));
  return u01(r.x);
}

// CartPole-v1 (gymnasium/envs/classic_control/cartpole.py, euler integrator) in fp32.
__device__ inline void cartpole_reset(uint64_t seed, uint32_t env, uint32_t episode, float (&s)[4]) {
  const u4 r = philox4x32_10(seed, env, 0xCA470000u, episode, 0u);
  s[0] = u01(r.x) * 0.1f - 0.05f;
  s[1] = u01(r.y) * 0.1f - 0.05f;
  s[2] = u01(r.z) * 0.1f - 0.05f;
  s[3] = u01(r.w) * 0.1f - 0.05f;
}
// The step in two halves (round 6): everything that does not depend on the ACTION - the trigonometry of the pole angle and
// the terms built on it - and the rest.  The fused rollout computes the first half on a service wave while the policy is
// still working on the step's action (orl_rollout2.h); cartpole_step() is the composition, so every route steps an env
// through the same expressions.
struct CartPolePre {
  float costh, sinth, t1, den;  // cos / sin of theta, polemass_length * theta_dot^2 * sin, the angular denominator
};
__device__ inline CartPolePre cartpole_pre(const float (&s)[4]) {
  const float masspole = 0.1f, total_mass = 1.1f, length = 0.5f, polemass_length = 0.05f;
  CartPolePre p;
  p.costh = cosf(s[2]);
  p.sinth = sinf(s[2]);
  p.t1 = polemass_length * s[3] * s[3] * p.sinth;
  p.den = length * (4.0f / 3.0f - masspole * p.costh * p.costh / total_mass);
  return p;
}
__device__ inline bool cartpole_post(float (&s)[4], const CartPolePre& p, int action) {
  const float gravity = 9.8f, total_mass = 1.1f, polemass_length = 0.05f, force_mag = 10.0f, tau = 0.02f;
  const float force = action == 1 ? force_mag : -force_mag;
  const float temp = (force + p.t1) / total_mass;
  const float thetaacc = (gravity * p.sinth - p.costh * temp) / p.den;
  const float xacc = temp - polemass_length * thetaacc * p.costh / total_mass;
  s[0] = s[0] + tau * s[1];
  s[1] = s[1] + tau * xacc;
  s[2] = s[2] + tau * s[3];
  s[3] = s[3] + tau * thetaacc;
  const float th_lim = 12.0f * 2.0f * 3.14159265358979323846f / 360.0f;
  return (s[0] < -2.4f) || (s[0] > 2.4f) || (s[2] < -th_lim) || (s[2] > th_lim);
}
__device__ inline bool cartpole_step(float (&s)[4], int action) {
  const CartPolePre p = cartpole_pre(s);
  return cartpole_post(s, p, action);
}


// One env.step of env `n` at global time `tg` on the env's state row `st` / episode statistics `e` (register or memory
// copies): writes the next observation to obs_out[0..D), returns the reward and whether the episode ended.  The one
// definition of the envs' arithmetic: every route steps through it.
template <int ENV>
__device__ inline void env_step_state(float* __restrict__ st, float* __restrict__ e, int n, int D, uint64_t seed,
                                      int episode_limit, uint64_t tg, int action, float* __restrict__ obs_out, float& r,
                                      bool& d) {
  if (ENV == ORL_ENV_SYNTH) {
    r = synth_reward(seed, (uint32_t)n, tg);
    const float c = st[0] + 1.f;
    d = c >= (float)episode_limit;
    st[0] = d ? 0.f : c;
    for (int b = 0; b < (D + 3) / 4; ++b) {
      float o[4];
      synth_obs_block(seed, (uint32_t)n, tg + 1, (uint32_t)b, o);
      for (int k = 0; k < 4; ++k)
        if (4 * b + k < D) obs_out[4 * b + k] = o[k];
    }
  } else {
    float s[4] = {st[0], st[1], st[2], st[3]};
    const bool term = cartpole_step(s, action);
    const float steps = st[4] + 1.f;
    d = term || steps >= (float)episode_limit;
    r = 1.0f;
    st[4] = d ? 0.f : steps;
    if (d) {
      st[5] += 1.f;
      cartpole_reset(seed, (uint32_t)n, (uint32_t)st[5], s);
    }
    for (int k = 0; k < 4; ++k) { st[k] = s[k]; obs_out[k] = s[k]; }
  }
  if (e != nullptr) {
    e[0] += r; e[1] += 1.f;
    if (d) { e[2] += e[0]; e[3] += 1.f; e[0] = 0.f; e[1] = 0.f; }
  }
}

// the same on the state arrays in memory (the stand-alone env_step_kernel's body)
template <int ENV>
__device__ inline void env_step_one(float* __restrict__ env_state, float* __restrict__ ep_stats, int n, int D,
                                    uint64_t seed, int episode_limit, uint64_t tg, int action,
                                    float* __restrict__ obs_out, float& r, bool& d) {
  constexpr int W = ENV == ORL_ENV_SYNTH ? SYNTH_STATE_W : CARTPOLE_STATE_W;
  env_step_state<ENV>(env_state + (size_t)n * W, ep_stats != nullptr ? ep_stats + (size_t)n * 4 : nullptr, n, D, seed,
                      episode_limit, tg, action, obs_out, r, d);
}

}  // namespace orl
