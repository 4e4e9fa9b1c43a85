// orl_buffer.hip - replay-buffer side of the hot path for gfx950:
//   K5 insert + mask construction, K6 GAE/return reverse scan (+K7a advantage & statistics),
//   K7b advantage normalisation + update-record packing, K8 minibatch gather, keyed permutation,
//   ValueNorm state update, minibatch return moments.
// All of these are HBM / latency bound integer-and-fp32 streaming kernels: coalesced row
// accesses over the [T(+1), N*A] lane grid, LDS staging for the time scan, wave64 shuffles for
// reductions.  No MFMA here by design.
#include "orl_common.h"
#include "orl_perm.h"

namespace orl {

thread_local char g_err[512] = {0};

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

int launch_status(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------
// K6: GAE / return scan.  One workgroup = 64 lanes x all T steps, processed in reverse time chunks
// staged through LDS: the 4 waves stream a [TC x 64] tile of each input with every thread keeping
// several independent 256-byte-row loads in flight, then wave 0 runs the sequential recurrence out
// of LDS and writes returns / raw advantages as coalesced 256-byte rows.
// ------------------------------------------------------------------------------------------------
constexpr int GAE_LANES = 16;   // lanes (env x agent columns) per workgroup: 4096 lanes -> 256 workgroups
constexpr int GAE_TC = 128;     // time chunk held in LDS
constexpr int GAE_THREADS = 512;
constexpr int PACK_ROWS_MAX = 256;  // rows of records assembled in LDS per workgroup and trip (fewer for wide records)

struct VnCoef {
  float sd, mean;
  int on;
};

__device__ inline VnCoef vn_coef(const float* vn_state) {
#pragma clang fp contract(off)
  VnCoef c;
  c.on = vn_state != nullptr;
  c.sd = 1.f;
  c.mean = 0.f;
  if (c.on) {
    // ValueNorm.running_mean_var (valuenorm.py:45-52): clamp(debias, 1e-5), var >= 1e-2
    const float deb = fmaxf(vn_state[2], 1e-5f);
    const float m = vn_state[0] / deb;
    const float msq = vn_state[1] / deb;
    const float var = fmaxf(msq - m * m, 1e-2f);
    c.sd = sqrtf(var);
    c.mean = m;
  }
  return c;
}

__device__ inline float vn_denorm(const VnCoef& c, float v) {
#pragma clang fp contract(off)
  if (!c.on) return v;
  const float t = v * c.sd;
  return t + c.mean;
}

// Three phases per time chunk, because only the carry is sequential:
//   A  (all 256 threads, one (t, lane) element each per pass): global loads, ValueNorm denormalisation, delta and the
//      per-step coefficient -> LDS.  Everything that does not depend on the carry happens here, in parallel.
//   B  (GAE_LANES threads): the reverse recurrence proper - 2-5 dependent fp32 operations per step, operands from LDS,
//      in exactly the reference's operation order (bit-exact, contraction off).
//   C  (all threads, same element mapping as A so V'(t) and the active mask are still in registers): returns, raw
//      advantages, the statistics sums.
template <bool USE_GAE, bool PROPER>
__global__ __launch_bounds__(GAE_THREADS) void gae_scan_kernel(
    const float* __restrict__ rewards, float* __restrict__ value_preds, const float* __restrict__ masks,
    const float* __restrict__ bad_masks, const float* __restrict__ next_value, const float* __restrict__ vn_state,
    float* __restrict__ returns, int T, int L, float gamma, float gl, const float* __restrict__ active_masks,
    float* __restrict__ adv_raw, double* __restrict__ stat_partials) {
#pragma clang fp contract(off)
  // LDS tiles, index [tt][lane].  GAE: d = delta, c = gamma*lambda*m(t+1).  Returns: d = r, c = m(t+1), y = (1-bad)*V'.
  __shared__ float s_d[GAE_TC][GAE_LANES];
  __shared__ float s_c[GAE_TC][GAE_LANES];
  __shared__ float s_b[PROPER ? GAE_TC : 1][GAE_LANES];
  __shared__ float s_y[(PROPER && !USE_GAE) ? GAE_TC : 1][GAE_LANES];
  __shared__ double s_red[GAE_THREADS / 64][8];
  constexpr int PASSES = GAE_TC * GAE_LANES / GAE_THREADS;  // elements per thread and chunk
  constexpr int TSTEP = GAE_THREADS / GAE_LANES;            // tt stride between a thread's elements

  const int lane0 = blockIdx.x * GAE_LANES;
  const int tid = threadIdx.x;
  const int la = tid % GAE_LANES, tg = tid / GAE_LANES;
  const int l = lane0 + la;
  const bool lane_ok = l < L;
  const VnCoef vc = vn_coef(vn_state);
  const bool want_stats = stat_partials != nullptr;

  // slot T gets next_value first (replay_data.py:323/360/384/418)
  float nv = 0.f;
  if (lane_ok) nv = next_value[l];
  if (tid < GAE_LANES && lane_ok) {
    if (USE_GAE) value_preds[(size_t)T * L + l] = nv;
    else returns[(size_t)T * L + l] = nv;
  }

  float carry = USE_GAE ? 0.f : nv;  // gae (USE_GAE) or returns[t+1] (otherwise); live in threads < GAE_LANES
  double s_all = 0, q_all = 0, n_all = 0, s_act = 0, q_act = 0, n_act = 0, s_ret = 0, q_ret = 0;

  for (int t_hi = T; t_hi > 0; t_hi -= GAE_TC) {
    const int t_lo = (t_hi - GAE_TC > 0) ? t_hi - GAE_TC : 0;
    const int nt = t_hi - t_lo;
    float v_cur[PASSES], act[PASSES];
    // ---- A ----
    float r_[PASSES], v_[PASSES], m_[PASSES], vn_[PASSES], b_[PASSES];
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const int tt = tg + p * TSTEP;
      r_[p] = v_[p] = m_[p] = vn_[p] = 0.f; b_[p] = 1.f; act[p] = 0.f;
      if (tt < nt && lane_ok) {
        const size_t t = (size_t)(t_lo + tt);
        r_[p] = rewards[t * L + l];
        v_[p] = value_preds[t * L + l];
        m_[p] = masks[(t + 1) * L + l];
        // V[t+1]: slot T is next_value (written above by another thread - read the source instead)
        if (USE_GAE) vn_[p] = ((int)t + 1 == T) ? nv : value_preds[(t + 1) * L + l];
        if (PROPER) b_[p] = bad_masks[(t + 1) * L + l];
        if (want_stats) act[p] = active_masks[t * L + l];
      }
    }
    __syncthreads();  // the previous chunk's phase C is done with s_d
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const int tt = tg + p * TSTEP;
      v_cur[p] = vn_denorm(vc, v_[p]);
      if (tt < nt) {
        if (USE_GAE) {
          const float v_nxt = vn_denorm(vc, vn_[p]);
          // delta = r + gamma * V'(t+1) * m(t+1) - V'(t)      (replay_data.py:390-396)
          const float a0 = gamma * v_nxt;
          const float a1 = a0 * m_[p];
          const float a2 = r_[p] + a1;
          s_d[tt][la] = a2 - v_cur[p];
          s_c[tt][la] = gl * m_[p];
          if (PROPER) s_b[tt][la] = b_[p];
        } else {
          s_d[tt][la] = r_[p];
          s_c[tt][la] = m_[p];
          if (PROPER) {
            s_b[tt][la] = b_[p];
            s_y[tt][la] = (1.f - b_[p]) * v_cur[p];
          }
        }
      }
    }
    __syncthreads();
    // ---- B ----
    if (tid < GAE_LANES) {
#pragma unroll 8
      for (int tt = nt - 1; tt >= 0; --tt) {
        if (USE_GAE) {
          // gae = delta + gamma*lambda * m(t+1) * gae        (:397-400)
          const float b1 = s_c[tt][la] * carry;
          float gae = s_d[tt][la] + b1;
          if (PROPER) gae = gae * s_b[tt][la];  // (:341/357)
          carry = gae;
        } else {
          // returns[t] = returns[t+1]*gamma*m(t+1) + r  [ *bad + (1-bad)*V'(t) ]  (:365-381, :420-423)
          const float c0 = carry * gamma;
          const float c1 = c0 * s_c[tt][la];
          float x = c1 + s_d[tt][la];
          if (PROPER) {
            const float y0 = x * s_b[tt][la];
            x = y0 + s_y[tt][la];
          }
          carry = x;
        }
        s_d[tt][la] = carry;
      }
    }
    __syncthreads();
    // ---- C ----
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const int tt = tg + p * TSTEP;
      if (tt < nt && lane_ok) {
        const size_t t = (size_t)(t_lo + tt);
        const float g = s_d[tt][la];
        const float ret = USE_GAE ? g + v_cur[p] : g;
        returns[t * L + l] = ret;
        if (adv_raw != nullptr || want_stats) {
          const float adv = ret - v_cur[p];  // ppo.py:394-400
          if (adv_raw != nullptr) adv_raw[t * L + l] = adv;
          if (want_stats) {
            const double a = (double)adv;
            s_all += a; q_all += a * a; n_all += 1.0;
            if (act[p] != 0.f) { s_act += a; q_act += a * a; n_act += 1.0; }
            const double rr = (double)ret;
            s_ret += rr; q_ret += rr * rr;
          }
        }
      }
    }
  }
  if (want_stats) {
    s_all = wave_sum(s_all); q_all = wave_sum(q_all); n_all = wave_sum(n_all);
    s_act = wave_sum(s_act); q_act = wave_sum(q_act); n_act = wave_sum(n_act);
    s_ret = wave_sum(s_ret); q_ret = wave_sum(q_ret);
    if ((tid & 63) == 0) {
      double* p = s_red[tid >> 6];
      p[0] = s_all; p[1] = q_all; p[2] = n_all; p[3] = s_act; p[4] = q_act; p[5] = n_act; p[6] = s_ret; p[7] = q_ret;
    }
    __syncthreads();
    if (tid < 8) {
      double x = 0;
      for (int w = 0; w < GAE_THREADS / 64; ++w) x += s_red[w][tid];
      stat_partials[(size_t)blockIdx.x * 8 + tid] = x;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// K7b: advantage normalisation + record packing.  Every workgroup re-reduces the (few) partial
// rows with wave shuffles, then streams its slice of the T*L samples.
// ------------------------------------------------------------------------------------------------
struct AdvCoef {
  float m1, s1;  // use_adv_normalize stage
  float m2, s2;  // active-mask nan-stats stage
};

__device__ inline void reduce_partials(const double* __restrict__ partials, int n, double out[8]) {
  __shared__ double s_red[4][8];
  double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] += partials[(size_t)i * 8 + k];
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = wave_sum(acc[k]);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int k = 0; k < 8; ++k) s_red[w][k] = acc[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    double s = 0;
    for (int ww = 0; ww < (int)(blockDim.x >> 6); ++ww) s += s_red[ww][k];
    out[k] = s;
  }
}

// NV = source elements per thread and slab that the batched-load path of the record assembly holds in registers (4: narrow
// records such as configuration 2's; 20: observations up to 20 wide at 256 rows per tile, wider ones at the smaller tiles
// their records get); shapes beyond NV take the slab-by-slab loops.
template <int NV>
__global__ __launch_bounds__(256, 2) void adv_normalize_pack_kernel(float* __restrict__ adv,
                                                                 const double* __restrict__ partials, int n_partials,
                                                                 long long M, int L, int use_adv_normalize,
                                                                 double* __restrict__ stats_out, orl_pack_src src,
                                                                 float* __restrict__ records, int R, int PACK_ROWS) {
#pragma clang fp contract(off)
  double st[8];
  reduce_partials(partials, n_partials, st);
  if (blockIdx.x == 0 && threadIdx.x == 0 && stats_out != nullptr) {
    for (int k = 0; k < 8; ++k) stats_out[k] = st[k];
    // {sum ret, sum ret^2, count}: the moments ValueNorm.update takes when one minibatch is the whole batch
    stats_out[8] = st[6]; stats_out[9] = st[7]; stats_out[10] = st[2];
  }
  AdvCoef c;
  // stage 1 (optional): numpy mean()/std() over ALL entries (ppo.py:402-403)
  double mean_all = st[0] / st[2];
  double var_all = st[1] / st[2] - mean_all * mean_all;
  if (var_all < 0) var_all = 0;
  double mean_act = st[3] / st[5];
  double var_act = st[4] / st[5] - mean_act * mean_act;
  if (var_act < 0) var_act = 0;
  double std_act = sqrt(var_act);
  if (use_adv_normalize) {
    c.m1 = (float)mean_all;
    c.s1 = (float)sqrt(var_all) + 1e-5f;
    // statistics of y = (x - m1)/s1 over the active subset follow analytically
    const double m1 = (double)c.m1, s1 = (double)c.s1;
    mean_act = (mean_act - m1) / s1;
    std_act = std_act / s1;
  } else {
    c.m1 = 0.f;
    c.s1 = 1.f;
  }
  // stage 2 (always): nanmean / nanstd over active entries (ppo.py:405-409)
  c.m2 = (float)mean_act;
  c.s2 = (float)std_act + 1e-5f;

  const long long stride = (long long)gridDim.x * blockDim.x;
  if (records == nullptr) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < M; i += stride) {
      float x = adv[i];
      if (use_adv_normalize) x = (x - c.m1) / c.s1;
      adv[i] = (x - c.m2) / c.s2;
    }
    return;
  }
  // fused: records are built PACK_ROWS rows at a time in LDS - every source array is read as the contiguous slab it is
  // (coalesced), the finished [PACK_ROWS][R] tile leaves as contiguous float4 stores
  extern __shared__ __attribute__((aligned(16))) float s_tile[];
  const int Dp = src.Dp, Dc = src.Dc, a = src.a, K = src.K;
  const int o_co = Dp, o_ac = o_co + Dc, o_lp = o_ac + a, o_adv = o_lp + a, o_vp = o_adv + 1, o_rt = o_vp + 1,
            o_am = o_rt + 1, o_mk = o_am + 1, o_end = o_mk + K;
  const long long n_tiles = (M + PACK_ROWS - 1) / PACK_ROWS;
  const int tid = threadIdx.x, nth = blockDim.x;
  auto slab = [&](const float* __restrict__ p, int w, int off, long long row0, int nrow) {  // [nrow][w] -> tile columns
    if (w == 1) {
      if (tid < nrow) s_tile[tid * R + off] = p[row0 + tid];
      return;
    }
    for (int e = tid; e < nrow * w; e += nth) {
      const int rr = e / w, c = e - rr * w;
      s_tile[rr * R + off + c] = p[row0 * w + e];
    }
  };
  // Every source slab of a tile <= NV elements per thread: all of a tile's global loads are issued into registers BEFORE
  // the first LDS write - one HBM round trip per tile.  The slab-by-slab form below is a chain of (4 loads -> waits ->
  // ds_writes) per slab: 5 serialized round trips per tile at configuration 2, 16 at configuration 5's 52-float records.
  int wmax = Dp > Dc ? Dp : Dc;
  wmax = wmax > a ? wmax : a;
  wmax = wmax > K ? wmax : K;
  constexpr int NA = NV < 16 ? NV : 16;  // registers for the action-side slabs (actions, log-probs, masks)
  const int wa = a > K ? a : K;
  const bool narrow = (long long)PACK_ROWS * wmax <= (long long)NV * nth && (long long)PACK_ROWS * wa <= (long long)NA * nth &&
                      PACK_ROWS <= nth;
  for (long long t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    const long long row0 = t * PACK_ROWS;
    const int nrow = (M - row0) < PACK_ROWS ? (int)(M - row0) : PACK_ROWS;
    if (narrow) {
      float v_po[NV], v_co[NV], v_ac[NA], v_lp[NA], v_mk[NA];
      auto ld = [&](const float* __restrict__ p, int w, auto& v) {
        constexpr int N = sizeof(v) / sizeof(float);
        const float* __restrict__ pb = p + row0 * w;  // wave-uniform base + a 32-bit lane offset: one address register per load
        const int n = nrow * w;
#pragma unroll
        for (int k = 0; k < N; ++k) {
          const int e = tid + k * nth;
          v[k] = e < n ? pb[e] : 0.f;
        }
      };
      auto st = [&](int w, int off, const auto& v) {
        constexpr int N = sizeof(v) / sizeof(float);
#pragma unroll
        for (int k = 0; k < N; ++k) {
          const int e = tid + k * nth;
          if (e < nrow * w) {
            const int rr = e / w;
            s_tile[rr * R + off + (e - rr * w)] = v[k];
          }
        }
      };
      const bool has_mk = K > 0 && src.action_masks != nullptr;
      const long long row = row0 + (tid < nrow ? tid : 0);  // clamped: the loads below are unconditional (one batch)
      float x = adv[row];
      const float pv = src.value_preds[row], rt = src.returns[row], am = src.active_masks[row];
      ld(src.policy_obs, Dp, v_po);
      ld(src.critic_obs, Dc, v_co);
      ld(src.actions, a, v_ac);
      ld(src.action_log_probs, a, v_lp);
      if (has_mk) ld(src.action_masks, K, v_mk);
      asm volatile("" ::"v"(x), "v"(pv), "v"(rt), "v"(am));  // keep them with the batch: one round trip per tile
      st(Dp, 0, v_po);
      st(Dc, o_co, v_co);
      st(a, o_ac, v_ac);
      st(a, o_lp, v_lp);
      if (has_mk) st(K, o_mk, v_mk);
      else
        for (int e = tid; e < nrow * K; e += nth) s_tile[(e / K) * R + o_mk + (e % K)] = 1.f;
      if (tid < nrow) {
        if (use_adv_normalize) x = (x - c.m1) / c.s1;
        const float v = (x - c.m2) / c.s2;
        adv[row0 + tid] = v;
        float* o = s_tile + tid * R;
        o[o_adv] = v;
        o[o_vp] = pv;
        o[o_rt] = rt;
        o[o_am] = am;
        for (int k = o_end; k < R; ++k) o[k] = 0.f;
      }
    } else {
    slab(src.policy_obs, Dp, 0, row0, nrow);
    slab(src.critic_obs, Dc, o_co, row0, nrow);
    slab(src.actions, a, o_ac, row0, nrow);
    slab(src.action_log_probs, a, o_lp, row0, nrow);
    if (tid < nrow) {
      const long long row = row0 + tid;
      float x = adv[row];
      if (use_adv_normalize) x = (x - c.m1) / c.s1;
      const float v = (x - c.m2) / c.s2;
      adv[row] = v;
      float* o = s_tile + tid * R;
      o[o_adv] = v;
      o[o_vp] = src.value_preds[row];
      o[o_rt] = src.returns[row];
      o[o_am] = src.active_masks[row];
      for (int k = o_end; k < R; ++k) o[k] = 0.f;
    }
    if (K > 0) {
      if (src.action_masks != nullptr) slab(src.action_masks, K, o_mk, row0, nrow);
      else
        for (int e = tid; e < nrow * K; e += nth) s_tile[(e / K) * R + o_mk + (e % K)] = 1.f;
    }
    }
    __syncthreads();
    {
      const int n4 = nrow * (R >> 2);
      float4* dst = (float4*)(records + row0 * R);
      const float4* sv = (const float4*)s_tile;
      for (int e = tid; e < n4; e += nth) dst[e] = sv[e];
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void buffer_insert_kernel(orl_buffer_ptrs b, int step,
                                                            const float* __restrict__ nobs_p,
                                                            const float* __restrict__ nobs_c,
                                                            const float* __restrict__ rew,
                                                            const uint8_t* __restrict__ dones,
                                                            const uint8_t* __restrict__ bad,
                                                            const float* __restrict__ namask,
                                                            float* __restrict__ h_policy, float* __restrict__ h_critic,
                                                            int H) {
  const int LA = b.N * b.A;
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t s1 = (size_t)(step + 1);
  for (long long i = i0; i < (long long)LA * b.Dp; i += stride) b.policy_obs[s1 * LA * b.Dp + i] = nobs_p[i];
  if (b.critic_obs != b.policy_obs || nobs_c != nobs_p) {
    for (long long i = i0; i < (long long)LA * b.Dc; i += stride) b.critic_obs[s1 * LA * b.Dc + i] = nobs_c[i];
  }
  if (b.action_masks != nullptr && namask != nullptr) {
    for (long long i = i0; i < (long long)LA * b.K; i += stride) b.action_masks[s1 * LA * b.K + i] = namask[i];
  }
  for (long long i = i0; i < LA; i += stride) {
    const int n = (int)(i / b.A);
    bool all_done = true;
    for (int a = 0; a < b.A; ++a) all_done = all_done && (dones[n * b.A + a] != 0);
    const bool d = dones[i] != 0;
    b.rewards[(size_t)step * LA + i] = rew[i];
    b.masks[s1 * LA + i] = all_done ? 0.f : 1.f;                 // onpolicy_driver.py:110-113
    b.active_masks[s1 * LA + i] = (d && !all_done) ? 0.f : 1.f;  // :119-125
    b.bad_masks[s1 * LA + i] = (bad != nullptr && bad[i] != 0) ? 0.f : 1.f;  // :126-138
  }
  // recurrent: rnn_states[dones_env == True] = 0 (onpolicy_driver.py:100-109) as the product with masks[step+1],
  // in place on slot step+1 where the act kernel left the new hidden states
  if (h_policy != nullptr) {
    for (long long i = i0; i < (long long)LA * H; i += stride) {
      const int n = (int)((i / H) / b.A);
      bool all_done = true;
      for (int a = 0; a < b.A; ++a) all_done = all_done && (dones[n * b.A + a] != 0);
      const float m = all_done ? 0.f : 1.f;
      h_policy[i] = h_policy[i] * m;
      if (h_critic != nullptr) h_critic[i] = h_critic[i] * m;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// ReplayData.after_update (replay_data.py:286-318): slot T -> slot 0 of every per-step array, one launch instead of
// one copy per array.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void multi_copy_kernel(orl_copy_desc d) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  for (int k = 0; k < d.count; ++k) {
    const float* __restrict__ src = d.src[k];
    float* __restrict__ dst = d.dst[k];
    for (long long i = i0; i < d.n[k]; i += stride) dst[i] = src[i];
  }
}

// ------------------------------------------------------------------------------------------------
// K8: multi-array row gather.  One thread per output float; descriptors live in kernel args.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gather_kernel(orl_gather_desc d, const int64_t* __restrict__ idx, int n_rows,
                                                     int total_w) {
  const long long total = (long long)n_rows * total_w;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    const int row = (int)(e / total_w);
    int col = (int)(e - (long long)row * total_w);
    const long long src_row = idx[row];
    for (int k = 0; k < d.count; ++k) {
      const int w = d.width[k];
      if (col < w) {
        d.dst[k][(size_t)row * w + col] = d.src[k][(size_t)src_row * w + col];
        break;
      }
      col -= w;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// ValueNorm.update (valuenorm.py:58-77) from batch sums, and minibatch return moments.
// ------------------------------------------------------------------------------------------------
__global__ void valuenorm_update_kernel(float* __restrict__ vn, const double* __restrict__ mom, float beta, float omw) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  valuenorm_update_body(vn, mom, beta, omw);
}

__global__ __launch_bounds__(256) void perm_feistel_kernel(PermJob J) { perm_job_block(J, blockIdx.x, gridDim.x); }

__global__ __launch_bounds__(256) void moments_partial_kernel(const float* __restrict__ records, int R, int col,
                                                              const int64_t* __restrict__ idx, int mb,
                                                              double* __restrict__ scratch) {
  double s = 0, q = 0;
  const int stride = gridDim.x * blockDim.x;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < mb; i += stride) {
    const long long r = idx != nullptr ? idx[i] : i;
    const double x = (double)records[(size_t)r * R + col];
    s += x;
    q += x * x;
  }
  s = wave_sum(s);
  q = wave_sum(q);
  __shared__ double sh[4][2];
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { sh[w][0] = s; sh[w][1] = q; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double ss = 0, qq = 0;
    for (int k = 0; k < 4; ++k) { ss += sh[k][0]; qq += sh[k][1]; }
    scratch[blockIdx.x * 2 + 0] = ss;
    scratch[blockIdx.x * 2 + 1] = qq;
  }
}

__global__ __launch_bounds__(64) void moments_final_kernel(const double* __restrict__ scratch, int nb, int mb,
                                                           double* __restrict__ moments) {
  double s = 0, q = 0;
  for (int i = threadIdx.x; i < nb; i += 64) { s += scratch[i * 2]; q += scratch[i * 2 + 1]; }
  s = wave_sum(s);
  q = wave_sum(q);
  if (threadIdx.x == 0) { moments[0] = s; moments[1] = q; moments[2] = (double)mb; }
}

// ------------------------------------------------------------------------------------------------
// K7a stand-alone: adv_raw = returns[:-1] - denorm(value_preds[:-1]) + per-block statistics, for
// callers that did not take the fused outputs of the GAE scan (ppo.py:384-400).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adv_stats_kernel(const float* __restrict__ returns,
                                                        const float* __restrict__ value_preds,
                                                        const float* __restrict__ active_masks,
                                                        const float* __restrict__ vn_state, long long M,
                                                        float* __restrict__ adv_raw, double* __restrict__ partials) {
#pragma clang fp contract(off)
  const VnCoef vc = vn_coef(vn_state);
  double s_all = 0, q_all = 0, n_all = 0, s_act = 0, q_act = 0, n_act = 0, s_ret = 0, q_ret = 0;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < M; i += stride) {
    const float ret = returns[i];
    const float adv = ret - vn_denorm(vc, value_preds[i]);
    adv_raw[i] = adv;
    const double a = (double)adv;
    s_all += a; q_all += a * a; n_all += 1.0;
    if (active_masks[i] != 0.f) { s_act += a; q_act += a * a; n_act += 1.0; }
    const double rr = (double)ret;
    s_ret += rr; q_ret += rr * rr;
  }
  double v[8] = {s_all, q_all, n_all, s_act, q_act, n_act, s_ret, q_ret};
  __shared__ double sh[4][8];
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = wave_sum(v[k]);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int k = 0; k < 8; ++k) sh[w][k] = v[k];
  }
  __syncthreads();
  if (threadIdx.x < 8) {
    double t = 0;
    for (int ww = 0; ww < 4; ++ww) t += sh[ww][threadIdx.x];
    partials[(size_t)blockIdx.x * 8 + threadIdx.x] = t;
  }
}

constexpr int MOM_BLOCKS = 256;

}  // namespace orl

using namespace orl;

extern "C" {

int orl_version(void) { return ORL_VERSION; }
int orl_build_experiments(void) { return ORL_BUILD_EXPERIMENTS; }
int orl_tower_split_terms(void) { return ORL_TOWER_F16 ? 2 : 3; }

int orl_abi_struct_size(int which) {
  switch (which) {
    case 0: return (int)sizeof(orl_net_desc);
    case 1: return (int)sizeof(orl_pack_src);
    case 2: return (int)sizeof(orl_buffer_ptrs);
    case 3: return (int)sizeof(orl_copy_desc);
    case 4: return (int)sizeof(orl_gather_desc);
    case 5: return (int)sizeof(orl_ppo_hparams);
    case 6: return (int)sizeof(orl_adam_state);
    case 7: return (int)sizeof(orl_rollout_args);
    case 8: return (int)sizeof(orl_rnn_batch);
    case 9: return (int)sizeof(orl_rnn_rollout_args);
    case 10: return (int)sizeof(orl_gen_mlp_desc);
    case 11: return (int)sizeof(orl_gt_desc);
    case 12: return (int)sizeof(orl_gt_loss);
    default: return ORL_E_INVALID;
  }
}
const char* orl_last_error_string(void) { return g_err; }

int orl_param_count(const orl_net_desc* net) {
  if (!net) return ORL_E_INVALID;
  return TowerLayout(*net).total;
}

int orl_raw_grad_count(const orl_net_desc* net) {
  if (!net) return ORL_E_INVALID;
  return RawLayout(*net).total;
}

int orl_gae_max_partials(int T, int L) {
  (void)T;
  return (L + GAE_LANES - 1) / GAE_LANES;
}

int orl_gae_scan(const float* rewards, float* value_preds, const float* masks, const float* bad_masks,
                 const float* next_value, const float* vn_state, float* returns, int T, int L, double gamma,
                 double gae_lambda, int flags, const float* active_masks, float* adv_raw, double* stat_partials,
                 int* n_partials, void* stream) {
  ORL_REQUIRE(rewards && value_preds && masks && next_value && returns, "orl_gae_scan: null buffer pointer");
  ORL_REQUIRE(T > 0 && L > 0, "orl_gae_scan: T=%d L=%d must be positive", T, L);
  const bool use_gae = flags & 1, proper = flags & 2;
  ORL_REQUIRE(!proper || bad_masks, "orl_gae_scan: use_proper_time_limits needs bad_masks");
  ORL_REQUIRE(!stat_partials || active_masks, "orl_gae_scan: statistics need active_masks");
  const int grid = (L + GAE_LANES - 1) / GAE_LANES;
  const float g = (float)gamma;
  const float gl = (float)(gamma * gae_lambda);  // python computes gamma*gae_lambda in double first
  hipStream_t s = (hipStream_t)stream;
#define ORL_GAE_LAUNCH(UG, PR)                                                                                     \
  hipLaunchKernelGGL((gae_scan_kernel<UG, PR>), dim3(grid), dim3(GAE_THREADS), 0, s, rewards, value_preds, masks, \
                     bad_masks, next_value, vn_state, returns, T, L, g, gl, active_masks, adv_raw, stat_partials)
  if (use_gae && proper) ORL_GAE_LAUNCH(true, true);
  else if (use_gae) ORL_GAE_LAUNCH(true, false);
  else if (proper) ORL_GAE_LAUNCH(false, true);
  else ORL_GAE_LAUNCH(false, false);
#undef ORL_GAE_LAUNCH
  if (n_partials) *n_partials = grid;
  return launch_status("orl_gae_scan");
}

int orl_adv_stats(const float* returns, const float* value_preds, const float* active_masks, const float* vn_state,
                  int T, int L, float* adv_raw, double* stat_partials, int* n_partials, void* stream) {
  ORL_REQUIRE(returns && value_preds && active_masks && adv_raw && stat_partials, "orl_adv_stats: null pointer");
  ORL_REQUIRE(T > 0 && L > 0, "orl_adv_stats: bad geometry");
  const long long M = (long long)T * L;
  int grid = (int)((M + 255) / 256);
  const int cap = orl_gae_max_partials(T, L);
  if (grid > cap) grid = cap;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(adv_stats_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, returns, value_preds,
                     active_masks, vn_state, M, adv_raw, stat_partials);
  if (n_partials) *n_partials = grid;
  return launch_status("orl_adv_stats");
}

int orl_record_width(int Dp, int Dc, int a, int K) {
  const int w = Dp + Dc + 2 * a + 4 + K;
  return (w + 3) & ~3;
}

int orl_adv_normalize_pack(float* adv, const double* stat_partials, int n_partials, int T, int L,
                           int use_adv_normalize, double* stats_out, const orl_pack_src* src, float* records,
                           void* stream) {
  ORL_REQUIRE(adv && stat_partials && n_partials > 0, "orl_adv_normalize_pack: null adv/partials");
  ORL_REQUIRE(T > 0 && L > 0, "orl_adv_normalize_pack: bad geometry");
  ORL_REQUIRE((records == nullptr) == (src == nullptr), "orl_adv_normalize_pack: records and src go together");
  orl_pack_src s0 = {};
  int R = 0;
  if (src) {
    s0 = *src;
    ORL_REQUIRE(s0.policy_obs && s0.critic_obs && s0.actions && s0.action_log_probs && s0.value_preds &&
                    s0.returns && s0.active_masks,
                "orl_adv_normalize_pack: null source array");
    R = orl_record_width(s0.Dp, s0.Dc, s0.a, s0.K);
  }
  const long long M = (long long)T * L;
  int prow = PACK_ROWS_MAX;  // one LDS tile of prow records per workgroup and trip, at most 64 KiB
  while (prow > 16 && (size_t)prow * R * sizeof(float) > 64 * 1024) prow >>= 1;
  int wmax = s0.Dp > s0.Dc ? s0.Dp : s0.Dc;
  wmax = wmax > s0.a ? wmax : s0.a;
  wmax = wmax > s0.K ? wmax : s0.K;
  // wide observations (configuration 4's 54-wide share_obs at 128 rows): shrink the tile until every source slab fits the kernel's
  // batched-load path (<= 20 elements per thread: ONE HBM round trip per tile) instead of falling to the slab-by-slab loops (five
  // serialised round trips and run-time divisions: 70 us for 100 MB)
  while (records && prow > 32 && (long long)prow * wmax > 20LL * 256) prow >>= 1;
  long long work = records ? (M + prow - 1) / prow * 256 : M;
  int grid = (int)((work + 255) / 256);
  // every workgroup re-reduces the GAE partial rows first and then walks its tiles: 1024 workgroups measured best
  // (24.8 us at config 2; 38.6 us at 4096, 32 us at 512)
  const int cap = records ? 1024 : 4096;
  if (grid > cap) grid = cap;
  if (grid < 1) grid = 1;
  const size_t tile_bytes = records ? (size_t)prow * R * sizeof(float) : 0;
  if ((long long)prow * wmax <= 4LL * 256)
    hipLaunchKernelGGL(adv_normalize_pack_kernel<4>, dim3(grid), dim3(256), tile_bytes, (hipStream_t)stream, adv,
                       stat_partials, n_partials, M, L, use_adv_normalize, stats_out, s0, records, R, prow);
  else
    hipLaunchKernelGGL(adv_normalize_pack_kernel<20>, dim3(grid), dim3(256), tile_bytes, (hipStream_t)stream, adv,
                       stat_partials, n_partials, M, L, use_adv_normalize, stats_out, s0, records, R, prow);
  return launch_status("orl_adv_normalize_pack");
}

int orl_buffer_insert(const orl_buffer_ptrs* buf, int step, const float* next_policy_obs,
                      const float* next_critic_obs, const float* rewards, const uint8_t* dones,
                      const uint8_t* bad_transition, const float* next_action_masks, void* stream) {
  return orl_buffer_insert_rnn(buf, step, next_policy_obs, next_critic_obs, rewards, dones, bad_transition,
                               next_action_masks, nullptr, nullptr, 0, stream);
}

int orl_buffer_insert_rnn(const orl_buffer_ptrs* buf, int step, const float* next_policy_obs,
                          const float* next_critic_obs, const float* rewards, const uint8_t* dones,
                          const uint8_t* bad_transition, const float* next_action_masks, float* h_policy_next,
                          float* h_critic_next, int hidden, void* stream) {
  ORL_REQUIRE(buf && next_policy_obs && next_critic_obs && rewards && dones, "orl_buffer_insert: null pointer");
  ORL_REQUIRE(!h_policy_next || hidden > 0, "orl_buffer_insert_rnn: hidden size %d", hidden);
  ORL_REQUIRE(h_policy_next || !h_critic_next, "orl_buffer_insert_rnn: critic states without policy states");
  ORL_REQUIRE(step >= 0 && step < buf->T, "orl_buffer_insert: step %d outside [0,%d)", step, buf->T);
  ORL_REQUIRE(buf->policy_obs && buf->critic_obs && buf->rewards && buf->masks && buf->bad_masks && buf->active_masks,
              "orl_buffer_insert: null buffer array");
  int wmax = buf->Dp > buf->Dc ? buf->Dp : buf->Dc;
  if (h_policy_next && hidden > wmax) wmax = hidden;
  const long long work = (long long)buf->N * buf->A * wmax;
  int grid = (int)((work + 255) / 256);
  if (grid > 1024) grid = 1024;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(buffer_insert_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, *buf, step,
                     next_policy_obs, next_critic_obs, rewards, dones, bad_transition, next_action_masks,
                     h_policy_next, h_critic_next, hidden);
  return launch_status("orl_buffer_insert");
}

int orl_multi_copy(const orl_copy_desc* desc, void* stream) {
  ORL_REQUIRE(desc && desc->count >= 0 && desc->count <= ORL_COPY_MAX, "orl_multi_copy: bad descriptor");
  long long most = 0;
  for (int k = 0; k < desc->count; ++k) {
    ORL_REQUIRE(desc->src[k] && desc->dst[k] && desc->n[k] >= 0, "orl_multi_copy: bad entry %d", k);
    if (desc->n[k] > most) most = desc->n[k];
  }
  if (desc->count == 0 || most == 0) return 0;
  int grid = (int)((most + 255) / 256);
  if (grid > 1024) grid = 1024;
  hipLaunchKernelGGL(multi_copy_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, *desc);
  return launch_status("orl_multi_copy");
}

int orl_gather_minibatch(const orl_gather_desc* desc, const int64_t* idx, int n_rows, void* stream) {
  if (n_rows == 0) return 0;  // empty minibatch: nothing to do (pointers may legitimately be NULL)
  ORL_REQUIRE(desc && idx, "orl_gather_minibatch: null pointer");
  ORL_REQUIRE(desc->count > 0 && desc->count <= ORL_GATHER_MAX, "orl_gather_minibatch: count %d", desc->count);
  ORL_REQUIRE(n_rows > 0, "orl_gather_minibatch: negative n_rows");
  int total_w = 0;
  for (int k = 0; k < desc->count; ++k) {
    ORL_REQUIRE(desc->src[k] && desc->dst[k] && desc->width[k] > 0, "orl_gather_minibatch: bad descriptor %d", k);
    total_w += desc->width[k];
  }
  const long long work = (long long)n_rows * total_w;
  int grid = (int)((work + 255) / 256);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(gather_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, *desc, idx, n_rows, total_w);
  return launch_status("orl_gather_minibatch");
}

static int launch_perm(const char* what, int64_t* idx, int64_t n, uint64_t seed, uint64_t stream_id, float* vn,
                       const double* mom, double beta, void* stream) {
  ORL_REQUIRE(idx && n > 0, "%s: bad arguments", what);
  ORL_REQUIRE(n <= ((int64_t)1 << 62), "%s: n too large", what);
  const PermJob J = make_perm_job(idx, n, seed, stream_id, vn, mom, beta);
  int grid = (int)((n + 255) / 256);
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(perm_feistel_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, J);
  return launch_status(what);
}

int orl_perm_feistel(int64_t* idx, int64_t n, uint64_t seed, uint64_t stream_id, void* stream) {
  return launch_perm("orl_perm_feistel", idx, n, seed, stream_id, nullptr, nullptr, 0.0, stream);
}

int orl_perm_feistel_vn(int64_t* idx, int64_t n, uint64_t seed, uint64_t stream_id, float* vn_state,
                        const double* moments, double beta, void* stream) {
  ORL_REQUIRE(vn_state && moments, "orl_perm_feistel_vn: null pointer");
  return launch_perm("orl_perm_feistel_vn", idx, n, seed, stream_id, vn_state, moments, beta, stream);
}

int orl_valuenorm_update(float* vn_state, const double* moments, double beta, void* stream) {
  ORL_REQUIRE(vn_state && moments, "orl_valuenorm_update: null pointer");
  hipLaunchKernelGGL(valuenorm_update_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, vn_state, moments, (float)beta, (float)(1.0 - beta));
  return launch_status("orl_valuenorm_update");
}

int orl_minibatch_moments(const float* records, int rec_width, int ret_col, const int64_t* idx, int mb,
                          double* scratch, double* moments, void* stream) {
  ORL_REQUIRE(records && scratch && moments && mb > 0, "orl_minibatch_moments: bad arguments");
  ORL_REQUIRE(ret_col >= 0 && ret_col < rec_width, "orl_minibatch_moments: column outside record");
  int nb = (mb + 255) / 256;
  if (nb > MOM_BLOCKS) nb = MOM_BLOCKS;
  hipLaunchKernelGGL(moments_partial_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, records, rec_width,
                     ret_col, idx, mb, scratch);
  hipLaunchKernelGGL(moments_final_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, scratch, nb, mb, moments);
  return launch_status("orl_minibatch_moments");
}

}  // extern "C"
