// orl_gen.hip - the GENERAL tower path for gfx950: any hidden_size / layer_N / activation / feature LayerNorm,
// shared policy+value trunks, MultiDiscrete heads.
//
// The fused kernels of orl_ppo_tower.h / orl_act.hip cover the reference's DEFAULT tower (hidden 64, layer_N 1, ReLU,
// no feature norm: what every BASELINE.json configuration runs).  Everything else the reference's MLPBase / MLPLayer
// (openrl/modules/networks/utils/mlp.py:8-46,100-180), PolicyValueNetwork (policy_value_network.py:34-230) and
// ACTLayer (utils/act.py:14-172) can express runs here, layer by layer, with activations in HBM:
//
//   orl_gemm            fp32 MFMA GEMM  C[M,N] = sum_k A(m,k) B(k,n)  with arbitrary element strides, so one kernel
//                       serves x W^T (forward), dz W (dgrad) and dz^T x (wgrad; rows are K -> deterministic split-K)
//   orl_row_fwd         bias + activation + LayerNorm of one linear layer (nn.Sequential(Linear, act, LayerNorm))
//   orl_row_bwd         its backward: LayerNorm affine / LayerNorm / activation, plus the column sums of
//                       d gamma, d beta, d bias as per-workgroup partial rows (reduced by orl_ppo_reduce)
//   orl_gather_cols     minibatch rows of a column range of the packed records -> dense matrix
//   orl_gen_policy_loss PPO clipped surrogate + entropy for Categorical / DiagGaussian / MultiDiscrete heads -> d loss / d logits (already divided by the masked-mean denominators) + statistics
//   orl_gen_value_loss  clipped huber / mse value loss -> d loss / d value
//   orl_gen_sample      action sampling / mode + log-probs from logits (rollout side)
//   orl_gen_adam        global grad norm, clip_grad_norm_, Adam on one flat parameter vector
//
// The layer loop itself lives on the host (openrl_amd/modules/generic_net.py), exactly where the reference has it
// (nn.Module.forward).  Throughput is secondary here - the hot configurations never take this path.
#include <string.h>
#include "orl_common.h"
#include "orl_mlp.h"
#include "orl_gen_act.h"
#include "orl_gen_sample.h"
#include "orl_gen_loss.h"

namespace orl {

// ---------------------------------------------------------------------------------------------------- GEMM
constexpr int GB_M = 64, GB_N = 64, GB_K = 16, G_LD = 68;

// A_KC: A's k index is the contiguous one (sak == 1), else m is; B_NC: B's n index is contiguous (sbn == 1), else k.
// Only affects which way the 256 threads walk the tile when loading (coalescing); any strides are correct.
template <bool A_KC, bool B_NC>
__global__ __launch_bounds__(256) void gemm_kernel(const float* __restrict__ A, long long sam, long long sak,
                                                   const float* __restrict__ B, long long sbk, long long sbn,
                                                   float* __restrict__ C, long long ldc, int M, int N, int K,
                                                   int k_per_split, float* __restrict__ partials) {
  __shared__ float As[GB_K][G_LD];  // As[k][m]
  __shared__ float Bs[GB_K][G_LD];  // Bs[k][n]
  const int m0 = blockIdx.y * GB_M, n0 = blockIdx.x * GB_N;
  const int kb = blockIdx.z * k_per_split;
  const int ke = min(K, kb + k_per_split);
  const int wave = threadIdx.x >> 6, l = threadIdx.x & 63, i = l & 15, q = l >> 4;
  f32x4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int k0 = kb; k0 < ke; k0 += GB_K) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int e = threadIdx.x + 256 * r;
      int am, ak, bk, bn;
      if (A_KC) { ak = e & 15; am = e >> 4; } else { am = e & 63; ak = e >> 6; }
      if (B_NC) { bn = e & 63; bk = e >> 6; } else { bk = e & 15; bn = e >> 4; }
      const int gm = m0 + am, gk = k0 + ak;
      As[ak][am] = (gm < M && gk < ke) ? A[(long long)gm * sam + (long long)gk * sak] : 0.f;
      const int gn = n0 + bn, gk2 = k0 + bk;
      Bs[bk][bn] = (gn < N && gk2 < ke) ? B[(long long)gk2 * sbk + (long long)gn * sbn] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < GB_K; kk += 4) {
      const float a = As[kk + q][16 * wave + i];
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = ORL_MFMA(a, Bs[kk + q][16 * t + i], acc[t]);
    }
    __syncthreads();
  }
  // D fragment: lane (n = i, q) reg r -> C[m0 + 16 wave + 4q + r][n0 + 16 t + i]
  float* out = partials ? partials + (size_t)blockIdx.z * (size_t)M * N : C;
  const long long ld = partials ? (long long)N : ldc;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int gm = m0 + 16 * wave + 4 * q + r, gn = n0 + 16 * t + i;
      if (gm < M && gn < N) out[(long long)gm * ld + gn] = acc[t][r];
    }
}

__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ partials, int n_split, long long MN,
                                                            int N, float* __restrict__ C, long long ldc) {
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < MN; e += (long long)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int k = 0; k < n_split; ++k) s += partials[(size_t)k * MN + e];  // fixed order: deterministic
    C[(e / N) * ldc + (e % N)] = s;
  }
}

// ---------------------------------------------------------------------------------------------------- rows
constexpr int ROW_MAX_PER_LANE = 8;  // H <= 512

// One wavefront per row; lane l owns columns l, l+64, ...
__global__ __launch_bounds__(256) void row_fwd_kernel(const float* __restrict__ z, const float* __restrict__ bias,
                                                      int act, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, int B, int H,
                                                      float* __restrict__ a_out, float* __restrict__ xhat_out,
                                                      float* __restrict__ rstd_out, float* __restrict__ y_out) {
  const int l = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6;
  for (int row = blockIdx.x * wpb + (threadIdx.x >> 6); row < B; row += gridDim.x * wpb) {
    float a[ROW_MAX_PER_LANE];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < ROW_MAX_PER_LANE; ++k) {
      const int h = l + 64 * k;
      a[k] = 0.f;
      if (h < H) {
        float v = z[(size_t)row * H + h];
        if (bias) v += bias[h];
        a[k] = act_fwd(v, act);
        if (a_out) a_out[(size_t)row * H + h] = a[k];
        s += a[k];
      }
    }
    if (gamma == nullptr) {  // no LayerNorm: y = act(z + bias)
#pragma unroll
      for (int k = 0; k < ROW_MAX_PER_LANE; ++k) {
        const int h = l + 64 * k;
        if (h < H && y_out) y_out[(size_t)row * H + h] = a[k];
      }
      continue;
    }
    // torch.nn.LayerNorm: biased variance of the centred values, eps = 1e-5
    const float mean = wave_sum(s) / (float)H;
    float v2 = 0.f;
#pragma unroll
    for (int k = 0; k < ROW_MAX_PER_LANE; ++k) {
      const int h = l + 64 * k;
      if (h < H) { a[k] -= mean; v2 += a[k] * a[k]; }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(v2) / (float)H + 1e-5f);
    if (l == 0 && rstd_out) rstd_out[row] = rstd;
#pragma unroll
    for (int k = 0; k < ROW_MAX_PER_LANE; ++k) {
      const int h = l + 64 * k;
      if (h < H) {
        const float xh = a[k] * rstd;
        if (xhat_out) xhat_out[(size_t)row * H + h] = xh;
        if (y_out) y_out[(size_t)row * H + h] = xh * gamma[h] + beta[h];
      }
    }
  }
}

// Backward of nn.Sequential(Linear, act, LayerNorm) below the Linear: dy -> dz (= gradient at the Linear's output),
// and per-workgroup partial rows [d gamma (H) | d beta (H) | d bias (H)].
__global__ __launch_bounds__(256) void row_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ gamma,
                                                      const float* __restrict__ xhat, const float* __restrict__ rstd,
                                                      const float* __restrict__ a, int act, int B, int H,
                                                      float* __restrict__ dz_out, float* __restrict__ partials) {
  extern __shared__ float sh_rb[];  // [waves][3H]
  const int l = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wpb = blockDim.x >> 6;
  float cg[ROW_MAX_PER_LANE], cb[ROW_MAX_PER_LANE], cz[ROW_MAX_PER_LANE];
#pragma unroll
  for (int k = 0; k < ROW_MAX_PER_LANE; ++k) cg[k] = cb[k] = cz[k] = 0.f;
  for (int row = blockIdx.x * wpb + wave; row < B; row += gridDim.x * wpb) {
    float d[ROW_MAX_PER_LANE], xh[ROW_MAX_PER_LANE];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < ROW_MAX_PER_LANE; ++k) {
      const int h = l + 64 * k;
      d[k] = 0.f; xh[k] = 0.f;
      if (h < H) {
        const float g = dy[(size_t)row * H + h];
        if (gamma) {
          xh[k] = xhat[(size_t)row * H + h];
          cg[k] += g * xh[k];
          cb[k] += g;
          d[k] = g * gamma[h];
          s1 += d[k];
          s2 += d[k] * xh[k];
        } else {
          d[k] = g;
        }
      }
    }
    float r = 1.f, c1 = 0.f, c2 = 0.f;
    if (gamma) {
      r = rstd[row];
      c1 = wave_sum(s1) / (float)H;
      c2 = wave_sum(s2) / (float)H;
    }
#pragma unroll
    for (int k = 0; k < ROW_MAX_PER_LANE; ++k) {
      const int h = l + 64 * k;
      if (h < H) {
        float da = gamma ? (d[k] - c1 - xh[k] * c2) * r : d[k];
        if (act != ORL_ACT_NONE) da *= act_bwd(a[(size_t)row * H + h], act);
        cz[k] += da;
        if (dz_out) dz_out[(size_t)row * H + h] = da;
      }
    }
  }
  float* mine = sh_rb + (size_t)wave * 3 * H;
#pragma unroll
  for (int k = 0; k < ROW_MAX_PER_LANE; ++k) {
    const int h = l + 64 * k;
    if (h < H) { mine[h] = cg[k]; mine[H + h] = cb[k]; mine[2 * H + h] = cz[k]; }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 3 * H; e += blockDim.x) {
    float s = 0.f;
    for (int w = 0; w < wpb; ++w) s += sh_rb[(size_t)w * 3 * H + e];
    partials[(size_t)blockIdx.x * 3 * H + e] = s;
  }
}

__global__ __launch_bounds__(256) void gather_cols_kernel(const float* __restrict__ rec, int R, int col0, int width,
                                                          const int64_t* __restrict__ idx, int mb,
                                                          float* __restrict__ out) {
  const long long n = (long long)mb * width;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) {
    const int i = (int)(e / width), c = (int)(e % width);
    const long long row = idx ? idx[i] : i;
    out[e] = rec[(size_t)row * R + col0 + c];
  }
}

// ---------------------------------------------------------------------------------------------------- losses
// denominators of the masked means (they depend on the records only): den[0] = sum(active), den[1] = rows
// Up to 256 workgroups sum slices of the rows; the workgroup that finishes last (a counter in the caller's scratch)
// adds the slice sums in index order - one launch, deterministic.  (One 256-thread workgroup walking 524 288 rows
// took 2 ms per call: 17 % of the general path's iteration at configuration 2's shape.)
constexpr int DEN_MAX_BLOCKS = 256;
__global__ __launch_bounds__(256) void denoms_kernel(const float* __restrict__ rec, int R, int col_active,
                                                     const int64_t* __restrict__ idx, int mb, float* __restrict__ den,
                                                     float* __restrict__ scratch) {
  __shared__ float sh[4];
  __shared__ int last;
  float s = 0.f;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < mb; i += gridDim.x * blockDim.x)
    s += rec[(size_t)(idx ? idx[i] : i) * R + col_active];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  unsigned* counter = (unsigned*)(scratch + DEN_MAX_BLOCKS);
  if (threadIdx.x == 0) {
    scratch[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
    __threadfence();
    last = atomicAdd(counter, 1u) == gridDim.x - 1 ? 1 : 0;
  }
  __syncthreads();
  if (last && threadIdx.x == 0) {
    __threadfence();
    float t = 0.f;
    for (unsigned b = 0; b < gridDim.x; ++b) t += __hip_atomic_load(scratch + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    den[0] = t;
    den[1] = (float)mb;
    *counter = 0u;  // ready for the next call
  }
}



__device__ inline float block_sum_256(float v, float* sh) {
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  const float t = (sh[0] + sh[1]) + (sh[2] + sh[3]);
  __syncthreads();
  return t;
}

// One thread per minibatch row.  partial row per workgroup: [ploss_sum, ent_sum, ratio_sum, dlogstd[n_out]...]
// mode 0: training (writes dlogits, stats); mode 1: evaluate_actions (writes logp_out [mb, a_w], ent_out [mb])
__global__ __launch_bounds__(256) void policy_loss_kernel(orl_head_desc hd, const float* __restrict__ logits,
                                                          const float* __restrict__ logstd, const float* __restrict__ rec,
                                                          int R, const int64_t* __restrict__ idx, int mb, GenCols c,
                                                          const float* __restrict__ den, orl_ppo_hparams hp,
                                                          float* __restrict__ dlogits, float* __restrict__ partials,
                                                          int mode, float* __restrict__ logp_out,
                                                          float* __restrict__ ent_out) {
  __shared__ float sh[4];
  const int NT = hd.n_out;  // total logits per row
  float st_loss = 0.f, st_ent = 0.f, st_ratio = 0.f;
  float dls[16];
  for (int k = 0; k < 16; ++k) dls[k] = 0.f;
  float inv_den = 1.f, inv_ent_den = 1.f;
  if (mode == 0) {
    const float d = hp.use_policy_active_masks ? den[0] : den[1];
    inv_den = 1.f / d;
    inv_ent_den = inv_den;
  }
  for (int i0 = blockIdx.x * blockDim.x; i0 < mb; i0 += gridDim.x * blockDim.x) {
    const int i = i0 + threadIdx.x;
    if (i < mb) {
      const float* r = rec + (size_t)(idx ? idx[i] : i) * R;
      const float* lgp = logits + (size_t)i * NT;
      float lg[GEN_MAX_OUT];
      for (int k = 0; k < NT; ++k) lg[k] = lgp[k];
      gen_policy_loss_row(hd, lg, logstd, r, c, inv_den, inv_ent_den, hp, mode, dlogits ? dlogits + (size_t)i * NT : nullptr,
                          logp_out ? logp_out + (size_t)i * c.a_w : nullptr, ent_out ? ent_out + i : nullptr, st_loss,
                          st_ent, st_ratio, dls);
    }
  }
  if (mode == 1) return;
  const float t0 = block_sum_256(st_loss, sh), t1 = block_sum_256(st_ent, sh), t2 = block_sum_256(st_ratio, sh);
  float* out = partials + (size_t)blockIdx.x * (4 + 16);
  if (threadIdx.x == 0) { out[0] = t0; out[1] = t1; out[2] = t2; out[3] = 0.f; }
  for (int k = 0; k < 16; ++k) {
    const float t = block_sum_256(dls[k], sh);
    if (threadIdx.x == 0) out[4 + k] = t;
  }
}

// cal_value_loss (ppo.py:178-220): one thread per row; partial row per workgroup [vloss_sum]
__global__ __launch_bounds__(256) void value_loss_kernel(const float* __restrict__ values, const float* __restrict__ rec,
                                                         int R, const int64_t* __restrict__ idx, int mb, GenCols c,
                                                         const float* __restrict__ vn_state,
                                                         const float* __restrict__ den, orl_ppo_hparams hp,
                                                         float* __restrict__ dvalues, float* __restrict__ partials) {
  __shared__ float sh[4];
  float vn_mean, vn_sd;
  gen_vn_coeffs(vn_state, hp, vn_mean, vn_sd);
  const float inv_den = 1.f / (hp.use_value_active_masks ? den[0] : den[1]);
  float st = 0.f;
  for (int i0 = blockIdx.x * blockDim.x; i0 < mb; i0 += gridDim.x * blockDim.x) {
    const int i = i0 + threadIdx.x;
    if (i < mb) {
      const float* r = rec + (size_t)(idx ? idx[i] : i) * R;
      dvalues[i] = gen_value_loss_row(values[i], r, c, vn_mean, vn_sd, inv_den, hp, st);
    }
  }
  const float t = block_sum_256(st, sh);
  if (threadIdx.x == 0) partials[blockIdx.x] = t;
}

// ---------------------------------------------------------------------------------------------------- sampling
__global__ __launch_bounds__(256) void gen_sample_kernel(orl_head_desc hd, const float* __restrict__ logits,
                                                         const float* __restrict__ logstd,
                                                         const float* __restrict__ amask, int B, int deterministic,
                                                         uint64_t seed, uint64_t row0, uint64_t rng_step,
                                                         const unsigned long long* __restrict__ rng_dev,
                                                         const float* __restrict__ forced, int a_w,
                                                         float* __restrict__ actions, float* __restrict__ logp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  const int NT = hd.n_out;
  float lg[GEN_MAX_OUT];
  for (int k = 0; k < NT; ++k) lg[k] = logits[(size_t)i * NT + k];
  gen_sample_row(hd, lg, logstd, amask ? amask + (size_t)i * NT : nullptr, deterministic, seed,
                 row0 + (uint64_t)i, rng_step + (rng_dev ? *rng_dev : 0ull), forced ? forced + (size_t)i * a_w : nullptr,
                 actions + (size_t)i * a_w, logp + (size_t)i * a_w);
}

// ---------------------------------------------------------------------------------------------------- optimiser
__global__ __launch_bounds__(256) void sqnorm_partials_kernel(const float* __restrict__ g, long long n,
                                                              float* __restrict__ partials) {
  __shared__ float sh[4];
  float s = 0.f;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x)
    s += g[e] * g[e];
  const float t = block_sum_256(s, sh);
  if (threadIdx.x == 0) partials[blockIdx.x] = t;
}

// clip_grad_norm_ (optionally twice - the shared model is clipped once as "actor" and once as "critic" parameters,
// ppo.py:127-145) + torch.optim.Adam.  info[slot_a] += norm before the first clip, info[slot_c] += norm before the second.
__global__ __launch_bounds__(256) void gen_adam_kernel(orl_adam_state ad, long long n, const float* __restrict__ partials,
                                                       int n_partials, float max_norm, int use_clip, int n_clips,
                                                       float* __restrict__ info, int slot_a, int slot_c) {
  float ss = 0.f;
  for (int k = 0; k < n_partials; ++k) ss += partials[k];  // every thread, same order: deterministic
  const float total = sqrtf(ss);
  float coef = 1.f;
  if (use_clip) coef = fminf(max_norm / (total + 1e-6f), 1.f);
  float total2 = total * coef, coef2 = 1.f;
  if (n_clips > 1 && use_clip) coef2 = fminf(max_norm / (total2 + 1e-6f), 1.f);
  if (blockIdx.x == 0 && threadIdx.x == 0 && info != nullptr) {
    if (slot_a >= 0) info[slot_a] += total;
    if (slot_c >= 0) info[slot_c] += n_clips > 1 ? total2 : total;
  }
  const double b1d = 0.9, b2d = 0.999;
  const float b2 = (float)b2d, omb1 = (float)(1.0 - b1d), omb2 = (float)(1.0 - b2d);
  const double bc1 = 1.0 - powi_d(b1d, (long long)ad.step), bc2 = 1.0 - powi_d(b2d, (long long)ad.step);
  const float step_size = (float)((double)ad.lr / bc1), bc2_sqrt = (float)sqrt(bc2);
  for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (long long)gridDim.x * blockDim.x) {
    float g = ad.grad[p] * coef * coef2;
    ad.grad[p] = g;
    float th = ad.theta[p];
    if (ad.weight_decay != 0.f) g += ad.weight_decay * th;
    float m = ad.m[p], v = ad.v[p];
    m = m + (g - m) * omb1;
    v = v * b2 + omb2 * (g * g);
    const float denom = sqrtf(v) / bc2_sqrt + ad.eps;
    th = th - step_size * (m / denom);
    ad.m[p] = m; ad.v[p] = v; ad.theta[p] = th;
  }
}

__global__ __launch_bounds__(256) void vec_add_kernel(float* __restrict__ dst, const float* __restrict__ src, long long n) {
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x)
    dst[e] += src[e];
}

// ---------------------------------------------------------------------------------------------------- GRU cell
// torch.nn.GRU (one layer) as RNNLayer runs it (networks/utils/rnn.py:28-99), with the two projections done as GEMMs:
//   gi = x W_ih^T + b_ih, gh = h_in W_hh^T + b_hh, h_in = h_prev * mask;  column blocks [r | z | n] of width H.
//   r = sigmoid(gi_r + gh_r), z = sigmoid(gi_z + gh_z), n = tanh(gi_n + r * gh_n), h = (1 - z) * n + z * h_in.
// The forward optionally stores (r, z, n, gh_n) for the backward and the NEXT step's masked input h * mask_next.
__device__ inline float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ __launch_bounds__(256) void gru_gate_fwd_kernel(const float* __restrict__ gi, const float* __restrict__ gh,
                                                           const float* __restrict__ h_in,
                                                           const float* __restrict__ mask_next, long long n_el, int H,
                                                           float* __restrict__ h_out, float* __restrict__ h_in_next,
                                                           float* __restrict__ save) {
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n_el; e += (long long)gridDim.x * blockDim.x) {
    const long long row = e / H;
    const int c = (int)(e % H);
    const float* a = gi + row * 3 * H;
    const float* b = gh + row * 3 * H;
    const float r = sigmoidf_(a[c] + b[c]);
    const float z = sigmoidf_(a[H + c] + b[H + c]);
    const float ghn = b[2 * H + c];
    const float n = tanhf(a[2 * H + c] + r * ghn);
    const float h = (1.0f - z) * n + z * h_in[e];
    h_out[e] = h;
    if (h_in_next) h_in_next[e] = h * mask_next[row];
    if (save) {
      float* sv = save + row * 4 * H;
      sv[c] = r; sv[H + c] = z; sv[2 * H + c] = n; sv[3 * H + c] = ghn;
    }
  }
}

// dh = gradient at h;  dgi / dgh = gradients at the two projections' outputs, dh_in = the direct path z * dh
__global__ __launch_bounds__(256) void gru_gate_bwd_kernel(const float* __restrict__ dh, const float* __restrict__ save,
                                                           const float* __restrict__ h_in, long long n_el, int H,
                                                           float* __restrict__ dgi, float* __restrict__ dgh,
                                                           float* __restrict__ dh_in) {
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n_el; e += (long long)gridDim.x * blockDim.x) {
    const long long row = e / H;
    const int c = (int)(e % H);
    const float* sv = save + row * 4 * H;
    const float r = sv[c], z = sv[H + c], n = sv[2 * H + c], ghn = sv[3 * H + c];
    const float g = dh[e];
    const float dn = g * (1.0f - z);
    const float dz = g * (h_in[e] - n);
    const float dpn = dn * (1.0f - n * n);
    const float dpr = dpn * ghn * r * (1.0f - r);
    const float dpz = dz * z * (1.0f - z);
    float* a = dgi + row * 3 * H;
    float* b = dgh + row * 3 * H;
    a[c] = dpr; a[H + c] = dpz; a[2 * H + c] = dpn;
    b[c] = dpr; b[H + c] = dpz; b[2 * H + c] = dpn * r;
    dh_in[e] = g * z;
  }
}

// torch.nn.LSTM cell (one layer) after its two projections gi = x W_ih^T + b_ih, gh = h_in W_hh^T + b_hh ([N, 4H], column
// blocks i | f | g | o):  i, f, o = sigmoid, g = tanh;  c = f * c_in + i * g;  h = o * tanh(c).
// save [N, 5H] = (i, f, g, o, tanh(c)); h_in_next / c_in_next = the next step's masked inputs.
__global__ __launch_bounds__(256) void lstm_gate_fwd_kernel(const float* __restrict__ gi, const float* __restrict__ gh,
                                                            const float* __restrict__ c_in,
                                                            const float* __restrict__ mask_next, long long n_el, int H,
                                                            float* __restrict__ h_out, float* __restrict__ c_out,
                                                            float* __restrict__ h_in_next, float* __restrict__ c_in_next,
                                                            float* __restrict__ save) {
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n_el; e += (long long)gridDim.x * blockDim.x) {
    const long long row = e / H;
    const int c = (int)(e % H);
    const float* a = gi + row * 4 * H;
    const float* b = gh + row * 4 * H;
    const float ig = sigmoidf_(a[c] + b[c]);
    const float fg = sigmoidf_(a[H + c] + b[H + c]);
    const float gg = tanhf(a[2 * H + c] + b[2 * H + c]);
    const float og = sigmoidf_(a[3 * H + c] + b[3 * H + c]);
    const float cn = fg * c_in[e] + ig * gg;
    const float tc = tanhf(cn);
    const float h = og * tc;
    h_out[e] = h;
    c_out[e] = cn;
    if (h_in_next) {
      const float m = mask_next[row];
      h_in_next[e] = h * m;
      c_in_next[e] = cn * m;
    }
    if (save) {
      float* sv = save + row * 5 * H;
      sv[c] = ig; sv[H + c] = fg; sv[2 * H + c] = gg; sv[3 * H + c] = og; sv[4 * H + c] = tc;
    }
  }
}

// dh, dc = gradients at h and c of this step (dc may be NULL = 0) -> dgates [N, 4H] (the gradient at BOTH projections'
// outputs) and dc_in = the gradient at c_in
__global__ __launch_bounds__(256) void lstm_gate_bwd_kernel(const float* __restrict__ dh, const float* __restrict__ dc,
                                                            const float* __restrict__ save, const float* __restrict__ c_in,
                                                            long long n_el, int H, float* __restrict__ dgates,
                                                            float* __restrict__ dc_in) {
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n_el; e += (long long)gridDim.x * blockDim.x) {
    const long long row = e / H;
    const int c = (int)(e % H);
    const float* sv = save + row * 5 * H;
    const float ig = sv[c], fg = sv[H + c], gg = sv[2 * H + c], og = sv[3 * H + c], tc = sv[4 * H + c];
    const float g = dh[e];
    const float dct = (dc ? dc[e] : 0.f) + g * og * (1.0f - tc * tc);
    float* d = dgates + row * 4 * H;
    d[c] = dct * gg * ig * (1.0f - ig);
    d[H + c] = dct * c_in[e] * fg * (1.0f - fg);
    d[2 * H + c] = dct * ig * (1.0f - gg * gg);
    d[3 * H + c] = g * tc * og * (1.0f - og);
    dc_in[e] = dct * fg;
  }
}

// out[row, :] = (a[row, :] + b[row, :]) * scale[row] + add[row, :]   (b, scale, add may be NULL): masks on hidden states
// and the carry of back-propagation through time
__global__ __launch_bounds__(256) void row_affine_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                         const float* __restrict__ scale, const float* __restrict__ add,
                                                         long long n_el, int H, float* __restrict__ out) {
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n_el; e += (long long)gridDim.x * blockDim.x) {
    float v = a[e];
    if (b) v += b[e];
    if (scale) v *= scale[e / H];
    if (add) v += add[e];
    out[e] = v;
  }
}

// train_info accumulation from the reduced loss statistics (everything stays on the device)
__global__ void gen_info_kernel(const float* __restrict__ psums, const float* __restrict__ vsums,
                                const float* __restrict__ den, orl_ppo_hparams hp, float ent_div, float ratio_div,
                                float* __restrict__ info) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const float den_p = hp.use_policy_active_masks ? den[0] : den[1];
  const float den_v = hp.use_value_active_masks ? den[0] : den[1];
  if (vsums) info[0] += vsums[0] / den_v;
  if (psums) {
    info[1] += psums[0] / den_p;
    info[2] += psums[1] / (den_p * ent_div);
    info[5] += psums[2] / (den[1] * ratio_div);
  }
}

}  // namespace orl

using namespace orl;

extern "C" {

int orl_gemm(const float* A, int64_t sam, int64_t sak, const float* B, int64_t sbk, int64_t sbn, float* C, int64_t ldc,
             int M, int N, int K, int n_split, float* partials, void* stream) {
  ORL_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0, "orl_gemm: bad arguments (M=%d N=%d K=%d)", M, N, K);
  ORL_REQUIRE(n_split >= 1 && (n_split == 1 || partials), "orl_gemm: split-K needs a partials buffer");
  int kps = (K + n_split - 1) / n_split;
  kps = (kps + GB_K - 1) / GB_K * GB_K;
  const int splits = (K + kps - 1) / kps;
  const dim3 grid((N + GB_N - 1) / GB_N, (M + GB_M - 1) / GB_M, splits);
  float* part = splits > 1 ? partials : nullptr;
  hipStream_t s = (hipStream_t)stream;
  const bool akc = sak == 1, bnc = sbn == 1;
#define ORL_GEMM_GO(AK, BN) \
  hipLaunchKernelGGL((gemm_kernel<AK, BN>), grid, dim3(256), 0, s, A, (long long)sam, (long long)sak, B, (long long)sbk, \
                     (long long)sbn, C, (long long)ldc, M, N, K, kps, part)
  if (akc && bnc) ORL_GEMM_GO(true, true);
  else if (akc) ORL_GEMM_GO(true, false);
  else if (bnc) ORL_GEMM_GO(false, true);
  else ORL_GEMM_GO(false, false);
#undef ORL_GEMM_GO
  int rc = launch_status("orl_gemm");
  if (rc || splits == 1) return rc;
  const long long MN = (long long)M * N;
  int g = (int)((MN + 255) / 256);
  if (g > 1024) g = 1024;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(g), dim3(256), 0, s, partials, splits, MN, N, C, (long long)ldc);
  return launch_status("orl_gemm(split-K reduce)");
}

int orl_row_fwd(const float* z, const float* bias, int act, const float* gamma, const float* beta, int B, int H,
                float* a_out, float* xhat_out, float* rstd_out, float* y_out, void* stream) {
  ORL_REQUIRE(z && B > 0 && H > 0 && H <= 64 * ROW_MAX_PER_LANE, "orl_row_fwd: bad arguments (B=%d, H=%d <= %d)", B, H,
              64 * ROW_MAX_PER_LANE);
  ORL_REQUIRE(act >= ORL_ACT_NONE && act <= ORL_ACT_ELU, "orl_row_fwd: activation id %d", act);
  ORL_REQUIRE((gamma == nullptr) == (beta == nullptr), "orl_row_fwd: gamma and beta come together");
  int grid = (B + 3) / 4;
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(row_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, z, bias, act, gamma, beta, B, H, a_out,
                     xhat_out, rstd_out, y_out);
  return launch_status("orl_row_fwd");
}

int orl_row_bwd(const float* dy, const float* gamma, const float* xhat, const float* rstd, const float* a, int act, int B,
                int H, float* dz_out, float* col_partials, int max_blocks, int* n_blocks_out, void* stream) {
  ORL_REQUIRE(dy && col_partials && n_blocks_out && B > 0 && H > 0 && H <= 64 * ROW_MAX_PER_LANE && max_blocks > 0,
              "orl_row_bwd: bad arguments");
  ORL_REQUIRE(!gamma || (xhat && rstd), "orl_row_bwd: LayerNorm backward needs xhat and rstd");
  ORL_REQUIRE(act == ORL_ACT_NONE || a, "orl_row_bwd: activation backward needs the activations");
  int grid = (B + 3) / 4;
  if (grid > max_blocks) grid = max_blocks;
  const size_t lds = (size_t)4 * 3 * H * sizeof(float);
  hipLaunchKernelGGL(row_bwd_kernel, dim3(grid), dim3(256), lds, (hipStream_t)stream, dy, gamma, xhat, rstd, a, act, B, H,
                     dz_out, col_partials);
  *n_blocks_out = grid;
  return launch_status("orl_row_bwd");
}

int orl_gather_cols(const float* records, int rec_width, int col0, int width, const int64_t* idx, int mb, float* out,
                    void* stream) {
  ORL_REQUIRE(records && out && mb > 0 && width > 0 && col0 >= 0 && col0 + width <= rec_width,
              "orl_gather_cols: bad arguments");
  long long n = (long long)mb * width;
  int grid = (int)((n + 255) / 256);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(gather_cols_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, records, rec_width, col0, width,
                     idx, mb, out);
  return launch_status("orl_gather_cols");
}

static int head_total(const orl_head_desc* h) {
  if (h->kind == ORL_HEAD_MIXED) return h->nvec[0] + h->nvec[1];
  if (h->kind == ORL_HEAD_MULTI_DISCRETE) {
    int t = 0;
    for (int k = 0; k < h->n_heads; ++k) t += h->nvec[k];
    return t;
  }
  return h->n_out;
}

static int check_head(const orl_head_desc* h, const char* who) {
  if (!h) return fail(ORL_E_INVALID, "%s: null head descriptor", who);
  if (h->kind < ORL_HEAD_CATEGORICAL || h->kind > ORL_HEAD_MIXED)
    return fail(ORL_E_UNSUPPORTED, "%s: head kind %d", who, h->kind);
  if (h->kind == ORL_HEAD_MIXED && (h->n_heads != 2 || h->nvec[0] < 1 || h->nvec[0] > 15 || h->nvec[1] < 1))
    return fail(ORL_E_UNSUPPORTED, "%s: mixed head needs nvec = {Box dims (1..15), Discrete classes}", who);
  if (h->kind == ORL_HEAD_MULTI_DISCRETE && (h->n_heads < 1 || h->n_heads > ORL_MAX_HEADS))
    return fail(ORL_E_UNSUPPORTED, "%s: MultiDiscrete with %d components (max %d)", who, h->n_heads, ORL_MAX_HEADS);
  if (h->n_out < 1 || h->n_out > GEN_MAX_OUT || head_total(h) != h->n_out)
    return fail(ORL_E_UNSUPPORTED, "%s: %d logits per row (max %d; must equal the sum of the components)", who, h->n_out,
                GEN_MAX_OUT);
  if (h->kind == ORL_HEAD_GAUSSIAN && h->n_out > 16) return fail(ORL_E_UNSUPPORTED, "%s: Box(%d) > 16 dims", who, h->n_out);
  return 0;
}


int orl_gen_denoms(const float* records, int rec_width, int Dp, int Dc, int a_w, const int64_t* idx, int mb, float* den,
                   float* scratch, void* stream) {
  ORL_REQUIRE(records && den && scratch && mb > 0, "orl_gen_denoms: bad arguments");
  const GenCols c = gen_cols(Dp, Dc, a_w, 0);
  int grid = (mb + 2047) / 2048;
  if (grid > DEN_MAX_BLOCKS) grid = DEN_MAX_BLOCKS;
  hipLaunchKernelGGL(denoms_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, records, rec_width, c.o_am, idx, mb,
                     den, scratch);
  return launch_status("orl_gen_denoms");
}

int orl_gen_policy_loss(const orl_head_desc* head, const float* logits, const float* logstd, const float* records,
                        int rec_width, int Dp, int Dc, int a_w, int K, const int64_t* idx, int mb, const float* den,
                        const orl_ppo_hparams* hp, float* dlogits, float* partials, int max_blocks, int* n_blocks_out,
                        float* logp_out, float* ent_out, void* stream) {
  int rc = check_head(head, "orl_gen_policy_loss");
  if (rc) return rc;
  const bool eval = dlogits == nullptr;
  ORL_REQUIRE(logits && records && hp && mb > 0, "orl_gen_policy_loss: null pointer");
  ORL_REQUIRE(eval ? (logp_out && ent_out) : (den && partials && n_blocks_out && max_blocks > 0),
              "orl_gen_policy_loss: training needs dlogits/den/partials, evaluation needs logp_out/ent_out");
  ORL_REQUIRE((head->kind != ORL_HEAD_GAUSSIAN && head->kind != ORL_HEAD_MIXED) || logstd,
              "orl_gen_policy_loss: Gaussian head without logstd");
  ORL_REQUIRE(orl_record_width(Dp, Dc, a_w, K) == rec_width, "orl_gen_policy_loss: record width %d != %d", rec_width,
              orl_record_width(Dp, Dc, a_w, K));
  int grid = (mb + 255) / 256;
  if (!eval && grid > max_blocks) grid = max_blocks;
  hipLaunchKernelGGL(policy_loss_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, *head, logits, logstd, records,
                     rec_width, idx, mb, gen_cols(Dp, Dc, a_w, K), den, *hp, dlogits, partials, eval ? 1 : 0, logp_out,
                     ent_out);
  if (n_blocks_out) *n_blocks_out = grid;
  return launch_status("orl_gen_policy_loss");
}

int orl_gen_value_loss(const float* values, const float* records, int rec_width, int Dp, int Dc, int a_w, int K,
                       const int64_t* idx, int mb, const float* vn_state, const float* den, const orl_ppo_hparams* hp,
                       float* dvalues, float* partials, int max_blocks, int* n_blocks_out, void* stream) {
  ORL_REQUIRE(values && records && den && hp && dvalues && partials && n_blocks_out && mb > 0 && max_blocks > 0,
              "orl_gen_value_loss: bad arguments");
  ORL_REQUIRE(orl_record_width(Dp, Dc, a_w, K) == rec_width, "orl_gen_value_loss: record width %d != %d", rec_width,
              orl_record_width(Dp, Dc, a_w, K));
  int grid = (mb + 255) / 256;
  if (grid > max_blocks) grid = max_blocks;
  hipLaunchKernelGGL(value_loss_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, values, records, rec_width, idx, mb,
                     gen_cols(Dp, Dc, a_w, K), vn_state, den, *hp, dvalues, partials);
  *n_blocks_out = grid;
  return launch_status("orl_gen_value_loss");
}

int orl_gen_sample(const orl_head_desc* head, const float* logits, const float* logstd, const float* action_masks, int B,
                   int deterministic, uint64_t seed, uint64_t row0, uint64_t rng_step, const uint64_t* rng_step_dev,
                   const float* forced_u, int a_w, float* actions, float* logp, void* stream) {
  int rc = check_head(head, "orl_gen_sample");
  if (rc) return rc;
  ORL_REQUIRE(logits && actions && logp && B > 0 && a_w > 0, "orl_gen_sample: bad arguments");
  ORL_REQUIRE((head->kind != ORL_HEAD_GAUSSIAN && head->kind != ORL_HEAD_MIXED) || logstd,
              "orl_gen_sample: Gaussian head without logstd");
  hipLaunchKernelGGL(gen_sample_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, *head, logits, logstd,
                     action_masks, B, deterministic, seed, row0, rng_step, (const unsigned long long*)rng_step_dev,
                     forced_u, a_w, actions, logp);
  return launch_status("orl_gen_sample");
}

int orl_gen_adam(const orl_adam_state* adam, int64_t n, float max_grad_norm, int use_max_grad_norm, int n_clips,
                 float* scratch, float* train_info_accum, int slot_first, int slot_second, void* stream) {
  ORL_REQUIRE(adam && adam->theta && adam->grad && adam->m && adam->v && scratch && n > 0 && adam->step >= 1,
              "orl_gen_adam: bad arguments");
  ORL_REQUIRE(n_clips == 1 || n_clips == 2, "orl_gen_adam: n_clips must be 1 or 2");
  int g = (int)((n + 255) / 256);
  if (g > 256) g = 256;  // scratch: 256 floats
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(sqnorm_partials_kernel, dim3(g), dim3(256), 0, s, adam->grad, (long long)n, scratch);
  int rc = launch_status("orl_gen_adam(norm)");
  if (rc) return rc;
  hipLaunchKernelGGL(gen_adam_kernel, dim3(g), dim3(256), 0, s, *adam, (long long)n, scratch, g, max_grad_norm,
                     use_max_grad_norm, n_clips, train_info_accum, slot_first, slot_second);
  return launch_status("orl_gen_adam");
}

int orl_vec_add(float* dst, const float* src, int64_t n, void* stream) {
  ORL_REQUIRE(dst && src && n > 0, "orl_vec_add: bad arguments");
  int g = (int)((n + 255) / 256);
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(vec_add_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, dst, src, (long long)n);
  return launch_status("orl_vec_add");
}

static inline int ew_grid(long long n) {
  long long g = (n + 255) / 256;
  return (int)(g > 4096 ? 4096 : g);
}

int orl_gen_gru_gate_fwd(const float* gi, const float* gh, const float* h_in, const float* mask_next, int N, int H,
                         float* h_out, float* h_in_next, float* save, void* stream) {
  ORL_REQUIRE(gi && gh && h_in && h_out && N > 0 && H > 0, "orl_gen_gru_gate_fwd: bad arguments");
  ORL_REQUIRE((h_in_next == nullptr) == (mask_next == nullptr), "orl_gen_gru_gate_fwd: h_in_next and mask_next come together");
  const long long n = (long long)N * H;
  hipLaunchKernelGGL(gru_gate_fwd_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, gi, gh, h_in, mask_next, n, H,
                     h_out, h_in_next, save);
  return launch_status("orl_gen_gru_gate_fwd");
}

int orl_gen_gru_gate_bwd(const float* dh, const float* save, const float* h_in, int N, int H, float* dgi, float* dgh,
                         float* dh_in, void* stream) {
  ORL_REQUIRE(dh && save && h_in && dgi && dgh && dh_in && N > 0 && H > 0, "orl_gen_gru_gate_bwd: bad arguments");
  const long long n = (long long)N * H;
  hipLaunchKernelGGL(gru_gate_bwd_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, dh, save, h_in, n, H, dgi, dgh,
                     dh_in);
  return launch_status("orl_gen_gru_gate_bwd");
}

int orl_gen_lstm_gate_fwd(const float* gi, const float* gh, const float* c_in, const float* mask_next, int N, int H,
                          float* h_out, float* c_out, float* h_in_next, float* c_in_next, float* save, void* stream) {
  ORL_REQUIRE(gi && gh && c_in && h_out && c_out && N > 0 && H > 0, "orl_gen_lstm_gate_fwd: bad arguments");
  ORL_REQUIRE((h_in_next == nullptr) == (mask_next == nullptr) && (h_in_next == nullptr) == (c_in_next == nullptr),
              "orl_gen_lstm_gate_fwd: h_in_next, c_in_next and mask_next come together");
  const long long n = (long long)N * H;
  hipLaunchKernelGGL(lstm_gate_fwd_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, gi, gh, c_in, mask_next, n, H,
                     h_out, c_out, h_in_next, c_in_next, save);
  return launch_status("orl_gen_lstm_gate_fwd");
}

int orl_gen_lstm_gate_bwd(const float* dh, const float* dc, const float* save, const float* c_in, int N, int H,
                          float* dgates, float* dc_in, void* stream) {
  ORL_REQUIRE(dh && save && c_in && dgates && dc_in && N > 0 && H > 0, "orl_gen_lstm_gate_bwd: bad arguments");
  const long long n = (long long)N * H;
  hipLaunchKernelGGL(lstm_gate_bwd_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, dh, dc, save, c_in, n, H,
                     dgates, dc_in);
  return launch_status("orl_gen_lstm_gate_bwd");
}

int orl_gen_row_affine(const float* a, const float* b, const float* row_scale, const float* add, int N, int H, float* out,
                       void* stream) {
  ORL_REQUIRE(a && out && N > 0 && H > 0, "orl_gen_row_affine: bad arguments");
  const long long n = (long long)N * H;
  hipLaunchKernelGGL(row_affine_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, a, b, row_scale, add, n, H, out);
  return launch_status("orl_gen_row_affine");
}

int orl_gen_info(const float* policy_sums, const float* value_sums, const float* den, const orl_ppo_hparams* hp,
                 float entropy_div, float ratio_div, float* train_info_accum, void* stream) {
  ORL_REQUIRE(den && hp && train_info_accum, "orl_gen_info: null pointer");
  hipLaunchKernelGGL(gen_info_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, policy_sums, value_sums, den, *hp,
                     entropy_div, ratio_div, train_info_accum);
  return launch_status("orl_gen_info");
}

}  // extern "C"
