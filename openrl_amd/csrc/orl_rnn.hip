// orl_rnn.hip - recurrent (GRU) branch of the hot path for gfx950 (use_recurrent_policy):
//   orl_rnn_act_step    : policy + value forward with hidden states and action sampling (rollout)
//   orl_rnn_chunk_rows  : recurrent_generator index arithmetic (chunks of data_chunk_length rows)
//   orl_rnn_ppo_fwd_bwd : per tower  (1) ROW kernel - forward over the L steps of 16 chunks per wavefront, PPO /
//                         value loss, back-propagation through time with the forward recomputed per step; every
//                         pre-activation gradient and layer input goes to a TAPE in HBM;  (2) WGRAD kernel - all
//                         weight gradients as one K = rows GEMM sweep over the tape (fp32 MFMA, accumulators in
//                         registers, 8 waves own disjoint output tiles);  then deterministic reductions
//   orl_rnn_ppo_apply   : raw sums -> gradients (LayerNorm-affine grads as dot products), clip, Adam
//
// Why a tape instead of the register-resident wgrad of orl_ppo_tower.h: the GRU adds 2 x 192 x 64 weights, i.e.
// 384 accumulator registers per lane on top of the 64 of W2 - they do not fit a wavefront.  Spread over the 8
// waves of a workgroup they do (<= 80 each), but then every wave needs every row's deltas: the tape is that
// exchange, sized for HBM3E (2.9 KB per row-step, written and read once per epoch).
#include <stdlib.h>
#include "orl_common.h"
#include "orl_mlp.h"
#include "orl_heads.h"
#include "orl_loss.h"
#include "orl_rnn.h"

namespace orl {

// =====================================================================================================
// rollout step
// =====================================================================================================
struct RnnActArgs {
  orl_net_desc pnet, cnet;
  const float *ptheta, *ctheta, *pobs, *cobs, *hp_in, *hc_in, *masks, *amask, *forced;
  const float* given;  // orl_rnn_eval_step: evaluate these actions [B, a] instead of sampling (NULL otherwise)
  float* ent_out;      //                    per-row entropy [B]
  float *values, *actions, *logp, *hp_out, *hc_out;
  int B, deterministic;
  uint64_t seed, row0, rng_step;
  const unsigned long long* rng_dev;  // optional device-side addend of rng_step (rng_step_dev argument)
};

// base -> GRU -> LayerNorm with every weight read straight from global memory (L2 resident; a rollout step
// touches each weight once per 16-row tile, staging 140 KB into LDS first would cost the same traffic)
template <class XB>
__device__ inline void rnn_tower_fwd_g(const float* __restrict__ th, const RnnLayout& tl, XB xb, const f32x4 (&hin)[4],
                                       f32x4 (&hnew)[4], f32x4 (&n3)[4], int j, int q) {
  f32x4 z[4], n1[4], n2[4];
  float rstd;
  load_vec_T(th + tl.ob1, q, z);
  fc1_g(th + tl.oW1, tl.D, xb, z, j, q);
  relu_T(z);
  ln_normalize_T(z, rstd);
  ln_affine_T(z, th + tl.og1, th + tl.obe1, q, n1);
  load_vec_T(th + tl.ob2, q, z);
  mm64_S<64>(th + tl.oW2, n1, z, j, q);
  ln_normalize_T(z, rstd);
  ln_affine_T(z, th + tl.og2, th + tl.obe2, q, n2);
  f32x4 r[4], zz[4], n[4], g[4];
  gru_fwd_T<64>(th + tl.oWih, th + tl.oWhh, th + tl.obih, th + tl.obhh, n2, hin, r, zz, n, g, hnew, j, q);
#pragma unroll
  for (int m = 0; m < 4; ++m) z[m] = hnew[m];
  ln_normalize_T(z, rstd);
  ln_affine_T(z, th + tl.og3, th + tl.obe3, q, n3);
}

template <int NO, int HEAD>
__global__ __launch_bounds__(128) void rnn_act_kernel(RnnActArgs A) {
  const int wave = threadIdx.x >> 6, l = threadIdx.x & 63, j = l & 15, q = l >> 4;
  const int row = blockIdx.x * TILE_B + j;
  const bool row_ok = row < A.B;
  const int rr = row_ok ? row : 0;
  const float mk = A.masks[rr];
  const bool pol = wave == 0;
  if (pol ? (A.ptheta == nullptr) : (A.ctheta == nullptr)) return;
  const RnnLayout tl(pol ? A.pnet : A.cnet);
  const float* th = pol ? A.ptheta : A.ctheta;
  const float* xrow = (pol ? A.pobs : A.cobs) + (size_t)rr * tl.D;
  const float* hrow = (pol ? A.hp_in : A.hc_in) + (size_t)rr * HID;
  float* hout = (pol ? A.hp_out : A.hc_out) + (size_t)rr * HID;
  const int D = tl.D;
  auto xb = [&](int s) -> float {
    const int k = 4 * s + q;
    return (k < D) ? xrow[k] : 0.f;
  };
  f32x4 hin[4], hnew[4], n3[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) hin[m] = *(const f32x4*)(hrow + 16 * m + 4 * q) * mk;
  rnn_tower_fwd_g(th, tl, xb, hin, hnew, n3, j, q);
  if (row_ok) {
#pragma unroll
    for (int m = 0; m < 4; ++m) *(f32x4*)(hout + 16 * m + 4 * q) = hnew[m];
  }
  if (pol) {
    float hd[NO];
    head_T<NO>(th + tl.oW3, th + tl.ob3, tl.n_out, n3, q, hd);
    const int n_out = tl.n_out;
    const int a_w = (HEAD == ORL_HEAD_CATEGORICAL) ? 1 : n_out;
    float act_o[NO], lp_o[NO];
    const float* am = (A.amask != nullptr && row_ok) ? A.amask + (size_t)row * n_out : nullptr;
    const float* fr = (A.forced != nullptr && row_ok) ? A.forced + (size_t)row * a_w : nullptr;
    if (A.given != nullptr) {  // evaluate mode: log-probs + entropy of the given actions
      float ent = 0.f;
      eval_head<NO, HEAD>(hd, n_out, th + tl.ologstd, am, A.given + (size_t)rr * a_w, lp_o, ent);
      if (row_ok && q == 0) {
#pragma unroll
        for (int c = 0; c < NO; ++c)
          if (c < a_w) A.logp[(size_t)row * a_w + c] = lp_o[c];
        if (A.ent_out) A.ent_out[row] = ent;
      }
    } else {
    sample_head<NO, HEAD>(hd, n_out, th + tl.ologstd, am, fr, A.deterministic, A.seed, A.row0 + (uint64_t)row,
                          A.rng_step + (A.rng_dev ? *A.rng_dev : 0ull), act_o, lp_o);
    if (row_ok && q == 0) {
#pragma unroll
      for (int c = 0; c < NO; ++c) {
        if (c < a_w) {
          A.actions[(size_t)row * a_w + c] = act_o[c];
          A.logp[(size_t)row * a_w + c] = lp_o[c];
        }
      }
    }
    }
  } else {
    float v[1];
    head_T<1>(th + tl.oW3, th + tl.ob3, 1, n3, q, v);
    if (row_ok && q == 0) A.values[row] = v[0];
  }
}

// Large batches: one workgroup = one tower (blocks [0, split) policy, the rest critic), its 142 KB of weights staged
// into LDS once, 4 waves x 16-row tiles.  At the cfg4 rollout batch (6144 rows) this replaces 768 waves that each pull
// every weight through L2 with global loads in front of the MFMAs (37 us) by 192 workgroups of LDS-fed GEMMs.
template <int NO, int HEAD>
__global__ __launch_bounds__(256, 1) void rnn_act_lds_kernel(RnnActArgs A, int split, int tiles_per_wg) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const bool pol = (int)blockIdx.x < split;
  const int bid = pol ? blockIdx.x : blockIdx.x - split;
  const orl_net_desc net = pol ? A.pnet : A.cnet;
  const float* th = pol ? A.ptheta : A.ctheta;
  const RnnLayout tl(net);
  const RnnLds tw(net.obs_dim, net.n_out, pol && HEAD == ORL_HEAD_GAUSSIAN);
  stage_rnn_tower(smem, th, tl, tw, threadIdx.x, blockDim.x);
  __syncthreads();
  const float* lw = smem;
  const int wave = threadIdx.x >> 6, l = threadIdx.x & 63, j = l & 15, q = l >> 4;
  const int D = tl.D;
  const int n_tiles = (A.B + TILE_B - 1) / TILE_B;
  for (int k = wave; k < tiles_per_wg; k += 4) {
    const int tile = bid * tiles_per_wg + k;
    if (tile >= n_tiles) break;
    const int row = tile * TILE_B + j;
    const bool row_ok = row < A.B;
    const int rr = row_ok ? row : 0;
    const float mk = A.masks[rr];
    const float* xrow = (pol ? A.pobs : A.cobs) + (size_t)rr * D;
    const float* hrow = (pol ? A.hp_in : A.hc_in) + (size_t)rr * HID;
    float* hout = (pol ? A.hp_out : A.hc_out) + (size_t)rr * HID;
    f32x4 hin[4], hnew[4], n3[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) hin[m] = *(const f32x4*)(hrow + 16 * m + 4 * q) * mk;
    rnn_tower_fwd_lds(lw, tw, [&](int s) -> float {
      const int c = 4 * s + q;
      return c < D ? xrow[c] : 0.f;
    }, hin, hnew, n3, j, q);
    if (row_ok) {
#pragma unroll
      for (int m = 0; m < 4; ++m) *(f32x4*)(hout + 16 * m + 4 * q) = hnew[m];
    }
    if (pol) {
      float hd[NO];
      head_T<NO>(lw + tw.W3, lw + tw.b3, tl.n_out, n3, q, hd);
      const int n_out = tl.n_out;
      const int a_w = (HEAD == ORL_HEAD_CATEGORICAL) ? 1 : n_out;
      float act_o[NO], lp_o[NO];
      const float* am = (A.amask != nullptr && row_ok) ? A.amask + (size_t)row * n_out : nullptr;
      const float* fr = (A.forced != nullptr && row_ok) ? A.forced + (size_t)row * a_w : nullptr;
      if (A.given != nullptr) {  // evaluate mode
        float ent = 0.f;
        eval_head<NO, HEAD>(hd, n_out, lw + tw.logstd, am, A.given + (size_t)rr * a_w, lp_o, ent);
        if (row_ok && q == 0) {
#pragma unroll
          for (int c = 0; c < NO; ++c)
            if (c < a_w) A.logp[(size_t)row * a_w + c] = lp_o[c];
          if (A.ent_out) A.ent_out[row] = ent;
        }
      } else {
      sample_head<NO, HEAD>(hd, n_out, lw + tw.logstd, am, fr, A.deterministic, A.seed, A.row0 + (uint64_t)row,
                            A.rng_step + (A.rng_dev ? *A.rng_dev : 0ull), act_o, lp_o);
      if (row_ok && q == 0) {
#pragma unroll
        for (int c = 0; c < NO; ++c) {
          if (c < a_w) {
            A.actions[(size_t)row * a_w + c] = act_o[c];
            A.logp[(size_t)row * a_w + c] = lp_o[c];
          }
        }
      }
      }
    } else {
      float v[1];
      head_T<1>(lw + tw.W3, lw + tw.b3, 1, n3, q, v);
      if (row_ok && q == 0) A.values[row] = v[0];
    }
  }
}

// =====================================================================================================
// recurrent_generator rows
// =====================================================================================================
__global__ void rnn_chunk_rows_kernel(const int64_t* __restrict__ chunk_idx, int n_chunks, int L, int T, int lanes,
                                      int64_t* __restrict__ rows) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_chunks * L) return;
  const int l = e / n_chunks, i = e - l * n_chunks;
  const int64_t c = chunk_idx ? chunk_idx[i] : (int64_t)i;
  const int64_t r = c * L + l;  // row of the [lane][t]-ordered flat batch (_cast, buffers/utils/util.py:96-97)
  const int64_t lane = r / T, t = r - lane * T;
  rows[e] = t * lanes + lane;
}

// recurrent_generator_v3 (buffers/replay_data.py:425-551, the joint-action loss): chunk c covers positions
// c*L .. c*L+L-1 of the (n*T + t)-ordered batch with the AGENT AXIS KEPT (_cast_v3, buffers/utils/util.py:100-101);
// a minibatch flattens [L, chunks, A], i.e. sequence (i, a) -> i*A + a.  agent0_only: one sequence per chunk (a = 0),
// what the critic of the joint-action loss evaluates (ppo.py:254-262, to_single_np).
__global__ void rnn_chunk_rows_v3_kernel(const int64_t* __restrict__ chunk_idx, int n_chunks, int L, int T, int A,
                                         int lanes, int agent0_only, int64_t* __restrict__ rows) {
  const int AA = agent0_only ? 1 : A;
  const int ns = n_chunks * AA;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= ns * L) return;
  const int l = e / ns, sq = e - l * ns;
  const int i = sq / AA, a = sq - i * AA;
  const int64_t c = chunk_idx ? chunk_idx[i] : (int64_t)i;
  const int64_t p = c * L + l;
  const int64_t n = p / T, t = p - n * T;
  rows[e] = t * lanes + n * A + a;
}

// Joint-action (JRPO) records: for every (step l, chunk i) of the minibatch the joint log-ratio
//   S = sum over agents a and action dims d of (logp_new[l, i*A+a, d] - old_logp[row(l,i,a), d])
// (ppo.py:281-300) is folded into the per-row fields the row kernel already reads, in a COPY of the records:
//   old_logp'[a, d] = old_logp[a, d] - (S - (logp_new[a, d] - old_logp[a, d]))   => exp(lp - old') = exp(S), d/dlp = ratio
//   adv'  = A * adv of agent 0,  active' = active of agent 0   (adv_targ[:, 0], active_masks_batch[:, 0]; the factor
//   A cancels the A-fold denominator the row kernel accumulates over the agent rows).
__global__ void rnn_jrpo_records_kernel(const float* __restrict__ rec, float* __restrict__ out, int R,
                                        const int64_t* __restrict__ rows /*[L][n_chunks*A]*/, int n_chunks, int L, int A,
                                        const float* __restrict__ logp_new /*[L][n_chunks*A][a_w]*/, int a_w, int o_lp,
                                        int o_adv, int o_am) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_chunks * L) return;
  const int l = e / n_chunks, i = e - l * n_chunks;
  const int ns = n_chunks * A;
  float S = 0.f;
  for (int a = 0; a < A; ++a) {
    const int64_t r = rows[(size_t)l * ns + i * A + a];
    for (int d = 0; d < a_w; ++d) S += logp_new[((size_t)l * ns + i * A + a) * a_w + d] - rec[(size_t)r * R + o_lp + d];
  }
  const int64_t r0 = rows[(size_t)l * ns + i * A];
  const float adv0 = rec[(size_t)r0 * R + o_adv], act0 = rec[(size_t)r0 * R + o_am];
  for (int a = 0; a < A; ++a) {
    const int64_t r = rows[(size_t)l * ns + i * A + a];
    for (int d = 0; d < a_w; ++d) {
      const float old = rec[(size_t)r * R + o_lp + d];
      const float own = logp_new[((size_t)l * ns + i * A + a) * a_w + d] - old;
      out[(size_t)r * R + o_lp + d] = old - (S - own);
    }
    out[(size_t)r * R + o_adv] = (float)A * adv0;
    out[(size_t)r * R + o_am] = act0;
  }
}

// =====================================================================================================
// update: row kernel
// =====================================================================================================
#ifdef ORL_PROF
// phase timing build (see orl_ppo_tower.h): wave 0 of workgroup 0 (policy tower), LDS-atomic accumulation;
// read with orl_debug_rnn_prof (tools/rnn_phase_prof.py)
__device__ unsigned long long g_rnn_prof[16];
#define RNN_T(k)                                                      \
  do {                                                                \
    if (prof_on) {                                                    \
      const unsigned long long t_now = __builtin_readcyclecounter();  \
      if (l == 0) atomicAdd(&rprof_lds[k], t_now - t_last);           \
      t_last = t_now;                                                 \
    }                                                                 \
  } while (0)
#else
#define RNN_T(k) ((void)0)
#endif
constexpr int RNN_ROW_BLOCKS = 256;
constexpr int RNN_WG_BLOCKS = 256;  // both towers together: 1 workgroup per CU (3-slot DMA ring in LDS)

struct RnnRowArgs {
  orl_net_desc net;
  const float* theta;
  const float* records;
  const int64_t* rows;  // [L][Nc]
  const float* masks;
  const float* hbuf;    // [T+1, lanes, H] stored states of this tower
  const float* vn_state;
  float* htape;         // [n_tiles*L][1024]   pre-mask input state of every step
  float* tape;          // [n_tiles*L][tape_block_floats]
  float* partials;      // [gridDim][n_logstd + ORL_N_STATS]
  orl_ppo_hparams hp;
  LossCols cols;
  int R, o_x, Nc, L;
};

// Body of the row kernel for one tower; `bid` / `nblk` = this workgroup's index / count within its tower's share
// of the launch (both towers run in ONE launch so that 2 x n_tiles work units spread over all CUs - with cfg4's
// 4800 tiles per tower a per-tower launch leaves 22 % of the wave slots idle in the last round).
template <int HEAD, int NO>
__device__ __forceinline__ void rnn_row_body(const RnnRowArgs& A, const int bid, const int nblk) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const RnnLayout tl(A.net);
  // wide categorical heads: logits and W3^T dhead as 16-MFMA GEMMs, the loss on the 4 logits the MFMA leaves per lane
  constexpr bool HMM = HEAD == ORL_HEAD_CATEGORICAL && NO > 4;
  const RnnLds tw(A.net.obs_dim, A.net.n_out, HEAD == ORL_HEAD_GAUSSIAN, HMM);
  stage_rnn_tower(smem, A.theta, tl, tw, threadIdx.x, blockDim.x, HMM);
  __syncthreads();
  const float* lw = smem;
  const int wave = threadIdx.x >> 6, l = threadIdx.x & 63, j = l & 15, q = l >> 4;
  const int D = A.net.obs_dim, n_out = A.net.n_out, DP = tw.DP;
  const int Nc = A.Nc, L = A.L;
  const orl_ppo_hparams hp = A.hp;
  const int BLK = tape_block_floats(D);
  const int ND = (D + 15) >> 4;

  float vn_mean = 0.f, vn_sd = 1.f;
  if (HEAD == ORL_HEAD_VALUE && hp.use_valuenorm && A.vn_state != nullptr) {
    const float deb = fmaxf(A.vn_state[2], 1e-5f);
    vn_mean = A.vn_state[0] / deb;
    const float msq = A.vn_state[1] / deb;
    vn_sd = sqrtf(fmaxf(msq - vn_mean * vn_mean, 1e-2f));
  }
  LossStats st = {0.f, 0.f, 0.f, 0.f, 0.f};
  float a_dls[NO];
#pragma unroll
  for (int c = 0; c < NO; ++c) a_dls[c] = 0.f;

  const int n_tiles = (Nc + TILE_B - 1) / TILE_B;
  const int nwv = blockDim.x >> 6;
#ifdef ORL_PROF
  __shared__ unsigned long long rprof_lds[16];
  const bool prof_on = blockIdx.x == 0 && wave == 0;
  if (prof_on && l < 16) rprof_lds[l] = 0ull;
  unsigned long long t_last = __builtin_readcyclecounter();
#endif
  for (int tile = bid * nwv + wave; tile < n_tiles; tile += nblk * nwv) {
    const int ci = tile * TILE_B + j;
    const bool valid = ci < Nc;
    const int cis = valid ? ci : 0;  // padding lanes shadow chunk 0: finite data, zero loss weight
    // observation B operands of one record row: xv[k] = x[4k + q] (zero beyond D), k < DP/4 <= 16
    auto load_x = [&](const float* rec, float (&xv)[16]) {
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const int c = 4 * k + q;
        xv[k] = (4 * k < DP && c < D) ? rec[A.o_x + c] : 0.f;
      }
    };
    // trunk forward of one row; returns what the backward pass needs
    auto trunk = [&](const float (&xv)[16], f32x4 (&xh1)[4], float& rstd1, unsigned& relu_bits, f32x4 (&xh2)[4],
                     float& rstd2, f32x4 (&n2)[4]) {
      f32x4 n1[4];
      load_vec_T(lw + tw.b1, q, xh1);
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        if (4 * k < DP) {
#pragma unroll
          for (int m = 0; m < 4; ++m) xh1[m] = ORL_MFMA(lw[tw.W1 + (16 * m + j) * DP + 4 * k + q], xv[k], xh1[m]);
        }
      }
      relu_bits = 0u;
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (xh1[m][r] > 0.f) relu_bits |= 1u << (4 * m + r);
          xh1[m][r] = fmaxf(xh1[m][r], 0.f);
        }
      ln_normalize_T(xh1, rstd1);
      ln_affine_T(xh1, lw + tw.g1, lw + tw.be1, q, n1);
      load_vec_T(lw + tw.b2, q, xh2);
      mm64_T(lw + tw.W2, n1, xh2, j, q);
      ln_normalize_T(xh2, rstd2);
      ln_affine_T(xh2, lw + tw.g2, lw + tw.be2, q, n2);
    };

    // The row index and mask of a step are fetched ONE STEP AHEAD (row -> record -> obs is a chain of three dependent
    // global loads when started inside the step; this way the observation loads can start at once).
    long long row_c = A.rows[cis];
    float mk_c = A.masks[row_c];

    // ---------------- forward sweep: hidden state entering every step -> htape ----------------
    {
      f32x4 h[4];
      const float* h0 = A.hbuf + (size_t)row_c * HID;
#pragma unroll
      for (int m = 0; m < 4; ++m) h[m] = *(const f32x4*)(h0 + 16 * m + 4 * q);
      for (int s = 0; s < L; ++s) {
        float* ht = A.htape + ((size_t)tile * L + s) * TV;
#pragma unroll
        for (int m = 0; m < 4; ++m) *(f32x4*)(ht + (m * 64 + l) * 4) = h[m];
        if (s == L - 1) break;  // row_c / mk_c / xv_c now describe step L-1: the first step of the backward sweep
        float xv_c[16];
        load_x(A.records + (size_t)row_c * A.R, xv_c);
        const long long row_n = A.rows[(size_t)(s + 1) * Nc + cis];
        const float mk_n = A.masks[row_n];
        f32x4 hin[4], xh1[4], xh2[4], n2[4], r[4], z[4], n[4], g[4];
        float r1, r2;
        unsigned rb;
#pragma unroll
        for (int m = 0; m < 4; ++m) hin[m] = h[m] * mk_c;
        trunk(xv_c, xh1, r1, rb, xh2, r2, n2);
        gru_fwd_T<W2S>(lw + tw.Wih, lw + tw.Whh, lw + tw.bih, lw + tw.bhh, n2, hin, r, z, n, g, h, j, q);
        row_c = row_n;
        mk_c = mk_n;
      }
    }

    RNN_T(0);  // forward sweep (L-1 forward-only steps + state tape)
    // ---------------- backward sweep (BPTT), forward recomputed per step ----------------
    f32x4 carry[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) carry[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int s = L - 1; s >= 0; --s) {
      const long long row = row_c;
      const float* rec = A.records + (size_t)row * A.R;
      const float mk = mk_c;
      float xv[16];
      load_x(rec, xv);
      // the scalar record fields the loss needs, requested NOW with the observation: read lazily inside the loss they
      // were a chain of exposed L2 round trips in the middle of the step
      const float rf_am = rec[A.cols.o_am], rf_adv = rec[A.cols.o_adv], rf_act = rec[A.cols.o_act];
      const float rf_lp = rec[A.cols.o_lp], rf_vp = rec[A.cols.o_vp], rf_rt = rec[A.cols.o_rt];
      float rf_mk[4] = {1.f, 1.f, 1.f, 1.f};  // wide categorical head: the masks of this lane's 4 classes
      if (HEAD == ORL_HEAD_CATEGORICAL && NO > 4 && A.cols.K > 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (4 * q + r < n_out) rf_mk[r] = rec[A.cols.o_mk + 4 * q + r];
      }
      auto recf = [&](int col) -> float {
        if (col == A.cols.o_am) return rf_am;
        if (col == A.cols.o_adv) return rf_adv;
        if (col == A.cols.o_act) return rf_act;
        if (col == A.cols.o_lp) return rf_lp;
        if (col == A.cols.o_vp) return rf_vp;
        if (col == A.cols.o_rt) return rf_rt;
        if (HEAD == ORL_HEAD_CATEGORICAL && NO > 4) {
          const int d = col - (A.cols.o_mk + 4 * q);
          if (d >= 0 && d < 4) return d == 0 ? rf_mk[0] : d == 1 ? rf_mk[1] : d == 2 ? rf_mk[2] : rf_mk[3];
        }
        return rec[col];
      };
      float* tb = A.tape + ((size_t)tile * L + s) * BLK;
      f32x4 hin[4];
      {
        const float* ht = A.htape + ((size_t)tile * L + s) * TV;
#pragma unroll
        for (int m = 0; m < 4; ++m) hin[m] = *(const f32x4*)(ht + (m * 64 + l) * 4) * mk;
      }
      if (s > 0) {  // next (earlier) step's inputs
        row_c = A.rows[(size_t)(s - 1) * Nc + cis];
        mk_c = A.masks[row_c];
      }
      RNN_T(1);  // step inputs: record / state-tape loads
      tape_store(tb + TV_HIN * TV, hin, j, q);
      float rstd1, rstd2, rstd3;
      unsigned relu_bits;
      f32x4 gr[4], gz[4], gn[4], ghn[4];
      float dh[NO], dls[NO];
      f32x4 dhv = f32x4{0.f, 0.f, 0.f, 0.f};  // wide categorical head: this lane's 4 head deltas
      {
        f32x4 xh1[4], xh2[4], n2[4], hnew[4];
        trunk(xv, xh1, rstd1, relu_bits, xh2, rstd2, n2);
        tape_store(tb + TV_XH1 * TV, xh1, j, q);
        tape_store(tb + TV_XH2 * TV, xh2, j, q);
        RNN_T(2);  // trunk recompute (fc1 + 64 MFMA) + tape stores
        gru_fwd_T<W2S>(lw + tw.Wih, lw + tw.Whh, lw + tw.bih, lw + tw.bhh, n2, hin, gr, gz, gn, ghn, hnew, j, q);
        RNN_T(3);  // GRU forward: 384 MFMA + gates
        ln_normalize_T(hnew, rstd3);  // hnew = xhat3
        tape_store(tb + TV_XH3 * TV, hnew, j, q);
        f32x4 n3[4];
        ln_affine_T(hnew, lw + tw.g3, lw + tw.be3, q, n3);
        if constexpr (HMM) {
          const int no4 = (n_out + 3) & ~3;
          f32x4 hv = f32x4{0.f, 0.f, 0.f, 0.f};
          if (4 * q < no4) hv = *(const f32x4*)(lw + tw.b3 + 4 * q);
#pragma unroll
          for (int mi = 0; mi < 4; ++mi) {
            const f32x4 a4 = *(const f32x4*)(lw + tw.W3P + j * W2S + 16 * mi + 4 * q);
#pragma unroll
            for (int r = 0; r < 4; ++r) hv = ORL_MFMA(a4[r], n3[mi][r], hv);
          }
          ppo_cat_loss_dist(hv, n_out, q, valid, recf, A.cols, hp, dhv, st, q == 0);
        } else {
          float hd[NO];
          head_T<NO>(lw + tw.W3, lw + tw.b3, n_out, n3, q, hd);
          ppo_row_loss<HEAD, NO>(hd, n_out, valid, recf, A.cols, hp, vn_mean, vn_sd, lw + tw.logstd, dh, dls, st, q == 0);
          if (q == 0) {
#pragma unroll
            for (int c = 0; c < NO; ++c) a_dls[c] += dls[c];
          }
        }
      }
      RNN_T(4);  // LN3, head, loss
      // head deltas -> tape (16-wide vector: lane (j,q) owns columns 4q..4q+3)
      {
        f32x4 dv = {0.f, 0.f, 0.f, 0.f};
        if constexpr (HMM) dv = dhv;
        else {
#pragma unroll
          for (int c = 0; c < NO; ++c)
            if ((c >> 2) == q) dv[c & 3] = dh[c];
        }
        *(f32x4*)(tb + TAPE_HEAD + tape_off(q, j)) = dv;
      }
      // observation tile -> tape
      for (int m = 0; m < ND; ++m) {
        f32x4 xv;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int f = 16 * m + 4 * q + r;
          xv[r] = f < D ? rec[A.o_x + f] : 0.f;
        }
        *(f32x4*)(tb + TAPE_X + m * 256 + tape_off(q, j)) = xv;
      }
      // d(features) = W3^T dhead, LN3 backward, + gradient carried from step s+1
      f32x4 dt[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) dt[m] = f32x4{0.f, 0.f, 0.f, 0.f};
      if constexpr (HMM) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int mo = 0; mo < 4; ++mo)
            dt[mo] = ORL_MFMA(lw[tw.W3P + (4 * q + r) * W2S + 16 * mo + j], dhv[r], dt[mo]);
      } else {
#pragma unroll
        for (int c = 0; c < NO; ++c) {
          if (c < n_out) {
#pragma unroll
            for (int m = 0; m < 4; ++m) dt[m] += *(const f32x4*)(lw + tw.W3 + c * HID + 16 * m + 4 * q) * dh[c];
          }
        }
      }
      {
        f32x4 xh3[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) xh3[m] = *(const f32x4*)(tb + TV_XH3 * TV + tape_off(m * 4 + q, j));
        ln_bwd_rnn(dt, xh3, lw + tw.g3, rstd3, q);
      }
#pragma unroll
      for (int m = 0; m < 4; ++m) dt[m] += carry[m];
      RNN_T(5);  // dhead / obs tape, W3^T dhead, LN3 backward
      // GRU cell backward (elementwise part); gr/gz/gn/ghn become dr/dz/dn/dghn, carry collects dt*z
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float rr = gr[m][k], zz = gz[m][k], nn = gn[m][k], gg = ghn[m][k], d = dt[m][k], hh = hin[m][k];
          const float dn_pre = d * (1.0f - zz) * (1.0f - nn * nn);
          const float dz_pre = d * (hh - nn) * zz * (1.0f - zz);
          const float dr_pre = dn_pre * gg * rr * (1.0f - rr);
          gr[m][k] = dr_pre;
          gz[m][k] = dz_pre;
          gn[m][k] = dn_pre;
          ghn[m][k] = dn_pre * rr;
          carry[m][k] = d * zz;
        }
      tape_store(tb + TV_DR * TV, gr, j, q);
      tape_store(tb + TV_DZ * TV, gz, j, q);
      tape_store(tb + TV_DN * TV, gn, j, q);
      tape_store(tb + TV_DGHN * TV, ghn, j, q);
      RNN_T(6);  // GRU elementwise backward + 4 tape vectors
      // dgrad through the hidden-to-hidden weights -> carry ; through the input weights -> dn2
      mm64_S_wt<W2S>(lw + tw.Whh, gr, carry, j, q);
      mm64_S_wt<W2S>(lw + tw.Whh + HID * W2S, gz, carry, j, q);
      mm64_S_wt<W2S>(lw + tw.Whh + 2 * HID * W2S, ghn, carry, j, q);
      f32x4 d2[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) d2[m] = f32x4{0.f, 0.f, 0.f, 0.f};
      mm64_S_wt<W2S>(lw + tw.Wih, gr, d2, j, q);
      mm64_S_wt<W2S>(lw + tw.Wih + HID * W2S, gz, d2, j, q);
      mm64_S_wt<W2S>(lw + tw.Wih + 2 * HID * W2S, gn, d2, j, q);
#pragma unroll
      for (int m = 0; m < 4; ++m) carry[m] = carry[m] * mk;  // h_in = h * mask
      RNN_T(7);  // GRU dgrad: 384 MFMA (column reads)
      // LN2 backward -> dz2
      {
        // xhat2 / xhat1 come back from the tape through a laundered pointer: otherwise the compiler forwards the
        // stored registers to these loads and keeps 32 VGPRs live across the GRU (-> scratch spills)
        const float* tbr = tb;
        asm volatile("" : "+v"(tbr));
        f32x4 xh2[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) xh2[m] = *(const f32x4*)(tbr + TV_XH2 * TV + tape_off(m * 4 + q, j));
        ln_bwd_rnn(d2, xh2, lw + tw.g2, rstd2, q);
      }
      tape_store(tb + TV_DZ2 * TV, d2, j, q);
      // dn1 = W2^T dz2, LN1 backward, relu backward -> dz1
      f32x4 d1[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) d1[m] = f32x4{0.f, 0.f, 0.f, 0.f};
      mm64_S_wt<W2S>(lw + tw.W2, d2, d1, j, q);
      {
        const float* tbr = tb;
        asm volatile("" : "+v"(tbr));
        f32x4 xh1[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) xh1[m] = *(const f32x4*)(tbr + TV_XH1 * TV + tape_off(m * 4 + q, j));
        ln_bwd_rnn(d1, xh1, lw + tw.g1, rstd1, q);
      }
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (!((relu_bits >> (4 * m + r)) & 1u)) d1[m][r] = 0.f;
      tape_store(tb + TV_DZ1 * TV, d1, j, q);
      RNN_T(8);  // LN2 backward, W2 dgrad (64 MFMA), LN1 / relu backward, tapes
    }
  }
#ifdef ORL_PROF
  if (prof_on && l < 12) atomicAdd(&g_rnn_prof[l], rprof_lds[l]);
  if (prof_on && l == 12) atomicAdd(&g_rnn_prof[12], 1ull);
#endif

  // ---- workgroup reduction of {dlogstd, stats}: fixed order ------------------------------------------
  __syncthreads();
  float* acc = smem;  // weights are dead
  const int PW = RnnRaw(A.net).n_logstd + ORL_N_STATS;
  const int nls = PW - ORL_N_STATS;
  for (int e = threadIdx.x; e < PW; e += blockDim.x) acc[e] = 0.f;
  __syncthreads();
  st.active = wave_sum(st.active); st.rows = wave_sum(st.rows); st.loss = wave_sum(st.loss);
  st.ent = wave_sum(st.ent); st.ratio = wave_sum(st.ratio);
#pragma unroll
  for (int c = 0; c < NO; ++c) a_dls[c] = wave_sum(a_dls[c]);
  for (int w = 0; w < nwv; ++w) {
    if (wave == w && l == 0) {
#pragma unroll
      for (int c = 0; c < NO; ++c)
        if (c < nls) acc[c] += a_dls[c];
      acc[nls + ST_ACTIVE_SUM] += st.active;
      acc[nls + ST_ROWS] += st.rows;
      if (HEAD == ORL_HEAD_VALUE) acc[nls + ST_VLOSS_SUM] += st.loss;
      else {
        acc[nls + ST_PLOSS_SUM] += st.loss;
        acc[nls + ST_ENT_SUM] += st.ent;
        acc[nls + ST_RATIO_SUM] += st.ratio;
      }
    }
    __syncthreads();
  }
  for (int e = threadIdx.x; e < PW; e += blockDim.x) A.partials[(size_t)bid * PW + e] = acc[e];
}

// blocks [0, split) = policy tower, [split, gridDim) = critic tower
template <int HEADP, int NOP>
__global__ __launch_bounds__(512, 2) void rnn_row_pair_kernel(RnnRowArgs P, RnnRowArgs Cc, int split) {
  if ((int)blockIdx.x < split) rnn_row_body<HEADP, NOP>(P, blockIdx.x, split);
  else rnn_row_body<ORL_HEAD_VALUE, 1>(Cc, blockIdx.x - split, gridDim.x - split);
}

}  // namespace orl
#if ORL_BUILD_EXPERIMENTS
#include "orl_rnn_stream.h"  // the same row kernel with its 64 x 64 GEMMs on the bf16 MFMA over streamed images (round 4)
#endif
#include "orl_rnn_l2.h"      // data_chunk_length == 2: both steps of a chunk resident in registers, no recompute (round 5)
namespace orl {

// =====================================================================================================
// update: weight-gradient GEMM over the tape
// =====================================================================================================
struct RnnWgArgs {
  orl_net_desc net;
  const float* tape;
  float* partials;  // [gridDim][raw.total - n_logstd]
  int n_blocks;     // tape blocks (tile-steps)
};

// MFMA operand of feature tile `m` of a staged 64-wide vector for k-step s (row tape_krow(s, q)), lane (c, q)
__device__ inline float tape_opnd(const float* __restrict__ v, int m, int s, int c, int q) {
  const int qq = c >> 2;
  return v[tape_off(m * 4 + qq, tape_krow(s, q)) + (c & 3)];
}

#ifndef ORL_RNN_WGRAD_REVERSE
#define ORL_RNN_WGRAD_REVERSE 1
#endif
__global__ __launch_bounds__(512, 2) void rnn_wgrad_kernel(RnnWgArgs P, RnnWgArgs Cc, int split) {
  extern __shared__ __attribute__((aligned(16))) float ring[];  // 3 slots of one tape block each
  // blocks [0, split) sweep the policy tower's tape, the rest the critic's (one launch for both)
  const bool pol = (int)blockIdx.x < split;
  const RnnWgArgs& A = pol ? P : Cc;
  const int bid = pol ? blockIdx.x : blockIdx.x - split;
  const int nblk = pol ? split : gridDim.x - split;
  const RnnRaw rl(A.net);
  const int D = A.net.obs_dim, K = A.net.n_out;
  const int ND = (D + 15) >> 4;
  const int BLK = tape_block_floats(D);
  const int wave = threadIdx.x >> 6, l = threadIdx.x & 63, c = l & 15, q = l >> 4;
  // waves 0-6: one 64 x 64 product each; wave 7: dz1 x obs (4 x ND tiles); waves 0-3 also one dhead x xhat3 tile
  //   wave   0        1        2        3        4        5          6
  //   delta  dr       dz       dn       dr       dz       dghn       dz2
  //   input  xhat2    xhat2    xhat2    h_in     h_in     h_in       xhat1
  const int g3 = wave < 3 ? wave : wave - 3;  // gate block of waves 0-5
  const int dv = wave == 6 ? TV_DZ2 : wave == 7 ? TV_DZ1 : (g3 == 0 ? TV_DR : g3 == 1 ? TV_DZ : wave == 2 ? TV_DN : TV_DGHN);
  const int iv = wave == 6 ? TV_XH1 : wave < 3 ? TV_XH2 : TV_HIN;
  f32x4 G[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) G[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  // waves 0-6: their 64 x 64 product runs on the 32x32x16 bf16 MFMA over three-term splits of the fp32 tape operands
  // (orl_mlp.h: exact hi + mid + lo, 6 of the 9 products; 24 MFMAs of 32 cycles per tape block instead of 64 fp32 MFMAs
  // of 32 cycles that also block the VALU) - 2 x 2 blocks of 32 x 32 accumulators, rows = delta features
  f32x16 GS[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) GS[a][b][r] = 0.f;
  f32x4 G5 = {0.f, 0.f, 0.f, 0.f};  // waves 0-3: dhead x xhat3 tile `wave` (S5 columns 16*wave .. +15)
  float bsum = 0.f;  // column sum of delta column `threadIdx.x` (threads 0..399)

  // Tape blocks stream HBM -> LDS by DMA (global_load_lds: no VGPRs, 1 KiB per wave instruction) through a ring of
  // three slots: while block i is multiplied, blocks i+1 and i+2 are in flight - two blocks (<= 90 KB) per CU keep
  // HBM busy, where a one-deep register prefetch left the memory pipe idle during commit + barriers (2.96 TB/s).
  const int n_inst = BLK >> 8;                       // 1 KiB DMA instructions per block (BLK is a multiple of 256)
  const int my_inst = (n_inst - wave + 7) >> 3;      // instructions k = wave, wave + 8, ... issued by this wave
  auto issue = [&](int b, int slot) {
#if ORL_RNN_WGRAD_REVERSE
    // the tape is walked from its END: the row kernel wrote the high blocks last, so they are what the 256 MiB Infinity Cache
    // still holds when this kernel starts - read in writing order, the oldest blocks come from HBM and evict the newest on the way
    const float* src = A.tape + (size_t)(A.n_blocks - 1 - b) * BLK + l * 4;
#else
    const float* src = A.tape + (size_t)b * BLK + l * 4;
#endif
    float* dst = ring + (size_t)slot * BLK;
    for (int k = wave; k < n_inst; k += 8)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + k * 256),
                                       (__attribute__((address_space(3))) void*)(dst + k * 256), 16, 0, 0);
  };
  if (bid < A.n_blocks) issue(bid, 0);
  if (bid + nblk < A.n_blocks) issue(bid + nblk, 1);
  int slot = 0;
  for (int b = bid; b < A.n_blocks; b += nblk) {
    // this wave's DMA for block b has landed when at most the younger block's instructions are outstanding
    if (b + nblk < A.n_blocks) {
      if (my_inst == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else if (my_inst == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();  // block b complete for every wave; slot (slot+2)%3 (block b - nblk) fully consumed
    if (b + 2 * nblk < A.n_blocks) issue(b + 2 * nblk, slot == 0 ? 2 : slot - 1);
    const float* blk = ring + (size_t)slot * BLK;
    slot = slot == 2 ? 0 : slot + 1;
    if (wave < 7) {
      const float* dvp = blk + dv * TV;
      const float* ivp = blk + iv * TV;
      {
        // lane (c32 = l & 31, kb = l >> 5): rows 8kb .. 8kb+7 of feature 32b + c32 of both vectors (element (row, f) of a
        // staged vector sits at tape_off(f >> 2, row) + (f & 3); with rot = g & 7 the 64 reads of an instruction hit 64 banks)
        const int c32 = l & 31, kb = l >> 5;
        u32x4 fa[2][3], fb[2][3];
#pragma unroll
        for (int b2 = 0; b2 < 2; ++b2) {
          const int f = 32 * b2 + c32, grp = f >> 2;
          float xa[8], xb[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const int off = tape_off(grp, 8 * kb + k) + (f & 3);
            xa[k] = dvp[off];
            xb[k] = ivp[off];
          }
          split8(xa, fa[b2][0], fa[b2][1], fa[b2][2]);
          split8(xb, fb[b2][0], fb[b2][1], fb[b2][2]);
        }
#pragma unroll
        for (int bo = 0; bo < 2; ++bo)
#pragma unroll
          for (int bi = 0; bi < 2; ++bi) {
            f32x16 g = GS[bo][bi];
            g = mfma_bf16_32(fa[bo][2], fb[bi][0], g);
            g = mfma_bf16_32(fa[bo][0], fb[bi][2], g);
            g = mfma_bf16_32(fa[bo][1], fb[bi][1], g);
            g = mfma_bf16_32(fa[bo][1], fb[bi][0], g);
            g = mfma_bf16_32(fa[bo][0], fb[bi][1], g);
            g = mfma_bf16_32(fa[bo][0], fb[bi][0], g);
            GS[bo][bi] = g;
          }
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        if (wave < 4) {
          const int qq = c >> 2;
          const float ah = blk[TAPE_HEAD + tape_off(qq, tape_krow(s, q)) + (c & 3)];
          G5 = ORL_MFMA(ah, tape_opnd(blk + TV_XH3 * TV, wave, s, c, q), G5);
        }
      }
    } else {
      const float* dvp = blk + TV_DZ1 * TV;
      const float* xp = blk + TAPE_X;       // ND m-blocks of 256 floats, same (qq,row,r) layout
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        float av[4], bv[4];
        const int qq = c >> 2;
        const int off = tape_off(qq, tape_krow(s, q)) + (c & 3);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          av[m] = tape_opnd(dvp, m, s, c, q);
          bv[m] = m < ND ? xp[m * 256 + off] : 0.f;
        }
#pragma unroll
        for (int mo = 0; mo < 4; ++mo)
#pragma unroll
          for (int mi = 0; mi < 4; ++mi)
            if (mi < ND) G[mo][mi] = ORL_MFMA(av[mo], bv[mi], G[mo][mi]);
      }
    }
    // bias gradients: column sums of the six delta vectors (384 columns) and of the head deltas (16)
    if (threadIdx.x < 400) {
      const int t = threadIdx.x;
      const float* base;
      int f;
      if (t < 384) { base = blk + (t >> 6) * TV; f = t & 63; }
      else { base = blk + TAPE_HEAD; f = t - 384; }
      const int g = f >> 2;  // (m, qq) group of 16 row slots; lanes of a wave start at different slots so that the
                             // 64 reads of one instruction spread over all LDS banks ((slot*4 + r) mod 32)
      const float* p = base + g * 64 + (f & 3);
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) s += p[((i + g) & 15) * 4];
      bsum += s;
    }
  }

  // ---- write this workgroup's partial row: every wave owns disjoint outputs --------------------------
  const int PW = rl.total - rl.n_logstd;
  float* out = A.partials + (size_t)bid * PW;
  // C fragment: lane (c, q), reg r -> out row 16*mo + 4q + r, in column 16*mi + c
  if (wave < 7) {
    float* o = out + (wave == 6 ? rl.oS2 : (wave < 3 ? rl.oS3 : rl.oP4) + g3 * 64 * 64);
    // 32x32 C fragment: lane (c32 = l & 31, kb = l >> 5), reg r -> row 32bo + (r & 3) + 8 (r >> 2) + 4kb, column 32bi + c32
#pragma unroll
    for (int bo = 0; bo < 2; ++bo)
#pragma unroll
      for (int bi = 0; bi < 2; ++bi)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          o[(32 * bo + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * HID + 32 * bi + (l & 31)] = GS[bo][bi][r];
    if (wave < 4) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int cc = 4 * q + r;
        if (cc < K) out[rl.oS5 + cc * HID + 16 * wave + c] = G5[r];
      }
    }
  } else {
#pragma unroll
    for (int mo = 0; mo < 4; ++mo)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int k = 16 * mi + c;
          if (mi < ND && k < D) out[rl.oP1 + (16 * mo + 4 * q + r) * D + k] = G[mo][mi][r];
        }
  }
  // bias sums: thread t < 384 holds column (vec = t/64, f = t%64); t in [384,400) head column
  __syncthreads();
  float* sh = ring;
  if (threadIdx.x < 400) sh[threadIdx.x] = bsum;
  __syncthreads();
  for (int e = threadIdx.x; e < 64; e += blockDim.x) {
    out[rl.odb1 + e] = sh[TV_DZ1 * 64 + e];
    out[rl.odb2 + e] = sh[TV_DZ2 * 64 + e];
  }
  for (int e = threadIdx.x; e < 192; e += blockDim.x) {
    const int g = e >> 6, f = e & 63;
    const float rz = sh[(g == 0 ? TV_DR : TV_DZ) * 64 + f];
    out[rl.odbih + e] = g < 2 ? rz : sh[TV_DN * 64 + f];
    out[rl.odbhh + e] = g < 2 ? rz : sh[TV_DGHN * 64 + f];
  }
  for (int e = threadIdx.x; e < K; e += blockDim.x) out[rl.odb3 + e] = sh[384 + e];
}

// =====================================================================================================
// apply
// =====================================================================================================
struct RnnApplyTower {
  orl_net_desc net;
  orl_adam_state ad;
  const float* raw;  // reduced raw sums of this tower, followed by the ORL_N_STATS stats
};

// Parameters whose gradient is a single raw sum or a two-term product.  The six LayerNorm-affine vectors
// (g1 be1 g2 be2 g3 be3) are 64-/192-/n_out-term dot products over the raw sums and are computed by the dedicated
// "dot" workgroups of rnn_grad_kernel; this function is never called for them.
__device__ inline bool rnn_is_dot_param(const RnnLayout& tl, int p) {
  return (p >= tl.og1 && p < tl.oW2) || (p >= tl.og2 && p < tl.oWih) || (p >= tl.og3 && p < tl.oW3);
}

__device__ inline float rnn_raw_to_grad(const float* __restrict__ raw, const float* __restrict__ th, const RnnLayout& tl,
                                        const RnnRaw& rl, int p, float inv_den) {
  const int H = HID;
  float g = 0.f;
  if (p < tl.ob1) g = raw[rl.oP1 + (p - tl.oW1)];
  else if (p < tl.og1) g = raw[rl.odb1 + (p - tl.ob1)];
  else if (p < tl.oW2) g = 0.f;  // g1, be1: dot workgroups
  else if (p < tl.ob2) {
    const int e = p - tl.oW2, o = e / H, i = e - o * H;
    g = th[tl.og1 + i] * raw[rl.oS2 + e] + th[tl.obe1 + i] * raw[rl.odb2 + o];
  } else if (p < tl.og2) g = raw[rl.odb2 + (p - tl.ob2)];
  else if (p < tl.oWih) g = 0.f;  // g2, be2
  else if (p < tl.oWhh) {
    const int e = p - tl.oWih, o = e / H, f = e - o * H;
    g = th[tl.og2 + f] * raw[rl.oS3 + e] + th[tl.obe2 + f] * raw[rl.odbih + o];
  } else if (p < tl.obih) g = raw[rl.oP4 + (p - tl.oWhh)];
  else if (p < tl.obhh) g = raw[rl.odbih + (p - tl.obih)];
  else if (p < tl.og3) g = raw[rl.odbhh + (p - tl.obhh)];
  else if (p < tl.oW3) g = 0.f;  // g3, be3
  else if (p < tl.ob3) {
    const int e = p - tl.oW3, cc = e / H, f = e - cc * H;
    g = th[tl.og3 + f] * raw[rl.oS5 + e] + th[tl.obe3 + f] * raw[rl.odb3 + cc];
  } else if (p < tl.ologstd) g = raw[rl.odb3 + (p - tl.ob3)];
  else g = raw[rl.odlogstd + (p - tl.ologstd)];
  return g * inv_den;
}

// blockIdx.y = tower (0 policy, 1 critic); grad[p] and per-block sums of squares -> scratch[tower*256 + block].
// Workgroups [0, nbm) take one parameter per thread; workgroups nbm + e, e = 0..5, are the "dot" workgroups of the
// LayerNorm-affine vectors g1 be1 g2 be2 g3 be3: lane f owns element f, the 4 waves split the reduction index (up to
// 192 terms, coalesced rows of W and of the raw sums) and combine through LDS - as one thread per element these loops
// were the critical path of the launch (18 us).  Their sums of squares go to scratch[tower*256 + nb_tower + e].
__global__ __launch_bounds__(256) void rnn_grad_kernel(RnnApplyTower P, RnnApplyTower Cc, orl_ppo_hparams hp,
                                                       float* __restrict__ scratch, int nbm, int nb_p, int nb_c) {
  __shared__ float sh[4][64];
  const int t = blockIdx.y;
  const RnnApplyTower& W = t == 0 ? P : Cc;
  const RnnLayout tl(W.net);
  const RnnRaw rl(W.net);
  const float* stv = W.raw + rl.total;
  const bool use_active = t == 0 ? hp.use_policy_active_masks : hp.use_value_active_masks;
  const float inv_den = 1.0f / (use_active ? stv[ST_ACTIVE_SUM] : stv[ST_ROWS]);
  const bool off = t == 0 && (hp.reserved & 1);
  const int nb_t = t == 0 ? nb_p : nb_c;
  if ((int)blockIdx.x >= nbm) {
    const int e = (int)blockIdx.x - nbm, w = threadIdx.x >> 6, f = threadIdx.x & 63;
    const int p0 = e == 0 ? tl.og1 : e == 1 ? tl.obe1 : e == 2 ? tl.og2 : e == 3 ? tl.obe2 : e == 4 ? tl.og3 : tl.obe3;
    const int th0 = e < 2 ? tl.oW2 : e < 4 ? tl.oWih : tl.oW3;
    const int len = e < 2 ? HID : e < 4 ? 3 * HID : tl.n_out;
    const int mat = e == 0 ? rl.oS2 : e == 2 ? rl.oS3 : rl.oS5;          // even e: elementwise against a raw matrix
    const int vec = e == 1 ? rl.odb2 : e == 3 ? rl.odbih : rl.odb3;      // odd e : against a raw bias-sum vector
    const float* th = W.ad.theta;
    float s = 0.f;
    if (!off) {
      if (e & 1) {
#pragma unroll 4
        for (int o = w; o < len; o += 4) s += th[th0 + o * HID + f] * W.raw[vec + o];
      } else {
#pragma unroll 4
        for (int o = w; o < len; o += 4) s += th[th0 + o * HID + f] * W.raw[mat + o * HID + f];
      }
    }
    sh[w][f] = s;
    __syncthreads();
    if (w == 0) {
      const float g = ((sh[0][f] + sh[1][f]) + (sh[2][f] + sh[3][f])) * inv_den;
      W.ad.grad[p0 + f] = g;
      const float ss = wave_sum(g * g);
      if (f == 0) scratch[t * 256 + nb_t + e] = ss;
    }
    return;
  }
  if ((int)blockIdx.x >= nb_t) return;  // padding workgroups of the smaller tower
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  float g = 0.f;
  if (p < tl.total && !rnn_is_dot_param(tl, p)) {
    if (!off) g = rnn_raw_to_grad(W.raw, W.ad.theta, tl, rl, p, inv_den);
    W.ad.grad[p] = g;
  }
  float ss = wave_sum(g * g);
  if ((threadIdx.x & 63) == 0) sh[0][threadIdx.x >> 6] = ss;
  __syncthreads();
  if (threadIdx.x == 0) scratch[t * 256 + blockIdx.x] = (sh[0][0] + sh[0][1]) + (sh[0][2] + sh[0][3]);
}

__global__ __launch_bounds__(256) void rnn_adam_kernel(RnnApplyTower P, RnnApplyTower Cc, orl_ppo_hparams hp,
                                                       const float* __restrict__ scratch, int nb_p, int nb_c,
                                                       float* __restrict__ info) {
  const int t = blockIdx.y;
  const RnnApplyTower& W = t == 0 ? P : Cc;
  const RnnLayout tl(W.net);
  const RnnRaw rl(W.net);
  const float* stv = W.raw + rl.total;
  const int nb = t == 0 ? nb_p : nb_c;
  if ((int)blockIdx.x >= nb) return;
  float ss = 0.f;
  for (int b = 0; b < nb + 6; ++b) ss += scratch[t * 256 + b];  // same order in every block: identical norm (+6 dot groups)
  const float total = sqrtf(ss);
  const bool off = t == 0 && (hp.reserved & 1);  // turn_on == False: no policy step (ppo.py:226-236)
  if (!off) {
    float coef = 1.f;
    if (hp.use_max_grad_norm) coef = fminf(hp.max_grad_norm / (total + 1e-6f), 1.f);
    const double b1d = 0.9, b2d = 0.999;
    const float b2 = (float)b2d, omb1 = (float)(1.0 - b1d), omb2 = (float)(1.0 - b2d);
    const double bc1 = 1.0 - powi_d(b1d, (long long)W.ad.step);
    const double bc2 = 1.0 - powi_d(b2d, (long long)W.ad.step);
    const float step_size = (float)((double)W.ad.lr / bc1);
    const float bc2_sqrt = (float)sqrt(bc2);
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < tl.total) {
      float g = W.ad.grad[p] * coef;
      W.ad.grad[p] = g;
      float th = W.ad.theta[p];
      if (W.ad.weight_decay != 0.f) g += W.ad.weight_decay * th;
      float m = W.ad.m[p], v = W.ad.v[p];
      m = m + (g - m) * omb1;
      v = v * b2 + omb2 * (g * g);
      const float denom = sqrtf(v) / bc2_sqrt + W.ad.eps;
      th = th - step_size * (m / denom);
      W.ad.m[p] = m; W.ad.v[p] = v; W.ad.theta[p] = th;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && info != nullptr) {
    if (t == 0) {
      const float den_p = hp.use_policy_active_masks ? stv[ST_ACTIVE_SUM] : stv[ST_ROWS];
      const bool gauss = P.net.head_kind == ORL_HEAD_GAUSSIAN;
      float ent_den = den_p;
      if (!hp.use_policy_active_masks && gauss) ent_den = den_p * (float)P.net.n_out;
      info[1] += stv[ST_PLOSS_SUM] / den_p;
      info[2] += stv[ST_ENT_SUM] / ent_den;
      if (!off) info[3] += total;
      info[5] += stv[ST_RATIO_SUM] / (stv[ST_ROWS] * (gauss ? (float)P.net.n_out : 1.f));
    } else {
      const float den_v = hp.use_value_active_masks ? stv[ST_ACTIVE_SUM] : stv[ST_ROWS];
      info[0] += stv[ST_VLOSS_SUM] / den_v;
      info[4] += total;
    }
  }
}

// column sums of two partial regions in one launch (fixed order => deterministic), each to its own output
__global__ __launch_bounds__(1024) void rnn_reduce_pair_kernel(const float* __restrict__ pa, int nb_a, int wa, int ga,
                                                               float* __restrict__ oa, const float* __restrict__ pb,
                                                               int nb_b, int wb, float* __restrict__ ob) {
  // 64 columns x 16 row groups per workgroup (same shape as ppo_reduce_pair_kernel, orl_apply.hip)
  __shared__ float sh[16][64];
  const bool first = (int)blockIdx.x < ga;
  const float* partials = first ? pa : pb;
  const int n_blocks = first ? nb_a : nb_b, width = first ? wa : wb;
  float* out = first ? oa : ob;
  const int lc = threadIdx.x & 63;
  const int col = (first ? blockIdx.x : blockIdx.x - ga) * 64 + lc;
  const int rg = threadIdx.x >> 6;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (col < width) {
    int b = rg;
    for (; b + 48 < n_blocks; b += 64) {
      s0 += partials[(size_t)b * width + col];
      s1 += partials[(size_t)(b + 16) * width + col];
      s2 += partials[(size_t)(b + 32) * width + col];
      s3 += partials[(size_t)(b + 48) * width + col];
    }
    for (; b < n_blocks; b += 16) s0 += partials[(size_t)b * width + col];
  }
  sh[rg][lc] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (rg == 0 && col < width) {
    float t[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) t[k] = (sh[4 * k][lc] + sh[4 * k + 1][lc]) + (sh[4 * k + 2][lc] + sh[4 * k + 3][lc]);
    out[col] = (t[0] + t[1]) + (t[2] + t[3]);
  }
}

// Round 5: the four partial regions of a recurrent optimiser step (weight-gradient partials and row-kernel partials of both
// towers) in ONE launch instead of two launches of rnn_reduce_pair_kernel - same per-column arithmetic and order
struct RnnReduce4 {
  const float* p[4];
  float* o[4];
  int nb[4], w[4], g0[5];  // g0[k] = first workgroup of region k (g0[4] = grid)
};
__global__ __launch_bounds__(1024) void rnn_reduce4_kernel(RnnReduce4 R) {
  __shared__ float sh[16][64];
  int k = 0;
#pragma unroll
  for (int i = 1; i < 4; ++i) k = (int)blockIdx.x >= R.g0[i] ? i : k;
  const float* partials = k == 0 ? R.p[0] : k == 1 ? R.p[1] : k == 2 ? R.p[2] : R.p[3];
  float* out = k == 0 ? R.o[0] : k == 1 ? R.o[1] : k == 2 ? R.o[2] : R.o[3];
  const int n_blocks = k == 0 ? R.nb[0] : k == 1 ? R.nb[1] : k == 2 ? R.nb[2] : R.nb[3];
  const int width = k == 0 ? R.w[0] : k == 1 ? R.w[1] : k == 2 ? R.w[2] : R.w[3];
  const int gbase = k == 0 ? R.g0[0] : k == 1 ? R.g0[1] : k == 2 ? R.g0[2] : R.g0[3];
  const int lc = threadIdx.x & 63;
  const int col = ((int)blockIdx.x - gbase) * 64 + lc;
  const int rg = threadIdx.x >> 6;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (col < width) {
    int b = rg;
    for (; b + 48 < n_blocks; b += 64) {
      s0 += partials[(size_t)b * width + col];
      s1 += partials[(size_t)(b + 16) * width + col];
      s2 += partials[(size_t)(b + 32) * width + col];
      s3 += partials[(size_t)(b + 48) * width + col];
    }
    for (; b < n_blocks; b += 16) s0 += partials[(size_t)b * width + col];
  }
  sh[rg][lc] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (rg == 0 && col < width) {
    float t[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) t[q] = (sh[4 * q][lc] + sh[4 * q + 1][lc]) + (sh[4 * q + 2][lc] + sh[4 * q + 3][lc]);
    out[col] = (t[0] + t[1]) + (t[2] + t[3]);
  }
}

__attribute__((unused)) static int orl_ppo_reduce_pair2(const float* pa, int nb_a, int wa, float* oa, const float* pb, int nb_b, int wb,
                                float* ob, hipStream_t s) {
  const int ga = (wa + 63) / 64, gb = (wb + 63) / 64;
  hipLaunchKernelGGL(rnn_reduce_pair_kernel, dim3(ga + gb), dim3(1024), 0, s, pa, nb_a, wa, ga, oa, pb, nb_b, wb, ob);
  return launch_status("orl_rnn_ppo_fwd_bwd(reduce)");
}

// ---- host helpers ------------------------------------------------------------------------------------
static int check_rnn_net(const orl_net_desc* n, const char* who, bool value) {
  if (!n) return fail(ORL_E_INVALID, "%s: null net descriptor", who);
  if (n->hidden != HID) return fail(ORL_E_UNSUPPORTED, "%s: hidden_size %d not built (only 64)", who, n->hidden);
  if (n->obs_dim < 1 || n->obs_dim > 64) return fail(ORL_E_UNSUPPORTED, "%s: obs_dim %d outside [1,64]", who, n->obs_dim);
  if (n->n_out < 1 || n->n_out > 16) return fail(ORL_E_UNSUPPORTED, "%s: n_out %d outside [1,16]", who, n->n_out);
  if (value && !(n->head_kind == ORL_HEAD_VALUE && n->n_out == 1)) return fail(ORL_E_INVALID, "%s: not a value head", who);
  if (!value && n->head_kind != ORL_HEAD_CATEGORICAL && n->head_kind != ORL_HEAD_GAUSSIAN)
    return fail(ORL_E_INVALID, "%s: policy head kind %d not built", who, n->head_kind);
  return 0;
}

struct RnnWs {  // workspace carve-up (floats) for one tower
  size_t htape, tape, rpart, wpart, img, total;
  int n_tiles, n_blocks, grid_row, grid_wg, rpw, wpw;
  RnnWs(const orl_net_desc& n, int n_chunks, int L) {
    const RnnRaw rl(n);
    n_tiles = (n_chunks + TILE_B - 1) / TILE_B;
    n_blocks = n_tiles * L;
    // each tower gets half of the CUs: its twin runs in the same launch
    grid_row = (n_tiles + 7) / 8;
    if (grid_row > RNN_ROW_BLOCKS / 2) grid_row = RNN_ROW_BLOCKS / 2;
    grid_wg = n_blocks < RNN_WG_BLOCKS / 2 ? n_blocks : RNN_WG_BLOCKS / 2;
    rpw = rl.n_logstd + ORL_N_STATS;
    wpw = rl.total - rl.n_logstd;
    size_t o = 0;
    htape = o; o += (size_t)n_blocks * TV;
    tape = o; o += (size_t)n_blocks * tape_block_floats(n.obs_dim);
    rpart = o; o += (size_t)RNN_ROW_BLOCKS * rpw;
    wpart = o; o += (size_t)RNN_WG_BLOCKS * wpw;
    o = (o + 63) & ~(size_t)63;
    img = o;
#if ORL_BUILD_EXPERIMENTS
    o += (size_t)RS_NIMG * RS_IMG_FLOATS + 128;  // bf16 images of W2 / Wih / Whh + the stream's two flag rows (streamed kernel)
#endif
    total = (o + 63) & ~(size_t)63;
  }
};

// the streamed split kernel: images first (one small launch per optimiser step), then the row pair
#if ORL_BUILD_EXPERIMENTS
template <int HEAD, int NO>
static int launch_rnn_rows_stream(const RnnRowArgs& P, const RnnRowArgs& Cc, float* img_p, float* img_c, int grid_p,
                                  int grid_c, hipStream_t s) {
  const RnnLds twp(P.net.obs_dim, P.net.n_out, HEAD == ORL_HEAD_GAUSSIAN, HEAD == ORL_HEAD_CATEGORICAL && NO > 4, true);
  const RnnLds twc(Cc.net.obs_dim, 1, false, false, true);
  const size_t lds = (size_t)((twp.total > twc.total ? twp.total : twc.total) + RS_NSLOT * RS_IMG_FLOATS +
                             RnnStream::extra_floats()) * sizeof(float);
  if (lds > 160 * 1024) return fail(ORL_E_UNSUPPORTED, "orl_rnn_ppo_fwd_bwd: tower needs %zu B of LDS", lds);
  hipLaunchKernelGGL(rnn_images_kernel, dim3(RS_NIMG, 2), dim3(256), 0, s, P.theta, Cc.theta, RnnLayout(P.net),
                     RnnLayout(Cc.net), img_p, img_c);
  int rc = launch_status("orl_rnn_ppo_fwd_bwd(images)");
  if (rc) return rc;
  if (P.hp.reserved & 16) {  // 4 waves per workgroup: one wave per SIMD, 512 registers (the caller sized the grid for it)
    (void)hipFuncSetAttribute((const void*)rnn_row_pair_stream_kernel<HEAD, NO, 4>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((rnn_row_pair_stream_kernel<HEAD, NO, 4>), dim3(grid_p + grid_c), dim3(256), lds, s, P, Cc,
                       (const float*)img_p, (const float*)img_c, grid_p);
    return launch_status("orl_rnn_ppo_fwd_bwd(row, streamed split, 4 waves)");
  }
  (void)hipFuncSetAttribute((const void*)rnn_row_pair_stream_kernel<HEAD, NO, 8>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds);
  hipLaunchKernelGGL((rnn_row_pair_stream_kernel<HEAD, NO, 8>), dim3(grid_p + grid_c), dim3(512), lds, s, P, Cc,
                     (const float*)img_p, (const float*)img_c, grid_p);
  return launch_status("orl_rnn_ppo_fwd_bwd(row, streamed split)");
}
#endif

// data_chunk_length == 2: rnn_row2_pair_kernel, 4 waves per workgroup (one per SIMD, 512 registers), one workgroup per CU
template <int HEAD, int NO>
static int launch_rnn_rows_l2(const RnnRowArgs& P, const RnnRowArgs& Cc, int grid_p, int grid_c, hipStream_t s) {
  const RnnLds twp(P.net.obs_dim, P.net.n_out, HEAD == ORL_HEAD_GAUSSIAN, HEAD == ORL_HEAD_CATEGORICAL && NO > 4, false, false,
                   rnn_l2_h2(HEAD, NO));
  const RnnLds twc(Cc.net.obs_dim, 1, false, false, false, false, rnn_l2_h2(ORL_HEAD_VALUE, 1));
  const size_t lds = (size_t)(twp.total > twc.total ? twp.total : twc.total) * sizeof(float);
  if (lds > 160 * 1024) return fail(ORL_E_UNSUPPORTED, "orl_rnn_ppo_fwd_bwd: tower needs %zu B of LDS", lds);
  (void)hipFuncSetAttribute((const void*)rnn_row2_pair_kernel<HEAD, NO>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds);
  hipLaunchKernelGGL((rnn_row2_pair_kernel<HEAD, NO>), dim3(grid_p + grid_c), dim3(256), lds, s, P, Cc, grid_p);
  return launch_status("orl_rnn_ppo_fwd_bwd(row, L = 2)");
}

template <int HEAD, int NO>
static int launch_rnn_rows(const RnnRowArgs& P, const RnnRowArgs& Cc, int grid_p, int grid_c, hipStream_t s) {
  const RnnLds twp(P.net.obs_dim, P.net.n_out, HEAD == ORL_HEAD_GAUSSIAN, HEAD == ORL_HEAD_CATEGORICAL && NO > 4);
  const RnnLds twc(Cc.net.obs_dim, 1, false);
  const size_t lds = (size_t)(twp.total > twc.total ? twp.total : twc.total) * sizeof(float);
  if (lds > 160 * 1024) return fail(ORL_E_UNSUPPORTED, "orl_rnn_ppo_fwd_bwd: tower needs %zu B of LDS", lds);
  (void)hipFuncSetAttribute((const void*)rnn_row_pair_kernel<HEAD, NO>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds);
  hipLaunchKernelGGL((rnn_row_pair_kernel<HEAD, NO>), dim3(grid_p + grid_c), dim3(512), lds, s, P, Cc, grid_p);
  return launch_status("orl_rnn_ppo_fwd_bwd(row)");
}

}  // namespace orl

using namespace orl;

extern "C" {

#ifdef ORL_PROF
int orl_debug_rnn_prof(unsigned long long* out16) {
  unsigned long long zero[16] = {0};
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_rnn_prof), sizeof(zero));
  hipMemcpyToSymbol(HIP_SYMBOL(g_rnn_prof), zero, sizeof(zero));
  return 0;
}
#endif

int orl_rnn_param_count(const orl_net_desc* net) {
  if (!net) return fail(ORL_E_INVALID, "orl_rnn_param_count: null");
  return RnnLayout(*net).total;
}

int orl_rnn_raw_grad_count(const orl_net_desc* net) {
  if (!net) return fail(ORL_E_INVALID, "orl_rnn_raw_grad_count: null");
  return RnnRaw(*net).total;
}

static int rnn_act_impl(const orl_net_desc* pnet, const float* ptheta, const orl_net_desc* cnet, const float* ctheta,
                        const float* policy_obs, const float* critic_obs, const float* h_policy_in,
                        const float* h_critic_in, const float* masks, const float* action_masks, int B,
                        int deterministic, uint64_t seed, uint64_t row0, uint64_t rng_step, const uint64_t* rng_step_dev,
                        const float* forced_u, const float* given, float* ent_out, float* values, float* actions,
                        float* logp, float* h_policy_out, float* h_critic_out, void* stream) {
  ORL_REQUIRE(ptheta || ctheta, "orl_rnn_act_step: neither tower given");
  ORL_REQUIRE(B > 0 && masks, "orl_rnn_act_step: B=%d / null masks", B);
  int rc;
  if (ptheta) {
    rc = check_rnn_net(pnet, "orl_rnn_act_step(policy)", false);
    if (rc) return rc;
    ORL_REQUIRE(policy_obs && (actions || given) && logp && h_policy_in && h_policy_out,
                "orl_rnn_act_step: null policy pointer");
  }
  if (ctheta) {
    rc = check_rnn_net(cnet, "orl_rnn_act_step(critic)", true);
    if (rc) return rc;
    ORL_REQUIRE(critic_obs && values && h_critic_in && h_critic_out, "orl_rnn_act_step: null critic pointer");
  }
  RnnActArgs A;
  A.pnet = ptheta ? *pnet : *cnet;
  A.cnet = ctheta ? *cnet : *pnet;
  A.ptheta = ptheta; A.ctheta = ctheta; A.pobs = policy_obs; A.cobs = critic_obs; A.hp_in = h_policy_in;
  A.hc_in = h_critic_in; A.masks = masks; A.amask = action_masks; A.forced = forced_u; A.values = values;
  A.given = given; A.ent_out = ent_out;
  A.actions = actions; A.logp = logp; A.hp_out = h_policy_out; A.hc_out = h_critic_out; A.B = B;
  A.deterministic = deterministic; A.seed = seed; A.row0 = row0; A.rng_step = rng_step; A.rng_dev = (const unsigned long long*)rng_step_dev;
  const int grid = (B + TILE_B - 1) / TILE_B;
  hipStream_t s = (hipStream_t)stream;
  const int no = A.pnet.n_out;
  if (ptheta && ctheta && grid >= 128) {  // big batch, both towers: LDS-staged workgroups (rnn_act_lds_kernel)
    int tpw = (grid + 95) / 96;           // tiles per workgroup: ~96 workgroups per tower, 4 waves each
    if (tpw < 4) tpw = 4;
    const int gw = (grid + tpw - 1) / tpw;
    const RnnLds twp(A.pnet.obs_dim, A.pnet.n_out, A.pnet.head_kind == ORL_HEAD_GAUSSIAN), twc(A.cnet.obs_dim, 1, false);
    const size_t lds = (size_t)(twp.total > twc.total ? twp.total : twc.total) * sizeof(float);
    if (lds <= 160 * 1024) {
#define ORL_RNN_ACT_LDS(NO, HD)                                                                                   \
  do {                                                                                                            \
    (void)hipFuncSetAttribute((const void*)rnn_act_lds_kernel<NO, HD>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                              (int)lds);                                                                          \
    hipLaunchKernelGGL((rnn_act_lds_kernel<NO, HD>), dim3(2 * gw), dim3(256), lds, s, A, gw, tpw);                \
  } while (0)
      if (A.pnet.head_kind == ORL_HEAD_CATEGORICAL) {
        if (no <= 2) ORL_RNN_ACT_LDS(2, ORL_HEAD_CATEGORICAL);
        else if (no <= 8) ORL_RNN_ACT_LDS(8, ORL_HEAD_CATEGORICAL);
        else ORL_RNN_ACT_LDS(16, ORL_HEAD_CATEGORICAL);
      } else {
        if (no <= 2) ORL_RNN_ACT_LDS(2, ORL_HEAD_GAUSSIAN);
        else if (no <= 8) ORL_RNN_ACT_LDS(8, ORL_HEAD_GAUSSIAN);
        else ORL_RNN_ACT_LDS(16, ORL_HEAD_GAUSSIAN);
      }
#undef ORL_RNN_ACT_LDS
      return launch_status("orl_rnn_act_step");
    }
  }
#define ORL_RNN_ACT(NO, HD) hipLaunchKernelGGL((rnn_act_kernel<NO, HD>), dim3(grid), dim3(128), 0, s, A)
  if (!ptheta || A.pnet.head_kind == ORL_HEAD_VALUE) ORL_RNN_ACT(2, ORL_HEAD_CATEGORICAL);  // value-only: wave 0 exits
  else if (A.pnet.head_kind == ORL_HEAD_CATEGORICAL) {
    if (no <= 2) ORL_RNN_ACT(2, ORL_HEAD_CATEGORICAL);
    else if (no <= 8) ORL_RNN_ACT(8, ORL_HEAD_CATEGORICAL);
    else ORL_RNN_ACT(16, ORL_HEAD_CATEGORICAL);
  } else {
    if (no <= 2) ORL_RNN_ACT(2, ORL_HEAD_GAUSSIAN);
    else if (no <= 8) ORL_RNN_ACT(8, ORL_HEAD_GAUSSIAN);
    else ORL_RNN_ACT(16, ORL_HEAD_GAUSSIAN);
  }
#undef ORL_RNN_ACT
  return launch_status("orl_rnn_act_step");
}

int orl_rnn_act_step(const orl_net_desc* pnet, const float* ptheta, const orl_net_desc* cnet, const float* ctheta,
                     const float* policy_obs, const float* critic_obs, const float* h_policy_in,
                     const float* h_critic_in, const float* masks, const float* action_masks, int B,
                     int deterministic, uint64_t seed, uint64_t row0, uint64_t rng_step, const uint64_t* rng_step_dev,
                     const float* forced_u, float* values, float* actions, float* logp, float* h_policy_out,
                     float* h_critic_out, void* stream) {
  return rnn_act_impl(pnet, ptheta, cnet, ctheta, policy_obs, critic_obs, h_policy_in, h_critic_in, masks, action_masks,
                      B, deterministic, seed, row0, rng_step, rng_step_dev, forced_u, nullptr, nullptr, values, actions,
                      logp, h_policy_out, h_critic_out, stream);
}

int orl_rnn_eval_step(const orl_net_desc* pnet, const float* ptheta, const orl_net_desc* cnet, const float* ctheta,
                      const float* policy_obs, const float* critic_obs, const float* h_policy_in,
                      const float* h_critic_in, const float* masks, const float* action_masks, const float* actions,
                      int B, float* values, float* logp, float* entropy, float* h_policy_out, float* h_critic_out,
                      void* stream) {
  ORL_REQUIRE(!ptheta || actions, "orl_rnn_eval_step: the policy tower needs the actions to evaluate");
  return rnn_act_impl(pnet, ptheta, cnet, ctheta, policy_obs, critic_obs, h_policy_in, h_critic_in, masks, action_masks,
                      B, 1, 0, 0, 0, nullptr, nullptr, actions, entropy, values, nullptr, logp, h_policy_out,
                      h_critic_out, stream);
}

int orl_rnn_chunk_rows(const int64_t* chunk_idx, int n_chunks, int L, int T, int lanes, int64_t* rows, void* stream) {
  ORL_REQUIRE(rows && n_chunks > 0 && L > 0 && T > 0 && lanes > 0, "orl_rnn_chunk_rows: bad arguments");
  const int n = n_chunks * L;
  hipLaunchKernelGGL(rnn_chunk_rows_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, chunk_idx,
                     n_chunks, L, T, lanes, rows);
  return launch_status("orl_rnn_chunk_rows");
}

int orl_rnn_chunk_rows_v3(const int64_t* chunk_idx, int n_chunks, int L, int T, int n_envs, int n_agents, int agent0_only,
                          int64_t* rows, void* stream) {
  ORL_REQUIRE(rows && n_chunks > 0 && L > 0 && T > 0 && n_envs > 0 && n_agents > 0, "orl_rnn_chunk_rows_v3: bad arguments");
  const int n = n_chunks * (agent0_only ? 1 : n_agents) * L;
  hipLaunchKernelGGL(rnn_chunk_rows_v3_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, chunk_idx,
                     n_chunks, L, T, n_agents, n_envs * n_agents, agent0_only, rows);
  return launch_status("orl_rnn_chunk_rows_v3");
}

int orl_rnn_jrpo_records(const float* records, float* records_out, int rec_width, int Dp, int Dc, int a_w,
                         const int64_t* rows, int n_chunks, int L, int n_agents, const float* logp_new, void* stream) {
  ORL_REQUIRE(records && records_out && rows && logp_new && n_chunks > 0 && L > 0 && n_agents > 0 && a_w > 0,
              "orl_rnn_jrpo_records: bad arguments");
  const int o_lp = Dp + Dc + a_w, o_adv = o_lp + a_w;
  const int n = n_chunks * L;
  hipLaunchKernelGGL(rnn_jrpo_records_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, records,
                     records_out, rec_width, rows, n_chunks, L, n_agents, logp_new, a_w, o_lp, o_adv, o_adv + 3);
  return launch_status("orl_rnn_jrpo_records");
}

int64_t orl_rnn_workspace_floats(const orl_net_desc* pnet, const orl_net_desc* cnet, int n_chunks, int L) {
  if (!pnet || !cnet || n_chunks <= 0 || L <= 0) return fail(ORL_E_INVALID, "orl_rnn_workspace_floats: bad arguments");
  return (int64_t)(RnnWs(*pnet, n_chunks, L).total + RnnWs(*cnet, n_chunks, L).total);
}

int orl_rnn_ppo_fwd_bwd(const orl_net_desc* pnet, const float* ptheta, const orl_net_desc* cnet, const float* ctheta,
                        const orl_rnn_batch* batch, const float* vn_state, const orl_ppo_hparams* hp,
                        float* workspace, float* sums, void* stream) {
  int rc = check_rnn_net(pnet, "orl_rnn_ppo_fwd_bwd(policy)", false);
  if (rc) return rc;
  rc = check_rnn_net(cnet, "orl_rnn_ppo_fwd_bwd(critic)", true);
  if (rc) return rc;
  ORL_REQUIRE(ptheta && ctheta && batch && hp && workspace && sums, "orl_rnn_ppo_fwd_bwd: null pointer");
  ORL_REQUIRE(batch->records && batch->rows && batch->masks && batch->h_policy && batch->h_critic,
              "orl_rnn_ppo_fwd_bwd: null batch pointer");
  ORL_REQUIRE(batch->n_chunks > 0 && batch->L > 0, "orl_rnn_ppo_fwd_bwd: empty minibatch");
  const int a_w = pnet->head_kind == ORL_HEAD_CATEGORICAL ? 1 : pnet->n_out;
  const int Dp = pnet->obs_dim, Dc = cnet->obs_dim;
  const int K = pnet->head_kind == ORL_HEAD_CATEGORICAL ? pnet->n_out : 0;
  ORL_REQUIRE(orl_record_width(Dp, Dc, a_w, K) == batch->rec_width, "orl_rnn_ppo_fwd_bwd: record width %d != %d",
              batch->rec_width, orl_record_width(Dp, Dc, a_w, K));
  hipStream_t s = (hipStream_t)stream;
  const int o_co = Dp, o_ac = o_co + Dc, o_lp = o_ac + a_w, o_adv = o_lp + a_w;
  RnnRowArgs A;
  A.records = batch->records; A.rows = batch->rows; A.masks = batch->masks; A.vn_state = vn_state; A.hp = *hp;
  A.R = batch->rec_width; A.Nc = batch->n_chunks; A.L = batch->L;
  A.cols.o_act = o_ac; A.cols.o_lp = o_lp; A.cols.o_adv = o_adv; A.cols.o_vp = o_adv + 1; A.cols.o_rt = o_adv + 2;
  A.cols.o_am = o_adv + 3; A.cols.o_mk = o_adv + 4; A.cols.K = K;
  const int nc_c = batch->rows_critic ? batch->n_chunks_critic : batch->n_chunks;  // joint-action loss: agent 0 only
  ORL_REQUIRE(nc_c > 0 && nc_c <= batch->n_chunks, "orl_rnn_ppo_fwd_bwd: %d critic sequences", nc_c);
  const RnnWs wp(*pnet, batch->n_chunks, batch->L), wc(*cnet, nc_c, batch->L);
  float* base_p = workspace;
  float* base_c = workspace + wp.total;
  RnnRowArgs C2 = A;
  A.net = *pnet; A.theta = ptheta; A.hbuf = batch->h_policy; A.o_x = 0;
  A.htape = base_p + wp.htape; A.tape = base_p + wp.tape; A.partials = base_p + wp.rpart;
  C2.net = *cnet; C2.theta = ctheta; C2.hbuf = batch->h_critic; C2.o_x = o_co;
  if (batch->rows_critic) { C2.rows = batch->rows_critic; C2.Nc = nc_c; }
  C2.htape = base_c + wc.htape; C2.tape = base_c + wc.tape; C2.partials = base_c + wc.rpart;
  // (1) row kernels of both towers, one launch.  hp.reserved selects the build (include/orl_hip.h):
  //   0 (the default): data_chunk_length == 2 -> rnn_row2_pair_kernel (both steps of a chunk in registers, no forward
  //     recompute, no state tape); any other length -> rnn_row_pair_kernel (forward sweep + BPTT sweep with recompute);
  //   4: rnn_row_pair_kernel whatever the length (comparison switch, cfg.amd_rnn_gemm = fp32_recompute);
  //   8 (| 16): the streamed bf16-split row kernel of round 4 with 8 (4) waves per workgroup (comparison switches)
  const int no = pnet->n_out;
  int grid_rp = wp.grid_row, grid_rc = wc.grid_row;  // workgroups (= rows of the row partials) per tower
  const bool streamed = (hp->reserved & 8) != 0;
  // (round 6, ADVICE r5: undefined combinations are refused instead of silently meaning something)
  if ((hp->reserved & 4) && (hp->reserved & (8 | 16)))
    return fail(ORL_E_INVALID, "orl_rnn_ppo_fwd_bwd: hparams.reserved GEMM bits %d: 4 (recompute kernel) excludes 8 / 16 (streamed "
                "kernels)", hp->reserved & 28);
  if ((hp->reserved & 16) && !streamed)
    return fail(ORL_E_INVALID, "orl_rnn_ppo_fwd_bwd: hparams.reserved & 16 (4 waves per workgroup) needs & 8 (the streamed kernel)");
#if !ORL_BUILD_EXPERIMENTS
  if (streamed)
    return fail(ORL_E_UNSUPPORTED, "orl_rnn_ppo_fwd_bwd: hparams.reserved & 8 selects the streamed bf16-split row kernel, which this "
                "library was built without (ORL_BUILD_EXPERIMENTS)");
#endif
  const bool l2 = !streamed && !(hp->reserved & 4) && batch->L == 2;
  if ((streamed && (hp->reserved & 16)) || l2) {  // 4 waves per workgroup: 4 tiles per group and round
    grid_rp = (wp.n_tiles + 3) / 4;
    grid_rc = (wc.n_tiles + 3) / 4;
    if (grid_rp > RNN_ROW_BLOCKS / 2) grid_rp = RNN_ROW_BLOCKS / 2;
    if (grid_rc > RNN_ROW_BLOCKS / 2) grid_rc = RNN_ROW_BLOCKS / 2;
    // Both towers capped at half of the CUs: the split follows the towers' cost per tile and minimises the later tower's
    // finish in whole rounds of tiles (as launch_pair_nd does for the feed-forward pair).  A policy tile with a wide head
    // (MFMA head + distributed loss) costs ~1.1 critic tiles: at cfg4 (4 800 tiles per tower) 136 / 120 workgroups =
    // 9 policy rounds against 10 critic rounds instead of 10 / 10 - 0.728 -> 0.718 ms per epoch in two alternations
    // (profiles/r05_experiments.md; 120 / 136 and 124 / 132 measured no better than even).
    if (l2 && grid_rp == RNN_ROW_BLOCKS / 2 && grid_rc == RNN_ROW_BLOCKS / 2) {
#ifdef ORL_RNN_L2_GRID_P  // build-time experiment: a fixed split
      grid_rp = ORL_RNN_L2_GRID_P;
#else
      const double w_p = (pnet->head_kind == ORL_HEAD_CATEGORICAL && no > 4) ? 1.10 : 1.0;
      double best = 1e30;
      for (int d = 0; d <= RNN_ROW_BLOCKS / 4; ++d) {  // nearest to an even split first: ties keep it
        for (int sgn = 1; sgn >= -1; sgn -= 2) {
          const int g = RNN_ROW_BLOCKS / 2 + sgn * d;
          const double tp = (double)((wp.n_tiles + 4 * g - 1) / (4 * g)) * w_p;
          const double tc = (double)((wc.n_tiles + 4 * (RNN_ROW_BLOCKS - g) - 1) / (4 * (RNN_ROW_BLOCKS - g)));
          const double t = tp > tc ? tp : tc;
          if (t < best - 1e-9) { best = t; grid_rp = g; }
          if (d == 0) break;
        }
      }
#endif
      grid_rc = RNN_ROW_BLOCKS - grid_rp;
    }
  }
  if (l2) {
    if (pnet->head_kind == ORL_HEAD_CATEGORICAL) {
      if (no <= 2) rc = launch_rnn_rows_l2<ORL_HEAD_CATEGORICAL, 2>(A, C2, grid_rp, grid_rc, s);
      else if (no <= 8) rc = launch_rnn_rows_l2<ORL_HEAD_CATEGORICAL, 8>(A, C2, grid_rp, grid_rc, s);
      else rc = launch_rnn_rows_l2<ORL_HEAD_CATEGORICAL, 16>(A, C2, grid_rp, grid_rc, s);
    } else {
      if (no <= 8) rc = launch_rnn_rows_l2<ORL_HEAD_GAUSSIAN, 8>(A, C2, grid_rp, grid_rc, s);
      else rc = launch_rnn_rows_l2<ORL_HEAD_GAUSSIAN, 16>(A, C2, grid_rp, grid_rc, s);
    }
  } else if (!streamed) {
    if (pnet->head_kind == ORL_HEAD_CATEGORICAL) {
      if (no <= 2) rc = launch_rnn_rows<ORL_HEAD_CATEGORICAL, 2>(A, C2, wp.grid_row, wc.grid_row, s);
      else if (no <= 8) rc = launch_rnn_rows<ORL_HEAD_CATEGORICAL, 8>(A, C2, wp.grid_row, wc.grid_row, s);
      else rc = launch_rnn_rows<ORL_HEAD_CATEGORICAL, 16>(A, C2, wp.grid_row, wc.grid_row, s);
    } else {
      if (no <= 8) rc = launch_rnn_rows<ORL_HEAD_GAUSSIAN, 8>(A, C2, wp.grid_row, wc.grid_row, s);
      else rc = launch_rnn_rows<ORL_HEAD_GAUSSIAN, 16>(A, C2, wp.grid_row, wc.grid_row, s);
    }
  } else {
#if ORL_BUILD_EXPERIMENTS
    float* ip = base_p + wp.img;
    float* ic = base_c + wc.img;
    if (pnet->head_kind == ORL_HEAD_CATEGORICAL) {
      if (no <= 2) rc = launch_rnn_rows_stream<ORL_HEAD_CATEGORICAL, 2>(A, C2, ip, ic, grid_rp, grid_rc, s);
      else if (no <= 8) rc = launch_rnn_rows_stream<ORL_HEAD_CATEGORICAL, 8>(A, C2, ip, ic, grid_rp, grid_rc, s);
      else rc = launch_rnn_rows_stream<ORL_HEAD_CATEGORICAL, 16>(A, C2, ip, ic, grid_rp, grid_rc, s);
    } else {
      if (no <= 8) rc = launch_rnn_rows_stream<ORL_HEAD_GAUSSIAN, 8>(A, C2, ip, ic, grid_rp, grid_rc, s);
      else rc = launch_rnn_rows_stream<ORL_HEAD_GAUSSIAN, 16>(A, C2, ip, ic, grid_rp, grid_rc, s);
    }
#endif
  }
  if (rc) return rc;
  // (2) weight-gradient GEMMs over both tapes, one launch
  RnnWgArgs Gp, Gc;
  Gp.net = *pnet; Gp.tape = A.tape; Gp.partials = base_p + wp.wpart; Gp.n_blocks = wp.n_blocks;
  Gc.net = *cnet; Gc.tape = C2.tape; Gc.partials = base_c + wc.wpart; Gc.n_blocks = wc.n_blocks;
  const int bp = tape_block_floats(pnet->obs_dim), bc = tape_block_floats(cnet->obs_dim);
  const size_t lds = 3 * (size_t)(bp > bc ? bp : bc) * sizeof(float);
  (void)hipFuncSetAttribute((const void*)rnn_wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(rnn_wgrad_kernel, dim3(wp.grid_wg + wc.grid_wg), dim3(512), lds, s, Gp, Gc, wp.grid_wg);
  rc = launch_status("orl_rnn_ppo_fwd_bwd(wgrad)");
  if (rc) return rc;
  // (3) sums of each tower: [raw without logstd | dlogstd | stats]
  float* sums_c = sums + RnnRaw(*pnet).total + ORL_N_STATS;
#ifdef ORL_RNN_REDUCE_TWO_LAUNCHES  // rounds 1 - 4 (comparison switch)
  rc = orl_ppo_reduce_pair2(Gp.partials, wp.grid_wg, wp.wpw, sums, Gc.partials, wc.grid_wg, wc.wpw, sums_c, s);
  if (rc) return rc;
  return orl_ppo_reduce_pair2(A.partials, grid_rp, wp.rpw, sums + wp.wpw, C2.partials, grid_rc, wc.rpw,
                              sums_c + wc.wpw, s);
#else
  RnnReduce4 R;
  R.p[0] = Gp.partials; R.nb[0] = wp.grid_wg; R.w[0] = wp.wpw; R.o[0] = sums;
  R.p[1] = Gc.partials; R.nb[1] = wc.grid_wg; R.w[1] = wc.wpw; R.o[1] = sums_c;
  R.p[2] = A.partials;  R.nb[2] = grid_rp;    R.w[2] = wp.rpw; R.o[2] = sums + wp.wpw;
  R.p[3] = C2.partials; R.nb[3] = grid_rc;    R.w[3] = wc.rpw; R.o[3] = sums_c + wc.wpw;
  R.g0[0] = 0;
  for (int k = 0; k < 4; ++k) R.g0[k + 1] = R.g0[k] + (R.w[k] + 63) / 64;
  hipLaunchKernelGGL(rnn_reduce4_kernel, dim3(R.g0[4]), dim3(1024), 0, s, R);
  return launch_status("orl_rnn_ppo_fwd_bwd(reduce)");
#endif
}

int orl_rnn_ppo_apply(const orl_net_desc* pnet, const orl_net_desc* cnet, const float* sums, const orl_ppo_hparams* hp,
                      const orl_adam_state* padam, const orl_adam_state* cadam, float* train_info_accum,
                      float* scratch, void* stream) {
  int rc = check_rnn_net(pnet, "orl_rnn_ppo_apply(policy)", false);
  if (rc) return rc;
  rc = check_rnn_net(cnet, "orl_rnn_ppo_apply(critic)", true);
  if (rc) return rc;
  ORL_REQUIRE(sums && hp && padam && cadam && scratch, "orl_rnn_ppo_apply: null pointer");
  ORL_REQUIRE(padam->theta && padam->grad && padam->m && padam->v && cadam->theta && cadam->grad && cadam->m && cadam->v,
              "orl_rnn_ppo_apply: null optimizer buffer");
  ORL_REQUIRE(padam->step >= 1 && cadam->step >= 1, "orl_rnn_ppo_apply: Adam step counts are 1-based");
  RnnApplyTower P, Cc;
  P.net = *pnet; P.ad = *padam; P.raw = sums;
  Cc.net = *cnet; Cc.ad = *cadam; Cc.raw = sums + RnnRaw(*pnet).total + ORL_N_STATS;
  const int nb_p = (RnnLayout(*pnet).total + 255) / 256, nb_c = (RnnLayout(*cnet).total + 255) / 256;
  ORL_REQUIRE(nb_p + 6 <= 256 && nb_c + 6 <= 256, "orl_rnn_ppo_apply: tower too large");
  const int nb = nb_p > nb_c ? nb_p : nb_c;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(rnn_grad_kernel, dim3(nb + 6, 2), dim3(256), 0, s, P, Cc, *hp, scratch, nb, nb_p, nb_c);
  hipLaunchKernelGGL(rnn_adam_kernel, dim3(nb, 2), dim3(256), 0, s, P, Cc, *hp, scratch, nb_p, nb_c,
                     train_info_accum);
  return launch_status("orl_rnn_ppo_apply");
}

}  // extern "C"
