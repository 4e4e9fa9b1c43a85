// orl_ppo.hip - PPO update side of the hot path for gfx950:
//   orl_ppo_fwd_bwd : K9-K12 fused tower forward + PPO/value/entropy loss + full backward, one
//                     launch per tower, producing per-workgroup partial sums of the RAW gradient
//   orl_ppo_reduce  : deterministic column sums of the partials (the vector a multi-GPU run all-reduces)
//   orl_ppo_apply   : raw sums -> parameter gradients, grad-norm clip (K13), Adam (K14), train_info
//
// Work decomposition of orl_ppo_fwd_bwd: a wavefront owns 16-row tiles of the minibatch and walks
// them with a stride of (all waves); per tile it runs, entirely out of registers + its private LDS
// slabs, the chain (fp32 MFMA 16x16x4, T layout - see orl_mlp.h):
//
//   z1 = W1 x + b1 -> relu -> LN1 -> n1 -> z2 = W2 n1 + b2 -> LN2 -> n2 -> head -> loss
//   dhead -> dn2 -> LN2' -> dz2 -> dn1 = W2^T dz2 -> LN1' -> relu' -> dz1
//
// The three 64x64 GEMMs per tile are: z2 (forward), dn1 (dgrad) and G += dz2^T xhat1 (wgrad).  The
// forward/dgrad GEMMs want batch rows across lanes (T layout), the wgrad GEMM and all bias / LN-affine
// gradients reduce OVER batch rows and want features across lanes ("F layout"): the 16x64 slabs
// xhat1, xhat2, dn2, dz2, dn1, dz1 are therefore bounced through LDS once (16-byte row stores in T
// layout, 4-byte column reads in F layout).  Weight-gradient accumulators (G: 64 AGPR/VGPRs) stay in
// registers across ALL tiles of a wave; the affine part of the LayerNorm is folded out of the wgrad
// GEMM algebraically:  dW2 = g1[i] * G[o][i] + be1[i] * db2[o]  with  G = sum_r dz2[r][o] xhat1[r][i].
#include "orl_common.h"
#include "orl_mlp.h"

namespace orl {

constexpr int PPO_WAVES = 8;                 // waves per workgroup (2 per SIMD)
constexpr int PPO_THREADS = PPO_WAVES * 64;
constexpr int PPO_MAX_BLOCKS = 256;          // one workgroup per CU
constexpr int TS = 68;                       // slab row stride (floats): 16-byte rows, bank-skewed
constexpr int SLAB = TILE_B * TS;

struct PpoArgs {
  orl_net_desc net;
  const float* theta;
  const float* records;
  const int64_t* idx;  // may be NULL (identity)
  const float* vn_state;
  float* partials;     // [gridDim][raw.total + ORL_N_STATS]
  orl_ppo_hparams hp;
  int R;               // record width
  int o_x;             // column of this tower's observation inside a record
  int o_act, o_lp, o_adv, o_vp, o_rt, o_am, o_mk;  // record columns
  int a_w;             // stored action width
  int K;               // action-mask width (0 = none)
  int mb;              // rows in this minibatch
};

__device__ inline float huber_or_mse(float e, float d, int use_huber, float& de) {
  if (use_huber) {
    const float ae = fabsf(e);
    if (ae <= d) { de = e; return e * e * 0.5f; }
    de = e > 0.f ? d : -d;
    return d * (ae - d * 0.5f);
  }
  de = e;
  return e * e * 0.5f;
}

// T-layout 16x64 register tile -> LDS slab rows (lane (j,q) writes 4 x 16 bytes of row j)
__device__ inline void store_slab_T(float* __restrict__ slab, const f32x4 (&x)[4], int j, int q) {
#pragma unroll
  for (int m = 0; m < 4; ++m) *(f32x4*)(slab + j * TS + 16 * m + 4 * q) = x[m];
}
__device__ inline void load_slab_T(const float* __restrict__ slab, f32x4 (&x)[4], int j, int q) {
#pragma unroll
  for (int m = 0; m < 4; ++m) x[m] = *(const f32x4*)(slab + j * TS + 16 * m + 4 * q);
}

// make LDS writes of this wave visible to its own later reads (single wave, in-order LDS queue);
// only the compiler must be kept from reordering.
__device__ inline void wave_lds_fence() {
  // LDS operations of one wave execute in issue order, so a later ds_read observes an earlier ds_write of
  // ANY lane of the same wave; the compiler still inserts the lgkmcnt wait before a read's first use.
  asm volatile("" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

// LayerNorm backward in T layout, in place: d <- rstd * (d*g - mean(d*g) - xhat * mean(d*g*xhat))
__device__ inline void ln_bwd_T(f32x4 (&d)[4], const f32x4 (&xhat)[4], const float* __restrict__ g, float rstd,
                                int q) {
  f32x4 t[4];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const f32x4 gg = *(const f32x4*)(g + 16 * m + 4 * q);
    d[m] = d[m] * gg;
    t[m] = d[m] * xhat[m];
    s1 += (d[m][0] + d[m][1]) + (d[m][2] + d[m][3]);
    s2 += (t[m][0] + t[m][1]) + (t[m][2] + t[m][3]);
  }
  s1 = row_allsum(s1);
  s2 = row_allsum(s2);
  const float c1 = s1 * (1.0f / 64.0f), c2 = s2 * (1.0f / 64.0f);
#pragma unroll
  for (int m = 0; m < 4; ++m) d[m] = (d[m] - c1 - xhat[m] * c2) * rstd;
}

// HEAD: ORL_HEAD_VALUE / _CATEGORICAL / _GAUSSIAN; NO: padded head width; ND: ceil(D/16) for the
// MFMA dW1 path, 0 = VALU path for D <= 4.
template <int HEAD, int NO, int ND>
__global__ __launch_bounds__(PPO_THREADS, 2) void ppo_tower_kernel(PpoArgs A) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const TowerLayout tl(A.net);
  const RawLayout rl(A.net);
  const TowerLds tw(A.net.obs_dim, A.net.n_out, HEAD == ORL_HEAD_GAUSSIAN, true);
  stage_tower(smem, A.theta, tl, tw, true, threadIdx.x, blockDim.x);
  const int DP = tw.DP;
  const int D = A.net.obs_dim;
  const int n_out = A.net.n_out;
  constexpr int NOP = (NO + 3) & ~3;
  const int wave = threadIdx.x >> 6, l = threadIdx.x & 63, j = l & 15, q = l >> 4;
  const int rts = (((A.R >> 2) + 3) >> 2) * 256;  // floats per record-ring slot
  const int per_wave = 3 * SLAB + 2 * rts + TILE_B * NOP;
  float* wl = smem + tw.total + wave * per_wave;
  float* X1 = wl;              // xhat1 slab
  float* X2 = wl + SLAB;       // xhat2 slab
  float* SS = wl + 2 * SLAB;   // scratch slab (dn2 -> dz2 -> dn1 -> dz1)
  float* XT = wl + 3 * SLAB;   // record ring: 2 slots of [chunk][16 rows][4 floats]
  float* DH = XT + 2 * rts;    // dhead [16][NOP]
  __syncthreads();

  const float* lw = smem;
  const orl_ppo_hparams hp = A.hp;

  // ---- persistent accumulators -------------------------------------------------------------------
  f32x4 G[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) G[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr int NDA = ND > 0 ? ND : 1;
  f32x4 G1[4][NDA];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < NDA; ++b) G1[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  float w1v[4] = {0.f, 0.f, 0.f, 0.f};  // VALU dW1 (ND == 0): lane f, k = 0..3
  float a_dg2 = 0.f, a_dbe2 = 0.f, a_db2 = 0.f, a_dg1 = 0.f, a_dbe1 = 0.f, a_db1 = 0.f, a_db3 = 0.f;
  float a_S3[NO];
#pragma unroll
  for (int c = 0; c < NO; ++c) a_S3[c] = 0.f;
  float a_dls = 0.f;  // Gaussian: dlogstd for c == lane (F pass)
  float st_active = 0.f, st_rows = 0.f, st_loss = 0.f, st_ent = 0.f, st_ratio = 0.f;

  // ValueNorm.normalize coefficients (valuenorm.py:79-91)
  float vn_mean = 0.f, vn_sd = 1.f;
  if (HEAD == ORL_HEAD_VALUE && hp.use_valuenorm && A.vn_state != nullptr) {
    const float deb = fmaxf(A.vn_state[2], 1e-5f);
    vn_mean = A.vn_state[0] / deb;
    const float msq = A.vn_state[1] / deb;
    vn_sd = sqrtf(fmaxf(msq - vn_mean * vn_mean, 1e-2f));
  }

  const int n_tiles = (A.mb + TILE_B - 1) / TILE_B;
  const int nwv = blockDim.x >> 6;  // waves in this workgroup (8 unless LDS forces fewer)
  const int wave_g = blockIdx.x * nwv + wave;
  const int n_waves = gridDim.x * nwv;

  // ---- record tile pipeline -----------------------------------------------------------------------
  // The 16 records of a tile are DMA'd global -> LDS (global_load_lds, 16 B per lane, no VGPRs) one tile
  // AHEAD into a 2-deep ring; layout RT[chunk][row][4 floats] (the DMA writes wave-base + lane*16 B with
  // lane = row + 16*q handling chunk 4g+q).  Minibatch indices are fetched two tiles ahead.
  const int nch = A.R >> 2;
  const int RTS = ((nch + 3) >> 2) * 256;  // floats per ring slot
  auto row_of = [&](int t) -> long long {
    const int ii = t * TILE_B + j;
    if (t >= n_tiles || ii >= A.mb) return 0;  // invalid lanes read row 0 (finite data, weight 0)
    return (A.idx != nullptr) ? A.idx[ii] : (long long)ii;
  };
  auto issue_dma = [&](float* slot, long long row) {
    const float* src = A.records + (size_t)row * A.R;
    for (int g = 0; 4 * g < nch; ++g) {
      const int c = 4 * g + q;
      if (c < nch)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 4 * c),
                                         (__attribute__((address_space(3))) void*)(slot + g * 256), 16, 0, 0);
    }
  };
  long long row_next = row_of(wave_g);
  issue_dma(XT, row_next);
  row_next = row_of(wave_g + n_waves);
  int ring = 0;

  for (int tile = wave_g; tile < n_tiles; tile += n_waves) {
    const int i = tile * TILE_B + j;
    const bool valid = i < A.mb;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this tile's records have landed in LDS
    const float* RT = XT + ring * RTS;
    issue_dma(XT + (ring ^ 1) * RTS, row_next);        // next tile's records, hidden behind this tile
    row_next = row_of(tile + 2 * n_waves);
    ring ^= 1;
#define REC(col) RT[(((col) >> 2) << 6) + (j << 2) + ((col) & 3)]
#define REC_R(r, col) RT[(((col) >> 2) << 6) + ((r) << 2) + ((col) & 3)]

    // ---------------- forward ----------------
    f32x4 z[4], xh1[4], n1[4], xh2[4];
    float rstd1, rstd2;
    load_vec_T(lw + tw.b1, q, z);
    // columns >= D of a record are other (finite) fields; W1's LDS image is zero-padded there
    fc1_T(lw + tw.W1, DP, [&](int s) -> float { return REC(A.o_x + 4 * s + q); }, z, j, q);
    unsigned relu_bits = 0u;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (z[m][r] > 0.f) relu_bits |= 1u << (4 * m + r);
        xh1[m][r] = fmaxf(z[m][r], 0.f);
      }
    ln_normalize_T(xh1, rstd1);
    store_slab_T(X1, xh1, j, q);
    ln_affine_T(xh1, lw + tw.g1, lw + tw.be1, q, n1);
    load_vec_T(lw + tw.b2, q, xh2);
    mm64_T(lw + tw.W2, n1, xh2, j, q);
    ln_normalize_T(xh2, rstd2);
    store_slab_T(X2, xh2, j, q);
    ln_affine_T(xh2, lw + tw.g2, lw + tw.be2, q, z);  // z = n2
    float hd[NO];
    head_T<NO>(lw + tw.W3, lw + tw.b3, n_out, z, q, hd);

    // ---------------- loss + dhead (per batch row; the 4 lanes of a row compute identically) --------
    float dh[NO];
#pragma unroll
    for (int c = 0; c < NO; ++c) dh[c] = 0.f;
    const float active = valid ? REC(A.o_am) : 0.f;
    if (HEAD == ORL_HEAD_VALUE) {
      const float w = valid ? (hp.use_value_active_masks ? active : 1.f) : 0.f;
      const float v = hd[0];
      const float vp = valid ? REC(A.o_vp) : 0.f;
      float rt = valid ? REC(A.o_rt) : 0.f;
      if (hp.use_valuenorm) rt = (rt - vn_mean) / vn_sd;
      // cal_value_loss (ppo.py:178-220)
      const float dv = v - vp;
      const float dvc = fminf(fmaxf(dv, -hp.clip_param), hp.clip_param);
      const bool inside = (dv >= -hp.clip_param) && (dv <= hp.clip_param);
      const float e_c = rt - (vp + dvc);
      const float e_o = rt - v;
      float de_c, de_o;
      const float l_c = huber_or_mse(e_c, hp.huber_delta, hp.use_huber_loss, de_c);
      const float l_o = huber_or_mse(e_o, hp.huber_delta, hp.use_huber_loss, de_o);
      float vl, g;  // g = d vl / d v
      if (hp.use_clipped_value_loss) {
        vl = fmaxf(l_o, l_c);
        if (l_o > l_c) g = -de_o;
        else if (l_o < l_c) g = inside ? -de_c : 0.f;
        else g = -0.5f * de_o + (inside ? -0.5f * de_c : 0.f);  // torch.max splits ties
      } else {
        vl = l_o;
        g = -de_o;
      }
      dh[0] = w * g * hp.value_loss_coef;
      if (q == 0 && valid) { st_active += active; st_rows += 1.f; st_loss += vl * w; }
    } else {
      const float w = valid ? (hp.use_policy_active_masks ? active : 1.f) : 0.f;
      const float adv = valid ? REC(A.o_adv) : 0.f;
      if (HEAD == ORL_HEAD_CATEGORICAL) {
        float mk[NO];  // action mask of this row (1 = legal)
#pragma unroll
        for (int c = 0; c < NO; ++c) mk[c] = (A.K > 0 && valid && c < n_out) ? REC(A.o_mk + c) : 1.f;
        const float lse = cat_lse<NO>(hd, n_out, mk);
        const int act = valid ? (int)REC(A.o_act) : 0;
        const float old_lp = valid ? REC(A.o_lp) : 0.f;
        const float lp = pick<NO>(hd, act) - lse;
        float ent = 0.f;
        float p[NO];
#pragma unroll
        for (int c = 0; c < NO; ++c) {
          p[c] = 0.f;
          if (c < n_out) {
            const float ell = hd[c] - lse;
            p[c] = expf(ell);
            ent -= p[c] * ell;
          }
        }
        float ratio = expf(lp - old_lp);
        const float ratio_raw = ratio;
        float dr_eff = 1.f;
        if (hp.dual_clip_ppo) {
          if (ratio > hp.dual_clip_coeff) { ratio = hp.dual_clip_coeff; dr_eff = 0.f; }
        }
        const float s1 = ratio * adv;
        const float s2 = fminf(fmaxf(ratio, 1.f - hp.clip_param), 1.f + hp.clip_param) * adv;
        const float surr = fminf(s1, s2);
        const float dsurr_dr = (s1 <= s2) ? adv : 0.f;
        const float gl = -dsurr_dr * dr_eff * ratio_raw;  // d(-surr)/d logp
#pragma unroll
        for (int c = 0; c < NO; ++c) {
          if (c < n_out) {
            const bool masked = mk[c] == 0.f;
            const float ell = hd[c] - lse;
            const float d = gl * ((c == act ? 1.f : 0.f) - p[c]) + hp.entropy_coef * p[c] * (ell + ent);
            dh[c] = masked ? 0.f : w * d;
          }
        }
        if (q == 0 && valid) {
          st_active += active; st_rows += 1.f; st_loss += -surr * w; st_ent += ent * w; st_ratio += ratio;
        }
      } else {
        // DiagGaussian, everything per action dimension (distributions.py:34-43, ppo.py:302-317)
        const float ent_scale = hp.use_policy_active_masks ? 1.f : 1.f / (float)n_out;
        float surr_sum = 0.f, ent_sum = 0.f, ratio_sum = 0.f;
#pragma unroll
        for (int c = 0; c < NO; ++c) {
          if (c < n_out) {
            const float ls = lw[tw.logstd + c];
            const float sd = expf(ls);
            const float av = valid ? REC(A.o_act + c) : 0.f;
            const float old_lp = valid ? REC(A.o_lp + c) : 0.f;
            const float dmu = av - hd[c];
            const float var = sd * sd;
            const float lp = -(dmu * dmu) / (2.f * var) - ls - 0.91893853320467274178f;
            float ratio = expf(lp - old_lp);
            const float ratio_raw = ratio;
            float dr_eff = 1.f;
            if (hp.dual_clip_ppo && ratio > hp.dual_clip_coeff) { ratio = hp.dual_clip_coeff; dr_eff = 0.f; }
            const float s1 = ratio * adv;
            const float s2 = fminf(fmaxf(ratio, 1.f - hp.clip_param), 1.f + hp.clip_param) * adv;
            surr_sum += fminf(s1, s2);
            const float gl = -((s1 <= s2) ? adv : 0.f) * dr_eff * ratio_raw;
            dh[c] = w * gl * dmu / var;
            // dlogstd row contribution, parked in DH's upper half via the F pass below
            const float dls = w * (gl * (dmu * dmu / var - 1.f) - hp.entropy_coef * ent_scale);
            if (q == 1) DH[j * NOP + c] = dls;  // temporarily; consumed before dhead is written
            ent_sum += 1.41893853320467274178f + ls;
            ratio_sum += ratio;
          }
        }
        if (q == 0 && valid) {
          st_active += active; st_rows += 1.f; st_loss += -surr_sum * w; st_ent += ent_sum * w;
          st_ratio += ratio_sum;
        }
        // fold the dlogstd column sums now (lane c < n_out sums DH[:, c] over the 16 rows)
        wave_lds_fence();
        if (l < n_out) {
          float s = 0.f;
          for (int r = 0; r < TILE_B; ++r) s += DH[r * NOP + l];
          a_dls += s;
        }
        wave_lds_fence();
      }
    }

    // ---------------- backward ----------------
    // dn2 = W3^T dhead   (T layout)
    f32x4 d2[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) d2[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < NO; ++c) {
      if (c < n_out) {
#pragma unroll
        for (int m = 0; m < 4; ++m) d2[m] += *(const f32x4*)(lw + tw.W3 + c * HID + 16 * m + 4 * q) * dh[c];
      }
    }
    store_slab_T(SS, d2, j, q);
    if (q == 0) {
#pragma unroll
      for (int c = 0; c < NO; ++c) DH[j * NOP + c] = dh[c];
    }
    wave_lds_fence();
    // F pass A (lane = feature f): dg2, dbe2, S3, db3
    {
      const int f = l;
      float s_dg = 0.f, s_dbe = 0.f, s_db3 = 0.f;
      float s3[NO];
#pragma unroll
      for (int c = 0; c < NO; ++c) s3[c] = 0.f;
      for (int r = 0; r < TILE_B; ++r) {
        const float dn = SS[r * TS + f];
        const float xh = X2[r * TS + f];
        s_dg += dn * xh;
        s_dbe += dn;
#pragma unroll
        for (int c = 0; c < NO; ++c) s3[c] += DH[r * NOP + c] * xh;
        if (f < NO) s_db3 += DH[r * NOP + (f < NO ? f : 0)];
      }
      a_dg2 += s_dg; a_dbe2 += s_dbe; a_db3 += s_db3;
#pragma unroll
      for (int c = 0; c < NO; ++c) a_S3[c] += s3[c];
    }
    // LN2 backward -> dz2
    ln_bwd_T(d2, xh2, lw + tw.g2, rstd2, q);
    wave_lds_fence();
    store_slab_T(SS, d2, j, q);
    wave_lds_fence();
    // F pass B: G += dz2^T xhat1 (MFMA, operands straight from the slabs), db2
    {
      float s_db = 0.f;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        float av[4], bv[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          av[m] = SS[(4 * s + q) * TS + 16 * m + j];
          bv[m] = X1[(4 * s + q) * TS + 16 * m + j];
        }
#pragma unroll
        for (int mo = 0; mo < 4; ++mo)
#pragma unroll
          for (int mi = 0; mi < 4; ++mi) G[mo][mi] = ORL_MFMA(av[mo], bv[mi], G[mo][mi]);
      }
      const int f = l;
      for (int r = 0; r < TILE_B; ++r) s_db += SS[r * TS + f];
      a_db2 += s_db;
    }
    // dn1 = W2^T dz2
    f32x4 d1[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) d1[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    mm64_T(lw + tw.W2T, d2, d1, j, q);
    wave_lds_fence();
    store_slab_T(SS, d1, j, q);
    wave_lds_fence();
    // F pass C: dg1, dbe1
    {
      const int f = l;
      float s_dg = 0.f, s_dbe = 0.f;
      for (int r = 0; r < TILE_B; ++r) {
        const float dn = SS[r * TS + f];
        s_dg += dn * X1[r * TS + f];
        s_dbe += dn;
      }
      a_dg1 += s_dg; a_dbe1 += s_dbe;
    }
    // LN1 backward, relu backward -> dz1
    load_slab_T(X1, xh1, j, q);
    ln_bwd_T(d1, xh1, lw + tw.g1, rstd1, q);
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (!((relu_bits >> (4 * m + r)) & 1u)) d1[m][r] = 0.f;
    wave_lds_fence();
    store_slab_T(SS, d1, j, q);
    wave_lds_fence();
    // F pass D: dW1, db1
    {
      const int f = l;
      float s_db = 0.f;
      for (int r = 0; r < TILE_B; ++r) s_db += SS[r * TS + f];
      a_db1 += s_db;
      if (ND == 0) {
        for (int r = 0; r < TILE_B; ++r) {
          const float dzv = SS[r * TS + f];
          const f32x4 xv = *(const f32x4*)(&REC_R(r, A.o_x));  // o_x % 4 == 0 on this path
          w1v[0] += dzv * xv[0]; w1v[1] += dzv * xv[1]; w1v[2] += dzv * xv[2]; w1v[3] += dzv * xv[3];
        }
      } else {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          float av[4], bv[NDA];
#pragma unroll
          for (int m = 0; m < 4; ++m) av[m] = SS[(4 * s + q) * TS + 16 * m + j];
#pragma unroll
          for (int mk = 0; mk < NDA; ++mk) bv[mk] = (16 * mk + j < D) ? REC_R(4 * s + q, A.o_x + 16 * mk + j) : 0.f;
#pragma unroll
          for (int mf = 0; mf < 4; ++mf)
#pragma unroll
            for (int mk = 0; mk < NDA; ++mk) G1[mf][mk] = ORL_MFMA(av[mf], bv[mk], G1[mf][mk]);
        }
      }
    }
    wave_lds_fence();
  }

#undef REC
#undef REC_R
  // ---- workgroup reduction of the 8 waves' accumulators, deterministic order -----------------------
  __syncthreads();
  float* acc = smem + tw.total;  // reuse the slab area: [rl.total + ORL_N_STATS]
  const int PW = rl.total + ORL_N_STATS;
  for (int e = threadIdx.x; e < PW; e += blockDim.x) acc[e] = 0.f;
  __syncthreads();
  st_active = wave_sum(st_active); st_rows = wave_sum(st_rows); st_loss = wave_sum(st_loss);
  st_ent = wave_sum(st_ent); st_ratio = wave_sum(st_ratio);
  for (int w = 0; w < nwv; ++w) {
    if (wave == w) {
      // G tiles: lane (c = j, q), reg r -> G[o = 16mo+4q+r][i = 16mi+c]
#pragma unroll
      for (int mo = 0; mo < 4; ++mo)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[rl.oG + (16 * mo + 4 * q + r) * HID + 16 * mi + j] += G[mo][mi][r];
      const int f = l;
      acc[rl.odg2 + f] += a_dg2; acc[rl.odbe2 + f] += a_dbe2; acc[rl.odb2 + f] += a_db2;
      acc[rl.odg1 + f] += a_dg1; acc[rl.odbe1 + f] += a_dbe1; acc[rl.odb1 + f] += a_db1;
#pragma unroll
      for (int c = 0; c < NO; ++c)
        if (c < n_out) acc[rl.oS3 + c * HID + f] += a_S3[c];
      if (f < n_out) acc[rl.odb3 + f] += a_db3;
      if (HEAD == ORL_HEAD_GAUSSIAN && f < n_out) acc[rl.odlogstd + f] += a_dls;
      if (ND == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (k < D) acc[rl.odW1 + f * D + k] += w1v[k];
      } else {
#pragma unroll
        for (int mf = 0; mf < 4; ++mf)
#pragma unroll
          for (int mk = 0; mk < NDA; ++mk)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int kk = 16 * mk + j;
              if (kk < D) acc[rl.odW1 + (16 * mf + 4 * q + r) * D + kk] += G1[mf][mk][r];
            }
      }
      if (l == 0) {
        acc[rl.total + ST_ACTIVE_SUM] += st_active;
        acc[rl.total + ST_ROWS] += st_rows;
        if (HEAD == ORL_HEAD_VALUE) acc[rl.total + ST_VLOSS_SUM] += st_loss;
        else {
          acc[rl.total + ST_PLOSS_SUM] += st_loss;
          acc[rl.total + ST_ENT_SUM] += st_ent;
          acc[rl.total + ST_RATIO_SUM] += st_ratio;
        }
      }
    }
    __syncthreads();
  }
  float* out = A.partials + (size_t)blockIdx.x * PW;
  for (int e = threadIdx.x; e < PW; e += blockDim.x) out[e] = acc[e];
}

// ---- reduce: column sums over workgroup partials -------------------------------------------------
__global__ __launch_bounds__(256) void ppo_reduce_kernel(const float* __restrict__ partials, int n_blocks, int width,
                                                         float* __restrict__ sums) {
  // block = 64 columns x 4 row groups: coalesced 256-byte row segments, 4-way split of the row walk,
  // fixed summation order (deterministic): rows rg, rg+4, ... then groups 0..3.
  __shared__ float sh[4][64];
  const int col = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rg = threadIdx.x >> 6;
  float s0 = 0.f, s1 = 0.f;
  if (col < width) {
    int b = rg;
    for (; b + 4 < n_blocks; b += 8) {
      s0 += partials[(size_t)b * width + col];
      s1 += partials[(size_t)(b + 4) * width + col];
    }
    if (b < n_blocks) s0 += partials[(size_t)b * width + col];
  }
  sh[rg][threadIdx.x & 63] = s0 + s1;
  __syncthreads();
  if (rg == 0 && col < width) sums[col] = (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
}

// ---- apply: raw sums -> grads -> clip -> Adam; one workgroup of 1024 threads ------------------------
struct ApplyTower {
  orl_net_desc net;
  orl_adam_state ad;
  int sums_off;  // offset of this tower's raw vector in `sums`
};

__device__ inline float block_sum_1024(float v, float* sh) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int k = 0; k < 16; ++k) t += sh[k];
  __syncthreads();
  return t;
}

// gradient of parameter p (parameter order) from the raw sums of one tower, already divided by den
__device__ inline float raw_to_grad(const float* __restrict__ raw, const float* __restrict__ theta,
                                    const TowerLayout& tl, const RawLayout& rl, int p, float inv_den) {
  const int H = HID, D = tl.D;
  float g;
  if (p < tl.ob1) g = raw[rl.odW1 + (p - tl.oW1)];
  else if (p < tl.og1) g = raw[rl.odb1 + (p - tl.ob1)];
  else if (p < tl.obe1) g = raw[rl.odg1 + (p - tl.og1)];
  else if (p < tl.oW2) g = raw[rl.odbe1 + (p - tl.obe1)];
  else if (p < tl.ob2) {
    const int e = p - tl.oW2, o = e / H, i = e - o * H;
    g = theta[tl.og1 + i] * raw[rl.oG + e] + theta[tl.obe1 + i] * raw[rl.odb2 + o];
  } else if (p < tl.og2) g = raw[rl.odb2 + (p - tl.ob2)];
  else if (p < tl.obe2) g = raw[rl.odg2 + (p - tl.og2)];
  else if (p < tl.oW3) g = raw[rl.odbe2 + (p - tl.obe2)];
  else if (p < tl.ob3) {
    const int e = p - tl.oW3, c = e / H, f = e - c * H;
    g = theta[tl.og2 + f] * raw[rl.oS3 + e] + theta[tl.obe2 + f] * raw[rl.odb3 + c];
  } else if (p < tl.ologstd) g = raw[rl.odb3 + (p - tl.ob3)];
  else g = raw[rl.odlogstd + (p - tl.ologstd)];
  (void)D;
  return g * inv_den;
}

__global__ __launch_bounds__(1024) void ppo_apply_kernel(ApplyTower P, ApplyTower Cc, const float* __restrict__ sums,
                                                         orl_ppo_hparams hp, float* __restrict__ info) {
  __shared__ float sh[16];
  const TowerLayout tlp(P.net), tlc(Cc.net);
  const RawLayout rlp(P.net), rlc(Cc.net);
  const float* rawp = sums + P.sums_off;
  const float* stp = rawp + rlp.total;
  const float* rawc = sums + Cc.sums_off;
  const float* stc = rawc + rlc.total;
  const float den_p = hp.use_policy_active_masks ? stp[ST_ACTIVE_SUM] : stp[ST_ROWS];
  const float den_v = hp.use_value_active_masks ? stc[ST_ACTIVE_SUM] : stc[ST_ROWS];
  float norms[2];
  for (int t = 0; t < 2; ++t) {
    const ApplyTower& W = t == 0 ? P : Cc;
    const TowerLayout& tl = t == 0 ? tlp : tlc;
    const RawLayout& rl = t == 0 ? rlp : rlc;
    const float* raw = t == 0 ? rawp : rawc;
    const float inv_den = 1.0f / (t == 0 ? den_p : den_v);
    if (t == 0 && (hp.reserved & 1)) {
      // turn_on == False: the policy loss is not in the loss list (ppo.py:226-236) -> no gradient, no step
      for (int p = threadIdx.x; p < tl.total; p += blockDim.x) W.ad.grad[p] = 0.f;
      norms[0] = 0.f;
      continue;
    }
    float ss = 0.f;
    for (int p = threadIdx.x; p < tl.total; p += blockDim.x) {
      float g = raw_to_grad(raw, W.ad.theta, tl, rl, p, inv_den);
      W.ad.grad[p] = g;
      ss += g * g;
    }
    const float total = sqrtf(block_sum_1024(ss, sh));
    norms[t] = total;
    // torch.nn.utils.clip_grad_norm_: coef = max_norm / (norm + 1e-6), clamped to 1
    float coef = 1.f;
    if (hp.use_max_grad_norm) coef = fminf(hp.max_grad_norm / (total + 1e-6f), 1.f);
    // torch.optim.Adam (single tensor math, betas (0.9, 0.999), amsgrad off)
    // scalar coefficients are python doubles in torch/optim/adam.py; only tensor math is fp32
    const double b1d = 0.9, b2d = 0.999;
    const float b2 = (float)b2d, omb1 = (float)(1.0 - b1d), omb2 = (float)(1.0 - b2d);
    const double bc1 = 1.0 - pow(b1d, (double)W.ad.step);
    const double bc2 = 1.0 - pow(b2d, (double)W.ad.step);
    const float step_size = (float)((double)W.ad.lr / bc1);
    const float bc2_sqrt = (float)sqrt(bc2);
    for (int p = threadIdx.x; p < tl.total; p += blockDim.x) {
      float g = W.ad.grad[p] * coef;
      W.ad.grad[p] = g;
      float th = W.ad.theta[p];
      if (W.ad.weight_decay != 0.f) g += W.ad.weight_decay * th;
      float m = W.ad.m[p], v = W.ad.v[p];
      m = m + (g - m) * omb1;          // exp_avg.lerp_(grad, 1 - beta1)
      v = v * b2 + omb2 * (g * g);     // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
      const float denom = sqrtf(v) / bc2_sqrt + W.ad.eps;
      th = th - step_size * (m / denom);
      W.ad.m[p] = m; W.ad.v[p] = v; W.ad.theta[p] = th;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0 && info != nullptr) {
    const float ploss = stp[ST_PLOSS_SUM] / den_p;
    float ent_den = den_p;
    if (!hp.use_policy_active_masks && P.net.head_kind == ORL_HEAD_GAUSSIAN) ent_den = den_p * (float)P.net.n_out;
    info[0] += stc[ST_VLOSS_SUM] / den_v;                 // value_loss
    info[1] += ploss;                                     // policy_loss
    info[2] += stp[ST_ENT_SUM] / ent_den;                 // dist_entropy
    info[3] += norms[0];                                  // actor_grad_norm
    info[4] += norms[1];                                  // critic_grad_norm
    const float a_w = P.net.head_kind == ORL_HEAD_GAUSSIAN ? (float)P.net.n_out : 1.f;
    info[5] += stp[ST_RATIO_SUM] / (stp[ST_ROWS] * a_w);  // ratio.mean()
  }
}

static int check_tower(const orl_net_desc* n, const char* who) {
  if (!n) return fail(ORL_E_INVALID, "%s: null net descriptor", who);
  if (n->hidden != HID) return fail(ORL_E_UNSUPPORTED, "%s: hidden_size %d not built (only 64)", who, n->hidden);
  if (n->obs_dim < 1 || n->obs_dim > 64) return fail(ORL_E_UNSUPPORTED, "%s: obs_dim %d outside [1,64]", who, n->obs_dim);
  if (n->n_out < 1 || n->n_out > 16) return fail(ORL_E_UNSUPPORTED, "%s: n_out %d outside [1,16]", who, n->n_out);
  return 0;
}

template <int HEAD, int NO, int ND>
static int launch_tower(const PpoArgs& A, int grid, hipStream_t s) {
  const TowerLds tw(A.net.obs_dim, A.net.n_out, HEAD == ORL_HEAD_GAUSSIAN, true);
  constexpr int NOP = (NO + 3) & ~3;
  const int rts = (((A.R >> 2) + 3) >> 2) * 256;
  const int per_wave = 3 * SLAB + 2 * rts + TILE_B * NOP;
  const RawLayout rl(A.net);
  // as many waves per workgroup (8, 6, 4, 2) as fit the 160 KiB of LDS next to the tower's weights
  const size_t need_acc = (size_t)tw.total + rl.total + ORL_N_STATS;
  int waves = PPO_WAVES;
  size_t fl = 0;
  for (; waves >= 2; waves -= 2) {
    fl = (size_t)tw.total + (size_t)waves * per_wave;
    if (need_acc > fl) fl = need_acc;
    if (fl * sizeof(float) <= 160 * 1024) break;
  }
  const size_t lds = fl * sizeof(float);
  if (waves < 2) return fail(ORL_E_UNSUPPORTED, "orl_ppo_fwd_bwd: tower needs %zu B of LDS (> 160 KiB)", lds);
  (void)hipFuncSetAttribute((const void*)ppo_tower_kernel<HEAD, NO, ND>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds);
  hipLaunchKernelGGL((ppo_tower_kernel<HEAD, NO, ND>), dim3(grid), dim3(waves * 64), lds, s, A);
  return launch_status("orl_ppo_fwd_bwd");
}

template <int HEAD, int NO>
static int launch_tower_nd(const PpoArgs& A, int grid, hipStream_t s) {
  const int D = A.net.obs_dim;
  if (D <= 4 && (A.o_x & 3) == 0) return launch_tower<HEAD, NO, 0>(A, grid, s);
  if (D <= 16) return launch_tower<HEAD, NO, 1>(A, grid, s);
  if (D <= 32) return launch_tower<HEAD, NO, 2>(A, grid, s);
  return launch_tower<HEAD, NO, 4>(A, grid, s);
}

}  // namespace orl

using namespace orl;

extern "C" {

int orl_ppo_max_blocks(void) { return PPO_MAX_BLOCKS; }

int orl_ppo_fwd_bwd(const orl_net_desc* pnet, const float* ptheta, const orl_net_desc* cnet, const float* ctheta,
                    const float* records, int rec_width, const int64_t* idx, int mb, const float* vn_state,
                    const orl_ppo_hparams* hp, float* partials, int* n_blocks_out, void* stream) {
  int rc = check_tower(pnet, "orl_ppo_fwd_bwd(policy)");
  if (rc) return rc;
  rc = check_tower(cnet, "orl_ppo_fwd_bwd(critic)");
  if (rc) return rc;
  ORL_REQUIRE(ptheta && ctheta && records && hp && partials, "orl_ppo_fwd_bwd: null pointer");
  ORL_REQUIRE(mb > 0, "orl_ppo_fwd_bwd: empty minibatch");
  ORL_REQUIRE(cnet->head_kind == ORL_HEAD_VALUE && cnet->n_out == 1, "orl_ppo_fwd_bwd: critic must be a value head");
  ORL_REQUIRE(pnet->head_kind == ORL_HEAD_CATEGORICAL || pnet->head_kind == ORL_HEAD_GAUSSIAN,
              "orl_ppo_fwd_bwd: policy head kind %d not built", pnet->head_kind);
  const int a_w = pnet->head_kind == ORL_HEAD_CATEGORICAL ? 1 : pnet->n_out;
  const int Dp = pnet->obs_dim, Dc = cnet->obs_dim;
  // record columns (orl_adv_normalize_pack): [pobs | cobs | act | logp | adv | vpred | ret | active | amask]
  const int o_co = Dp, o_ac = o_co + Dc, o_lp = o_ac + a_w, o_adv = o_lp + a_w;
  PpoArgs A;
  A.records = records; A.idx = idx; A.vn_state = vn_state; A.hp = *hp; A.R = rec_width;
  A.o_act = o_ac; A.o_lp = o_lp; A.o_adv = o_adv; A.o_vp = o_adv + 1; A.o_rt = o_adv + 2; A.o_am = o_adv + 3;
  A.o_mk = o_adv + 4; A.a_w = a_w; A.mb = mb;
  // action-mask width: categorical records always carry n_out mask floats (ReplayData keeps ones)
  A.K = pnet->head_kind == ORL_HEAD_CATEGORICAL ? pnet->n_out : 0;
  ORL_REQUIRE(orl_record_width(Dp, Dc, a_w, A.K) == rec_width, "orl_ppo_fwd_bwd: record width %d != %d", rec_width,
              orl_record_width(Dp, Dc, a_w, A.K));
  const int n_tiles = (mb + TILE_B - 1) / TILE_B;
  int grid = (n_tiles + PPO_WAVES - 1) / PPO_WAVES;
  if (grid > PPO_MAX_BLOCKS) grid = PPO_MAX_BLOCKS;
  hipStream_t s = (hipStream_t)stream;
  const RawLayout rlp(*pnet);

  // policy tower
  A.net = *pnet; A.theta = ptheta; A.o_x = 0; A.partials = partials;
  const int no = pnet->n_out;
  if (pnet->head_kind == ORL_HEAD_CATEGORICAL) {
    if (no <= 2) rc = launch_tower_nd<ORL_HEAD_CATEGORICAL, 2>(A, grid, s);
    else if (no <= 8) rc = launch_tower_nd<ORL_HEAD_CATEGORICAL, 8>(A, grid, s);
    else rc = launch_tower_nd<ORL_HEAD_CATEGORICAL, 16>(A, grid, s);
  } else {
    if (no <= 8) rc = launch_tower_nd<ORL_HEAD_GAUSSIAN, 8>(A, grid, s);
    else rc = launch_tower_nd<ORL_HEAD_GAUSSIAN, 16>(A, grid, s);
  }
  if (rc) return rc;
  // critic tower
  A.net = *cnet; A.theta = ctheta; A.o_x = o_co;
  A.partials = partials + (size_t)PPO_MAX_BLOCKS * (rlp.total + ORL_N_STATS);
  rc = launch_tower_nd<ORL_HEAD_VALUE, 1>(A, grid, s);
  if (rc) return rc;
  if (n_blocks_out) *n_blocks_out = grid;
  return 0;
}

int orl_ppo_reduce(const float* partials, int n_blocks, int width, float* sums, void* stream) {
  ORL_REQUIRE(partials && sums && n_blocks > 0 && width > 0, "orl_ppo_reduce: bad arguments");
  const int grid = (width + 63) / 64;
  hipLaunchKernelGGL(ppo_reduce_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, partials, n_blocks, width, sums);
  return launch_status("orl_ppo_reduce");
}

int orl_ppo_apply(const orl_net_desc* pnet, const orl_net_desc* cnet, const float* sums, const orl_ppo_hparams* hp,
                  const orl_adam_state* padam, const orl_adam_state* cadam, float* train_info_accum, void* stream) {
  int rc = check_tower(pnet, "orl_ppo_apply(policy)");
  if (rc) return rc;
  rc = check_tower(cnet, "orl_ppo_apply(critic)");
  if (rc) return rc;
  ORL_REQUIRE(sums && hp && padam && cadam, "orl_ppo_apply: null pointer");
  ORL_REQUIRE(padam->theta && padam->grad && padam->m && padam->v && cadam->theta && cadam->grad && cadam->m && cadam->v,
              "orl_ppo_apply: null optimizer buffer");
  ORL_REQUIRE(padam->step >= 1 && cadam->step >= 1, "orl_ppo_apply: Adam step counts are 1-based");
  ApplyTower P, Cc;
  P.net = *pnet; P.ad = *padam; P.sums_off = 0;
  Cc.net = *cnet; Cc.ad = *cadam; Cc.sums_off = RawLayout(*pnet).total + ORL_N_STATS;
  hipLaunchKernelGGL(ppo_apply_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, P, Cc, sums, *hp, train_info_accum);
  return launch_status("orl_ppo_apply");
}

}  // extern "C"
