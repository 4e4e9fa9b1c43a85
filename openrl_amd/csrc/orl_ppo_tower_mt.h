// orl_ppo_tower_mt.h - multi-tile variant of the fused PPO tower kernel (orl_ppo_tower.h): NT independent 16-row
// tiles per wavefront, ONE wavefront per SIMD (4 per workgroup, 512 registers each).
//
// Why: the single-tile kernel is latency-bound - MFMA pipe 46 % busy, VALU 33 %, LDS 34 %, waves 35 % in s_waitcnt
// (tools/pmc_tower.sh) - and its 256 VGPRs (64 wgrad accumulators + activations) keep a third wave per SIMD out.
// Here the parallelism is instruction-level instead: every phase of the tile pipeline is written as a loop over the
// NT tiles with no fence inside, so hipcc interleaves NT independent dependency chains in one instruction stream;
// the 64 accumulators and every weight operand read from LDS are shared by the NT tiles.  Same arithmetic, same
// per-workgroup partial rows, same deterministic reduction as the single-tile kernel.
#pragma once
#include "orl_common.h"
#include "orl_loss.h"
#include "orl_mlp.h"
#include "orl_ppo_tower.h"

namespace orl {

template <int HEAD, int NO, int ND, int NT>
__global__ __launch_bounds__(256, 1) void ppo_tower_mt_kernel(PpoArgs A) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const TowerLayout tl(A.net);
  const RawLayout rl(A.net);
  const TowerLds tw(ND == 0 ? 4 : A.net.obs_dim, NO <= 4 ? 4 : A.net.n_out, HEAD == ORL_HEAD_GAUSSIAN, false);
  stage_tower(smem, A.theta, tl, tw, false, threadIdx.x, blockDim.x);
  const int DP = tw.DP;
  const int D = A.net.obs_dim;
  const int n_out = A.net.n_out;
  constexpr int NOP = (NO + 3) & ~3;
  const int wave = threadIdx.x >> 6, l = threadIdx.x & 63, j = l & 15, q = l >> 4;
  const int nch = A.R >> 2;
  const int rts = ((nch + 3) >> 2) * 256;
  const int per_tile = 2 * SLAB + 2 * rts + TILE_B * NOP;
  float* wl = smem + tw.total + wave * NT * per_tile;
  __syncthreads();
  const float* lw = smem;
  const orl_ppo_hparams hp = A.hp;
  LossCols cols;
  cols.o_act = A.o_act; cols.o_lp = A.o_lp; cols.o_adv = A.o_adv; cols.o_vp = A.o_vp; cols.o_rt = A.o_rt;
  cols.o_am = A.o_am; cols.o_mk = A.o_mk; cols.K = A.K;

  // ---- persistent accumulators, shared by the NT tiles ----------------------------------------------------
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
  float w1v[4] = {0.f, 0.f, 0.f, 0.f};
  float a_db2 = 0.f, a_db1 = 0.f, a_db3 = 0.f;
  float a_S3[NO], a_dls[NO];
#pragma unroll
  for (int c = 0; c < NO; ++c) { a_S3[c] = 0.f; a_dls[c] = 0.f; }
  LossStats st = {0.f, 0.f, 0.f, 0.f, 0.f};

  float vn_mean = 0.f, vn_sd = 1.f;
  if (HEAD == ORL_HEAD_VALUE && hp.use_valuenorm && A.vn_state != nullptr) {
    const float deb = fmaxf(A.vn_state[2], 1e-5f);
    vn_mean = A.vn_state[0] / deb;
    const float msq = A.vn_state[1] / deb;
    vn_sd = sqrtf(fmaxf(msq - vn_mean * vn_mean, 1e-2f));
  }

  const int n_tiles = (A.mb + TILE_B - 1) / TILE_B;
  const int nwv = blockDim.x >> 6;
  const int first = (blockIdx.x * nwv + wave) * NT;   // first tile of this wave's first group
  const int stride = gridDim.x * nwv * NT;            // tiles between consecutive groups of a wave

  auto row_of = [&](int t) -> long long {
    const int ii = t * TILE_B + j;
    if (t >= n_tiles || ii >= A.mb) return 0;
    return (A.idx != nullptr) ? A.idx[ii] : (long long)ii;
  };
  auto issue_dma = [&](float* slot, long long row) {
    const float* src = A.records + (size_t)row * A.R;
#pragma unroll 1
    for (int g = 0; 4 * g < nch; ++g) {
      const int c = 4 * g + q;
      if (c < nch)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 4 * c),
                                         (__attribute__((address_space(3))) void*)(slot + g * 256), 16, 0, 0);
    }
  };
  long long row_next[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    issue_dma(wl + t * per_tile + 2 * SLAB, row_of(first + t));
    row_next[t] = row_of(first + stride + t);
  }
  int ring = 0;

  for (int tile0 = first; tile0 < n_tiles; tile0 += stride) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    bool valid[NT];
    const float* RT[NT];
    float *X1[NT], *SS[NT], *DH[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      float* base = wl + t * per_tile;
      X1[t] = base; SS[t] = base + SLAB; DH[t] = base + 2 * SLAB + 2 * rts;
      RT[t] = base + 2 * SLAB + ring * rts;
      valid[t] = (tile0 + t) * TILE_B + j < A.mb && tile0 + t < n_tiles;
      long long rd = row_next[t];
      asm volatile("" : "+v"(rd));
      row_next[t] = row_of(tile0 + 2 * stride + t);
      issue_dma(base + 2 * SLAB + (ring ^ 1) * rts, rd);
    }
    ring ^= 1;
#define MREC(t, col) RT[t][(((col) >> 2) << 6) + (j << 2) + ((col) & 3)]
#define MREC_R(t, r, col) RT[t][(((col) >> 2) << 6) + ((r) << 2) + ((col) & 3)]

    // ---------------- P1: forward (xhat1 -> X1, xhat2 -> SS, head outputs) ----------------
    float rstd1[NT], rstd2[NT];
    unsigned relu_bits[NT];
    float hd[NT][NO];
    {
      f32x4 n1[NT][4], xh2[NT][4];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        f32x4 z[4];
        load_vec_T(lw + tw.b1, q, z);
        fc1_T(lw + tw.W1, DP, [&](int s) -> float { return MREC(t, A.o_x + 4 * s + q); }, z, j, q);
        relu_bits[t] = 0u;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (z[m][r] > 0.f) relu_bits[t] |= 1u << (4 * m + r);
            z[m][r] = fmaxf(z[m][r], 0.f);
          }
        ln_normalize_T(z, rstd1[t]);
        store_slab_T(X1[t], z, j, q);
        ln_affine_T(z, lw + tw.g1, lw + tw.be1, q, n1[t]);
        load_vec_T(lw + tw.b2, q, xh2[t]);
      }
      // fc2 for all tiles: every W2 operand read from LDS feeds NT MFMAs
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        f32x4 a4[4];
#pragma unroll
        for (int mo = 0; mo < 4; ++mo) a4[mo] = *(const f32x4*)(lw + tw.W2 + (16 * mo + j) * W2S + 16 * mi + 4 * q);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int mo = 0; mo < 4; ++mo) xh2[t][mo] = ORL_MFMA(a4[mo][r], n1[t][mi][r], xh2[t][mo]);
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        f32x4 z[4];
        ln_normalize_T(xh2[t], rstd2[t]);
        store_slab_T(SS[t], xh2[t], j, q);
        ln_affine_T(xh2[t], lw + tw.g2, lw + tw.be2, q, z);
        head_T<NO>(lw + tw.W3, lw + tw.b3, n_out, z, q, hd[t]);
      }
    }

    // ---------------- P2: loss -> dhead ----------------
    float dh[NT][NO];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      float dls[NO];
      ppo_row_loss<HEAD, NO>(hd[t], n_out, valid[t], [&](int col) -> float { return MREC(t, col); }, cols, hp, vn_mean,
                             vn_sd, lw + tw.logstd, dh[t], dls, st, q == 0);
      if (q == 0) {
#pragma unroll
        for (int c = 0; c < NO; ++c) {
          a_dls[c] += dls[c];
          DH[t][j * NOP + c] = dh[t][c];
        }
      }
    }
    wave_lds_fence();

    // ---------------- P3: S3 / db3 column sums; dn2, LN2 backward (xhat2 still in SS) ----------------
    f32x4 d2[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int f = l;
      float s3[NO];
#pragma unroll
      for (int c = 0; c < NO; ++c) s3[c] = 0.f;
      float s_db3 = 0.f;
      for (int r = 0; r < TILE_B; ++r) {
        const float xh = SS[t][r * TS + f];
#pragma unroll
        for (int c = 0; c < NO; ++c) s3[c] += DH[t][r * NOP + c] * xh;
        s_db3 += DH[t][r * NOP + (f < NO ? f : 0)];
      }
#pragma unroll
      for (int c = 0; c < NO; ++c) a_S3[c] += s3[c];
      a_db3 += s_db3;
#pragma unroll
      for (int m = 0; m < 4; ++m) d2[t][m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < NO; ++c) {
        if (c < n_out) {
#pragma unroll
          for (int m = 0; m < 4; ++m) d2[t][m] += *(const f32x4*)(lw + tw.W3 + c * HID + 16 * m + 4 * q) * dh[t][c];
        }
      }
      f32x4 xh2[4];
      load_slab_T(SS[t], xh2, j, q);
      ln_bwd_T(d2[t], xh2, lw + tw.g2, rstd2[t], q);
    }
    wave_lds_fence();
#pragma unroll
    for (int t = 0; t < NT; ++t) store_slab_T(SS[t], d2[t], j, q);
    wave_lds_fence();

    // ---------------- P4: wgrad G += dz2^T xhat1, db2 ----------------
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      float s_db = 0.f;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        float av[4], bv[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          av[m] = SS[t][(4 * s + q) * TS + 16 * m + j];
          bv[m] = X1[t][(4 * s + q) * TS + 16 * m + j];
        }
#pragma unroll
        for (int mo = 0; mo < 4; ++mo)
#pragma unroll
          for (int mi = 0; mi < 4; ++mi) G[mo][mi] = ORL_MFMA(av[mo], bv[mi], G[mo][mi]);
      }
      for (int r = 0; r < TILE_B; ++r) s_db += SS[t][r * TS + l];
      a_db2 += s_db;
    }

    // ---------------- P5: dgrad dn1 = W2^T dz2 (W2 columns read once for all tiles), LN1 / relu backward ------
    f32x4 d1[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int m = 0; m < 4; ++m) d1[t][m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float* row = lw + tw.W2 + (16 * mi + 4 * q + r) * W2S + j;
        float a[4];
#pragma unroll
        for (int mo = 0; mo < 4; ++mo) a[mo] = row[16 * mo];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int mo = 0; mo < 4; ++mo) d1[t][mo] = ORL_MFMA(a[mo], d2[t][mi][r], d1[t][mo]);
      }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      f32x4 xh1[4];
      load_slab_T(X1[t], xh1, j, q);
      ln_bwd_T(d1[t], xh1, lw + tw.g1, rstd1[t], q);
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (!((relu_bits[t] >> (4 * m + r)) & 1u)) d1[t][m][r] = 0.f;
    }
    wave_lds_fence();
#pragma unroll
    for (int t = 0; t < NT; ++t) store_slab_T(SS[t], d1[t], j, q);
    wave_lds_fence();

    // ---------------- P6: dW1 += dz1^T x, db1 ----------------
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int f = l;
      float s_db = 0.f;
      if (ND == 0) {
        for (int r = 0; r < TILE_B; ++r) {
          const float dzv = SS[t][r * TS + f];
          const f32x4 xv = *(const f32x4*)(&MREC_R(t, r, A.o_x));
          s_db += dzv;
          w1v[0] += dzv * xv[0]; w1v[1] += dzv * xv[1]; w1v[2] += dzv * xv[2]; w1v[3] += dzv * xv[3];
        }
      } else {
        for (int r = 0; r < TILE_B; ++r) s_db += SS[t][r * TS + f];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          float av[4], bv[NDA];
#pragma unroll
          for (int m = 0; m < 4; ++m) av[m] = SS[t][(4 * s + q) * TS + 16 * m + j];
#pragma unroll
          for (int mk = 0; mk < NDA; ++mk) bv[mk] = (16 * mk + j < D) ? MREC_R(t, 4 * s + q, A.o_x + 16 * mk + j) : 0.f;
#pragma unroll
          for (int mf = 0; mf < 4; ++mf)
#pragma unroll
            for (int mk = 0; mk < NDA; ++mk) G1[mf][mk] = ORL_MFMA(av[mf], bv[mk], G1[mf][mk]);
        }
      }
      a_db1 += s_db;
    }
    wave_lds_fence();
  }
#undef MREC
#undef MREC_R

  // ---- workgroup reduction, fixed order (same layout as the single-tile kernel) ----------------------------
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  float* acc = smem + tw.total;
  const int PW = rl.total + ORL_N_STATS;
  for (int e = threadIdx.x; e < PW; e += blockDim.x) acc[e] = 0.f;
  __syncthreads();
  st.active = wave_sum(st.active); st.rows = wave_sum(st.rows); st.loss = wave_sum(st.loss);
  st.ent = wave_sum(st.ent); st.ratio = wave_sum(st.ratio);
#pragma unroll
  for (int c = 0; c < NO; ++c) a_dls[c] = wave_sum(a_dls[c]);
  for (int w = 0; w < nwv; ++w) {
    if (wave == w) {
#pragma unroll
      for (int mo = 0; mo < 4; ++mo)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[rl.oG + (16 * mo + 4 * q + r) * HID + 16 * mi + j] += G[mo][mi][r];
      const int f = l;
      acc[rl.odb2 + f] += a_db2;
      acc[rl.odb1 + f] += a_db1;
#pragma unroll
      for (int c = 0; c < NO; ++c)
        if (c < n_out) acc[rl.oS3 + c * HID + f] += a_S3[c];
      if (f < n_out) acc[rl.odb3 + f] += a_db3;
      if (HEAD == ORL_HEAD_GAUSSIAN && f == 0) {
#pragma unroll
        for (int c = 0; c < NO; ++c)
          if (c < n_out) acc[rl.odlogstd + c] += a_dls[c];
      }
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
        acc[rl.total + ST_ACTIVE_SUM] += st.active;
        acc[rl.total + ST_ROWS] += st.rows;
        if (HEAD == ORL_HEAD_VALUE) acc[rl.total + ST_VLOSS_SUM] += st.loss;
        else {
          acc[rl.total + ST_PLOSS_SUM] += st.loss;
          acc[rl.total + ST_ENT_SUM] += st.ent;
          acc[rl.total + ST_RATIO_SUM] += st.ratio;
        }
      }
    }
    __syncthreads();
  }
  float* out = A.partials + (size_t)blockIdx.x * PW;
  for (int e = threadIdx.x; e < PW; e += blockDim.x) out[e] = acc[e];
}

inline size_t tower_mt_lds_floats(const orl_net_desc& net, int R, int nop, int nt, bool gaussian) {
  const TowerLds tw(net.obs_dim, net.n_out, gaussian, false);
  const RawLayout rl(net);
  const int rts = (((R >> 2) + 3) >> 2) * 256;
  const size_t per_tile = 2 * SLAB + 2 * rts + TILE_B * nop;
  const size_t fl = (size_t)tw.total + 4 * (size_t)nt * per_tile;
  const size_t need_acc = (size_t)tw.total + rl.total + ORL_N_STATS;
  return fl > need_acc ? fl : need_acc;
}

}  // namespace orl
