// orl_gen_fused.hip - fused kernels of the GENERAL tower path for gfx950 (any hidden_size <= 512 / layer_N / activation).
//
// orl_gen.hip runs nn.Sequential(Linear, act, LayerNorm) as GEMM + row kernel with z / a / xhat / y all round-tripping
// HBM.  At hidden_size 128 and 524 288 minibatch rows every such array is 268 MB, so that path is a stream of HBM
// passes.  The kernels here keep a row tile of a layer on the CU across the whole layer (mlp.py:8-46, MLPLayer;
// torch.nn.LayerNorm: eps 1e-5, biased variance of the centred values):
//
//   orl_gen_layer_fwd   y = LN(act(x W^T + b)): the fp32 MFMA accumulators of a [64 x n_out] tile stay in registers;
//                       bias / activation on the MFMA D fragment, then through a wave-private LDS slab into a ROW layout
//                       (a row = up to 64 lanes x float4) for the LayerNorm statistics; a (post-activation),
//                       (mean, rstd) and y leave as whole coalesced rows.                               [x | a y]
//   orl_gen_layer_bwd   dy -> dz in the row layout (LayerNorm affine + LayerNorm + activation backward, column sums of
//                       d gamma / d beta / d bias in registers), dz to HBM once (the wgrad needs it) and through the
//                       LDS slab straight into the input gradient dx = dz W of the same tile.  Square layers up to
//                       128 x 128 keep W resident in LDS and run barrier-free (gen_layer_bwd_res_kernel). [dy a | dz dx]
//   orl_gen_wgrad       dW = dz^T x with the batch rows as K: persistent split-K, both operands row-major so every
//                       load is a coalesced float4, up to a 128 x 128 output block per workgroup.   [dz x | partials]
//   orl_gen_colsum      fixed-order column sums of per-workgroup partial rows, written to up to 3 destinations.
//   orl_gen_mlp_fwd     rollout side: the WHOLE tower (feature norm, every layer, 1-2 heads) of a 16-row tile in one
//                       launch, weights straight from L2 as the MFMA B operand.
//   orl_gen_act         a rollout step in one launch: the policy tower + ACTLayer sampling on its logits, and the
//                       critic tower (grid.y) or the shared network's value head.
//
// Widths that are not a multiple of 4 or parameter vectors that are not 16-byte aligned take the same kernels on a
// scalar / fragment-layout path; n_out > 512 is refused (GenNet refuses it first).
#include <string.h>
#include "orl_common.h"
#include "orl_mlp.h"
#include "orl_gen_act.h"
#include "orl_gen_sample.h"
#include "orl_gen_mlp.h"

namespace orl {

constexpr int GF_KC = 16;
constexpr int WG_KC = 32;  // batch rows per wgrad chunk

// sum over the 16 lanes (i = lane & 15) that hold one row of an MFMA D fragment; result in all of them
__device__ inline float sum16(float v) {
  v += __shfl_xor(v, 1);
  v += __shfl_xor(v, 2);
  v += __shfl_xor(v, 4);
  v += __shfl_xor(v, 8);
  return v;
}


// Row layout of a 16-row wave slab of NP = 16 NB columns: LPR lanes (a power of two <= 64) walk one row in float4
// slots s = lane % LPR + LPR j (j < VPL; valid while 4 s < n_out), 64 / LPR rows per pass, ITS passes per slab.  Loads
// and stores are whole coalesced rows instead of the MFMA fragment's 64-byte column groups.
template <int NB>
struct RowLay {
  static constexpr int VPRP = 4 * NB;
  static constexpr int LPR = VPRP < 64 ? VPRP : 64;
  static constexpr int VPL = VPRP / LPR;
  static constexpr int RPI = 64 / LPR;
  static constexpr int ITS = 16 / RPI;
  static constexpr int UB = ITS < 4 ? ITS : 4;  // passes whose loads are issued together
};
template <int LPR>
__device__ inline float sum_lanes(float v) {  // over the LPR lanes (aligned group) of one row
#pragma unroll
  for (int off = LPR >> 1; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}
template <int LPR>
__device__ inline float sum_groups(float v) {  // over the 64 / LPR row groups of the wave
#pragma unroll
  for (int off = 32; off >= LPR; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// ------------------------------------------------------------------------------------------------ forward
// Workgroup = WAVES waves, wave w owns rows [16 w, 16 w + 16) of the tile and ALL n_out <= 16 NB columns.
// D fragment: lane (i = l & 15, q = l >> 4), tile t, register r  ->  row 4 q + r, column 16 t + i.
// WKN: W is [n_in, n_out'] with row stride ldw (k-major: y = x W, an input gradient dz W) instead of nn.Linear's [n_out, n_in]
template <int NB, int WAVES, bool WKN = false>
__global__ __launch_bounds__(64 * WAVES) void gen_layer_fwd_kernel(
    const float* __restrict__ x, int B, int n_in, const float* __restrict__ W, const float* __restrict__ bias, int act,
    const float* __restrict__ gamma, const float* __restrict__ beta, int n_out, float* __restrict__ a_out,
    float* __restrict__ stats_out, float* __restrict__ y_out, int ldo, int ldw) {
  // ldo: row stride of a_out / y_out.  gridDim.y > 1: a wide layer WITHOUT LayerNorm computed in blocks of NP columns
  // (columns are independent then): block y owns columns [y NP, y NP + NP) of the ldo-wide output.
  constexpr int BM = 16 * WAVES, NP = 16 * NB, XLD = BM + 16, WLD = NP + 16, NTH = 64 * WAVES;
  if (gridDim.y > 1) {
    const int c0 = blockIdx.y * NP;
    W += WKN ? (long long)c0 : (long long)c0 * n_in;
    if (bias) bias += c0;
    if (a_out) a_out += c0;
    if (y_out) y_out += c0;
    n_out = n_out - c0 < NP ? n_out - c0 : NP;
  }
  constexpr int KQ = GF_KC / 4;                       // float4 units per chunk row
  constexpr int XU = (BM * KQ + NTH - 1) / NTH;       // float4 units of the x chunk per thread
  constexpr int WU = (NP * KQ + NTH - 1) / NTH;       // ... of the W chunk
  extern __shared__ float sh_gf[];
  float* xs = sh_gf;                    // [2][KC][XLD]   xs[k][m]
  float* ws = sh_gf + 2 * GF_KC * XLD;  // [2][KC][WLD]   ws[k][n]
  const int tid = threadIdx.x, wave = tid >> 6, l = tid & 63, i = l & 15, q = l >> 4;
  const long long m0 = (long long)blockIdx.x * BM;
  const bool vx = (n_in & 3) == 0 && aligned16(x);
  const bool vw = WKN ? ((ldw & 3) == 0 && aligned16(W)) : ((n_in & 3) == 0 && aligned16(W));

  f32x4 rx[XU];
  f32x4 rw[WU];
  auto load = [&](int k0) {
#pragma unroll
    for (int u0 = 0; u0 < XU; ++u0) {
      const int u = tid + NTH * u0, m = u / KQ, k = k0 + 4 * (u % KQ);
      rx[u0] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (u < BM * KQ && m0 + m < B && k < n_in) {
        const float* p = x + (m0 + m) * n_in + k;
        if (vx) rx[u0] = *(const f32x4*)p;
        else
#pragma unroll
          for (int j = 0; j < 4; ++j) if (k + j < n_in) rx[u0][j] = p[j];
      }
    }
#pragma unroll
    for (int u0 = 0; u0 < WU; ++u0) {
      const int u = tid + NTH * u0;
      rw[u0] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (WKN) {  // rows of W are k: a unit is 4 consecutive output columns of one k
        const int k = k0 + u / (NP / 4), n = 4 * (u % (NP / 4));
        if (u < NP * KQ && k < n_in && n < n_out) {
          const float* p = W + (long long)k * ldw + n;
          if (vw && n + 3 < n_out) rw[u0] = *(const f32x4*)p;
          else
#pragma unroll
            for (int j = 0; j < 4; ++j) if (n + j < n_out) rw[u0][j] = p[j];
        }
      } else {
        const int n = u / KQ, k = k0 + 4 * (u % KQ);
        if (u < NP * KQ && n < n_out && k < n_in) {
          const float* p = W + (long long)n * n_in + k;
          if (vw) rw[u0] = *(const f32x4*)p;
          else
#pragma unroll
            for (int j = 0; j < 4; ++j) if (k + j < n_in) rw[u0][j] = p[j];
        }
      }
    }
  };
  auto stash = [&](int buf) {
    float* xb = xs + buf * GF_KC * XLD;
    float* wb = ws + buf * GF_KC * WLD;
#pragma unroll
    for (int u0 = 0; u0 < XU; ++u0) {
      const int u = tid + NTH * u0, m = u / KQ, kq = 4 * (u % KQ);
      if (u < BM * KQ)
#pragma unroll
        for (int j = 0; j < 4; ++j) xb[(kq + j) * XLD + m] = rx[u0][j];
    }
#pragma unroll
    for (int u0 = 0; u0 < WU; ++u0) {
      const int u = tid + NTH * u0;
      if (u < NP * KQ) {
        if (WKN) {
          *(f32x4*)(wb + (u / (NP / 4)) * WLD + 4 * (u % (NP / 4))) = rw[u0];
        } else {
          const int n = u / KQ, kq = 4 * (u % KQ);
#pragma unroll
          for (int j = 0; j < 4; ++j) wb[(kq + j) * WLD + n] = rw[u0][j];
        }
      }
    }
  };

  f32x4 acc[NB];
#pragma unroll
  for (int t = 0; t < NB; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nch = (n_in + GF_KC - 1) / GF_KC;
  load(0);
  stash(0);
  __syncthreads();
  for (int c = 0; c < nch; ++c) {
    if (c + 1 < nch) load((c + 1) * GF_KC);
    const float* xb = xs + (c & 1) * GF_KC * XLD + 16 * wave + i;
    const float* wb = ws + (c & 1) * GF_KC * WLD + i;
#pragma unroll
    for (int kk = 0; kk < GF_KC; kk += 4) {
      const float av = xb[(kk + q) * XLD];
#pragma unroll
      for (int t = 0; t < NB; ++t) acc[t] = ORL_MFMA(av, wb[(kk + q) * WLD + 16 * t], acc[t]);
    }
    if (c + 1 < nch) stash((c + 1) & 1);
    __syncthreads();
  }

  // ---- epilogue
  const float inv_n = 1.0f / (float)n_out;
  if (gamma != nullptr && (n_out & 3) == 0 && aligned16(gamma) && aligned16(beta) && (!a_out || aligned16(a_out)) &&
      (!y_out || aligned16(y_out))) {
    // LayerNorm layers: bias + activation on the fragment, then through a wave-private LDS slab into the row layout
    constexpr int LDR = NP + 4;
    float* slab = sh_gf + wave * 16 * LDR;  // aliases xs / ws: the K loop ended with a barrier
#pragma unroll
    for (int t = 0; t < NB; ++t) {
      const int col = 16 * t + i;
      const float bv = (col < n_out && bias) ? bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) slab[(4 * q + r) * LDR + col] = col < n_out ? act_fwd(acc[t][r] + bv, act) : 0.f;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    using RL = RowLay<NB>;
    const int ls = l % RL::LPR, VPR = n_out >> 2;
    f32x4 g4[RL::VPL], b4[RL::VPL];
#pragma unroll
    for (int j = 0; j < RL::VPL; ++j) {
      const int sl = ls + RL::LPR * j;
      g4[j] = sl < VPR ? *(const f32x4*)(gamma + 4 * sl) : f32x4{0.f, 0.f, 0.f, 0.f};
      b4[j] = sl < VPR ? *(const f32x4*)(beta + 4 * sl) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int it = 0; it < RL::ITS; ++it) {
      const int rl = it * RL::RPI + l / RL::LPR;
      const long long row = m0 + 16 * wave + rl;
      const bool rv = row < B;
      f32x4 v[RL::VPL];
      float sm = 0.f;
#pragma unroll
      for (int j = 0; j < RL::VPL; ++j) {
        const int sl = ls + RL::LPR * j;
        const bool ok = sl < VPR;
        v[j] = ok ? *(const f32x4*)(slab + rl * LDR + 4 * sl) : f32x4{0.f, 0.f, 0.f, 0.f};
        sm += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
        if (ok && rv && a_out) *(f32x4*)(a_out + row * n_out + 4 * sl) = v[j];
      }
      const float mean = sum_lanes<RL::LPR>(sm) * inv_n;
      float v2 = 0.f;
#pragma unroll
      for (int j = 0; j < RL::VPL; ++j) {
        const bool ok = ls + RL::LPR * j < VPR;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[j][e] = ok ? v[j][e] - mean : 0.f;
          v2 += v[j][e] * v[j][e];
        }
      }
      const float rstd = 1.0f / sqrtf(sum_lanes<RL::LPR>(v2) * inv_n + 1e-5f);
      if (rv && stats_out && ls == 0) {
        stats_out[2 * row] = mean;
        stats_out[2 * row + 1] = rstd;
      }
#pragma unroll
      for (int j = 0; j < RL::VPL; ++j) {
        const int sl = ls + RL::LPR * j;
        if (sl < VPR && rv && y_out) {
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = v[j][e] * rstd * g4[j][e] + b4[j][e];
          *(f32x4*)(y_out + row * n_out + 4 * sl) = o;
        }
      }
    }
    return;
  }
  // any other shape (heads: no LayerNorm; odd widths): on the fragment
  float bs[NB], gm[NB], bt[NB];
#pragma unroll
  for (int t = 0; t < NB; ++t) {
    const int col = 16 * t + i;
    const bool cv = col < n_out;
    bs[t] = (cv && bias) ? bias[col] : 0.f;
    gm[t] = (cv && gamma) ? gamma[col] : 0.f;
    bt[t] = (cv && gamma) ? beta[col] : 0.f;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const long long row = m0 + 16 * wave + 4 * q + r;
    const bool rv = row < B;
    float v[NB];
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < NB; ++t) {
      const bool cv = 16 * t + i < n_out;
      v[t] = cv ? act_fwd(acc[t][r] + bs[t], act) : 0.f;
      s += v[t];
      if (rv && cv && a_out) a_out[row * ldo + 16 * t + i] = v[t];
    }
    if (gamma == nullptr) {
#pragma unroll
      for (int t = 0; t < NB; ++t)
        if (rv && 16 * t + i < n_out && y_out) y_out[row * ldo + 16 * t + i] = v[t];
      continue;
    }
    const float mean = sum16(s) * inv_n;
    float v2 = 0.f;
#pragma unroll
    for (int t = 0; t < NB; ++t) {
      v[t] = 16 * t + i < n_out ? v[t] - mean : 0.f;
      v2 += v[t] * v[t];
    }
    const float rstd = 1.0f / sqrtf(sum16(v2) * inv_n + 1e-5f);
    if (rv && stats_out && i == 0) {
      stats_out[2 * row] = mean;
      stats_out[2 * row + 1] = rstd;
    }
#pragma unroll
    for (int t = 0; t < NB; ++t)
      if (rv && 16 * t + i < n_out && y_out) y_out[row * ldo + 16 * t + i] = v[t] * rstd * gm[t] + bt[t];
  }
}

// ------------------------------------------------------------------------------------------------ backward
// Persistent over 16*WAVES-row tiles.  dx (dgrad) only for square layers (n_in == n_out), which is every hidden layer
// below the first; W == nullptr skips it.
template <int NB, int WAVES, int KC = GF_KC>
__global__ __launch_bounds__(64 * WAVES) void gen_layer_bwd_kernel(
    const float* __restrict__ dy, const float* __restrict__ a, const float* __restrict__ stats,
    const float* __restrict__ gamma, int act, int B, int n_out, const float* __restrict__ W, float* __restrict__ dz_out,
    float* __restrict__ dx_out, float* __restrict__ partials) {
  constexpr int BM = 16 * WAVES, NP = 16 * NB, WLD = NP + 16, NTH = 64 * WAVES, LDR = NP + 4;
  constexpr int WU = (KC * NP / 4 + NTH - 1) / NTH;
  extern __shared__ float sh_gb[];
  float* dzs = sh_gb;                      // [WAVES][16][LDR]  dzs[w][m][c]   (wave-private, row-major)
  float* ws = sh_gb + WAVES * 16 * LDR;    // [2][KC][WLD]      ws[k = c][n]
  const int tid = threadIdx.x, wave = tid >> 6, l = tid & 63, i = l & 15, q = l >> 4;
  const int n_in = n_out;
  const bool do_dx = W != nullptr && dx_out != nullptr;
  const bool vw = (n_in & 3) == 0 && aligned16(W);
  float* dzw = dzs + wave * 16 * LDR;
  // LayerNorm layers of aligned width walk phase 1 in the row layout (whole coalesced rows); anything else on the fragment
  const bool rowlay = gamma != nullptr && (n_out & 3) == 0 && aligned16(gamma) && aligned16(dy) && aligned16(a) &&
                      (!dz_out || aligned16(dz_out));
  using RL = RowLay<NB>;
  const int ls = l % RL::LPR, VPR = n_out >> 2;

  float gm[NB], cg[NB], cb[NB], cz[NB];
#pragma unroll
  for (int t = 0; t < NB; ++t) {
    gm[t] = (!rowlay && gamma && 16 * t + i < n_out) ? gamma[16 * t + i] : 0.f;
    cg[t] = cb[t] = cz[t] = 0.f;
  }
  f32x4 g4[RL::VPL], cg4[RL::VPL], cb4[RL::VPL], cz4[RL::VPL];
#pragma unroll
  for (int j = 0; j < RL::VPL; ++j) {
    const int sl = ls + RL::LPR * j;
    g4[j] = (rowlay && sl < VPR) ? *(const f32x4*)(gamma + 4 * sl) : f32x4{0.f, 0.f, 0.f, 0.f};
    cg4[j] = cb4[j] = cz4[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const float inv_n = 1.0f / (float)n_out;
  const long long ntiles = ((long long)B + BM - 1) / BM;
  for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long long m0 = tile * BM;
    // ---- phase 1: dy -> dz
    if (rowlay) {
#pragma unroll
      for (int it0 = 0; it0 < RL::ITS; it0 += RL::UB) {
        // all loads of UB passes first: (UB x VPL x 2) float4 in flight per lane
        f32x4 gy[RL::UB][RL::VPL], av[RL::UB][RL::VPL];
        float mean[RL::UB], rstd[RL::UB];
#pragma unroll
        for (int u = 0; u < RL::UB; ++u) {
          const int rl = (it0 + u) * RL::RPI + l / RL::LPR;
          const long long row = m0 + 16 * wave + rl;
          const bool rv = row < B;
          mean[u] = rv ? stats[2 * row] : 0.f;
          rstd[u] = rv ? stats[2 * row + 1] : 1.f;
#pragma unroll
          for (int j = 0; j < RL::VPL; ++j) {
            const int sl = ls + RL::LPR * j;
            const bool ok = rv && sl < VPR;
            gy[u][j] = ok ? *(const f32x4*)(dy + row * n_out + 4 * sl) : f32x4{0.f, 0.f, 0.f, 0.f};
            av[u][j] = ok ? *(const f32x4*)(a + row * n_out + 4 * sl) : f32x4{0.f, 0.f, 0.f, 0.f};
          }
        }
#pragma unroll
        for (int u = 0; u < RL::UB; ++u) {
          const int rl = (it0 + u) * RL::RPI + l / RL::LPR;
          const long long row = m0 + 16 * wave + rl;
          const bool rv = row < B;
          f32x4 d[RL::VPL], xh[RL::VPL];
          float s1 = 0.f, s2 = 0.f;
#pragma unroll
          for (int j = 0; j < RL::VPL; ++j) {
            const bool ok = rv && ls + RL::LPR * j < VPR;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              xh[j][e] = ok ? (av[u][j][e] - mean[u]) * rstd[u] : 0.f;
              cg4[j][e] += gy[u][j][e] * xh[j][e];
              cb4[j][e] += gy[u][j][e];
              d[j][e] = gy[u][j][e] * g4[j][e];
              s1 += d[j][e];
              s2 += d[j][e] * xh[j][e];
            }
          }
          const float c1 = sum_lanes<RL::LPR>(s1) * inv_n, c2 = sum_lanes<RL::LPR>(s2) * inv_n;
#pragma unroll
          for (int j = 0; j < RL::VPL; ++j) {
            const int sl = ls + RL::LPR * j;
            const bool sv = sl < VPR;
            const bool ok = rv && sv;
            f32x4 da;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float t = (d[j][e] - c1 - xh[j][e] * c2) * rstd[u];
              if (act != ORL_ACT_NONE) t *= act_bwd(av[u][j][e], act);
              da[e] = ok ? t : 0.f;
              cz4[j][e] += da[e];
            }
            if (ok && dz_out) *(f32x4*)(dz_out + row * n_out + 4 * sl) = da;
            if (do_dx && sv) *(f32x4*)(dzw + rl * LDR + 4 * sl) = da;
          }
        }
      }
      if (do_dx)  // slab columns past n_out feed the MFMA K loop: keep them zero
        for (int e = l; e < 16 * (NP - n_out); e += 64) dzw[(e / (NP - n_out)) * LDR + n_out + e % (NP - n_out)] = 0.f;
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long long row = m0 + 16 * wave + 4 * q + r;
        const bool rv = row < B;
        float d[NB], xh[NB], av[NB];
        float s1 = 0.f, s2 = 0.f, mean = 0.f, rstd = 1.f;
        if (gamma && rv) { mean = stats[2 * row]; rstd = stats[2 * row + 1]; }
#pragma unroll
        for (int t = 0; t < NB; ++t) {
          const bool ok = rv && 16 * t + i < n_out;
          const float g = ok ? dy[row * n_out + 16 * t + i] : 0.f;
          av[t] = (ok && a) ? a[row * n_out + 16 * t + i] : 0.f;
          if (gamma) {
            xh[t] = ok ? (av[t] - mean) * rstd : 0.f;
            cg[t] += g * xh[t];
            cb[t] += g;
            d[t] = g * gm[t];
            s1 += d[t];
            s2 += d[t] * xh[t];
          } else {
            d[t] = g;
            xh[t] = 0.f;
          }
        }
        float c1 = 0.f, c2 = 0.f;
        if (gamma) { c1 = sum16(s1) * inv_n; c2 = sum16(s2) * inv_n; }
#pragma unroll
        for (int t = 0; t < NB; ++t) {
          const bool ok = rv && 16 * t + i < n_out;
          float da = gamma ? (d[t] - c1 - xh[t] * c2) * rstd : d[t];
          if (act != ORL_ACT_NONE) da *= act_bwd(av[t], act);
          if (!ok) da = 0.f;
          cz[t] += da;
          if (ok && dz_out) dz_out[row * n_out + 16 * t + i] = da;
          if (do_dx) dzw[(4 * q + r) * LDR + 16 * t + i] = da;
        }
      }
    }
    if (!do_dx) continue;
    // ---- phase 2: dx[m][n] = sum_c dz[m][c] W[c][n]
    f32x4 acc[NB];
#pragma unroll
    for (int t = 0; t < NB; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 rw[WU];
    auto load = [&](int c0) {
#pragma unroll
      for (int u0 = 0; u0 < WU; ++u0) {
        const int u = tid + NTH * u0, k = u / (NP / 4), n = 4 * (u % (NP / 4));
        rw[u0] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (u < KC * NP / 4 && c0 + k < n_out && n < n_in) {
          const float* p = W + (long long)(c0 + k) * n_in + n;
          if (vw) rw[u0] = *(const f32x4*)p;
          else
#pragma unroll
            for (int j = 0; j < 4; ++j) if (n + j < n_in) rw[u0][j] = p[j];
        }
      }
    };
    auto stash = [&](int buf) {
      float* wb = ws + buf * KC * WLD;
#pragma unroll
      for (int u0 = 0; u0 < WU; ++u0) {
        const int u = tid + NTH * u0, k = u / (NP / 4), n = 4 * (u % (NP / 4));
        if (u < KC * NP / 4) *(f32x4*)(wb + k * WLD + n) = rw[u0];
      }
    };
    const int nch = (n_out + KC - 1) / KC;
    load(0);
    __syncthreads();  // the previous tile's MFMA loop is done with both ws buffers
    stash(0);
    __syncthreads();
    for (int c = 0; c < nch; ++c) {
      if (c + 1 < nch) load((c + 1) * KC);
      const float* wb = ws + (c & 1) * KC * WLD + i;
      const float* ab = dzw + i * LDR + c * KC + q;
#pragma unroll
      for (int kk = 0; kk < KC; kk += 4) {
        const float av = ab[kk];
#pragma unroll
        for (int t = 0; t < NB; ++t) acc[t] = ORL_MFMA(av, wb[(kk + q) * WLD + 16 * t], acc[t]);
      }
      if (c + 1 < nch) stash((c + 1) & 1);
      __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long long row = m0 + 16 * wave + 4 * q + r;
      if (row < B)
#pragma unroll
        for (int t = 0; t < NB; ++t)
          if (16 * t + i < n_in) dx_out[row * n_in + 16 * t + i] = acc[t][r];
    }
  }
  // ---- column sums of this workgroup: within the wave, then over the waves -> one partial row [dg | dbeta | dbias]
  __syncthreads();
  float* cs = sh_gb;  // [WAVES][3 NP]
  if (rowlay) {
#pragma unroll
    for (int j = 0; j < RL::VPL; ++j) {
      const int sl = ls + RL::LPR * j;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float g = sum_groups<RL::LPR>(cg4[j][e]), b = sum_groups<RL::LPR>(cb4[j][e]), z = sum_groups<RL::LPR>(cz4[j][e]);
        if (l < RL::LPR && sl < VPR) {
          cs[wave * 3 * NP + 4 * sl + e] = g;
          cs[wave * 3 * NP + NP + 4 * sl + e] = b;
          cs[wave * 3 * NP + 2 * NP + 4 * sl + e] = z;
        }
      }
    }
  } else {
#pragma unroll
    for (int t = 0; t < NB; ++t) {
      const float g = row_allsum(cg[t]), b = row_allsum(cb[t]), z = row_allsum(cz[t]);
      if (q == 0) {
        cs[wave * 3 * NP + 16 * t + i] = g;
        cs[wave * 3 * NP + NP + 16 * t + i] = b;
        cs[wave * 3 * NP + 2 * NP + 16 * t + i] = z;
      }
    }
  }
  __syncthreads();
  for (int e = tid; e < 3 * NP; e += NTH) {
    const int k = e / NP, col = e % NP;
    if (col < n_out) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < WAVES; ++w) s += cs[w * 3 * NP + e];
      partials[(size_t)blockIdx.x * 3 * n_out + (size_t)k * n_out + col] = s;
    }
  }
}

// ------------------------------------------------------------------------------------------------ resident-W backward
// Square layers up to 128 x 128 on training-sized batches: the workgroup (8 waves, one per CU) loads W into LDS ONCE and
// every wave then walks its own 16-row tiles with no workgroup barrier at all - dz reaches the MFMA A operand through a
// wave-private row-major slab, so the memory phase of one wave overlaps the MFMA loop of the other wave of its SIMD
// (fc3 backward at hidden 128: 500 -> 265 us, the HBM time of its four 268 MB passes).  The same structure was tried for
// the forward and lost to the streaming kernel above (304 vs 270 us: with W resident only one 8-wave workgroup fits a
// CU and its MFMA loop is bound by the LDS operand reads, 288 B per MFMA; blocking 32 rows per wave does not fit LDS).
// square layer (n_in == n_out <= 128) with the input gradient: W row-major in LDS for the whole launch
template <int NB>
__global__ __launch_bounds__(512) void gen_layer_bwd_res_kernel(
    const float* __restrict__ dy, const float* __restrict__ a, const float* __restrict__ stats,
    const float* __restrict__ gamma, int act, int B, int n_out, const float* __restrict__ W, float* __restrict__ dz_out,
    float* __restrict__ dx_out, float* __restrict__ partials) {
  constexpr int NP = 16 * NB, WLD = NP + 16, SLD = NP + 4;
  using RL = RowLay<NB>;
  extern __shared__ float sh_br[];
  float* ws = sh_br;                // [NP][WLD]  ws[c][n] = W[c][n]
  float* slabs = sh_br + NP * WLD;  // [8][16][SLD]
  const int tid = threadIdx.x, wave = tid >> 6, l = tid & 63, i = l & 15, q = l >> 4;
  const int VPR = n_out >> 2;
  for (int u = tid; u < n_out * VPR; u += 512) {
    const int c = u / VPR, n = 4 * (u % VPR);
    *(f32x4*)(ws + c * WLD + n) = *(const f32x4*)(W + (long long)c * n_out + n);
  }
  __syncthreads();
  float* slab = slabs + wave * 16 * SLD;
  const int ls = l % RL::LPR;
  f32x4 g4[RL::VPL], cg4[RL::VPL], cb4[RL::VPL], cz4[RL::VPL];
#pragma unroll
  for (int j = 0; j < RL::VPL; ++j) {
    const int sl = ls + RL::LPR * j;
    g4[j] = sl < VPR ? *(const f32x4*)(gamma + 4 * sl) : f32x4{0.f, 0.f, 0.f, 0.f};
    cg4[j] = cb4[j] = cz4[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const float inv_n = 1.0f / (float)n_out;
  const long long ntiles = ((long long)B + 15) / 16, stride = (long long)gridDim.x * 8;
  for (long long tile = (long long)blockIdx.x * 8 + wave; tile < ntiles; tile += stride) {
    const long long m0 = tile * 16;
    // ---- phase 1 (row layout): dy -> dz, to HBM and into the slab
#pragma unroll
    for (int it0 = 0; it0 < RL::ITS; it0 += RL::UB) {
      f32x4 gy[RL::UB][RL::VPL], av[RL::UB][RL::VPL];
      float mean[RL::UB], rstd[RL::UB];
#pragma unroll
      for (int u = 0; u < RL::UB; ++u) {
        const int rl = (it0 + u) * RL::RPI + l / RL::LPR;
        const long long row = m0 + rl;
        const bool rv = row < B;
        mean[u] = rv ? stats[2 * row] : 0.f;
        rstd[u] = rv ? stats[2 * row + 1] : 1.f;
#pragma unroll
        for (int j = 0; j < RL::VPL; ++j) {
          const int sl = ls + RL::LPR * j;
          const bool ok = rv && sl < VPR;
          gy[u][j] = ok ? *(const f32x4*)(dy + row * n_out + 4 * sl) : f32x4{0.f, 0.f, 0.f, 0.f};
          av[u][j] = ok ? *(const f32x4*)(a + row * n_out + 4 * sl) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
#pragma unroll
      for (int u = 0; u < RL::UB; ++u) {
        const int rl = (it0 + u) * RL::RPI + l / RL::LPR;
        const long long row = m0 + rl;
        const bool rv = row < B;
        f32x4 d[RL::VPL], xh[RL::VPL];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < RL::VPL; ++j) {
          const bool ok = rv && ls + RL::LPR * j < VPR;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            xh[j][e] = ok ? (av[u][j][e] - mean[u]) * rstd[u] : 0.f;
            cg4[j][e] += gy[u][j][e] * xh[j][e];
            cb4[j][e] += gy[u][j][e];
            d[j][e] = gy[u][j][e] * g4[j][e];
            s1 += d[j][e];
            s2 += d[j][e] * xh[j][e];
          }
        }
        const float c1 = sum_lanes<RL::LPR>(s1) * inv_n, c2 = sum_lanes<RL::LPR>(s2) * inv_n;
#pragma unroll
        for (int j = 0; j < RL::VPL; ++j) {
          const int sl = ls + RL::LPR * j;
          const bool sv = sl < VPR;
          const bool ok = rv && sv;
          f32x4 da;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float t = (d[j][e] - c1 - xh[j][e] * c2) * rstd[u];
            if (act != ORL_ACT_NONE) t *= act_bwd(av[u][j][e], act);
            da[e] = ok ? t : 0.f;
            cz4[j][e] += da[e];
          }
          if (ok && dz_out) *(f32x4*)(dz_out + row * n_out + 4 * sl) = da;
          if (sv) *(f32x4*)(slab + rl * SLD + 4 * sl) = da;
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // ---- phase 2: dx = dz W on MFMA, no workgroup barrier
    f32x4 acc[NB];
#pragma unroll
    for (int t = 0; t < NB; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* ab = slab + i * SLD + q;
    const float* wb = ws + q * WLD + i;
#pragma unroll 4
    for (int c0 = 0; c0 < n_out; c0 += 4) {
      const float av = ab[c0];
#pragma unroll
      for (int t = 0; t < NB; ++t) acc[t] = ORL_MFMA(av, wb[c0 * WLD + 16 * t], acc[t]);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int t = 0; t < NB; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) slab[(4 * q + r) * SLD + 16 * t + i] = acc[t][r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int it = 0; it < RL::ITS; ++it) {
      const int rl = it * RL::RPI + l / RL::LPR;
      const long long row = m0 + rl;
#pragma unroll
      for (int j = 0; j < RL::VPL; ++j) {
        const int sl = ls + RL::LPR * j;
        if (row < B && sl < VPR) *(f32x4*)(dx_out + row * n_out + 4 * sl) = *(const f32x4*)(slab + rl * SLD + 4 * sl);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  // ---- column sums -> one partial row [dg | dbeta | dbias] per workgroup
  __syncthreads();
  float* cs = slabs;  // [8][3 NP]
#pragma unroll
  for (int j = 0; j < RL::VPL; ++j) {
    const int sl = ls + RL::LPR * j;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float g = sum_groups<RL::LPR>(cg4[j][e]), b = sum_groups<RL::LPR>(cb4[j][e]), z = sum_groups<RL::LPR>(cz4[j][e]);
      if (l < RL::LPR && sl < VPR) {
        cs[wave * 3 * NP + 4 * sl + e] = g;
        cs[wave * 3 * NP + NP + 4 * sl + e] = b;
        cs[wave * 3 * NP + 2 * NP + 4 * sl + e] = z;
      }
    }
  }
  __syncthreads();
  for (int e = tid; e < 3 * NP; e += 512) {
    const int k = e / NP, col = e % NP;
    if (col < n_out) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) s += cs[w * 3 * NP + e];
      partials[(size_t)blockIdx.x * 3 * n_out + (size_t)k * n_out + col] = s;
    }
  }
}

template <int NBW, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void gen_mlp_fwd_kernel(MlpArgs A, int SLD) {
  extern __shared__ float slab[];  // [16][SLD]
  mlp_tile<NBW, WAVES>(A, SLD, slab, nullptr);
}

// Rollout step of a policy / critic pair: blockIdx.y = 0 runs the policy tower and samples, 1 the critic tower.
template <int NBW, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void gen_act_kernel(MlpArgs Ap, MlpArgs Ac, ActArgs S, int SLD) {
  extern __shared__ float slab[];  // [16][SLD] + [16][LGS_LD]
  if (blockIdx.y == 0) mlp_tile<NBW, WAVES>(Ap, SLD, slab, &S);
  else mlp_tile<NBW, WAVES>(Ac, SLD, slab, nullptr);
}

// ------------------------------------------------------------------------------------------------ wgrad
// partials[z][m][n] = sum over the rows k of split z of dz[k][m0 + m] * x[k][n0 + n];  block = (64 MT) x (16 NT).
template <int MT, int NT>
__global__ __launch_bounds__(256) void gen_wgrad_kernel(const float* __restrict__ dz, const float* __restrict__ x, int B,
                                                        int n_out, int n_in, int rows_per_split,
                                                        float* __restrict__ partials) {
  constexpr int BM = 64 * MT, BN = 16 * NT, ALD = BM + 16, BLD = BN + 16;
  constexpr int AUN = WG_KC * BM / 4, BUN = WG_KC * BN / 4;  // float4 units of one chunk
  constexpr int AU = (AUN + 255) / 256, BU = (BUN + 255) / 256;
  extern __shared__ float sh_gw[];
  float* As = sh_gw;                     // [2][KC][ALD]  As[k][m]
  float* Bs = sh_gw + 2 * WG_KC * ALD;   // [2][KC][BLD]  Bs[k][n]
  const int tid = threadIdx.x, wave = tid >> 6, l = tid & 63, i = l & 15, q = l >> 4;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const long long kb = (long long)blockIdx.z * rows_per_split;
  const long long ke = min((long long)B, kb + rows_per_split);
  const bool va = (n_out & 3) == 0 && aligned16(dz), vb = (n_in & 3) == 0 && aligned16(x);
  f32x4 ra[AU], rb[BU];
  auto load = [&](long long k0) {
#pragma unroll
    for (int u0 = 0; u0 < AU; ++u0) {
      const int u = tid + 256 * u0, k = u / (BM / 4), m = m0 + 4 * (u % (BM / 4));
      ra[u0] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (u < AUN && k0 + k < ke && m < n_out) {
        const float* p = dz + (k0 + k) * n_out + m;
        if (va) ra[u0] = *(const f32x4*)p;
        else
#pragma unroll
          for (int j = 0; j < 4; ++j) if (m + j < n_out) ra[u0][j] = p[j];
      }
    }
#pragma unroll
    for (int u0 = 0; u0 < BU; ++u0) {
      const int u = tid + 256 * u0, k = u / (BN / 4), n = n0 + 4 * (u % (BN / 4));
      rb[u0] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (u < BUN && k0 + k < ke && n < n_in) {
        const float* p = x + (k0 + k) * n_in + n;
        if (vb) rb[u0] = *(const f32x4*)p;
        else
#pragma unroll
          for (int j = 0; j < 4; ++j) if (n + j < n_in) rb[u0][j] = p[j];
      }
    }
  };
  auto stash = [&](int buf) {
    float* ab = As + buf * WG_KC * ALD;
    float* bb = Bs + buf * WG_KC * BLD;
#pragma unroll
    for (int u0 = 0; u0 < AU; ++u0) {
      const int u = tid + 256 * u0;
      if (u < AUN) *(f32x4*)(ab + (u / (BM / 4)) * ALD + 4 * (u % (BM / 4))) = ra[u0];
    }
#pragma unroll
    for (int u0 = 0; u0 < BU; ++u0) {
      const int u = tid + 256 * u0;
      if (u < BUN) *(f32x4*)(bb + (u / (BN / 4)) * BLD + 4 * (u % (BN / 4))) = rb[u0];
    }
  };
  f32x4 acc[MT][NT];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[a][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const long long nch = (ke - kb + WG_KC - 1) / WG_KC;
  if (nch > 0) {
    load(kb);
    stash(0);
  }
  __syncthreads();
  for (long long c = 0; c < nch; ++c) {
    if (c + 1 < nch) load(kb + (c + 1) * WG_KC);
    const float* ab = As + (c & 1) * WG_KC * ALD + 16 * MT * wave + i;
    const float* bb = Bs + (c & 1) * WG_KC * BLD + i;
#pragma unroll
    for (int kk = 0; kk < WG_KC; kk += 4) {
      float av[MT];
#pragma unroll
      for (int a = 0; a < MT; ++a) av[a] = ab[(kk + q) * ALD + 16 * a];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const float bv = bb[(kk + q) * BLD + 16 * t];
#pragma unroll
        for (int a = 0; a < MT; ++a) acc[a][t] = ORL_MFMA(av[a], bv, acc[a][t]);
      }
    }
    if (c + 1 < nch) stash((c + 1) & 1);
    __syncthreads();
  }
  float* out = partials + (size_t)blockIdx.z * (size_t)n_out * n_in;
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gm = m0 + 16 * (MT * wave + a) + 4 * q + r, gn = n0 + 16 * t + i;
        if (gm < n_out && gn < n_in) out[(size_t)gm * n_in + gn] = acc[a][t][r];
      }
}

// ------------------------------------------------------------------------------------------------ column sums
struct ColsumDst {
  float* p[3];
  int w[3];
};
// sums[col] = sum_b partials[b][col] in a fixed order (16 row groups of a strided walk, then the groups in index order);
// column col of segment k (widths w[0..2]) goes to p[k][col - start_k] when p[k] != nullptr.
__global__ __launch_bounds__(1024) void gen_colsum_kernel(const float* __restrict__ partials, int n_rows, int width,
                                                          ColsumDst dst) {
  __shared__ float sh[16][64];
  const int c = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + c;
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  if (col < width) {
    int b = rg;
    for (; b + 48 < n_rows; b += 64) {
#pragma unroll
      for (int j = 0; j < 4; ++j) s[j] += partials[(size_t)(b + 16 * j) * width + col];
    }
    for (int j = 0; b < n_rows; b += 16, ++j) s[j & 3] += partials[(size_t)b * width + col];
  }
  sh[rg][c] = (s[0] + s[1]) + (s[2] + s[3]);
  __syncthreads();
  if (rg == 0 && col < width) {
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) t += sh[g][c];
    int k = 0, start = 0;
    while (k < 2 && col >= start + dst.w[k]) { start += dst.w[k]; ++k; }
    if (dst.p[k]) dst.p[k][col - start] = t;
  }
}

// partials[b][c] = sum of x[r][c] over the rows r = b, b + gridDim.x, ... (fixed order); gen_colsum_kernel finishes
__global__ __launch_bounds__(256) void gen_rowslab_sum_kernel(const float* __restrict__ x, long long n_rows, int width,
                                                              float* __restrict__ partials) {
  const long long g = gridDim.x;
  for (int c = threadIdx.x; c < width; c += 256) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    long long r = blockIdx.x;
    for (; r + 3 * g < n_rows; r += 4 * g) {  // four rows in flight per thread
      s0 += x[r * width + c];
      s1 += x[(r + g) * width + c];
      s2 += x[(r + 2 * g) * width + c];
      s3 += x[(r + 3 * g) * width + c];
    }
    for (; r < n_rows; r += g) s0 += x[r * width + c];
    partials[(size_t)blockIdx.x * width + c] = (s0 + s1) + (s2 + s3);
  }
}

// ------------------------------------------------------------------------------------------------ host
template <int NB, int WAVES, bool WKN = false>
static int launch_fwd(const float* x, int B, int n_in, const float* W, const float* bias, int act, const float* gamma,
                      const float* beta, int n_out, float* a_out, float* stats_out, float* y_out, hipStream_t s,
                      int ldo = 0, int ldw = 0, int col_blocks = 1) {
  constexpr int BM = 16 * WAVES, NP = 16 * NB;
  if (ldo == 0) ldo = n_out;
  size_t fl = (size_t)2 * GF_KC * ((BM + 16) + (NP + 16));
  if (fl < (size_t)WAVES * 16 * (NP + 4)) fl = (size_t)WAVES * 16 * (NP + 4);
  const size_t lds = fl * sizeof(float);
  (void)hipFuncSetAttribute((const void*)gen_layer_fwd_kernel<NB, WAVES, WKN>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds);
  const unsigned grid = (unsigned)(((long long)B + BM - 1) / BM);
  hipLaunchKernelGGL((gen_layer_fwd_kernel<NB, WAVES, WKN>), dim3(grid, col_blocks), dim3(64 * WAVES), lds, s, x, B, n_in, W,
                     bias, act, gamma, beta, n_out, a_out, stats_out, y_out, ldo, ldw);
  return launch_status("orl_gen_layer_fwd");
}

template <int NB, int WAVES, int KC = GF_KC>
static int launch_bwd(const float* dy, const float* a, const float* stats, const float* gamma, int act, int B, int n_out,
                      const float* W, float* dz_out, float* dx_out, float* partials, int max_blocks, int* n_blocks_out,
                      hipStream_t s) {
  constexpr int BM = 16 * WAVES, NP = 16 * NB;
  size_t fl = (size_t)WAVES * 16 * (NP + 4) + (size_t)2 * KC * (NP + 16);
  if (fl < (size_t)WAVES * 3 * NP) fl = (size_t)WAVES * 3 * NP;
  const size_t lds = fl * sizeof(float);
  (void)hipFuncSetAttribute((const void*)gen_layer_bwd_kernel<NB, WAVES, KC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  long long grid = ((long long)B + BM - 1) / BM;
  const int cap = max_blocks < 768 ? max_blocks : 768;  // 3 workgroups per CU
  if (grid > cap) grid = cap;
  hipLaunchKernelGGL((gen_layer_bwd_kernel<NB, WAVES, KC>), dim3((unsigned)grid), dim3(64 * WAVES), lds, s, dy, a, stats, gamma, act,
                     B, n_out, W, dz_out, dx_out, partials);
  *n_blocks_out = (int)grid;
  return launch_status("orl_gen_layer_bwd");
}

template <int NB>
static int launch_bwd_res(const float* dy, const float* a, const float* stats, const float* gamma, int act, int B, int n_out,
                          const float* W, float* dz_out, float* dx_out, float* partials, int max_blocks, int* n_blocks_out,
                          hipStream_t s) {
  constexpr int NP = 16 * NB;
  const size_t lds = ((size_t)NP * (NP + 16) + (size_t)8 * 16 * (NP + 4)) * sizeof(float);
  (void)hipFuncSetAttribute((const void*)gen_layer_bwd_res_kernel<NB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  long long grid = (((long long)B + 15) / 16 + 7) / 8;
  const int cap = max_blocks < 256 ? max_blocks : 256;
  if (grid > cap) grid = cap;
  hipLaunchKernelGGL((gen_layer_bwd_res_kernel<NB>), dim3((unsigned)grid), dim3(512), lds, s, dy, a, stats, gamma, act, B,
                     n_out, W, dz_out, dx_out, partials);
  *n_blocks_out = (int)grid;
  return launch_status("orl_gen_layer_bwd");
}

static inline bool al16(const void* p) { return p == nullptr || (((unsigned long long)p) & 15ull) == 0; }

template <int MT, int NT>
static int launch_wgrad(const float* dz, const float* x, int B, int n_out, int n_in, int n_split, int rows_per_split,
                        float* partials, hipStream_t s) {
  constexpr int BM = 64 * MT, BN = 16 * NT;
  const size_t lds = (size_t)2 * WG_KC * ((BM + 16) + (BN + 16)) * sizeof(float);
  (void)hipFuncSetAttribute((const void*)gen_wgrad_kernel<MT, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const dim3 grid((n_in + BN - 1) / BN, (n_out + BM - 1) / BM, n_split);
  hipLaunchKernelGGL((gen_wgrad_kernel<MT, NT>), grid, dim3(256), lds, s, dz, x, B, n_out, n_in, rows_per_split, partials);
  return launch_status("orl_gen_wgrad");
}

}  // namespace orl

using namespace orl;

extern "C" {

int orl_gen_layer_fwd(const float* x, int B, int n_in, const float* W, const float* bias, int act, const float* gamma,
                      const float* beta, int n_out, float* a_out, float* stats_out, float* y_out, void* stream) {
  ORL_REQUIRE(x && W && B > 0 && n_in > 0 && n_out > 0 && (n_out <= 512 || gamma == nullptr),
              "orl_gen_layer_fwd: bad arguments (B=%d n_in=%d n_out=%d; <= 512 with LayerNorm)", B, n_in, n_out);
  ORL_REQUIRE(act >= ORL_ACT_NONE && act <= ORL_ACT_ELU, "orl_gen_layer_fwd: activation id %d", act);
  ORL_REQUIRE((gamma == nullptr) == (beta == nullptr), "orl_gen_layer_fwd: gamma and beta come together");
  hipStream_t s = (hipStream_t)stream;
#define ORL_GF_FWD(NB, WV) \
  return launch_fwd<NB, WV>(x, B, n_in, W, bias, act, gamma, beta, n_out, a_out, stats_out, y_out, s)
  // rollout-sized batches: 16-row workgroups so that a few thousand rows still cover the chip
  const bool skinny = B <= 16 * 1024;
  if (gamma == nullptr && n_out > 128) {
    // no LayerNorm: output columns are independent, so a wide projection (the GRU's 3 H gate columns) runs as 128-column
    // blocks of the 8-tile kernel instead of one 32-tile-per-wave launch (256 registers, 2 waves per workgroup)
    const int cb = (n_out + 127) / 128;
    return skinny ? launch_fwd<8, 1>(x, B, n_in, W, bias, act, nullptr, nullptr, n_out, a_out, nullptr, y_out, s, n_out, 0, cb)
                  : launch_fwd<8, 4>(x, B, n_in, W, bias, act, nullptr, nullptr, n_out, a_out, nullptr, y_out, s, n_out, 0, cb);
  }
  if (n_out <= 16) { if (skinny) ORL_GF_FWD(1, 1); ORL_GF_FWD(1, 4); }
  if (n_out <= 32) { if (skinny) ORL_GF_FWD(2, 1); ORL_GF_FWD(2, 4); }
  if (n_out <= 64) { if (skinny) ORL_GF_FWD(4, 1); ORL_GF_FWD(4, 4); }
  if (n_out <= 128) { if (skinny) ORL_GF_FWD(8, 1); ORL_GF_FWD(8, 4); }
  if (n_out <= 256) { if (skinny) ORL_GF_FWD(16, 1); ORL_GF_FWD(16, 4); }
  ORL_GF_FWD(32, 2);
#undef ORL_GF_FWD
}

int orl_gen_layer_bwd(const float* dy, const float* a, const float* stats, const float* gamma, int act, int B, int n_out,
                      const float* W, int n_in, float* dz_out, float* dx_out, float* col_partials, int max_blocks,
                      int* n_blocks_out, void* stream) {
  ORL_REQUIRE(dy && col_partials && n_blocks_out && B > 0 && n_out > 0 && n_out <= 512 && max_blocks > 0,
              "orl_gen_layer_bwd: bad arguments (B=%d n_out=%d <= 512)", B, n_out);
  ORL_REQUIRE(!gamma || (a && stats), "orl_gen_layer_bwd: LayerNorm backward needs the activations and the row statistics");
  ORL_REQUIRE(act == ORL_ACT_NONE || a, "orl_gen_layer_bwd: activation backward needs the activations");
  ORL_REQUIRE((dx_out == nullptr) || (W && n_in == n_out), "orl_gen_layer_bwd: the fused dgrad takes square layers (n_in %d, n_out %d)",
              n_in, n_out);
  hipStream_t s = (hipStream_t)stream;
  const float* Wd = dx_out ? W : nullptr;
  if (dx_out && B > 16 * 1024 && gamma && n_out > 16 && n_out <= 128 && (n_out & 3) == 0 && al16(dy) && al16(a) && al16(W) &&
      al16(gamma) && al16(dz_out) && al16(dx_out)) {
    if (n_out <= 32) return launch_bwd_res<2>(dy, a, stats, gamma, act, B, n_out, W, dz_out, dx_out, col_partials, max_blocks, n_blocks_out, s);
    if (n_out <= 64) return launch_bwd_res<4>(dy, a, stats, gamma, act, B, n_out, W, dz_out, dx_out, col_partials, max_blocks, n_blocks_out, s);
    return launch_bwd_res<8>(dy, a, stats, gamma, act, B, n_out, W, dz_out, dx_out, col_partials, max_blocks, n_blocks_out, s);
  }
#define ORL_GF_BWD(NB, WV) \
  return launch_bwd<NB, WV>(dy, a, stats, gamma, act, B, n_out, Wd, dz_out, dx_out, col_partials, max_blocks, n_blocks_out, s)
  if (n_out <= 16) ORL_GF_BWD(1, 4);
  if (n_out <= 32) ORL_GF_BWD(2, 4);
  if (n_out <= 64) ORL_GF_BWD(4, 4);
  if (n_out <= 128) ORL_GF_BWD(8, 4);
  if (n_out <= 256) {
    // with the input gradient the 4-wave workgroup (101 KB of LDS, 253 registers) runs alone on its CU: 8 waves with
    // 8-k weight chunks fit 150 KB and put two waves on every SIMD
    if (dx_out) return launch_bwd<16, 8, 8>(dy, a, stats, gamma, act, B, n_out, Wd, dz_out, dx_out, col_partials, max_blocks, n_blocks_out, s);
    ORL_GF_BWD(16, 4);
  }
  ORL_GF_BWD(32, 2);
#undef ORL_GF_BWD
}

int orl_gen_colsum(const float* partials, int n_rows, int width, float* dst0, int w0, float* dst1, int w1, float* dst2,
                   int w2, void* stream) {
  ORL_REQUIRE(partials && n_rows > 0 && width > 0 && w0 >= 0 && w1 >= 0 && w2 >= 0 && w0 + w1 + w2 == width,
              "orl_gen_colsum: bad arguments (rows %d, width %d = %d + %d + %d)", n_rows, width, w0, w1, w2);
  ColsumDst d;
  d.p[0] = dst0; d.p[1] = dst1; d.p[2] = dst2;
  d.w[0] = w0; d.w[1] = w1; d.w[2] = w2;
  hipLaunchKernelGGL(gen_colsum_kernel, dim3((width + 63) / 64), dim3(1024), 0, (hipStream_t)stream, partials, n_rows, width, d);
  return launch_status("orl_gen_colsum");
}

// validates a tower description; widest layer -> *wmax, slab columns -> *width
static int check_mlp_desc(const orl_gen_mlp_desc* desc, bool out0, bool out1, bool feats, const char* who, int* wmax_io,
                          int* width_io) {
  ORL_REQUIRE(desc->n_layers >= 1 && desc->n_heads >= 0 && desc->n_heads <= 2 && (desc->n_heads >= 1 || feats) &&
                  desc->n_layers + desc->n_heads <= ORL_GEN_MLP_MAX_LAYERS,
              "%s: %d layers + %d heads (at most %d entries, 0-2 heads, features or a head wanted)", who, desc->n_layers,
              desc->n_heads, ORL_GEN_MLP_MAX_LAYERS);
  ORL_REQUIRE((desc->n_heads < 1 || out0) && (desc->n_heads < 2 || out1), "%s: a head has no output buffer", who);
  ORL_REQUIRE((desc->fn_gamma == nullptr) == (desc->fn_beta == nullptr), "%s: fn_gamma and fn_beta come together", who);
  int wmax = *wmax_io, width = *width_io;
  const int w0 = (desc->layer[0].n_in + 15) & ~15;
  if (w0 > width) width = w0;
  for (int L = 0; L < desc->n_layers + desc->n_heads; ++L) {
    const orl_gen_mlp_layer& ly = desc->layer[L];
    const bool head = L >= desc->n_layers;
    ORL_REQUIRE(ly.W && ly.n_in > 0 && ly.n_out > 0 && ly.n_out <= 256, "%s: entry %d: n_in %d, n_out %d (<= 256)", who, L, ly.n_in,
                ly.n_out);
    ORL_REQUIRE(ly.act >= ORL_ACT_NONE && ly.act <= ORL_ACT_ELU, "%s: entry %d: activation id %d", who, L, ly.act);
    if (!head) {
      ORL_REQUIRE(ly.gamma && ly.beta && (ly.n_out & 3) == 0, "%s: layer %d needs LayerNorm parameters and a width that is a multiple of 4", who, L);
      ORL_REQUIRE(L == 0 || ly.n_in == desc->layer[L - 1].n_out, "%s: layer %d reads %d columns, layer %d writes %d", who, L, ly.n_in, L - 1, desc->layer[L - 1].n_out);
    } else {
      ORL_REQUIRE(ly.n_in == desc->layer[desc->n_layers - 1].n_out && ly.act == ORL_ACT_NONE && !ly.gamma,
                  "%s: head %d must be a plain Linear on the trunk's %d features", who, L - desc->n_layers,
                  desc->layer[desc->n_layers - 1].n_out);
    }
    if (ly.n_out > wmax) wmax = ly.n_out;
    const int w16 = (ly.n_out + 15) & ~15;
    if (w16 > width) width = w16;
  }
  ORL_REQUIRE(width <= 1024, "%s: %d columns do not fit the wave's LDS slab", who, width);
  *wmax_io = wmax; *width_io = width;
  return 0;
}

int orl_gen_mlp_fwd(const orl_gen_mlp_desc* desc, const float* x, int B, float* head_out0, float* head_out1,
                    float* feats_out, void* stream) {
  ORL_REQUIRE(desc && x && B > 0, "orl_gen_mlp_fwd: bad arguments");
  int wmax = 0, width = 0;
  int rc = check_mlp_desc(desc, head_out0 != nullptr, head_out1 != nullptr, feats_out != nullptr, "orl_gen_mlp_fwd", &wmax, &width);
  if (rc) return rc;
  MlpArgs A;
  A.d = *desc; A.x = x; A.B = B; A.head_out[0] = head_out0; A.head_out[1] = head_out1; A.feats = feats_out;
  // NBW output tiles per wave: widths up to 64 with 4 waves, beyond that 8 waves - a rollout step is one wave per SIMD
  // walking a latency chain whose length is its instruction count, so wide layers are split over more waves
  const int NBW = wmax <= 128 ? 1 : 2, WV = wmax <= 64 ? 4 : 8;
  if (width < 16 * WV * NBW) width = 16 * WV * NBW;
  const int SLD = width + 4;
  const size_t lds = (size_t)16 * SLD * sizeof(float);
  const unsigned grid = (unsigned)(((long long)B + 15) / 16);
#define ORL_MLP_LAUNCH(NBX, WVX)                                                                                     \
  do {                                                                                                               \
    (void)hipFuncSetAttribute((const void*)gen_mlp_fwd_kernel<NBX, WVX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL((gen_mlp_fwd_kernel<NBX, WVX>), dim3(grid), dim3(64 * WVX), lds, (hipStream_t)stream, A, SLD); \
  } while (0)
  if (WV == 4) ORL_MLP_LAUNCH(1, 4);
  else if (NBW == 1) ORL_MLP_LAUNCH(1, 8);
  else ORL_MLP_LAUNCH(2, 8);
#undef ORL_MLP_LAUNCH
  return launch_status("orl_gen_mlp_fwd");
}

int orl_gen_act(const orl_gen_mlp_desc* policy, const float* obs, const orl_gen_mlp_desc* critic, const float* critic_obs,
                int B, float* logits_out, float* values, const orl_head_desc* head, const float* logstd,
                const float* action_masks, int deterministic, uint64_t seed, uint64_t row0, uint64_t rng_step,
                const uint64_t* rng_step_dev, const float* forced_u, int a_w, float* actions, float* logp, void* stream) {
  ORL_REQUIRE(policy && obs && head && actions && logp && B > 0 && a_w > 0, "orl_gen_act: bad arguments");
  ORL_REQUIRE(policy->n_heads >= 1 && (critic == nullptr || (policy->n_heads == 1 && critic->n_heads == 1 && critic_obs)),
              "orl_gen_act: the policy carries the action head (+ the value head of a shared network), a separate critic one value head");
  ORL_REQUIRE((policy->n_heads == 2 || critic) == (values != nullptr), "orl_gen_act: values goes with a value head");
  ORL_REQUIRE(head->kind == ORL_HEAD_CATEGORICAL || head->kind == ORL_HEAD_GAUSSIAN || head->kind == ORL_HEAD_MULTI_DISCRETE ||
                  head->kind == ORL_HEAD_MIXED,
              "orl_gen_act: head kind %d", head->kind);
  ORL_REQUIRE(head->n_out >= 1 && head->n_out <= GEN_MAX_OUT && head->n_out == policy->layer[policy->n_layers].n_out,
              "orl_gen_act: the head has %d logits (1..%d), the tower's first head %d", head->n_out, GEN_MAX_OUT,
              policy->layer[policy->n_layers].n_out);
  if (head->kind == ORL_HEAD_MULTI_DISCRETE) {
    ORL_REQUIRE(head->n_heads >= 1 && head->n_heads <= ORL_MAX_HEADS && a_w == head->n_heads, "orl_gen_act: %d components, a_w %d", head->n_heads, a_w);
    int tot = 0;
    for (int h = 0; h < head->n_heads; ++h) { ORL_REQUIRE(head->nvec[h] >= 1, "orl_gen_act: nvec[%d] = %d", h, head->nvec[h]); tot += head->nvec[h]; }
    ORL_REQUIRE(tot == head->n_out, "orl_gen_act: nvec sums to %d, n_out is %d", tot, head->n_out);
  } else if (head->kind == ORL_HEAD_MIXED) {
    ORL_REQUIRE(head->n_heads == 2 && head->nvec[0] >= 1 && head->nvec[0] <= 15 && head->nvec[1] >= 1 &&
                    head->nvec[0] + head->nvec[1] == head->n_out && a_w == head->nvec[0] + 1,
                "orl_gen_act: mixed head {%d, %d}, n_out %d, a_w %d", head->nvec[0], head->nvec[1], head->n_out, a_w);
  } else {
    ORL_REQUIRE(a_w == (head->kind == ORL_HEAD_GAUSSIAN ? head->n_out : 1), "orl_gen_act: a_w %d for head kind %d", a_w, head->kind);
  }
  ORL_REQUIRE((head->kind != ORL_HEAD_GAUSSIAN && head->kind != ORL_HEAD_MIXED) || logstd, "orl_gen_act: Gaussian head without logstd");
  int wmax = 0, width = 0;
  int rc = check_mlp_desc(policy, true, policy->n_heads == 2, false, "orl_gen_act(policy)", &wmax, &width);
  if (rc) return rc;
  if (critic) {
    ORL_REQUIRE(critic->layer[critic->n_layers].n_out == 1, "orl_gen_act: the critic's head has %d outputs", critic->layer[critic->n_layers].n_out);
    rc = check_mlp_desc(critic, true, false, false, "orl_gen_act(critic)", &wmax, &width);
    if (rc) return rc;
  } else if (policy->n_heads == 2) {
    ORL_REQUIRE(policy->layer[policy->n_layers + 1].n_out == 1, "orl_gen_act: the value head has %d outputs", policy->layer[policy->n_layers + 1].n_out);
  }
  MlpArgs Ap, Ac;
  Ap.d = *policy; Ap.x = obs; Ap.B = B; Ap.head_out[0] = logits_out; Ap.head_out[1] = critic ? nullptr : values; Ap.feats = nullptr;
  Ac = Ap;
  if (critic) { Ac.d = *critic; Ac.x = critic_obs; Ac.head_out[0] = values; Ac.head_out[1] = nullptr; }
  ActArgs S;
  S.hd = *head; S.logstd = logstd; S.amask = action_masks; S.deterministic = deterministic; S.seed = seed; S.row0 = row0;
  S.rng_step = rng_step; S.rng_dev = (const unsigned long long*)rng_step_dev; S.forced = forced_u; S.a_w = a_w;
  S.actions = actions; S.logp = logp;
  const int NBW = wmax <= 128 ? 1 : 2, WV = wmax <= 64 ? 4 : 8;  // as orl_gen_mlp_fwd picks them
  if (width < 16 * WV * NBW) width = 16 * WV * NBW;
  const int SLD = width + 4;
  const size_t lds = ((size_t)16 * SLD + 16 * LGS_LD) * sizeof(float);
  const dim3 grid((unsigned)(((long long)B + 15) / 16), critic ? 2 : 1);
#define ORL_ACT_LAUNCH(NBX, WVX)                                                                                     \
  do {                                                                                                               \
    (void)hipFuncSetAttribute((const void*)gen_act_kernel<NBX, WVX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL((gen_act_kernel<NBX, WVX>), grid, dim3(64 * WVX), lds, (hipStream_t)stream, Ap, Ac, S, SLD);  \
  } while (0)
  if (WV == 4) ORL_ACT_LAUNCH(1, 4);
  else if (NBW == 1) ORL_ACT_LAUNCH(1, 8);
  else ORL_ACT_LAUNCH(2, 8);
#undef ORL_ACT_LAUNCH
  return launch_status("orl_gen_act");
}

int orl_gen_matmul(const float* x, int B, int K, const float* W, int N, float* y, void* stream) {
  ORL_REQUIRE(x && W && y && B > 0 && K > 0 && N > 0, "orl_gen_matmul: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const bool skinny = B <= 16 * 1024;
  const int cb = (N + 127) / 128;  // 128 output columns (8 accumulator tiles per wave) per block column
  return skinny ? launch_fwd<8, 1, true>(x, B, K, W, nullptr, ORL_ACT_NONE, nullptr, nullptr, N, nullptr, nullptr, y, s, N, N, cb)
                : launch_fwd<8, 4, true>(x, B, K, W, nullptr, ORL_ACT_NONE, nullptr, nullptr, N, nullptr, nullptr, y, s, N, N, cb);
}

int orl_gen_colsum_rows(const float* x, int n_rows, int width, float* dst, float* partials, int64_t partials_floats,
                        void* stream) {
  ORL_REQUIRE(x && dst && partials && n_rows > 0 && width > 0, "orl_gen_colsum_rows: bad arguments");
  long long nb = partials_floats / width;
  if (nb > 512) nb = 512;
  if (nb > n_rows) nb = n_rows;
  ORL_REQUIRE(nb >= 1, "orl_gen_colsum_rows: the partials buffer (%lld floats) does not hold one row of %d columns",
              (long long)partials_floats, width);
  hipLaunchKernelGGL(gen_rowslab_sum_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, x, (long long)n_rows,
                     width, partials);
  int rc = launch_status("orl_gen_colsum_rows");
  if (rc) return rc;
  return orl_gen_colsum(partials, (int)nb, width, dst, width, nullptr, 0, nullptr, 0, stream);
}

int orl_gen_wgrad(const float* dz, const float* x, int B, int n_out, int n_in, float* dW, float* partials,
                  int64_t partials_floats, void* stream) {
  ORL_REQUIRE(dz && x && dW && partials && B > 0 && n_out > 0 && n_in > 0, "orl_gen_wgrad: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const bool big = n_out > 64 && n_in > 64;
  const int BM = big ? 128 : 64, BN = big ? 128 : (n_in <= 16 ? 16 : n_in > 64 ? 128 : 64);
  const int tiles = ((n_out + BM - 1) / BM) * ((n_in + BN - 1) / BN);
  long long n_split = 512 / tiles;
  if (n_split < 1) n_split = 1;
  const long long by_rows = ((long long)B + 255) / 256;  // at least 256 rows per split
  if (n_split > by_rows) n_split = by_rows;
  const long long by_mem = partials_floats / ((long long)n_out * n_in);
  ORL_REQUIRE(by_mem >= 1, "orl_gen_wgrad: the partials buffer (%lld floats) does not hold one %d x %d block",
              (long long)partials_floats, n_out, n_in);
  if (n_split > by_mem) n_split = by_mem;
  long long rps = (((long long)B + n_split - 1) / n_split + WG_KC - 1) / WG_KC * WG_KC;
  n_split = ((long long)B + rps - 1) / rps;
  int rc;
  if (big) rc = launch_wgrad<2, 8>(dz, x, B, n_out, n_in, (int)n_split, (int)rps, partials, s);
  else if (n_in <= 16) rc = launch_wgrad<1, 1>(dz, x, B, n_out, n_in, (int)n_split, (int)rps, partials, s);
  else if (n_in > 64) rc = launch_wgrad<1, 8>(dz, x, B, n_out, n_in, (int)n_split, (int)rps, partials, s);
  else rc = launch_wgrad<1, 4>(dz, x, B, n_out, n_in, (int)n_split, (int)rps, partials, s);
  if (rc) return rc;
  return orl_gen_colsum(partials, (int)n_split, n_out * n_in, dW, n_out * n_in, nullptr, 0, nullptr, 0, stream);
}

}  // extern "C"
