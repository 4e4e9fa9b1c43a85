// orl_gen_mlp.h - the whole-MLP inference tile of the general tower path (one 16-row tile through every layer and up to
// two heads, activations in an LDS slab, optional ACTLayer sampling on the first head's logits), shared by the
// one-launch rollout step (orl_gen_fused.hip: orl_gen_mlp_fwd / orl_gen_act) and the fused general rollout
// (orl_gen_rollout.hip: all episode_length steps in one launch).  Not part of the C ABI.
#pragma once
#include <string.h>
#include "orl_common.h"
#include "orl_mlp.h"
#include "orl_gen_act.h"
#include "orl_gen_sample.h"

namespace orl {

template <int LPR>
__device__ inline float sum_row(float v) {  // sum16's order, over an aligned group of LPR lanes
#pragma unroll
  for (int off = 1; off < LPR; off <<= 1) v += __shfl_xor(v, off);
  return v;
}

__device__ inline bool aligned16(const void* p) { return (((unsigned long long)p) & 15ull) == 0; }

// ------------------------------------------------------------------------------------------------ whole-MLP inference
// Rollout side: the entire tower (optional feature LayerNorm, every MLPLayer, up to two heads) in ONE launch.  A WAVES-wave
// workgroup (4 up to 64 columns, 8 beyond: the tile is a latency chain as long as one wave's instruction stream, see
// DESIGN.md section 11) owns 16 rows whose activations never leave an LDS slab; wave w computes the output tiles [w NBW, (w+1) NBW)
// of every layer.  The B operand (weights) comes straight from L2 as one float4 per lane and 16-k block - lane (i, q)
// reads W[16 t + i][k0 + 4 q .. + 3], i.e. the MFMA of sub-step s multiplies k = k0 + 4 q + s, and the A operand is read
// from the slab with the same k permutation (a dot product does not care) - and ALL of a layer's weight loads (up to
// 128 k) are issued at once, the next layer's right after this layer's MFMA loop, so the L2 latency hides behind the
// bias / activation / LayerNorm phase: a rollout step is a latency chain, not a throughput problem.
struct MlpArgs {
  orl_gen_mlp_desc d;
  const float* x;
  int B;
  float* head_out[2];
  float* feats;  // [B, n_out of the last trunk layer] or NULL
};

constexpr int MLP_KPRE = 8;  // 16-k blocks whose weights are in flight together

// LDS-resident copy of a tower's parameters (the fused general rollout stages it once and walks it for all T steps:
// per step the weights then cost an LDS round trip instead of an L2 one on a latency chain): per entry L the weight
// matrix [n_out][ld] (ld = n_in rounded up to 4, + 4: zero padded, rows 16 bytes apart in bank space) at off[L], then
// bias / gamma / beta [n_out rounded up to 4] each at voff[L].  Offsets in floats from `base`.
struct MlpLds {
  int off[ORL_GEN_MLP_MAX_LAYERS], ld[ORL_GEN_MLP_MAX_LAYERS], voff[ORL_GEN_MLP_MAX_LAYERS];
  int total;
};
__host__ inline MlpLds mlp_lds_layout(const orl_gen_mlp_desc& d) {
  MlpLds w;
  memset(&w, 0, sizeof(w));
  int o = 0;
  for (int L = 0; L < d.n_layers + d.n_heads; ++L) {
    const int n4 = (d.layer[L].n_in + 3) & ~3, o4 = (d.layer[L].n_out + 3) & ~3;
    w.ld[L] = n4 + 4;
    w.off[L] = o; o += d.layer[L].n_out * w.ld[L];
    w.voff[L] = o; o += 3 * o4;
  }
  w.total = o;
  return w;
}
// cooperative staging (any thread count); the caller synchronises
__device__ inline void mlp_lds_stage(const orl_gen_mlp_desc& d, const MlpLds& w, float* __restrict__ base, int tid, int nthreads) {
  for (int L = 0; L < d.n_layers + d.n_heads; ++L) {
    const orl_gen_mlp_layer& ly = d.layer[L];
    const int n_in = ly.n_in, n_out = ly.n_out, ld = w.ld[L], o4 = (n_out + 3) & ~3;
    for (int e = tid; e < n_out * ld; e += nthreads) {
      const int n = e / ld, k = e - n * ld;
      base[w.off[L] + e] = k < n_in ? ly.W[(long long)n * n_in + k] : 0.f;
    }
    for (int e = tid; e < 3 * o4; e += nthreads) {
      const int which = e / o4, c = e - which * o4;
      const float* src = which == 0 ? ly.bias : which == 1 ? ly.gamma : ly.beta;
      base[w.voff[L] + e] = (src != nullptr && c < n_out) ? src[c] : 0.f;
    }
  }
}

// ACTLayer.forward on the first head's logits inside the same launch (orl_gen_act): the 16 rows' logits go through a
// [16][LGS_LD] LDS tile behind the slab and lanes 0-15 of wave 0 run gen_sample_row ON that tile (its working copy) - the arithmetic and Philox
// counters of orl_gen_sample on the same fp32 logits, so the two routes agree bit for bit.
constexpr int LGS_LD = GEN_MAX_OUT + 1;  // odd stride: the 16 sampling lanes walk their rows conflict-free

struct ActArgs {
  orl_head_desc hd;
  const float* logstd;
  const float* amask;
  int deterministic;
  uint64_t seed, row0, rng_step;
  const unsigned long long* rng_dev;
  const float* forced;
  int a_w;
  float* actions;
  float* logp;
};

template <int NBW, int WAVES, bool WL = false>
// WL: the parameters come from the LDS copy `wl` / `wbase` (mlp_lds_stage) instead of global memory.
// roff: row offset applied to the input rows, to every head / feature output and to the sampling's action-mask / action /
// log-prob rows (the fused rollout walks the buffer's [T + 1][N] slots with roff = t * N; the tile's own rows stay
// m0 .. m0 + 15 < B = N and keep their Philox row counter); rng_add is added to the sampling's step counter.  Both are
// plain scalars on purpose: a per-step COPY of the argument structs would live in scratch memory.
__device__ __forceinline__ void mlp_tile(const MlpArgs& A, int SLD, float* __restrict__ slab, const ActArgs* S,
                                         const long long roff = 0, const unsigned long long rng_add = 0,
                                         const MlpLds* wl = nullptr, const float* __restrict__ wbase = nullptr,
                                         const bool input_in_slab = false, float* __restrict__ act_lds = nullptr) {
  // input_in_slab: the caller has already put the 16 raw input rows (zero padded to 16 columns) into the slab;
  // act_lds: the sampling writes row r's actions / log-probs to act_lds[r][0..a_w) / [a_w..2 a_w) (LDS) instead of S's
  // global arrays - the fused rollout's env step needs the action at once, a global round trip would sit on its chain
  const int tid = threadIdx.x, wave = tid >> 6, l = tid & 63, i = l & 15, q = l >> 4;
  const long long m0 = (long long)blockIdx.x * 16;
  const int B = A.B;
  const int n_total = A.d.n_layers + A.d.n_heads;
  f32x4 wr[MLP_KPRE][NBW];
  // weights of entry L, k blocks [kb0, kb0 + MLP_KPRE), this wave's tiles
  auto loadw = [&](int L, int kb0) {
    const orl_gen_mlp_layer& ly = A.d.layer[L];
    const int n_in = ly.n_in, n_out = ly.n_out;
    if constexpr (WL) {  // zero-padded rows of ld floats, 16-byte aligned: whole float4 reads
      const float* Wl = wbase + wl->off[L];
      const int ld = wl->ld[L];
#pragma unroll
      for (int kb = 0; kb < MLP_KPRE; ++kb)
#pragma unroll
        for (int j = 0; j < NBW; ++j) {
          wr[kb][j] = f32x4{0.f, 0.f, 0.f, 0.f};
          const int n = 16 * (wave * NBW + j) + i, k = 16 * (kb0 + kb) + 4 * q;
          if (n < n_out && k < n_in) wr[kb][j] = *(const f32x4*)(Wl + n * ld + k);
        }
      return;
    }
    const bool vec = (n_in & 3) == 0 && aligned16(ly.W);
#pragma unroll
    for (int kb = 0; kb < MLP_KPRE; ++kb)
#pragma unroll
      for (int j = 0; j < NBW; ++j) {
        wr[kb][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int n = 16 * (wave * NBW + j) + i, k = 16 * (kb0 + kb) + 4 * q;
        if (n < n_out && k < n_in) {
          const float* p = ly.W + (long long)n * n_in + k;
          if (vec) wr[kb][j] = *(const f32x4*)p;
          else
#pragma unroll
            for (int e = 0; e < 4; ++e) if (k + e < n_in) wr[kb][j][e] = p[e];
        }
      }
  };
  loadw(0, 0);
  // ---- input rows (+ MLPBase.feature_norm) into the slab, zero-padded to a multiple of 16 columns
  {
    const int D = A.d.layer[0].n_in, DP = (D + 15) & ~15;
    if (!input_in_slab) {
      for (int e = tid; e < 16 * DP; e += 64 * WAVES) {
        const int r = e / DP, c = e % DP;
        slab[r * SLD + c] = (c < D && m0 + r < B) ? A.x[(roff + m0 + r) * D + c] : 0.f;
      }
    }
    __syncthreads();
    if (A.d.fn_gamma != nullptr) {
      if (tid < 16) {  // one lane per row: D is an observation width
        float s = 0.f;
        for (int c = 0; c < D; ++c) s += slab[tid * SLD + c];
        const float mean = s / (float)D;
        float v2 = 0.f;
        for (int c = 0; c < D; ++c) { const float t = slab[tid * SLD + c] - mean; v2 += t * t; }
        const float rstd = 1.0f / sqrtf(v2 / (float)D + 1e-5f);
        for (int c = 0; c < D; ++c) slab[tid * SLD + c] = (slab[tid * SLD + c] - mean) * rstd * A.d.fn_gamma[c] + A.d.fn_beta[c];
      }
      __syncthreads();
    }
  }
  for (int L = 0; L < n_total; ++L) {
    const orl_gen_mlp_layer& ly = A.d.layer[L];
    const bool is_head = L >= A.d.n_layers;
    const int n_in = ly.n_in, n_out = ly.n_out;
    f32x4 acc[NBW];
#pragma unroll
    for (int j = 0; j < NBW; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nkb = (n_in + 15) >> 4;
    for (int kb0 = 0; kb0 < nkb; kb0 += MLP_KPRE) {
      if (kb0 > 0) loadw(L, kb0);
#pragma unroll
      for (int kb = 0; kb < MLP_KPRE; ++kb) {
        if (kb0 + kb < nkb) {
          const f32x4 a4 = *(const f32x4*)(slab + i * SLD + 16 * (kb0 + kb) + 4 * q);
#pragma unroll
          for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int j = 0; j < NBW; ++j) acc[j] = ORL_MFMA(a4[s], wr[kb][j][s], acc[j]);
        }
      }
    }
    if (L + 1 < n_total) loadw(L + 1, 0);  // in flight during this layer's epilogue
    if (is_head) {  // heads read the trunk's features and leave the slab alone (a second head reads them again)
      float* out = A.head_out[L - A.d.n_layers];
      const bool samp = S != nullptr && L == A.d.n_layers;
      float* lgs = slab + 16 * SLD;  // [16][LGS_LD], only there when S is
#pragma unroll
      for (int j = 0; j < NBW; ++j) {
        const int col = 16 * (wave * NBW + j) + i;
        if (col < n_out) {
          const float bv = WL ? wbase[wl->voff[L] + col] : (ly.bias ? ly.bias[col] : 0.f);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float v = acc[j][r] + bv;
            if (samp) lgs[(4 * q + r) * LGS_LD + col] = v;
            if (out && m0 + 4 * q + r < B) out[(roff + m0 + 4 * q + r) * n_out + col] = v;
          }
        }
      }
      continue;
    }
    __syncthreads();  // every wave is done reading this layer's input
#pragma unroll
    for (int j = 0; j < NBW; ++j) {
      const int col = 16 * (wave * NBW + j) + i;
      if (col < ((n_out + 15) & ~15)) {
        const float bv = WL ? (col < n_out ? wbase[wl->voff[L] + col] : 0.f) : ((col < n_out && ly.bias) ? ly.bias[col] : 0.f);
#pragma unroll
        for (int r = 0; r < 4; ++r) slab[(4 * q + r) * SLD + col] = col < n_out ? act_fwd(acc[j][r] + bv, ly.act) : 0.f;
      }
    }
    __syncthreads();
    // LayerNorm in place: wave w takes rows RPW w .. RPW w + RPW - 1, LPR lanes x float4 slots per row (n_out % 4 == 0)
    {
      constexpr int RPW = 16 / WAVES, LPR = 64 / RPW;
      const int rl = RPW * wave + l / LPR, ls = l % LPR, VPR = n_out >> 2;
      const float inv_n = 1.0f / (float)n_out;
      const bool gal = aligned16(ly.gamma) && aligned16(ly.beta);
      f32x4 v[NBW];
      float sm = 0.f;
#pragma unroll
      for (int j = 0; j < NBW; ++j) {
        const int sl = ls + LPR * j;
        v[j] = sl < VPR ? *(const f32x4*)(slab + rl * SLD + 4 * sl) : f32x4{0.f, 0.f, 0.f, 0.f};
        sm += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
      }
      const float mean = sum_row<LPR>(sm) * inv_n;
      float v2 = 0.f;
#pragma unroll
      for (int j = 0; j < NBW; ++j) {
        const bool ok = ls + LPR * j < VPR;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[j][e] = ok ? v[j][e] - mean : 0.f;
          v2 += v[j][e] * v[j][e];
        }
      }
      const float rstd = 1.0f / sqrtf(sum_row<LPR>(v2) * inv_n + 1e-5f);
#pragma unroll
      for (int j = 0; j < NBW; ++j) {
        const int sl = ls + LPR * j;
        if (sl < VPR) {
          f32x4 g, b, o;
          if constexpr (WL) {
            const int o4 = (n_out + 3) & ~3;
            g = *(const f32x4*)(wbase + wl->voff[L] + o4 + 4 * sl);
            b = *(const f32x4*)(wbase + wl->voff[L] + 2 * o4 + 4 * sl);
          } else if (gal) { g = *(const f32x4*)(ly.gamma + 4 * sl); b = *(const f32x4*)(ly.beta + 4 * sl); }
          else
#pragma unroll
            for (int e = 0; e < 4; ++e) { g[e] = ly.gamma[4 * sl + e]; b[e] = ly.beta[4 * sl + e]; }
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = v[j][e] * rstd * g[e] + b[e];
          *(f32x4*)(slab + rl * SLD + 4 * sl) = o;
          if (A.feats && L == A.d.n_layers - 1 && m0 + rl < B) {  // the trunk's features (a recurrent cell follows)
            float* fo = A.feats + (roff + m0 + rl) * n_out + 4 * sl;
            if (aligned16(A.feats)) *(f32x4*)fo = o;
            else
#pragma unroll
              for (int e = 0; e < 4; ++e) fo[e] = o[e];
          }
        }
      }
    }
    __syncthreads();
  }
  // sampling comes last, when nothing of the tile's state is live any more (its registers would otherwise add to the
  // layer loop's and halve the occupancy); the logits tile sits behind the slab and is still intact
  if (S != nullptr) {
    float* lgs = slab + 16 * SLD;
    __syncthreads();
    if (tid < 16 && m0 + tid < B) {
      const long long row = m0 + tid;
      const int NT = S->hd.n_out;
      gen_sample_row(S->hd, lgs + tid * LGS_LD, S->logstd, S->amask ? S->amask + (roff + row) * NT : nullptr, S->deterministic,
                     S->seed, S->row0 + (uint64_t)row, S->rng_step + rng_add + (S->rng_dev ? *S->rng_dev : 0ull),
                     S->forced ? S->forced + row * S->a_w : nullptr,
                     act_lds ? act_lds + tid * 2 * S->a_w : S->actions + (roff + row) * S->a_w,
                     act_lds ? act_lds + tid * 2 * S->a_w + S->a_w : S->logp + (roff + row) * S->a_w);
    }
  }
}

}  // namespace orl
