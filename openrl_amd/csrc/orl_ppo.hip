// orl_ppo.hip - PPO update side of the hot path for gfx950:
//   orl_ppo_fwd_bwd : K9-K12 fused tower forward + PPO/value/entropy loss + full backward, one launch
//                     per tower (kernel in orl_ppo_tower.h), producing per-workgroup partial sums of the
//                     RAW gradient
//   orl_ppo_reduce  : deterministic column sums of the partials (the vector a multi-GPU run all-reduces)
//   orl_ppo_apply   : raw sums -> parameter gradients, grad-norm clip (K13), Adam (K14), train_info
#include <stdlib.h>
#include "orl_common.h"
#include "orl_mlp.h"
#include "orl_ppo_tower.h"
#include "orl_ppo_tower_mt.h"

namespace orl {

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
  if (rg == 0 && col < width)
    sums[col] = (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
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

// beta ** step for the Adam bias corrections: square-and-multiply in double (a dozen multiplies; libm's pow is a few
// hundred cold instructions in a kernel whose run time is instruction fetch).  Agrees with pow() to ~1e-15 relative.
__device__ inline double powi_d(double b, long long n) {
  double r = 1.0;
  while (n > 0) {
    if (n & 1) r *= b;
    b *= b;
    n >>= 1;
  }
  return r;
}

// gradient of parameter p (parameter order) from the raw sums of one tower, already divided by den
__device__ inline float raw_to_grad(const float* __restrict__ raw, const float* __restrict__ theta,
                                    const TowerLayout& tl, const RawLayout& rl, int p, float inv_den) {
  const int H = HID;
  float g;
  if (p < tl.ob1) g = raw[rl.odW1 + (p - tl.oW1)];
  else if (p < tl.og1) g = raw[rl.odb1 + (p - tl.ob1)];
  else if (p < tl.obe1) {  // LN1 weight: dg1[i] = sum_o W2[o][i] * G[o][i]
    const int i = p - tl.og1;
    g = 0.f;
    for (int o = 0; o < H; ++o) g += theta[tl.oW2 + o * H + i] * raw[rl.oG + o * H + i];
  } else if (p < tl.oW2) {  // LN1 bias: dbe1[i] = sum_o W2[o][i] * db2[o]
    const int i = p - tl.obe1;
    g = 0.f;
    for (int o = 0; o < H; ++o) g += theta[tl.oW2 + o * H + i] * raw[rl.odb2 + o];
  } else if (p < tl.ob2) {
    const int e = p - tl.oW2, o = e / H, i = e - o * H;
    g = theta[tl.og1 + i] * raw[rl.oG + e] + theta[tl.obe1 + i] * raw[rl.odb2 + o];
  } else if (p < tl.og2) g = raw[rl.odb2 + (p - tl.ob2)];
  else if (p < tl.obe2) {  // LN2 weight: dg2[f] = sum_c W3[c][f] * S3[c][f]
    const int f = p - tl.og2;
    g = 0.f;
    for (int c = 0; c < tl.n_out; ++c) g += theta[tl.oW3 + c * H + f] * raw[rl.oS3 + c * H + f];
  } else if (p < tl.oW3) {  // LN2 bias: dbe2[f] = sum_c W3[c][f] * db3[c]
    const int f = p - tl.obe2;
    g = 0.f;
    for (int c = 0; c < tl.n_out; ++c) g += theta[tl.oW3 + c * H + f] * raw[rl.odb3 + c];
  } else if (p < tl.ob3) {
    const int e = p - tl.oW3, c = e / H, f = e - c * H;
    g = theta[tl.og2 + f] * raw[rl.oS3 + e] + theta[tl.obe2 + f] * raw[rl.odb3 + c];
  } else if (p < tl.ologstd) g = raw[rl.odb3 + (p - tl.ob3)];
  else g = raw[rl.odlogstd + (p - tl.ologstd)];
  return g * inv_den;
}

// One workgroup per tower (blockIdx.x = 0 policy, 1 critic).  The tower's raw sums and parameters are
// staged in LDS first so the 64-term LayerNorm-affine dot products of raw_to_grad run out of LDS.
__global__ __launch_bounds__(1024) void ppo_apply_kernel(ApplyTower P, ApplyTower Cc, const float* __restrict__ sums,
                                                         orl_ppo_hparams hp, float* __restrict__ info) {
  extern __shared__ __attribute__((aligned(16))) float s_apply[];
  __shared__ float sh[16];
  const TowerLayout tlp(P.net), tlc(Cc.net);
  const RawLayout rlp(P.net), rlc(Cc.net);
  const float* rawp = sums + P.sums_off;
  const float* stp = rawp + rlp.total;
  const float* rawc = sums + Cc.sums_off;
  const float* stc = rawc + rlc.total;
  const float den_p = hp.use_policy_active_masks ? stp[ST_ACTIVE_SUM] : stp[ST_ROWS];
  const float den_v = hp.use_value_active_masks ? stc[ST_ACTIVE_SUM] : stc[ST_ROWS];
  float norms[2] = {0.f, 0.f};
  {
    const int t = blockIdx.x;
    const ApplyTower& W = t == 0 ? P : Cc;
    const TowerLayout& tl = t == 0 ? tlp : tlc;
    const RawLayout& rl = t == 0 ? rlp : rlc;
    const float* raw_g = t == 0 ? rawp : rawc;
    const float inv_den = 1.0f / (t == 0 ? den_p : den_v);
    if (t == 0 && (hp.reserved & 1)) {
      // turn_on == False: the policy loss is not in the loss list (ppo.py:226-236) -> no gradient, no step
      for (int p = threadIdx.x; p < tl.total; p += blockDim.x) W.ad.grad[p] = 0.f;
      if (threadIdx.x == 0 && info != nullptr) {
        info[1] += stp[ST_PLOSS_SUM] / den_p;
        float ent_den0 = den_p;
        if (!hp.use_policy_active_masks && P.net.head_kind == ORL_HEAD_GAUSSIAN) ent_den0 = den_p * (float)P.net.n_out;
        info[2] += stp[ST_ENT_SUM] / ent_den0;
        const float aw0 = P.net.head_kind == ORL_HEAD_GAUSSIAN ? (float)P.net.n_out : 1.f;
        info[5] += stp[ST_RATIO_SUM] / (stp[ST_ROWS] * aw0);
      }
      return;
    }
    float* raw = s_apply;
    float* th_s = s_apply + rl.total;
    // latency plan for a one-shot, two-workgroup kernel: everything that comes from HBM (raw sums, parameters, Adam
    // moments) is requested up front into LDS, the double-precision bias corrections are evaluated while those loads
    // are in flight, gradients stay in LDS between the norm and the Adam pass, and every loop stays ROLLED - the code
    // is fetched cold on each launch, so instruction bytes cost more than loop overhead (an unrolled variant of this
    // body measured 17.2 us against 13.3 us).
    float* m_s = th_s + tl.total;
    float* v_s = m_s + tl.total;
    float* g_s = v_s + tl.total;
#pragma unroll 1
    for (int e = threadIdx.x; e < rl.total; e += blockDim.x) raw[e] = raw_g[e];
#pragma unroll 1
    for (int e = threadIdx.x; e < tl.total; e += blockDim.x) {
      th_s[e] = W.ad.theta[e];
      m_s[e] = W.ad.m[e];
      v_s[e] = W.ad.v[e];
    }
    // torch.optim.Adam (single tensor math, betas (0.9, 0.999), amsgrad off)
    // scalar coefficients are python doubles in torch/optim/adam.py; only tensor math is fp32
    const double b1d = 0.9, b2d = 0.999;
    const float b2 = (float)b2d, omb1 = (float)(1.0 - b1d), omb2 = (float)(1.0 - b2d);
    const double bc1 = 1.0 - powi_d(b1d, (long long)W.ad.step);
    const double bc2 = 1.0 - powi_d(b2d, (long long)W.ad.step);
    const float step_size = (float)((double)W.ad.lr / bc1);
    const float bc2_sqrt = (float)sqrt(bc2);
    __syncthreads();
    float ss = 0.f;
#pragma unroll 1
    for (int p = threadIdx.x; p < tl.total; p += blockDim.x) {
      const float g = raw_to_grad(raw, th_s, tl, rl, p, inv_den);
      g_s[p] = g;
      ss += g * g;
    }
    const float total = sqrtf(block_sum_1024(ss, sh));
    norms[t] = total;
    // torch.nn.utils.clip_grad_norm_: coef = max_norm / (norm + 1e-6), clamped to 1
    float coef = 1.f;
    if (hp.use_max_grad_norm) coef = fminf(hp.max_grad_norm / (total + 1e-6f), 1.f);
#pragma unroll 1
    for (int p = threadIdx.x; p < tl.total; p += blockDim.x) {
      float g = g_s[p] * coef;
      W.ad.grad[p] = g;
      float th = th_s[p];
      if (W.ad.weight_decay != 0.f) g += W.ad.weight_decay * th;
      float m = m_s[p], v = v_s[p];
      m = m + (g - m) * omb1;          // exp_avg.lerp_(grad, 1 - beta1)
      v = v * b2 + omb2 * (g * g);     // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
      const float denom = sqrtf(v) / bc2_sqrt + W.ad.eps;
      th = th - step_size * (m / denom);
      W.ad.m[p] = m; W.ad.v[p] = v; W.ad.theta[p] = th;
    }
  }
  if (threadIdx.x == 0 && info != nullptr) {
    if (blockIdx.x == 0) {
      float ent_den = den_p;
      if (!hp.use_policy_active_masks && P.net.head_kind == ORL_HEAD_GAUSSIAN) ent_den = den_p * (float)P.net.n_out;
      info[1] += stp[ST_PLOSS_SUM] / den_p;                 // policy_loss
      info[2] += stp[ST_ENT_SUM] / ent_den;                 // dist_entropy
      info[3] += norms[0];                                  // actor_grad_norm
      const float a_w = P.net.head_kind == ORL_HEAD_GAUSSIAN ? (float)P.net.n_out : 1.f;
      info[5] += stp[ST_RATIO_SUM] / (stp[ST_ROWS] * a_w);  // ratio.mean()
    } else {
      info[0] += stc[ST_VLOSS_SUM] / den_v;                 // value_loss
      info[4] += norms[1];                                  // critic_grad_norm
    }
  }
}

// both towers' partial regions in one launch: blocks [0, gp) reduce the policy region, the rest the critic's
__global__ __launch_bounds__(1024) void ppo_reduce_pair_kernel(const float* __restrict__ pp, int nb_p, int wp, int gp,
                                                               const float* __restrict__ pc, int nb_c, int wc,
                                                               float* __restrict__ sums) {
  // 64 columns x 16 row groups per workgroup: at 256 partial rows every thread has 16 independent loads in flight
  __shared__ float sh[16][64];
  const bool pol = (int)blockIdx.x < gp;
  const float* partials = pol ? pp : pc;
  const int n_blocks = pol ? nb_p : nb_c, width = pol ? wp : wc;
  float* out = pol ? sums : sums + wp;
  const int lc = threadIdx.x & 63;
  const int col = (pol ? blockIdx.x : blockIdx.x - gp) * 64 + lc;
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

static int check_tower(const orl_net_desc* n, const char* who) {
  if (!n) return fail(ORL_E_INVALID, "%s: null net descriptor", who);
  if (n->hidden != HID) return fail(ORL_E_UNSUPPORTED, "%s: hidden_size %d not built (only 64)", who, n->hidden);
  if (n->obs_dim < 1 || n->obs_dim > 64) return fail(ORL_E_UNSUPPORTED, "%s: obs_dim %d outside [1,64]", who, n->obs_dim);
  if (n->n_out < 1 || n->n_out > 16) return fail(ORL_E_UNSUPPORTED, "%s: n_out %d outside [1,16]", who, n->n_out);
  return 0;
}

template <int HEAD, int NO, int ND, int WPS, bool PC>
static int launch_tower_w(const PpoArgs& A, int waves, size_t lds, hipStream_t s) {
  const int n_tiles = (A.mb + TILE_B - 1) / TILE_B;
  const int walkers = PC ? 8 : waves;  // waves per workgroup that walk tiles
  int grid = (n_tiles + walkers - 1) / walkers;
  if (grid > PPO_MAX_BLOCKS) grid = PPO_MAX_BLOCKS;
  (void)hipFuncSetAttribute((const void*)ppo_tower_kernel<HEAD, NO, ND, WPS, PC>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((ppo_tower_kernel<HEAD, NO, ND, WPS, PC>), dim3(grid), dim3(waves * 64), lds, s, A);
  return grid;
}

// Launch one tower; returns the number of workgroups (> 0) or a negative error code.
template <int HEAD, int NO, int ND>
static int launch_tower(const PpoArgs& A, hipStream_t s) {
  constexpr int NOP = (NO + 3) & ~3;
  // as many waves per workgroup (8, 6, 4, 2) as fit the 160 KiB of LDS next to the tower's weights.
  // 12 waves (3 per SIMD, <= 168 VGPRs) fit the LDS budget too but were measured SLOWER on MI355X
  // (5.94 vs 5.21 ms per iteration): at 168 VGPRs hipcc spills 260 B per lane around the 64 wgrad
  // accumulators, so the kernel is built for 2 waves per SIMD.
  static const int kWaves[4] = {8, 6, 4, 2};
  static const int max_waves = []() {  // tuning knob for A/B runs: ORL_PPO_WAVES=4 caps the workgroup size
    const char* e = getenv("ORL_PPO_WAVES");
    return e ? atoi(e) : 8;  // 12 = producer/consumer build: correct but measured slower (0.408 vs 0.384 ms)
  }();
  static const int mt = []() {  // ORL_PPO_MT=2: A/B knob for the multi-tile kernel (2 tiles per wave, 1 wave / SIMD)
    const char* e = getenv("ORL_PPO_MT");
    return e ? atoi(e) : 0;
  }();
  if constexpr (ND == 0) {
    if (mt == 2) {  // measured SLOWER than the default (0.457 vs 0.367 ms per pair), kept for A/B runs: DESIGN.md 6
      const size_t lds = tower_mt_lds_floats(A.net, A.R, NOP, 2, HEAD == ORL_HEAD_GAUSSIAN) * sizeof(float);
      if (lds <= 160 * 1024) {
        const int n_tiles = (A.mb + TILE_B - 1) / TILE_B;
        int grid = (n_tiles + 7) / 8;
        if (grid > PPO_MAX_BLOCKS) grid = PPO_MAX_BLOCKS;
        (void)hipFuncSetAttribute((const void*)ppo_tower_mt_kernel<HEAD, NO, ND, 2>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((ppo_tower_mt_kernel<HEAD, NO, ND, 2>), dim3(grid), dim3(256), lds, s, A);
        const int rc = launch_status("orl_ppo_fwd_bwd(mt)");
        return rc ? -1000 - rc : grid;
      }
    }
  }
  if (max_waves >= 12) {  // producer / consumer build: 8 producers + 4 consumers, 3 waves per SIMD
    const size_t lds = tower_lds_floats(A.net, A.R, NOP, 12, HEAD == ORL_HEAD_GAUSSIAN, true) * sizeof(float);
    if (lds <= 160 * 1024) {
      const int grid = launch_tower_w<HEAD, NO, ND, 3, true>(A, 12, lds, s);
      const int rc = launch_status("orl_ppo_fwd_bwd");
      return rc ? -1000 - rc : grid;
    }
  }
  for (int k = 0; k < 4; ++k) {
    const int waves = kWaves[k];
    if (waves > max_waves) continue;
    const size_t lds = tower_lds_floats(A.net, A.R, NOP, waves, HEAD == ORL_HEAD_GAUSSIAN) * sizeof(float);
    if (lds > 160 * 1024) continue;
    const int grid = launch_tower_w<HEAD, NO, ND, 2, false>(A, waves, lds, s);
    const int rc = launch_status("orl_ppo_fwd_bwd");
    return rc ? -1000 - rc : grid;
  }
  fail(ORL_E_UNSUPPORTED, "orl_ppo_fwd_bwd: tower (obs %d, record %d floats) does not fit 160 KiB of LDS", A.net.obs_dim,
       A.R);
  return ORL_E_UNSUPPORTED;
}

template <int HEAD, int NO>
static int launch_tower_nd(const PpoArgs& A, hipStream_t s) {
  const int D = A.net.obs_dim;
  if (D <= 4 && (A.o_x & 3) == 0) return launch_tower<HEAD, NO, 0>(A, s);
  if (D <= 16) return launch_tower<HEAD, NO, 1>(A, s);
  if (D <= 32) return launch_tower<HEAD, NO, 2>(A, s);
  return launch_tower<HEAD, NO, 4>(A, s);
}

}  // namespace orl

using namespace orl;

extern "C" {

int orl_ppo_max_blocks(void) { return PPO_MAX_BLOCKS; }

#ifdef ORL_PROF
// debug build only: cumulative per-phase cycle counts of wave 0 / workgroup 0 of every tower launch; reset on read
int orl_debug_prof(unsigned long long* out16) {
  unsigned long long zero[16] = {0};
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_orl_prof), sizeof(zero));
  hipMemcpyToSymbol(HIP_SYMBOL(g_orl_prof), zero, sizeof(zero));
  return 0;
}
#endif

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
  hipStream_t s = (hipStream_t)stream;
  const RawLayout rlp(*pnet);

  // policy tower
  A.net = *pnet; A.theta = ptheta; A.o_x = 0; A.partials = partials;
  const int no = pnet->n_out;
  int gp;
  if (pnet->head_kind == ORL_HEAD_CATEGORICAL) {
    if (no <= 2) gp = launch_tower_nd<ORL_HEAD_CATEGORICAL, 2>(A, s);
    else if (no <= 8) gp = launch_tower_nd<ORL_HEAD_CATEGORICAL, 8>(A, s);
    else gp = launch_tower_nd<ORL_HEAD_CATEGORICAL, 16>(A, s);
  } else {
    if (no <= 8) gp = launch_tower_nd<ORL_HEAD_GAUSSIAN, 8>(A, s);
    else gp = launch_tower_nd<ORL_HEAD_GAUSSIAN, 16>(A, s);
  }
  if (gp <= 0) return gp <= -1000 ? -(gp + 1000) : gp;
  // critic tower
  A.net = *cnet; A.theta = ctheta; A.o_x = o_co;
  A.partials = partials + (size_t)PPO_MAX_BLOCKS * (rlp.total + ORL_N_STATS);
  const int gc = launch_tower_nd<ORL_HEAD_VALUE, 1>(A, s);
  if (gc <= 0) return gc <= -1000 ? -(gc + 1000) : gc;
  if (n_blocks_out) { n_blocks_out[0] = gp; n_blocks_out[1] = gc; }
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
  const size_t lp = (size_t)(RawLayout(*pnet).total + 4 * TowerLayout(*pnet).total) * sizeof(float);  // raw | theta m v g
  const size_t lc = (size_t)(RawLayout(*cnet).total + 4 * TowerLayout(*cnet).total) * sizeof(float);
  const size_t lds = lp > lc ? lp : lc;
  (void)hipFuncSetAttribute((const void*)ppo_apply_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(ppo_apply_kernel, dim3(2), dim3(1024), lds, (hipStream_t)stream, P, Cc, sums, *hp,
                     train_info_accum);
  return launch_status("orl_ppo_apply");
}

int orl_ppo_reduce_pair(const float* partials, int n_blocks_policy, int width_policy, int n_blocks_critic,
                        int width_critic, float* sums, void* stream) {
  ORL_REQUIRE(partials && sums && n_blocks_policy > 0 && n_blocks_critic > 0 && width_policy > 0 && width_critic > 0,
              "orl_ppo_reduce_pair: bad arguments");
  const int gp = (width_policy + 63) / 64, gc = (width_critic + 63) / 64;
  hipLaunchKernelGGL(ppo_reduce_pair_kernel, dim3(gp + gc), dim3(1024), 0, (hipStream_t)stream, partials,
                     n_blocks_policy, width_policy, gp, partials + (size_t)PPO_MAX_BLOCKS * width_policy,
                     n_blocks_critic, width_critic, sums);
  return launch_status("orl_ppo_reduce_pair");
}

}  // extern "C"
