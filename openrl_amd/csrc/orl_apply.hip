// orl_apply.hip - the optimiser-step side of the PPO update for gfx950 (built with strict IEEE arithmetic: the Adam
// update, the ValueNorm moments and the gradient-norm clip follow torch's fp32 / python-double operation order):
//   orl_ppo_reduce / orl_ppo_reduce_pair : deterministic column sums of the tower kernels' per-workgroup partials
//                                          (the vector a multi-GPU run all-reduces)
//   orl_ppo_apply                        : raw sums -> parameter gradients, grad-norm clip (K13), Adam (K14), train_info
//   orl_ppo_apply_perm                   : the same launch also produces the next epoch's permutation / ValueNorm.update
//   orl_ppo_reduce_pair_comm / orl_ppo_apply_comm : the multi-GPU optimiser step in the SAME two launches - the reduce
//                                          pushes every column sum it produces to the peers' inboxes, the apply sums
//                                          the G contributions in rank order while it stages them (orl_comm.h)
#include <stdlib.h>
#include <string.h>
#include "orl_common.h"
#include "orl_perm.h"
#include "orl_comm.h"
#include "orl_mlp.h"
#undef ORL_PROF  // the phase-timing symbols of the tower header belong to orl_ppo.hip
#include "orl_ppo_tower.h"

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

// n floats HBM -> LDS by DMA (global_load_lds: no registers, nothing to wait for per element): every iteration of the
// rolled register loop this replaces was one serialized HBM round trip (load -> s_waitcnt vmcnt(0) -> ds_write), ~4.5 of
// them per array at configuration 2's towers.  The caller waits vmcnt(0) + barrier once, after its last call.
__device__ inline void stage_dma(float* lds_dst, const float* __restrict__ src, int n) {
  const int lane = threadIdx.x & 63;
#pragma unroll 1
  for (int e0 = (threadIdx.x & ~63); e0 < n; e0 += blockDim.x) {
    if (e0 + lane < n)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + e0 + lane),
                                       (__attribute__((address_space(3))) void*)(lds_dst + e0), 4, 0, 0);
  }
}

__device__ inline float block_sum_1024(float v, float* sh) {
  v = wave_sum_dpp(v);  // (on the VALU: the ds_bpermute butterfly was six LDS round trips on the optimiser step's serial chain)
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
  const int H = HID;
  float g;
  if (p < tl.ob1) g = raw[rl.odW1 + (p - tl.oW1)];
  else if (p < tl.og1) g = raw[rl.odb1 + (p - tl.ob1)];
  else if (p < tl.obe1) {  // LN1 weight: dg1[i] = sum_o W2[o][i] * G[o][i]
    // (both 64-term loops: operands of 8 terms requested together, summed in the same order - the rolled form was 64
    // dependent LDS round trips on one wave while the rest of the workgroup waited at the norm's barrier)
    const int i = p - tl.og1;
    g = 0.f;
#pragma unroll 8
    for (int o = 0; o < H; ++o) g += theta[tl.oW2 + o * H + i] * raw[rl.oG + o * H + i];
  } else if (p < tl.oW2) {  // LN1 bias: dbe1[i] = sum_o W2[o][i] * db2[o]
    const int i = p - tl.obe1;
    g = 0.f;
#pragma unroll 8
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

// The optimiser step of tower t (0 policy, 1 critic) by ONE workgroup of 1024 threads.  The tower's raw sums and
// parameters are staged in LDS first so the 64-term LayerNorm-affine dot products of raw_to_grad run out of LDS.
// wait_ctr (orl_ppo_step, round 6): this workgroup was launched TOGETHER with the workgroups that produce `sums`; it requests
// everything that does not depend on them first (parameters, Adam moments, the bias corrections), then waits until the ticket
// word says that all `wait_target` producers have published, and reads the sums with system-scope loads (the producers stored
// them write-through: orl_ppo_step's protocol below).  Returns false when the bounded wait ran out (nothing is written then
// except NaN into the train_info slots - a loud failure, not a hang).
__device__ inline bool apply_tower_block(const int t, const ApplyTower& P, const ApplyTower& Cc, float* __restrict__ sums,
                                         const orl_ppo_hparams& hp, float* __restrict__ info, const int stage_mv,
                                         const CommDev& CM, const int use_comm, float* s_apply, float* sh,
                                         unsigned* __restrict__ wait_ctr = nullptr, const unsigned wait_target = 0u) {
  const TowerLayout tlp(P.net), tlc(Cc.net);
  const RawLayout rlp(P.net), rlc(Cc.net);
  float norms[2] = {0.f, 0.f};
  // this tower's raw sums + statistics, staged in LDS first.  Multi-GPU (use_comm): every element is the sum over
  // ranks, in rank order, of the local column sum and the peers' pushed granules (orl_ppo_reduce_pair_comm); the
  // reduced vector is also written back to `sums` so the caller observes the global sums.
  float* raw = s_apply;
  const float* st;  // this tower's statistics (ST_*), in LDS
  float den_p = 1.f, den_v = 1.f;
  const bool policy_off = t == 0 && (hp.reserved & 1);
  bool params_staged = false;
  {
    const RawLayout& rl0 = t == 0 ? rlp : rlc;
    const int off = t == 0 ? P.sums_off : Cc.sums_off;
    const int nraw = rl0.total + ORL_N_STATS;
    if (wait_ctr != nullptr) {
      // (1) what does not depend on the producers: parameters and moments on their way into LDS
      if (!policy_off) {
        const ApplyTower& W0 = t == 0 ? P : Cc;
        const TowerLayout& tl0 = t == 0 ? tlp : tlc;
        float* th0 = s_apply + rl0.total + ORL_N_STATS;
        stage_dma(th0, W0.ad.theta, tl0.total);
        if (stage_mv) {
          stage_dma(th0 + 2 * tl0.total, W0.ad.m, tl0.total);
          stage_dma(th0 + 3 * tl0.total, W0.ad.v, tl0.total);
        }
        params_staged = true;
      }
      // (2) the ticket: one lane polls (relaxed agent-scope loads, s_sleep between them), bounded
      __shared__ int s_wait_ok;
      if (threadIdx.x == 0) {
        int ok = 1;
        unsigned spins = 0;
        while (__hip_atomic_load(wait_ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < wait_target) {
          __builtin_amdgcn_s_sleep(2);
          if (++spins > (1u << 22)) { ok = 0; break; }  // ~1 s: the producers of THIS launch never take that long
        }
        s_wait_ok = ok;
        if (ok) __hip_atomic_store(wait_ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // nobody draws again in this launch
      }
      __syncthreads();
      if (!s_wait_ok) {
        if (threadIdx.x == 0 && info != nullptr)
          for (int k = 0; k < 6; ++k) info[k] = __builtin_bit_cast(float, 0x7fc00000u);
        return false;
      }
      // (3) the sums: system-scope loads (the producers' write-through stores are in memory; no L2 line of this XCD is trusted)
#pragma unroll 1
      for (int e = threadIdx.x; e < nraw; e += blockDim.x) {
        float v = __hip_atomic_load(sums + off + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (use_comm) {
          v = comm_sum(CM, off + e, v);
          sums[off + e] = v;
        }
        raw[e] = v;
      }
    } else if (use_comm) {
#pragma unroll 1
      for (int e = threadIdx.x; e < nraw; e += blockDim.x) {
        const float v = comm_sum(CM, off + e, sums[off + e]);
        sums[off + e] = v;
        raw[e] = v;
      }
    } else {
      stage_dma(raw, sums + off, nraw);
    }
    st = raw + rl0.total;
  }
  {
    const ApplyTower& W = t == 0 ? P : Cc;
    const TowerLayout& tl = t == 0 ? tlp : tlc;
    const RawLayout& rl = t == 0 ? rlp : rlc;
    if (policy_off) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMA (stage_dma above) has landed, as on the main path
      __syncthreads();
      den_p = hp.use_policy_active_masks ? st[ST_ACTIVE_SUM] : st[ST_ROWS];
      const float* stp = st;
      // turn_on == False: the policy loss is not in the loss list (ppo.py:226-236) -> no gradient, no step
      for (int p = threadIdx.x; p < tl.total; p += blockDim.x) W.ad.grad[p] = 0.f;
      if (threadIdx.x == 0 && info != nullptr) {
        if (hp.reserved & 32) info[1] = info[2] = info[3] = info[5] = 0.f;  // first step of a train() call: overwrite
        info[1] += stp[ST_PLOSS_SUM] / den_p;
        float ent_den0 = den_p;
        if (!hp.use_policy_active_masks && P.net.head_kind == ORL_HEAD_GAUSSIAN) ent_den0 = den_p * (float)P.net.n_out;
        info[2] += stp[ST_ENT_SUM] / ent_den0;
        const float aw0 = P.net.head_kind == ORL_HEAD_GAUSSIAN ? (float)P.net.n_out : 1.f;
        info[5] += stp[ST_RATIO_SUM] / (stp[ST_ROWS] * aw0);
      }
      return true;
    }
    float* th_s = s_apply + rl.total + ORL_N_STATS;
    // latency plan for a one-shot, two-workgroup kernel: everything that comes from HBM (raw sums, parameters, Adam
    // moments) is requested up front into LDS by DMA, the double-precision bias corrections are evaluated while those
    // loads are in flight, gradients stay in LDS between the norm and the Adam pass, and every loop stays ROLLED - the code
    // is fetched cold on each launch, so instruction bytes cost more than loop overhead (an unrolled variant of this
    // body measured 17.2 us against 13.3 us).
    float* g_s = th_s + tl.total;
    float* m_s = g_s + tl.total;  // Adam moments: staged too when everything fits 160 KiB (stage_mv), else read from HBM
    float* v_s = m_s + tl.total;
    if (!params_staged) {
      stage_dma(th_s, W.ad.theta, tl.total);
      if (stage_mv) {
        stage_dma(m_s, W.ad.m, tl.total);
        stage_dma(v_s, W.ad.v, tl.total);
      }
    }
    // torch.optim.Adam (single tensor math, betas (0.9, 0.999), amsgrad off)
    // scalar coefficients are python doubles in torch/optim/adam.py; only tensor math is fp32
    const double b1d = 0.9, b2d = 0.999;
    const float b2 = (float)b2d, omb1 = (float)(1.0 - b1d), omb2 = (float)(1.0 - b2d);
    const double bc1 = 1.0 - powi_d(b1d, (long long)W.ad.step);
    const double bc2 = 1.0 - powi_d(b2d, (long long)W.ad.step);
    const float step_size = (float)((double)W.ad.lr / bc1);
    const float bc2_sqrt = (float)sqrt(bc2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMA has landed
    __syncthreads();
    den_p = hp.use_policy_active_masks ? st[ST_ACTIVE_SUM] : st[ST_ROWS];
    den_v = hp.use_value_active_masks ? st[ST_ACTIVE_SUM] : st[ST_ROWS];
    const float inv_den = 1.0f / (t == 0 ? den_p : den_v);
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
    // The loop reads LDS (or, for the widest towers, the moments in HBM through their own pointers) and only STORES to
    // HBM: one loop over generic pointers for both cases made every iteration drain the previous iteration's stores
    // (flat loads -> s_waitcnt vmcnt(0) lgkmcnt(0) inside the loop, ~1 us each).
    auto adam = [&](int p, float m, float v) {
      float g = g_s[p] * coef;
      W.ad.grad[p] = g;
      float th = th_s[p];
      if (W.ad.weight_decay != 0.f) g += W.ad.weight_decay * th;
      m = m + (g - m) * omb1;          // exp_avg.lerp_(grad, 1 - beta1)
      v = v * b2 + omb2 * (g * g);     // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
      const float denom = sqrtf(v) / bc2_sqrt + W.ad.eps;
      th = th - step_size * (m / denom);
      W.ad.m[p] = m; W.ad.v[p] = v; W.ad.theta[p] = th;
    };
    if (stage_mv) {
#pragma unroll 1
      for (int p = threadIdx.x; p < tl.total; p += blockDim.x) adam(p, m_s[p], v_s[p]);
    } else {
#pragma unroll 1
      for (int p = threadIdx.x; p < tl.total; p += blockDim.x) adam(p, W.ad.m[p], W.ad.v[p]);
    }
  }
  if (threadIdx.x == 0 && info != nullptr) {
    const float* stp = st;
    const float* stc = st;
    // each workgroup owns its slots (one writer): the running sums are kept with return-less float atomics - a plain
    // `info[k] += x` is a load -> add -> store chain, one more HBM round trip on the tail of a latency-bound launch.
    // hp.reserved & 32: the first optimiser step of a train() call STORES (the caller then needs no zero-fill launch).
    const bool first = (hp.reserved & 32) != 0;
    auto acc = [&](int k, float x) {
      if (first) info[k] = x;
      else unsafeAtomicAdd(info + k, x);
    };
    if (t == 0) {
      float ent_den = den_p;
      if (!hp.use_policy_active_masks && P.net.head_kind == ORL_HEAD_GAUSSIAN) ent_den = den_p * (float)P.net.n_out;
      acc(1, stp[ST_PLOSS_SUM] / den_p);                 // policy_loss
      acc(2, stp[ST_ENT_SUM] / ent_den);                 // dist_entropy
      acc(3, norms[0]);                                  // actor_grad_norm
      const float a_w = P.net.head_kind == ORL_HEAD_GAUSSIAN ? (float)P.net.n_out : 1.f;
      acc(5, stp[ST_RATIO_SUM] / (stp[ST_ROWS] * a_w));  // ratio.mean()
    } else {
      acc(0, stc[ST_VLOSS_SUM] / den_v);                 // value_loss
      acc(4, norms[1]);                                  // critic_grad_norm
    }
  }
  return true;
}

// One workgroup per tower (blockIdx.x = 0 policy, 1 critic).  Workgroups >= 2 (orl_ppo_apply_perm only) produce the NEXT
// epoch's minibatch permutation and ValueNorm.update in the same launch: both are independent of this optimiser step, and
// the two apply workgroups leave 254 CUs idle.
__global__ __launch_bounds__(1024) void ppo_apply_kernel(ApplyTower P, ApplyTower Cc, float* __restrict__ sums,
                                                         orl_ppo_hparams hp, float* __restrict__ info, PermJob J,
                                                         int stage_mv, CommDev CM, int use_comm) {
  extern __shared__ __attribute__((aligned(16))) float s_apply[];
  if (blockIdx.x >= 2) {
    perm_job_block(J, (int)blockIdx.x - 2, (int)gridDim.x - 2);
    return;
  }
  __shared__ float sh[16];
  apply_tower_block((int)blockIdx.x, P, Cc, sums, hp, info, stage_mv, CM, use_comm, s_apply, sh);
}

// 64 columns x 16 row groups of one tower's partial region by a workgroup of 1024 threads: at 256 partial rows every
// thread has 16 independent loads in flight.  `sh` = 16 x 64 floats of LDS.  Multi-GPU: the column sum also goes
// straight to every peer's inbox (one 8-byte granule each).
// PUBLISH (orl_ppo_step): the column sums are stored WRITE-THROUGH with system scope instead of plain - they are read in the
// same launch by a workgroup on another CU / XCD, which no kernel boundary orders behind these stores
template <bool PUBLISH = false>
__device__ inline void reduce_columns_block(const float* __restrict__ partials, int n_blocks, int width, int col0,
                                            float* __restrict__ out, int comm_off, const CommDev& CM, int use_comm,
                                            float (*sh)[64]) {
  const int lc = threadIdx.x & 63;
  const int col = col0 + lc;
  const int rg = threadIdx.x >> 6;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (col < width) {
    if (n_blocks <= 256) {
      // the usual case (<= PPO_MAX_BLOCKS partial rows = <= 16 per thread): every load of the thread in ONE batch, then the
      // additions of the loop below in its order (rows rg + 64 k + {0, 16, 32, 48} into s0..s3, the remainder into s0) -
      // the loop form is one HBM round trip per 4 rows
      float v[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const int b = rg + 16 * k;
        v[k] = b < n_blocks ? partials[(size_t)b * width + col] : 0.f;
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (rg + 64 * g + 48 < n_blocks) {  // a complete group of four rows (these form a prefix of the groups)
          s0 += v[4 * g]; s1 += v[4 * g + 1]; s2 += v[4 * g + 2]; s3 += v[4 * g + 3];
        } else {                            // the remainder (all rows of any later group are >= n_blocks)
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (rg + 64 * g + 16 * k < n_blocks) s0 += v[4 * g + k];
        }
      }
    } else {
      int b = rg;
      for (; b + 48 < n_blocks; b += 64) {
        s0 += partials[(size_t)b * width + col];
        s1 += partials[(size_t)(b + 16) * width + col];
        s2 += partials[(size_t)(b + 32) * width + col];
        s3 += partials[(size_t)(b + 48) * width + col];
      }
      for (; b < n_blocks; b += 16) s0 += partials[(size_t)b * width + col];
    }
  }
  sh[rg][lc] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (rg == 0 && col < width) {
    float t[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) t[k] = (sh[4 * k][lc] + sh[4 * k + 1][lc]) + (sh[4 * k + 2][lc] + sh[4 * k + 3][lc]);
    const float v = (t[0] + t[1]) + (t[2] + t[3]);
    if (PUBLISH) __hip_atomic_store(out + col, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else out[col] = v;
    if (use_comm) {
      for (int p = 0; p < CM.world; ++p)
        if (p != CM.rank) comm_push(CM, p, comm_off + col, v);
    }
  }
}

#if ORL_BUILD_EXPERIMENTS
// Reduce + optimiser step in ONE launch (orl_ppo_reduce_apply): workgroups [0, gp) / [gp, gp + gc) sum 64 columns each
// of the policy / critic partial region, then take a ticket; the workgroup that draws a tower's LAST ticket has every
// column sum of that tower behind a device-scope fence and runs the tower's optimiser step (apply_tower_block).  Every
// other workgroup - and the extra ones behind gp + gc - draws one share of the next epoch's permutation job.  One
// kernel boundary (~2.4 us of dependent-dispatch latency on this stack) less than orl_ppo_reduce_pair +
// orl_ppo_apply_perm - and measured SLOWER: 15.7 us against 4.4 + 9.3 us, the fences and the ticket's round trip to
// memory cost more than the boundary (cfg.amd_optim_step keeps two launches as the default).  ctr[0..2]: tickets of the two towers
// and of the permutation shares; each is reset by the workgroup that draws its last value, so a zero-initialised array
// serves every launch on the stream.
__global__ __launch_bounds__(1024) void ppo_reduce_apply_kernel(const float* __restrict__ pp, int nb_p, int wp, int gp,
                                                                const float* __restrict__ pc, int nb_c, int wc, int gc,
                                                                ApplyTower P, ApplyTower Cc, float* __restrict__ sums,
                                                                orl_ppo_hparams hp, float* __restrict__ info, PermJob J,
                                                                int stage_mv, CommDev CM, int use_comm,
                                                                unsigned* __restrict__ ctr) {
  extern __shared__ __attribute__((aligned(16))) float s_apply[];
  __shared__ float sh[16];
  __shared__ unsigned s_ticket;
  const int b = blockIdx.x;
  if (b < gp + gc) {
    const bool pol = b < gp;
    // the reduce's 4 KB of scratch are the head of the apply's staging area (used strictly before it)
    reduce_columns_block(pol ? pp : pc, pol ? nb_p : nb_c, pol ? wp : wc, (pol ? b : b - gp) * 64, pol ? sums : sums + wp,
                         pol ? 0 : wp, CM, use_comm, (float(*)[64])s_apply);
    // Wave 0 wrote the 64 column sums: ITS release fence (stores retired, written back so that the other XCDs' L2s can
    // see them) precedes the ticket.  One fence per workgroup - a __threadfence() by all 16 waves of all 146 workgroups was
    // 2 336 L2 write-back + invalidate sequences and cost 35 us per launch.
    if (threadIdx.x < 64) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      if (threadIdx.x == 0) s_ticket = atomicAdd(ctr + (pol ? 0 : 1), 1u);
    }
    __syncthreads();
    if (s_ticket == (unsigned)((pol ? gp : gc) - 1)) {
      if (threadIdx.x == 0) ctr[pol ? 0 : 1] = 0u;  // nobody draws from it again in this launch
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // the other workgroups' column sums, not a stale line
      apply_tower_block(pol ? 0 : 1, P, Cc, sums, hp, info, stage_mv, CM, use_comm, s_apply, sh);
      return;
    }
  }
  if (J.idx == nullptr) return;
  const unsigned shares = gridDim.x - 2u;  // every workgroup but the two that ran an optimiser step
  __syncthreads();
  if (threadIdx.x == 0) s_ticket = atomicAdd(ctr + 2, 1u);
  __syncthreads();
  const unsigned k = s_ticket;
  if (k == shares - 1u && threadIdx.x == 0) ctr[2] = 0u;
  perm_job_block(J, (int)k, (int)shares);
}

#endif

// The optimiser step in ONE launch (orl_ppo_step, round 6).  Workgroups [0, gp) / [gp, gp + gc) sum 64
// columns each of the policy / critic partial region exactly as ppo_reduce_pair does (bit-identical sums) and publish them:
// write-through stores (system scope), s_waitcnt vmcnt(0), one agent-scope ticket.  Workgroups gp + gc and gp + gc + 1 are the
// two optimiser workgroups - DESIGNATED, not "the last one to arrive" as in round 5's orl_ppo_reduce_apply: they are launched
// with the producers, request parameters / moments / bias corrections while the producers work, poll their tower's ticket word
// and then read the sums with system-scope loads.  No L2 write-back / invalidate fence anywhere (they were what made the
// ticketed form slower than two launches: 15.7 us against 4.4 + 9.3), and no co-residency assumption: only the two optimiser
// workgroups ever wait, every producer runs to completion whatever is resident.  Workgroups behind them draw the next epoch's
// permutation job.  ctr[0..1]: the towers' ticket words, zero before the first call; each is reset by its optimiser workgroup.
__global__ __launch_bounds__(1024) void ppo_step_kernel(const float* __restrict__ pp, int nb_p, int wp, int gp,
                                                        const float* __restrict__ pc, int nb_c, int wc, int gc, ApplyTower P,
                                                        ApplyTower Cc, float* __restrict__ sums, orl_ppo_hparams hp,
                                                        float* __restrict__ info, PermJob J, int stage_mv, CommDev CM,
                                                        int use_comm, unsigned* __restrict__ ctr) {
  extern __shared__ __attribute__((aligned(16))) float s_apply[];
  __shared__ float sh[16];
  const int b = blockIdx.x;
  if (b < gp + gc) {
    const bool pol = b < gp;
    reduce_columns_block<true>(pol ? pp : pc, pol ? nb_p : nb_c, pol ? wp : wc, (pol ? b : b - gp) * 64, pol ? sums : sums + wp,
                               pol ? 0 : wp, CM, use_comm, (float(*)[64])s_apply);
    if (threadIdx.x < 64) {  // wave 0 wrote the 64 column sums: its stores have left the CU before the ticket is drawn
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (threadIdx.x == 0) __hip_atomic_fetch_add(ctr + (pol ? 0 : 1), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return;
  }
  if (b < gp + gc + 2) {
    const int t = b - gp - gc;
    apply_tower_block(t, P, Cc, sums, hp, info, stage_mv, CM, use_comm, s_apply, sh, ctr + t, (unsigned)(t == 0 ? gp : gc));
    return;
  }
  perm_job_block(J, b - (gp + gc + 2), (int)gridDim.x - (gp + gc + 2));
}

// both towers' partial regions in one launch: blocks [0, gp) reduce the policy region, the rest the critic's
__global__ __launch_bounds__(1024) void ppo_reduce_pair_kernel(const float* __restrict__ pp, int nb_p, int wp, int gp,
                                                               const float* __restrict__ pc, int nb_c, int wc,
                                                               float* __restrict__ sums, CommDev CM, int use_comm) {
  __shared__ float sh[16][64];
  const bool pol = (int)blockIdx.x < gp;
  reduce_columns_block(pol ? pp : pc, pol ? nb_p : nb_c, pol ? wp : wc, (pol ? blockIdx.x : blockIdx.x - gp) * 64,
                       pol ? sums : sums + wp, pol ? 0 : wp, CM, use_comm, sh);
}

static int check_tower(const orl_net_desc* n, const char* who) {
  if (!n) return fail(ORL_E_INVALID, "%s: null net descriptor", who);
  if (n->hidden != HID) return fail(ORL_E_UNSUPPORTED, "%s: hidden_size %d not built (only 64)", who, n->hidden);
  if (n->obs_dim < 1 || n->obs_dim > 64) return fail(ORL_E_UNSUPPORTED, "%s: obs_dim %d outside [1,64]", who, n->obs_dim);
  if (n->n_out < 1 || n->n_out > 16) return fail(ORL_E_UNSUPPORTED, "%s: n_out %d outside [1,16]", who, n->n_out);
  return 0;
}


}  // namespace orl

using namespace orl;

extern "C" {

int orl_ppo_reduce(const float* partials, int n_blocks, int width, float* sums, void* stream) {
  ORL_REQUIRE(partials && sums && n_blocks > 0 && width > 0, "orl_ppo_reduce: bad arguments");
  const int grid = (width + 63) / 64;
  hipLaunchKernelGGL(ppo_reduce_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, partials, n_blocks, width, sums);
  return launch_status("orl_ppo_reduce");
}

static int launch_apply(const char* what, const orl_net_desc* pnet, const orl_net_desc* cnet, float* sums,
                        const orl_ppo_hparams* hp, const orl_adam_state* padam, const orl_adam_state* cadam,
                        float* train_info_accum, const PermJob& J, void* stream, orl_comm* comm = nullptr) {
  int rc = check_tower(pnet, what);
  if (rc) return rc;
  rc = check_tower(cnet, what);
  if (rc) return rc;
  ORL_REQUIRE(sums && hp && padam && cadam, "%s: null pointer", what);
  ORL_REQUIRE(padam->theta && padam->grad && padam->m && padam->v && cadam->theta && cadam->grad && cadam->m && cadam->v,
              "%s: null optimizer buffer", what);
  ORL_REQUIRE(padam->step >= 1 && cadam->step >= 1, "%s: Adam step counts are 1-based", what);
  ApplyTower P, Cc;
  P.net = *pnet; P.ad = *padam; P.sums_off = 0;
  Cc.net = *cnet; Cc.ad = *cadam; Cc.sums_off = RawLayout(*pnet).total + ORL_N_STATS;
  // raw | theta g [| m v]: the Adam moments are staged with the rest when that fits 160 KiB (every tower up to obs ~40),
  // the widest towers (obs 64 x 16 outputs: 191 KB) read them from HBM in the Adam pass
  auto need = [&](int k) {
    const size_t lp = (size_t)(RawLayout(*pnet).total + ORL_N_STATS + k * TowerLayout(*pnet).total) * sizeof(float);
    const size_t lc = (size_t)(RawLayout(*cnet).total + ORL_N_STATS + k * TowerLayout(*cnet).total) * sizeof(float);
    return lp > lc ? lp : lc;
  };
  const int stage_mv = need(4) <= 160 * 1024;
  const size_t lds = need(stage_mv ? 4 : 2);
  ORL_REQUIRE(lds <= 160 * 1024, "%s: tower needs %zu B of LDS", what, lds);
  int perm_blocks = 0;
  if (J.idx != nullptr) {
    perm_blocks = (int)((J.n + 1023) / 1024);
    if (perm_blocks > 254) perm_blocks = 254;  // one workgroup per CU next to the two apply workgroups (their LDS footprint is per launch)
  }
  CommDev CM;
  memset(&CM, 0, sizeof(CM));
  int use_comm = 0;
  if (comm != nullptr) {  // the push half ran in orl_ppo_reduce_pair_comm under the SAME sequence number
    const int total = Cc.sums_off + RawLayout(*cnet).total + ORL_N_STATS;
    ORL_REQUIRE(total <= orl_comm_capacity(comm), "%s: the sums vector (%d floats) exceeds the comm's capacity %d", what,
                total, orl_comm_capacity(comm));
    rc = orl_comm_current(comm, &CM);
    if (rc) return rc;
    use_comm = CM.world > 1;
  }
  (void)hipFuncSetAttribute((const void*)ppo_apply_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(ppo_apply_kernel, dim3(2 + perm_blocks), dim3(1024), lds, (hipStream_t)stream, P, Cc, sums, *hp,
                     train_info_accum, J, stage_mv, CM, use_comm);
  return launch_status(what);
}

int orl_ppo_apply(const orl_net_desc* pnet, const orl_net_desc* cnet, const float* sums, const orl_ppo_hparams* hp,
                  const orl_adam_state* padam, const orl_adam_state* cadam, float* train_info_accum, void* stream) {
  PermJob J;
  memset(&J, 0, sizeof(J));
  return launch_apply("orl_ppo_apply", pnet, cnet, (float*)sums, hp, padam, cadam, train_info_accum, J, stream);
}

int orl_ppo_apply_comm(orl_comm* comm, const orl_net_desc* pnet, const orl_net_desc* cnet, float* sums,
                       const orl_ppo_hparams* hp, const orl_adam_state* padam, const orl_adam_state* cadam,
                       float* train_info_accum, int64_t* next_idx, int64_t n, uint64_t seed, uint64_t stream_id,
                       float* vn_state, const double* moments, double beta, void* stream) {
  ORL_REQUIRE(comm, "orl_ppo_apply_comm: null comm");
  PermJob J;
  memset(&J, 0, sizeof(J));
  if (next_idx != nullptr) {
    ORL_REQUIRE(n > 0 && n <= ((int64_t)1 << 62), "orl_ppo_apply_comm: bad permutation arguments");
    ORL_REQUIRE(!vn_state || moments, "orl_ppo_apply_comm: vn_state needs moments");
    J = make_perm_job(next_idx, n, seed, stream_id, vn_state, moments, beta);
  }
  return launch_apply("orl_ppo_apply_comm", pnet, cnet, sums, hp, padam, cadam, train_info_accum, J, stream, comm);
}

int orl_ppo_apply_perm(const orl_net_desc* pnet, const orl_net_desc* cnet, const float* sums, const orl_ppo_hparams* hp,
                       const orl_adam_state* padam, const orl_adam_state* cadam, float* train_info_accum,
                       int64_t* next_idx, int64_t n, uint64_t seed, uint64_t stream_id, float* vn_state,
                       const double* moments, double beta, void* stream) {
  ORL_REQUIRE(next_idx && n > 0 && n <= ((int64_t)1 << 62), "orl_ppo_apply_perm: bad permutation arguments");
  ORL_REQUIRE(!vn_state || moments, "orl_ppo_apply_perm: vn_state needs moments");
  const PermJob J = make_perm_job(next_idx, n, seed, stream_id, vn_state, moments, beta);
  return launch_apply("orl_ppo_apply_perm", pnet, cnet, (float*)sums, hp, padam, cadam, train_info_accum, J, stream);
}

int orl_ppo_reduce_apply(orl_comm* comm, const float* partials, int n_blocks_policy, int width_policy,
                         int n_blocks_critic, int width_critic, float* sums, const orl_net_desc* pnet,
                         const orl_net_desc* cnet, const orl_ppo_hparams* hp, const orl_adam_state* padam,
                         const orl_adam_state* cadam, float* train_info_accum, int64_t* next_idx, int64_t n, uint64_t seed,
                         uint64_t stream_id, float* vn_state, const double* moments, double beta, uint32_t* sync_ctr,
                         void* stream) {
  const char* what = "orl_ppo_reduce_apply";
#if !ORL_BUILD_EXPERIMENTS
  return fail(ORL_E_UNSUPPORTED, "%s: the ticketed one-launch optimiser step lost its A/B (15.7 us against 4.4 + 9.3 us for "
              "orl_ppo_reduce_pair + orl_ppo_apply) and is only present in an ORL_BUILD_EXPERIMENTS library", what);
#else
  int rc = check_tower(pnet, what);
  if (rc) return rc;
  rc = check_tower(cnet, what);
  if (rc) return rc;
  ORL_REQUIRE(partials && sums && hp && padam && cadam && sync_ctr, "%s: null pointer", what);
  ORL_REQUIRE(n_blocks_policy > 0 && n_blocks_critic > 0, "%s: bad arguments", what);
  ORL_REQUIRE(padam->theta && padam->grad && padam->m && padam->v && cadam->theta && cadam->grad && cadam->m && cadam->v,
              "%s: null optimizer buffer", what);
  ORL_REQUIRE(padam->step >= 1 && cadam->step >= 1, "%s: Adam step counts are 1-based", what);
  ApplyTower P, Cc;
  P.net = *pnet; P.ad = *padam; P.sums_off = 0;
  Cc.net = *cnet; Cc.ad = *cadam; Cc.sums_off = RawLayout(*pnet).total + ORL_N_STATS;
  ORL_REQUIRE(width_policy == Cc.sums_off && width_critic == RawLayout(*cnet).total + ORL_N_STATS,
              "%s: partial widths %d / %d do not match the towers' raw vectors (%d / %d)", what, width_policy, width_critic,
              Cc.sums_off, RawLayout(*cnet).total + ORL_N_STATS);
  PermJob J;
  memset(&J, 0, sizeof(J));
  if (next_idx != nullptr) {
    ORL_REQUIRE(n > 0 && n <= ((int64_t)1 << 62), "%s: bad permutation arguments", what);
    ORL_REQUIRE(!vn_state || moments, "%s: vn_state needs moments", what);
    J = make_perm_job(next_idx, n, seed, stream_id, vn_state, moments, beta);
  }
  auto need = [&](int k) {  // launch_apply's LDS plan; the reduce's 4 KB of scratch fit its head
    const size_t lp = (size_t)(RawLayout(*pnet).total + ORL_N_STATS + k * TowerLayout(*pnet).total) * sizeof(float);
    const size_t lc = (size_t)(RawLayout(*cnet).total + ORL_N_STATS + k * TowerLayout(*cnet).total) * sizeof(float);
    return lp > lc ? lp : lc;
  };
  const int stage_mv = need(4) <= 156 * 1024;
  const size_t lds = need(stage_mv ? 4 : 2);
  ORL_REQUIRE(lds <= 156 * 1024 && lds >= 16 * 64 * sizeof(float), "%s: tower needs %zu B of LDS", what, lds);
  const int gp = (width_policy + 63) / 64, gc = (width_critic + 63) / 64;
  int extra = 0;
  if (J.idx != nullptr) {  // one workgroup per CU (the launch's LDS footprint): fill the chip, never less than one share
    extra = (int)((J.n + 1023) / 1024) - (gp + gc - 2);
    if (extra > 256 - gp - gc) extra = 256 - gp - gc;
    if (extra < (gp + gc == 2 ? 1 : 0)) extra = gp + gc == 2 ? 1 : 0;
  }
  CommDev CM;
  memset(&CM, 0, sizeof(CM));
  int use_comm = 0;
  if (comm != nullptr) {
    ORL_REQUIRE(width_policy + width_critic <= orl_comm_capacity(comm), "%s: %d floats exceed the comm's capacity %d", what,
                width_policy + width_critic, orl_comm_capacity(comm));
    rc = orl_comm_next(comm, &CM);  // one collective: pushed by the reducing workgroups, summed by the optimiser step's
    if (rc) return rc;
    use_comm = CM.world > 1;
  }
  (void)hipFuncSetAttribute((const void*)ppo_reduce_apply_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(ppo_reduce_apply_kernel, dim3(gp + gc + extra), dim3(1024), lds, (hipStream_t)stream, partials,
                     n_blocks_policy, width_policy, gp, partials + (size_t)PPO_MAX_BLOCKS * width_policy, n_blocks_critic,
                     width_critic, gc, P, Cc, sums, *hp, train_info_accum, J, stage_mv, CM, use_comm, sync_ctr);
  return launch_status(what);
#endif
}

int orl_ppo_step(orl_comm* comm, const float* partials, int n_blocks_policy, int width_policy, int n_blocks_critic,
                 int width_critic, float* sums, const orl_net_desc* pnet, const orl_net_desc* cnet,
                 const orl_ppo_hparams* hp, const orl_adam_state* padam, const orl_adam_state* cadam,
                 float* train_info_accum, int64_t* next_idx, int64_t n, uint64_t seed, uint64_t stream_id, float* vn_state,
                 const double* moments, double beta, uint32_t* sync_ctr, void* stream) {
  const char* what = "orl_ppo_step";
  int rc = check_tower(pnet, what);
  if (rc) return rc;
  rc = check_tower(cnet, what);
  if (rc) return rc;
  ORL_REQUIRE(partials && sums && hp && padam && cadam && sync_ctr, "%s: null pointer", what);
  ORL_REQUIRE(n_blocks_policy > 0 && n_blocks_critic > 0, "%s: bad arguments", what);
  ORL_REQUIRE(padam->theta && padam->grad && padam->m && padam->v && cadam->theta && cadam->grad && cadam->m && cadam->v,
              "%s: null optimizer buffer", what);
  ORL_REQUIRE(padam->step >= 1 && cadam->step >= 1, "%s: Adam step counts are 1-based", what);
  ApplyTower P, Cc;
  P.net = *pnet; P.ad = *padam; P.sums_off = 0;
  Cc.net = *cnet; Cc.ad = *cadam; Cc.sums_off = RawLayout(*pnet).total + ORL_N_STATS;
  ORL_REQUIRE(width_policy == Cc.sums_off && width_critic == RawLayout(*cnet).total + ORL_N_STATS,
              "%s: partial widths %d / %d do not match the towers' raw vectors (%d / %d)", what, width_policy, width_critic,
              Cc.sums_off, RawLayout(*cnet).total + ORL_N_STATS);
  PermJob J;
  memset(&J, 0, sizeof(J));
  if (next_idx != nullptr) {
    ORL_REQUIRE(n > 0 && n <= ((int64_t)1 << 62), "%s: bad permutation arguments", what);
    ORL_REQUIRE(!vn_state || moments, "%s: vn_state needs moments", what);
    J = make_perm_job(next_idx, n, seed, stream_id, vn_state, moments, beta);
  }
  auto need = [&](int k) {  // launch_apply's LDS plan; the reduce workgroups' 4 KB of scratch fit its head
    const size_t lp = (size_t)(RawLayout(*pnet).total + ORL_N_STATS + k * TowerLayout(*pnet).total) * sizeof(float);
    const size_t lc = (size_t)(RawLayout(*cnet).total + ORL_N_STATS + k * TowerLayout(*cnet).total) * sizeof(float);
    return lp > lc ? lp : lc;
  };
  const int stage_mv = need(4) <= 160 * 1024;
  const size_t lds = need(stage_mv ? 4 : 2);
  ORL_REQUIRE(lds <= 160 * 1024 && lds >= 16 * 64 * sizeof(float), "%s: tower needs %zu B of LDS", what, lds);
  const int gp = (width_policy + 63) / 64, gc = (width_critic + 63) / 64;
  int perm_blocks = 0;
  if (J.idx != nullptr) {
    perm_blocks = (int)((J.n + 1023) / 1024);
    const int room = 254 - (gp + gc) > 8 ? 254 - (gp + gc) : 8;  // one workgroup per CU (the launch's LDS footprint)
    if (perm_blocks > room) perm_blocks = room;
  }
  CommDev CM;
  memset(&CM, 0, sizeof(CM));
  int use_comm = 0;
  if (comm != nullptr) {
    ORL_REQUIRE(width_policy + width_critic <= orl_comm_capacity(comm), "%s: %d floats exceed the comm's capacity %d", what,
                width_policy + width_critic, orl_comm_capacity(comm));
    rc = orl_comm_next(comm, &CM);  // one collective: pushed by the reducing workgroups, summed by the optimiser workgroups
    if (rc) return rc;
    use_comm = CM.world > 1;
  }
  (void)hipFuncSetAttribute((const void*)ppo_step_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(ppo_step_kernel, dim3(gp + gc + 2 + perm_blocks), dim3(1024), lds, (hipStream_t)stream, partials,
                     n_blocks_policy, width_policy, gp, partials + (size_t)PPO_MAX_BLOCKS * width_policy, n_blocks_critic,
                     width_critic, gc, P, Cc, sums, *hp, train_info_accum, J, stage_mv, CM, use_comm, sync_ctr);
  return launch_status(what);
}

static int launch_reduce_pair(const char* what, orl_comm* comm, const float* partials, int n_blocks_policy,
                              int width_policy, int n_blocks_critic, int width_critic, float* sums, void* stream) {
  ORL_REQUIRE(partials && sums && n_blocks_policy > 0 && n_blocks_critic > 0 && width_policy > 0 && width_critic > 0,
              "%s: bad arguments", what);
  CommDev CM;
  memset(&CM, 0, sizeof(CM));
  int use_comm = 0;
  if (comm != nullptr) {
    ORL_REQUIRE(width_policy + width_critic <= orl_comm_capacity(comm), "%s: %d floats exceed the comm's capacity %d",
                what, width_policy + width_critic, orl_comm_capacity(comm));
    const int rc = orl_comm_next(comm, &CM);  // a new collective: its poll half is orl_ppo_apply_comm
    if (rc) return rc;
    use_comm = CM.world > 1;
  }
  const int gp = (width_policy + 63) / 64, gc = (width_critic + 63) / 64;
  hipLaunchKernelGGL(ppo_reduce_pair_kernel, dim3(gp + gc), dim3(1024), 0, (hipStream_t)stream, partials,
                     n_blocks_policy, width_policy, gp, partials + (size_t)PPO_MAX_BLOCKS * width_policy,
                     n_blocks_critic, width_critic, sums, CM, use_comm);
  return launch_status(what);
}

int orl_ppo_reduce_pair(const float* partials, int n_blocks_policy, int width_policy, int n_blocks_critic,
                        int width_critic, float* sums, void* stream) {
  return launch_reduce_pair("orl_ppo_reduce_pair", nullptr, partials, n_blocks_policy, width_policy, n_blocks_critic,
                            width_critic, sums, stream);
}

int orl_ppo_reduce_pair_comm(orl_comm* comm, const float* partials, int n_blocks_policy, int width_policy,
                             int n_blocks_critic, int width_critic, float* sums, void* stream) {
  ORL_REQUIRE(comm, "orl_ppo_reduce_pair_comm: null comm");
  return launch_reduce_pair("orl_ppo_reduce_pair_comm", comm, partials, n_blocks_policy, width_policy, n_blocks_critic,
                            width_critic, sums, stream);
}

}  // extern "C"
