// orl_common.h - shared device/host helpers for the gfx950 kernels (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/orl_hip.h"

#define ORL_WAVE 64

// Round 6: kernels that LOST their A/B against the production path - the streamed bf16-split recurrent row
// kernel (orl_rnn_stream.h), the ticketed one-launch optimiser step (orl_ppo_reduce_apply), the fp32-MFMA and two-image
// variants of the tower pair - are compiled only with -DORL_BUILD_EXPERIMENTS=1 (ORL_BUILD_DEFS of csrc/build.py).  The shipped
// library leaves them out: their entry points / hparams.reserved bits return ORL_E_UNSUPPORTED, orl_build_experiments() says
// which build is loaded, and the tests of those paths skip unless it is the experimental one (run once per round:
// profiles/rNN_pytest_gpu_experiments.log).
// 1 = the MLP towers' 64-wide GEMMs as two-term fp16 splits (orl_mlp.h, round 6), 0 = the three-term bf16 splits of rounds 3 - 5
#ifndef ORL_TOWER_F16
#define ORL_TOWER_F16 1
#endif
#ifndef ORL_BUILD_EXPERIMENTS
#define ORL_BUILD_EXPERIMENTS 0
#endif

namespace orl {

// ---- error reporting -----------------------------------------------------------------------
extern thread_local char g_err[512];
int fail(int code, const char* fmt, ...);
int launch_status(const char* what);  // hipGetLastError -> return code (+message)

#define ORL_REQUIRE(cond, ...)                            \
  do {                                                    \
    if (!(cond)) return ::orl::fail(ORL_E_INVALID, __VA_ARGS__); \
  } while (0)

// ---- parameter layout of one tower (matches the reference state_dict order) -------------------------
struct TowerLayout {
  int D, H, n_out, head;
  int oW1, ob1, og1, obe1, oW2, ob2, og2, obe2, oW3, ob3, ologstd, total;
  __host__ __device__ TowerLayout() {}
  __host__ __device__ explicit TowerLayout(const orl_net_desc& n) {
    D = n.obs_dim; H = n.hidden; n_out = n.n_out; head = n.head_kind;
    int o = 0;
    oW1 = o; o += H * D;
    ob1 = o; o += H;
    og1 = o; o += H;
    obe1 = o; o += H;
    oW2 = o; o += H * H;
    ob2 = o; o += H;
    og2 = o; o += H;
    obe2 = o; o += H;
    oW3 = o; o += n_out * H;
    ob3 = o; o += n_out;
    ologstd = o;
    if (head == ORL_HEAD_GAUSSIAN) o += n_out;
    total = o;
  }
};

// Raw gradient-sum vector of one tower (what the fused backward accumulates; see orl_ppo.hip):
//   G[H*H]   = sum_r dz2[r][o] * xhat1[r][i]          (dW2 = g1[i]*G + be1[i]*db2[o])
//   S3[n_out*H] = sum_r dhead[r][c] * xhat2[r][f]     (dW3 = g2[f]*S3 + be2[f]*db3[c])
//   db3[n_out], db2[H], dW1[H*D], db1[H], dlogstd[n_out]
// The LayerNorm affine gradients need NO accumulator of their own - they are linear images of the above:
//   dn2 = W3^T dhead  =>  dg2[f]  = sum_r dn2[r][f] xhat2[r][f] = sum_c W3[c][f] * S3[c][f],
//                         dbe2[f] = sum_r dn2[r][f]             = sum_c W3[c][f] * db3[c];
//   dn1 = W2^T dz2    =>  dg1[i]  = sum_o W2[o][i] * G[o][i],     dbe1[i] = sum_o W2[o][i] * db2[o].
// orl_ppo_apply evaluates them (fp32 dot products of <= 64 terms) from the reduced sums.
struct RawLayout {
  int oG, oS3, odb3, odb2, odW1, odb1, odlogstd, total;
  __host__ __device__ RawLayout() {}
  __host__ __device__ explicit RawLayout(const orl_net_desc& n) {
    const int H = n.hidden, D = n.obs_dim, K = n.n_out;
    int o = 0;
    oG = o; o += H * H;
    oS3 = o; o += K * H;
    odb3 = o; o += K;
    odb2 = o; o += H;
    odW1 = o; o += H * D;
    odb1 = o; o += H;
    odlogstd = o; o += (n.head_kind == ORL_HEAD_GAUSSIAN ? K : 0);
    total = o;
  }
};

// stats vector slots (sums over the minibatch rows; see orl_ppo.hip)
enum {
  ST_ACTIVE_SUM = 0,   // sum(active)
  ST_ROWS = 1,         // number of rows
  ST_VLOSS_SUM = 2,    // sum(value_loss_row * (active or 1))
  ST_PLOSS_SUM = 3,    // sum(-sum_dim surr * (active or 1))
  ST_ENT_SUM = 4,      // sum(entropy_row * (active or 1))
  ST_RATIO_SUM = 5,    // sum over rows and dims of ratio
  ST_RATIO_CNT = 6,    // rows * dims
};

// ---- Philox4x32-10 (Salmon et al. 2011), counter-based RNG used for action sampling and the
// synthetic env.  The oracle restates it in numpy (oracle/philox.py) so streams are comparable.
struct u4 { uint32_t x, y, z, w; };

__host__ __device__ inline void philox_round(u4& c, uint32_t k0, uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
  const uint64_t p0 = (uint64_t)M0 * c.x;
  const uint64_t p1 = (uint64_t)M1 * c.z;
  const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
  const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
  u4 r;
  r.x = hi1 ^ c.y ^ k0;
  r.y = lo1;
  r.z = hi0 ^ c.w ^ k1;
  r.w = lo0;
  c = r;
}

__host__ __device__ inline u4 philox4x32_10(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3) {
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  u4 c{c0, c1, c2, c3};
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return c;
}

// uniform in [0,1) with 24 bits; uniform in (0,1] for log()
__host__ __device__ inline float u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }
__host__ __device__ inline float u01_open0(uint32_t x) { return ((float)(x >> 8) + 1.0f) * (1.0f / 16777216.0f); }

// ---- wave helpers ------------------------------------------------------------------------------
// Sum over the 4 lanes {l, l^16, l^32, l^48} that share a batch row of a 16-row MFMA tile, result in all
// of them.  gfx950 v_permlane16_swap / v_permlane32_swap are plain VALU ops (no LDS round trip like
// ds_bpermute): swapping a value with a copy of itself and adding the two results is x + x[lane^16]
// (resp. x + x[lane^32]).
__device__ inline float row_allsum(float s) {
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  // Written as inline asm on purpose: with ROCm 7.2's hipcc the __builtin_amdgcn_permlane{16,32}_swap
  // pair-result is mis-lowered inside larger kernels (the add after the swap reads the FIRST result twice:
  // `v_permlane16_swap v12, v13 ; v_add_f32 v12, v12, v12`), which silently turns the sum into 2*x.
  // The swap rewrites both operands in place, hence the two "+v" copies; the s_nop's are the VALU-write ->
  // permlane-read wait states hipcc itself inserts (it pads nothing inside an asm statement).
  float a = s, b = s;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  const float t = a + b;
  a = t;
  b = t;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  return a + b;
}

// Two independent row sums in one go: the swaps of the two values share their wait states and the dependent
// add -> swap -> add chain of one value hides behind the other's (LayerNorm needs its two statistics together).
#ifndef ORL_ROWSUM2_HALVES
#define ORL_ROWSUM2_HALVES 1   // build-time A/B switch: 0 = two independent butterflies (rounds 1 - 4)
#endif
__device__ inline void row_allsum2(float& x, float& y) {
#if ORL_ROWSUM2_HALVES
  // Round 5: the two sums share ONE butterfly.  permlane32_swap(a, b) exchanges a's upper half-wave with b's lower one, so
  // with a = x, b = y the sum a + b holds x[l] + x[l ^ 32] in lanes 0 - 31 and y[l] + y[l ^ 32] in lanes 32 - 63; one 16-lane
  // level on that register finishes both sums, each in its own half-wave, and a last permlane32_swap of the result with a
  // copy of itself spreads the halves: the first operand ends up holding x's total in every lane, the second y's.
  // 3 swaps + 2 adds + 2 copies instead of 4 swaps + 4 adds + 4 copies.
  float a = x, b = y;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  const float t = a + b;
  a = t;
  b = t;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  const float u = a + b;
  a = u;
  b = u;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  x = a;
  y = b;
#else
  float a = x, b = x, c = y, d = y;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\ts_nop 1"
               : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
  const float t = a + b, u = c + d;
  a = t; b = t; c = u; d = u;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\ts_nop 1"
               : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
  x = a + b;
  y = c + d;
#endif
}

// The 4 values of a row (lanes q = 0..3 of the tile row), in every one of its lanes: permlane16_swap of a value with
// a copy of itself leaves the even-q value of each lane pair in one operand and the odd-q value in the other;
// permlane32_swap of each of those with a copy of itself separates the two pairs.
__device__ inline void row_gather4(float s, float (&o)[4]) {
  float a = s, b = s;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  float c = a, d = a, e = b, f = b;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\ts_nop 1"
               : "+v"(c), "+v"(d), "+v"(e), "+v"(f));
  o[0] = c; o[1] = e; o[2] = d; o[3] = f;
}

// max over the same 4 lanes, result in all of them
__device__ inline float row_allmax(float s) {
  float a = s, b = s;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  const float t = fmaxf(a, b);
  a = t;
  b = t;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  return fmaxf(a, b);
}

// two independent row maxima through one butterfly (row_allsum2's scheme; max is exact, so the results are those of two row_allmax)
__device__ inline void row_allmax2(float& x, float& y) {
  float a = x, b = y;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  const float t = fmaxf(a, b);
  a = t;
  b = t;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  const float u = fmaxf(a, b);
  a = u;
  b = u;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  x = a;
  y = b;
}

// x[l] + x[l ^ 32] in every lane (one v_permlane32_swap instead of a ds_bpermute round trip)
__device__ inline float xor32_sum(float x) {
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  return a + b;
}

// sum over the wave, in every lane, on the VALU: four DPP rotations inside the 16-lane rows, then row_allsum's two half-wave
// swaps - wave_sum below takes six ds_bpermute round trips through the LDS queue (a different order of additions: results
// differ in the last bits)
__device__ inline float wave_sum_dpp(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0u, __builtin_bit_cast(unsigned, v), 0x121, 0xf, 0xf, false));  // row_ror:1
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0u, __builtin_bit_cast(unsigned, v), 0x122, 0xf, 0xf, false));  // row_ror:2
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0u, __builtin_bit_cast(unsigned, v), 0x124, 0xf, 0xf, false));  // row_ror:4
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0u, __builtin_bit_cast(unsigned, v), 0x128, 0xf, 0xf, false));  // row_ror:8
  return row_allsum(v);
}

__device__ inline double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}
__device__ inline float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
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

}  // namespace orl
