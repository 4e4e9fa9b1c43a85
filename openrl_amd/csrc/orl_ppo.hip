// orl_ppo.hip - PPO update side of the hot path for gfx950:
//   orl_ppo_fwd_bwd : K9-K12 fused tower forward + PPO/value/entropy loss + full backward, one launch
//                     per tower (kernel in orl_ppo_tower.h), producing per-workgroup partial sums of the
//                     RAW gradient
// (the reduction of those partials and the optimiser step live in orl_apply.hip, built WITHOUT -ffast-math)
#include "orl_common.h"
#include "orl_mlp.h"
#include "orl_ppo_tower.h"

// build-time A/B switch (ORL_BUILD_DEFS="-DORL_TOWER_FIRST_VARIANT=1" = never a full-split build); never read at run time
#ifndef ORL_TOWER_FIRST_VARIANT
#define ORL_TOWER_FIRST_VARIANT -1
#endif
#ifndef ORL_TOWER_TR       // wide-observation pair builds (ND >= 1): try the transposing-read full split first
#define ORL_TOWER_TR 1
#endif
#ifndef ORL_ND_REM         // build-time experiment: remainder columns (D - 16 ND, up to this many) of dW1 on the VALU; 0 = a
#define ORL_ND_REM 4       // whole extra 16-column MFMA block instead (D = 17..20 take the ND = 2 build)
#endif
// cost of a wide-head policy tile relative to a critic tile (the side-by-side CU split follows it).  Round 2 measured 1.24
// on the fp32-GEMM builds; with the full split under the wide towers the head's fp32 MFMAs weigh less - re-measured at the
// cfg3 shape (pair launch, two alternations on one box): 1.24 -> 155.9 us, 1.15 -> 154.8, 1.10 -> 154.1
#ifndef ORL_PAIR_WP_WIDE
#define ORL_PAIR_WP_WIDE 1.10
#endif
// ... and with a wide GAUSSIAN head, re-measured after the fp16 split (tools/r06_calls/r06_call33.sh, cfg3's shape: 1.10 -> 134 + 122
// workgroups, 12 / 14 tiles per wave, pair launch 0.1165 - 0.1186 ms; 1.30 .. 1.50 -> 146 + 110, 11 / 15 tiles, 0.1077 - 0.1095 ms)
#ifndef ORL_PAIR_WP_WIDE_GAUSS
#define ORL_PAIR_WP_WIDE_GAUSS 1.30
#endif
#ifndef ORL_TOWER_TR_ND0   // build-time experiment: the same for the small-observation build (it fits both images anyway)
#define ORL_TOWER_TR_ND0 0
#endif
#ifndef ORL_TOWER_WGRAD_SPLIT  // build-time experiment: 0 = the fallback variants keep the fp32 wgrad
#define ORL_TOWER_WGRAD_SPLIT 1
#endif
#ifndef ORL_PAIR_WAVES  // build-time experiment: waves per workgroup of the pair launch (8 = two per SIMD)
#define ORL_PAIR_WAVES 8
#endif

namespace orl {

static int check_tower(const orl_net_desc* n, const char* who) {
  if (!n) return fail(ORL_E_INVALID, "%s: null net descriptor", who);
  if (n->hidden != HID) return fail(ORL_E_UNSUPPORTED, "%s: hidden_size %d not built (only 64)", who, n->hidden);
  if (n->obs_dim < 1 || n->obs_dim > 64) return fail(ORL_E_UNSUPPORTED, "%s: obs_dim %d outside [1,64]", who, n->obs_dim);
  if (n->n_out < 1 || n->n_out > 16) return fail(ORL_E_UNSUPPORTED, "%s: n_out %d outside [1,16]", who, n->n_out);
  return 0;
}

template <int HEAD, int NO, int ND>
static int launch_tower_w(const PpoArgs& A, int waves, size_t lds, hipStream_t s) {
  const int n_tiles = (A.mb + TILE_B - 1) / TILE_B;
  int grid = (n_tiles + waves - 1) / waves;
  if (grid > PPO_MAX_BLOCKS) grid = PPO_MAX_BLOCKS;
  (void)hipFuncSetAttribute((const void*)ppo_tower_kernel<HEAD, NO, ND>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds);
  hipLaunchKernelGGL((ppo_tower_kernel<HEAD, NO, ND>), dim3(grid), dim3(waves * 64), lds, s, A);
  return grid;
}

// Launch one tower; returns the number of workgroups (> 0) or a negative error code.
template <int HEAD, int NO, int ND>
static int launch_tower(const PpoArgs& A, hipStream_t s) {
  constexpr int NOP = NO > 4 ? 16 : ((NO + 3) & ~3);  // wide heads: 16-wide dhead tile + W3 MFMA image
  // as many waves per workgroup (8, 6, 4, 2) as fit the 160 KiB of LDS next to the tower's weights; the kernel is
  // built for 2 waves per SIMD (3 per SIMD at <= 168 VGPRs spills around the 64 wgrad accumulators and a producer /
  // consumer split of the wgrad were both measured slower: DESIGN.md section 6)
  static const int kWaves[4] = {8, 6, 4, 2};
  for (int k = 0; k < 4; ++k) {
    const int waves = kWaves[k];
#ifdef ORL_TOWER_MAXWAVES  // build-time experiment switch (occupancy A/B), never defined in the shipped build
    if (waves > ORL_TOWER_MAXWAVES) continue;
#endif
    // the transposed W2 copy (17 KB, +2 % on the dgrad GEMM) is the first thing to go when it would cost a pair of waves
    for (int w2t = 1; w2t >= (ND == 0 ? 1 : 0); --w2t) {
      PpoArgs B = A;
      const int ring_nch = ring_chunks(B, ND > 0);
      const size_t lds = tower_lds_floats(A.net, A.R, NOP, waves, HEAD == ORL_HEAD_GAUSSIAN, w2t != 0, false, ring_nch) *
                         sizeof(float);
      if (lds > 160 * 1024) continue;
      B.use_w2t = w2t;
      B.lds_floats = (int)(lds / sizeof(float));
      const int grid = launch_tower_w<HEAD, NO, ND>(B, waves, lds, s);
      const int rc = launch_status("orl_ppo_fwd_bwd");
      return rc ? -1000 - rc : grid;
    }
  }
  fail(ORL_E_UNSUPPORTED, "orl_ppo_fwd_bwd: tower (obs %d, record %d floats) does not fit 160 KiB of LDS", A.net.obs_dim,
       A.R);
  return ORL_E_UNSUPPORTED;
}

// Both towers in one launch (ppo_tower_pair_kernel), 8 waves per workgroup.  Returns 0 when the pair launch does not
// apply (the caller then launches the towers one by one), > 0 = launched (gp, gc through the out arguments), < 0 = error.
template <int HEADP, int NOP_, int ND>
static int launch_pair_nd(const PpoArgs& P, const PpoArgs& Cc, int* gp_out, int* gc_out, hipStream_t s) {
  constexpr int NOPP = NOP_ > 4 ? 16 : ((NOP_ + 3) & ~3);
  // variants in order of preference: the full split build (bf16 MFMAs over three-term splits for fc2 / dgrad / wgrad;
  // needs both bf16 images of W2 in LDS), then fp32 GEMMs + split wgrad with and without the W2^T copy (same LDS policy
  // as launch_tower: W2^T goes before a pair of waves)
  // hp.reserved & 4: the caller asks for the fp32-MFMA GEMMs (measurement / comparison switch, bench.py --tower-gemm fp32)
  // variant -1 (the wide-observation builds' first choice, ORL_TOWER_TR; the small-observation build takes it only with
  // -DORL_TOWER_TR_ND0): the full split with the dgrad through transposing reads of W2's image - no W2^T image
#if !ORL_BUILD_EXPERIMENTS
  if (P.hp.reserved & (4 | 8))
    return -1000 - fail(ORL_E_UNSUPPORTED, "orl_ppo_fwd_bwd: hparams.reserved & %d selects a comparison build (fp32-MFMA / two-image "
                        "tower pair) that this library was built without (ORL_BUILD_EXPERIMENTS)", P.hp.reserved & (4 | 8));
#endif
  for (int var = (P.hp.reserved & 4) ? 1 : ORL_TOWER_FIRST_VARIANT; var < 3; ++var) {
    const bool spt = var == -1;
    const bool sp = var <= 0;
    const int w2t = var == 0 || var == 1;
    if (var == 2 && ND == 0) break;
    if (spt && !(ND > 0 ? ORL_TOWER_TR : ORL_TOWER_TR_ND0)) continue;
    if (spt && (P.hp.reserved & 8)) continue;  // amd_tower_gemm=split_two_image: round 3's variants (comparison switch)
    PpoArgs P2 = P, C2 = Cc;
    const int nch_p = ring_chunks(P2, ND > 0), nch_c = ring_chunks(C2, ND > 0);
    const size_t lp = tower_lds_floats(P.net, P.R, NOPP, ORL_PAIR_WAVES, HEADP == ORL_HEAD_GAUSSIAN, w2t != 0, sp, nch_p) *
                      sizeof(float);
    const size_t lc = tower_lds_floats(Cc.net, Cc.R, 4, ORL_PAIR_WAVES, false, w2t != 0, sp, nch_c) * sizeof(float);
    const size_t lds = lp > lc ? lp : lc;
    if (lds > 160 * 1024) continue;
    P2.use_w2t = w2t; C2.use_w2t = w2t;
    P2.lds_floats = C2.lds_floats = (int)(lds / sizeof(float));
    const int n_tiles = (P.mb + TILE_B - 1) / TILE_B;
    // One workgroup fits a CU (LDS).  Two ways to run 2 x g workgroups on 256 CUs:
    //  * back to back: 256 + 256 workgroups, a critic workgroup starts on a CU when its policy workgroup retires - exact
    //    load balance, but the per-workgroup prologue / epilogue (~12 us: staging, first record DMA, accumulator
    //    reduction) is paid twice in sequence;
    //  * side by side: the CUs are split between the towers (both resident from the first cycle, one prologue /
    //    epilogue per launch), each wave loops over about twice as many tiles.  The split follows the towers' cost per
    //    tile (single-tower launches at the three BASELINE shapes: policy / critic = 1.05 with a narrow head,
    //    ORL_PAIR_WP_WIDE with the wide-head MFMA path of either distribution) and minimises the later tower's finish in
    //    whole tiles per wave.
    // Measured pair launch, back to back -> side by side: configuration 2's towers 63.7 -> 55.2 us at 512 envs,
    // 100.4 -> 94.3 us at 1024, 323.9 -> 319.8 us at 4096; cfg3 shape (12 800 tiles) 198 -> 186 us; cfg5 shape (51 200
    // tiles, wide head) 678 -> 694 us.  Hence side by side when the towers are nearly equal or the launch is short.
    int gp = (n_tiles + 7) / 8, gc = gp;
    if (gp > PPO_MAX_BLOCKS) gp = gc = PPO_MAX_BLOCKS;
    const double w_p = NOP_ > 4 ? (HEADP == ORL_HEAD_GAUSSIAN ? ORL_PAIR_WP_WIDE_GAUSS : ORL_PAIR_WP_WIDE) : 1.05;
    const bool side_by_side = w_p < 1.1 || n_tiles <= 12 * 8 * PPO_MAX_BLOCKS;
    if (gp + gc > PPO_MAX_BLOCKS && side_by_side) {
      double best = 1e30;
      int best_g = PPO_MAX_BLOCKS / 2;
      for (int d = 0; d <= PPO_MAX_BLOCKS / 4; ++d) {  // nearest to an even split first: ties keep it
        for (int sgn = 1; sgn >= -1; sgn -= 2) {
          const int g = PPO_MAX_BLOCKS / 2 + sgn * d;
          const double tp = (double)((n_tiles + 8 * g - 1) / (8 * g)) * w_p;
          const double tc = (double)((n_tiles + 8 * (PPO_MAX_BLOCKS - g) - 1) / (8 * (PPO_MAX_BLOCKS - g)));
          const double t = tp > tc ? tp : tc;
          if (t < best - 1e-9) { best = t; best_g = g; }
          if (d == 0) break;
        }
      }
      gp = best_g;
      gc = PPO_MAX_BLOCKS - best_g;
    }
    // Which builds exist (round 6): the shipped library instantiates only what the default path can reach - the full split
    // with two images for the small-observation towers (ND == 0), the transposing-read full split and its LDS fallback (fp32
    // GEMMs + split wgrad) for the wide ones - 25 of the 60 (head, ND, SP) combinations; the rest (the transposing build at
    // ND == 0, two images at ND >= 1, every fp32-MFMA pair) are comparison builds of earlier rounds (ORL_BUILD_EXPERIMENTS).
    constexpr bool HAVE_SPT = ND > 0 ? (ORL_TOWER_TR != 0) : (ORL_TOWER_TR_ND0 != 0);
    constexpr bool HAVE_SP2 = ND == 0 || ORL_BUILD_EXPERIMENTS;
    constexpr bool HAVE_FB = ND > 0 || ORL_BUILD_EXPERIMENTS;
    bool launched = false;
    if (spt) {
      if constexpr (HAVE_SPT) {
        (void)hipFuncSetAttribute((const void*)ppo_tower_pair_kernel<HEADP, NOP_, ND, 3>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((ppo_tower_pair_kernel<HEADP, NOP_, ND, 3>), dim3(gp + gc), dim3(64 * ORL_PAIR_WAVES), lds, s, P2, C2, gp);
        launched = true;
      }
    } else if (sp) {
      if constexpr (HAVE_SP2) {
        (void)hipFuncSetAttribute((const void*)ppo_tower_pair_kernel<HEADP, NOP_, ND, 2>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((ppo_tower_pair_kernel<HEADP, NOP_, ND, 2>), dim3(gp + gc), dim3(64 * ORL_PAIR_WAVES), lds, s, P2, C2, gp);
        launched = true;
      }
    } else {
      if (P.hp.reserved & 4) {  // every GEMM on v_mfma_f32_16x16x4_f32 (round 2's kernel + the LayerNorm fold)
#if ORL_BUILD_EXPERIMENTS
        (void)hipFuncSetAttribute((const void*)ppo_tower_pair_kernel<HEADP, NOP_, ND, 0>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((ppo_tower_pair_kernel<HEADP, NOP_, ND, 0>), dim3(gp + gc), dim3(64 * ORL_PAIR_WAVES), lds, s, P2, C2, gp);
        launched = true;
#endif
      } else if constexpr (HAVE_FB) {
        (void)hipFuncSetAttribute((const void*)ppo_tower_pair_kernel<HEADP, NOP_, ND, ORL_TOWER_WGRAD_SPLIT>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((ppo_tower_pair_kernel<HEADP, NOP_, ND, ORL_TOWER_WGRAD_SPLIT>), dim3(gp + gc), dim3(64 * ORL_PAIR_WAVES), lds, s, P2, C2, gp);
        launched = true;
      }
    }
    if (!launched) continue;  // a build this library does not carry: the next variant
    const int rc = launch_status("orl_ppo_fwd_bwd(pair)");
    if (rc) return -1000 - rc;
    *gp_out = gp; *gc_out = gc;
    return 1;
  }
  return 0;  // 8 waves do not fit: the one-by-one launches pick fewer waves
}

// Both towers in one launch (ppo_tower_pair_kernel) when both take the same build (same ND, 8 waves).  Returns 0 when the
// pair launch does not apply (the caller then launches the towers one by one), > 0 = launched (workgroups per tower
// through the out arguments), < 0 = error.
template <int HEADP, int NOP_>
static int try_launch_pair(const PpoArgs& P, const PpoArgs& Cc, int* gp_out, int* gc_out, hipStream_t s) {
#ifdef ORL_TOWER_MAXWAVES
  return 0;
#endif
  auto nd_of = [](const PpoArgs& A) {  // = launch_tower_nd's choice
    const int D = A.net.obs_dim;
    return (D <= 4 && (A.o_x & 3) == 0) ? 0 : D <= 16 + ORL_ND_REM ? 1 : D <= 32 + ORL_ND_REM ? 2 : 4;
  };
  const int nd = nd_of(P);
  if (nd != nd_of(Cc)) return 0;
  if (nd == 0) return launch_pair_nd<HEADP, NOP_, 0>(P, Cc, gp_out, gc_out, s);
  if (nd == 1) return launch_pair_nd<HEADP, NOP_, 1>(P, Cc, gp_out, gc_out, s);
  if (nd == 2) return launch_pair_nd<HEADP, NOP_, 2>(P, Cc, gp_out, gc_out, s);
  return 0;  // obs > 36 (ND == 4): rare, one launch per tower
}

template <int HEAD, int NO>
static int launch_tower_nd(const PpoArgs& A, hipStream_t s) {
  const int D = A.net.obs_dim;
  if (D <= 4 && (A.o_x & 3) == 0) return launch_tower<HEAD, NO, 0>(A, s);
  // ND = 16-column MFMA blocks of the dW1 accumulator; up to 4 remainder columns go to the VALU (17..20, 33..36)
  if (D <= 16 + ORL_ND_REM) return launch_tower<HEAD, NO, 1>(A, s);
  if (D <= 32 + ORL_ND_REM) return launch_tower<HEAD, NO, 2>(A, s);
  return launch_tower<HEAD, NO, 4>(A, s);
}

}  // namespace orl

using namespace orl;

extern "C" {

int orl_ppo_max_blocks(void) { return PPO_MAX_BLOCKS; }

#ifdef ORL_PROF
// debug build only: cumulative per-phase cycle counts of wave 0 / workgroup 0 of every tower launch; reset on read
int orl_debug_prof(unsigned long long* out16) {  // 24 counters
  unsigned long long zero[24] = {0};
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
  A.o_mk = o_adv + 4; A.a_w = a_w; A.mb = mb; A.use_w2t = 1; A.lds_floats = 0;
  // action-mask width: categorical records always carry n_out mask floats (ReplayData keeps ones)
  A.K = pnet->head_kind == ORL_HEAD_CATEGORICAL ? pnet->n_out : 0;
  ORL_REQUIRE(orl_record_width(Dp, Dc, a_w, A.K) == rec_width, "orl_ppo_fwd_bwd: record width %d != %d", rec_width,
              orl_record_width(Dp, Dc, a_w, A.K));
  hipStream_t s = (hipStream_t)stream;
  const RawLayout rlp(*pnet);

  // both towers in one launch where that build applies
  {
    PpoArgs P = A, Cc = A;
    P.net = *pnet; P.theta = ptheta; P.o_x = 0; P.partials = partials;
    Cc.net = *cnet; Cc.theta = ctheta; Cc.o_x = o_co;
    Cc.partials = partials + (size_t)PPO_MAX_BLOCKS * (rlp.total + ORL_N_STATS);
    int g1 = 0, g2 = 0, pr = 0;
    const int no_ = pnet->n_out;
    if (pnet->head_kind == ORL_HEAD_CATEGORICAL) {
      if (no_ <= 2) pr = try_launch_pair<ORL_HEAD_CATEGORICAL, 2>(P, Cc, &g1, &g2, s);
      else if (no_ <= 8) pr = try_launch_pair<ORL_HEAD_CATEGORICAL, 8>(P, Cc, &g1, &g2, s);
      else pr = try_launch_pair<ORL_HEAD_CATEGORICAL, 16>(P, Cc, &g1, &g2, s);
    } else {
      if (no_ <= 8) pr = try_launch_pair<ORL_HEAD_GAUSSIAN, 8>(P, Cc, &g1, &g2, s);
      else pr = try_launch_pair<ORL_HEAD_GAUSSIAN, 16>(P, Cc, &g1, &g2, s);
    }
    if (pr < 0) return -(pr + 1000);
    if (pr > 0) {
      if (n_blocks_out) { n_blocks_out[0] = g1; n_blocks_out[1] = g2; }
      return 0;
    }
  }
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

}  // extern "C"
