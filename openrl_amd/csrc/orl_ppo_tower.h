// orl_ppo_tower.h - the fused per-tower PPO forward + loss + backward kernel (K9-K12) for gfx950.
//
// A wavefront owns 16-row tiles of the minibatch (stride = all waves of the grid).  Per tile, out of
// registers + two private 16x64 LDS slabs, with fp32 MFMA 16x16x4 in the T layout of orl_mlp.h:
//
//   forward   x -> fc1 -> relu -> LN1 (xhat1 -> slab X1) -> n1 -> W2 (64 MFMA) -> LN2 (xhat2) -> n2 -> head
//   loss      PPO clipped surrogate + entropy / clipped huber value loss  ->  dhead   (per batch row)
//   backward  dn2 = W3^T dhead -> LN2' -> dz2 -> slab SS
//             G  += dz2^T xhat1        (wgrad, 64 MFMA, operands read from SS / X1 in F layout), db2
//             dn1 = W2^T dz2           (dgrad, 64 MFMA, A operand = W2 columns) -> LN1' -> relu' -> dz1 -> SS
//             dW1 += dz1^T x, db1
//
// Only xhat1, xhat2, dz2 and dz1 are bounced through LDS (T-layout 16-byte row stores, F-layout 4-byte
// column reads); the LayerNorm-affine gradients are NOT accumulated here at all - they are linear images
// of G / S3 / db2 / db3 and are evaluated once per update in orl_ppo_apply (see RawLayout).
//
// Records arrive through a 2-deep global->LDS DMA ring (global_load_lds, no VGPRs) one tile ahead.
// Occupancy: 2 waves per SIMD (8-wave workgroups, <= 256 VGPRs; the 64 wgrad accumulators keep a third wave out).
#pragma once
#include "orl_common.h"
#include "orl_mlp.h"

namespace orl {

// A/B knob: raise the wave's issue priority around MFMA bursts (cdna guide T5).  ORL_USE_SETPRIO = 1: the fc2 burst only (the
// round-2 form), 2: the dgrad and wgrad bursts too.
#ifndef ORL_USE_SETPRIO
#define ORL_USE_SETPRIO 2   // round 5 (six bf16 products): 1 was - 0.8 % on the pair launch, 2 no better; round 6 (three fp16 products): 2 is - 3 % against 1, 0 is + 2 % (tools/r06_calls/r06_call27.sh).  In the fp16 builds "2" raises the priority around the DGRAD burst only: the fp16 wgrad block carries no s_setprio, and with one it is + 2 % (0.1820 against 0.1780 ms)
#endif
#if ORL_USE_SETPRIO >= 1
#define ORL_PRIO(x) __builtin_amdgcn_s_setprio(x)
#else
#define ORL_PRIO(x) ((void)0)
#endif
#if ORL_USE_SETPRIO >= 2
// (the small-observation build only: the wide ones measured + 1 % with it at cfg5's shape, no difference at cfg3's; 3 = everywhere)
#define ORL_PRIO2(x)                                                           \
  do {                                                                         \
    if constexpr (ND == 0 || ORL_USE_SETPRIO >= 3) __builtin_amdgcn_s_setprio(x); \
  } while (0)
#else
#define ORL_PRIO2(x) ((void)0)
#endif

#ifdef ORL_PROF
// Phase timing build (python -m openrl_amd.csrc.build --prof; tools/tower_phase_prof.py): wave 0 of workgroup 0 reads
// s_memtime after each phase of its tile loop and lane 0 adds the delta to an LDS counter (ds_add, no return value:
// nothing on the vmcnt path - register accumulators were spilled to scratch by hipcc and global atomics sit in front
// of the tile's own s_waitcnt vmcnt(0), both of which distorted the phases being measured).
__device__ unsigned long long g_orl_prof[24];
#define ORL_T(k)                                                                  \
  do {                                                                            \
    if (prof_on) {                                                                \
      const unsigned long long t_now = __builtin_readcyclecounter();              \
      if (l == 0) atomicAdd(&prof_lds[k], t_now - t_last);                        \
      t_last = t_now;                                                             \
    }                                                                             \
  } while (0)
#elif defined(ORL_MARK)
// ISA phase markers (tools/tower_valu_budget.py: hipcc -S -DORL_MARK): an asm comment per phase boundary; the statement is
// volatile, so the markers keep their order and the instruction mix between two of them is the phase's
#define ORL_T(k) asm volatile("; ORL_PHASE " #k ::: "memory")
#else
#define ORL_T(k) ((void)0)
#endif

// 1 = ONE running scale of dz2 per wave (that of the largest tile so far) for both the dgrad and the wgrad, applied by explicit
// v_ldexp: xhat1 enters the wgrad unscaled and db2 needs no rescaling - 18 VALU per tile less than a scale per tile (0), - 1.7 % of the
// pair launch (tools/r06_calls/r06_call36.sh).  A tile 2^-k below the running maximum keeps an absolute error 2^-36 of that maximum;
// everything downstream of dz2 ends in sums over the tiles.  (Folding the power into rstd2 instead of the v_ldexp pass was 1 % SLOWER.)
#ifndef ORL_TOWER_RUNSCALE
#define ORL_TOWER_RUNSCALE 1
#endif
#ifndef ORL_DMA_WAIT_DEP
#define ORL_DMA_WAIT_DEP 1
#endif
#ifndef ORL_WGRAD_HALVES
#define ORL_WGRAD_HALVES 1
#endif
#ifndef ORL_HMM_XH2_RELOAD
#define ORL_HMM_XH2_RELOAD 0   // build-time experiment (round 5): no fewer spilled registers (20 / 29 either way) - off
#endif
#ifndef ORL_DB3_ROWLANE
#define ORL_DB3_ROWLANE 1   // build-time A/B switch (round 5): db3 of narrow heads from per-row register partials
#endif
constexpr int PPO_MAX_BLOCKS = 256;  // one workgroup per CU
constexpr int TS = 68;               // slab row stride (floats): 16-byte rows, bank-skewed
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
  int use_w2t;         // keep a transposed LDS copy of W2 for the dgrad GEMM (dropped when it costs a pair of waves)
  // record ring of the wide-observation builds (ND >= 1): only the 16-byte chunks this tower reads travel to LDS - its
  // observation columns (chunks [c0_beg, c0_beg + c0_n)) and the tail fields from its first one on ([c1_beg, R / 4)) -
  // stored back to back (ring_chunks()); the ND == 0 build keeps whole records (4 chunks)
  int c0_beg, c0_n, c1_beg;
  int lds_floats;  // dynamic LDS of the launch, in floats (the epilogue spreads the waves' accumulators over all of it)
};

// which chunks of a record a tower needs: fills A.c0_beg / c0_n / c1_beg, returns the number of chunks per row in the ring
__host__ __device__ inline int ring_chunks(PpoArgs& A, bool compact) {
  const int nch = A.R >> 2;
  if (!compact) { A.c0_beg = 0; A.c0_n = nch; A.c1_beg = nch; return nch; }
  const int DP = (A.net.obs_dim + 3) & ~3;
  const int first_tail = A.net.head_kind == ORL_HEAD_VALUE ? A.o_vp : A.o_act;
  A.c0_beg = A.o_x >> 2;
  int c0_end = ((A.o_x + DP - 1) >> 2) + 1;   // fc1 reads DP columns from o_x (W1's image is zero beyond D)
  if (c0_end > nch) c0_end = nch;
  A.c0_n = c0_end - A.c0_beg;
  A.c1_beg = first_tail >> 2;
  if (A.c1_beg < c0_end) A.c1_beg = c0_end;   // ranges touch / overlap: one contiguous run
  return A.c0_n + (nch - A.c1_beg);
}

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

// T-layout 16x64 register tile <-> LDS slab rows (lane (j,q) owns 4 x 16 bytes of row j)
__device__ inline void store_slab_T(float* __restrict__ slab, const f32x4 (&x)[4], int j, int q) {
#pragma unroll
  for (int m = 0; m < 4; ++m) *(f32x4*)(slab + j * TS + 16 * m + 4 * q) = x[m];
}
__device__ inline void load_slab_T(const float* __restrict__ slab, f32x4 (&x)[4], int j, int q) {
#pragma unroll
  for (int m = 0; m < 4; ++m) x[m] = *(const f32x4*)(slab + j * TS + 16 * m + 4 * q);
}

// LayerNorm backward in T layout, in place, on d = d loss / d xhat (the affine's gamma is folded into the weights the
// incoming gradient was multiplied with, stage_tower(fold)):  d <- rstd * (d - mean(d) - xhat * mean(d * xhat))
__device__ inline void ln_bwd_T(f32x4 (&d)[4], const f32x4 (&xhat)[4], float rstd) {
  float s1 = lane_sum16(d), s2 = lane_dot16(d, xhat);
  row_allsum2(s1, s2);
  // rstd * (d*g - c1 - xhat*c2) as two FMAs per element
  const float k1 = -s1 * (1.0f / 64.0f) * rstd, k2 = -s2 * (1.0f / 64.0f) * rstd;
#pragma unroll
  for (int m = 0; m < 4; ++m) d[m] = xhat[m] * k2 + (d[m] * rstd + k1);
}

// acc[mo] += sum_o W[o][16mo + j-th column] * in[o]   i.e.  acc = W^T in  with W row-major [64][W2S]:
// the dgrad GEMM reads W2 by columns instead of keeping a transposed copy in LDS (rows 4q apart hit
// banks 16 apart, so the 4-byte column reads are conflict free).
__device__ inline void mm64_T_wt(const float* __restrict__ Ws, const f32x4 (&in)[4], f32x4 (&acc)[4], int j, int q) {
  // column reads of k-step k+1 issued before the MFMAs of k-step k (see mm64_S_wt in orl_rnn.h)
  float a[2][4];
#pragma unroll
  for (int mo = 0; mo < 4; ++mo) a[0][mo] = Ws[(4 * q) * W2S + j + 16 * mo];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int mi = k >> 2, r = k & 3;
    if (k < 15) {
      const float* row = Ws + (16 * ((k + 1) >> 2) + 4 * q + ((k + 1) & 3)) * W2S + j;
#pragma unroll
      for (int mo = 0; mo < 4; ++mo) a[(k + 1) & 1][mo] = row[16 * mo];
    }
#pragma unroll
    for (int mo = 0; mo < 4; ++mo) acc[mo] = ORL_MFMA(a[k & 1][mo], in[mi][r], acc[mo]);
  }
}

// HEAD: ORL_HEAD_VALUE / _CATEGORICAL / _GAUSSIAN; NO: padded head width; ND: ceil(D/16) for the MFMA
// dW1 path, 0 = VALU path (D <= 4, 16-byte aligned obs column).
// SP: bf16 MFMAs over three-term splits of the fp32 operands (orl_mlp.h) instead of v_mfma_f32_16x16x4_f32:
//   2 = the three 64-wide GEMMs of a tile (fc2, dgrad, wgrad); needs the bf16 images of W2 and W2^T in LDS (+ 20 KB);
//   1 = the wgrad only - its operands come from the slabs, no LDS image is involved - for towers whose records leave no
//       room for the bf16 images (wide observations / heads);
//   3 = as 2, but the dgrad reads W2's image through the transposing LDS read (mm64_T_split_tr) instead of a second,
//       transposed image: 27.6 KB less LDS - what lets the wide-observation towers (cfg3 / cfg5) take the full split;
//   0 = none.  fc1, the wide-head GEMMs and the MFMA dW1 path stay on the fp32 MFMA (a few instructions per tile).
template <int HEAD, int NO, int ND, int SP_ = 0>
__device__ __forceinline__ void ppo_tower_body(const PpoArgs& A, const int bid, const int nblk) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
#ifdef ORL_PROF
  const unsigned long long t_entry = __builtin_readcyclecounter();
#endif
  const TowerLayout tl(A.net);
  const RawLayout rl(A.net);
  // On the ND == 0 path the padded obs width is 4 and for NO <= 4 the padded head width is 4: every LDS
  // offset of the tower image becomes a compile-time constant (immediate offsets instead of address VGPRs).
  // wide heads (NO > 4: Discrete(5..16), Box(5..16)) run their three head GEMMs - logits, dn2 = W3^T dhead,
  // S3 += dhead^T xhat2 - on MFMA; with NO x 64 scalar FMAs per lane they were 2/3 of the tile and spilled 360 VGPRs
  constexpr bool HMM = NO > 4;
  constexpr bool SP = SP_ >= 2;   // the GEMMs with LDS images
  constexpr bool SPT = SP_ == 3;  // ... the dgrad through transposing reads of W2's image (no W2^T image)
  constexpr bool SPW = SP_ >= 1;  // the wgrad
  // compile-time true on the small-observation path and in the two-image split builds, false with transposing reads
  const bool w2t = SPT ? false : (ND == 0 || SP || A.use_w2t);
  const TowerLds tw(ND == 0 ? 4 : A.net.obs_dim, NO <= 4 ? 4 : A.net.n_out, HEAD == ORL_HEAD_GAUSSIAN, w2t, HMM, SP);
  const int DP = tw.DP;
  const int D = A.net.obs_dim;
  const int n_out = A.net.n_out;
  constexpr int NOP = HMM ? 16 : ((NO + 3) & ~3);  // width of the dhead tile in LDS
  const int wave = threadIdx.x >> 6, l = threadIdx.x & 63, j = l & 15, q = l >> 4;
  // chunks per record row in the ring: whole records on the small-observation path, this tower's chunks otherwise
  const int nch = ND == 0 ? (A.R >> 2) : A.c0_n + ((A.R >> 2) - A.c1_beg);
  // floats per record-ring slot (ring chunk c of row r at c*64 + r*4); the small-observation path keeps the form that is
  // visibly a multiple of 256 (measured: 1.2 % faster tower at config 2), wider records are sized exactly
  const int rts = ND == 0 ? ((nch + 3) >> 2) * 256 : nch * 64;
  const int per_wave = 2 * SLAB + 2 * rts + TILE_B * NOP;
  float* wl = smem + tw.total + wave * per_wave;
  float* X1 = wl;              // xhat1 slab (forward -> wgrad / LN1 backward)
  float* SS = wl + SLAB;       // scratch slab: xhat2 -> dz2 -> dz1
  float* RR = wl + 2 * SLAB;   // record ring: 2 slots of [chunk][16 rows][4 floats]
  float* DH = RR + 2 * rts;    // dhead [16][NOP]

  const float* lw = smem;
  const orl_ppo_hparams hp = A.hp;

  // ---- persistent accumulators (live across all tiles of this wave) -----------------------------------
  f32x4 G[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) G[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  // split builds: G as 2 x 2 blocks of the 32x32x16 bf16 MFMA (rows o = 32 bo + .., columns i = 32 bi + ..), and db2 as
  // per-lane partial column sums of the wgrad's own operand reads (lane (c, kb): column 32 bo + c over rows 8kb .. 8kb+7)
  f32x16 GS[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) GS[a][b][r] = 0.f;
  float a_db2s[2] = {0.f, 0.f};
  // split builds with one observation block (ND == 1: obs <= 20 columns): dW1 = dz1^T x on the 32x32x16 bf16 MFMA like the
  // wgrad - 2 row blocks x ONE 32-column block (columns >= obs_dim are dead), operands split in registers, db1 from the
  // operand reads.  Replaces 16 VALU-blocking fp32 MFMAs + the <= 4-column VALU remainder + a 16-read column sum: the
  // phase profile had this phase at 3 200 of a tile's 21 200 cycles at obs 18 (534 of 15 900 on the small-observation path).
  constexpr bool W1S = SPW && ND == 1;
  f32x16 G1S[2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) G1S[a][r] = 0.f;
  float a_db1s[2] = {0.f, 0.f};
  constexpr int NDA = ND > 0 ? ND : 1;
  f32x4 G1[4][NDA];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < NDA; ++b) G1[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  float w1v[4] = {0.f, 0.f, 0.f, 0.f};  // VALU dW1 (ND == 0): lane f, k = 0..3
  float a_db2 = 0.f, a_db1 = 0.f, a_db3 = 0.f, a_dls = 0.f;
  float a_db3r[NO];  // narrow heads (ORL_DB3_ROWLANE): db3 partials of this lane's batch row, combined in the epilogue
#pragma unroll
  for (int c = 0; c < NO; ++c) a_db3r[c] = 0.f;
  constexpr int NS3 = HMM ? 1 : NO;
  float a_S3[NS3];  // narrow heads: S3[c][f = lane] sums on the VALU
#pragma unroll
  for (int c = 0; c < NS3; ++c) a_S3[c] = 0.f;
  f32x4 S3acc[4];   // wide heads: MFMA tiles, lane (j, q) reg r -> S3[c = 4q+r][f = 16mi+j]
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) S3acc[mi] = f32x4{0.f, 0.f, 0.f, 0.f};
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
  const int nwv = blockDim.x >> 6;
  const int wave_g = bid * nwv + wave;
  const int n_waves = nblk * nwv;

  // ---- record tile pipeline: DMA global -> LDS one tile ahead, indices two tiles ahead ----------------
  auto row_of = [&](int t) -> long long {
    const int ii = t * TILE_B + j;
    if (t >= n_tiles || ii >= A.mb) return 0;  // invalid lanes read row 0 (finite data, weight 0)
    return (A.idx != nullptr) ? A.idx[ii] : (long long)ii;
  };
  // The DMA is issued from an asm statement on purpose.  hipcc counts __builtin_amdgcn_global_load_lds in vmcnt and,
  // having no alias information for it, drains it (s_waitcnt vmcnt(0)) before the next LDS read of the wave - i.e.
  // right after the issue, a full HBM round trip per tile with nothing overlapped (2 900 of a tile's 24 000 cycles in
  // the phase profile).  An asm statement is invisible to that bookkeeping; the one wait this pipeline needs is the
  // explicit vmcnt(0) at the top of the tile loop, a whole tile after the issue (guide section 5.7: M0 is written in
  // the same statement that reads it; the s_nop covers the SALU-write -> M0 use hazard).
  auto issue_dma = [&](float* slot, long long row) {
    const float* src = A.records + (size_t)row * A.R;
#pragma unroll 1
    for (int g = 0; 4 * g < nch; ++g) {
      const int c = 4 * g + q;  // ring chunk; record chunk = c (ND == 0) or through the tower's two ranges
      const int rc = ND == 0 ? c : (c < A.c0_n ? A.c0_beg + c : A.c1_beg + (c - A.c0_n));
      const unsigned lds_dst = __builtin_amdgcn_readfirstlane(
          (unsigned)(size_t)(__attribute__((address_space(3))) float*)(slot + g * 256));
      if (c < nch) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(src + 4 * rc), "s"(lds_dst)
                     : "memory");
      }
    }
  };
  long long row_next = 0;
  int ring = 0;
  row_next = row_of(wave_g);
  issue_dma(RR, row_next);
  row_next = row_of(wave_g + n_waves);
  // the tower image is staged AFTER the first tile's index loads and record DMA are in flight (their latency chain -
  // index -> record rows -> LDS - overlaps the parameter loads instead of following them); the record ring and the
  // image are disjoint LDS regions
  stage_tower(smem, A.theta, tl, tw, w2t, threadIdx.x, blockDim.x, HMM, SP, true);
  __syncthreads();
  // fp16 split builds (orl_mlp.h, ORL_TOWER_F16): the weight image's scale 2^kw (fc2's accumulators and LayerNorm 2 run scaled:
  // eps x 4^kw), and the scale 2^LS of this wave's wgrad accumulators: with ORL_TOWER_RUNSCALE every tile's dz2 is scaled by the
  // power of two of the largest tile so far (e_run) and LS = 138 - e_run; without it a tile's gradients are scaled by their own
  // maximum and its xhat1 operand by what is left of 2^LS.  LS only ever decreases, and the accumulators follow it.
  constexpr bool F16W = SPW && ORL_TOWER_F16;  // the wgrad (and with it the tile scale of dz2)
  constexpr bool F16G = SP && ORL_TOWER_F16;   // fc2 / dgrad over the fp16 images
  const int kw = F16G ? (int)smem[tw.wsc] : 0;
  const float w_scale = F16G ? smem[tw.wsc + 1] : 1.f, ln2_eps = F16G ? smem[tw.wsc + 3] : 1e-5f;
  int LS = 1000;
  int e_run = 0;  // ORL_TOWER_RUNSCALE: the largest tile exponent so far

#ifdef ORL_PROF
  __shared__ unsigned long long prof_lds[16];
  const bool prof_on = bid == 0 && wave == 0;
  if (prof_on && l < 16) prof_lds[l] = 0ull;
  unsigned long long t_last = __builtin_readcyclecounter();
  const unsigned long long t_loop0 = t_last;
#endif
  for (int tile = wave_g; tile < n_tiles; tile += n_waves) {
    const int i = tile * TILE_B + j;
    const bool valid = i < A.mb;
    // This tile's records have landed in LDS.  On the small-observation build the wait orders only what reads the ring: it
    // "redefines" the ring pointer instead of clobbering memory, so hipcc keeps scheduling the tile's other LDS traffic across
    // it (- 1.5 % of the pair launch, tools/r06_calls/r06_call22.sh; the wide builds pay 23 more registers for it and spill: they
    // keep the barrier form).  The laundered pointer is a GENERIC one: the ring's 20 reads per tile become flat_load (the vector
    // memory path) - kept on purpose: laundering it as an address-space-3 pointer (ds_read again, 27 registers less) measured
    // 0.1856 / 0.1832 ms per launch against 0.1777 / 0.1734 for this form (tools/r06_calls/r06_call29.sh).
    // ORL_PROBE_NO_DMA_WAIT: timing probe, never shipped (reads the ring without waiting).
    const float* RT = RR + ring * rts;
#if !defined(ORL_PROBE_NO_DMA_WAIT)
    if constexpr (ND == 0 && ORL_DMA_WAIT_DEP) asm volatile("s_waitcnt vmcnt(0)" : "+v"(RT));
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    // Order matters: the index prefetch (a load hipcc counts) is re-issued BEFORE the DMA (which it cannot see).  The
    // other way round, hipcc's own "previous load into these registers must have landed" wait sits behind the DMA
    // and drains it.
    const long long row_cur = row_next;
    row_next = row_of(tile + 2 * n_waves);
    issue_dma(RR + (ring ^ 1) * rts, row_cur);        // next tile's records, hidden behind this tile
    ring ^= 1;
    // column -> ring position: observation columns live in range 0 (ring chunk = record chunk - c0_beg), every other field
    // in range 1 (ring chunk = record chunk - c1_beg + c0_n); both biases fold into the base pointer.  ND == 0: identity.
    const float* RTX = ND == 0 ? RT : RT - 64 * A.c0_beg;
    const float* RTT = ND == 0 ? RT : RT + 64 * (A.c0_n - A.c1_beg);
#define REC(col) RTT[(((col) >> 2) << 6) + (j << 2) + ((col) & 3)]
#define RECX(col) RTX[(((col) >> 2) << 6) + (j << 2) + ((col) & 3)]
#define REC_R(r, col) RTX[(((col) >> 2) << 6) + ((r) << 2) + ((col) & 3)]

    ORL_T(0);  // record DMA wait + next issue
    // ---------------- forward ----------------
    float rstd1, rstd2;
    // relu'(z1) as 16 wave-wide lane masks (one v_cmp each, kept in SGPR pairs) instead of a per-lane bit field that
    // costs an or per element here and an and + compare per element in the backward pass
    unsigned long long relu_mask[16];
    float hd[NO];
    f32x4 hv = f32x4{0.f, 0.f, 0.f, 0.f};  // wide heads: this lane's 4 logits
    // xhat2 stays in registers from LN2 to LN2' (16 VGPRs through the head / loss / S3 phases, where pressure is low): its
    // T-layout reload from the scratch slab was an exposed LDS round trip in front of the backward pass
    f32x4 xh2[4];
    {
      f32x4 z[4];
      load_vec_T(lw + tw.b1, q, z);
      // columns >= D of a record are other (finite) fields; W1's LDS image is zero-padded there
      if constexpr (ND == 0 || ND > 2) {  // (ND = 4: 17 k-steps of operands would not fit the registers - rolled loop)
        fc1_T(lw + tw.W1, DP, [&](int s) -> float { return RECX(A.o_x + 4 * s + q); }, z, j, q);
      } else {
        // wide observations: every operand of the <= 4 ND + 1 k-steps is requested BEFORE the first MFMA (the rolled loop
        // of fc1_T opened each k-step with an exposed LDS round trip: this phase was 3 560 of a tile's 20 900 cycles at obs
        // 18, against 1 500 on the small-observation path)
        constexpr int MAXS = 4 * ND + 1;
        const int nks = DP >> 2;
        float bq[MAXS], aq[MAXS][4];
#pragma unroll
        for (int s = 0; s < MAXS; ++s) {
          const bool on = s < nks;
          bq[s] = on ? RECX(A.o_x + 4 * s + q) : 0.f;
#pragma unroll
          for (int m = 0; m < 4; ++m) aq[s][m] = on ? lw[tw.W1 + (16 * m + j) * DP + 4 * s + q] : 0.f;
        }
#pragma unroll
        for (int s = 0; s < MAXS; ++s) {
          if (s < nks) {
#pragma unroll
            for (int m = 0; m < 4; ++m) z[m] = ORL_MFMA(aq[s][m], bq[s], z[m]);
          }
        }
      }
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          relu_mask[4 * m + r] = __builtin_amdgcn_ballot_w64(z[m][r] > 0.f);
          z[m][r] = fmaxf(z[m][r], 0.f);
        }
      ln_normalize_T(z, rstd1);  // z = xhat1
      store_slab_T(X1, z, j, q);
      load_vec_T(lw + tw.b2, q, xh2);  // the folded bias b2 + W2 be1; the GEMM runs on xhat1 itself (W2 diag(g1) image)
      ORL_T(1);  // fc1, relu, LN1, slab store, affine
      ORL_PRIO(1);
      if constexpr (F16G) {
        u32x4 xs[2][2];
        split_Th(z, xs);
        mm64_T_h2((const unsigned short*)(lw + tw.W2), xs, xh2, j, q);  // 2^kw z2 (the bias slot holds 2^kw b2')
      } else if constexpr (SP) {
        u32x4 xs[2][3];
        split_T(z, xs);
        mm64_T_split((const unsigned short*)(lw + tw.W2), xs, xh2, j, q);
      } else {
        mm64_T(lw + tw.W2, z, xh2, j, q);
      }
      ORL_PRIO(0);
      ORL_T(2);  // fc2: 64 MFMA
      ln_normalize_T(xh2, rstd2, ln2_eps);
      if constexpr (F16G) rstd2 *= w_scale;  // of the unscaled row
      store_slab_T(SS, xh2, j, q);  // parked in the scratch slab: read back in F layout (S3) and T layout (LN2')
#pragma unroll
      for (int m = 0; m < 4; ++m) z[m] = xh2[m];  // the head runs on xhat2 (W3 diag(g2) image, folded bias)
      if constexpr (HMM) {
        // logits^T[16 c x 16 rows] = W3p n2^T: lane (j, q) keeps logits c = 4q..4q+3 of row j - the layout the loss,
        // the dhead tile and the dn2 GEMM below all work in (no redistribution)
        const int no4 = (n_out + 3) & ~3;
        hv = f32x4{0.f, 0.f, 0.f, 0.f};
        if (4 * q < no4) hv = *(const f32x4*)(lw + tw.b3 + 4 * q);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
          const f32x4 a4 = *(const f32x4*)(lw + tw.W3P + j * W2S + 16 * mi + 4 * q);
#pragma unroll
          for (int r = 0; r < 4; ++r) hv = ORL_MFMA(a4[r], z[mi][r], hv);
        }
      } else {
        head_T<NO>(lw + tw.W3, lw + tw.b3, n_out, z, q, hd);
      }
    }

    ORL_T(3);  // LN2, slab store, affine, head
    // ---------------- loss + dhead (per batch row; the 4 lanes of a row compute identically) --------
    float dh[NO];
#pragma unroll
    for (int c = 0; c < NO; ++c) dh[c] = 0.f;
    f32x4 dhv = f32x4{0.f, 0.f, 0.f, 0.f};  // wide heads: d loss / d logit of this lane's 4 classes / dimensions
    const float active = valid ? REC(A.o_am) : 0.f;
    if constexpr (HMM) {
      // Each lane owns classes / action dimensions c = 4q + r; row-wide quantities are in-lane partials combined
      // over the 4 lanes of the row (row_allsum / row_allmax).  Same formulas as the narrow-head code below.
      const float w = valid ? (hp.use_policy_active_masks ? active : 1.f) : 0.f;
      const float adv = valid ? REC(A.o_adv) : 0.f;
      if (HEAD == ORL_HEAD_CATEGORICAL) {
        float lg[4], mk[4];
        float mx = -3.0e38f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = 4 * q + r;
          mk[r] = (A.K > 0 && valid && c < n_out) ? REC(A.o_mk + c) : 1.f;
          lg[r] = (mk[r] == 0.f) ? -6e4f : hv[r];
          if (c < n_out) mx = fmaxf(mx, lg[r]);
        }
        mx = row_allmax(mx);
        float se = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (4 * q + r < n_out) se += __expf(lg[r] - mx);
        se = row_allsum(se);
        const float lse = mx + __logf(se);
        const int act = valid ? (int)REC(A.o_act) : 0;
        const float old_lp = valid ? REC(A.o_lp) : 0.f;
        float pk = 0.f, entp = 0.f;
        float p[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = 4 * q + r;
          p[r] = 0.f;
          if (c < n_out) {
            const float ell = lg[r] - lse;
            p[r] = __expf(ell);
            entp -= p[r] * ell;
            if (c == act) pk = lg[r];
          }
        }
        const float lp = row_allsum(pk) - lse;
        const float ent = row_allsum(entp);
        float ratio = __expf(lp - old_lp);
        const float ratio_raw = ratio;
        float dr_eff = 1.f;
        if (hp.dual_clip_ppo) {
          if (ratio > hp.dual_clip_coeff) { ratio = hp.dual_clip_coeff; dr_eff = 0.f; }
        }
        const float s1 = ratio * adv;
        const float s2 = fminf(fmaxf(ratio, 1.f - hp.clip_param), 1.f + hp.clip_param) * adv;
        float surr = fminf(s1, s2);
        float gl = -((s1 <= s2) ? adv : 0.f) * dr_eff * ratio_raw;  // d(-surr)/d logp
        if (hp.reserved & 2) {  // A2C
          surr = adv * lp;
          gl = -adv;
          ratio = 0.f;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = 4 * q + r;
          if (c < n_out) {
            const float ell = lg[r] - lse;
            const float d = gl * ((c == act ? 1.f : 0.f) - p[r]) + hp.entropy_coef * p[r] * (ell + ent);
            dhv[r] = (mk[r] == 0.f) ? 0.f : w * d;
          }
        }
        if (q == 0 && valid) {
          st_active += active; st_rows += 1.f; st_loss += -surr * w; st_ent += ent * w; st_ratio += ratio;
        }
      } else {
        const float ent_scale = hp.use_policy_active_masks ? 1.f : 1.f / (float)n_out;
        float surr_sum = 0.f, ent_sum = 0.f, ratio_sum = 0.f;
        f32x4 dlsv = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = 4 * q + r;
          if (c < n_out) {
            const float ls = lw[tw.logstd + c];
            const float sd = expf(ls);
            const float av = valid ? REC(A.o_act + c) : 0.f;
            const float old_lp = valid ? REC(A.o_lp + c) : 0.f;
            const float dmu = av - hv[r];
            const float var = sd * sd;
            const float lp = -(dmu * dmu) / (2.f * var) - ls - 0.91893853320467274178f;
            float ratio = __expf(lp - old_lp);
            const float ratio_raw = ratio;
            float dr_eff = 1.f;
            if (hp.dual_clip_ppo && ratio > hp.dual_clip_coeff) { ratio = hp.dual_clip_coeff; dr_eff = 0.f; }
            const float s1 = ratio * adv;
            const float s2 = fminf(fmaxf(ratio, 1.f - hp.clip_param), 1.f + hp.clip_param) * adv;
            float surr_c = fminf(s1, s2);
            float gl = -((s1 <= s2) ? adv : 0.f) * dr_eff * ratio_raw;
            if (hp.reserved & 2) {  // A2C
              surr_c = adv * lp;
              gl = -adv;
              ratio = 0.f;
            }
            surr_sum += surr_c;
            dhv[r] = w * gl * dmu / var;
            dlsv[r] = w * (gl * (dmu * dmu / var - 1.f) - hp.entropy_coef * ent_scale);
            ent_sum += 1.41893853320467274178f + ls;
            ratio_sum += ratio;
          }
        }
        surr_sum = row_allsum(surr_sum);
        ent_sum = row_allsum(ent_sum);
        ratio_sum = row_allsum(ratio_sum);
        if (q == 0 && valid) {
          st_active += active; st_rows += 1.f; st_loss += -surr_sum * w; st_ent += ent_sum * w;
          st_ratio += ratio_sum;
        }
        // dlogstd: column sums of the [16 rows][16 dims] tile
        *(f32x4*)(DH + j * NOP + 4 * q) = dlsv;
        wave_lds_fence();
        if (l < n_out) {
          float s = 0.f;
          for (int r = 0; r < TILE_B; ++r) s += DH[r * NOP + l];
          a_dls += s;
        }
        wave_lds_fence();
      }
    } else
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
            p[c] = __expf(ell);
            ent -= p[c] * ell;
          }
        }
        float ratio = __expf(lp - old_lp);
        const float ratio_raw = ratio;
        float dr_eff = 1.f;
        if (hp.dual_clip_ppo) {
          if (ratio > hp.dual_clip_coeff) { ratio = hp.dual_clip_coeff; dr_eff = 0.f; }
        }
        const float s1 = ratio * adv;
        const float s2 = fminf(fmaxf(ratio, 1.f - hp.clip_param), 1.f + hp.clip_param) * adv;
        float surr = fminf(s1, s2);
        const float dsurr_dr = (s1 <= s2) ? adv : 0.f;
        float gl = -dsurr_dr * dr_eff * ratio_raw;  // d(-surr)/d logp
        if (hp.reserved & 2) {  // A2C (algorithms/a2c.py:88-98): loss = -adv * logp, train_info ratio = 0
          surr = adv * lp;
          gl = -adv;
          ratio = 0.f;
        }
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
            float ratio = __expf(lp - old_lp);
            const float ratio_raw = ratio;
            float dr_eff = 1.f;
            if (hp.dual_clip_ppo && ratio > hp.dual_clip_coeff) { ratio = hp.dual_clip_coeff; dr_eff = 0.f; }
            const float s1 = ratio * adv;
            const float s2 = fminf(fmaxf(ratio, 1.f - hp.clip_param), 1.f + hp.clip_param) * adv;
            float surr_c = fminf(s1, s2);
            float gl = -((s1 <= s2) ? adv : 0.f) * dr_eff * ratio_raw;
            if (hp.reserved & 2) {  // A2C
              surr_c = adv * lp;
              gl = -adv;
              ratio = 0.f;
            }
            surr_sum += surr_c;
            dh[c] = w * gl * dmu / var;
            // dlogstd contribution of this row, column-summed through DH right below
            const float dls = w * (gl * (dmu * dmu / var - 1.f) - hp.entropy_coef * ent_scale);
            if (q == 1) DH[j * NOP + c] = dls;
            ent_sum += 1.41893853320467274178f + ls;
            ratio_sum += ratio;
          }
        }
        if (q == 0 && valid) {
          st_active += active; st_rows += 1.f; st_loss += -surr_sum * w; st_ent += ent_sum * w;
          st_ratio += ratio_sum;
        }
        wave_lds_fence();
        if (l < n_out) {
          float s = 0.f;
          for (int r = 0; r < TILE_B; ++r) s += DH[r * NOP + l];
          a_dls += s;
        }
        wave_lds_fence();
      }
    }

    ORL_T(4);  // loss
    // ---------------- backward ----------------
    // S3 += dhead^T xhat2, db3 (F layout: lane = feature f); xhat2 comes from the scratch slab
    if constexpr (HMM) {
      *(f32x4*)(DH + j * NOP + 4 * q) = dhv;
    } else if (q == 0) {
#pragma unroll
      for (int c = 0; c < NOP; ++c) DH[j * NOP + c] = c < NO ? dh[c < NO ? c : 0] : 0.f;
    }
    wave_lds_fence();
    if constexpr (HMM) {
      // K = 16 rows: A = dhead[row = 4s+q][c = j], B = xhat2[row = 4s+q][f = 16mi+j], both straight from LDS
      float s_db3 = 0.f;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float a = DH[(4 * s + q) * NOP + j];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) S3acc[mi] = ORL_MFMA(a, SS[(4 * s + q) * TS + 16 * mi + j], S3acc[mi]);
      }
      for (int r = 0; r < TILE_B; ++r) s_db3 += DH[r * NOP + (l & 15)];
      a_db3 += s_db3;
    } else {
      const int f = l;
#if ORL_DB3_ROWLANE
      // db3 = the column sums of dhead: every lane of a batch row holds that row's dh[] already, so the q == 0 lane of each
      // row accumulates it in registers across tiles and the 16 rows are combined ONCE, in the epilogue (wave_sum) - no
      // per-tile LDS reads for it, and the S3 loop is a plain chain of FMAs (with db3 inside, hipcc paired {S3, db3} into
      // v_pk_add_f32 with a v_mov + v_mul per row: 3 instructions where one v_fmac does)
#pragma unroll
      for (int c = 0; c < NO; ++c) a_db3r[c] += q == 0 ? dh[c] : 0.f;
#pragma unroll
      for (int r = 0; r < TILE_B; ++r) {
        const float xh = SS[r * TS + f];
#pragma unroll
        for (int c = 0; c < NO; ++c) a_S3[c < NS3 ? c : 0] += DH[r * NOP + c] * xh;
      }
#else
      float s3[NO];
#pragma unroll
      for (int c = 0; c < NO; ++c) s3[c] = 0.f;
      float s_db3 = 0.f;
      for (int r = 0; r < TILE_B; ++r) {
        const float xh = SS[r * TS + f];
#pragma unroll
        for (int c = 0; c < NO; ++c) s3[c] += DH[r * NOP + c] * xh;
        s_db3 += DH[r * NOP + (f < NO ? f : 0)];
      }
#pragma unroll
      for (int c = 0; c < NO; ++c) a_S3[c < NS3 ? c : 0] += s3[c];
      a_db3 += s_db3;
#endif
    }
    ORL_T(5);  // dhead store, S3 / db3 column sums
    // dn2 = W3^T dhead (T layout), LN2 backward -> dz2
    f32x4 d2[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) d2[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (HMM) {
      // dn2^T[64 f x 16 rows] = W3p^T[64 x 16 c] dhead^T: the dhead tile is this lane's B operand in T layout
      const f32x4 dv = dhv;
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int mo = 0; mo < 4; ++mo)
          d2[mo] = ORL_MFMA(lw[tw.W3P + (4 * q + r) * W2S + 16 * mo + j], dv[r], d2[mo]);
    } else {
      // (no test on n_out: dh[c] is 0 and W3's LDS image is zero padded for classes >= n_out - the test was a branch around
      // sixteen v_mov 0 per tile)
#pragma unroll
      for (int c = 0; c < NO; ++c) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const f32x4 w3 = *(const f32x4*)(lw + tw.W3 + c * HID + 16 * m + 4 * q);
          d2[m] = c == 0 ? w3 * dh[0] : w3 * dh[c] + d2[m];
        }
      }
    }
#if ORL_HMM_XH2_RELOAD
    if constexpr ((HMM && ND > 0) || ORL_HMM_XH2_RELOAD == 2) {
      // wide head + wide observations: x-hat2 comes back from the scratch slab here (it still holds it: dz2 is stored below)
      // instead of staying in 16 registers through the loss - the build sits at 256 registers and spilled 20 - 33 of them
      // around the loss phase (profiles/r04_tower_isa_budget.txt: 7 - 15 scratch operations per tile there)
      const float* ssr = SS;
      asm volatile("" : "+v"(ssr));  // (a laundered pointer: otherwise the stored registers are forwarded and stay live)
      f32x4 xh2r[4];
      load_slab_T(ssr, xh2r, j, q);
      ln_bwd_T(d2, xh2r, rstd2);
    } else
#endif
    ln_bwd_T(d2, xh2, rstd2);
    int eA = 138;  // the tile's dz2 leaves scaled by 2^(138 - eA): its largest |element| in [2^11, 2^12)
    if constexpr (F16W) {
      float mx = 0.f;
#pragma unroll
      for (int m = 0; m < 4; ++m) mx = fmaxf(fmaxf(mx, fmaxf(fabsf(d2[m][0]), fabsf(d2[m][1]))), fmaxf(fabsf(d2[m][2]), fabsf(d2[m][3])));
      eA = __builtin_amdgcn_readfirstlane(scale_exponent(wave_absmax(mx)));
#if ORL_TOWER_RUNSCALE
      // ONE running scale (that of the largest tile so far) for dz2 in both the dgrad and the wgrad - xhat1 enters the wgrad
      // unscaled, G / db2 are in units of 2^(138 - e_run); a larger tile rescales them (exact, rare after a wave's first tiles)
      if (eA > e_run) {
        const int dl = e_run - eA;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
#pragma unroll
          for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) GS[a][b][r] = __builtin_ldexpf(GS[a][b][r], dl);
          a_db2s[a] = __builtin_ldexpf(a_db2s[a], dl);
        }
        e_run = eA;
      }
      eA = e_run;
      LS = 138 - e_run;
#endif
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) d2[m][r] = __builtin_ldexpf(d2[m][r], 138 - eA);
#if !ORL_TOWER_RUNSCALE
      // G's scale: 2^LS x (dz2^T xhat1) = (2^(138 - eA) dz2)^T (2^LB xhat1) with LB = LS - (138 - eA) <= 12 (|xhat1| < 8 stays
      // below 2^15); a tile with larger gradients than any before lowers LS, and the accumulators are rescaled (exact)
      const int ls_new = 12 + 138 - eA;
      if (ls_new < LS) {
        const int dl = ls_new - LS;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
#pragma unroll
          for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) GS[a][b][r] = __builtin_ldexpf(GS[a][b][r], dl);
          a_db2s[a] = __builtin_ldexpf(a_db2s[a], dl);
        }
        LS = ls_new;
      }
#endif
    }
    wave_lds_fence();
    store_slab_T(SS, d2, j, q);
    wave_lds_fence();
    ORL_T(6);  // dn2, LN2 backward, dz2 slab store
    if constexpr (SPW) {
      // wgrad on the 32x32x16 bf16 MFMA (K = the tile's 16 rows): lane (c = l & 31, kb = l >> 5) reads rows 8kb..8kb+7 of
      // columns 32b + c of both slabs (32 conflict-free 4-byte reads, as many as the fp32 path), splits them in
      // registers and issues 6 products x 4 blocks = 24 MFMAs.  db2 falls out of the same reads.
      const int c = l & 31, kb = l >> 5;
#if ORL_TOWER_F16
      {
        const int LB = LS - (138 - eA);
        u32x4 fb[2][2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          float xb[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) xb[k] = ORL_TOWER_RUNSCALE ? X1[(8 * kb + k) * TS + 32 * b + c] : __builtin_ldexpf(X1[(8 * kb + k) * TS + 32 * b + c], LB);
          split8h(xb, fb[b][0], fb[b][1]);
        }
#pragma unroll
        for (int bo = 0; bo < 2; ++bo) {
          float xa[8];
          u32x4 fa[2];
#pragma unroll
          for (int k = 0; k < 8; ++k) xa[k] = SS[(8 * kb + k) * TS + 32 * bo + c];
          a_db2s[bo] += ORL_TOWER_RUNSCALE ? ((xa[0] + xa[1]) + (xa[2] + xa[3])) + ((xa[4] + xa[5]) + (xa[6] + xa[7]))
                                           : __builtin_ldexpf(((xa[0] + xa[1]) + (xa[2] + xa[3])) + ((xa[4] + xa[5]) + (xa[6] + xa[7])), LB);
          split8h(xa, fa[0], fa[1]);
#pragma unroll
          for (int bi = 0; bi < 2; ++bi) {
            f32x16 g = GS[bo][bi];
            g = mfma_f16_32(fa[1], fb[bi][0], g);
            g = mfma_f16_32(fa[0], fb[bi][1], g);
            g = mfma_f16_32(fa[0], fb[bi][0], g);
            GS[bo][bi] = g;
          }
        }
      }
      if (false) {
#elif ORL_WGRAD_HALVES
      // round 5: the dz2 fragments of ONE 32-row block at a time - 12 registers less at the kernel's pressure peak.  The wide
      // builds (256 registers, wide head + MFMA dW1) spill 10 / 19 registers instead of 20 / 27 with it and every scratch reload
      // of a tile waits vmcnt(0) behind the record DMA: cfg3 pair 0.1448 / 0.1437 / 0.1503 -> 0.1382 / 0.1392 / 0.1388 ms, cfg5
      // 0.4627 / 0.4670 / 0.4631 -> 0.4450 / 0.4487 / 0.4441 ms, cfg2 unchanged (three alternations, profiles/r05_experiments.md)
      {
        u32x4 fb[2][3];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          float xb[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) xb[k] = X1[(8 * kb + k) * TS + 32 * b + c];
          split8(xb, fb[b][0], fb[b][1], fb[b][2]);
        }
#pragma unroll
        for (int bo = 0; bo < 2; ++bo) {
          float xa[8];
          u32x4 fa[3];
#pragma unroll
          for (int k = 0; k < 8; ++k) xa[k] = SS[(8 * kb + k) * TS + 32 * bo + c];
          a_db2s[bo] += ((xa[0] + xa[1]) + (xa[2] + xa[3])) + ((xa[4] + xa[5]) + (xa[6] + xa[7]));
          split8(xa, fa[0], fa[1], fa[2]);
#pragma unroll
          for (int bi = 0; bi < 2; ++bi) {
            f32x16 g = GS[bo][bi];
            ORL_IF_FULL(g = mfma_bf16_32(fa[2], fb[bi][0], g);)
            ORL_IF_FULL(g = mfma_bf16_32(fa[0], fb[bi][2], g);)
            ORL_IF_FULL(g = mfma_bf16_32(fa[1], fb[bi][1], g);)
            g = mfma_bf16_32(fa[1], fb[bi][0], g);
            g = mfma_bf16_32(fa[0], fb[bi][1], g);
            g = mfma_bf16_32(fa[0], fb[bi][0], g);
            GS[bo][bi] = g;
          }
        }
      }
      if (false) {
#else
      {
#endif
      u32x4 fa[2][3], fb[2][3];
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        float xa[8], xb[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          xa[k] = SS[(8 * kb + k) * TS + 32 * b + c];
          xb[k] = X1[(8 * kb + k) * TS + 32 * b + c];
        }
        a_db2s[b] += ((xa[0] + xa[1]) + (xa[2] + xa[3])) + ((xa[4] + xa[5]) + (xa[6] + xa[7]));
        split8(xa, fa[b][0], fa[b][1], fa[b][2]);
        split8(xb, fb[b][0], fb[b][1], fb[b][2]);
      }
      ORL_PRIO2(1);
#ifdef ORL_IGLP_WGRAD  // build-time experiment (round 5): let hipcc's IGroupLP pipeline the splits under the MFMAs of this block
      __builtin_amdgcn_iglp_opt(ORL_IGLP_WGRAD);
#endif
#pragma unroll
      for (int bo = 0; bo < 2; ++bo)
#pragma unroll
        for (int bi = 0; bi < 2; ++bi) {
          f32x16 g = GS[bo][bi];
          ORL_IF_FULL(g = mfma_bf16_32(fa[bo][2], fb[bi][0], g);)
          ORL_IF_FULL(g = mfma_bf16_32(fa[bo][0], fb[bi][2], g);)
          ORL_IF_FULL(g = mfma_bf16_32(fa[bo][1], fb[bi][1], g);)
          g = mfma_bf16_32(fa[bo][1], fb[bi][0], g);
          g = mfma_bf16_32(fa[bo][0], fb[bi][1], g);
          g = mfma_bf16_32(fa[bo][0], fb[bi][0], g);
          GS[bo][bi] = g;
        }
      ORL_PRIO2(0);
      }
    } else
    {
      // wgrad: G += dz2^T xhat1 (operands straight from the slabs in F layout), db2.  Issued as a burst of its own:
      // an fp32 MFMA blocks the issuing wave's VALU for its whole 32 cycles on gfx950 (tools/mfma_valu_overlap.hip),
      // so interleaving these MFMAs with the LN1' chain below buys nothing (measured: 4 862 cycles interleaved vs
      // 2 986 + 1 884 as two phases).
      float s_db = 0.f;
      // operands of k-step s+1 are read before the 16 MFMAs of k-step s (same double buffering as mm64_T)
      float av[2][4], bv[2][4];
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        av[0][m] = SS[q * TS + 16 * m + j];
        bv[0][m] = X1[q * TS + 16 * m + j];
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        if (s < 3) {
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            av[(s + 1) & 1][m] = SS[(4 * (s + 1) + q) * TS + 16 * m + j];
            bv[(s + 1) & 1][m] = X1[(4 * (s + 1) + q) * TS + 16 * m + j];
          }
        }
#pragma unroll
        for (int mo = 0; mo < 4; ++mo)
#pragma unroll
          for (int mi = 0; mi < 4; ++mi) G[mo][mi] = ORL_MFMA(av[s & 1][mo], bv[s & 1][mi], G[mo][mi]);
      }
      const int f = l;
      for (int r = 0; r < TILE_B; ++r) s_db += SS[r * TS + f];
      a_db2 += s_db;
    }
    ORL_T(7);  // wgrad: 64 MFMA from F-layout slab reads, db2
    // dgrad: dn1 = W2^T dz2 straight from the LN2' output registers (no T-layout re-read of the slab), then LN1
    // backward and relu backward -> dz1
    f32x4 d1[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) d1[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    // dgrad A operand: a transposed copy of W2 in LDS read by 16-byte rows (16 reads per 64 MFMA); reading W2 by
    // columns (64 four-byte reads, one per MFMA) made this the slowest phase of the tile: 5 100 vs 3 100 cycles
    if constexpr (F16G) {
      u32x4 xs[2][2];
      split_Th(d2, xs);
      ORL_PRIO2(1);
      if constexpr (SPT) mm64_T_h2_tr((const unsigned short*)(lw + tw.W2), xs, d1, j, q);
      else mm64_T_h2((const unsigned short*)(lw + tw.W2T), xs, d1, j, q);
      ORL_PRIO2(0);
    } else if constexpr (SP) {
      u32x4 xs[2][3];
      split_T(d2, xs);
      ORL_PRIO2(1);
      if constexpr (SPT) mm64_T_split_tr((const unsigned short*)(lw + tw.W2), xs, d1, j, q);
      else mm64_T_split((const unsigned short*)(lw + tw.W2T), xs, d1, j, q);
      ORL_PRIO2(0);
    } else {
      if (w2t) mm64_T(lw + tw.W2T, d2, d1, j, q);
      else mm64_T_wt(lw + tw.W2, d2, d1, j, q);
    }
    ORL_T(8);  // dgrad: 64 MFMA
    {
      f32x4 xh1[4];
      load_slab_T(X1, xh1, j, q);
      // (fp16 builds: d1 = 2^(kw + 138 - eA) W2g^T dz2 - LayerNorm's backward is linear in it, the inverse power rides on rstd1)
      ln_bwd_T(d1, xh1, F16W ? __builtin_ldexpf(rstd1, eA - 138 - kw) : rstd1);
    }
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        d1[m][r] = __builtin_amdgcn_inverse_ballot_w64(relu_mask[4 * m + r]) ? d1[m][r] : 0.f;
    wave_lds_fence();
    store_slab_T(SS, d1, j, q);
    wave_lds_fence();
    ORL_T(9);  // LN1 backward, relu backward, dz1 slab store
    // dW1 += dz1^T x, db1
    if constexpr (W1S) {
      const int c = l & 31, kb = l >> 5;
      // this lane's observation column, clamped into the tower's staged columns (results of columns >= D are dropped)
      const int colx = A.o_x + (c < DP ? c : DP - 1);
      const float* xcol = RTX + ((colx >> 2) << 6) + (colx & 3);
      float xv8[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) xv8[k] = xcol[(8 * kb + k) << 2];
      u32x4 fx[3];
      split8(xv8, fx[0], fx[1], fx[2]);
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        float xa[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) xa[k] = SS[(8 * kb + k) * TS + 32 * b + c];
        a_db1s[b] += ((xa[0] + xa[1]) + (xa[2] + xa[3])) + ((xa[4] + xa[5]) + (xa[6] + xa[7]));
        u32x4 fa[3];
        split8(xa, fa[0], fa[1], fa[2]);
        f32x16 g = G1S[b];
        ORL_IF_FULL(g = mfma_bf16_32(fa[2], fx[0], g);)
        ORL_IF_FULL(g = mfma_bf16_32(fa[0], fx[2], g);)
        ORL_IF_FULL(g = mfma_bf16_32(fa[1], fx[1], g);)
        g = mfma_bf16_32(fa[1], fx[0], g);
        g = mfma_bf16_32(fa[0], fx[1], g);
        g = mfma_bf16_32(fa[0], fx[0], g);
        G1S[b] = g;
      }
    } else {
      const int f = l;
      float s_db = 0.f;
      if (ND == 0) {
        // written on f32x4 / pairs so that it compiles to 2 v_pk_fma_f32 per row and 1 v_pk_add_f32 per two rows (left to its
        // SLP pass hipcc built the pairs with v_mov copies: 37 - 75 VALU for this phase instead of 42)
        f32x4 w4 = f32x4{w1v[0], w1v[1], w1v[2], w1v[3]};
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        f32x2 sd2 = f32x2{0.f, 0.f};
#pragma unroll
        for (int r = 0; r < TILE_B; r += 2) {
          const f32x2 dz = f32x2{SS[r * TS + f], SS[(r + 1) * TS + f]};
          const f32x4 x0 = *(const f32x4*)(&REC_R(r, A.o_x));  // o_x % 4 == 0 on this path
          const f32x4 x1 = *(const f32x4*)(&REC_R(r + 1, A.o_x));
          sd2 += dz;
          w4 = x0 * dz[0] + w4;
          w4 = x1 * dz[1] + w4;
        }
        s_db += sd2[0] + sd2[1];
        w1v[0] = w4[0]; w1v[1] = w4[1]; w1v[2] = w4[2]; w1v[3] = w4[3];
      } else {
        // columns [0, 16 ND) of dW1 on MFMA; a remainder of up to 4 columns (D = 17..20, 33..36: launch_tower_nd) on
        // the VALU like the ND == 0 path instead of a whole extra 16-column accumulator block
        const int n_rem = D - 16 * ND;
        if (n_rem > 0) {
          // all four remainder columns are read and accumulated unconditionally (columns >= D are other, finite or
          // not, fields of the record: their accumulators are never written out) - with a per-column test on n_rem
          // every read was its own branch + LDS round trip: 5 350 cycles of the tile's 30 100 at obs 18
#pragma unroll
          for (int r = 0; r < TILE_B; ++r) {
            const float dzv = SS[r * TS + f];
            s_db += dzv;
#pragma unroll
            for (int k = 0; k < 4; ++k) w1v[k] += dzv * REC_R(r, A.o_x + 16 * ND + k);
          }
        } else {
          for (int r = 0; r < TILE_B; ++r) s_db += SS[r * TS + f];
        }
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
      a_db1 += s_db;
    }
    wave_lds_fence();
    ORL_T(10);  // dW1 / db1
  }
#ifdef ORL_PROF
  if (prof_on && l < 12) atomicAdd(&g_orl_prof[l], prof_lds[l]);
  if (prof_on && l == 12) atomicAdd(&g_orl_prof[12], 1ull);
  if (prof_on && l == 13)  // tiles the probe wave walked (depends on how launch_pair_nd split the CUs)
    atomicAdd(&g_orl_prof[13], (unsigned long long)((n_tiles - wave_g + n_waves - 1) / n_waves));
  if (prof_on && l == 14) atomicAdd(&g_orl_prof[14], t_loop0 - t_entry);  // prologue: entry -> first tile
  const unsigned long long t_loop1 = __builtin_readcyclecounter();
#endif
#undef REC
#undef RECX
#undef REC_R

  if constexpr (F16W) {  // the wgrad accumulators back to their own units (a wave that walked no tile: zeros stay zeros)
#pragma unroll
    for (int a = 0; a < 2; ++a) {
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) GS[a][b][r] = __builtin_ldexpf(GS[a][b][r], -LS);
      a_db2s[a] = __builtin_ldexpf(a_db2s[a], -LS);
    }
  }
  // ---- workgroup reduction of the waves' accumulators, fixed order (deterministic) ---------------------
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the last (unused) DMA before LDS is reused
#ifdef ORL_PROF
  if (prof_on && l == 16) atomicAdd(&g_orl_prof[16], __builtin_readcyclecounter() - t_loop1);  // ... the DMA drain
#endif
  __syncthreads();
#ifdef ORL_PROF
  if (prof_on && l == 17) atomicAdd(&g_orl_prof[17], __builtin_readcyclecounter() - t_loop1);  // ... + waiting for the other waves
#endif
  // The waves' accumulators are combined in a few PARALLEL rounds instead of one serial round per wave: the slab area
  // holds up to NREG full images of the partial row; in round p waves [p*NREG, (p+1)*NREG) each write (p == 0) or add
  // (p > 0) their accumulators into image (wave % NREG), and the images are then summed element-wise in image order.
  // Every element is produced by the same fixed sequence of additions on every launch (deterministic), the 8 serial
  // read-modify-write rounds + zero fill of the first version become 2 rounds + one pass at configuration 2.
  // All of the workgroup's LDS is free now (the tower image is dead too).  One image per wave where that fits (configuration
  // 2 with the launch's LDS rounded up to 8 x 18.4 KB): ONE round of plain stores - no read-modify-write round, no second
  // barrier; otherwise as many images as divide the wave count.
  const int PW = rl.total + ORL_N_STATS;
  const int lds_have = A.lds_floats > 0 ? A.lds_floats : tw.total + nwv * per_wave;
  constexpr int NREG_MAX = 8;  // images the final element-wise sum below requests together (v[NREG_MAX])
  // Round 5 (ORL_EPI_FRAG): the 64 x 64 block G - 4 096 of a partial row's ~4 600 elements, 64 registers per lane - travels in
  // the MFMA's FRAGMENT order: every wave stores its registers as sixteen 16-byte rows into an image of its own (no
  // read-modify-write round, 16 ds_write_b128 per lane instead of 64 scattered ds_write_b32), the final pass sums the waves'
  // images chunk by chunk in wave order (8 ds_read_b128 per chunk) and scatters a chunk's four values to the row-major
  // positions the fragment stands for (for one register, 32 consecutive lanes are 32 consecutive columns: coalesced).  Only
  // the remaining elements (biases, S3, dW1, statistics: PW - 4 096) go through the image rounds below.  rl.oG == 0.
#ifndef ORL_EPI_FRAG
#define ORL_EPI_FRAG 1
#endif
  constexpr int GN = HID * HID;
  // (small-observation builds only: with wide observations / heads the remaining elements are ~1 900 per row and their image
  // rounds double next to 128 KB of fragment images - measured no gain at the cfg3 / cfg5 shapes, - 2 % of the launch at a
  // 512-env shard of configuration 2, profiles/r05_experiments.md)
  const bool frag_g = ORL_EPI_FRAG && SPW && ND == 0 && nwv <= NREG_MAX && lds_have >= nwv * GN + (PW - GN);
  const int PWI = frag_g ? PW - GN : PW;   // floats per image of the rounds below
  const int IOFF = frag_g ? GN : 0;        // row element e lives at image offset e - IOFF
  float* gimg = smem;                      // frag_g: nwv fragment images of G
  float* img0 = smem + (frag_g ? nwv * GN : 0);
  int NREG = (lds_have - (frag_g ? nwv * GN : 0)) / PWI;
  NREG = NREG > nwv ? nwv : NREG < 1 ? 1 : NREG;
  NREG = NREG > NREG_MAX ? NREG_MAX : NREG;
  while (nwv % NREG) --NREG;
  // only the statistics tail has slots no wave writes: zero it in every image, the rest is written by round 0
  for (int e = threadIdx.x; e < NREG * ORL_N_STATS; e += blockDim.x)
    img0[(size_t)(e / ORL_N_STATS) * PWI + rl.total - IOFF + (e % ORL_N_STATS)] = 0.f;
  if constexpr (SPW) {
    if (frag_g) {
      float* gw = gimg + (size_t)wave * GN + l * 4;
#pragma unroll
      for (int bo = 0; bo < 2; ++bo)
#pragma unroll
        for (int bi = 0; bi < 2; ++bi)
#pragma unroll
          for (int rq = 0; rq < 4; ++rq)
            *(f32x4*)(gw + ((bo * 2 + bi) * 4 + rq) * 256) = f32x4{GS[bo][bi][4 * rq], GS[bo][bi][4 * rq + 1],
                                                                   GS[bo][bi][4 * rq + 2], GS[bo][bi][4 * rq + 3]};
    }
  }
  __syncthreads();
  // (VALU reductions: the ds_bpermute form of wave_sum was 6 LDS round trips per sum in front of the image rounds)
  st_active = wave_sum_dpp(st_active); st_rows = wave_sum_dpp(st_rows); st_loss = wave_sum_dpp(st_loss);
  st_ent = wave_sum_dpp(st_ent); st_ratio = wave_sum_dpp(st_ratio);
#if ORL_DB3_ROWLANE
  if constexpr (!HMM) {  // lane c of the wave ends up with db3[c], as the per-tile column sums left it
#pragma unroll
    for (int c = 0; c < NO; ++c) {
      const float tot = wave_sum_dpp(a_db3r[c]);
      if (l == c) a_db3 = tot;
    }
  }
#endif
  float* acc = img0 + (size_t)(wave % NREG) * PWI - IOFF;  // (indexed with row positions >= IOFF only)
  for (int p = 0; p < nwv / NREG; ++p) {
    if (wave / NREG == p) {
      const bool first = p == 0;
      // (later rounds as LDS float atomics instead of read-add-write were measured 4x SLOWER: 18 000 -> 79 000 cycles at
      // configuration 3's two rounds)
      auto put = [&](int idx, float v) { acc[idx] = first ? v : acc[idx] + v; };
      const int f = l;
      if constexpr (SPW) {
        // 32x32 C fragment: lane (c = l & 31, kb = l >> 5), reg r -> G[o = 32bo + (r&3) + 8(r>>2) + 4kb][i = 32bi + c]
        // G is 64 of a lane's ~90 elements.  A later round reads ALL 64 old values first, then adds and stores: element by
        // element (`put`) the compiler kept every read-add-write a dependent LDS round trip - 11 000 cycles for the second
        // round at configuration 3.
        auto g_idx = [&](int bo, int bi, int r) {
          return rl.oG + (32 * bo + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * HID + 32 * bi + (l & 31);
        };
        if (frag_g) {
          // (G went out in fragment order above)
        } else if (first) {
#pragma unroll
          for (int bo = 0; bo < 2; ++bo)
#pragma unroll
            for (int bi = 0; bi < 2; ++bi)
#pragma unroll
              for (int r = 0; r < 16; ++r) acc[g_idx(bo, bi, r)] = GS[bo][bi][r];
        } else {
          float old[2][2][16];
#pragma unroll
          for (int bo = 0; bo < 2; ++bo)
#pragma unroll
            for (int bi = 0; bi < 2; ++bi)
#pragma unroll
              for (int r = 0; r < 16; ++r) old[bo][bi][r] = acc[g_idx(bo, bi, r)];
#pragma unroll
          for (int bo = 0; bo < 2; ++bo)
#pragma unroll
            for (int bi = 0; bi < 2; ++bi)
#pragma unroll
              for (int r = 0; r < 16; ++r) acc[g_idx(bo, bi, r)] = old[bo][bi][r] + GS[bo][bi][r];
        }
        const float d0 = xor32_sum(a_db2s[0]), d1s = xor32_sum(a_db2s[1]);
        if (l < 32) {
          put(rl.odb2 + l, d0);
          put(rl.odb2 + 32 + l, d1s);
        }
      } else {
        // G tiles: lane (c = j, q), reg r -> G[o = 16mo+4q+r][i = 16mi+c]
        auto g_idx = [&](int mo, int mi, int r) { return rl.oG + (16 * mo + 4 * q + r) * HID + 16 * mi + j; };
        if (first) {
#pragma unroll
          for (int mo = 0; mo < 4; ++mo)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
              for (int r = 0; r < 4; ++r) acc[g_idx(mo, mi, r)] = G[mo][mi][r];
        } else {  // all old values first, then add + store (see the split build above)
          float old[4][4][4];
#pragma unroll
          for (int mo = 0; mo < 4; ++mo)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
              for (int r = 0; r < 4; ++r) old[mo][mi][r] = acc[g_idx(mo, mi, r)];
#pragma unroll
          for (int mo = 0; mo < 4; ++mo)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
              for (int r = 0; r < 4; ++r) acc[g_idx(mo, mi, r)] = old[mo][mi][r] + G[mo][mi][r];
        }
        put(rl.odb2 + f, a_db2);
      }
      if constexpr (W1S) {
        const float e0 = xor32_sum(a_db1s[0]), e1 = xor32_sum(a_db1s[1]);
        if (l < 32) {
          put(rl.odb1 + l, e0);
          put(rl.odb1 + 32 + l, e1);
        }
      } else {
        put(rl.odb1 + f, a_db1);
      }
      if constexpr (HMM) {
        // wide heads: 16 more elements per lane, batched like G
        auto s_idx = [&](int mi, int r) { return rl.oS3 + (4 * q + r) * HID + 16 * mi + j; };
        float old[4][4];
        if (!first) {
#pragma unroll
          for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int r = 0; r < 4; ++r) old[mi][r] = 4 * q + r < n_out ? acc[s_idx(mi, r)] : 0.f;
        }
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (4 * q + r < n_out) acc[s_idx(mi, r)] = first ? S3acc[mi][r] : old[mi][r] + S3acc[mi][r];
      } else {
#pragma unroll
        for (int c = 0; c < NO; ++c)
          if (c < n_out) put(rl.oS3 + c * HID + f, a_S3[c < NS3 ? c : 0]);
      }
      if (f < n_out) put(rl.odb3 + f, a_db3);
      if (HEAD == ORL_HEAD_GAUSSIAN && f < n_out) put(rl.odlogstd + f, a_dls);
      if constexpr (W1S) {
        // 32x32 C fragment: lane (c = l & 31, kb = l >> 5), reg r -> dW1[f = 32bo + (r&3) + 8(r>>2) + 4kb][column c]
        if ((l & 31) < D) {  // 32 elements per lane, batched like G
          auto w_idx = [&](int bo, int r) { return rl.odW1 + (32 * bo + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * D + (l & 31); };
          float old[2][16];
          if (!first) {
#pragma unroll
            for (int bo = 0; bo < 2; ++bo)
#pragma unroll
              for (int r = 0; r < 16; ++r) old[bo][r] = acc[w_idx(bo, r)];
          }
#pragma unroll
          for (int bo = 0; bo < 2; ++bo)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[w_idx(bo, r)] = first ? G1S[bo][r] : old[bo][r] + G1S[bo][r];
        }
      } else if (ND == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (k < D) put(rl.odW1 + f * D + k, w1v[k]);
      } else {
#pragma unroll
        for (int mf = 0; mf < 4; ++mf)
#pragma unroll
          for (int mk = 0; mk < NDA; ++mk)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int kk = 16 * mk + j;
              if (kk < D) put(rl.odW1 + (16 * mf + 4 * q + r) * D + kk, G1[mf][mk][r]);
            }
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (16 * ND + k < D) put(rl.odW1 + f * D + 16 * ND + k, w1v[k]);
      }
      if (l == 0) {
        put(rl.total + ST_ACTIVE_SUM, st_active);
        put(rl.total + ST_ROWS, st_rows);
        if (HEAD == ORL_HEAD_VALUE) put(rl.total + ST_VLOSS_SUM, st_loss);
        else {
          put(rl.total + ST_PLOSS_SUM, st_loss);
          put(rl.total + ST_ENT_SUM, st_ent);
          put(rl.total + ST_RATIO_SUM, st_ratio);
        }
      }
    }
    __syncthreads();
  }
#ifdef ORL_PROF
  if (prof_on && l == 18) atomicAdd(&g_orl_prof[18], __builtin_readcyclecounter() - t_loop1);  // ... + the accumulator images
#endif
  float* out = A.partials + (size_t)bid * PW;
  if (frag_g) {
    // chunk k = (register quad Rq = k >> 6, lane ln = k & 63) of the fragment images: the quad's registers 4 Rq .. + 3 of block
    // (bo, bi) = (Rq >> 3, (Rq >> 2) & 1) stand for rows 32 bo + 8 (Rq & 3) + 4 (ln >> 5) + 0 .. 3, column 32 bi + (ln & 31)
    for (int k = threadIdx.x; k < GN / 4; k += blockDim.x) {
      const int Rq = k >> 6, ln = k & 63;
      f32x4 v[NREG_MAX];
#pragma unroll
      for (int w = 0; w < NREG_MAX; ++w)
        v[w] = w < nwv ? *(const f32x4*)(gimg + (size_t)w * GN + Rq * 256 + ln * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
      f32x4 t = v[0];
#pragma unroll
      for (int w = 1; w < NREG_MAX; ++w) t += v[w];  // wave order; + 0 for a wave that is not there
      const int o0 = 32 * (Rq >> 3) + 8 * (Rq & 3) + 4 * (ln >> 5), i = 32 * ((Rq >> 2) & 1) + (ln & 31);
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) out[(o0 + rr) * HID + i] = t[rr];
    }
  }
  // all images of an element are requested together and summed in image order (the rolled form was NREG dependent LDS round
  // trips per element: 6 500 of the epilogue's 12 600 cycles at configuration 2)
#pragma unroll 2
  for (int e = IOFF + threadIdx.x; e < PW; e += blockDim.x) {
    float v[NREG_MAX];
#pragma unroll
    for (int g = 0; g < NREG_MAX; ++g) v[g] = g < NREG ? img0[(size_t)g * PWI + e - IOFF] : 0.f;
    float t = v[0];
#pragma unroll
    for (int g = 1; g < NREG_MAX; ++g) t += v[g];  // + 0.f for an image that is not there
    out[e] = t;
  }
#ifdef ORL_PROF
  if (prof_on && l == 15) atomicAdd(&g_orl_prof[15], __builtin_readcyclecounter() - t_loop1);  // epilogue: last tile -> end
#endif
}

template <int HEAD, int NO, int ND>
__global__ __launch_bounds__(512, 2) void ppo_tower_kernel(PpoArgs A) {
  ppo_tower_body<HEAD, NO, ND>(A, (int)blockIdx.x, (int)gridDim.x);
}

// Both towers of one minibatch in ONE launch (the default 8-wave build): workgroups [0, gp) are the policy tower,
// the rest the critic tower.  No kernel boundary between the two: critic workgroups start on a CU as soon as its
// policy workgroup retires instead of waiting for the slowest policy workgroup of the whole chip.
// ORL_PAIR_WAVES (build-time experiment, orl_ppo.hip): waves per workgroup of the pair launch; 8 = two per SIMD (256 registers
// each), 12 = three per SIMD (168 registers)
#ifndef ORL_PAIR_WAVES
#define ORL_PAIR_WAVES 8
#endif
template <int HEADP, int NOP_, int ND, int SP = 0>
__global__ __launch_bounds__(64 * ORL_PAIR_WAVES, (ORL_PAIR_WAVES + 3) / 4) void ppo_tower_pair_kernel(PpoArgs P, PpoArgs Cc, int gp) {
  if ((int)blockIdx.x < gp) ppo_tower_body<HEADP, NOP_, ND, SP>(P, (int)blockIdx.x, gp);
  else ppo_tower_body<ORL_HEAD_VALUE, 1, ND, SP>(Cc, (int)blockIdx.x - gp, (int)gridDim.x - gp);
}

// LDS bytes needed by `waves` waves of this tower
// ring_nch: chunks per record row in the ring (ring_chunks(); R / 4 when whole records are staged)
inline size_t tower_lds_floats(const orl_net_desc& net, int R, int nop, int waves, bool gaussian, bool w2t = true,
                               bool split = false, int ring_nch = -1) {
  const TowerLds tw(net.obs_dim, net.n_out, gaussian, w2t, nop == 16, split);
  const RawLayout rl(net);
  if (ring_nch < 0) ring_nch = R >> 2;
  const int rts = net.obs_dim <= 4 ? (((R >> 2) + 3) >> 2) * 256 : ring_nch * 64;  // >= what the kernel uses
  const size_t per_wave = 2 * SLAB + 2 * rts + TILE_B * nop;
  size_t fl = (size_t)tw.total + (size_t)waves * per_wave;
  const size_t pw = (size_t)rl.total + ORL_N_STATS;
  const size_t need_acc = pw;                      // the epilogue needs at least one image of the partial row
  const size_t all_waves = (size_t)waves * pw;     // ... and takes one per wave when that still fits the CU's 160 KiB
  if (fl < all_waves && all_waves * sizeof(float) <= 160 * 1024) fl = all_waves;
  return fl > need_acc ? fl : need_acc;
}

}  // namespace orl
