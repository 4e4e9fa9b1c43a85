// orl_rnn_stream.h - the recurrent ROW kernel with its fourteen 64 x 64 GEMMs per BPTT step on the bf16 MFMA (round 4).
// Included by orl_rnn.hip after rnn_row_body (it uses RnnRowArgs / RNN_T and the loss helpers of that translation unit).
// NOT the default: cfg.amd_rnn_gemm = split selects it (orl_ppo_hparams.reserved & 4 clear in orl_rnn_ppo_fwd_bwd); the
// default stays rnn_row_body's fp32 MFMAs, which is faster (0.80 vs 0.83 ms per epoch at the cfg4 shape).  Measured why,
// with the stream's waits compiled out (-DORL_RS_NODMA) this kernel runs the epoch in 0.65 ms, with the
// tape stores out too 0.55 - the bf16 GEMMs themselves are worth 0.15 ms, the image stream gives it back.
//
// rnn_row_body forms W2, the GRU's W_ih / W_hh gate blocks and their transposes with v_mfma_f32_16x16x4_f32: ~900
// VALU-blocking fp32 MFMAs per 16-row tile and step.  The three-term bf16 images of the seven matrices are 7 x 27.6 KB =
// 193 KB - they do not fit the 160 KB of LDS next to each other, so they are STREAMED: a small kernel splits the weights
// once per optimiser step into global images (rnn_images_kernel, 14 x 27.6 KB, L2 resident), and the row kernel's waves
// walk their tiles through one cyclic schedule of images,
//     forward group  : W2  Wih_r Wih_z Wih_n  Whh_r Whh_z Whh_n                      (every step phase starts with it)
//     backward group : Whh_r Wih_r  Whh_z Wih_z  Whh_n  Wih_n  W2, read TRANSPOSED out of the same images
//                      (mm64_T_split_tr, ds_read_b64_tr_b16) - the order that lets each of dr / dz / dghn / dn / dz2 be
//                      split once,
// through a 4-slot LDS ring filled by global_load_lds DMA three chunks ahead; readiness and slot reuse are signalled
// through LDS flag rows / counters (RnnStream below) - no workgroup barrier and no vmcnt wait in the tile loop.  A tile
// runs as ONE loop over its 2 L - 1 step phases so that the forward code exists once (instruction cache).  The B operands
// (activations / deltas) are split in registers (split_T): n1, n2, h_in per forward, dr, dz, dghn, dn, dz2 per backward
// step.  Arithmetic: 6 of the 9 bf16 products, fp32 accumulation - the feed-forward tower's (error <= the fp32 MFMA's own,
// profiles/r03_split_bf16_gemm.txt); fc1, the heads and everything element-wise are unchanged.
#pragma once
#include "orl_rnn.h"

namespace orl {

constexpr int RS_NIMG = 7;                     // W2 | Wih r z n | Whh r z n
constexpr int RS_NSLOT = 4;                    // LDS ring slots
constexpr int RS_IMG_FLOATS = WB_IMG_FLOATS;   // 7 680 floats at WBS = 80
// the streamed ring + the widest tower image must fit the CU: checked at build time for the largest supported observation
// (a future WBS change fails here instead of returning ORL_E_UNSUPPORTED at run time)
constexpr int RS_IMG_BYTES = RS_IMG_FLOATS * 4;  // 27 648 B = 27 x 1 KiB
static_assert(RS_IMG_BYTES % 1024 == 0, "an image is a whole number of 1 KiB DMA blocks");
// the ring beside the widest tower's resident arrays (obs 64: W1 64 x 64, ~24 vectors of 64, the 16 x W2S head image) and the
// flag rows must fit the CU's 160 KiB - at build time (WBS 72 -> 80 grew the ring by 12 KB; only the run-time check
// of launch_rnn_rows_stream guarded it)
static_assert((RS_NSLOT * RS_IMG_FLOATS + HID * 64 + 24 * HID + 16 * W2S + RS_NSLOT * 64 + 16) * 4 <= 160 * 1024,
              "the streamed row kernel's LDS ring does not fit beside the widest tower: shrink WBS or RS_NSLOT");
constexpr int RS_SPIN_LIMIT = 1 << 22;         // polls (~100 cycles each) before a wait is declared dead: ~0.2 s

// global images of one optimiser step: grid (7 images, 2 towers), any block size
__global__ __launch_bounds__(256) void rnn_images_kernel(const float* __restrict__ ptheta, const float* __restrict__ ctheta,
                                                         RnnLayout tlp, RnnLayout tlc, float* __restrict__ img_p,
                                                         float* __restrict__ img_c) {
  const int im = blockIdx.x;
  const bool pol = blockIdx.y == 0;
  const float* th = pol ? ptheta : ctheta;
  const RnnLayout& tl = pol ? tlp : tlc;
  unsigned short* img = (unsigned short*)((pol ? img_p : img_c) + (size_t)im * RS_IMG_FLOATS);
  const float* W = th + (im == 0 ? tl.oW2 : im <= 3 ? tl.oWih + (im - 1) * HID * HID : tl.oWhh + (im - 4) * HID * HID);
  for (int e = 2 * threadIdx.x; e < HID * HID; e += 2 * blockDim.x) split_weight_store2(img, e >> 6, e & 63, W[e], W[e + 1]);
  if (im == 0) {  // the two constant rows the stream's READY flags are loaded from (RnnStream::issue)
    unsigned* fr = (unsigned*)((pol ? img_p : img_c) + (size_t)RS_NIMG * RS_IMG_FLOATS);
    for (int e = threadIdx.x; e < 128; e += blockDim.x) fr[e] = (unsigned)(e >> 6);
  }
}

// LDS helpers with explicit instructions (the compiler must neither cache nor reorder these)
__device__ __forceinline__ unsigned rs_lds_peek(unsigned addr) {
  unsigned v;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  return v;
}

// The stream WITHOUT workgroup barriers and WITHOUT vmcnt waits (the first version had one barrier per GEMM and the
// issuing wave draining vmcnt(0) - i.e. every tape store it had in flight - in front of it: 3 700 cycles per GEMM in the
// phase profile against ~1 000 of work, profiles/r04_rnn_phase_prof.txt):
//  * READY: the wave that issues chunk m appends one more DMA load behind the chunk's 27 - 64 dwords of a constant row of
//    the value (m / NSLOT) & 1 into the slot's flag row.  Loads of a wave return in order, so when the flag row shows the
//    expected parity the image is complete; consumers poll the row (one LDS read), nobody waits on vmcnt;
//  * FREE: a wave that has finished its GEMM on a slot adds 1 to the slot's counter (LDS operations of a wave execute in
//    order, so the add follows its last fragment read); the issuer of the slot's next occupant waits for
//    nwaves x (uses so far) before it overwrites the slot.
// The waves of a workgroup therefore run up to NSLOT - 1 chunks apart instead of in lockstep.
struct RnnStream {
  const char* gsrc;   // this tower's images in global memory, + lane * 16
  char* lbase;        // ring slot 0 (generic pointer)
  unsigned lds0;      // ... its LDS byte address (flag rows and counters follow the ring)
  int n, total;       // next chunk to consume; chunks this workgroup consumes in all
  int wave, wmask, nwaves, lane;
  int cur;            // slot of the chunk handed out by the last consume()
  int dead;           // a wait ran into RS_SPIN_LIMIT: the wave went on with whatever the slot held; the kernel then poisons
                      // its loss statistics with NaN, so the failure is in train_info instead of being silent (or a hang)

  static __host__ __device__ constexpr int extra_floats() { return RS_NSLOT * 64 + 16; }  // flag rows + counters
  __device__ __forceinline__ unsigned flags0() const { return lds0 + (unsigned)RS_NSLOT * (unsigned)RS_IMG_BYTES; }
  __device__ __forceinline__ unsigned done0() const { return flags0() + (unsigned)RS_NSLOT * 256u; }
  // Image of the chunk consumed at GEMM site S of a step phase.  Every phase starts with the forward group
  // (W2, Wih r z n, Whh r z n = images 0..6, sites 0..6); a BPTT phase continues with the backward group
  // (Whh_r Wih_r Whh_z Wih_z Whh_n Wih_n W2 = images 4 1 5 2 6 3 0, sites 7..13).  The chunk issued from site S is the one
  // consumed RS_NSLOT - 1 sites later - in this phase, or at the start of the next one (always a forward group).
  static __device__ __forceinline__ constexpr int site_image(int site) {
    return site < 7 ? site : (int)((0x0362514u >> (4 * (site - 7))) & 15u);
  }
  template <int S>
  static __device__ __forceinline__ int ahead_image(bool bwd) {
    constexpr int A = S + RS_NSLOT - 1;
    if constexpr (A < 7) return site_image(A);
    else if constexpr (S < 7) return bwd ? site_image(A) : site_image(A - 7);   // (A < 14 here: S <= 6, NSLOT <= 8)
    else if constexpr (A < 14) return site_image(A);
    else return site_image(A - 14);
  }
  // ring / flag / counter addresses; every thread of the workgroup calls this, then a __syncthreads(), then start()
  __device__ __forceinline__ void init(const float* images, float* ring, int total_, int wave_, int lane_, int nwaves_) {
    gsrc = (const char*)images + lane_ * 16;
    lbase = (char*)ring;
    lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lbase;
    n = 0; total = total_; wave = wave_; wmask = nwaves_ - 1; nwaves = nwaves_; lane = lane_; cur = 0; dead = 0;
    unsigned* fl = (unsigned*)(lbase + (size_t)RS_NSLOT * RS_IMG_BYTES);
    for (int e = wave_ * 64 + lane_; e < RS_NSLOT * 64 + 16; e += nwaves_ * 64) fl[e] = e < RS_NSLOT * 64 ? 1u : 0u;
  }
  __device__ __forceinline__ void issue(int m, int image) {  // the whole chunk + its flag row, by ONE wave
    const int slot = m & (RS_NSLOT - 1), use = m / RS_NSLOT;
    if (use > 0) {  // the slot's previous occupants have been released by every wave
      const unsigned need = (unsigned)(nwaves * use);
      int spins = 0;
#pragma nounroll
      while (__builtin_amdgcn_readfirstlane(rs_lds_peek(done0() + 4u * (unsigned)slot)) < need) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > RS_SPIN_LIMIT) { dead = 1; break; }  // a protocol error must never hang the GPU: see `dead`
      }
    }
    const char* src = gsrc + (size_t)image * RS_IMG_BYTES;
    const unsigned dst = lds0 + (unsigned)slot * (unsigned)RS_IMG_BYTES;
#pragma unroll 1
    for (int off = 0; off < RS_IMG_BYTES; off += 1024) {
      unsigned keep;
      const unsigned m0v = __builtin_amdgcn_readfirstlane(dst + (unsigned)off);
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
                   "s_mov_b32 m0, %0"
                   : "=&s"(keep)
                   : "v"(src + off), "s"(m0v)
                   : "memory");
    }
    {  // the READY row: 64 dwords of (use & 1) from the constant rows behind the images (lane * 4 = (lane * 16) / 4)
      unsigned keep;
      const char* fsrc = gsrc - lane * 12 + (size_t)RS_NIMG * RS_IMG_BYTES + 256 * (use & 1);
      const unsigned m0v = __builtin_amdgcn_readfirstlane(flags0() + 256u * (unsigned)slot);
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\t"
                   "s_mov_b32 m0, %0"
                   : "=&s"(keep)
                   : "v"(fsrc), "s"(m0v)
                   : "memory");
    }
  }
  // WHO issues chunk m: wave m % nwaves.  (Tried: "the first wave that needs it" - one LDS compare-and-swap on a next-chunk
  // counter - 0.849 against 0.827 ms per epoch: the waits are not caused by a lagging issuer)
  __device__ __forceinline__ void start() {  // the first chunks of the first phase: forward group, sites 0 ..
    for (int m = 0; m < RS_NSLOT - 1 && m < total; ++m)
      if (wave == (m & wmask)) issue(m, site_image(m));
  }
  // the image of chunk n (GEMM site S of the current phase), complete in LDS
  template <int S>
  __device__ __forceinline__ const unsigned short* consume(bool bwd) {
#ifdef ORL_RS_NODMA  // TIMING experiment only (wrong results): every GEMM reads whatever the first fill left in its slot
    cur = n & (RS_NSLOT - 1);
    ++n;
    (void)bwd;
    return (const unsigned short*)(lbase + (size_t)cur * RS_IMG_BYTES);
#endif
    const int ahead = n + RS_NSLOT - 1;
    if (ahead < total && wave == (ahead & wmask)) issue(ahead, ahead_image<S>(bwd));
    cur = n & (RS_NSLOT - 1);
    const unsigned want = (unsigned)((n / RS_NSLOT) & 1);
    const unsigned fa = flags0() + 256u * (unsigned)cur + 4u * (unsigned)lane;
    int spins = 0;
#pragma nounroll
    while (__builtin_amdgcn_ballot_w64(rs_lds_peek(fa) == want) != ~0ull) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > RS_SPIN_LIMIT) { dead = 1; break; }
    }
    ++n;
    return (const unsigned short*)(lbase + (size_t)cur * RS_IMG_BYTES);
  }
  // this wave is done with the chunk of the last consume()
  __device__ __forceinline__ void release() {
    if (lane == 0) asm volatile("ds_add_u32 %0, %1" ::"v"(done0() + 4u * (unsigned)cur), "v"(1u) : "memory");
    else asm volatile("" ::: "memory");
  }
};
static_assert((RS_NSLOT & (RS_NSLOT - 1)) == 0 && RS_NSLOT <= 8, "ring slots: a power of two, at most 8");

template <int HEAD, int NO>
__device__ __forceinline__ void rnn_row_body_stream(const RnnRowArgs& A, const float* __restrict__ images, const int bid,
                                                    const int nblk) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const RnnLayout tl(A.net);
  constexpr bool HMM = HEAD == ORL_HEAD_CATEGORICAL && NO > 4;
  const RnnLds tw(A.net.obs_dim, A.net.n_out, HEAD == ORL_HEAD_GAUSSIAN, HMM, true);
  // the wave index as a SCALAR (readfirstlane): everything derived from it - the wave's tile, tile_ok, its DMA turns - is
  // then wave-uniform for the compiler too (s_cbranch instead of exec masking around every predicated store)
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int l = threadIdx.x & 63, j = l & 15, q = l >> 4;
  const int D = A.net.obs_dim, n_out = A.net.n_out, DP = tw.DP;
  const int Nc = A.Nc, L = A.L;
  const orl_ppo_hparams hp = A.hp;
  const int BLK = tape_block_floats(D);
  const int ND = (D + 15) >> 4;
  const int n_tiles = (Nc + TILE_B - 1) / TILE_B;
  const int nwv = blockDim.x >> 6;
  const int per = nblk * nwv;
  // every wave of a workgroup runs the same number of tile iterations (the stream is consumed in lockstep): the
  // workgroup's tiles of iteration `it` are it * per + bid * nwv + [0, nwv); only the last group can be ragged
  int n_iter = 0;
  if (bid * nwv < n_tiles) n_iter = (n_tiles - bid * nwv + per - 1) / per;
  RnnStream st_w;
  st_w.init(images, smem + tw.total, n_iter * (7 * (L - 1) + 14 * L), wave, l, nwv);
  __syncthreads();  // flag rows / counters initialised before the first DMA may land in them
  st_w.start();
  stage_rnn_tower(smem, A.theta, tl, tw, threadIdx.x, blockDim.x, HMM, true);
  __syncthreads();
  const float* lw = smem;

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
#ifdef ORL_PROF
  __shared__ unsigned long long rprof_lds[16];
  const bool prof_on = blockIdx.x == 0 && wave == 0;
  if (prof_on && l < 16) rprof_lds[l] = 0ull;
  unsigned long long t_last = __builtin_readcyclecounter();
#endif

  // acc += image x (forward) / image^T x (backward) for the next image of the stream; xs = split_T of the operand
  // `site` (GtIdx<S>): the GEMM's position in the phase's image schedule (RnnStream::site_image); bwd_phase: the current
  // phase is a BPTT step (set at the top of the phase loop)
  bool bwd_phase = false;
  auto gemm = [&](auto site, const u32x4 (&xs)[2][3], f32x4 (&acc)[4]) {
    mm64_T_split(st_w.template consume<decltype(site)::value>(bwd_phase), xs, acc, j, q);
    st_w.release();
  };
  auto gemm_t = [&](auto site, const u32x4 (&xs)[2][3], f32x4 (&acc)[4]) {
    mm64_T_split_tr(st_w.template consume<decltype(site)::value>(bwd_phase), xs, acc, j, q);
    st_w.release();
  };

  for (int it = 0; it < n_iter; ++it) {
    int tile = it * per + bid * nwv + wave;
    const bool tile_ok = tile < n_tiles;  // ragged last group: the wave shadows the last tile - all loads, NO stores,
    if (!tile_ok) tile = n_tiles - 1;     // and its loss statistics are dropped below
    const LossStats st_keep = st;
    float dls_keep[NO];
#pragma unroll
    for (int c = 0; c < NO; ++c) dls_keep[c] = a_dls[c];
    const int ci = tile * TILE_B + j;
    const bool valid = tile_ok && ci < Nc;
    const int cis = ci < Nc ? ci : 0;  // padding lanes shadow chunk 0: finite data, zero loss weight
    auto load_x = [&](const float* rec, float (&xv)[16]) {
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const int c = 4 * k + q;
        xv[k] = (4 * k < DP && c < D) ? rec[A.o_x + c] : 0.f;
      }
    };
    // trunk forward of one row (fc1 on the fp32 MFMA out of LDS, fc2 on the streamed image)
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
      {
        u32x4 xs[2][3];
        split_T(n1, xs);
        gemm(GtIdx<0>{}, xs, xh2);  // W2
      }
      ln_normalize_T(xh2, rstd2);
      ln_affine_T(xh2, lw + tw.g2, lw + tw.be2, q, n2);
    };
    // one GRU step (gru_fwd_T's arithmetic; images Wih r z n, then Whh r z n)
    auto gru = [&](const f32x4 (&x)[4], const f32x4 (&hin)[4], f32x4 (&r)[4], f32x4 (&z)[4], f32x4 (&n)[4],
                   f32x4 (&ghn)[4], f32x4 (&hnew)[4]) {
      f32x4 t[4];
      load_vec_T(lw + tw.bih, q, r);
      load_vec_T(lw + tw.bhh, q, t);
#pragma unroll
      for (int m = 0; m < 4; ++m) r[m] += t[m];
      load_vec_T(lw + tw.bih + HID, q, z);
      load_vec_T(lw + tw.bhh + HID, q, t);
#pragma unroll
      for (int m = 0; m < 4; ++m) z[m] += t[m];
      load_vec_T(lw + tw.bih + 2 * HID, q, n);
      load_vec_T(lw + tw.bhh + 2 * HID, q, ghn);
      {
        u32x4 xs[2][3];
        split_T(x, xs);
        gemm(GtIdx<1>{}, xs, r);
        gemm(GtIdx<2>{}, xs, z);
        gemm(GtIdx<3>{}, xs, n);
      }
      {
        u32x4 xs[2][3];
        split_T(hin, xs);
        gemm(GtIdx<4>{}, xs, r);
        gemm(GtIdx<5>{}, xs, z);
        gemm(GtIdx<6>{}, xs, ghn);
      }
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float rr = sigmoid_f(r[m][k]);
          const float zz = sigmoid_f(z[m][k]);
          const float nn = tanh_f(n[m][k] + rr * ghn[m][k]);
          r[m][k] = rr;
          z[m][k] = zz;
          n[m][k] = nn;
          hnew[m][k] = (1.0f - zz) * nn + zz * hin[m][k];
        }
    };

    // ONE loop over the 2 L - 1 step phases of a tile - forward-only steps s = 0 .. L-2 (the hidden state entering every step
    // goes to htape), then the BPTT steps s = L-1 .. 0 with the forward recomputed - so that the forward code (trunk, GRU:
    // 7 GEMM sites, 3 operand splits, the gates) exists ONCE in the kernel.  With a separate forward sweep the tile loop
    // was ~64 KB of instructions per tower body: it did not fit the 64 KB instruction cache two CUs share, and every GEMM
    // took 3 700 - 5 000 cycles instead of ~1 000 (profiles/r04_rnn_phase_prof.txt).
    // `state`: the running hidden state during the forward phases, the gradient carried back in time (dt * z + Whh^T ...)
    // during the backward phases - never both.
    auto step_of = [&](int ph) -> int { return ph >= L - 1 ? 2 * L - 2 - ph : ph; };
    const int n_ph = 2 * L - 1;
    long long row_c = A.rows[(size_t)step_of(0) * Nc + cis];
    float mk_c = A.masks[row_c];
    f32x4 state[4];
    {
      const float* h0 = A.hbuf + (size_t)row_c * HID;  // (L > 1: the row of step 0; L == 1: the only row)
#pragma unroll
      for (int m = 0; m < 4; ++m) state[m] = *(const f32x4*)(h0 + 16 * m + 4 * q);
    }
#pragma unroll 1
    for (int ph = 0; ph < n_ph; ++ph) {
      const bool bwd = ph >= L - 1;
      bwd_phase = bwd;
      const int s = step_of(ph);
      const long long row = row_c;
      const float* rec = A.records + (size_t)row * A.R;
      const float mk = mk_c;
      float xv[16];
      load_x(rec, xv);
      const float rf_am = rec[A.cols.o_am], rf_adv = rec[A.cols.o_adv], rf_act = rec[A.cols.o_act];
      const float rf_lp = rec[A.cols.o_lp], rf_vp = rec[A.cols.o_vp], rf_rt = rec[A.cols.o_rt];
      float rf_mk[4] = {1.f, 1.f, 1.f, 1.f};
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
      // a shadowing wave (ragged last group) must not touch the tape: every store below is predicated on tile_ok
      float* tb = A.tape + ((size_t)tile * L + s) * BLK;
      f32x4 hin[4];
      if (bwd && s < L - 1) {  // the state entering step s, left by the forward phases
        const float* ht = A.htape + ((size_t)tile * L + s) * TV;
#pragma unroll
        for (int m = 0; m < 4; ++m) hin[m] = *(const f32x4*)(ht + (m * 64 + l) * 4) * mk;
      } else {                 // forward phases, and the first backward step (s = L - 1): the running state
        if (!bwd && tile_ok) {
          float* ht = A.htape + ((size_t)tile * L + s) * TV;
#pragma unroll
          for (int m = 0; m < 4; ++m) *(f32x4*)(ht + (m * 64 + l) * 4) = state[m];
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) hin[m] = state[m] * mk;
        if (bwd) {             // from here on `state` is the carried gradient
#pragma unroll
          for (int m = 0; m < 4; ++m) state[m] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
      if (ph + 1 < n_ph) {  // the next phase's row and mask, one phase ahead
        row_c = A.rows[(size_t)step_of(ph + 1) * Nc + cis];
        mk_c = A.masks[row_c];
      }
      RNN_T(1);  // step inputs: record / state-tape loads
      if (bwd && tile_ok) tape_store(tb + TV_HIN * TV, hin, j, q);
      float rstd1, rstd2, rstd3;
      unsigned relu_bits;
      f32x4 gr[4], gz[4], gn[4], ghn[4];
      float dh[NO], dls[NO];
      f32x4 dhv = f32x4{0.f, 0.f, 0.f, 0.f};
      f32x4 xh3[4];  // xhat3 stays in registers from LN3 to its backward (head + loss in between: low pressure) - its
                     // reload from the tape queued behind the tape stores just issued (vmcnt returns in order)
      {
        f32x4 xh1[4], xh2[4], n2[4], hnew[4];
        trunk(xv, xh1, rstd1, relu_bits, xh2, rstd2, n2);
        if (bwd && tile_ok) {
          tape_store(tb + TV_XH1 * TV, xh1, j, q);
          tape_store(tb + TV_XH2 * TV, xh2, j, q);
        }
        RNN_T(2);  // trunk (fc1 + W2) + tape stores
        gru(n2, hin, gr, gz, gn, ghn, hnew);
        RNN_T(3);  // GRU forward: 6 GEMMs + gates
        if (!bwd) {  // a forward-only phase ends here: the new state enters step s + 1
#pragma unroll
          for (int m = 0; m < 4; ++m) state[m] = hnew[m];
        } else {
        ln_normalize_T(hnew, rstd3);  // hnew = xhat3
        if (tile_ok) tape_store(tb + TV_XH3 * TV, hnew, j, q);
#pragma unroll
        for (int m = 0; m < 4; ++m) xh3[m] = hnew[m];
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
        }  // bwd: LN3, head, loss
      }
      if (bwd) {  // ---- the backward half of a BPTT step
      RNN_T(4);  // LN3, head, loss
      if (tile_ok) {
        // head deltas and the observation tile -> tape
        f32x4 dv = {0.f, 0.f, 0.f, 0.f};
        if constexpr (HMM) dv = dhv;
        else {
#pragma unroll
          for (int c = 0; c < NO; ++c)
            if ((c >> 2) == q) dv[c & 3] = dh[c];
        }
        *(f32x4*)(tb + TAPE_HEAD + tape_off(q, j)) = dv;
        for (int m = 0; m < ND; ++m) {
          f32x4 xo;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int f = 16 * m + 4 * q + r;
            xo[r] = f < D ? rec[A.o_x + f] : 0.f;
          }
          *(f32x4*)(tb + TAPE_X + m * 256 + tape_off(q, j)) = xo;
        }
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
      // (xhat2 / xhat1 come back from the tape further down, through a laundered pointer as in rnn_row_body: otherwise the
      // stored registers are forwarded to those loads and 32 VGPRs stay live across the GRU.  A shadowing wave reads
      // whatever the tile's owner has or has not written there - its results are dropped.)
      ln_bwd_rnn(dt, xh3, lw + tw.g3, rstd3, q);
#pragma unroll
      for (int m = 0; m < 4; ++m) dt[m] += state[m];
      RNN_T(5);  // dhead / obs tape, W3^T dhead, LN3 backward
      // GRU cell backward (element-wise part): gr/gz/gn/ghn become dr/dz/dn/dghn, carry collects dt * z
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
          state[m][k] = d * zz;
        }
      if (tile_ok) {
        tape_store(tb + TV_DR * TV, gr, j, q);
        tape_store(tb + TV_DZ * TV, gz, j, q);
        tape_store(tb + TV_DN * TV, gn, j, q);
        tape_store(tb + TV_DGHN * TV, ghn, j, q);
      }
      RNN_T(6);  // GRU elementwise backward + 4 tape vectors
      // dgrad: state (the carried gradient) += Whh^T [dr dz dghn], dn2 = Wih^T [dr dz dn]; stream order Whh_r Wih_r Whh_z Wih_z Whh_n Wih_n
      f32x4 d2[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) d2[m] = f32x4{0.f, 0.f, 0.f, 0.f};
      {
        u32x4 xs[2][3];
        split_T(gr, xs);
        gemm_t(GtIdx<7>{}, xs, state);
        gemm_t(GtIdx<8>{}, xs, d2);
      }
      {
        u32x4 xs[2][3];
        split_T(gz, xs);
        gemm_t(GtIdx<9>{}, xs, state);
        gemm_t(GtIdx<10>{}, xs, d2);
      }
      {
        u32x4 xs[2][3];
        split_T(ghn, xs);
        gemm_t(GtIdx<11>{}, xs, state);
      }
      {
        u32x4 xs[2][3];
        split_T(gn, xs);
        gemm_t(GtIdx<12>{}, xs, d2);
      }
#pragma unroll
      for (int m = 0; m < 4; ++m) state[m] = state[m] * mk;  // h_in = h * mask
      RNN_T(7);  // GRU dgrad: 6 transposed GEMMs
      {
        const float* tbr = tb;
        asm volatile("" : "+v"(tbr));
        f32x4 xh2[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) xh2[m] = *(const f32x4*)(tbr + TV_XH2 * TV + tape_off(m * 4 + q, j));
        ln_bwd_rnn(d2, xh2, lw + tw.g2, rstd2, q);
      }
      if (tile_ok) tape_store(tb + TV_DZ2 * TV, d2, j, q);
      f32x4 d1[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) d1[m] = f32x4{0.f, 0.f, 0.f, 0.f};
      {
        u32x4 xs[2][3];
        split_T(d2, xs);
        gemm_t(GtIdx<13>{}, xs, d1);  // W2^T
      }
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
      if (tile_ok) tape_store(tb + TV_DZ1 * TV, d1, j, q);
      RNN_T(8);  // LN2 backward, W2 dgrad, LN1 / relu backward, tapes
      }  // if (bwd)
    }
    if (!tile_ok) {  // a shadowing wave contributes nothing
      st = st_keep;
#pragma unroll
      for (int c = 0; c < NO; ++c) a_dls[c] = dls_keep[c];
    }
  }

#ifdef ORL_PROF
  if (prof_on && l < 12) atomicAdd(&g_rnn_prof[l], rprof_lds[l]);
  if (prof_on && l == 12) atomicAdd(&g_rnn_prof[12], 1ull);
  if (prof_on && l == 13) atomicAdd(&g_rnn_prof[13], (unsigned long long)n_iter);
#endif
  if (st_w.dead) st.loss = u2f(0x7fc00000u);  // a stream wait timed out (RnnStream::dead): NaN into train_info
  // ---- workgroup reduction of {dlogstd, stats}: fixed order (as rnn_row_body) ----
  __syncthreads();
  float* acc = smem;  // the resident image is dead, the ring has no DMA in flight (every issued chunk was consumed)
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

// blocks [0, split) = policy tower, [split, gridDim) = critic tower.  NW = 8: two waves per SIMD, 256 registers each;
// NW = 4: ONE wave per SIMD with the whole 512-entry register budget (256 VGPRs + 256 AccVGPRs the allocator spills into
// instead of scratch memory) - the row kernel's hot loop has ~170 live vector registers per wave before address
// arithmetic, and a scratch reload is a VMEM operation that queues behind the tape stores (vmcnt is in order)
template <int HEADP, int NOP, int NW>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 1 : 2) void rnn_row_pair_stream_kernel(RnnRowArgs P, RnnRowArgs Cc,
                                                                                    const float* img_p,
                                                                                    const float* img_c, int split) {
  if ((int)blockIdx.x < split) rnn_row_body_stream<HEADP, NOP>(P, img_p, blockIdx.x, split);
  else rnn_row_body_stream<ORL_HEAD_VALUE, 1>(Cc, img_c, blockIdx.x - split, gridDim.x - split);
}

}  // namespace orl
