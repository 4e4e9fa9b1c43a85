// orl_rnn_stream.h - the recurrent ROW kernel with its fourteen 64 x 64 GEMMs per step on the bf16 MFMA (round 4).
// Included by orl_rnn.hip after rnn_row_body (it uses RnnRowArgs / RNN_T and the loss helpers of that translation unit).
//
// rnn_row_body (orl_rnn.hip) forms W2, the GRU's W_ih / W_hh gate blocks and their transposes with
// v_mfma_f32_16x16x4_f32: ~900 VALU-blocking fp32 MFMAs per 16-row tile and step, the M / (M + V) regime the split
// removed from the feed-forward tower.  The three-term bf16 images of the seven matrices are
// 7 x 27.6 KB = 193 KB - they do not fit the 160 KB of LDS next to each other, so they are STREAMED: a small kernel
// splits the weights once per optimiser step into global images (rnn_images_kernel, 14 x 27.6 KB, L2 resident), and the row
// kernel's 8 waves walk their tiles in LOCKSTEP through one cyclic schedule of images,
//     forward sweep step   : W2  Wih_r Wih_z Wih_n  Whh_r Whh_z Whh_n
//     backward sweep step  : the same seven (forward recompute), then Whh_r Wih_r  Whh_z Wih_z  Whh_n  Wih_n  W2 read
//                            TRANSPOSED out of the same images (mm64_T_split_tr, ds_read_b64_tr_b16) - the order that lets
//                            each of dr / dz / dghn / dn / dz2 be split once,
// through a 4-slot LDS ring filled by global_load_lds DMA three chunks ahead.  Chunk m is issued as a whole (27 x 1 KiB) by
// wave m % 8 right after the barrier of chunk m - 3; that wave alone waits for it (s_waitcnt vmcnt(0) at the top of its
// consume(m)), every wave meets at ONE workgroup barrier per GEMM, which is also what frees slot (m - 1) % 4 for chunk
// m + 3.  The B operands (activations / deltas) are split in registers (split_T): n1, n2, h_in per forward, dr, dz, dghn,
// dn, dz2 per backward step.  Arithmetic: 6 of the 9 bf16 products, fp32 accumulation - the feed-forward tower's
// (error <= the fp32 MFMA's own, profiles/r03_split_bf16_gemm.txt); fc1, the heads and everything element-wise are unchanged.
// orl_ppo_hparams.reserved & 4 selects the fp32 kernel (comparison switch, cfg.amd_tower_gemm = fp32).
#pragma once
#include "orl_rnn.h"

namespace orl {

constexpr int RS_NIMG = 7;                     // W2 | Wih r z n | Whh r z n
constexpr int RS_NSLOT = 4;                    // LDS ring slots
constexpr int RS_IMG_FLOATS = WB_IMG_FLOATS;   // 6 912 floats
constexpr int RS_IMG_BYTES = RS_IMG_FLOATS * 4;  // 27 648 B = 27 x 1 KiB
static_assert(RS_IMG_BYTES % 1024 == 0, "an image is a whole number of 1 KiB DMA blocks");

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
  for (int e = threadIdx.x; e < HID * HID; e += blockDim.x) split_weight_store(img, e >> 6, e & 63, W[e]);
}

struct RnnStream {
  const char* gsrc;  // this tower's images in global memory, + lane * 16
  char* lbase;       // ring slot 0 (generic pointer)
  unsigned lds0;     // ... its LDS byte address
  int n, total;      // next chunk to consume; chunks this workgroup consumes in all
  int cyc, nfw;      // chunks per tile iteration: 7 (L - 1) forward-sweep chunks, then 14 per backward step
  int wave;

  __device__ __forceinline__ int img_of(int m) const {
    const unsigned FW = 0x6543210u, BW = 0x0362514u;  // nibble k = image of position k of a forward / backward group
    const int p = m % cyc;
    if (p < nfw) return (int)((FW >> (4 * (p % 7))) & 15u);
    const int u = (p - nfw) % 14;
    return (int)(((u < 7 ? FW >> (4 * u) : BW >> (4 * (u - 7)))) & 15u);
  }
  __device__ __forceinline__ void issue(int m) {  // the whole chunk, by ONE wave (asm: see orl_ppo_tower.h issue_dma)
    const char* src = gsrc + (size_t)img_of(m) * RS_IMG_BYTES;
    const unsigned dst = lds0 + (unsigned)(m % RS_NSLOT) * (unsigned)RS_IMG_BYTES;
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
  }
  __device__ __forceinline__ void start(const float* images, float* ring, int cyc_, int nfw_, int total_, int wave_, int lane) {
    gsrc = (const char*)images + lane * 16;
    lbase = (char*)ring;
    lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lbase;
    n = 0; total = total_; cyc = cyc_; nfw = nfw_; wave = wave_;
    for (int m = 0; m < RS_NSLOT - 1 && m < total; ++m)
      if (wave == (m & 7)) issue(m);
  }
  // chunk n is complete in LDS for every wave; returns its image.  ONE workgroup barrier.
  __device__ __forceinline__ const unsigned short* consume() {
    if (wave == (n & 7)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the issuing wave's DMA has landed
    __syncthreads();
    const int ahead = n + RS_NSLOT - 1;  // its slot held chunk n - 1, which every wave left before this barrier
    if (ahead < total && wave == (ahead & 7)) issue(ahead);
    const unsigned short* r = (const unsigned short*)(lbase + (size_t)(n % RS_NSLOT) * RS_IMG_BYTES);
    ++n;
    return r;
  }
};

template <int HEAD, int NO>
__device__ __forceinline__ void rnn_row_body_stream(const RnnRowArgs& A, const float* __restrict__ images, const int bid,
                                                    const int nblk) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const RnnLayout tl(A.net);
  constexpr bool HMM = HEAD == ORL_HEAD_CATEGORICAL && NO > 4;
  const RnnLds tw(A.net.obs_dim, A.net.n_out, HEAD == ORL_HEAD_GAUSSIAN, HMM, true);
  const int wave = threadIdx.x >> 6, l = threadIdx.x & 63, j = l & 15, q = l >> 4;
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
  st_w.start(images, smem + tw.total, 7 * (L - 1) + 14 * L, 7 * (L - 1), n_iter * (7 * (L - 1) + 14 * L), wave, l);
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

  // acc += image x (forward) / image^T x (backward) for the next image of the stream; xs = split_T of the operand
  auto gemm = [&](const u32x4 (&xs)[2][3], f32x4 (&acc)[4]) { mm64_T_split(st_w.consume(), xs, acc, j, q); };
  auto gemm_t = [&](const u32x4 (&xs)[2][3], f32x4 (&acc)[4]) { mm64_T_split_tr(st_w.consume(), xs, acc, j, q); };

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
        gemm(xs, xh2);  // W2
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
        gemm(xs, r);
        gemm(xs, z);
        gemm(xs, n);
      }
      {
        u32x4 xs[2][3];
        split_T(hin, xs);
        gemm(xs, r);
        gemm(xs, z);
        gemm(xs, ghn);
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

    long long row_c = A.rows[cis];
    float mk_c = A.masks[row_c];

    // ---------------- forward sweep: hidden state entering every step -> htape ----------------
    {
      f32x4 h[4];
      const float* h0 = A.hbuf + (size_t)row_c * HID;
#pragma unroll
      for (int m = 0; m < 4; ++m) h[m] = *(const f32x4*)(h0 + 16 * m + 4 * q);
      for (int s = 0; s < L; ++s) {
        if (tile_ok) {
          float* ht = A.htape + ((size_t)tile * L + s) * TV;
#pragma unroll
          for (int m = 0; m < 4; ++m) *(f32x4*)(ht + (m * 64 + l) * 4) = h[m];
        }
        if (s == L - 1) break;
        float xv_c[16];
        load_x(A.records + (size_t)row_c * A.R, xv_c);
        const long long row_n = A.rows[(size_t)(s + 1) * Nc + cis];
        const float mk_n = A.masks[row_n];
        f32x4 hin[4], xh1[4], xh2[4], n2[4], r[4], z[4], n[4], g[4];
        float r1, r2;
        unsigned rb;
#pragma unroll
        for (int m = 0; m < 4; ++m) hin[m] = h[m] * mk_c;
        trunk(xv_c, xh1, r1, rb, xh2, r2, n2);
        gru(n2, hin, r, z, n, g, h);
        row_c = row_n;
        mk_c = mk_n;
      }
    }

    // ---------------- backward sweep (BPTT), forward recomputed per step ----------------
    f32x4 carry[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) carry[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int s = L - 1; s >= 0; --s) {
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
      {
        const float* ht = A.htape + ((size_t)tile * L + s) * TV;
#pragma unroll
        for (int m = 0; m < 4; ++m) hin[m] = *(const f32x4*)(ht + (m * 64 + l) * 4) * mk;
      }
      if (s > 0) {
        row_c = A.rows[(size_t)(s - 1) * Nc + cis];
        mk_c = A.masks[row_c];
      }
      if (tile_ok) tape_store(tb + TV_HIN * TV, hin, j, q);
      float rstd1, rstd2, rstd3;
      unsigned relu_bits;
      f32x4 gr[4], gz[4], gn[4], ghn[4];
      float dh[NO], dls[NO];
      f32x4 dhv = f32x4{0.f, 0.f, 0.f, 0.f};
      {
        f32x4 xh1[4], xh2[4], n2[4], hnew[4];
        trunk(xv, xh1, rstd1, relu_bits, xh2, rstd2, n2);
        if (tile_ok) {
          tape_store(tb + TV_XH1 * TV, xh1, j, q);
          tape_store(tb + TV_XH2 * TV, xh2, j, q);
        }
        gru(n2, hin, gr, gz, gn, ghn, hnew);
        ln_normalize_T(hnew, rstd3);  // hnew = xhat3
        if (tile_ok) tape_store(tb + TV_XH3 * TV, hnew, j, q);
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
      }
      if (tile_ok) {
        // head deltas and the observation tile -> tape
        f32x4 dv = {0.f, 0.f, 0.f, 0.f};
        if constexpr (HMM) dv = dhv;
        else {
#pragma unroll
          for (int c = 0; c < NO; ++c)
            if ((c >> 2) == q) dv[c & 3] = dh[c];
        }
        *(f32x4*)(tb + TAPE_HEAD + (q * 16 + ((j + 4 * q) & 15)) * 4) = dv;
        for (int m = 0; m < ND; ++m) {
          f32x4 xo;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int f = 16 * m + 4 * q + r;
            xo[r] = f < D ? rec[A.o_x + f] : 0.f;
          }
          *(f32x4*)(tb + TAPE_X + m * 256 + (q * 16 + ((j + 4 * q) & 15)) * 4) = xo;
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
      // xhat3 / xhat2 / xhat1 come back from the tape (through a laundered pointer, as in rnn_row_body: otherwise the
      // stored registers are forwarded to these loads and 48 VGPRs stay live across the GRU).  A shadowing wave reads
      // whatever the tile's owner has or has not written there - its results are dropped.
      {
        const float* tbr = tb;
        asm volatile("" : "+v"(tbr));
        f32x4 xh3[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) xh3[m] = *(const f32x4*)(tbr + TV_XH3 * TV + ((m * 4 + q) * 16 + ((j + 4 * q) & 15)) * 4);
        ln_bwd_rnn(dt, xh3, lw + tw.g3, rstd3, q);
      }
#pragma unroll
      for (int m = 0; m < 4; ++m) dt[m] += carry[m];
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
          carry[m][k] = d * zz;
        }
      if (tile_ok) {
        tape_store(tb + TV_DR * TV, gr, j, q);
        tape_store(tb + TV_DZ * TV, gz, j, q);
        tape_store(tb + TV_DN * TV, gn, j, q);
        tape_store(tb + TV_DGHN * TV, ghn, j, q);
      }
      // dgrad: carry += Whh^T [dr dz dghn], dn2 = Wih^T [dr dz dn]; stream order Whh_r Wih_r Whh_z Wih_z Whh_n Wih_n
      f32x4 d2[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) d2[m] = f32x4{0.f, 0.f, 0.f, 0.f};
      {
        u32x4 xs[2][3];
        split_T(gr, xs);
        gemm_t(xs, carry);
        gemm_t(xs, d2);
      }
      {
        u32x4 xs[2][3];
        split_T(gz, xs);
        gemm_t(xs, carry);
        gemm_t(xs, d2);
      }
      {
        u32x4 xs[2][3];
        split_T(ghn, xs);
        gemm_t(xs, carry);
      }
      {
        u32x4 xs[2][3];
        split_T(gn, xs);
        gemm_t(xs, d2);
      }
#pragma unroll
      for (int m = 0; m < 4; ++m) carry[m] = carry[m] * mk;  // h_in = h * mask
      {
        const float* tbr = tb;
        asm volatile("" : "+v"(tbr));
        f32x4 xh2[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) xh2[m] = *(const f32x4*)(tbr + TV_XH2 * TV + ((m * 4 + q) * 16 + ((j + 4 * q) & 15)) * 4);
        ln_bwd_rnn(d2, xh2, lw + tw.g2, rstd2, q);
      }
      if (tile_ok) tape_store(tb + TV_DZ2 * TV, d2, j, q);
      f32x4 d1[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) d1[m] = f32x4{0.f, 0.f, 0.f, 0.f};
      {
        u32x4 xs[2][3];
        split_T(d2, xs);
        gemm_t(xs, d1);  // W2^T
      }
      {
        const float* tbr = tb;
        asm volatile("" : "+v"(tbr));
        f32x4 xh1[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) xh1[m] = *(const f32x4*)(tbr + TV_XH1 * TV + ((m * 4 + q) * 16 + ((j + 4 * q) & 15)) * 4);
        ln_bwd_rnn(d1, xh1, lw + tw.g1, rstd1, q);
      }
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (!((relu_bits >> (4 * m + r)) & 1u)) d1[m][r] = 0.f;
      if (tile_ok) tape_store(tb + TV_DZ1 * TV, d1, j, q);
    }
    if (!tile_ok) {  // a shadowing wave contributes nothing
      st = st_keep;
#pragma unroll
      for (int c = 0; c < NO; ++c) a_dls[c] = dls_keep[c];
    }
  }

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

// blocks [0, split) = policy tower, [split, gridDim) = critic tower
template <int HEADP, int NOP>
__global__ __launch_bounds__(512, 2) void rnn_row_pair_stream_kernel(RnnRowArgs P, RnnRowArgs Cc, const float* img_p,
                                                                     const float* img_c, int split) {
  if ((int)blockIdx.x < split) rnn_row_body_stream<HEADP, NOP>(P, img_p, blockIdx.x, split);
  else rnn_row_body_stream<ORL_HEAD_VALUE, 1>(Cc, img_c, blockIdx.x - split, gridDim.x - split);
}

}  // namespace orl
