// orl_rnn_l2.h - the recurrent ROW kernel for data_chunk_length == 2 (the reference default, cfg4) with both steps of a
// chunk RESIDENT IN REGISTERS (round 5).  Included by orl_rnn.hip after rnn_row_body (RnnRowArgs, RNN_T, loss helpers).
// Reference: openrl/modules/networks/utils/rnn.py:39-99 (masked GRU over a chunk), openrl/buffers/replay_data.py:1062-1258
// (recurrent_generator: chunks of data_chunk_length steps, the stored state of the chunk's first step), openrl/algorithms/
// ppo.py:363-372.
//
// rnn_row_body walks a chunk as "forward sweep (state tape) + backward sweep with the forward RECOMPUTED per step" at two
// waves per SIMD and 256 registers per wave: 35 64 x 64 GEMMs per 16-chunk tile at L = 2 (step 0 forward twice, and the
// hidden-state dgrad of step 0, whose result nobody reads), x-hat vectors parked on the HBM tape and reloaded behind the
// tape's own stores (vmcnt returns in order: the reload waits for every store in front of it).  With two steps the whole
// chunk fits ONE wave's register file when the wave has a SIMD to itself (4-wave workgroups, 512 registers: 256 VGPRs + 256
// AccVGPRs the allocator moves values through):
//
//   step 0 forward   trunk, GRU, LN3, head, loss  -> keeps x-hat1/2, the gates, h_in, the relu mask and
//                    dt0 = LN3'(W3^T dhead_0) (the loss's own gradient w.r.t. h1)
//   step 1 forward   the same from h1 * mask_1
//   step 1 backward  gate deltas, W_hh^T / W_ih^T dgrad (carry -> h1), LN2', W2^T, LN1', relu'
//   step 0 backward  d = dt0 + carry; gate deltas; W_ih^T dgrad only (the gradient w.r.t. the stored state h0 is not
//                    needed); LN2', W2^T, LN1', relu'
//
// = 7 + 7 + 7 + 4 = 25 GEMMs per tile instead of 35, no forward recompute, no state tape (htape), and NO global load behind a
// tape store inside a tile: the x-hat vectors never leave the registers, the tile's inputs (row indices, records, masks,
// stored states) are requested one tile ahead.  The wgrad tape is written exactly as rnn_row_body writes it (same block
// layout, same values up to summation order), so rnn_wgrad_kernel and everything downstream are unchanged.
// Round 6: the 25 GEMMs run as two-term fp16 splits (3 products on v_mfma_f32_16x16x32_f16) over resident fp16 IMAGES of the seven
// matrices (orl_rnn.h: ORL_RNN_L2_H2, RnnLds h2; per-row power-of-two scales of the gate deltas) - round 5's form, v_mfma_f32_16x16x4_f32
// out of resident fp32 rows, remains for the wide Gaussian instances and under -DORL_RNN_L2_H2=0.
#pragma once
#include "orl_rnn.h"

namespace orl {

// (round 5's experiment with the fp32 LDS rows split on the fly - ORL_RNN_L2_OSPLIT, 0.807 against 0.716 ms per epoch - is gone: round
// 6's fp16 IMAGES, ORL_RNN_L2_H2 in orl_rnn.h, are what the idea needed)

// what one step of one 16-chunk tile reads from global memory (requested a tile ahead)
struct Row2In {
  f32x4 xo[4];        // observation columns 16m + 4q .. + 3 of this lane's row (0 beyond D): fc1's B operands AND the tape tile
  float am, adv, act, lp, vp, rt;  // the record's scalar loss fields
  float mkc[4];       // wide categorical head: action masks of this lane's 4 classes
  float mask;         // masks[row]
  int row;            // the record row (fields outside the prefetched set - Gaussian per-dimension columns - are read through it)
};

template <int HEAD, int NO>
__device__ __forceinline__ void rnn_row2_body(const RnnRowArgs& A, const int bid, const int nblk) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const RnnLayout tl(A.net);
  constexpr bool HMM = HEAD == ORL_HEAD_CATEGORICAL && NO > 4;
  // the seven 64 x 64 matrices as scaled two-term fp16 images (orl_rnn.h); the wide Gaussian instances keep the fp32 rows (they
  // sit at 512 registers already and spill 33 of them with the images' extra state)
  constexpr bool H2 = rnn_l2_h2(HEAD, NO);
  const RnnLds tw(A.net.obs_dim, A.net.n_out, HEAD == ORL_HEAD_GAUSSIAN, HMM, false, false, H2);
  stage_rnn_tower(smem, A.theta, tl, tw, threadIdx.x, blockDim.x, HMM, false, false, H2);
  __syncthreads();
  const float* lw = smem;
  // image scales: fc2 accumulates 2^kw2 z2 (LayerNorm 2 takes it with eps x 4^kw2), the GRU's products 2^kwg (.)
  const int kw2 = H2 ? (int)smem[tw.wsc] : 0, kwg = H2 ? (int)smem[tw.wsc + 4] : 0;
  const float sc2 = H2 ? smem[tw.wsc + 1] : 1.f, ln2_eps = H2 ? smem[tw.wsc + 3] : 1e-5f, ginv = H2 ? smem[tw.wsc + 6] : 1.f;
  const unsigned short* iW2 = (const unsigned short*)(smem + tw.W2);
  const unsigned short* iWih = (const unsigned short*)(smem + tw.Wih);
  const unsigned short* iWhh = (const unsigned short*)(smem + tw.Whh);
  constexpr int IMG = 2 * RIMG_FLOATS;  // ushorts per matrix image
  const int wave = threadIdx.x >> 6, l = threadIdx.x & 63;
  int j = l & 15, q = l >> 4;  // (not const: redefined opaquely at the top of every tile, see the tile loop)
  const int D = A.net.obs_dim, n_out = A.net.n_out, DP = tw.DP;
  const int Nc = A.Nc;
  const orl_ppo_hparams hp = A.hp;
  const int BLK = tape_block_floats(D);
  const int ND = (D + 15) >> 4;

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

  const int n_tiles = (Nc + TILE_B - 1) / TILE_B;
  const int nwv = blockDim.x >> 6;
  const int stride = nblk * nwv;

  // ---- input pipeline: row indices two tiles ahead, everything read through them one tile ahead ---------------------
  // (row indices are kept as 32-bit values: they live in registers across a whole tile - see the prefetch point below - and a
  // spilled one comes back through a scratch load, whose vmcnt(0) drains every tape store in flight)
  auto rows_of = [&](int tile, int& r0, int& r1) {
    const int ci = tile * TILE_B + j;
    const int cis = (tile < n_tiles && ci < Nc) ? ci : 0;  // padding lanes / tiles shadow chunk 0: finite data, zero weight
    r0 = (int)A.rows[cis];
    r1 = (int)A.rows[(size_t)Nc + cis];
  };
  // Every load is unconditional (clamped column) and NOTHING is computed from a loaded value here: a test per column became
  // an exec-mask branch around each global load, and a select right behind the load is a first use - hipcc schedules it
  // where it is written and waits for the load on the spot (vmcnt(0): the tape stores in flight drain).  Columns >= D are
  // zeroed where they are used, with the NEXT tile's (opaquely redefined) lane coordinates: that cannot move up here.
  auto fetch = [&](int row, Row2In& I) {
    const float* rec = A.records + (size_t)row * A.R;
    I.row = row;
    I.mask = A.masks[row];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      I.xo[m] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (m < ND) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int f = 16 * m + 4 * q + r;
          I.xo[m][r] = rec[A.o_x + (f < D ? f : D - 1)];
        }
      }
    }
    I.am = rec[A.cols.o_am]; I.adv = rec[A.cols.o_adv]; I.act = rec[A.cols.o_act];
    I.lp = rec[A.cols.o_lp]; I.vp = rec[A.cols.o_vp]; I.rt = rec[A.cols.o_rt];
#pragma unroll
    for (int r = 0; r < 4; ++r) I.mkc[r] = 1.f;
    if (HEAD == ORL_HEAD_CATEGORICAL && A.cols.K > 0) {  // wide heads: this lane's classes 4q .. 4q + 3; narrow ones: 0 .. 3
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = (HMM ? 4 * q : 0) + r;
        I.mkc[r] = rec[A.cols.o_mk + (c < n_out ? c : n_out - 1)];  // (classes >= n_out are never looked at)
      }
    }
  };
  auto fetch_h = [&](int row, f32x4 (&h)[4]) {
    const float* h0 = A.hbuf + (size_t)row * HID;
#pragma unroll
    for (int m = 0; m < 4; ++m) h[m] = *(const f32x4*)(h0 + 16 * m + 4 * q);
  };

  // ---- pieces of a step ---------------------------------------------------------------------------------------------
  // trunk: fc1 enumerates its reduction index like the 64-wide layers do ((m, r) -> column 16m + 4q + r), so the observation
  // registers are the tape's tile as they are; W1's LDS image is read 16 bytes at a time (columns >= DP of a row run into
  // the next row / the vectors behind W1: finite values that meet x = 0)
  auto trunk = [&](const f32x4 (&xo)[4], f32x4 (&xh1)[4], float& rstd1, unsigned& relu_bits, f32x4 (&xh2)[4], float& rstd2,
                   f32x4 (&n2)[4]) {
    f32x4 n1[4];
    load_vec_T(lw + tw.b1, q, xh1);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      if (m < ND) {
        f32x4 a4[4];
#pragma unroll
        for (int mo = 0; mo < 4; ++mo) a4[mo] = *(const f32x4*)(lw + tw.W1 + (16 * mo + j) * DP + 16 * m + 4 * q);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int mo = 0; mo < 4; ++mo) xh1[mo] = ORL_MFMA(a4[mo][r], xo[m][r], xh1[mo]);
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
    if constexpr (H2) {
      u32x4 xs[2][2];
      split_Th(n1, xs);
      mm64_R_h2(iW2, xs, xh2, j, q);
      ln_normalize_T(xh2, rstd2, ln2_eps);
      rstd2 *= sc2;  // of the unscaled row
    } else {
      mm64_T(lw + tw.W2, n1, xh2, j, q);
      ln_normalize_T(xh2, rstd2);
    }
    ln_affine_T(xh2, lw + tw.g2, lw + tw.be2, q, n2);
  };
  // LN3 + head + loss of a step from its new hidden state; leaves x-hat3, the head deltas (dh / dhv) and
  // dt = LN3'(W3^T dhead) and writes x-hat3 and the head deltas to the step's tape block
  auto head_loss = [&](const f32x4 (&hnew)[4], const Row2In& I, bool valid, float* __restrict__ tb, f32x4 (&dt)[4]) {
    auto recf = [&](int col) -> float {
      if (col == A.cols.o_am) return I.am;
      if (col == A.cols.o_adv) return I.adv;
      if (col == A.cols.o_act) return I.act;
      if (col == A.cols.o_lp) return I.lp;
      if (col == A.cols.o_vp) return I.vp;
      if (col == A.cols.o_rt) return I.rt;
      if (HEAD == ORL_HEAD_CATEGORICAL) {
        const int d = col - (A.cols.o_mk + (HMM ? 4 * q : 0));
        if (d >= 0 && d < 4) return d == 0 ? I.mkc[0] : d == 1 ? I.mkc[1] : d == 2 ? I.mkc[2] : I.mkc[3];
      }
      return A.records[(size_t)I.row * A.R + col];
    };
    f32x4 xh3[4], n3[4];
    float rstd3;
#pragma unroll
    for (int m = 0; m < 4; ++m) xh3[m] = hnew[m];
    ln_normalize_T(xh3, rstd3);
    tape_store(tb + TV_XH3 * TV, xh3, j, q);
    ln_affine_T(xh3, lw + tw.g3, lw + tw.be3, q, n3);
    float dh[NO], dls[NO];
    f32x4 dhv = f32x4{0.f, 0.f, 0.f, 0.f};
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
#pragma unroll
      for (int c = 0; c < NO; ++c) a_dls[c] += q == 0 ? dls[c] : 0.f;  // (a select: no exec-masked region in the tile loop)
    }
    {  // head deltas -> tape (16-wide vector: lane (j, q) owns columns 4q .. 4q + 3)
      f32x4 dv = {0.f, 0.f, 0.f, 0.f};
      if constexpr (HMM) dv = dhv;
      else {
#pragma unroll
        for (int c = 0; c < NO; ++c)
          if ((c >> 2) == q) dv[c & 3] = dh[c];
      }
      *(f32x4*)(tb + TAPE_HEAD + tape_off(q, j)) = dv;
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) dt[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (HMM) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int mo = 0; mo < 4; ++mo) dt[mo] = ORL_MFMA(lw[tw.W3P + (4 * q + r) * W2S + 16 * mo + j], dhv[r], dt[mo]);
    } else {
#pragma unroll
      for (int c = 0; c < NO; ++c) {
        if (c < n_out) {
#pragma unroll
          for (int m = 0; m < 4; ++m) dt[m] += *(const f32x4*)(lw + tw.W3 + c * HID + 16 * m + 4 * q) * dh[c];
        }
      }
    }
    ln_bwd_rnn(dt, xh3, lw + tw.g3, rstd3, q);
  };
  // the observation tile of a step -> its tape block (constant indices: a rolled loop over ND would put xo into scratch memory)
  auto obs_to_tape = [&](const f32x4 (&xo)[4], float* __restrict__ tb) {
#pragma unroll
    for (int m = 0; m < 4; ++m)
      if (m < ND) *(f32x4*)(tb + TAPE_X + m * 256 + tape_off(q, j)) = xo[m];
  };
  // a 64-wide vector this wave stored to its tape block earlier, back into registers (tape_store's layout)
  auto tape_load = [&](const float* __restrict__ v, f32x4 (&x)[4]) {
#pragma unroll
    for (int m = 0; m < 4; ++m) x[m] = *(const f32x4*)(v + tape_off(m * 4 + q, j));
  };
  // GRU cell backward (element-wise): gates -> deltas in place, carry = d * z; the four delta vectors go to the tape
  auto gate_bwd = [&](const f32x4 (&d)[4], const f32x4 (&hin)[4], f32x4 (&gr)[4], f32x4 (&gz)[4], f32x4 (&gn)[4],
                      f32x4 (&ghn)[4], f32x4 (&carry)[4], float* __restrict__ tb) {
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float rr = gr[m][k], zz = gz[m][k], nn = gn[m][k], gg = ghn[m][k], dd = d[m][k], hh = hin[m][k];
        const float dn_pre = dd * (1.0f - zz) * (1.0f - nn * nn);
        const float dz_pre = dd * (hh - nn) * zz * (1.0f - zz);
        const float dr_pre = dn_pre * gg * rr * (1.0f - rr);
        gr[m][k] = dr_pre;
        gz[m][k] = dz_pre;
        gn[m][k] = dn_pre;
        ghn[m][k] = dn_pre * rr;
        carry[m][k] = dd * zz;
      }
    tape_store(tb + TV_DR * TV, gr, j, q);
    tape_store(tb + TV_DZ * TV, gz, j, q);
    tape_store(tb + TV_DN * TV, gn, j, q);
    tape_store(tb + TV_DGHN * TV, ghn, j, q);
  };
  // input-side dgrad of the GRU, then LN2' -> dz2 -> W2^T -> LN1' -> relu' -> dz1 (both to the tape)
  // (H2: the gate deltas arrive scaled by 2^shg per row - row_shift - and every product carries its image's 2^kw: LayerNorm's
  // backward is linear in the incoming gradient, the inverse powers ride on rstd)
  auto trunk_bwd = [&](const f32x4 (&gr)[4], const f32x4 (&gz)[4], const f32x4 (&gn)[4], const f32x4 (&xh1)[4], float rstd1,
                       unsigned relu_bits, const f32x4 (&xh2)[4], float rstd2, float* __restrict__ tb, int shg) {
    f32x4 d2[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) d2[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (H2) {
      u32x4 xs[2][2];
      split_Th(gr, xs);
      mm64_R_h2_tr(iWih, xs, d2, j, q);
      split_Th(gz, xs);
      mm64_R_h2_tr(iWih + IMG, xs, d2, j, q);
      split_Th(gn, xs);
      mm64_R_h2_tr(iWih + 2 * IMG, xs, d2, j, q);
      rstd2 = __builtin_ldexpf(rstd2, -shg - kwg);
    } else {
      mm64_S_wt<W2S>(lw + tw.Wih, gr, d2, j, q);
      mm64_S_wt<W2S>(lw + tw.Wih + HID * W2S, gz, d2, j, q);
      mm64_S_wt<W2S>(lw + tw.Wih + 2 * HID * W2S, gn, d2, j, q);
    }
    ln_bwd_rnn(d2, xh2, lw + tw.g2, rstd2, q);
    tape_store(tb + TV_DZ2 * TV, d2, j, q);
    f32x4 d1[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) d1[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (H2) {
      const int shd = row_shift(absmax16(d2, 0.f));  // (d2 is on the tape already: scaled in place)
      ldexp16(d2, shd);
      u32x4 xs[2][2];
      split_Th(d2, xs);
      mm64_R_h2_tr(iW2, xs, d1, j, q);
      rstd1 = __builtin_ldexpf(rstd1, -shd - kw2);
    } else {
      mm64_S_wt<W2S>(lw + tw.W2, d2, d1, j, q);
    }
    ln_bwd_rnn(d1, xh1, lw + tw.g1, rstd1, q);
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (!((relu_bits >> (4 * m + r)) & 1u)) d1[m][r] = 0.f;
    tape_store(tb + TV_DZ1 * TV, d1, j, q);
  };

#ifdef ORL_PROF
  // Timing build.  The probe of rnn_row_body (lane 0 adds to an LDS counter: an exec-masked region per stamp) FAULTS in this
  // kernel ("Memory access fault", round 5): with 480 live registers the allocator parks values in AccVGPRs wherever it
  // likes, also inside the probe's divergent regions.  Here every lane of the probe wave adds to ITS OWN LDS slot under a
  // wave-uniform branch - no exec-masked region is added to the tile loop.
  __shared__ unsigned rprof_lds[16 * 64];  // (32-bit sums: 4 KB - the fp16 images leave the timing build no room for 8)
  const bool prof_on = __builtin_amdgcn_readfirstlane((int)(blockIdx.x == 0 && wave == 0)) != 0;
  if (prof_on) {
#pragma unroll
    for (int k = 0; k < 16; ++k) rprof_lds[k * 64 + l] = 0u;
  }
  unsigned long long t_last = __builtin_readcyclecounter();
#undef RNN_T
#define RNN_T(k)                                                      \
  do {                                                                \
    if (prof_on) {                                                    \
      const unsigned long long t_now = __builtin_readcyclecounter();  \
      rprof_lds[(k) * 64 + l] += (unsigned)(t_now - t_last);          \
      t_last = t_now;                                                 \
    }                                                                 \
  } while (0)
#endif

  const int tile0 = bid * nwv + wave;
  // Input pipeline.  A wait on a load also waits for everything OLDER in the wave's VMEM queue (vmcnt returns in order), so
  // a load is harmless when it is issued long before its first use and fatal when it is issued right in front of it (the
  // tape stores in flight are drained).  Hence:
  //   row indices      two tiles ahead (rn0 / rn1: two registers that live across the whole tile);
  //   step 0's inputs  (record fields, mask, stored state) one tile ahead, in front of the previous tile's last phase;
  //   step 1's inputs  at the top of their own tile, a whole step-0 forward before their first use.
  int rc1, rn0, rn1;         // row of this tile's step 1; rows of the NEXT tile's two steps
  Row2In In0, In1;           // inputs of the CURRENT tile's two steps
  f32x4 h0[4];
  {
    int r0;
    rows_of(tile0, r0, rc1);
    rows_of(tile0 + stride, rn0, rn1);
    fetch(r0, In0);
    fetch_h(r0, h0);
  }
  for (int tile = tile0; tile < n_tiles; tile += stride) {
    // The tower image never changes inside this loop, and hipcc knows it: left alone it hoists the loop-invariant LDS reads
    // (16 bias / LayerNorm vectors x 16 registers, the first fragments of every GEMM: 54 ds_read_b128 in front of the loop)
    // and holds them across the whole tile - on top of a working set that already fills the file (72 - 240 spilled registers,
    // and a spill's reload drains the tape stores through its vmcnt(0)).  An opaque redefinition of the lane coordinates per
    // tile makes every LDS address loop-variant; immediate offsets from the known base still fold.
    asm volatile("" : "+v"(j), "+v"(q));
    const bool valid = tile * TILE_B + j < Nc;
    fetch(rc1, In1);
    float* tb0 = A.tape + ((size_t)tile * 2 + 0) * BLK;
    float* tb1 = A.tape + ((size_t)tile * 2 + 1) * BLK;
    RNN_T(1);  // input hand-over

    auto zero_pad = [&](f32x4 (&xo)[4]) {  // observation columns >= D (clamped re-reads of column D - 1) -> 0
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) xo[m][r] = (m < ND && 16 * m + 4 * q + r < D) ? xo[m][r] : 0.f;
    };
    // ---------------- step 0 forward ----------------
    zero_pad(In0.xo);
    obs_to_tape(In0.xo, tb0);
    // x-hat1 / x-hat2 of step 0 are NOT held through step 1 (32 registers at the tile's pressure peak): they are on the tape
    // already and come back at the prefetch point, three GEMMs ahead of their first use (h_in is needed at once there: kept)
    f32x4 hin0[4], r0g[4], z0g[4], n0g[4], g0g[4], h1[4], dt0[4];
    float rs1_0, rs2_0;
    unsigned rb0;
    {
      f32x4 n2[4], xh1_0[4], xh2_0[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) hin0[m] = h0[m] * In0.mask;
      tape_store(tb0 + TV_HIN * TV, hin0, j, q);
      trunk(In0.xo, xh1_0, rs1_0, rb0, xh2_0, rs2_0, n2);
      tape_store(tb0 + TV_XH1 * TV, xh1_0, j, q);
      tape_store(tb0 + TV_XH2 * TV, xh2_0, j, q);
      RNN_T(2);  // trunk (fc1 + 64 MFMA) + tape stores
      if constexpr (H2) {
        gru_fwd_T_h2(iWih, iWhh, lw + tw.bih, lw + tw.bhh, ginv, n2, hin0, r0g, z0g, n0g, g0g, h1, j, q);
      } else {
        gru_fwd_T<W2S>(lw + tw.Wih, lw + tw.Whh, lw + tw.bih, lw + tw.bhh, n2, hin0, r0g, z0g, n0g, g0g, h1, j, q);
      }
      RNN_T(3);  // GRU forward: 384 MFMA + gates
      head_loss(h1, In0, valid, tb0, dt0);
      RNN_T(4);  // LN3, head, loss, W3^T dhead, LN3'
    }
    // ---------------- step 1 forward + backward ----------------
    f32x4 carry[4];
    {
      f32x4 hin1[4], xh1_1[4], xh2_1[4], gr[4], gz[4], gn[4], ghn[4], dt1[4];
      float rs1_1, rs2_1;
      unsigned rb1;
      {
        f32x4 n2[4], h2[4];
        zero_pad(In1.xo);
        obs_to_tape(In1.xo, tb1);
#pragma unroll
        for (int m = 0; m < 4; ++m) hin1[m] = h1[m] * In1.mask;
        tape_store(tb1 + TV_HIN * TV, hin1, j, q);
        trunk(In1.xo, xh1_1, rs1_1, rb1, xh2_1, rs2_1, n2);
        tape_store(tb1 + TV_XH1 * TV, xh1_1, j, q);
        tape_store(tb1 + TV_XH2 * TV, xh2_1, j, q);
        RNN_T(2);
        if constexpr (H2) {
          gru_fwd_T_h2(iWih, iWhh, lw + tw.bih, lw + tw.bhh, ginv, n2, hin1, gr, gz, gn, ghn, h2, j, q);
        } else {
          gru_fwd_T<W2S>(lw + tw.Wih, lw + tw.Whh, lw + tw.bih, lw + tw.bhh, n2, hin1, gr, gz, gn, ghn, h2, j, q);
        }
        RNN_T(3);
        head_loss(h2, In1, valid, tb1, dt1);
        RNN_T(4);
      }
      gate_bwd(dt1, hin1, gr, gz, gn, ghn, carry, tb1);
      RNN_T(6);  // GRU element-wise backward + 4 tape vectors
      int shg1 = 0;
      if constexpr (H2) {
        // the four delta vectors (on the tape already) scaled in place by ONE power of two per row; W_hh^T's three products go
        // into an accumulator of their own and join the carry unscaled
        shg1 = row_shift(absmax16(ghn, absmax16(gn, absmax16(gz, absmax16(gr, 0.f)))));
        ldexp16(gr, shg1); ldexp16(gz, shg1); ldexp16(gn, shg1); ldexp16(ghn, shg1);
        f32x4 ct[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) ct[m] = f32x4{0.f, 0.f, 0.f, 0.f};
        u32x4 xs[2][2];
        split_Th(gr, xs);
        mm64_R_h2_tr(iWhh, xs, ct, j, q);
        split_Th(gz, xs);
        mm64_R_h2_tr(iWhh + IMG, xs, ct, j, q);
        split_Th(ghn, xs);
        mm64_R_h2_tr(iWhh + 2 * IMG, xs, ct, j, q);
        ldexp16(ct, -shg1 - kwg);
#pragma unroll
        for (int m = 0; m < 4; ++m) carry[m] += ct[m];
      } else {
        mm64_S_wt<W2S>(lw + tw.Whh, gr, carry, j, q);
        mm64_S_wt<W2S>(lw + tw.Whh + HID * W2S, gz, carry, j, q);
        mm64_S_wt<W2S>(lw + tw.Whh + 2 * HID * W2S, ghn, carry, j, q);
      }
#pragma unroll
      for (int m = 0; m < 4; ++m) carry[m] = carry[m] * In1.mask;  // h_in = h * mask
      RNN_T(7);  // hidden-state dgrad: 192 MFMA (column reads)
      trunk_bwd(gr, gz, gn, xh1_1, rs1_1, rb1, xh2_1, rs2_1, tb1, shg1);
      RNN_T(8);  // W_ih^T dgrad, LN2', W2^T, LN1', relu', tapes
    }
    // The next tile's step-0 inputs are requested HERE, in front of the last phase (4 GEMMs, ~3 us: longer than a global load's
    // round trip), not at the top of the tile: ~110 registers of prefetched values held through both steps pushed the
    // allocation past 512 registers into scratch.  The row indices travel one tile further ahead.
    Row2In Nx0;
    f32x4 hn0[4];
    fetch(rn0, Nx0);
    fetch_h(rn0, hn0);
    rc1 = rn1;
    rows_of(tile + 2 * stride, rn0, rn1);
    f32x4 xh1_0[4], xh2_0[4];
    {
      const float* tbr = tb0;
      asm volatile("" : "+v"(tbr));  // (a laundered pointer: otherwise the stored registers are forwarded and stay live)
      tape_load(tbr + TV_XH1 * TV, xh1_0);
      tape_load(tbr + TV_XH2 * TV, xh2_0);
    }
    // ---------------- step 0 backward ----------------
    {
      f32x4 d[4], cdead[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) d[m] = dt0[m] + carry[m];
      gate_bwd(d, hin0, r0g, z0g, n0g, g0g, cdead, tb0);  // (the carry into the stored state h0 is not needed)
      RNN_T(6);
      int shg0 = 0;
      if constexpr (H2) {
        shg0 = row_shift(absmax16(n0g, absmax16(z0g, absmax16(r0g, 0.f))));
        ldexp16(r0g, shg0); ldexp16(z0g, shg0); ldexp16(n0g, shg0);
      }
      trunk_bwd(r0g, z0g, n0g, xh1_0, rs1_0, rb0, xh2_0, rs2_0, tb0, shg0);
      RNN_T(8);
    }
    // hand the prefetched inputs over
    In0 = Nx0;
#pragma unroll
    for (int m = 0; m < 4; ++m) h0[m] = hn0[m];
  }
#ifdef ORL_PROF
  if (prof_on && l < 12) atomicAdd(&g_rnn_prof[l], (unsigned long long)rprof_lds[l * 64]);
  if (prof_on && l == 12) atomicAdd(&g_rnn_prof[12], 1ull);
  if (prof_on && l == 13) atomicAdd(&g_rnn_prof[13], (unsigned long long)((n_tiles - tile0 + stride - 1) / stride));
  if (prof_on && l == 14) atomicAdd(&g_rnn_prof[14], 1ull);  // marks the L = 2 kernel for tools/rnn_phase_prof.py
#endif

  // ---- workgroup reduction of {dlogstd, stats}: fixed order (as rnn_row_body) ------------------------------------------
  __syncthreads();
  float* acc = smem;  // weights are dead
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

// blocks [0, split) = policy tower, [split, gridDim) = critic tower; 4 waves per workgroup = one wave per SIMD
template <int HEADP, int NOP>
__global__ __launch_bounds__(256, 1) void rnn_row2_pair_kernel(RnnRowArgs P, RnnRowArgs Cc, int split) {
  if ((int)blockIdx.x < split) rnn_row2_body<HEADP, NOP>(P, blockIdx.x, split);
  else rnn_row2_body<ORL_HEAD_VALUE, 1>(Cc, blockIdx.x - split, gridDim.x - split);
}

}  // namespace orl
