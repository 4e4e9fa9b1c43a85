// orl_rnn.h - layouts and wave-level primitives of the recurrent (GRU) towers for gfx950
// (use_recurrent_policy: openrl/modules/networks/utils/rnn.py:5-99).
//
// Everything stays in the T layout of orl_mlp.h (16 batch rows per wavefront, features down the MFMA M
// dimension), so the GRU's six 64x64 GEMMs chain straight off the trunk's C fragments:
//   r = sigma(Wir x + bir + Whr h + bhr),  z = sigma(Wiz x + biz + Whz h + bhz),
//   n = tanh(Win x + bin + r * (Whn h + bhn)),  h' = (1 - z) * n + z * h          (torch.nn.GRU, gates r,z,n)
#pragma once
#include "orl_common.h"
#include "orl_mlp.h"

namespace orl {

// ---- parameter order of a recurrent tower (reference model.parameters()) ---------------------------------
struct RnnLayout {
  int D, H, n_out, head;
  int oW1, ob1, og1, obe1, oW2, ob2, og2, obe2, oWih, oWhh, obih, obhh, og3, obe3, oW3, ob3, ologstd, total;
  __host__ __device__ RnnLayout() {}
  __host__ __device__ explicit RnnLayout(const orl_net_desc& n) {
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
    oWih = o; o += 3 * H * H;
    oWhh = o; o += 3 * H * H;
    obih = o; o += 3 * H;
    obhh = o; o += 3 * H;
    og3 = o; o += H;
    obe3 = o; o += H;
    oW3 = o; o += n_out * H;
    ob3 = o; o += n_out;
    ologstd = o;
    if (head == ORL_HEAD_GAUSSIAN) o += n_out;
    total = o;
  }
};

// Raw sums the update accumulates for one recurrent tower (delta = gradient w.r.t. a pre-activation):
//   P1[H*D]   = sum dz1 (x) obs                     = dW1
//   S2[H*H]   = sum dz2 (x) xhat1                   (dW2 = g1*S2 + be1*db2;  dg1 = sum_o W2*S2, dbe1 = W2^T db2)
//   S3[3H*H]  = sum [dr,dz,dn] (x) xhat2            (dWih = g2*S3 + be2*dbih; dg2 = sum_o Wih*S3, dbe2 = Wih^T dbih)
//   P4[3H*H]  = sum [dr,dz,dghn] (x) h_in           = dWhh
//   S5[K*H]   = sum dhead (x) xhat3                 (dW3 = g3*S5 + be3*db3;  dg3 = sum_c W3*S5, dbe3 = W3^T db3)
//   db1[H] db2[H] dbih[3H] dbhh[3H] db3[K] | dlogstd[K]
// The LayerNorm-affine gradients are linear images of these (same argument as RawLayout in orl_common.h).
struct RnnRaw {
  int oP1, oS2, oS3, oP4, oS5, odb1, odb2, odbih, odbhh, odb3, odlogstd, total, n_logstd;
  __host__ __device__ RnnRaw() {}
  __host__ __device__ explicit RnnRaw(const orl_net_desc& n) {
    const int H = n.hidden, D = n.obs_dim, K = n.n_out;
    int o = 0;
    oP1 = o; o += H * D;
    oS2 = o; o += H * H;
    oS3 = o; o += 3 * H * H;
    oP4 = o; o += 3 * H * H;
    oS5 = o; o += K * H;
    odb1 = o; o += H;
    odb2 = o; o += H;
    odbih = o; o += 3 * H;
    odbhh = o; o += 3 * H;
    odb3 = o; o += K;
    odlogstd = o;
    n_logstd = (n.head_kind == ORL_HEAD_GAUSSIAN ? K : 0);
    o += n_logstd;
    total = o;
  }
};

// ---- LDS image of a recurrent tower (row kernel): 64x64 blocks padded to W2S columns ----------------------
// Round 6 (ORL_RNN_L2_H2): the seven 64 x 64 matrices of the register-resident L = 2 row kernel as two-term fp16 IMAGES (orl_mlp.h:
// hi = rn16(x), lo = rn16(x - hi), 3 products per fp32 product) instead of fp32 rows under v_mfma_f32_16x16x4_f32.  At the row
// stride RWBS = 72 elements two parts of a matrix are 18 KB (fp32 rows at stride 68: 17 KB) - seven of them fit beside W1 where
// the fp32 rows are; the conflict-free stride 80 (orl_mlp.h: WBS) would not.  Scales: W2 by 2^kw2 from its own maximum, the six GRU
// matrices by ONE 2^kwg (their products are added inside the gates); biases stored scaled.  The forward products' B operands - a
// LayerNorm output after its affine (g xhat + be, |xhat| < 8) and the masked hidden state (|h| < 1) - enter unscaled: fp16's range
// (65 504) is assumed to cover them, i.e. |g| + |be| / 8 < 8 000; the backward products' operands (gate deltas, dz2) are scaled per
// row from their own maximum.
#ifndef ORL_RNN_L2_H2
#define ORL_RNN_L2_H2 1
#endif
// which instances of the L = 2 row kernel take the images (host launch and kernel body agree through this one function)
__host__ __device__ constexpr bool rnn_l2_h2(int head, int no) { return ORL_RNN_L2_H2 != 0 && !(head == ORL_HEAD_GAUSSIAN && no > 4); }
constexpr int RWBS = 72;
constexpr int RIMG_FLOATS = 2 * HID * RWBS / 2;  // floats per matrix image (two parts x 64 rows)
__device__ __forceinline__ int rwb_off(int p, int o, int h, int q) { return (p * HID + o) * RWBS + h * 32 + q * 8; }

struct RnnLds {
  int DP, W1, b1, g1, be1, W2, b2, g2, be2, Wih, Whh, bih, bhh, g3, be3, W3, b3, logstd, W3P, wsc, total;
  __host__ __device__ RnnLds() {}
  // with_w3p: W3 zero padded to [16][W2S] - the MFMA operand of the wide categorical head in the row kernel
  // stream: the seven 64 x 64 matrices (W2, Wih, Whh) are NOT resident - the streamed row kernel (orl_rnn_stream.h) pulls
  // their bf16 images through a ring behind `total`
  // no_w1: W1 is not resident either (the cooperative rollout's critic holds its 16 rows of W1 in registers)
  // h2: W2 / Wih / Whh are fp16 images (RIMG_FLOATS each); wsc = {kw2, 2^kw2, 2^-kw2, 1e-5 x 4^kw2, kwg, 2^kwg, 2^-kwg, -} + 32
  // words of scratch for the staging's maxima
  __host__ __device__ RnnLds(int D, int n_out, bool gaussian, bool with_w3p = false, bool stream = false,
                             bool no_w1 = false, bool h2 = false) {
    const int mat = h2 ? RIMG_FLOATS : HID * W2S;
    DP = (D + 3) & ~3;
    const int no4 = (n_out + 3) & ~3;
    int o = 0;
    W1 = o; o += no_w1 ? 0 : HID * DP;
    b1 = o; o += HID;
    g1 = o; o += HID;
    be1 = o; o += HID;
    W2 = o; o += stream ? 0 : mat;
    b2 = o; o += HID;
    g2 = o; o += HID;
    be2 = o; o += HID;
    Wih = o; o += stream ? 0 : 3 * mat;
    Whh = o; o += stream ? 0 : 3 * mat;
    bih = o; o += 3 * HID;
    bhh = o; o += 3 * HID;
    g3 = o; o += HID;
    be3 = o; o += HID;
    W3 = o; o += no4 * HID;
    b3 = o; o += no4;
    logstd = o; o += gaussian ? no4 : 0;
    W3P = o; o += with_w3p ? 16 * W2S : 0;
    wsc = o; o += h2 ? 40 : 0;
    total = o;
  }
};

__device__ inline void stage_rnn_tower(float* __restrict__ lds, const float* __restrict__ theta, const RnnLayout& tl,
                                       const RnnLds& tw, int tid, int nthreads, bool with_w3p = false,
                                       bool stream = false, bool no_w1 = false, bool h2 = false) {
  const int D = tl.D;
  // fp16 images: the two scales first (one pass over the seven matrices, a wave maximum per wave, one barrier)
  float sc2 = 1.f, scg = 1.f;
  if (h2) {
    float m2 = 0.f, mg = 0.f;
    for (int e = tid; e < HID * HID; e += nthreads) m2 = fmaxf(m2, fabsf(theta[tl.oW2 + e]));
    for (int e = tid; e < 3 * HID * HID; e += nthreads)
      mg = fmaxf(mg, fmaxf(fabsf(theta[tl.oWih + e]), fabsf(theta[tl.oWhh + e])));
    m2 = wave_absmax(m2);
    mg = wave_absmax(mg);
    if ((tid & 63) == 0) {
      lds[tw.wsc + 8 + (tid >> 6)] = m2;
      lds[tw.wsc + 24 + (tid >> 6)] = mg;
    }
    __syncthreads();
    m2 = mg = 0.f;
    for (int w = 0; w < (nthreads + 63) / 64; ++w) {
      m2 = fmaxf(m2, lds[tw.wsc + 8 + w]);
      mg = fmaxf(mg, lds[tw.wsc + 24 + w]);
    }
    auto kof = [](float mx) -> int {  // the power of two that moves mx into [2^13, 2^14)
      const int eb = (int)(f2u(mx) >> 23) & 0xff;
      int k = (eb > 0 && eb < 255) ? 13 - (eb - 127) : 0;
      return k < -40 ? -40 : (k > 40 ? 40 : k);
    };
    const int k2 = kof(m2), kg = kof(mg);
    sc2 = __builtin_ldexpf(1.f, k2);
    scg = __builtin_ldexpf(1.f, kg);
    if (tid == 0) {
      lds[tw.wsc + 0] = (float)k2; lds[tw.wsc + 1] = sc2; lds[tw.wsc + 2] = __builtin_ldexpf(1.f, -k2);
      lds[tw.wsc + 3] = __builtin_ldexpf(1e-5f, 2 * k2);
      lds[tw.wsc + 4] = (float)kg; lds[tw.wsc + 5] = scg; lds[tw.wsc + 6] = __builtin_ldexpf(1.f, -kg);
    }
  }
  if (with_w3p) {
    for (int e = tid; e < 16 * W2S; e += nthreads) {
      const int c = e / W2S, i = e - c * W2S;
      lds[tw.W3P + e] = (c < tl.n_out && i < HID) ? theta[tl.oW3 + c * HID + i] : 0.f;
    }
  }
  for (int e = tid; e < (no_w1 ? 0 : HID * tw.DP); e += nthreads) {
    const int f = e / tw.DP, k = e - f * tw.DP;
    lds[tw.W1 + e] = (k < D) ? theta[tl.oW1 + f * D + k] : 0.f;
  }
  for (int e = tid; e < HID; e += nthreads) {
    lds[tw.b1 + e] = theta[tl.ob1 + e];
    lds[tw.g1 + e] = theta[tl.og1 + e];
    lds[tw.be1 + e] = theta[tl.obe1 + e];
    lds[tw.b2 + e] = theta[tl.ob2 + e] * sc2;
    lds[tw.g2 + e] = theta[tl.og2 + e];
    lds[tw.be2 + e] = theta[tl.obe2 + e];
    lds[tw.g3 + e] = theta[tl.og3 + e];
    lds[tw.be3 + e] = theta[tl.obe3 + e];
  }
  for (int e = tid; e < 3 * HID; e += nthreads) {
    lds[tw.bih + e] = theta[tl.obih + e] * scg;
    lds[tw.bhh + e] = theta[tl.obhh + e] * scg;
  }
  if (h2) {
    // images: element pairs (k even, k + 1) of row o -> one dword per part (split_weight_store2h's addressing at stride RWBS)
    auto image = [&](const float* __restrict__ src, unsigned short* __restrict__ img, int n_rows, float sc) {
      for (int e = tid; e < n_rows * (HID / 2); e += nthreads) {
        const int o = e >> 5, k = 2 * (e & 31);
        const float w0 = src[o * HID + k] * sc, w1 = src[o * HID + k + 1] * sc;
        const int m = k >> 4, qq = (k >> 2) & 3, r = k & 3, h = m >> 1, sl = (m & 1) * 4 + r;
        unsigned short* mat = img + (o >> 6) * (2 * RIMG_FLOATS);  // (o >> 6: which of the stacked matrices; in ushorts)
        const int oo = o & 63;
        const unsigned hi = cvt_pk_f16(w0, w1);
        *(unsigned*)(mat + rwb_off(0, oo, h, qq) + sl) = hi;
        *(unsigned*)(mat + rwb_off(1, oo, h, qq) + sl) = cvt_pk_f16(rem16_lo(hi, w0), rem16_hi(hi, w1));
      }
    };
    image(theta + tl.oW2, (unsigned short*)(lds + tw.W2), HID, sc2);
    image(theta + tl.oWih, (unsigned short*)(lds + tw.Wih), 3 * HID, scg);
    image(theta + tl.oWhh, (unsigned short*)(lds + tw.Whh), 3 * HID, scg);
  } else if (!stream) {
    // The seven 64 x 64 matrices are plain copies with a new row stride: one LDS-DMA instruction per row (a wave's 64 lanes
    // = the row's 64 floats, written at a wave-uniform LDS address), all 448 of a workgroup in flight at once.  The register
    // loops they replace were 8 loads in flight per trip - 7 to 28 global round trips in a row on 512 to 128 threads.
    const int wave = tid >> 6, lane = tid & 63, nw = nthreads >> 6;  // every caller launches whole waves
    // (one loop per matrix with scalar bases: selecting the layout fields by a matrix index sent them through scratch, and
    // every trip then waited vmcnt(0) - for its predecessor's DMA - before it could issue its own)
    auto rows = [&](const float* __restrict__ src, float* __restrict__ dst, int n_rows) {
      for (int r = wave; r < n_rows; r += nw)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + r * HID + lane),
                                         (__attribute__((address_space(3))) void*)(dst + r * W2S), 4, 0, 0);
    };
    rows(theta + tl.oW2, lds + tw.W2, HID);
    rows(theta + tl.oWih, lds + tw.Wih, 3 * HID);
    rows(theta + tl.oWhh, lds + tw.Whh, 3 * HID);
  }
  const int no4 = (tl.n_out + 3) & ~3;
  for (int e = tid; e < no4 * HID; e += nthreads) lds[tw.W3 + e] = (e < tl.n_out * HID) ? theta[tl.oW3 + e] : 0.f;
  for (int e = tid; e < no4; e += nthreads) {
    lds[tw.b3 + e] = (e < tl.n_out) ? theta[tl.ob3 + e] : 0.f;
    if (tl.head == ORL_HEAD_GAUSSIAN) lds[tw.logstd + e] = (e < tl.n_out) ? theta[tl.ologstd + e] : 0.f;
  }
  if (!stream && !h2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMA rows have landed (callers barrier next)
}

// acc += W[64 x 64, row stride S] * in   (T layout; S = W2S for the LDS image, 64 for weights read from global)
template <int S>
__device__ inline void mm64_S(const float* __restrict__ Ws, const f32x4 (&in)[4], f32x4 (&acc)[4], int j, int q) {
  // A operands double-buffered like mm64_T: k-block mi+1's four 16-byte reads are issued before k-block mi's MFMAs
  f32x4 a4[2][4];
#pragma unroll
  for (int mo = 0; mo < 4; ++mo) a4[0][mo] = *(const f32x4*)(Ws + (16 * mo + j) * S + 4 * q);
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    if (mi < 3) {
#pragma unroll
      for (int mo = 0; mo < 4; ++mo)
        a4[(mi + 1) & 1][mo] = *(const f32x4*)(Ws + (16 * mo + j) * S + 16 * (mi + 1) + 4 * q);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int mo = 0; mo < 4; ++mo) acc[mo] = ORL_MFMA(a4[mi & 1][mo][r], in[mi][r], acc[mo]);
    }
  }
}

// acc += W^T in  with W row-major [64][S] read by columns (dgrad)
template <int S>
__device__ inline void mm64_S_wt(const float* __restrict__ Ws, const f32x4 (&in)[4], f32x4 (&acc)[4], int j, int q) {
  // the four column reads of k-step (mi, r) + 1 are issued before the four MFMAs of k-step (mi, r): read by columns
  // every MFMA needs its own 4-byte LDS read, and without the look-ahead each group of 4 opened with an exposed LDS
  // round trip (the GRU dgrad ran at 59 cycles per MFMA)
  float a[2][4];
#pragma unroll
  for (int mo = 0; mo < 4; ++mo) a[0][mo] = Ws[(4 * q) * S + j + 16 * mo];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int mi = k >> 2, r = k & 3;
    if (k < 15) {
      const float* row = Ws + (16 * ((k + 1) >> 2) + 4 * q + ((k + 1) & 3)) * S + j;
#pragma unroll
      for (int mo = 0; mo < 4; ++mo) a[(k + 1) & 1][mo] = row[16 * mo];
    }
#pragma unroll
    for (int mo = 0; mo < 4; ++mo) acc[mo] = ORL_MFMA(a[k & 1][mo], in[mi][r], acc[mo]);
  }
}

// fc1 with W1 read from GLOBAL memory (row stride D, no padding)
template <class XB>
__device__ inline void fc1_g(const float* __restrict__ W1, int D, XB xb, f32x4 (&acc)[4], int j, int q) {
  for (int s = 0; 4 * s < D; ++s) {
    const float b = xb(s);
    const int k = 4 * s + q;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const float a = (k < D) ? W1[(16 * m + j) * D + k] : 0.f;
      acc[m] = ORL_MFMA(a, b, acc[m]);
    }
  }
}

// Gate non-linearities on the hardware transcendental units (v_exp_f32, v_rcp_f32 via __builtin_amdgcn_rcpf -
// __frcp_rn still expands to the IEEE division sequence): 1-ulp instructions, so
// sigma and tanh cost ~5 VALU ops instead of the ~25-35 of libm's expf + IEEE division / tanhf (the GRU evaluates
// 3 x 64 of them per row and step, forward and recomputed forward).  Absolute error <= 3e-7, far inside the
// stated fp32 tolerance of the parity tests.
__device__ inline float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ inline float tanh_f(float x) {
  const float xc = fminf(fmaxf(x, -15.0f), 15.0f);  // tanh(15) == 1 in fp32; keeps exp finite
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * xc));
}

// LayerNorm backward in T layout, in place: d <- rstd * (d*g - mean(d*g) - xhat * mean(d*g*xhat))
__device__ inline void ln_bwd_rnn(f32x4 (&d)[4], const f32x4 (&xhat)[4], const float* __restrict__ g, float rstd,
                                  int q) {
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const f32x4 gg = *(const f32x4*)(g + 16 * m + 4 * q);
    d[m] = d[m] * gg;
  }
  // (round 5: packed in-lane sums and ONE shared cross-lane butterfly for the two statistics, as the feed-forward tower's ln_bwd_T)
  float s1 = lane_sum16(d), s2 = lane_dot16(d, xhat);
  row_allsum2(s1, s2);
  const float c1 = s1 * (1.0f / 64.0f), c2 = s2 * (1.0f / 64.0f);
#pragma unroll
  for (int m = 0; m < 4; ++m) d[m] = (d[m] - c1 - xhat[m] * c2) * rstd;
}

// One GRU step in T layout.  Wih/Whh: three stacked 64 x 64 blocks (r, z, n) with row stride S; bih/bhh [192].
// Outputs the gates (needed by the backward pass) and h' = (1-z)*n + z*hin.
template <int S>
__device__ inline void gru_fwd_T(const float* __restrict__ Wih, const float* __restrict__ Whh,
                                 const float* __restrict__ bih, const float* __restrict__ bhh, const f32x4 (&x)[4],
                                 const f32x4 (&hin)[4], f32x4 (&r)[4], f32x4 (&z)[4], f32x4 (&n)[4], f32x4 (&ghn)[4],
                                 f32x4 (&hnew)[4], int j, int q) {
  f32x4 t[4];
  load_vec_T(bih, q, r);
  load_vec_T(bhh, q, t);
#pragma unroll
  for (int m = 0; m < 4; ++m) r[m] += t[m];
  mm64_S<S>(Wih, x, r, j, q);
  mm64_S<S>(Whh, hin, r, j, q);
  load_vec_T(bih + HID, q, z);
  load_vec_T(bhh + HID, q, t);
#pragma unroll
  for (int m = 0; m < 4; ++m) z[m] += t[m];
  mm64_S<S>(Wih + HID * S, x, z, j, q);
  mm64_S<S>(Whh + HID * S, hin, z, j, q);
  load_vec_T(bih + 2 * HID, q, n);
  mm64_S<S>(Wih + 2 * HID * S, x, n, j, q);
  load_vec_T(bhh + 2 * HID, q, ghn);
  mm64_S<S>(Whh + 2 * HID * S, hin, ghn, j, q);
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
}

// ---- round 6: the same GEMMs over the two-term fp16 images (RnnLds h2) ----------------------------------------------------
// acc += W in / acc += W^T in (T layout) - orl_mlp.h's mm64_T_h2 / mm64_T_h2_tr at the recurrent images' stride
__device__ __forceinline__ void mm64_R_h2(const unsigned short* __restrict__ Wb, const u32x4 (&xs)[2][2], f32x4 (&acc)[4], int j,
                                          int q) {
  u32x4 w[2][2];
#pragma unroll
  for (int p = 0; p < 2; ++p) w[0][p] = *(const u32x4*)(Wb + rwb_off(p, j, 0, q));
#pragma unroll
  for (int st = 0; st < 8; ++st) {
    const int h = st >> 2, mo = st & 3;
    if (st < 7) {
      const int h2 = (st + 1) >> 2, mo2 = (st + 1) & 3;
#pragma unroll
      for (int p = 0; p < 2; ++p) w[(st + 1) & 1][p] = *(const u32x4*)(Wb + rwb_off(p, 16 * mo2 + j, h2, q));
    }
    const u32x4 wh = w[st & 1][0], wl = w[st & 1][1];
    acc[mo] = mfma_f16_16(wl, xs[h][0], acc[mo]);
    acc[mo] = mfma_f16_16(wh, xs[h][1], acc[mo]);
    acc[mo] = mfma_f16_16(wh, xs[h][0], acc[mo]);
  }
}
__device__ __forceinline__ void mm64_R_h2_tr(const unsigned short* __restrict__ Wb, const u32x4 (&xs)[2][2], f32x4 (&acc)[4],
                                             int j, int q) {
  const unsigned short* base = Wb + (4 * q + (j >> 2)) * RWBS + (j & 3) * 8;
  auto frag = [&](int p, int h, int mo) -> u32x4 {
    const unsigned short* a = base + (p * HID + 32 * h) * RWBS + (mo >> 1) * 32 + (mo & 1) * 4;
    const u32x2 lo = ds_read_tr16(a), hi = ds_read_tr16(a + 16 * RWBS);
    return u32x4{lo[0], lo[1], hi[0], hi[1]};
  };
  u32x4 w[2][2];
#pragma unroll
  for (int p = 0; p < 2; ++p) w[0][p] = frag(p, 0, 0);
#pragma unroll
  for (int st = 0; st < 8; ++st) {
    const int h = st >> 2, mo = st & 3;
    if (st < 7) {
#pragma unroll
      for (int p = 0; p < 2; ++p) w[(st + 1) & 1][p] = frag(p, (st + 1) >> 2, (st + 1) & 3);
    }
    const u32x4 wh = w[st & 1][0], wl = w[st & 1][1];
    acc[mo] = mfma_f16_16(wl, xs[h][0], acc[mo]);
    acc[mo] = mfma_f16_16(wh, xs[h][1], acc[mo]);
    acc[mo] = mfma_f16_16(wh, xs[h][0], acc[mo]);
  }
}
// per-ROW scales of gradient vectors (lane (j, q) holds 16 values of batch row j per vector): absmax16 folds a vector into a
// running lane maximum; row_shift turns the row's maximum (over its 4 lanes) into the power of two that moves it into
// [2^11, 2^12); ldexp16 applies it
__device__ __forceinline__ float absmax16(const f32x4 (&v)[4], float mx) {
#pragma unroll
  for (int m = 0; m < 4; ++m) mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v[m][0]), fabsf(v[m][1]))), fmaxf(fabsf(v[m][2]), fabsf(v[m][3])));
  return mx;
}
__device__ __forceinline__ int row_shift(float lane_max) { return 138 - scale_exponent(row_allmax(lane_max)); }
__device__ __forceinline__ void ldexp16(f32x4 (&v)[4], int sh) {
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) v[m][r] = __builtin_ldexpf(v[m][r], sh);
}
// gru_fwd_T over the images: Wih / Whh = three stacked images (r, z, n), bih / bhh scaled by 2^kwg like the images; ginv = 2^-kwg
__device__ inline void gru_fwd_T_h2(const unsigned short* __restrict__ Wih, const unsigned short* __restrict__ Whh,
                                    const float* __restrict__ bih, const float* __restrict__ bhh, const float ginv,
                                    const f32x4 (&x)[4], const f32x4 (&hin)[4], f32x4 (&r)[4], f32x4 (&z)[4], f32x4 (&n)[4],
                                    f32x4 (&ghn)[4], f32x4 (&hnew)[4], int j, int q) {
  constexpr int IMG = 2 * RIMG_FLOATS;  // ushorts per matrix image
  u32x4 xs[2][2], hs[2][2];
  split_Th(x, xs);
  split_Th(hin, hs);
  f32x4 t[4];
  load_vec_T(bih, q, r);
  load_vec_T(bhh, q, t);
#pragma unroll
  for (int m = 0; m < 4; ++m) r[m] += t[m];
  mm64_R_h2(Wih, xs, r, j, q);
  mm64_R_h2(Whh, hs, r, j, q);
  load_vec_T(bih + HID, q, z);
  load_vec_T(bhh + HID, q, t);
#pragma unroll
  for (int m = 0; m < 4; ++m) z[m] += t[m];
  mm64_R_h2(Wih + IMG, xs, z, j, q);
  mm64_R_h2(Whh + IMG, hs, z, j, q);
  load_vec_T(bih + 2 * HID, q, n);
  mm64_R_h2(Wih + 2 * IMG, xs, n, j, q);
  load_vec_T(bhh + 2 * HID, q, ghn);
  mm64_R_h2(Whh + 2 * IMG, hs, ghn, j, q);
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gg = ghn[m][k] * ginv;  // (the backward pass reads W_hn h + b_hn itself)
      const float rr = sigmoid_f(r[m][k] * ginv);
      const float zz = sigmoid_f(z[m][k] * ginv);
      const float nn = tanh_f(n[m][k] * ginv + rr * gg);
      r[m][k] = rr;
      z[m][k] = zz;
      n[m][k] = nn;
      ghn[m][k] = gg;
      hnew[m][k] = (1.0f - zz) * nn + zz * hin[m][k];
    }
}

// base -> GRU -> LayerNorm of one 16-row tile with the tower's LDS image (RnnLds): the rollout-side forward shared by
// rnn_act_lds_kernel and the fused recurrent rollout (orl_rnn_rollout.hip).  xb(s) = obs column 4s + q of this lane's row.
template <class XB>
__device__ inline void rnn_tower_fwd_lds(const float* __restrict__ lw, const RnnLds& tw, XB xb, const f32x4 (&hin)[4],
                                         f32x4 (&hnew)[4], f32x4 (&n3)[4], int j, int q) {
  f32x4 z[4], n1[4], n2[4];
  float rstd;
  load_vec_T(lw + tw.b1, q, z);
  fc1_T(lw + tw.W1, tw.DP, xb, z, j, q);
  relu_T(z);
  ln_normalize_T(z, rstd);
  ln_affine_T(z, lw + tw.g1, lw + tw.be1, q, n1);
  load_vec_T(lw + tw.b2, q, z);
  mm64_T(lw + tw.W2, n1, z, j, q);
  ln_normalize_T(z, rstd);
  ln_affine_T(z, lw + tw.g2, lw + tw.be2, q, n2);
  f32x4 r[4], zz[4], n[4], g[4];
  gru_fwd_T<W2S>(lw + tw.Wih, lw + tw.Whh, lw + tw.bih, lw + tw.bhh, n2, hin, r, zz, n, g, hnew, j, q);
#pragma unroll
  for (int m = 0; m < 4; ++m) z[m] = hnew[m];
  ln_normalize_T(z, rstd);
  ln_affine_T(z, lw + tw.g3, lw + tw.be3, q, n3);
}

// ---- wgrad tape ---------------------------------------------------------------------------------------------
// One block per (tile, step): 10 activation vectors of 16 rows x 64 features, the head deltas (16 x 16) and the
// observation tile (16 x 16*ND).  Element (row, feature f = 16m + 4qq + r) of a 64-wide vector sits at float
//     ((m*4 + qq)*16 + ((row + rot(m*4 + qq)) & 15))*4 + r          (tape_off below)
// i.e. the writer's own register layout with the 16 rows of each 4-feature group rotated: the row kernel's
// stores stay 1 KiB-contiguous per m, and the wgrad kernel's 4-byte MFMA operand reads (16 consecutive features
// x 4 rows per instruction) fall on 32 distinct LDS banks.
constexpr int TV = 1024;  // floats per 64-wide vector
enum { TV_DZ1 = 0, TV_DZ2, TV_DR, TV_DZ, TV_DN, TV_DGHN, TV_XH1, TV_XH2, TV_HIN, TV_XH3, TV_NVEC };
constexpr int TAPE_HEAD = TV_NVEC * TV;  // head deltas: 256 floats
constexpr int TAPE_X = TAPE_HEAD + 256;  // observation m-blocks: ND * 256 floats
__host__ __device__ inline int tape_block_floats(int D) { return TAPE_X + ((D + 15) >> 4) * 256; }

// Round 5: the rotation of 4-feature group g is g & 7 instead of 4 * (g & 3).  The wgrad kernel's bf16 operand reads take 8 rows of
// 32 consecutive features (8 groups) x 2 row halves per instruction: with 4 * (g & 3) they fell on 32 of the 64 LDS banks (2-way
// conflict, SQ_LDS_BANK_CONFLICT 3.7 x SQ_ACTIVE_INST_LDS in profiles/r05_pmc_rnn.txt), with g & 7 the 16 (group, half) pairs
// take 16 different row slots = all 64 banks.  Its fp32-MFMA operand reads (tape_opnd) then walk rows s, s+4, s+8, s+12 per
// k-step instead of 4s .. 4s+3 (tape_krow: 4q + qq is again a bijection onto the 16 slots).  -DORL_TAPE_ROT8=0: rounds 2 - 4.
#ifndef ORL_TAPE_ROT8
#define ORL_TAPE_ROT8 1
#endif
__host__ __device__ constexpr int tape_rot(int g) { return ORL_TAPE_ROT8 ? (g & 7) : 4 * (g & 3); }
// float offset of (group g, row) inside a 64-wide vector (g = 4m + qq) or a 16-wide block (head deltas, observation m-blocks: g = qq)
__host__ __device__ constexpr int tape_off(int g, int row) { return (g * 16 + ((row + tape_rot(g)) & 15)) * 4; }
// the row lane q supplies at k-step s of a 16x16x4 operand read (any fixed bijection of the 16 rows serves both operands alike)
__host__ __device__ constexpr int tape_krow(int s, int q) { return ORL_TAPE_ROT8 ? s + 4 * q : 4 * s + q; }

__device__ inline void tape_store(float* __restrict__ v, const f32x4 (&x)[4], int j, int q) {
#pragma unroll
  for (int m = 0; m < 4; ++m) *(f32x4*)(v + tape_off(m * 4 + q, j)) = x[m];
}

}  // namespace orl
