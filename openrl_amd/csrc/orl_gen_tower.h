// orl_gen_tower.h - cross-layer FUSED update kernels of the general feed-forward towers (hidden_size 64 / 128, any
// layer_N the registers hold, four activations, feature norm, one or two heads): the shapes outside the default tower
// that orl_gen_fused.hip runs layer by layer with every activation array crossing HBM.
//
// What is computed: MLPBase.forward (openrl/modules/networks/utils/mlp.py:8-48, 100-180) + the Linear heads of
// ACTLayer / v_out on the rows idx[0..mb) of the update records, and the backward of that chain.
//
//   orl_gt_prep  theta -> "image": LayerNorm affines folded into the next Linear (exact algebra:
//                W (xhat g + be) + b = (W diag g) xhat + (b + W be)), the H x H matrices as three-term bf16 split
//                images (orl_mlp.h) of W' and W'^T in the chunk order the kernels stream them
//   gt_fwd_kernel<H, NL, ND>  forward: rows -> head outputs (logits / values); activations never leave the chip
//   gt_bwd_kernel<H, NL, ND, NW>  backward: recomputes the forward of its rows, takes d loss / d head outputs and
//                accumulates the RAW gradient sums G_l = dz_l^T xhat_{l-1}, db_l = sum dz_l in registers
//   gt_finalize  raw sums -> gradients in the parameter vector's layout (dW = G diag g + db be^T, d g_l / d be_l as
//                linear images of the next layer's sums - the relations of orl_common.h's RawLayout)
//
// Work distribution.  A workgroup = 8 waves = one CU; a pass = 128 rows, 16 per wave.  Within a wave the rows live in
// the T layout of orl_mlp.h (features down the MFMA M dimension, the 16 rows across N): a layer's C fragment is the
// next layer's B operand, LayerNorm statistics are in-lane sums + two permlane swaps.  The H x H GEMMs (forward and
// input gradient) run on v_mfma_f32_16x16x32_bf16 over exact three-term splits - the six largest of the nine partial
// products, error below the fp32 MFMA's own (profiles/r03_split_bf16_gemm.txt).  Their A operands (the bf16 images)
// do not fit LDS beside everything else at H = 128, so they are STREAMED: 32 output rows of the three parts per chunk
// (26 KB at H = 128), a linear global -> LDS DMA (global_load_lds_dwordx4) one chunk ahead into a double buffer, one
// workgroup barrier per chunk; all 8 waves consume the same chunk, so a CU reads each image once per pass from L2.
// The weight gradients cannot stay wave-private at H = 128 (128 x 128 accumulators = 256 VGPRs): they are COOPERATIVE -
// wave w owns the output-feature slice [16 os, 16 os + 16) x an input-feature part of every G_l, the pass's dz_l and
// xhat_{l-1} go through one [feature][row] LDS slab (first dz_l: every wave takes its slice as 32 A registers, then
// xhat_{l-1}: streamed as the B operand), products on v_mfma_f32_16x16x4_f32 with the 128 rows as K.
#pragma once
#include "orl_common.h"
#include "orl_mlp.h"
#include "orl_gen_act.h"
#include "orl_gen_loss.h"

namespace orl {

constexpr int GT_WAVES = 8;          // waves per workgroup of the forward kernel (and of the backward's 8-wave build)
constexpr int GT_HEADS = 16;         // head outputs (all heads together) are padded to 16
constexpr int GT_LOSS_SUMS = 24;     // orl_gt_train: {policy-loss sum, entropy sum, ratio sum, -, dlogstd[16], value-loss sum, -, -, -}

template <int H>
struct GtC {
  static constexpr int NT = H / 16;                        // M tiles of a layer
  static constexpr int KS = H / 32;                        // bf16 k-steps
  static constexpr int KC = H / 32;                        // chunks (32 A rows each) per GEMM
  static constexpr int WBS = H + 8;                        // bf16 image row stride (elements): 16-byte skew
  static constexpr int CB = 3 * 32 * WBS * 2;              // chunk bytes
  static constexpr int CBP = (CB + 1023) / 1024 * 1024;    // padded to whole-wave DMA blocks
};

// Partition of the cooperative weight gradients over the NW waves of a workgroup (a pass = 16 NW rows): wave w owns OSW
// output-feature slices of 16 x an input-feature part of NTW 16-column tiles of every G_l.
template <int H, int NW>
struct GtW {
  static constexpr int NT = H / 16;
  static constexpr int R = 16 * NW;                        // rows per pass
  static constexpr int RS = R + 4;                         // slab row stride (floats): [feature][row]
  static constexpr int RG = R / 16;                        // 16-row groups of the pass = b128 reads per operand column
  static constexpr int OSW = NT >= NW ? NT / NW : 1;       // output slices per wave
  static constexpr int NOSG = NT / OSW;                    // groups of output slices
  static constexpr int IPARTS = NW / NOSG;                 // input-feature parts
  static constexpr int NTW = NT / IPARTS;                  // input tiles per owned output slice
  static constexpr int G3T = (NT + NW - 1) / NW;           // head feature tiles per wave
  static constexpr int SLAB_FLOATS = H * RS;
};

// Offsets (floats) of the image and of the raw gradient-sum vector; evaluated on host and device from the descriptor
// (closed forms only: no local arrays, everything stays in scalar registers).
struct GtLay {
  int H, D, NLt, NL, DP, DPS, DP16, ntot;
  int iW0, ib0, ib3, iW3T, iW3P, res_bwd, res_fwd, iChunks, chunk_floats, n_chunks, img_total;
  int rG0, rG1, rG3, rdb0, rdb3, raw_total;
  __host__ __device__ GtLay() {}
  __host__ __device__ explicit GtLay(const orl_gt_desc& d) {
    H = d.H; D = d.D; NLt = d.n_layers; NL = d.n_layers - 1;
    DP = (D + 3) & ~3;
    DPS = ((DP >> 2) & 1) ? DP : DP + 4;  // DPS / 4 odd: 16 rows of a 4-byte column read land on distinct banks
    DP16 = (D + 15) & ~15;
    ntot = d.head_n[0] + (d.n_heads > 1 ? d.head_n[1] : 0);
    int o = 0;
    iW0 = o; o += H * DPS;
    ib0 = o; o += NLt * H;
    ib3 = o; o += GT_HEADS;
    iW3T = o; o += H * 20;
    iW3P = o; o += GT_HEADS * (H + 4);
    res_fwd = o;
    res_bwd = o;  // (the backward kernel evaluates the heads too when it computes the losses itself)
    o = (o + 255) & ~255;  // chunks start on a 1 KB boundary
    iChunks = o;
    const int cbp = H == 128 ? GtC<128>::CBP : GtC<64>::CBP;
    chunk_floats = cbp / 4;
    n_chunks = 2 * NL * (H / 32);  // forward images of layers 1..NL, then the transposed images of layers NL..1
    o += n_chunks * chunk_floats;
    img_total = o;
    int r = 0;
    rG0 = r; r += H * DP16;
    rG1 = r; r += NL * H * H;
    rG3 = r; r += GT_HEADS * H;
    rdb0 = r; r += NLt * H;
    rdb3 = r; r += GT_HEADS;
    raw_total = r;
    // (a partial row of the backward kernel carries GT_LOSS_SUMS more floats: the loss / logging sums of orl_gt_train)
  }
  __host__ __device__ int ib(int l) const { return ib0 + l * H; }                 // folded bias of layer l
  __host__ __device__ int rG(int l) const { return rG1 + (l - 1) * H * H; }      // G_l, l >= 1
  __host__ __device__ int rdb(int l) const { return rdb0 + l * H; }
  // chunk index (in consumption order) of chunk c of layer l's forward / transposed image
  __host__ __device__ int fchunk(int l, int c) const { return (l - 1) * (H / 32) + c; }
  __host__ __device__ int bchunk(int l, int c) const { return (NL + (NL - l)) * (H / 32) + c; }
};

// orl_gt_train: the backward kernel evaluates the heads and the losses of its rows itself (no forward launch, no loss
// launches, no head outputs / gradients through HBM); x = the update records, ldx = the record width
struct GtLossArgs {
  int on;                      // 0: d loss / d head outputs come from dh0 / dh1
  int policy_head, value_head; // index of the head the policy / value loss applies to, or -1
  int policy_grad;             // 0: the policy head's gradient is dropped (turn_on = False on a shared network)
  orl_head_desc hd;
  const float* logstd;
  const float* den;
  const float* vn_state;
  orl_ppo_hparams hp;
  GenCols c;
};

struct GtArgs {
  orl_gt_desc d;
  GtLossArgs loss;
  const float* image;
  const float* x;          // rows: x + row * ldx + col0
  int ldx, col0;
  const long long* idx;    // minibatch order (NULL: identity)
  int mb;
  float* out0;             // forward: head outputs [mb, head_n]
  float* out1;
  const float* dh0;        // backward: d loss / d head outputs [mb, head_n]
  const float* dh1;
  float* partials;         // backward: [gridDim.x][raw_total + GT_LOSS_SUMS]
};

// ------------------------------------------------------------------------------------------------ chunk stream
// Two buffers: chunk n lands in buffer n & 1.  Buffer 1 may ALIAS the backward kernel's exchange slab: a GEMM has an even
// number of chunks and starts in buffer 0, the chunk prefetched across the exchange phases is always a GEMM's first one
// (buffer 0), and buffer 1 is only written after the barrier of a consume() inside a GEMM, when every wave has left the
// preceding exchange phase; the slab is only written after a barrier that follows the GEMM's last chunk.
template <int H, int NW>
struct GtStream {
  static constexpr int CBP = GtC<H>::CBP;
  const char* gsrc;   // this lane's global source of chunk 0: image chunks + wave * 1024 + lane * 16
  unsigned ldst[2];   // LDS byte address of this wave's first block in the two buffers
  char* lbase[2];     // generic pointers to the buffers
  int cpp, total, n, pos, wave;

  __device__ __forceinline__ void issue(int p, int buf) {
    const char* src = gsrc + (size_t)p * CBP;
    const unsigned dst = ldst[buf];
#pragma unroll
    for (int off = 0; off < CBP; off += NW * 1024) {
      if (off + wave * 1024 < CBP) {
        // (asm on purpose - see orl_ppo_tower.h: hipcc would drain a __builtin_amdgcn_global_load_lds right after the issue)
        unsigned keep;
        const unsigned m0v = __builtin_amdgcn_readfirstlane(dst + (unsigned)off);
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(src + off), "s"(m0v)
                     : "memory");
      }
    }
  }
  __device__ __forceinline__ void start(const float* chunks, float* buf0, float* buf1, int cpp_, int total_, int wave_,
                                        int lane) {
    gsrc = (const char*)chunks + wave_ * 1024 + lane * 16;
    lbase[0] = (char*)buf0;
    lbase[1] = (char*)buf1;
    ldst[0] = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lbase[0] + (unsigned)wave_ * 1024u;
    ldst[1] = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lbase[1] + (unsigned)wave_ * 1024u;
    cpp = cpp_; total = total_; n = 0; pos = 0; wave = wave_;
    if (total > 0) issue(0, 0);
  }
  // chunk n is complete in LDS for every wave, chunk n + 1 is on its way; returns chunk n
  __device__ __forceinline__ const unsigned short* consume() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int nx = n + 1;
    if (nx < total) {
      pos = pos + 1 == cpp ? 0 : pos + 1;
      issue(pos, nx & 1);
    }
    const unsigned short* r = (const unsigned short*)((n & 1) ? lbase[1] : lbase[0]);
    n = nx;
    return r;
  }
};

// ------------------------------------------------------------------------------------------------ T-layout helpers
template <int NT>
__device__ __forceinline__ void gt_split(const f32x4 (&in)[NT], u32x4 (&xs)[NT / 2][3]) {
#pragma unroll
  for (int h = 0; h < NT / 2; ++h) {
    float x[8];
#pragma unroll
    for (int r = 0; r < 4; ++r) { x[r] = in[2 * h][r]; x[4 + r] = in[2 * h + 1][r]; }
    split8(x, xs[h][0], xs[h][1], xs[h][2]);
  }
}

// (pinned LDS reads gt_ds_read128 / gt_lds_wait and the compile-time loop gt_static_for: orl_mlp.h)

// acc += A in over the streamed chunks of one image (A = the image, T layout in / out): KC chunks x 2 row blocks x KS
// k-steps x 6 products.  The three A fragments of a (row block, k-step) are read PF steps ahead of their MFMAs.
template <int H, int NW, int PF>
__device__ __forceinline__ void gt_gemm(GtStream<H, NW>& st, const u32x4 (&xs)[H / 32][3], f32x4 (&acc)[H / 16], int j, int q) {
  constexpr int KS = GtC<H>::KS, KC = GtC<H>::KC, WBS = GtC<H>::WBS, NS = 2 * KS;
  gt_static_for<0, KC>([&](auto ci) {
    constexpr int c = decltype(ci)::value;
    const unsigned short* Wb = st.consume();
    const unsigned base = gt_lds_addr(Wb) + (unsigned)(j * WBS + q * 8) * 2u;
    u32x4 w[PF + 1][3];
#define GT_FRAG_OFF(p, s) ((((p) * 32 + 16 * ((s) / KS)) * WBS + ((s) % KS) * 32) * 2)
    gt_static_for<0, PF>([&](auto si) {
      constexpr int s = decltype(si)::value;
      w[s][0] = gt_ds_read128<GT_FRAG_OFF(0, s)>(base);
      w[s][1] = gt_ds_read128<GT_FRAG_OFF(1, s)>(base);
      w[s][2] = gt_ds_read128<GT_FRAG_OFF(2, s)>(base);
    });
    gt_static_for<0, NS>([&](auto si) {
      constexpr int s = decltype(si)::value;
      if constexpr (s + PF < NS) {
        w[(s + PF) % (PF + 1)][0] = gt_ds_read128<GT_FRAG_OFF(0, s + PF)>(base);
        w[(s + PF) % (PF + 1)][1] = gt_ds_read128<GT_FRAG_OFF(1, s + PF)>(base);
        w[(s + PF) % (PF + 1)][2] = gt_ds_read128<GT_FRAG_OFF(2, s + PF)>(base);
      }
      constexpr int ahead = (NS - 1 - s) < PF ? (NS - 1 - s) : PF;  // younger steps in flight
      u32x4& wh = w[s % (PF + 1)][0];
      u32x4& wm = w[s % (PF + 1)][1];
      u32x4& wl = w[s % (PF + 1)][2];
      gt_lds_wait<3 * ahead>(wh, wm, wl);
      constexpr int mo = 2 * c + s / KS, h = s % KS;
      acc[mo] = mfma_bf16_16(wl, xs[h][0], acc[mo]);
      acc[mo] = mfma_bf16_16(wh, xs[h][2], acc[mo]);
      acc[mo] = mfma_bf16_16(wm, xs[h][1], acc[mo]);
      acc[mo] = mfma_bf16_16(wm, xs[h][0], acc[mo]);
      acc[mo] = mfma_bf16_16(wh, xs[h][1], acc[mo]);
      acc[mo] = mfma_bf16_16(wh, xs[h][0], acc[mo]);
    });
#undef GT_FRAG_OFF
  });
}

template <int NT>
__device__ __forceinline__ void gt_load_vec(const float* __restrict__ v, int q, f32x4 (&acc)[NT]) {
#pragma unroll
  for (int m = 0; m < NT; ++m) acc[m] = *(const f32x4*)(v + 16 * m + 4 * q);
}

// activation in place; returns the sign bit field (bit 4m + r set when the pre-activation is > 0) used by the backward
template <int NT, int ACT>
__device__ __forceinline__ unsigned gt_act_k(f32x4 (&z)[NT]) {
  unsigned bits = 0u;
#pragma unroll
  for (int m = 0; m < NT; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float v = z[m][r];
      bits |= (v > 0.f ? 1u : 0u) << (4 * m + r);
      z[m][r] = act_fwd(v, ACT);
    }
  return bits;
}
template <int NT>
__device__ __forceinline__ unsigned gt_act(f32x4 (&z)[NT], int act) {
  switch (act) {
    case ORL_ACT_TANH: return gt_act_k<NT, ORL_ACT_TANH>(z);
    case ORL_ACT_RELU: return gt_act_k<NT, ORL_ACT_RELU>(z);
    case ORL_ACT_LEAKY_RELU: return gt_act_k<NT, ORL_ACT_LEAKY_RELU>(z);
    case ORL_ACT_ELU: return gt_act_k<NT, ORL_ACT_ELU>(z);
    default: return 0u;
  }
}

// LayerNorm statistics over the NT * 16 features of a row + normalisation in place (eps 1e-5, biased variance)
template <int NT>
__device__ __forceinline__ void gt_ln(f32x4 (&x)[NT], float& mean, float& rstd) {
  float s = 0.f;
#pragma unroll
  for (int m = 0; m < NT; ++m) s += (x[m][0] + x[m][1]) + (x[m][2] + x[m][3]);
  s = row_allsum(s);
  mean = s * (1.0f / (16 * NT));
  float v = 0.f;
#pragma unroll
  for (int m = 0; m < NT; ++m) {
    x[m] = x[m] - mean;
    v += (x[m][0] * x[m][0] + x[m][1] * x[m][1]) + (x[m][2] * x[m][2] + x[m][3] * x[m][3]);
  }
  v = row_allsum(v);
  rstd = __builtin_amdgcn_rsqf(v * (1.0f / (16 * NT)) + 1e-5f);
#pragma unroll
  for (int m = 0; m < NT; ++m) x[m] = x[m] * rstd;
}

// d <- gradient at the Linear's output, given d = gradient at xhat (the LayerNorm's affine lives in the next layer's
// image), xhat, the row statistics and the activation between Linear and LayerNorm
template <int NT, int ACT>
__device__ __forceinline__ void gt_ln_act_bwd_k(f32x4 (&d)[NT], const f32x4 (&xhat)[NT], float mean, float rstd, unsigned bits) {
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int m = 0; m < NT; ++m) {
    s1 += (d[m][0] + d[m][1]) + (d[m][2] + d[m][3]);
    s2 += (d[m][0] * xhat[m][0] + d[m][1] * xhat[m][1]) + (d[m][2] * xhat[m][2] + d[m][3] * xhat[m][3]);
  }
  row_allsum2(s1, s2);
  const float m1 = s1 * (1.0f / (16 * NT)), m2 = s2 * (1.0f / (16 * NT));
  const float sig = 1.0f / rstd;
#pragma unroll
  for (int m = 0; m < NT; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float t = rstd * (d[m][r] - m1 - xhat[m][r] * m2);
      if (ACT != ORL_ACT_NONE) {
        const bool pos = (bits >> (4 * m + r)) & 1u;
        const float a = xhat[m][r] * sig + mean;  // the activation's output, what orl_gen_layer_bwd reads back
        float g;
        if (ACT == ORL_ACT_TANH) g = 1.f - a * a;
        else if (ACT == ORL_ACT_RELU) g = pos ? 1.f : 0.f;
        else if (ACT == ORL_ACT_LEAKY_RELU) g = pos ? 1.f : 0.01f;
        else g = pos ? 1.f : a + 1.f;  // ELU
        t *= g;
      }
      d[m][r] = t;
    }
}
template <int NT>
__device__ __forceinline__ void gt_ln_act_bwd(f32x4 (&d)[NT], const f32x4 (&xhat)[NT], float mean, float rstd, int act,
                                              unsigned bits) {
  switch (act) {
    case ORL_ACT_TANH: gt_ln_act_bwd_k<NT, ORL_ACT_TANH>(d, xhat, mean, rstd, bits); break;
    case ORL_ACT_RELU: gt_ln_act_bwd_k<NT, ORL_ACT_RELU>(d, xhat, mean, rstd, bits); break;
    case ORL_ACT_LEAKY_RELU: gt_ln_act_bwd_k<NT, ORL_ACT_LEAKY_RELU>(d, xhat, mean, rstd, bits); break;
    case ORL_ACT_ELU: gt_ln_act_bwd_k<NT, ORL_ACT_ELU>(d, xhat, mean, rstd, bits); break;
    default: gt_ln_act_bwd_k<NT, ORL_ACT_NONE>(d, xhat, mean, rstd, bits); break;
  }
}

// T layout -> slab [feature][row]: feature 16m + 4q + r of row 16 wave + j
template <int NT, int RS>
__device__ __forceinline__ void gt_slab_store(float* __restrict__ slab, const f32x4 (&x)[NT], int wave, int j, int q) {
#pragma unroll
  for (int m = 0; m < NT; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) slab[(16 * m + 4 * q + r) * RS + 16 * wave + j] = x[m][r];
}

// the A registers of a 16-feature slice: lane (m = j, kq = q) takes feature f0 + j, rows 16 g + 4 q + e
template <int RG, int RS>
__device__ __forceinline__ void gt_slab_A(const float* __restrict__ slab, int f0, int j, int q, f32x4 (&A)[RG], float& colsum) {
  float s = 0.f;
#pragma unroll
  for (int g = 0; g < RG; ++g) {
    A[g] = *(const f32x4*)(slab + (f0 + j) * RS + 16 * g + 4 * q);
    s += (A[g][0] + A[g][1]) + (A[g][2] + A[g][3]);
  }
  colsum += s;
}

// acc[t] (16 x 16 tiles) += A^T-slice (registers) x slab features [16 t, 16 t + 16) beyond `base` over the pass's rows:
// 4 RG fp32 MFMAs per tile; the B operand reads (one 16-byte read per 4 MFMAs) run two reads ahead (see gt_ds_read128).
// base = LDS byte address of this lane's first read: slab + ((first feature + j) * RS + 4 q) * 4.
template <int NTILES, int RG, int RS>
__device__ __forceinline__ void gt_wgrad_tiles(unsigned base, const f32x4 (&A)[RG], f32x4 (&acc)[NTILES]) {
  constexpr int N = NTILES * RG, PFW = 2;
  u32x4 b[PFW + 1];
#define GT_B_OFF(n) ((((n) / RG) * 16 * RS + ((n) % RG) * 16) * 4)
  gt_static_for<0, (PFW < N ? PFW : N)>([&](auto ni) {
    constexpr int n = decltype(ni)::value;
    b[n] = gt_ds_read128<GT_B_OFF(n)>(base);
  });
  gt_static_for<0, N>([&](auto ni) {
    constexpr int n = decltype(ni)::value;
    if constexpr (n + PFW < N) b[(n + PFW) % (PFW + 1)] = gt_ds_read128<GT_B_OFF(n + PFW)>(base);
    constexpr int ahead = (N - 1 - n) < PFW ? (N - 1 - n) : PFW;
    u32x4& bb = b[n % (PFW + 1)];
    gt_lds_wait<ahead>(bb);
    const f32x4 bv = __builtin_bit_cast(f32x4, bb);
    constexpr int t = n / RG, g = n % RG;
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[t] = ORL_MFMA(A[g][e], bv[e], acc[t]);
  });
#undef GT_B_OFF
}

#ifdef ORL_PROF
// phase timing build (python -m openrl_amd.csrc.build --prof; tools/gt_phase_prof.py): wave 0 of workgroup 0 adds the
// s_memtime delta of each phase of its pass loop to an LDS counter
__device__ unsigned long long g_gt_prof[16];
#define GT_T(k)                                                                   \
  do {                                                                            \
    if (prof_on) {                                                                \
      const unsigned long long t_now = __builtin_readcyclecounter();              \
      if (l == 0) atomicAdd(&prof_lds[k], t_now - t_last);                        \
      t_last = t_now;                                                             \
    }                                                                             \
  } while (0)
#else
#define GT_T(k) ((void)0)
#endif

// ------------------------------------------------------------------------------------------------ the kernel
// H: hidden width (64 / 128); NL: number of H x H layers (n_layers - 1); ND: 16-column blocks of observation registers
// (1: D <= 16, 4: D <= 64); BWD: backward kernel (forward recompute + gradients) or forward kernel (head outputs).
// NW: waves per workgroup - the backward kernel is built with 4 (two workgroups per CU when the LDS allows, their phases
// interleave) and 8 (one workgroup per CU, wide observations); the forward kernel with 8.
template <int H, int NL, int ND, bool BWD, int NW>
__device__ __forceinline__ void gt_body(const GtArgs& A) {
  using Cn = GtC<H>;
  using Wn = GtW<H, NW>;
  constexpr int NT = Cn::NT, KS = Cn::KS, RS = Wn::RS, RG = Wn::RG, OSW = Wn::OSW, NTW = Wn::NTW;
#ifdef ORL_GT_PF_BWD  // build-time experiment
  constexpr int PF = BWD ? ORL_GT_PF_BWD : 2;
#else
  constexpr int PF = BWD ? 1 : 2;  // A-fragment read-ahead (the backward kernel has no registers for a second step)
#endif
  extern __shared__ __attribute__((aligned(1024))) float smem[];
  const GtLay ly(A.d);
  const int wave = threadIdx.x >> 6, l = threadIdx.x & 63, j = l & 15, q = l >> 4;
  const int res = BWD ? ly.res_bwd : ly.res_fwd;
  const int res_pad = (res + 255) & ~255;
  float* cbuf = smem + res_pad;                                   // chunk buffer 0
  // backward: the exchange slab [H][RS], which is also chunk buffer 1 (see GtStream); forward: a second chunk buffer
  float* slab = cbuf + Cn::CBP / 4;
  // fused losses: per wave a [16 rows][16] head-output tile (the row's loss turns it into d loss / d outputs in place)
  // and a [16 rows][16] d logstd accumulator
  float* ltile = slab + Wn::SLAB_FLOATS + wave * 256;
  float* dls_rows = slab + Wn::SLAB_FLOATS + NW * 256 + wave * 256;
  const float* lw = smem;

  const int n_tiles = (A.mb + 15) >> 4;
  const int n_pass = (n_tiles + NW - 1) / NW;
  const int my_pass = ((int)blockIdx.x < n_pass) ? (n_pass - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
  const int cpp = (BWD ? 2 : 1) * NL * Cn::KC;

  GtStream<H, NW> st;
  st.start(A.image + ly.iChunks, cbuf, slab, cpp, my_pass * cpp, wave, l);
  // resident part of the image: fc1's fp32 matrix, the folded biases, the head matrices
  for (int e = threadIdx.x * 4; e < res; e += NW * 64 * 4) *(f32x4*)(smem + e) = *(const f32x4*)(A.image + e);
  __syncthreads();

  const int D = ly.D, DPS = ly.DPS, nks = ly.DP >> 2;
  const int n0 = A.d.head_n[0], n1 = A.d.n_heads > 1 ? A.d.head_n[1] : 0;
  const bool fn = A.d.o_fn_g >= 0;

  // ---- persistent accumulators of the cooperative weight gradients (backward)
  const int osg = wave % Wn::NOSG, ip = wave / Wn::NOSG;  // output slices 16 (osg OSW + a), input tiles ip NTW + t
  f32x4 G[NL > 0 ? NL : 1][OSW][NTW];
  f32x4 G0[OSW][ND];
  f32x4 G3[Wn::G3T];
  float dbs[NL + 1][OSW];
  float db3 = 0.f;
  if constexpr (BWD) {
#pragma unroll
    for (int a = 0; a < NL; ++a)
#pragma unroll
      for (int o = 0; o < OSW; ++o)
#pragma unroll
        for (int t = 0; t < NTW; ++t) G[a][o][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int o = 0; o < OSW; ++o)
#pragma unroll
      for (int t = 0; t < ND; ++t) G0[o][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < Wn::G3T; ++u) G3[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a <= NL; ++a)
#pragma unroll
      for (int o = 0; o < OSW; ++o) dbs[a][o] = 0.f;
  }

  // fused losses: denominators, ValueNorm coefficients, per-lane sums (lanes q == 0 own a row each)
  float inv_den_p = 1.f, inv_den_v = 1.f, vn_mean = 0.f, vn_sd = 1.f;
  float st_loss = 0.f, st_ent = 0.f, st_ratio = 0.f, st_v = 0.f;
  if constexpr (BWD) {
    if (A.loss.on) {
      inv_den_p = 1.f / (A.loss.hp.use_policy_active_masks ? A.loss.den[0] : A.loss.den[1]);
      inv_den_v = 1.f / (A.loss.hp.use_value_active_masks ? A.loss.den[0] : A.loss.den[1]);
      gen_vn_coeffs(A.loss.vn_state, A.loss.hp, vn_mean, vn_sd);
      for (int e = l; e < 256; e += 64) dls_rows[e] = 0.f;
    }
  }
  auto row_of = [&](int pass) -> long long {
    const int ii = (pass * NW + wave) * 16 + j;
    const int iv = (pass < n_pass && ii < A.mb) ? ii : 0;  // invalid lanes read row 0 (finite data, zero gradient)
    return (A.idx != nullptr) ? A.idx[iv] : (long long)iv;
  };
  long long row_next = row_of(blockIdx.x);
#ifdef ORL_PROF
  __shared__ unsigned long long prof_lds[16];
  const bool prof_on = BWD && blockIdx.x == 0 && wave == 0;
  if (prof_on && l < 16) prof_lds[l] = 0ull;
  unsigned long long t_last = __builtin_readcyclecounter();
#endif
  for (int pass = blockIdx.x; pass < n_pass; pass += gridDim.x) {
    GT_T(9);
    const int i = (pass * NW + wave) * 16 + j;
    const bool valid = i < A.mb;
    // this pass's row index was loaded one pass ago; the next pass's is requested now
    const long long row = row_next;
    row_next = row_of(pass + gridDim.x);
    const float* xr = A.x + (size_t)row * A.ldx + A.col0;

    // ---- observations: lane (j, q) holds x[j][4 s + q]; feature norm (LayerNorm over the D real columns)
    float x0[4 * ND];
#pragma unroll
    for (int s = 0; s < 4 * ND; ++s) x0[s] = (4 * s + q < D) ? xr[4 * s + q] : 0.f;
    if (fn) {
      float sm = 0.f;
#pragma unroll
      for (int s = 0; s < 4 * ND; ++s) sm += x0[s];
      sm = row_allsum(sm);
      const float mean = sm / (float)D;
      float v = 0.f;
#pragma unroll
      for (int s = 0; s < 4 * ND; ++s) {
        x0[s] = (4 * s + q < D) ? x0[s] - mean : 0.f;
        v += x0[s] * x0[s];
      }
      v = row_allsum(v);
      const float rs = __builtin_amdgcn_rsqf(v / (float)D + 1e-5f);
#pragma unroll
      for (int s = 0; s < 4 * ND; ++s) x0[s] *= rs;
    }

    // ---- forward (recomputed in the backward kernel): xh[l] = LayerNorm output of layer l without its affine
    f32x4 xh[NL + 1][NT];
    float mu[NL + 1], rstd[NL + 1];
    unsigned bits[NL + 1];
    gt_load_vec<NT>(lw + ly.ib(0), q, xh[0]);
#pragma unroll
    for (int s = 0; s < 4 * ND; ++s) {
      if (s < nks) {
#pragma unroll
        for (int m = 0; m < NT; ++m) {
          const float a = lw[ly.iW0 + (16 * m + j) * DPS + 4 * s + q];
          xh[0][m] = ORL_MFMA(a, x0[s], xh[0][m]);
        }
      }
    }
    bits[0] = gt_act<NT>(xh[0], A.d.act[0]);
    gt_ln<NT>(xh[0], mu[0], rstd[0]);
    GT_T(0);  // observations, fc1, activation, LayerNorm
#pragma unroll
    for (int k = 1; k <= NL; ++k) {
      u32x4 xs[KS][3];
      gt_split<NT>(xh[k - 1], xs);
      gt_load_vec<NT>(lw + ly.ib(k), q, xh[k]);
      gt_gemm<H, NW, PF>(st, xs, xh[k], j, q);
      bits[k] = gt_act<NT>(xh[k], A.d.act[k]);
      gt_ln<NT>(xh[k], mu[k], rstd[k]);
    }
    GT_T(1);  // forward H x H layers: split, streamed GEMM, activation, LayerNorm

    if constexpr (!BWD) {
      // ---- heads: out^T[16 c x 16 rows] = W3' xhat_NL^T + b3'; lane (j, q) keeps outputs c = 4q .. 4q + 3 of row j
      f32x4 hv = *(const f32x4*)(lw + ly.ib3 + 4 * q);
#pragma unroll
      for (int m = 0; m < NT; ++m) {
        const f32x4 a4 = *(const f32x4*)(lw + ly.iW3P + j * (H + 4) + 16 * m + 4 * q);
#pragma unroll
        for (int r = 0; r < 4; ++r) hv = ORL_MFMA(a4[r], xh[NL][m][r], hv);
      }
      if (valid) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = 4 * q + r;
          if (c < n0) A.out0[(size_t)i * n0 + c] = hv[r];
          else if (c - n0 < n1) A.out1[(size_t)i * n1 + (c - n0)] = hv[r];
        }
      }
    } else {
      // ---- d loss / d head outputs of this lane's row: c = 4q + s
      float dh[4];
      if (A.loss.on) {
        // heads: out^T[16 c x 16 rows] = W3' xhat_NL^T + b3' (as the forward kernel), through the wave's tile to the row's
        // owner lane (q == 0), which evaluates the losses of its row with the stand-alone kernels' code (orl_gen_loss.h)
        f32x4 hv = *(const f32x4*)(lw + ly.ib3 + 4 * q);
#pragma unroll
        for (int m = 0; m < NT; ++m) {
          const f32x4 a4 = *(const f32x4*)(lw + ly.iW3P + j * (H + 4) + 16 * m + 4 * q);
#pragma unroll
          for (int r = 0; r < 4; ++r) hv = ORL_MFMA(a4[r], xh[NL][m][r], hv);
        }
        *(f32x4*)(ltile + j * 16 + 4 * q) = hv;
        wave_lds_fence();
        if (q == 0) {
          float* trow = ltile + j * 16;
          const float* rr = A.x + (size_t)row * A.ldx;
          if (A.loss.policy_head >= 0) {
            float* lg = trow + (A.loss.policy_head == 0 ? 0 : n0);
            const int np = A.loss.hd.n_out;
            if (valid && A.loss.policy_grad) {
              float (&dls)[16] = *(float(*)[16])(dls_rows + j * 16);
              gen_policy_loss_row(A.loss.hd, lg, A.loss.logstd, rr, A.loss.c, inv_den_p, inv_den_p, A.loss.hp, 0, lg, nullptr,
                                  nullptr, st_loss, st_ent, st_ratio, dls);
            } else {
              if (valid) {  // statistics only (the logging sums of a policy that is not stepped), gradient dropped
                float scratch_dls[16];
                float (&dls)[16] = scratch_dls;
                for (int k = 0; k < 16; ++k) dls[k] = 0.f;
                gen_policy_loss_row(A.loss.hd, lg, A.loss.logstd, rr, A.loss.c, inv_den_p, inv_den_p, A.loss.hp, 0, lg, nullptr,
                                    nullptr, st_loss, st_ent, st_ratio, dls);
              }
              for (int k = 0; k < np; ++k) lg[k] = 0.f;
            }
          }
          if (A.loss.value_head >= 0) {
            float* vv = trow + (A.loss.value_head == 0 ? 0 : n0);
            vv[0] = valid ? gen_value_loss_row(vv[0], rr, A.loss.c, vn_mean, vn_sd, inv_den_v, A.loss.hp, st_v) : 0.f;
          }
        }
        wave_lds_fence();
        const f32x4 dv = *(const f32x4*)(ltile + j * 16 + 4 * q);
#pragma unroll
        for (int s = 0; s < 4; ++s) dh[s] = (4 * q + s < n0 + n1) ? dv[s] : 0.f;
        wave_lds_fence();
      } else {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const int c = 4 * q + s;
          float v = 0.f;
          if (valid) {
            if (c < n0) v = A.dh0[(size_t)i * n0 + c];
            else if (c - n0 < n1) v = A.dh1[(size_t)i * n1 + (c - n0)];
          }
          dh[s] = v;
        }
      }
      // ---- G3 += dhead^T xhat_NL, db3 += sum dhead
      f32x4 Areg[OSW][RG];
      __syncthreads();  // (the forward's last chunk - buffer 1 = the slab - has been consumed by every wave)
#pragma unroll
      for (int s = 0; s < 4; ++s) slab[(4 * q + s) * RS + 16 * wave + j] = dh[s];
      __syncthreads();
      {
        float cs = 0.f;
        gt_slab_A<RG, RS>(slab, 0, j, q, Areg[0], cs);
        db3 += cs;
      }
      __syncthreads();
      gt_slab_store<NT, RS>(slab, xh[NL], wave, j, q);
      __syncthreads();
      const unsigned slab_lane = gt_lds_addr(slab) + (unsigned)(j * RS + 4 * q) * 4u;  // this lane's operand-read origin
#pragma unroll
      for (int u = 0; u < Wn::G3T; ++u)
        if (wave + NW * u < NT) {
          f32x4 (&g3)[1] = *(f32x4(*)[1])&G3[u];
          gt_wgrad_tiles<1, RG, RS>(slab_lane + (unsigned)(16 * (wave + NW * u) * RS) * 4u, Areg[0], g3);
        }
      // ---- d xhat_NL = W3'^T dhead (A = the [H][20] transposed head image, K = the 16 head outputs)
      f32x4 d[NT];
#pragma unroll
      for (int m = 0; m < NT; ++m) {
        d[m] = f32x4{0.f, 0.f, 0.f, 0.f};
        const f32x4 a4 = *(const f32x4*)(lw + ly.iW3T + (16 * m + j) * 20 + 4 * q);
#pragma unroll
        for (int s = 0; s < 4; ++s) d[m] = ORL_MFMA(a4[s], dh[s], d[m]);
      }
      GT_T(2);  // dhead loads, the head's exchange + G3, head input gradient
#pragma unroll
      for (int k = NL; k >= 0; --k) {
        // d = gradient at xhat_k  ->  dz_k
        gt_ln_act_bwd<NT>(d, xh[k], mu[k], rstd[k], A.d.act[k], bits[k]);
        GT_T(3);  // LayerNorm + activation backward
        __syncthreads();  // every wave has finished reading the slab's previous contents (operand reads / chunk reads)
        gt_slab_store<NT, RS>(slab, d, wave, j, q);
        __syncthreads();
#pragma unroll
        for (int o = 0; o < OSW; ++o) gt_slab_A<RG, RS>(slab, 16 * (osg * OSW + o), j, q, Areg[o], dbs[k][o]);
        __syncthreads();
        GT_T(4);  // dz through the slab: 3 barriers, store, A reads
        if (k > 0) {
          gt_slab_store<NT, RS>(slab, xh[k - 1], wave, j, q);
          __syncthreads();
          GT_T(5);  // xhat through the slab: store + barrier
#pragma unroll
          for (int o = 0; o < OSW; ++o)
            gt_wgrad_tiles<NTW, RG, RS>(slab_lane + (unsigned)(16 * ip * NTW * RS) * 4u, Areg[o], G[k - 1][o]);
          GT_T(6);  // G_k MFMAs
          // input gradient d xhat_{k-1} = W_k'^T dz_k through the transposed image's chunks
          u32x4 xs[KS][3];
          gt_split<NT>(d, xs);
#pragma unroll
          for (int m = 0; m < NT; ++m) d[m] = f32x4{0.f, 0.f, 0.f, 0.f};
          gt_gemm<H, NW, PF>(st, xs, d, j, q);
          GT_T(7);  // input gradient: split + streamed GEMM
        } else {
          // the observations as the B operand: rows d = 4 s + q of the slab
#pragma unroll
          for (int s = 0; s < 4 * ND; ++s) slab[(4 * s + q) * RS + 16 * wave + j] = x0[s];
          __syncthreads();
#pragma unroll
          for (int o = 0; o < OSW; ++o)
#pragma unroll
            for (int t = 0; t < ND; ++t)
              if (16 * t < D && (t % Wn::IPARTS) == ip) {
                f32x4 (&g0)[1] = *(f32x4(*)[1])&G0[o][t];
                gt_wgrad_tiles<1, RG, RS>(slab_lane + (unsigned)(16 * t * RS) * 4u, Areg[o], g0);
              }
          GT_T(8);  // observations through the slab + G0
        }
      }
    }
  }

#ifdef ORL_PROF
  if (prof_on && l < 12) atomicAdd(&g_gt_prof[l], prof_lds[l]);
  if (prof_on && l == 12) atomicAdd(&g_gt_prof[12], (unsigned long long)my_pass);
#endif
  if constexpr (BWD) {
    // ---- this workgroup's partial row of the raw sums.  A tile's lane (j, q) register r holds
    // (output feature 16 slice + 4 q + r, input feature 16 tile + j)
    float* P = A.partials + (size_t)blockIdx.x * (ly.raw_total + GT_LOSS_SUMS);
#pragma unroll
    for (int k = 1; k <= NL; ++k)
#pragma unroll
      for (int o = 0; o < OSW; ++o)
#pragma unroll
        for (int t = 0; t < NTW; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            P[ly.rG(k) + (16 * (osg * OSW + o) + 4 * q + r) * H + 16 * (ip * NTW + t) + j] = G[k - 1][o][t][r];
#pragma unroll
    for (int o = 0; o < OSW; ++o)
#pragma unroll
      for (int t = 0; t < ND; ++t)
        if (16 * t < ly.DP16 && (t % Wn::IPARTS) == ip) {
#pragma unroll
          for (int r = 0; r < 4; ++r) P[ly.rG0 + (16 * (osg * OSW + o) + 4 * q + r) * ly.DP16 + 16 * t + j] = G0[o][t][r];
        }
#pragma unroll
    for (int u = 0; u < Wn::G3T; ++u)
      if (wave + NW * u < NT) {
#pragma unroll
        for (int r = 0; r < 4; ++r) P[ly.rG3 + (4 * q + r) * H + 16 * (wave + NW * u) + j] = G3[u][r];
      }
    // loss / logging sums of orl_gt_train: GT_LOSS_SUMS floats behind the raw sums, fixed summation order
    if (A.loss.on) {
      __syncthreads();
      float* red = slab;  // (the slab is free: every wave is past its last pass)
      const float v4[4] = {wave_sum(st_loss), wave_sum(st_ent), wave_sum(st_ratio), wave_sum(st_v)};
      if (l == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) red[wave * 4 + k] = v4[k];
      }
      __syncthreads();
      if (wave == 0) {
        if (l < 4) {
          float t = 0.f;
          for (int w = 0; w < NW; ++w) t += red[w * 4 + l];
          if (l < 3) P[ly.raw_total + l] = t;
          else { P[ly.raw_total + 20] = t; P[ly.raw_total + 3] = 0.f; }
        } else if (l >= 16 && l < 32) {  // d logstd column l - 16: the rows of all waves in order
          const float* dr = slab + Wn::SLAB_FLOATS + NW * 256;
          float t = 0.f;
          for (int rw = 0; rw < NW * 16; ++rw) t += dr[rw * 16 + (l - 16)];
          P[ly.raw_total + 4 + (l - 16)] = t;
        } else if (l >= 32 && l < 35) {
          P[ly.raw_total + 21 + (l - 32)] = 0.f;
        }
      }
    }
    // column sums: lane (m = j, kq = q) holds a partial of feature 16 slice + j (every wave of an output slice holds the
    // same sums: the ip == 0 wave writes); db3: feature c = j, identical in all waves - wave 0 writes
#pragma unroll
    for (int k = 0; k <= NL; ++k)
#pragma unroll
      for (int o = 0; o < OSW; ++o) {
        const float sm = row_allsum(dbs[k][o]);
        if (ip == 0 && q == 0) P[ly.rdb(k) + 16 * (osg * OSW + o) + j] = sm;
      }
    {
      const float sm = row_allsum(db3);
      if (wave == 0 && q == 0) P[ly.rdb3 + j] = sm;
    }
  }
}

}  // namespace orl
