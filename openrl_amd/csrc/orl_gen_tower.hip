// orl_gen_tower.hip - C ABI of the cross-layer fused general towers (kernels in orl_gen_tower.h): orl_gt_supported,
// orl_gt_image_floats, orl_gt_raw_floats, orl_gt_prep, orl_gt_fwd, orl_gt_bwd.
#include "orl_gen_tower_launch.h"

namespace orl {

// ------------------------------------------------------------------------------------------------ image
// LayerNorm affine feeding layer l (l = 0: the feature norm, or identity)
__device__ __forceinline__ float gt_gin(const orl_gt_desc& d, int l, int i) {
  if (l == 0) return d.o_fn_g >= 0 ? d.theta[d.o_fn_g + i] : 1.f;
  return d.theta[d.og[l - 1] + i];
}
__device__ __forceinline__ float gt_bein(const orl_gt_desc& d, int l, int i) {
  if (l == 0) return d.o_fn_be >= 0 ? d.theta[d.o_fn_be + i] : 0.f;
  return d.theta[d.obe[l - 1] + i];
}
// row c of the stacked head matrix [ntot][H] and its bias
__device__ __forceinline__ const float* gt_head_row(const orl_gt_desc& d, int c) {
  return c < d.head_n[0] ? d.theta + d.head_oW[0] + c * d.H : d.theta + d.head_oW[1] + (c - d.head_n[0]) * d.H;
}
__device__ __forceinline__ float gt_head_bias(const orl_gt_desc& d, int c) {
  return c < d.head_n[0] ? d.theta[d.head_ob[0] + c] : d.theta[d.head_ob[1] + (c - d.head_n[0])];
}

// element (row mrow, reduction index k) of an image: chunk mrow / 32, then [part][row % 32][k in fragment order]
__device__ __forceinline__ void gt_image_store(unsigned short* __restrict__ chunks, int chunk0, int chunk_ushorts, int WBS,
                                               int mrow, int k, float w) {
  const int m = k >> 4, qq = (k >> 2) & 3, r = k & 3, h = m >> 1, sl = (m & 1) * 4 + r;
  unsigned short* base = chunks + (size_t)(chunk0 + (mrow >> 5)) * chunk_ushorts + (mrow & 31) * WBS + h * 32 + qq * 8 + sl;
  const float h1 = u2f(f2u(w) & 0xffff0000u), r1 = w - h1;
  const float m1 = u2f(f2u(r1) & 0xffff0000u), l1 = r1 - m1;
  base[0] = (unsigned short)(f2u(h1) >> 16);
  base[32 * WBS] = (unsigned short)(f2u(m1) >> 16);
  base[64 * WBS] = (unsigned short)(f2u(l1) >> 16);
}

// sum over the 16 lanes of an aligned lane group, result in all of them
__device__ __forceinline__ float gt_sum16(float v) {
  v += __shfl_xor(v, 1);
  v += __shfl_xor(v, 2);
  v += __shfl_xor(v, 4);
  v += __shfl_xor(v, 8);
  return v;
}

// Work split of the two small kernels: DOT items (a reduction of up to 128 terms) take 16 lanes each, 4 items per wave
// and round, every lane of a wave running the same number of rounds (the shuffles are wave-wide); ELEMENT items take one
// thread each.
__global__ __launch_bounds__(256) void gt_prep_kernel(orl_gt_desc d, float* __restrict__ img) {
  const GtLay ly(d);
  const int H = ly.H, D = ly.D, NLt = ly.NLt, ntot = ly.ntot;
  const int lane = threadIdx.x & 63, sub = lane & 15, grp = lane >> 4;
  const int wave_g = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
  // ---- folded biases: item = (layer l, output o) for l < NLt, then the head outputs c
  const int n_dot = NLt * H + GT_HEADS;
  for (int base = wave_g * 4; base < n_dot; base += n_waves * 4) {
    const int it = base + grp;
    float acc = 0.f, b = 0.f;
    int dst = -1;
    if (it < NLt * H) {
      const int l = it / H, o = it - l * H, n_in = l == 0 ? D : H;
      dst = ly.ib(l) + o;
      b = d.theta[d.ob[l] + o];
      if (l > 0 || d.o_fn_be >= 0) {
        const float* wrow = d.theta + d.oW[l] + o * n_in;
        for (int k = sub; k < n_in; k += 16) acc += wrow[k] * gt_bein(d, l, k);
      }
    } else if (it < n_dot) {
      const int c = it - NLt * H;
      dst = ly.ib3 + c;
      if (c < ntot) {
        const float* wrow = gt_head_row(d, c);
        b = gt_head_bias(d, c);
        for (int f = sub; f < H; f += 16) acc += wrow[f] * d.theta[d.obe[NLt - 1] + f];
      }
    }
    acc = gt_sum16(acc);
    if (dst >= 0 && sub == 0) img[dst] = b + acc;
  }
  // ---- element items
  const int nA = H * ly.DPS, nD = H * 20, nE = GT_HEADS * (H + 4), nF = ly.NL * H * H;
  const int total = nA + nD + nE + nF;
  const int WBS = H + 8, chunk_ushorts = ly.chunk_floats * 2;
  unsigned short* chunks = (unsigned short*)(img + ly.iChunks);
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    int t = e;
    if (t < nA) {  // fc1: W0 diag(g_fn), zero padded to DPS columns
      const int o = t / ly.DPS, k = t - o * ly.DPS;
      img[ly.iW0 + t] = k < D ? d.theta[d.oW[0] + o * D + k] * gt_gin(d, 0, k) : 0.f;
      continue;
    }
    t -= nA;
    if (t < nD) {  // (W3 diag(g_NL))^T as [H][20]
      const int f = t / 20, c = t - f * 20;
      img[ly.iW3T + t] = c < ntot ? gt_head_row(d, c)[f] * d.theta[d.og[NLt - 1] + f] : 0.f;
      continue;
    }
    t -= nD;
    if (t < nE) {  // W3 diag(g_NL) as [16][H + 4]
      const int c = t / (H + 4), f = t - c * (H + 4);
      img[ly.iW3P + t] = (c < ntot && f < H) ? gt_head_row(d, c)[f] * d.theta[d.og[NLt - 1] + f] : 0.f;
      continue;
    }
    t -= nE;
    {  // H x H layers: three-term bf16 images of W' = W diag(g_{l-1}) (rows o, reduction i) and W'^T (rows i, reduction o)
      const int l = 1 + t / (H * H), oi = t - (l - 1) * H * H, o = oi / H, i = oi - o * H;
      const float w = d.theta[d.oW[l] + oi] * gt_gin(d, l, i);
      gt_image_store(chunks, ly.fchunk(l, 0), chunk_ushorts, WBS, o, i, w);
      gt_image_store(chunks, ly.bchunk(l, 0), chunk_ushorts, WBS, i, o, w);
    }
  }
}

// ------------------------------------------------------------------------------------------------ raw sums -> gradients
__global__ __launch_bounds__(256) void gt_finalize_kernel(orl_gt_desc d, const float* __restrict__ raw, float* __restrict__ grad) {
  const GtLay ly(d);
  const int H = ly.H, D = ly.D, NLt = ly.NLt, ntot = ly.ntot;
  const bool fn = d.o_fn_g >= 0;
  const int lane = threadIdx.x & 63, sub = lane & 15, grp = lane >> 4;
  const int wave_g = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
  // ---- DOT items: the LayerNorm affines as linear images of the NEXT Linear's sums.  item = (layer l, which, column i),
  // l = -1 for the feature norm: d g[i] = sum_o W[o][i] G[o][i], d be[i] = sum_o W[o][i] db[o]
  const int n_fn = fn ? 2 * D : 0, n_dot = n_fn + NLt * 2 * H;
  for (int base = wave_g * 4; base < n_dot; base += n_waves * 4) {
    const int it = base + grp;
    float acc = 0.f;
    int dst = -1;
    if (it < n_fn) {
      const int which = it / D, k = it - which * D;
      dst = (which == 0 ? d.o_fn_g : d.o_fn_be) + k;
      for (int o = sub; o < H; o += 16)
        acc += d.theta[d.oW[0] + o * D + k] * (which == 0 ? raw[ly.rG0 + o * ly.DP16 + k] : raw[ly.rdb(0) + o]);
    } else if (it < n_dot) {
      const int t = it - n_fn, l = t / (2 * H), r = t - l * 2 * H, which = r / H, i = r - which * H;
      dst = (which == 0 ? d.og[l] : d.obe[l]) + i;
      if (l + 1 < NLt) {
        const float* W = d.theta + d.oW[l + 1];
        for (int o = sub; o < H; o += 16)
          acc += W[o * H + i] * (which == 0 ? raw[ly.rG(l + 1) + o * H + i] : raw[ly.rdb(l + 1) + o]);
      } else {
        for (int c = sub; c < ntot; c += 16)
          acc += gt_head_row(d, c)[i] * (which == 0 ? raw[ly.rG3 + c * H + i] : raw[ly.rdb3 + c]);
      }
    }
    acc = gt_sum16(acc);
    if (dst >= 0 && sub == 0) grad[dst] = acc;
  }
  // ---- ELEMENT items: per layer (n_in H weights + H biases), per head output (H weights + 1 bias)
  int total = 0;
  for (int l = 0; l < NLt; ++l) total += (l == 0 ? D : H) * H + H;
  total += ntot * H + ntot;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    int t = e;
    bool done = false;
    for (int l = 0; l < NLt && !done; ++l) {
      const int n_in = l == 0 ? D : H;
      const int nW = n_in * H;
      if (t < nW) {  // dW_l[o][i] = G_l[o][i] g_in[i] + db_l[o] be_in[i]
        const int o = t / n_in, i = t - o * n_in;
        const float g = l == 0 ? raw[ly.rG0 + o * ly.DP16 + i] : raw[ly.rG(l) + o * H + i];
        grad[d.oW[l] + t] = g * gt_gin(d, l, i) + raw[ly.rdb(l) + o] * gt_bein(d, l, i);
        done = true;
        break;
      }
      t -= nW;
      if (t < H) {
        grad[d.ob[l] + t] = raw[ly.rdb(l) + t];
        done = true;
        break;
      }
      t -= H;
    }
    if (done) continue;
    if (t < ntot * H) {  // head weights: G3[c][f] g_NL[f] + db3[c] be_NL[f]
      const int c = t / H, f = t - c * H;
      const float v = raw[ly.rG3 + t] * d.theta[d.og[NLt - 1] + f] + raw[ly.rdb3 + c] * d.theta[d.obe[NLt - 1] + f];
      if (c < d.head_n[0]) grad[d.head_oW[0] + c * H + f] = v;
      else grad[d.head_oW[1] + (c - d.head_n[0]) * H + f] = v;
      continue;
    }
    t -= ntot * H;
    if (t < d.head_n[0]) grad[d.head_ob[0] + t] = raw[ly.rdb3 + t];
    else grad[d.head_ob[1] + (t - d.head_n[0])] = raw[ly.rdb3 + t];
  }
}

// ------------------------------------------------------------------------------------------------ host side
// bwd_waves = 0: the forward kernel; 4 / 8: the backward kernel's builds
static size_t gt_lds_bytes(const GtLay& ly, int bwd_waves) {
  const int res = bwd_waves ? ly.res_bwd : ly.res_fwd;
  size_t fl = (size_t)((res + 255) & ~255) + (size_t)ly.chunk_floats;
  // backward: the exchange slab [H][16 waves + 4], which doubles as the second chunk buffer
  const size_t slab = (size_t)ly.H * (16 * bwd_waves + 4);
  fl += bwd_waves ? (slab > (size_t)ly.chunk_floats ? slab : (size_t)ly.chunk_floats) : (size_t)ly.chunk_floats;
  if (bwd_waves) fl += (size_t)bwd_waves * 512;  // per wave: the head-output tile and the d logstd rows of the fused losses
  return fl * sizeof(float);
}
// Waves per backward workgroup: 4 (one workgroup per CU, one wave per SIMD, 512 registers, no scratch traffic) when its
// LDS fits 80 KB, else 8.  (Round 3 measured TWO 4-wave workgroups per CU at 256 registers - ORL_GT_BWD_MINWAVES = 2 -
// slower, 531 vs 495 us: twice the G accumulators per wave, 169 spilled VGPRs; with the whole register file per wave the
// 4-wave form costs the same time as the 8-wave one and spills nothing.)
static int gt_bwd_waves(const GtLay& ly) {
  // ORL_GT_BWD_LDS_CAP_KB: towers whose 4-wave LDS footprint exceeds it take the 8-wave build.  Measured (round 4, same box,
  // profiles/r04_experiments.md): hidden 64 (<= 80 KB) - both forms the same time, so the scratch-free 4-wave form ships;
  // hidden 128 (85 KB) - 4 waves 13.69 ms per iteration with 37 + 21 MB of HBM traffic per backward launch, 8 waves 12.69 ms
  // with 221 + 394 MB (its 121 - 177 spilled VGPRs): the spilling build is 8 % FASTER there and stays the default;
  // -DORL_GT_BWD_LDS_CAP_KB=160 ships the scratch-free form everywhere.  (Round 3's "no-spill variant, same time" at hidden
  // 128 compared the 8-wave kernel with itself: the 80 KB bound of the two-workgroups-per-CU form had been left in.)
  const size_t cap = (size_t)ORL_GT_BWD_LDS_CAP_KB * 1024;
  return (ORL_GT_BWD_WAVES == 4 && gt_lds_bytes(ly, 4) <= cap) ? 4 : 8;
}

static int gt_check(const orl_gt_desc* d, const char* who) {
  ORL_REQUIRE(d && d->theta, "%s: null descriptor", who);
  if (d->H != 64 && d->H != 128) return fail(ORL_E_UNSUPPORTED, "%s: hidden_size %d (the fused towers take 64 / 128)", who, d->H);
  const int max_layers = ORL_GT_MAX_LAYERS;  // (three 128-wide hidden layers spill ~700 VGPRs; the kernel is issue-bound, it still wins)
  if (d->n_layers < 2 || d->n_layers > max_layers)
    return fail(ORL_E_UNSUPPORTED, "%s: %d layers at hidden_size %d (2..%d)", who, d->n_layers, d->H, max_layers);
  if (d->D < 1 || d->D > 64) return fail(ORL_E_UNSUPPORTED, "%s: obs_dim %d outside [1, 64]", who, d->D);
  if (d->n_heads < 1 || d->n_heads > 2) return fail(ORL_E_UNSUPPORTED, "%s: %d heads", who, d->n_heads);
  const int ntot = d->head_n[0] + (d->n_heads > 1 ? d->head_n[1] : 0);
  if (d->head_n[0] < 1 || ntot > GT_HEADS) return fail(ORL_E_UNSUPPORTED, "%s: %d head outputs (<= 16)", who, ntot);
  for (int l = 0; l < d->n_layers; ++l)
    if (d->act[l] < ORL_ACT_NONE || d->act[l] > ORL_ACT_ELU) return fail(ORL_E_INVALID, "%s: activation %d", who, d->act[l]);
  const GtLay ly(*d);
  if (gt_lds_bytes(ly, 8) > 160 * 1024)
    return fail(ORL_E_UNSUPPORTED, "%s: %zu bytes of LDS (obs_dim %d at hidden_size %d)", who, gt_lds_bytes(ly, 8), d->D, d->H);
  return 0;
}

// bwd_waves = 0: forward
static int gt_launch(const GtArgs& A, int bwd_waves, int grid, size_t lds, hipStream_t s) {
  const int H = A.d.H, NL = A.d.n_layers - 1, ND = A.d.D <= 16 ? 1 : 4;
  ORL_GT_CASE(64, 1)
  ORL_GT_CASE(64, 2)
  ORL_GT_CASE(64, 3)
  if (H == 128) return gt_launch_h128(A, bwd_waves, grid, lds, s);
  return fail(ORL_E_UNSUPPORTED, "orl_gt: no kernel for hidden_size %d with %d layers", H, NL + 1);
}

}  // namespace orl

using namespace orl;

extern "C" {

#ifdef ORL_PROF
int orl_gt_debug_prof(unsigned long long* out16) {
  unsigned long long zero[16] = {0};
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_gt_prof), sizeof(zero));
  hipMemcpyToSymbol(HIP_SYMBOL(g_gt_prof), zero, sizeof(zero));
  return 0;
}
#endif

int orl_gt_supported(const orl_gt_desc* d) { return gt_check(d, "orl_gt_supported") == 0 ? 1 : 0; }

int64_t orl_gt_image_floats(const orl_gt_desc* d) {
  if (gt_check(d, "orl_gt_image_floats")) return -1;
  return GtLay(*d).img_total;
}

int64_t orl_gt_raw_floats(const orl_gt_desc* d) {
  if (gt_check(d, "orl_gt_raw_floats")) return -1;
  return GtLay(*d).raw_total;
}

int orl_gt_prep(const orl_gt_desc* d, float* image, void* stream) {
  int rc = gt_check(d, "orl_gt_prep");
  if (rc) return rc;
  ORL_REQUIRE(image, "orl_gt_prep: null image");
  hipLaunchKernelGGL(gt_prep_kernel, dim3(128), dim3(256), 0, (hipStream_t)stream, *d, image);
  return launch_status("orl_gt_prep");
}

int orl_gt_fwd(const orl_gt_desc* d, const float* image, const float* x, int ldx, int col0, const int64_t* idx, int mb,
               float* head_out0, float* head_out1, void* stream) {
  int rc = gt_check(d, "orl_gt_fwd");
  if (rc) return rc;
  ORL_REQUIRE(image && x && head_out0 && mb > 0 && ldx >= col0 + d->D && col0 >= 0, "orl_gt_fwd: bad arguments");
  ORL_REQUIRE(d->n_heads == 1 || head_out1, "orl_gt_fwd: two heads need two outputs");
  const GtLay ly(*d);
  GtArgs A{};
  A.d = *d; A.image = image; A.x = x; A.ldx = ldx; A.col0 = col0; A.idx = (const long long*)idx; A.mb = mb;
  A.out0 = head_out0; A.out1 = head_out1;
  const int n_pass = ((mb + 15) / 16 + GT_WAVES - 1) / GT_WAVES;
  const size_t lds_fwd = gt_lds_bytes(ly, 0);
  const int per_cu = lds_fwd <= 80 * 1024 ? 2 : 1;
  const int grid = n_pass < 256 * per_cu ? n_pass : 256 * per_cu;
  rc = gt_launch(A, 0, grid, lds_fwd, (hipStream_t)stream);
  if (rc) return rc;
  return launch_status("orl_gt_fwd");
}

static int gt_backward(const orl_gt_desc* d, const GtLossArgs& loss, const float* image, const float* x, int ldx, int col0,
                       const int64_t* idx, int mb, const float* dhead0, const float* dhead1, float* partials,
                       int64_t partials_floats, float* raw, float* grad, float* sums_out, void* stream, const char* who) {
  const GtLay ly(*d);
  const int row_floats = ly.raw_total + GT_LOSS_SUMS;
  ORL_REQUIRE(partials_floats >= row_floats, "%s: the partials buffer holds %lld floats, one row is %d", who,
              (long long)partials_floats, row_floats);
  GtArgs A{};
  A.d = *d; A.loss = loss; A.image = image; A.x = x; A.ldx = ldx; A.col0 = col0; A.idx = (const long long*)idx; A.mb = mb;
  A.dh0 = dhead0; A.dh1 = dhead1; A.partials = partials;
  const int nw = gt_bwd_waves(ly);
  const size_t lds_bwd = gt_lds_bytes(ly, nw);
  const int n_pass = ((mb + 15) / 16 + nw - 1) / nw;
  const int max_grid = 256 * ((lds_bwd <= 80 * 1024 && ORL_GT_BWD_MINWAVES > 1) ? 2 : 1);
  int grid = n_pass < max_grid ? n_pass : max_grid;
  const int64_t fit = partials_floats / row_floats;
  if (grid > fit) grid = (int)fit;
  int rc = gt_launch(A, nw, grid, lds_bwd, (hipStream_t)stream);
  if (rc) return rc;
  rc = launch_status(who);
  if (rc) return rc;
  rc = orl_gen_colsum(partials, grid, row_floats, raw, ly.raw_total, sums_out, GT_LOSS_SUMS, nullptr, 0, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(gt_finalize_kernel, dim3(128), dim3(256), 0, (hipStream_t)stream, *d, raw, grad);
  return launch_status(who);
}

int orl_gt_bwd(const orl_gt_desc* d, const float* image, const float* x, int ldx, int col0, const int64_t* idx, int mb,
               const float* dhead0, const float* dhead1, float* partials, int64_t partials_floats, float* raw, float* grad,
               void* stream) {
  int rc = gt_check(d, "orl_gt_bwd");
  if (rc) return rc;
  ORL_REQUIRE(image && x && dhead0 && partials && raw && grad && mb > 0 && ldx >= col0 + d->D && col0 >= 0,
              "orl_gt_bwd: bad arguments");
  ORL_REQUIRE(d->n_heads == 1 || dhead1, "orl_gt_bwd: two heads need two gradients");
  GtLossArgs loss{};
  return gt_backward(d, loss, image, x, ldx, col0, idx, mb, dhead0, dhead1, partials, partials_floats, raw, grad, nullptr,
                     stream, "orl_gt_bwd");
}

int orl_gt_train(const orl_gt_desc* d, const float* image, const float* records, int rec_width, int col0,
                 const int64_t* idx, int mb, const orl_gt_loss* L, float* partials, int64_t partials_floats, float* raw,
                 float* grad, float* sums_out, void* stream) {
  int rc = gt_check(d, "orl_gt_train");
  if (rc) return rc;
  ORL_REQUIRE(image && records && L && partials && raw && grad && sums_out && mb > 0 && col0 >= 0 &&
                  rec_width >= col0 + d->D, "orl_gt_train: bad arguments");
  ORL_REQUIRE(L->den, "orl_gt_train: null denominators");
  ORL_REQUIRE(orl_record_width(L->Dp, L->Dc, L->a_w, L->K) == rec_width, "orl_gt_train: record width %d != %d", rec_width,
              orl_record_width(L->Dp, L->Dc, L->a_w, L->K));
  ORL_REQUIRE(L->policy_head < d->n_heads && L->value_head < d->n_heads && (L->policy_head >= 0 || L->value_head >= 0) &&
                  L->policy_head != L->value_head, "orl_gt_train: heads %d / %d of %d", L->policy_head, L->value_head,
              d->n_heads);
  if (L->policy_head >= 0) {
    ORL_REQUIRE(L->head.n_out == d->head_n[L->policy_head], "orl_gt_train: the policy head has %d outputs, the loss %d",
                d->head_n[L->policy_head], L->head.n_out);
    ORL_REQUIRE((L->head.kind != ORL_HEAD_GAUSSIAN && L->head.kind != ORL_HEAD_MIXED) || L->logstd,
                "orl_gt_train: Gaussian head without logstd");
  }
  if (L->value_head >= 0) ORL_REQUIRE(d->head_n[L->value_head] == 1, "orl_gt_train: the value head must have one output");
  GtLossArgs loss{};
  loss.on = 1;
  loss.policy_head = L->policy_head; loss.value_head = L->value_head; loss.policy_grad = L->policy_grad;
  loss.hd = L->head; loss.logstd = L->logstd; loss.den = L->den; loss.vn_state = L->vn_state; loss.hp = L->hp;
  loss.c = gen_cols(L->Dp, L->Dc, L->a_w, L->K);
  return gt_backward(d, loss, image, records, rec_width, col0, idx, mb, nullptr, nullptr, partials, partials_floats, raw,
                     grad, sums_out, stream, "orl_gt_train");
}

}  // extern "C"
