// split_bf16_gemm.hip - can three-term bf16 splitting replace the fp32-input MFMA in the PPO tower kernel?
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/sbg tools/split_bf16_gemm.hip && /tmp/sbg
//
// The tower kernel (openrl_amd/csrc/orl_ppo_tower.h) spends 56 % of its SIMD cycles in v_mfma_f32_16x16x4_f32, which
// runs at the fp32 VECTOR rate (32 cycles per 2 048 flop) and blocks the SIMD's VALU meanwhile
// (tools/mfma_valu_overlap.hip).  v_mfma_f32_16x16x32_bf16 retires 16 384 flop in ~16 cycles and does overlap VALU.
// An fp32 number is EXACTLY hi + mid + lo with three bf16 terms when each term is a truncation of the running
// remainder (8 + 8 + 8 significand bits), so a . b = sum of 9 bf16 products; dropping the three smallest
// (mid.lo, lo.mid, lo.lo <= 2^-24 |a||b|) leaves 6 products whose error is of the size of one fp32 rounding.
//
// One wave-phase of the tower = Z^T[64 x 16] = W[64 x 64] X^T[64 x 16] (T layout of orl_mlp.h: the C fragment of one
// GEMM is the B operand of the next, weights come from LDS).  Measured here:
//   accuracy  max / rms error against an fp64 reference on N(0,1) inputs: native fp32 MFMA, bf16x3 with 6 products,
//             bf16x3 with 9 products, plain bf16 (1 product)
//   cycles    per chained wave-phase (output -> scaled -> next input), 1 and 2 waves per SIMD:
//               native  64 x v_mfma_f32_16x16x4_f32, A = 16-byte LDS reads                        (mm64_T today)
//               split   48 x v_mfma_f32_16x16x32_bf16, A = pre-split bf16 images in LDS, the activation operand is
//                       split in registers every phase (11 VALU per 2 elements)
//               split-mfma-only  the same 48 MFMAs on an operand split once (the MFMA-side floor)
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define W2S 68   // fp32 image row stride (floats)
#define WBS 72   // bf16 image row stride (elements): 144-byte rows, 16 lanes x 16 bytes hit 64 distinct banks

__device__ __forceinline__ unsigned fbits(float x) { return __builtin_bit_cast(unsigned, x); }
__device__ __forceinline__ float bitsf(unsigned u) { return __builtin_bit_cast(float, u); }
// {hi16(b), hi16(a)} -> one dword, a in the low half
__device__ __forceinline__ unsigned pack_hi(float a, float b) {
  return __builtin_amdgcn_perm(fbits(b), fbits(a), 0x07060302u);
}

// split the 8 activations of one K-step (x[0..3] = in[2h], x[4..7] = in[2h+1]) into three bf16x8 fragments
__device__ __forceinline__ void split8(const float (&x)[8], u32x4& hi, u32x4& mid, u32x4& lo) {
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const float a = x[2 * p], b = x[2 * p + 1];
    hi[p] = pack_hi(a, b);
    const float ra = a - bitsf(fbits(a) & 0xffff0000u), rb = b - bitsf(fbits(b) & 0xffff0000u);
    mid[p] = pack_hi(ra, rb);
    const float sa = ra - bitsf(fbits(ra) & 0xffff0000u), sb = rb - bitsf(fbits(rb) & 0xffff0000u);
    lo[p] = pack_hi(sa, sb);
  }
}

__device__ __forceinline__ f32x4 mfma_bf16(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// bf16 weight images: part p (0 hi, 1 mid, 2 lo), row o, K-step h, lane group q, slot s <-> feature 16(2h + s/4) + 4q + s%4
__device__ __forceinline__ int wb_off(int p, int o, int h, int q) { return (p * 64 + o) * WBS + h * 32 + q * 8; }

__device__ void stage_weights(const float* __restrict__ W, float* __restrict__ Wf, unsigned short* __restrict__ Wb, int tid,
                              int nt) {
  for (int e = tid; e < 64 * 64; e += nt) {
    const int o = e >> 6, f = e & 63;
    const float w = W[e];
    Wf[o * W2S + f] = w;
    const int m = f >> 4, q = (f >> 2) & 3, r = f & 3, h = m >> 1, s = (m & 1) * 4 + r;
    const float h1 = bitsf(fbits(w) & 0xffff0000u), r1 = w - h1;
    const float m1 = bitsf(fbits(r1) & 0xffff0000u), l1 = r1 - m1;
    Wb[wb_off(0, o, h, q) + s] = (unsigned short)(fbits(h1) >> 16);
    Wb[wb_off(1, o, h, q) + s] = (unsigned short)(fbits(m1) >> 16);
    Wb[wb_off(2, o, h, q) + s] = (unsigned short)(fbits(l1) >> 16);
  }
}

// native: acc += W in  (orl_mlp.h::mm64_T)
__device__ __forceinline__ void mm_native(const float* __restrict__ Wf, const f32x4 (&in)[4], f32x4 (&acc)[4], int j, int q) {
  f32x4 a4[2][4];
#pragma unroll
  for (int mo = 0; mo < 4; ++mo) a4[0][mo] = *(const f32x4*)(Wf + (16 * mo + j) * W2S + 4 * q);
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    if (mi < 3) {
#pragma unroll
      for (int mo = 0; mo < 4; ++mo) a4[(mi + 1) & 1][mo] = *(const f32x4*)(Wf + (16 * mo + j) * W2S + 16 * (mi + 1) + 4 * q);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int mo = 0; mo < 4; ++mo)
        acc[mo] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[mi & 1][mo][r], in[mi][r], acc[mo], 0, 0, 0);
  }
}

// split: NP = 1 (hi.hi), 6 or 9 products.  xs[h][part] = the activation fragments of K-step h
template <int NP>
__device__ __forceinline__ void mm_split_frag(const unsigned short* __restrict__ Wb, const u32x4 (&xs)[2][3], f32x4 (&acc)[4],
                                              int j, int q) {
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int mo = 0; mo < 4; ++mo) {
      const u32x4 wh = *(const u32x4*)(Wb + wb_off(0, 16 * mo + j, h, q));
      // smallest terms first: the accumulator sees them before the large hi.hi term
      if (NP >= 6) {
        const u32x4 wm = *(const u32x4*)(Wb + wb_off(1, 16 * mo + j, h, q));
        const u32x4 wl = *(const u32x4*)(Wb + wb_off(2, 16 * mo + j, h, q));
        if (NP == 9) {
          acc[mo] = mfma_bf16(wl, xs[h][2], acc[mo]);
          acc[mo] = mfma_bf16(wl, xs[h][1], acc[mo]);
          acc[mo] = mfma_bf16(wm, xs[h][2], acc[mo]);
        }
        acc[mo] = mfma_bf16(wl, xs[h][0], acc[mo]);
        acc[mo] = mfma_bf16(wh, xs[h][2], acc[mo]);
        acc[mo] = mfma_bf16(wm, xs[h][1], acc[mo]);
        acc[mo] = mfma_bf16(wm, xs[h][0], acc[mo]);
        acc[mo] = mfma_bf16(wh, xs[h][1], acc[mo]);
      }
      acc[mo] = mfma_bf16(wh, xs[h][0], acc[mo]);
    }
  }
}

__device__ __forceinline__ void split_in(const f32x4 (&in)[4], u32x4 (&xs)[2][3]) {
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    float x[8];
#pragma unroll
    for (int r = 0; r < 4; ++r) { x[r] = in[2 * h][r]; x[4 + r] = in[2 * h + 1][r]; }
    split8(x, xs[h][0], xs[h][1], xs[h][2]);
  }
}

// ---------------------------------------------------------------------------------------------- accuracy
// X: [16 rows][64 features]; out[v]: [64 o][16 rows] for v = native, split6, split9, bf16x1
__global__ void accuracy_kernel(const float* __restrict__ W, const float* __restrict__ X, float* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) float Wf[64 * W2S];
  __shared__ __attribute__((aligned(16))) unsigned short Wb[3 * 64 * WBS];
  stage_weights(W, Wf, Wb, threadIdx.x, blockDim.x);
  __syncthreads();
  const int l = threadIdx.x & 63, j = l & 15, q = l >> 4;
  f32x4 in[4];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) in[m][r] = X[j * 64 + 16 * m + 4 * q + r];
  u32x4 xs[2][3];
  split_in(in, xs);
  for (int v = 0; v < 4; ++v) {
    f32x4 acc[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (v == 0) mm_native(Wf, in, acc, j, q);
    else if (v == 1) mm_split_frag<6>(Wb, xs, acc, j, q);
    else if (v == 2) mm_split_frag<9>(Wb, xs, acc, j, q);
    else mm_split_frag<1>(Wb, xs, acc, j, q);
    // C fragment: lane (j, q) reg r of block mo = out[o = 16mo + 4q + r][row j]
#pragma unroll
    for (int mo = 0; mo < 4; ++mo)
#pragma unroll
      for (int r = 0; r < 4; ++r) out[v * 1024 + (16 * mo + 4 * q + r) * 16 + j] = acc[mo][r];
  }
}

// ---------------------------------------------------------------------------------------------- timing
// MODE 0 native, 1 split (operand split every phase), 2 split MFMAs only (operand split once)
template <int MODE>
__global__ __launch_bounds__(512) void timing_kernel(const float* __restrict__ W, const float* __restrict__ X,
                                                     float* __restrict__ out, long long* __restrict__ cyc, int iters) {
  __shared__ __attribute__((aligned(16))) float Wf[64 * W2S];
  __shared__ __attribute__((aligned(16))) unsigned short Wb[3 * 64 * WBS];
  stage_weights(W, Wf, Wb, threadIdx.x, blockDim.x);
  __syncthreads();
  const int l = threadIdx.x & 63, j = l & 15, q = l >> 4, wave = threadIdx.x >> 6;
  f32x4 in[4];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) in[m][r] = X[j * 64 + 16 * m + 4 * q + r];
  u32x4 xs[2][3];
  split_in(in, xs);
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    f32x4 acc[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (MODE == 0) mm_native(Wf, in, acc, j, q);
    else {
      if (MODE == 1) split_in(in, xs);
      mm_split_frag<6>(Wb, xs, acc, j, q);
    }
    // chain: the next phase's operand is this phase's result (kept O(1): W is N(0,1)/8)
#pragma unroll
    for (int m = 0; m < 4; ++m) in[m] = acc[m];
    if (MODE == 2) xs[0][0][0] ^= (fbits(acc[0][0]) & 1u);  // keep the MFMAs dependent on the previous phase
  }
  const long long t1 = __builtin_readcyclecounter();
  if (l == 0) cyc[wave] = t1 - t0;
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) out[(threadIdx.x * 16) + 4 * m + r] = in[m][r];
}

static double gauss() {
  double u1 = (rand() + 1.0) / (RAND_MAX + 2.0), u2 = (rand() + 1.0) / (RAND_MAX + 2.0);
  return sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
}

template <int MODE>
static void time_mode(const char* name, int threads, const float* dW, const float* dX, float* dout, long long* dcyc) {
  const int iters = 2048;
  timing_kernel<MODE><<<1, threads>>>(dW, dX, dout, dcyc, 16);
  timing_kernel<MODE><<<1, threads>>>(dW, dX, dout, dcyc, iters);
  hipDeviceSynchronize();
  long long h[8];
  hipMemcpy(h, dcyc, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-54s waves/SIMD %d: %8.1f cycles per wave-phase (wave 0)", name, threads / 256, (double)h[0] / iters);
  if (threads > 256) printf(", %8.1f (wave 4)", (double)h[4] / iters);
  printf("\n");
}

int main() {
  srand(12345);
  std::vector<float> W(4096), X(1024);
  float *dW, *dX, *dout;
  long long* dcyc;
  hipMalloc(&dW, 4096 * 4); hipMalloc(&dX, 1024 * 4); hipMalloc(&dout, 512 * 16 * 4 + 4 * 1024 * 4); hipMalloc(&dcyc, 64);
  printf("== accuracy: Z^T[64 x 16] = W[64 x 64] X^T, W and X ~ N(0,1), error against fp64 (20 draws)\n");
  const char* names[4] = {"native fp32 MFMA (v_mfma_f32_16x16x4_f32)", "bf16x3, 6 products (48 x v_mfma_f32_16x16x32_bf16)",
                          "bf16x3, 9 products (72 MFMAs)", "plain bf16 (hi.hi only, 8 MFMAs)"};
  double maxe[4] = {0, 0, 0, 0}, sse[4] = {0, 0, 0, 0}, ssr = 0;
  long long n = 0;
  for (int draw = 0; draw < 20; ++draw) {
    for (auto& w : W) w = (float)gauss();
    for (auto& x : X) x = (float)gauss();
    hipMemcpy(dW, W.data(), 4096 * 4, hipMemcpyHostToDevice);
    hipMemcpy(dX, X.data(), 1024 * 4, hipMemcpyHostToDevice);
    accuracy_kernel<<<1, 64>>>(dW, dX, dout);
    std::vector<float> o(4096);
    hipMemcpy(o.data(), dout, 4096 * 4, hipMemcpyDeviceToHost);
    for (int oo = 0; oo < 64; ++oo)
      for (int jj = 0; jj < 16; ++jj) {
        double ref = 0;
        for (int f = 0; f < 64; ++f) ref += (double)W[oo * 64 + f] * (double)X[jj * 64 + f];
        ssr += ref * ref;
        ++n;
        for (int v = 0; v < 4; ++v) {
          const double e = fabs((double)o[v * 1024 + oo * 16 + jj] - ref);
          if (e > maxe[v]) maxe[v] = e;
          sse[v] += e * e;
        }
      }
  }
  const double rms_ref = sqrt(ssr / n);
  for (int v = 0; v < 4; ++v)
    printf("  %-52s max |err| %.3e   rms err %.3e   (rms |Z| %.2f -> relative rms %.2e)\n", names[v], maxe[v],
           sqrt(sse[v] / n), rms_ref, sqrt(sse[v] / n) / rms_ref);

  printf("== cycles per chained wave-phase (s_memtime = shader cycles), one workgroup\n");
  for (auto& w : W) w = (float)(gauss() / 8.0);
  hipMemcpy(dW, W.data(), 4096 * 4, hipMemcpyHostToDevice);
  for (int t = 256; t <= 512; t += 256) {
    time_mode<0>("native: 64 fp32 MFMA", t, dW, dX, dout, dcyc);
    time_mode<2>("split, MFMAs only: 48 bf16 MFMA (operand split once)", t, dW, dX, dout, dcyc);
    time_mode<1>("split: 48 bf16 MFMA + operand split each phase (88 VALU)", t, dW, dX, dout, dcyc);
  }
  return 0;
}
