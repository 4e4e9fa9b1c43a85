// Probe (round 5): is  a - bf16_trunc(a)  through v_dot2c_f32_bf16 bit-identical to the v_and_b32 + v_sub_f32 form?
//   hipcc --offload-arch=gfx950 -O3 tools/split_dot2c_probe.hip -o /tmp/split_probe && /tmp/split_probe
// (First run, constants (lo = -1, hi = +0) / 0xbf800000: hipcc encoded the first as the inline constant -1.0, which the hardware
// reads as 0xbf800000 - every pair came out wrong.  With a sign bit in the unused half both constants are 32-bit literals.)
// Both forms of the three-term split run over 2^24 operand pairs: N(0,1) x 2^e for e in [-40, 40], every exponent of the
// normal range once, signed zeros, denormals.  Prints the number of differing (hi, mid, lo) words and the first few.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include <random>
#include <cmath>

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned f2u(float x) { return __builtin_bit_cast(unsigned, x); }
__device__ __forceinline__ float u2f(unsigned u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ unsigned pack_hi16(float a, float b) { return __builtin_amdgcn_perm(f2u(b), f2u(a), 0x07060302u); }

__global__ void probe(const float* __restrict__ x, unsigned* __restrict__ out_and, unsigned* __restrict__ out_dot, int n_pairs) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_pairs) return;
  const float a = x[2 * i], b = x[2 * i + 1];
  {  // and / sub
    const unsigned hi = pack_hi16(a, b);
    const float ra = a - u2f(f2u(a) & 0xffff0000u), rb = b - u2f(f2u(b) & 0xffff0000u);
    const unsigned mid = pack_hi16(ra, rb);
    const float sa = ra - u2f(f2u(ra) & 0xffff0000u), sb = rb - u2f(f2u(rb) & 0xffff0000u);
    out_and[3 * i] = hi; out_and[3 * i + 1] = mid; out_and[3 * i + 2] = pack_hi16(sa, sb);
  }
  {  // dot2c
    const bf16x2 m_lo = __builtin_bit_cast(bf16x2, 0x8000bf80u), m_hi = __builtin_bit_cast(bf16x2, 0xbf808000u);  // -0 in the unused half: no inline constant
    const unsigned hi = pack_hi16(a, b);
    const float ra = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, hi), m_lo, a, false);
    const float rb = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, hi), m_hi, b, false);
    const unsigned mid = pack_hi16(ra, rb);
    const float sa = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, mid), m_lo, ra, false);
    const float sb = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, mid), m_hi, rb, false);
    out_dot[3 * i] = hi; out_dot[3 * i + 1] = mid; out_dot[3 * i + 2] = pack_hi16(sa, sb);
  }
}

int main() {
  const int n_pairs = 1 << 24;
  std::vector<float> h(2 * (size_t)n_pairs);
  std::mt19937 g(1234);
  std::normal_distribution<float> nd(0.f, 1.f);
  for (size_t i = 0; i < h.size(); ++i) h[i] = std::ldexp(nd(g), (int)(i % 81) - 40);
  // edge cases at the front: every normal exponent with a full significand, +-0, denormals
  size_t k = 0;
  for (int e = 1; e < 255; ++e) { uint32_t u = ((uint32_t)e << 23) | 0x7fffffu; float f; std::memcpy(&f, &u, 4); h[k++] = f; h[k++] = -f; }
  h[k++] = 0.f; h[k++] = -0.f;
  for (uint32_t u : {1u, 0x7fffffu, 0x00012345u, 0x80000001u}) { float f; std::memcpy(&f, &u, 4); h[k++] = f; }
  const size_t n_edge_pairs = (k + 1) / 2;
  float* dx; unsigned *da, *dd;
  hipMalloc(&dx, h.size() * 4); hipMalloc(&da, 3 * (size_t)n_pairs * 4); hipMalloc(&dd, 3 * (size_t)n_pairs * 4);
  hipMemcpy(dx, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3((n_pairs + 255) / 256), dim3(256), 0, 0, dx, da, dd, n_pairs);
  std::vector<unsigned> a(3 * (size_t)n_pairs), d(3 * (size_t)n_pairs);
  hipMemcpy(a.data(), da, a.size() * 4, hipMemcpyDeviceToHost);
  hipMemcpy(d.data(), dd, d.size() * 4, hipMemcpyDeviceToHost);
  size_t diff = 0, diff_edge = 0, shown = 0;
  double worst = 0.0;
  for (size_t i = 0; i < (size_t)n_pairs; ++i) {
    bool bad = false;
    for (int p = 0; p < 3; ++p) bad |= a[3 * i + p] != d[3 * i + p];
    if (!bad) continue;
    ++diff;
    if (i < n_edge_pairs) ++diff_edge;
    // reconstruction error of the dot2c form for both values of the pair
    for (int half = 0; half < 2; ++half) {
      double rec = 0.0;
      for (int p = 0; p < 3; ++p) { uint32_t u = (half ? (d[3 * i + p] & 0xffff0000u) : (d[3 * i + p] << 16)); float f; std::memcpy(&f, &u, 4); rec += f; }
      const double x = h[2 * i + half];
      if (x != 0.0) worst = std::fmax(worst, std::fabs(rec - x) / std::fabs(x));
    }
    if (shown++ < 8)
      std::printf("  pair %zu (%a, %a): and/sub %08x %08x %08x  dot2c %08x %08x %08x\n", i, h[2 * i], h[2 * i + 1], a[3 * i],
                  a[3 * i + 1], a[3 * i + 2], d[3 * i], d[3 * i + 1], d[3 * i + 2]);
  }
  std::printf("split_dot2c_probe: %d pairs, %zu differ (%zu of the %zu edge-case pairs); worst relative reconstruction error of a "
              "differing dot2c split %.3e\n", n_pairs, diff, diff_edge, n_edge_pairs, worst);
  return 0;
}
