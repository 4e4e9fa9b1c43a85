// mfma_valu_overlap.hip - does a gfx950 SIMD overlap fp32-input MFMA with fp32 VALU work?
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mvo tools/mfma_valu_overlap.hip && /tmp/mvo
//
// The fused PPO tower kernel (openrl_amd/csrc/orl_ppo_tower.h) is priced against the fp32 MFMA peak, but its tiles
// also carry ~770 VALU instructions per 196 MFMAs.  Whether those can hide under the MFMAs decides what the kernel's
// real ceiling is.  Each test runs ONE workgroup and reads the shader clock (s_memtime) inside the kernel:
//   A  mfma  : one wave per SIMD, 16 independent accumulators, MFMAs only
//   B  valu  : one wave per SIMD, K independent v_fma_f32 per MFMA slot, no MFMA
//   C  mixed : one wave per SIMD, the SAME wave issues 1 MFMA + K v_fma_f32, interleaved
//   D  split : two waves per SIMD: waves 0-3 run A's stream, waves 4-7 run B's stream (same SIMDs)
// If the pipes were independent, C and D would take max(A, B); if MFMA_F32 executes on the VALU datapath they take
// A + B.  The same four tests are repeated with a bf16 MFMA (v_mfma_f32_16x16x32_bf16) as the control.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define ITERS 512
#define NACC 16

// Every instruction of the measured streams is a volatile asm statement: hipcc keeps volatile statements in program
// order, so the interleave below is exactly what the SIMD sees (pure builtins float across sched_barrier at IR level).
// None of the VALU registers is an MFMA operand and 16 accumulators rotate, so no wait state is needed inside.
template <int K>
__device__ __forceinline__ void valu_block(float (&v)[8], float m) {
#pragma unroll
  for (int k = 0; k < K; ++k) asm volatile("v_fma_f32 %0, %0, %1, 1.0" : "+v"(v[k & 7]) : "v"(m));
}
__device__ __forceinline__ void mfma_f32(f32x4& acc, float a, float b) {
  asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_bf16(f32x4& acc, bf16x8 a, bf16x8 b) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}

// MODE: 0 mfma only, 1 valu only, 2 mixed in one wave, 3 split by wave half.  BF: 0 fp32 MFMA, 1 bf16 MFMA
template <int MODE, int K, int BF>
__global__ __launch_bounds__(1024) void probe(float* out, long long* cycles, float seed) {
  const int wave = threadIdx.x >> 6;
  const bool do_mfma = MODE == 0 || MODE == 2 || (MODE == 3 && wave < 4);
  const bool do_valu = MODE == 1 || MODE == 2 || (MODE == 3 && wave >= 4);
  f32x4 acc[NACC];
#pragma unroll
  for (int a = 0; a < NACC; ++a) acc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
  float v[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = seed + k;
  const float a = seed * 0.5f + threadIdx.x, b = seed * 0.25f;
  bf16x8 ab, bb;
#pragma unroll
  for (int k = 0; k < 8; ++k) { ab[k] = (__bf16)(a + k); bb[k] = (__bf16)(b + k); }
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  if (do_mfma && do_valu) {
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
      for (int s = 0; s < NACC; ++s) {
        if (BF) mfma_bf16(acc[s], ab, bb);
        else mfma_f32(acc[s], a, b);
        valu_block<K>(v, b);
      }
    }
  } else if (do_mfma) {
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
      for (int s = 0; s < NACC; ++s) {
        if (BF) mfma_bf16(acc[s], ab, bb);
        else mfma_f32(acc[s], a, b);
      }
    }
  } else if (do_valu) {
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
      for (int s = 0; s < NACC; ++s) {
        valu_block<K>(v, b);
      }
    }
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // MFMA results -> VALU readers
  const long long t1 = __builtin_readcyclecounter();
  float r = 0.f;
#pragma unroll
  for (int s = 0; s < NACC; ++s) r += acc[s][0] + acc[s][1] + acc[s][2] + acc[s][3];
#pragma unroll
  for (int k = 0; k < 8; ++k) r += v[k];
  out[threadIdx.x] = r;
  if ((threadIdx.x & 63) == 0) cycles[wave] = t1 - t0;
}

template <int MODE, int K, int BF>
static void run(const char* name, int threads, float* out, long long* cyc) {
  long long h[16] = {0};
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL((probe<MODE, K, BF>), dim3(1), dim3(threads), 0, 0, out, cyc, 1.0f + rep);
    hipDeviceSynchronize();
  }
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  const double slots = (double)ITERS * NACC;
  printf("%-34s waves %d: wave0 %7.1f cyc/slot", name, threads / 64, h[0] / slots);
  if (threads >= 512) printf("   wave4 %7.1f cyc/slot", h[4] / slots);
  printf("\n");
}

int main() {
  float* out;
  long long* cyc;
  hipMalloc(&out, 1024 * sizeof(float));
  hipMalloc(&cyc, 16 * sizeof(long long));
  printf("one slot = 1 MFMA and/or K v_fma_f32 (wave64); s_memtime ticks = shader cycles\n");
  printf("---- fp32 MFMA v_mfma_f32_16x16x4_f32, K = 4\n");
  run<0, 4, 0>("A mfma only (1 wave/SIMD)", 256, out, cyc);
  run<1, 4, 0>("B valu only (1 wave/SIMD)", 256, out, cyc);
  run<2, 4, 0>("C mfma+valu in ONE wave", 256, out, cyc);
  run<3, 4, 0>("D mfma waves 0-3 | valu waves 4-7", 512, out, cyc);
  printf("---- VALU pipe occupancy: the same v_fma stream on 1 / 2 / 4 waves per SIMD (K = 4 and 8 independent chains)\n");
  run<1, 4, 0>("B valu only, 2 waves/SIMD", 512, out, cyc);
  run<1, 4, 0>("B valu only, 4 waves/SIMD", 1024, out, cyc);
  run<1, 8, 0>("B valu only K=8, 2 waves/SIMD", 512, out, cyc);
  run<1, 8, 0>("B valu only K=8, 4 waves/SIMD", 1024, out, cyc);
  run<0, 4, 0>("A mfma only, 2 waves/SIMD", 512, out, cyc);
  printf("---- fp32 MFMA, K = 8\n");
  run<1, 8, 0>("B valu only", 256, out, cyc);
  run<2, 8, 0>("C mfma+valu in ONE wave", 256, out, cyc);
  run<3, 8, 0>("D mfma waves 0-3 | valu waves 4-7", 512, out, cyc);
  printf("---- control: bf16 MFMA v_mfma_f32_16x16x32_bf16, K = 4\n");
  run<0, 4, 1>("A mfma only", 256, out, cyc);
  run<2, 4, 1>("C mfma+valu in ONE wave", 256, out, cyc);
  run<3, 4, 1>("D mfma waves 0-3 | valu waves 4-7", 512, out, cyc);
  printf("---- control: bf16 MFMA, K = 2\n");
  run<1, 2, 1>("B valu only", 256, out, cyc);
  run<2, 2, 1>("C mfma+valu in ONE wave", 256, out, cyc);
  run<3, 2, 1>("D mfma waves 0-3 | valu waves 4-7", 512, out, cyc);
  return 0;
}
