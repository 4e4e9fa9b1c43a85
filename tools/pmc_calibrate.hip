// pmc_calibrate.hip - what do rocprofv3's FETCH_SIZE / WRITE_SIZE report per byte on gfx950, by access pattern?
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/pmc_cal tools/pmc_calibrate.hip
//   rocprofv3 --pmc FETCH_SIZE -d out1 --output-format csv -- /tmp/pmc_cal
//   rocprofv3 --pmc WRITE_SIZE -d out2 --output-format csv -- /tmp/pmc_cal        (tools/pmc_calibrate.sh does both)
//
// MI355X_MICROARCH.md: FETCH_SIZE reports exactly 1/2 of the bytes of a wide (16 B / lane) streaming read and is
// uncalibrated for other widths, WRITE_SIZE uncalibrated.  The PPO kernels use three patterns - 16-byte
// global_load_lds DMA (record fetch of the tower kernels), 4-byte-per-lane coalesced rows (GAE scan, pack) and
// 16-byte rows - so each is measured here on a known byte count (512 MiB, larger than the 256 MiB Infinity Cache).
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f4 __attribute__((ext_vector_type(4)));
#define N_BYTES (512ll << 20)

__global__ __launch_bounds__(256) void read16_write16(const f4* __restrict__ in, f4* __restrict__ out, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    out[i] = in[i] * 2.0f;
}
__global__ __launch_bounds__(256) void read4_write4(const float* __restrict__ in, float* __restrict__ out, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    out[i] = in[i] * 2.0f;
}
// 16-byte global -> LDS DMA (one 1 KiB piece per wave instruction), result summed so the loads are not dead
__global__ __launch_bounds__(256) void dma16_read(const float* __restrict__ in, float* __restrict__ out, long long n_f) {
  __shared__ __attribute__((aligned(16))) float lds[4 * 256];
  const int wave = threadIdx.x >> 6, l = threadIdx.x & 63;
  float acc = 0.f;
  const long long per_it = (long long)gridDim.x * 4 * 256;
  for (long long base = ((long long)blockIdx.x * 4 + wave) * 256; base < n_f; base += per_it) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(in + base + 4 * l),
                                     (__attribute__((address_space(3))) void*)(lds + wave * 256), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    acc += lds[wave * 256 + l] + lds[wave * 256 + 64 + l] + lds[wave * 256 + 128 + l] + lds[wave * 256 + 192 + l];
  }
  if (acc == 123.456f) out[threadIdx.x] = acc;
}
// the tower kernels' record gather: 64-byte rows at permuted positions, 16 B per lane
__global__ __launch_bounds__(256) void gather64_read(const f4* __restrict__ in, float* __restrict__ out, long long n_rows) {
  float acc = 0.f;
  const long long stride = (long long)gridDim.x * blockDim.x / 4;
  for (long long r = ((long long)blockIdx.x * blockDim.x + threadIdx.x) / 4; r < n_rows; r += stride) {
    const long long row = (r * 2654435761ll) % n_rows;  // a permutation-like scatter of whole rows
    const f4 v = in[row * 4 + (threadIdx.x & 3)];
    acc += v[0] + v[1] + v[2] + v[3];
  }
  if (acc == 123.456f) out[threadIdx.x] = acc;
}

int main() {
  float *a, *b;
  hipMalloc(&a, N_BYTES);
  hipMalloc(&b, N_BYTES);
  hipMemset(a, 0, N_BYTES);
  hipMemset(b, 0, N_BYTES);
  const long long nf = N_BYTES / 4;
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(read16_write16, dim3(4096), dim3(256), 0, 0, (const f4*)a, (f4*)b, nf / 4);
    hipLaunchKernelGGL(read4_write4, dim3(4096), dim3(256), 0, 0, a, b, nf);
    hipLaunchKernelGGL(dma16_read, dim3(2048), dim3(256), 0, 0, a, b, nf);
    hipLaunchKernelGGL(gather64_read, dim3(4096), dim3(256), 0, 0, (const f4*)a, b, nf / 16);
  }
  hipDeviceSynchronize();
  printf("bytes per kernel: read %lld, write %lld (copies) / 0 (read-only kernels)\n", (long long)N_BYTES, (long long)N_BYTES);
  return 0;
}
