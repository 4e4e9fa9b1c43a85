// ds_read_tr_probe.hip - what exactly does gfx950's ds_read_b64_tr_b16 return?
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/trp tools/ds_read_tr_probe.hip && /tmp/trp
//
// LDS is filled with the 16-bit value of its own index (element e holds e).  Every lane passes its own byte address;
// three address patterns are tried and the four 16-bit results of every lane are printed, so the lane -> (source lane,
// element) mapping can be read off:
//   A  lane l -> byte address 8 * l                (lane-linear: each lane "owns" 4 consecutive elements)
//   B  lane l -> row (l & 15) of a matrix with 64-byte rows, column block (l >> 4): address (l & 15) * 64 + (l >> 4) * 8
//   C  all lanes -> address 0
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ void probe(unsigned* out, int pattern) {
  __shared__ unsigned short lds[4096];
  for (int e = threadIdx.x; e < 4096; e += 64) lds[e] = (unsigned short)e;
  __syncthreads();
  const int l = threadIdx.x;
  unsigned addr = 0;
  if (pattern == 0) addr = 8 * l;
  else if (pattern == 1) addr = (l & 15) * 64 + (l >> 4) * 8;
  const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned short*)lds + addr;
  unsigned lo, hi;
  unsigned long long v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(base) : "memory");
  lo = (unsigned)v;
  hi = (unsigned)(v >> 32);
  out[2 * l] = lo;
  out[2 * l + 1] = hi;
}

int main() {
  unsigned* d;
  hipMalloc(&d, 128 * sizeof(unsigned));
  const char* names[3] = {"A: addr = 8*lane", "B: addr = (lane&15)*64 + (lane>>4)*8", "C: addr = 0 for all lanes"};
  for (int p = 0; p < 3; ++p) {
    probe<<<1, 64>>>(d, p);
    unsigned h[128];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("pattern %s\n", names[p]);
    for (int l = 0; l < 64; ++l) {
      printf("  lane %2d: %4u %4u %4u %4u", l, h[2 * l] & 0xffff, h[2 * l] >> 16, h[2 * l + 1] & 0xffff, h[2 * l + 1] >> 16);
      if ((l & 3) == 3) printf("\n");
    }
  }
  return 0;
}
