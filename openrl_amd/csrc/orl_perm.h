// orl_perm.h - keyed minibatch permutation + ValueNorm.update bodies shared by orl_buffer.hip (stand-alone launches)
// and orl_ppo.hip (the same work riding in the optimiser-step launch of the previous epoch).
#pragma once
#include "orl_common.h"

namespace orl {

// ------------------------------------------------------------------------------------------------
// Keyed permutation of [0,n): balanced Feistel network over 2*hb bits (2^(2hb) >= n) with
// cycle walking.  The four 32-bit round keys come from ONE Philox4x32-10 block of (stream_id) under
// `seed`; the round function is the murmur3 32-bit finaliser of (half ^ key) - a bijection-preserving
// Feistel needs no more than a well-mixed round function, and this keeps the kernel store-bound.
// ------------------------------------------------------------------------------------------------
__host__ __device__ inline uint32_t fmix32(uint32_t x) {
  x ^= x >> 16;
  x *= 0x85EBCA6Bu;
  x ^= x >> 13;
  x *= 0xC2B2AE35u;
  x ^= x >> 16;
  return x;
}

__host__ __device__ inline uint64_t feistel_perm(uint64_t i, uint64_t n, int hb, const u4& keys) {
  const uint32_t hmask = (uint32_t)(((uint64_t)1 << hb) - 1);
  const uint32_t k[4] = {keys.x, keys.y, keys.z, keys.w};
  uint64_t x = i;
  do {
    uint32_t lft = (uint32_t)(x >> hb), rgt = (uint32_t)x & hmask;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const uint32_t fv = fmix32(rgt ^ k[r]) & hmask;
      const uint32_t nl = rgt;
      rgt = lft ^ fv;
      lft = nl;
    }
    x = ((uint64_t)lft << hb) | rgt;
  } while (x >= n);
  return x;
}

// ValueNorm.update (valuenorm.py:58-77) from batch sums {sum, sum of squares, count}
__device__ __forceinline__ void valuenorm_update_body(float* __restrict__ vn, const double* __restrict__ mom, float beta, float omw) {
  // included from translation units built with and without -ffast-math: pin the IEEE evaluation here
#pragma clang fp contract(off) reassociate(off) reciprocal(off)
  const double cnt = mom[2];
  const float bm = (float)(mom[0] / cnt);
  const float bsq = (float)(mom[1] / cnt);
  // omw = float32(1.0 - weight) with the subtraction done in double on the host (python float semantics)
  const float t0 = bm * omw;
  const float t1 = bsq * omw;
  vn[0] = vn[0] * beta + t0;
  vn[1] = vn[1] * beta + t1;
  vn[2] = vn[2] * beta + omw;
}


// One permutation job: idx[i] = feistel_perm(i) for i < n and, when vn != NULL, the ValueNorm.update that precedes the
// loss of the epoch the permutation is for (ppo.py:190-195; the two are independent).
struct PermJob {
  int64_t* idx;
  long long n;
  int hb;
  uint64_t seed, stream_id;
  float* vn;
  const double* mom;
  float beta, omw;
};

inline PermJob make_perm_job(int64_t* idx, int64_t n, uint64_t seed, uint64_t stream_id, float* vn, const double* mom,
                             double beta) {
  int bits = 1;
  while (((int64_t)1 << bits) < n) ++bits;
  PermJob J;
  J.idx = idx; J.n = (long long)n; J.hb = (bits + 1) / 2; J.seed = seed; J.stream_id = stream_id;
  J.vn = vn; J.mom = mom; J.beta = (float)beta; J.omw = (float)(1.0 - beta);
  return J;
}

// workgroup b of the nb workgroups that share the job (any block size)
__device__ inline void perm_job_block(const PermJob& J, int b, int nb) {
  if (b == 0 && threadIdx.x == 0 && J.vn != nullptr) valuenorm_update_body(J.vn, J.mom, J.beta, J.omw);
  const u4 keys = philox4x32_10(J.seed, (uint32_t)J.stream_id, (uint32_t)(J.stream_id >> 32), 0x5EED5EEDu, 0u);
  const long long stride = (long long)nb * blockDim.x;
  for (long long i = (long long)b * blockDim.x + threadIdx.x; i < J.n; i += stride)
    J.idx[i] = (int64_t)feistel_perm((uint64_t)i, (uint64_t)J.n, J.hb, keys);
}

}  // namespace orl
