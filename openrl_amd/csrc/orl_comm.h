// orl_comm.h - device-side view + push / poll primitives of the one-shot small all-reduce (orl_comm.hip); shared with
// orl_apply.hip, which fuses the two halves into the PPO optimiser step.  Not part of the C ABI.
#pragma once
#include "orl_common.h"

#define ORL_COMM_MAX_WORLD 8
#define ORL_COMM_TIMEOUT_TICKS 1000000000ull  // 10 s of the 100 MHz wall clock: a missing peer is an error, not a hang

namespace orl {

struct CommDev {
  unsigned long long* inbox[ORL_COMM_MAX_WORLD];  // inbox[p] = rank p's inbox ([2 parities][world][cap] granules)
  int* err;                                       // local error word (set on a poll timeout)
  int rank, world, cap;
  unsigned seq;                                   // sequence number of THIS collective (never 0)
};

// granule of element i from rank `src` in the inbox of rank `dst` for the collective C.seq
__device__ inline unsigned long long* comm_slot(const CommDev& C, int dst, int src, int i) {
  return C.inbox[dst] + ((size_t)((C.seq & 1u) * C.world + src) * C.cap + i);
}

__device__ inline void comm_push(const CommDev& C, int peer, int i, float v) {
  const unsigned long long g = ((unsigned long long)C.seq << 32) | (unsigned long long)__float_as_uint(v);
  __hip_atomic_store(comm_slot(C, peer, C.rank, i), g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// One wave-level bounded spin per granule.  The bound is wall-clock (100 MHz constant counter): a peer that never
// arrives sets the error word instead of hanging the GPU.
__device__ inline float comm_poll(const unsigned long long* __restrict__ slot, unsigned seq, int* __restrict__ err) {
  unsigned long long g = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if ((unsigned)(g >> 32) != seq) {
    const unsigned long long t0 = wall_clock64();
    do {
      __builtin_amdgcn_s_sleep(2);
      g = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if ((unsigned)(g >> 32) == seq) break;
      if (wall_clock64() - t0 > ORL_COMM_TIMEOUT_TICKS) {
        *err = 1;
        return 0.f;
      }
    } while (true);
  }
  return __uint_as_float((unsigned)g);
}

// sum over ranks 0..world-1 in rank order; `mine` is this rank's own contribution
__device__ inline float comm_sum(const CommDev& C, int i, float mine) {
  float s = 0.f;
  for (int r = 0; r < C.world; ++r) s += (r == C.rank) ? mine : comm_poll(comm_slot(C, C.rank, r, i), C.seq, C.err);
  return s;
}

}  // namespace orl

struct orl_comm;
int orl_comm_next(orl_comm* c, orl::CommDev* out);  // advance the sequence number, return the device view
int orl_comm_current(orl_comm* c, orl::CommDev* out);  // device view of the collective opened last
int orl_comm_capacity(const orl_comm* c);
