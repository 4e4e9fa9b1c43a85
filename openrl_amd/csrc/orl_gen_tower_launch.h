// orl_gen_tower_launch.h - the __global__ wrappers of the cross-layer fused general towers and their launch helpers, shared by
// the two translation units that instantiate them (orl_gen_tower.hip: hidden_size 64, orl_gen_tower128.hip: hidden_size 128 -
// split in round 6 so that the library builds in the time of its slowest unit, not of their sum).
#pragma once
#include "orl_gen_tower.h"

namespace orl {

// backward: one 8-wave workgroup per CU (256 VGPRs per wave); forward: 8-wave workgroups at <= 128 VGPRs and ~75 KB of
// LDS, TWO per CU, so that one's barrier / LDS waits overlap the other's MFMAs
// Round 4: the backward launch ships as 4-wave workgroups, ONE per CU, 512 registers per wave (ORL_GT_BWD_WAVES = 4,
// ORL_GT_BWD_MINWAVES = 1: no scratch traffic) wherever its LDS fits ORL_GT_BWD_LDS_CAP_KB (gt_bwd_waves below has the
// measurements that set the cap); -DORL_GT_BWD_WAVES=8 -DORL_GT_BWD_MINWAVES=2 rebuilds round 3's.
#ifndef ORL_GT_BWD_MINWAVES
#define ORL_GT_BWD_MINWAVES 1
#endif
#ifndef ORL_GT_BWD_WAVES
#define ORL_GT_BWD_WAVES 4
#endif
#ifndef ORL_GT_BWD_LDS_CAP_KB
#define ORL_GT_BWD_LDS_CAP_KB 80
#endif
template <int H, int NL, int ND, int NW>
__global__ __launch_bounds__(NW * 64, ORL_GT_BWD_MINWAVES) void gt_bwd_kernel(GtArgs A) {
  gt_body<H, NL, ND, true, NW>(A);
}
template <int H, int NL, int ND>
__global__ __launch_bounds__(GT_WAVES * 64, 4) void gt_fwd_kernel(GtArgs A) {
  gt_body<H, NL, ND, false, GT_WAVES>(A);
}

template <int H, int NL, int ND>
static inline void gt_launch_fwd(const GtArgs& A, int grid, size_t lds, hipStream_t s) {
  (void)hipFuncSetAttribute((const void*)gt_fwd_kernel<H, NL, ND>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((gt_fwd_kernel<H, NL, ND>), dim3(grid), dim3(GT_WAVES * 64), lds, s, A);
}
template <int H, int NL, int ND, int NW>
static inline void gt_launch_bwd(const GtArgs& A, int grid, size_t lds, hipStream_t s) {
  (void)hipFuncSetAttribute((const void*)gt_bwd_kernel<H, NL, ND, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((gt_bwd_kernel<H, NL, ND, NW>), dim3(grid), dim3(NW * 64), lds, s, A);
}

#if ORL_GT_BWD_WAVES == 4
#define ORL_GT_NW4 4
#else
#define ORL_GT_NW4 8  // the 4-wave build is not instantiated
#endif

// the H == 128 instantiations live in orl_gen_tower128.hip; returns 0 when it launched, ORL_E_UNSUPPORTED otherwise
int gt_launch_h128(const GtArgs& A, int bwd_waves, int grid, size_t lds, hipStream_t s);

#define ORL_GT_CASE3(h, nl, nd)                                                    \
  if (H == h && NL == nl && ND == nd) {                                            \
    if (bwd_waves == 0) gt_launch_fwd<h, nl, nd>(A, grid, lds, s);                 \
    else if (bwd_waves == 4) gt_launch_bwd<h, nl, nd, ORL_GT_NW4>(A, grid, lds, s); \
    else gt_launch_bwd<h, nl, nd, 8>(A, grid, lds, s);                             \
    return 0;                                                                      \
  }
#define ORL_GT_CASE(h, nl) ORL_GT_CASE3(h, nl, 1) ORL_GT_CASE3(h, nl, 4)

}  // namespace orl
