// orl_gen_tower128.hip - the hidden_size 128 instantiations of the cross-layer fused general towers (orl_gen_tower.h); the
// hidden_size 64 ones and the C ABI are in orl_gen_tower.hip.  A translation unit of its own since round 6: the general towers
// were the slowest unit of the build (99 s) and are compiled side by side now.
#include "orl_gen_tower_launch.h"

namespace orl {

int gt_launch_h128(const GtArgs& A, int bwd_waves, int grid, size_t lds, hipStream_t s) {
  const int H = A.d.H, NL = A.d.n_layers - 1, ND = A.d.D <= 16 ? 1 : 4;
  ORL_GT_CASE(128, 1)
  ORL_GT_CASE(128, 2)
  ORL_GT_CASE(128, 3)
  return fail(ORL_E_UNSUPPORTED, "orl_gt: no kernel for hidden_size %d with %d layers", H, NL + 1);
}

}  // namespace orl
