// orl_gen_act.h - activation functions of the general tower path (activation_id of the reference's MLPLayer,
// openrl/modules/networks/utils/mlp.py:14-17: [Tanh, ReLU, LeakyReLU, ELU]); shared by orl_gen.hip and orl_gen_fused.hip.
#pragma once
#include "orl_common.h"

namespace orl {

__device__ inline float act_fwd(float z, int act) {
  switch (act) {
    case ORL_ACT_TANH: return tanhf(z);
    case ORL_ACT_RELU: return fmaxf(z, 0.f);
    case ORL_ACT_LEAKY_RELU: return z > 0.f ? z : 0.01f * z;
    case ORL_ACT_ELU: return z > 0.f ? z : expm1f(z);
    default: return z;
  }
}
// derivative expressed through the OUTPUT a = act(z) (what the forward stores)
__device__ inline float act_bwd(float a, int act) {
  switch (act) {
    case ORL_ACT_TANH: return 1.f - a * a;
    case ORL_ACT_RELU: return a > 0.f ? 1.f : 0.f;
    case ORL_ACT_LEAKY_RELU: return a > 0.f ? 1.f : 0.01f;
    case ORL_ACT_ELU: return a > 0.f ? 1.f : a + 1.f;
    default: return 1.f;
  }
}

}  // namespace orl
