// One element of the fused gradient clamp + Adam step (hint_optim.hip has the reference lines): shared by the stand-alone
// optimizer kernel and by the slab reduction with the optimizer folded in (hint_wgrad.hip), so that both take the same step
// bit for bit.
#pragma once
#include <hip/hip_runtime.h>

namespace hint {

// hint_wreduce_kernel with the optimizer folded in (hint_chain_backward_adam).  p, m, v: the model-wide arenas the blocks'
// parameter slices live in (a block's offset: its params pointer minus p); st: the device-side step factors
// {.., [3] lr / (1 - beta1^t), [4] 1 / sqrt(1 - beta2^t)} written by the step prologue.  p == nullptr: off.
struct AdamFuse {
    float* p; float* m; float* v; const float* st;
    float b1, b2, eps, wd, gscale, gclamp;
};

__device__ __forceinline__ void adam_update(float& p, float& m, float& v, float g, float lr_t, float b1, float b2,
                                            float inv_sqrt_bc2, float eps, float wd, float gscale, float gclamp) {
    float gj = g * gscale;
    gj = fminf(fmaxf(gj, -gclamp), gclamp);
    gj = gj + wd * p;
    m = b1 * m + (1.f - b1) * gj;
    v = b2 * v + (1.f - b2) * gj * gj;
    const float denom = sqrtf(v) * inv_sqrt_bc2 + eps;
    p = p - lr_t * (m / denom);
}

}  // namespace hint
