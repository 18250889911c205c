// Fused gradient clamp + Adam over the flat parameter arena (gfx950).
//
// Reference step being replaced (read-only):
//   /root/reference/train_unconditional.py:140-141  p.grad.data.clamp_(-5, 5) for every p
//   /root/reference/train_unconditional.py:144,174-176  torch.optim.Adam(lr, betas, eps=1e-4,
//                                                        weight_decay=l2_weight_reg).step()
// torch.optim.Adam (non-AMSGrad, L2 weight decay folded into the gradient):
//   g = clamp(g*gscale) + wd*p ; m = b1*m + (1-b1)*g ; v = b2*v + (1-b2)*g*g
//   p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
// `gscale` carries the 1/world_size of the data-parallel all-reduce (sum -> mean), applied
// BEFORE the clamp because the reference clamps the final, averaged gradient.
// With zero_grads the kernel also clears the gradient arena it has just consumed, so the next
// step's backward kernels can accumulate into it without a separate memset.
// Pure HBM streaming: 16 B/lane loads and stores, 4 arrays read + 3 (4) written = 28-32 B/param.
#include "hint_adam.hpp"

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void hint_adam_kernel(float* __restrict__ p, float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v,
                                                        long n4, long n, float lr_t, float b1, float b2,
                                                        float inv_sqrt_bc2, float eps, float wd,
                                                        float gscale, float gclamp, int zero_grads,
                                                        const float* __restrict__ dev_state) {
    // dev_state (hint_adam_step_dev): step-dependent factors computed on the device by the step
    // prologue (hint_pack_group_run_ex), so that the launch can sit inside a captured graph
    if (dev_state != nullptr) { lr_t = dev_state[3]; inv_sqrt_bc2 = dev_state[4]; }
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        f32x4 pp = ((f32x4*)p)[i], gg = ((f32x4*)g)[i], mm = ((f32x4*)m)[i], vv = ((f32x4*)v)[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float pj = pp[j], mj = mm[j], vj = vv[j];
            hint::adam_update(pj, mj, vj, gg[j], lr_t, b1, b2, inv_sqrt_bc2, eps, wd, gscale, gclamp);
            pp[j] = pj; mm[j] = mj; vv[j] = vj;
        }
        ((f32x4*)p)[i] = pp; ((f32x4*)m)[i] = mm; ((f32x4*)v)[i] = vv;
        if (zero_grads) ((f32x4*)g)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // ragged tail (n not a multiple of 4)
    const long tail0 = n4 * 4;
    const long t = tail0 + (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) {
        float pj = p[t], mj = m[t], vj = v[t];
        hint::adam_update(pj, mj, vj, g[t], lr_t, b1, b2, inv_sqrt_bc2, eps, wd, gscale, gclamp);
        p[t] = pj; m[t] = mj; v[t] = vj;
        if (zero_grads) g[t] = 0.f;
    }
}

namespace hint {
hipError_t launch_adam(float* p, float* g, float* m, float* v, long n, float lr_t, float b1, float b2,
                       float inv_sqrt_bc2, float eps, float wd, float gscale, float gclamp, int zero_grads, int num_cu,
                       const float* dev_state, hipStream_t stream) {
    const long n4 = n / 4;
    long blocks = (n4 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > (long)num_cu * 8) blocks = (long)num_cu * 8;
    hipLaunchKernelGGL(hint_adam_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, p, g, m, v, n4, n, lr_t,
                       b1, b2, inv_sqrt_bc2, eps, wd, gscale, gclamp, zero_grads, dev_state);
    return hipGetLastError();
}
}  // namespace hint
