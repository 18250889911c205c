// Weight packing (gfx950): flat torch-layout parameters -> MFMA fragment order (hint_dev.h), plus the
// small streaming helpers of the C ABI.  Pure data movement: HBM / L2 bound, 16 bytes per lane.
#include "hint_device.hpp"

using namespace hint;

__device__ __forceinline__ void pack_body(int bid, const PackSeg* __restrict__ segs,
                                          const int2* __restrict__ ptiles, int n_tiles,
                                          const int32_t* __restrict__ bmap, int n_bias, long bias_off,
                                          const float* __restrict__ P, float* __restrict__ packed) {
    if (bid >= n_tiles) {
        // trailing workgroups: biases b1, b2, b3 per unit, zero padded to 16
        const int i = (bid - n_tiles) * 256 + (int)threadIdx.x;
        if (i < n_bias) { const int off = bmap[i]; packed[bias_off + i] = off >= 0 ? P[off] : 0.f; }
        return;
    }
    const int2 pt = ptiles[bid];
    const PackSeg sg = segs[pt.x];
    const int nt = pt.y;
    if (sg.kmap == 2) {
        // vector layout (thin layers): dst[(nt*KV + k)*16 + f] = Wlog[nt*16 + f][k] for k < K, zero up to Kp = NB;
        // vector Kp: the bias
        const int Kp = sg.NB, KV = Kp + (sg.src2 >= 0 ? 1 : 0);
        for (int idx = (int)threadIdx.x; idx < 16 * KV; idx += 256) {
            const int k = idx >> 4, f = idx & 15, n = nt * 16 + f;
            float val = 0.f;
            if (n < sg.N) {
                if (k < sg.K) val = sg.trans ? P[sg.src + (int64_t)k * sg.ld + n] : P[sg.src + (int64_t)n * sg.ld + k];
                else if (k == Kp) val = P[sg.src2 + n];
            }
            packed[sg.dst + (int64_t)nt * KV * 16 + idx] = val;
        }
        return;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = nt * 16 + (lane & 15), kq = lane >> 4;
    for (int kb = wave; kb < sg.NB; kb += 4) {
        f32x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float val = 0.f;
            const int k = kb * 16 + 4 * kq + i;
            if (n < sg.N && k < sg.K) val = sg.trans ? P[sg.src + (int64_t)k * sg.ld + n] : P[sg.src + (int64_t)n * sg.ld + k];
            v[i] = val;
        }
        ((f32x4*)(packed + sg.dst + ((int64_t)nt * sg.NB + kb) * 256))[lane] = v;
    }
}

__global__ __launch_bounds__(256) void hint_pack_kernel(const PackSeg* __restrict__ segs,
                                                        const int2* __restrict__ ptiles, int n_tiles,
                                                        const int32_t* __restrict__ bmap, int n_bias,
                                                        long bias_off, const float* __restrict__ P,
                                                        float* __restrict__ packed) {
    pack_body((int)blockIdx.x, segs, ptiles, n_tiles, bmap, n_bias, bias_off, P, packed);
}

// all blocks of a flow in ONE launch (the trainer re-packs every block after each optimizer step)
// The launch doubles as the prologue of a training step: one extra workgroup clears the loss sums
// of the step before and advances the noise counter (hint_pack_group_run_ex).
__global__ __launch_bounds__(256) void hint_pack_many_kernel(const PackItem* __restrict__ items, int n_items,
                                                             int pack_grid, float* __restrict__ zero_buf, int zero_floats,
                                                             unsigned long long* __restrict__ rng_state,
                                                             float* __restrict__ opt_state) {
    if ((int)blockIdx.x >= pack_grid) {
        for (int i = threadIdx.x; i < zero_floats; i += 256) zero_buf[i] = 0.f;
        if (rng_state != nullptr && threadIdx.x == 0) {
            const unsigned long long step = rng_state[1] + 1ull;
            rng_state[1] = step;
            if (opt_state != nullptr) {
                // Adam's bias corrections of this step (torch.optim.Adam evaluates them in double):
                // opt_state = {lr, beta1, beta2, -> lr/(1-beta1^t), -> 1/sqrt(1-beta2^t)}
                const double b1 = opt_state[1], b2 = opt_state[2], t = (double)step;
                opt_state[3] = (float)((double)opt_state[0] / (1.0 - pow(b1, t)));
                opt_state[4] = (float)(1.0 / sqrt(1.0 - pow(b2, t)));
            }
        }
        return;
    }
    // which item: one parallel look at every item's first workgroup instead of a chain of dependent loads
    int it = 0;
    for (int i0 = 0; i0 < n_items; i0 += 64) {
        const int i = i0 + (int)(threadIdx.x & 63);
        const bool ge = i < n_items && (int)blockIdx.x >= items[i].grid_begin;
        it += __builtin_popcountll(__ballot(ge));
    }
    it = __builtin_amdgcn_readfirstlane(it - 1);
    const PackItem q = items[it];
    pack_body((int)blockIdx.x - q.grid_begin, q.segs, (const int2*)q.ptiles, q.n_tiles, q.bmap, q.n_bias, q.bias_off, q.params,
              q.packed);
}

__global__ __launch_bounds__(256) void hint_zero_kernel(float* __restrict__ p, long n4, long n) {
    const long stride = (long)gridDim.x * blockDim.x;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) ((f32x4*)p)[i] = z;
    const long t = n4 * 4 + (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) p[t] = 0.f;
}

// ---- element-wise helpers of hint_block_inverse_backward (hint_invgrad.cpp): B x d lane tiles, `lower` marks the lanes a level
//      transforms.  op 0: out = lower ? 1 : 0;  op 1: out = lower ? g / e : 0;  op 2: g = lower ? a : g - b ----
__global__ __launch_bounds__(256) void hint_inv_lane_kernel(int op, float* out, const float* g,             // (op 2 runs in place: out == g)
                                                            const float* __restrict__ a, const float* __restrict__ b,
                                                            const uint8_t* __restrict__ lower, long n, int d) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const bool lo = lower[i % d] != 0;
        float v;
        if (op == 0) v = lo ? 1.f : 0.f;
        else if (op == 1) v = lo ? g[i] / a[i] : 0.f;
        else v = lo ? a[i] : g[i] - b[i];
        out[i] = v;
    }
}

// dst = (keep ? dst : 0) - src
__global__ __launch_bounds__(256) void hint_inv_minus_kernel(float* __restrict__ dst, const float* __restrict__ src, long n, int keep) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = (keep ? dst[i] : 0.f) - src[i];
}

// y = x P (row tile by d x d matrix, d <= 128: the node permutations in front of a block; a few MFLOP)
__global__ __launch_bounds__(256) void hint_inv_rowmat_kernel(const float* __restrict__ x, const float* __restrict__ P,
                                                              float* __restrict__ y, long n, int d) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const long row = i / d;
        const int j = (int)(i - row * d);
        float acc = 0.f;
        for (int k = 0; k < d; ++k) acc = fmaf(x[row * d + k], P[(long)k * d + j], acc);
        y[i] = acc;
    }
}

namespace hint {

static inline unsigned small_grid(long n, int num_cu) {
    long blocks = (n + 255) / 256;
    return (unsigned)(blocks < 1 ? 1 : (blocks > (long)num_cu * 8 ? (long)num_cu * 8 : blocks));
}

hipError_t launch_inv_lane(int op, float* out, const float* g, const float* a, const float* b, const uint8_t* lower, long n, int d,
                           int num_cu, hipStream_t stream) {
    hipLaunchKernelGGL(hint_inv_lane_kernel, dim3(small_grid(n, num_cu)), dim3(256), 0, stream, op, out, g, a, b, lower, n, d);
    return hipGetLastError();
}

hipError_t launch_inv_minus(float* dst, const float* src, long n, int keep, int num_cu, hipStream_t stream) {
    if (n > 0) hipLaunchKernelGGL(hint_inv_minus_kernel, dim3(small_grid(n, num_cu)), dim3(256), 0, stream, dst, src, n, keep);
    return hipGetLastError();
}

hipError_t launch_inv_rowmat(const float* x, const float* P, float* y, long n, int d, int num_cu, hipStream_t stream) {
    hipLaunchKernelGGL(hint_inv_rowmat_kernel, dim3(small_grid(n, num_cu)), dim3(256), 0, stream, x, P, y, n, d);
    return hipGetLastError();
}

hipError_t launch_pack(const PackSeg* segs, const int2* ptiles, int n_tiles, const int32_t* bmap, int n_bias,
                       long bias_off, const float* params, float* packed, hipStream_t stream) {
    const int grid = n_tiles + (n_bias + 255) / 256;
    if (grid > 0)
        hipLaunchKernelGGL(hint_pack_kernel, dim3(grid), dim3(256), 0, stream, segs, ptiles, n_tiles, bmap, n_bias,
                           bias_off, params, packed);
    return hipGetLastError();
}

hipError_t launch_pack_many(const PackItem* items, int n_items, int grid, float* zero_buf, int zero_floats,
                            unsigned long long* rng_state, float* opt_state, hipStream_t stream) {
    const int extra = (zero_floats > 0 || rng_state != nullptr) ? 1 : 0;
    if (grid + extra > 0)
        hipLaunchKernelGGL(hint_pack_many_kernel, dim3(grid + extra), dim3(256), 0, stream, items, n_items, grid, zero_buf,
                           zero_floats, rng_state, opt_state);
    return hipGetLastError();
}

hipError_t launch_zero(float* p, long n, int num_cu, hipStream_t stream) {
    const long n4 = n / 4;
    long blocks = (n4 + 255) / 256;
    blocks = blocks < 1 ? 1 : (blocks > (long)num_cu * 4 ? (long)num_cu * 4 : blocks);
    hipLaunchKernelGGL(hint_zero_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, p, n4, n);
    return hipGetLastError();
}

}  // namespace hint
