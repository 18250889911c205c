// Microbenchmark: how fast can every CU stream the SAME small buffer (the packed weights of a
// block) out of L2?  Variants: lock-step identical order vs per-workgroup rotated start,
// loads in flight per wave, waves per workgroup.   hipcc --offload-arch=gfx950 -O3 tools/l2bench.hip -o l2bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int DEPTH>
__global__ __launch_bounds__(512) void stream(const f32x4* __restrict__ buf, int n_tiles /*1KB each*/, int passes,
                                              int rot_stride, float* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    f32x4 acc = {0, 0, 0, 0};
    const int rot = (int)((blockIdx.x * (unsigned)rot_stride) % (unsigned)n_tiles);
    for (int p = 0; p < passes; ++p) {
        for (int t = wave * DEPTH; t < n_tiles; t += nw * DEPTH) {
            f32x4 v[DEPTH];
#pragma unroll
            for (int i = 0; i < DEPTH; ++i) {
                int tt = t + i + rot; if (tt >= n_tiles) tt -= n_tiles; if (tt >= n_tiles) tt -= n_tiles;
                v[i] = buf[(size_t)tt * 64 + lane];
            }
#pragma unroll
            for (int i = 0; i < DEPTH; ++i) acc += v[i];
        }
    }
    if (acc.x == 123.456f) out[0] = acc.y;   // keep the loads alive
}

int main() {
    const int n_tiles = 290;   // 290 KB, like one block's forward weights
    f32x4* buf; float* out;
    hipMalloc(&buf, (size_t)n_tiles * 1024); hipMalloc(&out, 16);
    hipMemset(buf, 0, (size_t)n_tiles * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, auto kern, int grid, int threads, int passes, int rot) {
        for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), 0, 0, buf, n_tiles, passes, rot, out);
        hipDeviceSynchronize();
        const int reps = 20;
        hipEventRecord(e0);
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), 0, 0, buf, n_tiles, passes, rot, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1e3 / reps;
        const double bytes_per_cu = (double)n_tiles * 1024 * passes;
        printf("%-34s grid %4d thr %4d passes %2d rot %3d : %8.2f us/launch  %7.1f GB/s per WG  %6.2f TB/s total\n", name, grid,
               threads, passes, rot, us, bytes_per_cu / us * 1e-3, bytes_per_cu * grid / us * 1e-6);
    };
    for (int passes : {1, 8}) {
        for (int rot : {0, 7, 37}) {
            run("depth4  8 waves", stream<4>, 256, 512, passes, rot);
            run("depth8  8 waves", stream<8>, 256, 512, passes, rot);
            run("depth16 8 waves", stream<16>, 256, 512, passes, rot);
        }
        run("depth8  4 waves", stream<8>, 256, 256, passes, 37);
        run("depth8  8 waves, 2 WG/CU", stream<8>, 512, 512, passes, 37);
        run("depth8  8 waves, 32 WGs only", stream<8>, 32, 512, passes, 37);
        run("depth8  8 waves, 8 WGs only", stream<8>, 8, 512, passes, 37);
    }
    return 0;
}
