// Micro-benchmark of ONE GEMM stage of the block kernels (out[16][N] = relu(A[16][K] W^T + b), A in LDS,
// W streamed from L2 in MFMA fragment order) to compare inner-loop structures outside the full kernel.
//   hipcc -O3 --offload-arch=gfx950 tools/stage_bench.hip -o gpurun_out/stage_bench && gpurun_out/stage_bench
// Shapes: NTILES output tiles of 16 columns, NB k-blocks of 16 (L2 of the power_hint_8 root: 18 x 9).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int NW = 8, NTHREADS = 64 * NW, ROWS = 16;
#define LDS_AS __attribute__((address_space(3)))

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// ---- variant 1: NT output tiles per wavefront share every A fragment; k-loop software pipelined 2 ahead ----
template <int NT>
__device__ __forceinline__ void job_nt(const float* arow, int NB, const f32x4* wp, int tstride, const float* bias,
                                       float* orow, int ldo, int ocol, int lane) {
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 b0[NT], b1[NT], b2[NT], a0, a1;
#pragma unroll
    for (int t = 0; t < NT; ++t) { b0[t] = wp[(size_t)(t * tstride + 0) * 64]; b1[t] = wp[(size_t)(t * tstride + 1) * 64]; }
    a0 = *(const f32x4*)(arow);
    int kb = 0;
    // NB >= 2 assumed; loads past the end are clamped to the last block
#define STEP(AC, AN, BC, BN2)                                                                        \
    {                                                                                                \
        const int kn = kb + 2 < NB ? kb + 2 : NB - 1;                                                \
        _Pragma("unroll") for (int t = 0; t < NT; ++t) BN2[t] = wp[(size_t)(t * tstride + kn) * 64]; \
        const int ka = kb + 1 < NB ? kb + 1 : NB - 1;                                                \
        AN = *(const f32x4*)(arow + 16 * ka);                                                        \
        _Pragma("unroll") for (int t = 0; t < NT; ++t) acc[t] = mfma4(AC.x, BC[t].x, acc[t]);        \
        _Pragma("unroll") for (int t = 0; t < NT; ++t) acc[t] = mfma4(AC.y, BC[t].y, acc[t]);        \
        _Pragma("unroll") for (int t = 0; t < NT; ++t) acc[t] = mfma4(AC.z, BC[t].z, acc[t]);        \
        _Pragma("unroll") for (int t = 0; t < NT; ++t) acc[t] = mfma4(AC.w, BC[t].w, acc[t]);        \
        ++kb;                                                                                        \
    }
    while (true) {
        STEP(a0, a1, b0, b2) if (kb >= NB) break;
        STEP(a1, a0, b1, b0) if (kb >= NB) break;
        STEP(a0, a1, b2, b1) if (kb >= NB) break;
        STEP(a1, a0, b0, b2) if (kb >= NB) break;
        STEP(a0, a1, b1, b0) if (kb >= NB) break;
        STEP(a1, a0, b2, b1) if (kb >= NB) break;
    }
#undef STEP
    const int nl = lane & 15;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const float bi = bias[ocol + 16 * t + nl];
        float* o = orow + ocol + 16 * t + nl;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i * ldo] = fmaxf(acc[t][i] + bi, 0.f);
    }
}

// ---- variant 4: variant 1 with the first two k-blocks' B tiles handed in (loaded one stage ahead) ----
template <int NT>
__device__ __forceinline__ void job_pre(const float* arow, int NB, const f32x4* wp, int tstride, const float* bias,
                                        float* orow, int ldo, int ocol, int lane, f32x4 (&b0)[3], f32x4 (&b1)[3],
                                        const f32x4* wp_next) {
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 b2[NT], a0, a1;
    a0 = *(const f32x4*)(arow);
    int kb = 0;
#define STEP(AC, AN, BC, BN2)                                                                        \
    {                                                                                                \
        const int kn = kb + 2 < NB ? kb + 2 : NB - 1;                                                \
        _Pragma("unroll") for (int t = 0; t < NT; ++t) BN2[t] = wp[(size_t)(t * tstride + kn) * 64]; \
        const int ka = kb + 1 < NB ? kb + 1 : NB - 1;                                                \
        AN = *(const f32x4*)(arow + 16 * ka);                                                        \
        _Pragma("unroll") for (int t = 0; t < NT; ++t) acc[t] = mfma4(AC.x, BC[t].x, acc[t]);        \
        _Pragma("unroll") for (int t = 0; t < NT; ++t) acc[t] = mfma4(AC.y, BC[t].y, acc[t]);        \
        _Pragma("unroll") for (int t = 0; t < NT; ++t) acc[t] = mfma4(AC.z, BC[t].z, acc[t]);        \
        _Pragma("unroll") for (int t = 0; t < NT; ++t) acc[t] = mfma4(AC.w, BC[t].w, acc[t]);        \
        ++kb;                                                                                        \
    }
    while (true) {
        STEP(a0, a1, b0, b2) if (kb >= NB) break;
        STEP(a1, a0, b1, b0) if (kb >= NB) break;
        STEP(a0, a1, b2, b1) if (kb >= NB) break;
        STEP(a1, a0, b0, b2) if (kb >= NB) break;
        STEP(a0, a1, b1, b0) if (kb >= NB) break;
        STEP(a1, a0, b2, b1) if (kb >= NB) break;
    }
#undef STEP
    // next stage's first two k-blocks, before the epilogue
#pragma unroll
    for (int t = 0; t < NT; ++t) { b0[t] = wp_next[(size_t)(t * tstride + 0) * 64]; b1[t] = wp_next[(size_t)(t * tstride + 1) * 64]; }
    const int nl = lane & 15;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const float bi = bias[ocol + 16 * t + nl];
        float* o = orow + ocol + 16 * t + nl;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i * ldo] = fmaxf(acc[t][i] + bi, 0.f);
    }
}

// ---- variant 2: like variant 1 but the whole stage's k-loop is shared by up to 3 tiles and the B loads run
//      DEPTH k-blocks ahead through a register ring indexed at compile time (NB must be a template constant) ----
template <int NT, int NB, int DEPTH>
__device__ __forceinline__ void job_static(const float* arow, const f32x4* wp, const float* bias, float* orow, int ldo,
                                           int ocol, int lane) {
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 b[DEPTH + 1][NT], a[2];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
        for (int t = 0; t < NT; ++t) b[d][t] = wp[(size_t)(t * NB + (d < NB ? d : NB - 1)) * 64];
    a[0] = *(const f32x4*)(arow);
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) {
        if (kb + DEPTH < NB) {
#pragma unroll
            for (int t = 0; t < NT; ++t) b[(kb + DEPTH) % (DEPTH + 1)][t] = wp[(size_t)(t * NB + kb + DEPTH) * 64];
        }
        if (kb + 1 < NB) a[(kb + 1) & 1] = *(const f32x4*)(arow + 16 * (kb + 1));
        const f32x4 ac = a[kb & 1];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = mfma4(ac.x, b[kb % (DEPTH + 1)][t].x, acc[t]);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = mfma4(ac.y, b[kb % (DEPTH + 1)][t].y, acc[t]);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = mfma4(ac.z, b[kb % (DEPTH + 1)][t].z, acc[t]);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = mfma4(ac.w, b[kb % (DEPTH + 1)][t].w, acc[t]);
    }
    const int nl = lane & 15;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const float bi = bias[ocol + 16 * t + nl];
        float* o = orow + ocol + 16 * t + nl;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i * ldo] = fmaxf(acc[t][i] + bi, 0.f);
    }
}

template <int VARIANT, int NTILES, int NB>
__global__ __launch_bounds__(NTHREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void stage_kernel(
    const float* __restrict__ packed, const float* __restrict__ x, float* __restrict__ out, int iters,
    unsigned long long* __restrict__ cyc) {
    constexpr int K = NB * 16, N = NTILES * 16;
    constexpr int lda = K + 4, ldo = N + 4;
    __shared__ __attribute__((aligned(16))) float A[ROWS * lda];
    __shared__ __attribute__((aligned(16))) float O[ROWS * ldo];
    __shared__ float bias[N];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < ROWS * K; i += NTHREADS) A[(i / K) * lda + i % K] = x[(size_t)blockIdx.x * ROWS * K + i];
    for (int i = tid; i < N; i += NTHREADS) bias[i] = 0.01f * i;
    __syncthreads();
    const float* arow = A + (lane & 15) * lda + 4 * (lane >> 4);
    float* orow = O + 4 * (lane >> 4) * ldo;
    const f32x4* wp = (const f32x4*)packed + lane;
    unsigned long long t0 = 0, t1 = 0;
    f32x4 pb0[3], pb1[3];
    if (VARIANT == 4) {
        constexpr int base = NTILES / NW, extra = NTILES % NW;
        const int t0i = wave * base + (wave < extra ? wave : extra);
#pragma unroll
        for (int t = 0; t < 3; ++t) { pb0[t] = wp[(size_t)(t0i * NB + t * NB) * 64]; pb1[t] = wp[(size_t)(t0i * NB + t * NB + (NB > 1)) * 64]; }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
        // every iteration reads a different copy of the weights (as consecutive stages of the block do)
        const f32x4* w = wp + (size_t)(it & 7) * NTILES * NB * 64;
        if (VARIANT == 4) {
            constexpr int base = NTILES / NW, extra = NTILES % NW;
            const int nt = base + (wave < extra ? 1 : 0);
            const int t0i = wave * base + (wave < extra ? wave : extra);
            const f32x4* wn = wp + (size_t)((it + 1) & 7) * NTILES * NB * 64 + (size_t)t0i * NB * 64;
            if (nt == 3) job_pre<3>(arow, NB, w + (size_t)t0i * NB * 64, NB, bias, orow, ldo, 16 * t0i, lane, pb0, pb1, wn);
            else if (nt == 2) job_pre<2>(arow, NB, w + (size_t)t0i * NB * 64, NB, bias, orow, ldo, 16 * t0i, lane, pb0, pb1, wn);
            else if (nt == 1) job_pre<1>(arow, NB, w + (size_t)t0i * NB * 64, NB, bias, orow, ldo, 16 * t0i, lane, pb0, pb1, wn);
        } else if (VARIANT == 1) {
            // tiles dealt round-robin in runs of 3/2: wave w owns tiles [t0, t0+nt)
            constexpr int base = NTILES / NW, extra = NTILES % NW;
            const int nt = base + (wave < extra ? 1 : 0);
            const int t0i = wave * base + (wave < extra ? wave : extra);
            if (nt == 3) job_nt<3>(arow, NB, w + (size_t)t0i * NB * 64, NB, bias, orow, ldo, 16 * t0i, lane);
            else if (nt == 2) job_nt<2>(arow, NB, w + (size_t)t0i * NB * 64, NB, bias, orow, ldo, 16 * t0i, lane);
            else if (nt == 1) job_nt<1>(arow, NB, w + (size_t)t0i * NB * 64, NB, bias, orow, ldo, 16 * t0i, lane);
        } else if (VARIANT == 2 || VARIANT == 3) {
            constexpr int DEPTH = VARIANT == 2 ? 2 : 3;
            constexpr int base = NTILES / NW, extra = NTILES % NW;
            const int nt = base + (wave < extra ? 1 : 0);
            const int t0i = wave * base + (wave < extra ? wave : extra);
            if (nt == 3) job_static<3, NB, DEPTH>(arow, w + (size_t)t0i * NB * 64, bias, orow, ldo, 16 * t0i, lane);
            else if (nt == 2) job_static<2, NB, DEPTH>(arow, w + (size_t)t0i * NB * 64, bias, orow, ldo, 16 * t0i, lane);
            else if (nt == 1) job_static<1, NB, DEPTH>(arow, w + (size_t)t0i * NB * 64, bias, orow, ldo, 16 * t0i, lane);
        } else {
            // variant 0: one tile per job, jobs dealt round-robin (what the chunk interpreter does, minus the interpreter)
            for (int t = wave; t < NTILES; t += NW) job_nt<1>(arow, NB, w + (size_t)t * NB * 64, NB, bias, orow, ldo, 16 * t, lane);
        }
        lds_barrier();
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (blockIdx.x == 0 && lane == 0) cyc[wave] = t1 - t0;
    for (int i = tid; i < ROWS * N; i += NTHREADS) out[(size_t)blockIdx.x * ROWS * N + i] = O[(i / N) * ldo + i % N];
}

template <int VARIANT, int NTILES, int NB>
static void run(const char* name, const float* packed, const float* x, float* out, unsigned long long* cyc) {
    const int iters = 400, grid = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((stage_kernel<VARIANT, NTILES, NB>), dim3(grid), dim3(NTHREADS), 0, 0, packed, x, out, iters, cyc);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((stage_kernel<VARIANT, NTILES, NB>), dim3(grid), dim3(NTHREADS), 0, 0, packed, x, out, iters, cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[NW];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    const double us = ms * 1e3 / 5 / iters;
    // MFMA bound: the busiest SIMD (waves w and w+4) x 4 MFMAs x 32 cycles per k-block
    int per_wave[NW];
    for (int w = 0; w < NW; ++w) per_wave[w] = NTILES / NW + (w < NTILES % NW ? 1 : 0);
    int worst = 0;
    for (int s = 0; s < 4; ++s) worst = std::max(worst, per_wave[s] + per_wave[s + 4]);
    const double bound = worst * NB * 4 * 32.0;
    printf("%-34s tiles %2d x kb %2d: %7.3f us/stage  %7.0f cyc/stage (wave0)  MFMA bound %6.0f cyc  -> %4.1f %%\n", name, NTILES, NB,
           us, (double)h[0] / iters, bound, 100.0 * bound / ((double)h[0] / iters));
}

int main() {
    const size_t wfloats = (size_t)8 * 24 * 24 * 256 + 4096;
    const size_t xfloats = (size_t)256 * ROWS * 24 * 16;
    float *packed, *x, *out;
    unsigned long long* cyc;
    hipMalloc(&packed, wfloats * 4); hipMalloc(&x, xfloats * 4); hipMalloc(&out, xfloats * 4); hipMalloc(&cyc, 64);
    std::vector<float> h(wfloats);
    for (size_t i = 0; i < wfloats; ++i) h[i] = 0.01f * (float)((i * 2654435761u) % 97) - 0.4f;
    hipMemcpy(packed, h.data(), wfloats * 4, hipMemcpyHostToDevice);
    hipMemcpy(x, h.data(), xfloats * 4, hipMemcpyHostToDevice);
    run<0, 18, 9>("v0 one tile/job", packed, x, out, cyc);
    run<1, 18, 9>("v1 shared A, runtime NB", packed, x, out, cyc);
    run<4, 18, 9>("v4 = v1 + next stage prefetched", packed, x, out, cyc);
    run<2, 18, 9>("v2 shared A, static NB, depth 2", packed, x, out, cyc);
    run<3, 18, 9>("v3 shared A, static NB, depth 3", packed, x, out, cyc);
    run<0, 20, 5>("v0 one tile/job", packed, x, out, cyc);
    run<1, 20, 5>("v1 shared A, runtime NB", packed, x, out, cyc);
    run<4, 20, 5>("v4 = v1 + next stage prefetched", packed, x, out, cyc);
    run<2, 20, 5>("v2 shared A, static NB, depth 2", packed, x, out, cyc);
    run<0, 16, 8>("v0 one tile/job", packed, x, out, cyc);
    run<1, 16, 8>("v1 shared A, runtime NB", packed, x, out, cyc);
    run<2, 16, 8>("v2 shared A, static NB, depth 2", packed, x, out, cyc);
    run<0, 18, 1>("v0 one tile/job (L1-like)", packed, x, out, cyc);
    run<2, 18, 1>("v2 (L1-like)", packed, x, out, cyc);
    run<4, 18, 1>("v4 (L1-like)", packed, x, out, cyc);
    run<4, 24, 9>("v4 24x9", packed, x, out, cyc);
    run<4, 16, 9>("v4 16x9", packed, x, out, cyc);
    return 0;
}
