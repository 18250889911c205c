// Microbenchmark of the wave-local kernels' k-loop step (hint_wl.hpp: WL_STEP) in isolation: 256 workgroups x 8
// wavefronts, every wavefront runs `steps` steps of [prefetch 3 weight tiles][thin vectors from LDS -> B operand]
// [4 x ntt MFMAs], wavefronts 0,1 with 3 tiles, the others with 2 (the root level of POWER d = 6).  Prints shader
// cycles per step (median over workgroups of the slowest wavefront) for: MFMAs only / + weight stream / + thin layer.
//   hipcc -O3 --offload-arch=gfx950 -fno-slp-vectorize tools/wl_step_bench.hip -o tools/wl_step_bench && tools/wl_step_bench
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define LDS_AS __attribute__((address_space(3)))
#define GLOBAL_AS __attribute__((address_space(1)))

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 fma4(const f32x4 w, float s, const f32x4 a) {
    return f32x4{fmaf(w.x, s, a.x), fmaf(w.y, s, a.y), fmaf(w.z, s, a.z), fmaf(w.w, s, a.w)};
}

template <int MODE, int DIST>     // MODE bit 0: weight stream, bit 1: thin layer, bit 2: every workgroup starts its stream elsewhere, bit 3: exact loads (no third tile for 2-tile rows);  DIST: ring depth - 1
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void step_kernel(
    const float* __restrict__ wts, long wfloats, const float* __restrict__ thin, int steps, float* out,
    unsigned long long* cyc) {
    __shared__ __attribute__((aligned(16))) float par[8192];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 15, kq = lane >> 4;
    for (int i = tid; i < 8192; i += 512) par[i] = thin[i];
    __syncthreads();
    const int ntt = wave < 2 ? 3 : 2;
    float vin[4];
    for (int k = 0; k < 4; ++k) vin[k] = par[m * 7 + k];
    const LDS_AS f32x4* tq = (const LDS_AS f32x4*)par + kq;
    const GLOBAL_AS char* p = (const GLOBAL_AS char*)wts + lane * 16;
    const int tmask = (int)(wfloats / 256) - 1;           // (a power of two of tiles)
    const int base = wave * 40 + ((MODE & 4) ? (int)(blockIdx.x >> 3) * 67 : 0);      // tile index of this wavefront's stream
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = acc0, acc2 = acc0;
    constexpr int RING = DIST + 1;
    f32x4 ring[RING][3];
    auto load = [&](f32x4 (&dst)[3], int t) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            if ((MODE & 8) && j >= ntt) break;
            const int jj = j < ntt ? j : ntt - 1;
            const bool stream = (MODE & 1) && !((MODE & 32) && wave >= 4);
            const int idx = stream ? ((base + t + jj * 9) & tmask) : jj;
            if (MODE & 16) dst[j] = __builtin_nontemporal_load((const GLOBAL_AS f32x4*)(p + (size_t)idx * 1024));
            else dst[j] = *(const GLOBAL_AS f32x4*)(p + (size_t)idx * 1024);
        }
    };
#pragma unroll
    for (int s = 0; s < DIST; ++s) load(ring[s], s);
    f32x4 qv[5];
    auto q_load = [&](int kb) {
        const LDS_AS f32x4* q = tq + 20 * (kb % 9);
#pragma unroll
        for (int k = 0; k < 5; ++k) qv[k] = q[4 * k];
    };
    auto q_frag = [&]() -> f32x4 {
        const f32x4 r = fma4(qv[3], vin[3], fma4(qv[2], vin[2], fma4(qv[1], vin[1], fma4(qv[0], vin[0], qv[4]))));
        return f32x4{fmaxf(r.x, 0.f), fmaxf(r.y, 0.f), fmaxf(r.z, 0.f), fmaxf(r.w, 0.f)};
    };
    f32x4 b4n = {vin[0], vin[1], vin[2], vin[3]};
    if (MODE & 2) { q_load(0); b4n = q_frag(); q_load(1); }
#define PIN() asm volatile("" : "+v"(acc0), "+v"(acc1), "+v"(acc2) : : "memory");
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if ((MODE & 64) && wave >= 4) __builtin_amdgcn_s_sleep(5);          // stagger the SIMD partners by about half a step
    for (int k0 = 0; k0 < steps; k0 += RING) {
#pragma unroll
        for (int s = 0; s < RING; ++s) {
            load(ring[(s + DIST) % RING], k0 + s + DIST);
            PIN()
            f32x4 b4 = b4n;
            const bool late = (MODE & 128) && wave >= 4;                // these wavefronts prepare the next operand BEHIND their MFMAs
            if ((MODE & 2) && !late) { b4n = q_frag(); q_load(k0 + s + 2); }
            if (ntt >= 3) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc0 = mfma4(ring[s][0][i], b4[i], acc0); acc1 = mfma4(ring[s][1][i], b4[i], acc1);
                    acc2 = mfma4(ring[s][2][i], b4[i], acc2);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) { acc0 = mfma4(ring[s][0][i], b4[i], acc0); acc1 = mfma4(ring[s][1][i], b4[i], acc1); }
            }
            if ((MODE & 2) && late) { PIN() b4n = q_frag(); q_load(k0 + s + 2); }
            PIN()
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const f32x4 r = acc0 + acc1 + acc2;
    out[(size_t)blockIdx.x * 512 + tid] = r.x + r.y + r.z + r.w;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int MODE, int DIST>
static void run(const char* name, const float* wts, long wfloats, const float* thin, float* out, unsigned long long* cyc, int steps) {
    for (int it = 0; it < 3; ++it) {
        hipLaunchKernelGGL((step_kernel<MODE, DIST>), dim3(256), dim3(512), 0, 0, wts, wfloats, thin, steps, out, cyc);
    }
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * 8);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> w3, w2;
    for (int b = 0; b < 256; ++b) {
        w3.push_back(std::max(h[b * 8 + 0], h[b * 8 + 1]) / (double)steps);
        double mx = 0;
        for (int w = 2; w < 8; ++w) mx = std::max(mx, (double)h[b * 8 + w]);
        w2.push_back(mx / steps);
    }
    std::sort(w3.begin(), w3.end()); std::sort(w2.begin(), w2.end());
    printf("%-34s ring %d: cycles/step  3-tile waves %.0f   2-tile waves %.0f   (bound: 20 MFMAs x 32 = 640 per SIMD pair 3+2, 512 for 2+2)\n", name,
           DIST + 1, w3[128], w2[128]);
}

int main(int argc, char** argv) {
    const long wfloats = (argc > 1 ? atol(argv[1]) : 512) * 1024;      // 2 MB of weight tiles by default (a power of two)
    float *wts, *thin, *out;
    unsigned long long* cyc;
    hipMalloc(&wts, wfloats * 4); hipMalloc(&thin, 8192 * 4); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
    std::vector<float> h(wfloats);
    for (long i = 0; i < wfloats; ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
    hipMemcpy(wts, h.data(), wfloats * 4, hipMemcpyHostToDevice);
    hipMemcpy(thin, h.data(), 8192 * 4, hipMemcpyHostToDevice);
    const int steps = 960;
    printf("weight tiles: %ld KiB\n", wfloats * 4 / 1024);
    run<0, 1>("MFMAs only (tiles re-read, L1)", wts, wfloats, thin, out, cyc, steps);
    run<1, 1>("+ weight stream (L2)", wts, wfloats, thin, out, cyc, steps);
    run<17, 1>("weight stream, nontemporal", wts, wfloats, thin, out, cyc, steps);
    run<33, 1>("weight stream, 4 of 8 waves", wts, wfloats, thin, out, cyc, steps);
    run<33, 3>("weight stream, 4 of 8 waves", wts, wfloats, thin, out, cyc, steps);
    run<3 + 64, 1>("both, waves 4-7 start late", wts, wfloats, thin, out, cyc, steps);
    run<3 + 128, 1>("both, waves 4-7 operand behind MFMAs", wts, wfloats, thin, out, cyc, steps);
    run<3 + 64 + 128, 1>("both, late + behind", wts, wfloats, thin, out, cyc, steps);
    if (argc > 2) return 0;
    run<2, 1>("+ thin layer (LDS + 20 VALU)", wts, wfloats, thin, out, cyc, steps);
    run<3, 1>("+ both", wts, wfloats, thin, out, cyc, steps);
    run<3, 2>("+ both", wts, wfloats, thin, out, cyc, steps);
    run<3, 3>("+ both", wts, wfloats, thin, out, cyc, steps);
    run<1, 3>("+ weight stream (L2)", wts, wfloats, thin, out, cyc, steps);
    run<5, 1>("weight stream, staggered WGs", wts, wfloats, thin, out, cyc, steps);
    run<7, 1>("both, staggered WGs", wts, wfloats, thin, out, cyc, steps);
    run<9, 1>("weight stream, exact loads", wts, wfloats, thin, out, cyc, steps);
    run<13, 1>("weight stream, exact, staggered", wts, wfloats, thin, out, cyc, steps);
    run<15, 1>("both, exact, staggered", wts, wfloats, thin, out, cyc, steps);
    run<15, 3>("both, exact, staggered", wts, wfloats, thin, out, cyc, steps);
    return 0;
}
