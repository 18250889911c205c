// What does other work cost beside v_mfma_f32_16x16x4_f32 on gfx950 (MI355X)?  Every wavefront runs ITER iterations of NM
// independent MFMAs (nine accumulators) with NX instructions of one kind interleaved in program order; 1, 2, 4 wavefronts per
// SIMD.  Finding (round 4): vector-ALU instructions do NOT hide under the matrix pipe - a SIMD's time is
// 32 x MFMAs + ~5 x VALU instructions, whatever the number of wavefronts - which makes the VALU count the quantity to minimise.
// (the s_add_i32 kind is kept in the kernel but not run: its loop is folded away by the compiler and times nothing)
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_valu_bench.hip -o tools/mfma_valu_bench && tools/mfma_valu_bench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

enum Kind { NONE, VFMA, VPKFMA, SADD, DSREAD, VCNDMASK, VMOV, GLOAD };
static const char* names[] = {"none", "v_fma_f32", "v_pk_fma_f32", "s_add_i32", "ds_read_b128", "v_cndmask_b32", "v_mov_b32", "global_load_dwordx4"};

template <int NM, int NX, int KIND>
__global__ __launch_bounds__(256) void k(float* out, const f32x4* src, int iters, float seed) {
    __shared__ f32x4 lds[256];
    lds[threadIdx.x] = f32x4{seed, seed, seed, seed};
    __syncthreads();
    f32x4 acc[9];
    for (int i = 0; i < 9; ++i) acc[i] = f32x4{seed, seed, seed, seed};
    float a = seed + threadIdx.x, b = seed * 0.5f;
    float v[8];
    f32x2 w[4];
    f32x4 ld[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
    for (int i = 0; i < 8; ++i) v[i] = seed + i;
    for (int i = 0; i < 4; ++i) w[i] = f32x2{seed, seed + i};
    int sacc = __builtin_amdgcn_readfirstlane(iters) * 3 + 1;      // (a value of its own: the asm below must not share the loop bound's register)
    const f32x4* gp = src + threadIdx.x;
    constexpr int PER = NM > 0 ? (NX + NM - 1) / NM : NX;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < (NM > 0 ? NM : 1); ++m) {
            if (NM > 0) acc[m % 9] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m % 9], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < PER; ++q) {
                if (m * PER + q >= NX) continue;
                const int r = (m + q) & 7;
                if (KIND == VFMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[r]) : "v"(b), "v"(a));
                if (KIND == VPKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(w[r & 3]) : "v"(w[(r + 1) & 3]));
                if (KIND == SADD) asm volatile("s_add_i32 %0, %0, 1" : "+s"(sacc));
                if (KIND == DSREAD) asm volatile("ds_read_b128 %0, %1" : "=v"(ld[r & 1]) : "v"((int)(threadIdx.x * 16)) : "memory");
                if (KIND == VCNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[r]) : "v"(b) : );
                if (KIND == VMOV) asm volatile("v_mov_b32 %0, %1" : "=v"(v[r]) : "v"(b));
                if (KIND == GLOAD) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ld[r & 1]) : "v"(gp) : "memory");
            }
        }
        if (KIND == DSREAD) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (KIND == GLOAD) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    float r = 0.f;
    for (int i = 0; i < 9; ++i) r += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    for (int i = 0; i < 8; ++i) r += v[i];
    for (int i = 0; i < 4; ++i) r += w[i].x + w[i].y;
    r += ld[0].x + ld[1].y;
    if (r == 12345.678f) out[threadIdx.x] = r + sacc;
}

template <int NM, int NX, int KIND>
static void run(int waves_per_simd, float* d, const f32x4* src) {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int iters = 2000;
    const int wgs = p.multiProcessorCount * waves_per_simd;   // 4 waves per workgroup = one per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NM, NX, KIND><<<wgs, 256>>>(d, src, 10, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NM, NX, KIND><<<wgs, 256>>>(d, src, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double clk = p.clockRate * 1e3;      // Hz (nominal)
    const double cyc = ms * 1e-3 * clk / iters;       // cycles per iteration of ALL waves of a SIMD
    const double mf = NM * 32.0 * waves_per_simd;
    printf("%d waves/SIMD, %2d MFMA + %3d %-20s: %8.1f cycles per round (MFMAs alone %5.0f) -> %5.2f cycles per extra instruction, MFMA busy %.2f\n",
           waves_per_simd, NM, NX, names[KIND], cyc, mf, NX > 0 ? (cyc - mf * 1.01) / (NX * waves_per_simd) : 0.0, mf / cyc);
}

int main() {
    float* d; hipMalloc(&d, 4096);
    f32x4* src; hipMalloc(&src, 1 << 20);
    for (int w : {1, 4}) {
        run<36, 0, NONE>(w, d, src);
        run<36, 144, VFMA>(w, d, src);
        run<36, 144, VPKFMA>(w, d, src);
        run<36, 144, VCNDMASK>(w, d, src);
        run<36, 144, VMOV>(w, d, src);
        run<36, 36, DSREAD>(w, d, src);
        run<36, 36, GLOAD>(w, d, src);
        run<0, 144, VFMA>(w, d, src);
        run<0, 144, VPKFMA>(w, d, src);
    }
    return 0;
}
