// HIP kernels (gfx950 / CDNA4 only) for HINT's recursive affine-coupling block.
//
// Arithmetic reproduced (reference, read-only): /root/reference/hint.py:62-101
//   per node:  v = [u | c];  s = mlp_s(v), t = mlp_t(v)            (hint.py:76-77, :10-13)
//              a = alpha*atan(s), alpha = clamp*0.636               (hint.py:56-60)
//   forward    l' = exp(a)*l + t ;  J += sum a   (children first)   (hint.py:70-80,97-99)
//   inverse    l  = (l' - t)/exp(a); J -= sum a  (root first)       (hint.py:82-88)
//
// Design: samples are independent, so one workgroup owns a tile of 16 batch
// rows (= one M-tile of v_mfma_f32_16x16x4_f32) and carries it through ALL tree levels of
// ALL blocks of a flow inside one launch; the lane tile, the conditioning input, both hidden
// activations and s/t live in LDS for the whole pass, HBM sees x once in and z, J once out
// (plus the training tape).  The eight wavefronts split the 16-wide output tiles of the s- and
// t-subnets of every node of a level as jobs of 1-3 tiles; weights stream from L2 as
// pre-packed MFMA B-fragments (one coalesced 1 KiB load per 16x16 tile), two k-blocks ahead of
// the MFMAs that consume them.  fp32 MFMA is an exact fp32 FMA chain, so results differ from
// the CPU reference only by summation order.
#include <hip/hip_runtime.h>
#include "hint_dev.h"

using namespace hint;

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Explicit LDS (address space 3) pointers for the tables staged in LDS.  A pointer that loses
// its address space is read with flat_load, and a flat access forces `s_waitcnt vmcnt(0)
// lgkmcnt(0)`, i.e. drains the whole packed-weight prefetch queue on every table read.
#define LDS_AS __attribute__((address_space(3)))
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef const LDS_AS GJob* lds_jobs_t;
__device__ __forceinline__ int lds_int(const LDS_AS int32_t* p) { return __builtin_amdgcn_readfirstlane(*p); }
#define GF(G, FIELD) lds_int(&(G)->FIELD)     // wave-uniform read of a DGroup field from LDS

// Diagnostic build only (-DHINT_STAMPS): shader-clock stamps of workgroup 0 at stage boundaries,
// written to a buffer nothing else reads (cdna_hip_programming.md §7 "In-kernel stamps").
#ifdef HINT_STAMPS
// Stamps are collected in LDS and flushed when the kernel ends: a global store per stamp would sit in
// the in-order vmcnt queue in front of the weight loads of the next stage and distort what it measures.
// Ids 0..127: stage boundaries; 128..255: job starts inside the GEMM stages (STAMP_JOBS / STAMP_JOB).
#define STAMP_IDS 512
__device__ unsigned long long* g_hint_stamps = nullptr;
__shared__ unsigned long long hint_stamp_lds[hint::NWAVES * STAMP_IDS];
__shared__ int hint_stamp_jb[hint::NWAVES];
__shared__ int hint_stamp_sub[hint::NWAVES];
#define STAMP(ID)                                                                              \
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) {                                           \
        unsigned long long t_;                                                                 \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");             \
        hint_stamp_lds[(threadIdx.x >> 6) * STAMP_IDS + (ID)] = t_;                             \
    }
#define STAMP_INIT()                                                                           \
    {                                                                                          \
        for (int i_ = threadIdx.x; i_ < hint::NWAVES * STAMP_IDS; i_ += blockDim.x) hint_stamp_lds[i_] = 0ull; \
        if (threadIdx.x < hint::NWAVES) { hint_stamp_jb[threadIdx.x] = 0; hint_stamp_sub[threadIdx.x] = 0; } \
        __syncthreads();                                                                       \
    }
#define STAMP_FLUSH()                                                                          \
    {                                                                                          \
        __syncthreads();                                                                       \
        if (g_hint_stamps != nullptr && blockIdx.x == 0)                                        \
            for (int i_ = threadIdx.x; i_ < hint::NWAVES * STAMP_IDS; i_ += blockDim.x)         \
                if (hint_stamp_lds[i_] != 0ull) g_hint_stamps[i_] = hint_stamp_lds[i_];         \
    }
#define STAMP_JOBS(BASE) { if ((threadIdx.x & 63) == 0) hint_stamp_jb[threadIdx.x >> 6] = (BASE); }
#define STAMP_JOB(JI)                                                                          \
    {                                                                                          \
        const int b_ = hint_stamp_jb[threadIdx.x >> 6];                                         \
        if (b_ != 0) STAMP(b_ + ((JI) < 11 ? (JI) : 11))                                        \
    }
// a sequential log (ids 256..511: section K << 56 | time) inside the stage whose job base is HINT_STAMP_STAGE
#ifdef HINT_STAMP_STAGE
#define STAMP_SUB(K)                                                                           \
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && hint_stamp_jb[threadIdx.x >> 6] == (HINT_STAMP_STAGE)) { \
        const int n_ = hint_stamp_sub[threadIdx.x >> 6];                                        \
        if (n_ < 256) {                                                                        \
            unsigned long long t_;                                                             \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");         \
            hint_stamp_lds[(threadIdx.x >> 6) * STAMP_IDS + 256 + n_] = t_ | ((unsigned long long)(K) << 56); \
            hint_stamp_sub[threadIdx.x >> 6] = n_ + 1;                                          \
        }                                                                                      \
    }
#else
#define STAMP_SUB(K)
#endif
#else
#define STAMP_SUB(K)
#define STAMP(ID)
#define STAMP_INIT()
#define STAMP_FLUSH()
#define STAMP_JOBS(BASE)
#define STAMP_JOB(JI)
#endif

// Workgroup barrier that orders LDS traffic only.  __syncthreads() makes hipcc drain vmcnt(0)
// first, which would retire the packed-weight prefetch of the next stage at every stage
// boundary; here only LDS operations (lgkmcnt) are waited for, global loads stay in flight
// across the barrier (cdna_hip_programming.md §5 "Pipelining across barriers").  Global
// stores issued before it are never read back inside the kernel.
__device__ __forceinline__ void lds_barrier() {
#ifdef HINT_NO_BARRIER      // diagnostic only (results are wrong): how much do the stage barriers cost?
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}

// floor(i / d) for 0 <= i < 2^20 through the float unit: three instructions instead of the ~25 of an
// integer division by a run-time divisor (the tile loops below split a flat index into row and column;
// they made up a third of the vector instructions of the backward kernel).  inv = frcp(d).
__device__ __forceinline__ float frcp(int d) { return __builtin_amdgcn_rcpf((float)d); }
__device__ __forceinline__ int fdiv(int i, float inv) { return (int)(((float)i + 0.5f) * inv); }

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
#ifdef HINT_ABLATE_MFMA     // diagnostic: keep operands alive, skip the matrix pipe
    asm volatile("" ::"v"(a), "v"(b));
    return c;
#else
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
#endif
}

// =======================================================================================
// weight packing: flat torch-layout parameters -> MFMA fragment order (see hint_dev.h)
// =======================================================================================
__device__ __forceinline__ void pack_body(int bid, const PackSeg* __restrict__ segs,
                                          const int2* __restrict__ ptiles, int n_tiles,
                                          const int32_t* __restrict__ bmap, int n_bias, long bias_off,
                                          const float* __restrict__ P, float* __restrict__ packed) {
    if (bid >= n_tiles) {
        // trailing workgroups: biases laid out per group in LDS column order (zero for padding)
        const int i = (bid - n_tiles) * 256 + (int)threadIdx.x;
        if (i < n_bias) { const int off = bmap[i]; packed[bias_off + i] = off >= 0 ? P[off] : 0.f; }
        return;
    }
    const int2 pt = ptiles[bid];
    const PackSeg sg = segs[pt.x];
    const int nt = pt.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = nt * 16 + (lane & 15), kq = lane >> 4;
    for (int kb = wave; kb < sg.NB; kb += 4) {
        f32x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = kb * 16 + 4 * kq + i;
            float val = 0.f;
            if (n < sg.N && k < sg.K) {
                if (sg.mode == 0) val = P[sg.src0 + (int64_t)n * sg.ld + k];
                else if (sg.mode == 1) val = P[sg.src0 + (int64_t)k * sg.ld + n];
                else {
                    const int net = k / sg.hp, k2 = k - net * sg.hp;
                    if (k2 < sg.h) val = P[(net ? sg.src1 : sg.src0) + (int64_t)k2 * sg.ld + n];
                }
            }
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

// =======================================================================================
// Generic GEMM stage:  out[16][N] (+)= A[16][K] * Wlog^T  for every (node, net) of a group.
// The host cuts the stage into JOBS of 1..3 adjacent 16-column output tiles that share their A
// operand (same node, same net input) and deals them to the 8 wavefronts (hint_plan.cpp,
// emit_stage).  A wavefront runs a job as ONE plain k-loop over the job's 16-wide k-blocks:
//     step kb :  global fetch of the packed B tiles of k-block kb+2   (3 register sets rotate)
//                LDS read of the A fragment of k-block kb+1           (shared by the NT tiles)
//                4*NT MFMAs of k-block kb                              (NT independent accumulators)
// and the epilogue of all NT tiles at the end.  The first two k-blocks of a job are fetched
// by its PREDECESSOR right before that one's epilogue - also across the workgroup barrier
// between two stages (weights do not depend on the previous stage, only A in LDS does) - so
// a job's loop starts with its weights in flight or landed.  A micro-benchmark of this loop
// against a per-chunk interpreter (tools/stage_bench.hip) is what the structure comes from:
// sharing A and keeping the inner loop free of record decoding is worth 1.3-1.9x per stage.
//   MFMA lane map (16x16x4 f32): lane l supplies A[m = l&15][kslot = l>>4] and
//   B[kslot][n = l&15]; result reg i = C[4*(l>>4)+i][l&15].  Slot kq of step i of a 16-wide
//   k-block is k = 16*kb + 4*kq + i on both operands, so each lane reads 4 consecutive k with
//   one 128-bit access (LDS for A, global for packed B).
// Job records and biases live in LDS (staged per group, one group ahead).
// =======================================================================================
enum { EPI_RELU = 0, EPI_LINEAR = 1, EPI_MASK = 2, EPI_PLAIN = 3 };

struct JobU { int wtile, acol, ocol, nb, nt, nvalid, slab, tstride, count; };

__device__ __forceinline__ JobU decode_job(i32x4 raw) {
    JobU u;
    u.wtile = __builtin_amdgcn_readfirstlane(raw.x);
    const unsigned y = (unsigned)__builtin_amdgcn_readfirstlane(raw.y);
    const unsigned z = (unsigned)__builtin_amdgcn_readfirstlane(raw.z);
    const unsigned w = (unsigned)__builtin_amdgcn_readfirstlane(raw.w);
    u.acol = (int)(y & 0xffffu);
    u.ocol = (int)(y >> 16);
    u.nb = (int)(z & 0xffu);
    u.nt = (int)((z >> 8) & 0xffu);
    u.nvalid = (int)((z >> 16) & 0xffu);
    u.slab = (int)(z >> 24);
    u.tstride = (int)(w & 0xffffu);
    u.count = (int)(w >> 16);
    return u;
}

// What a wavefront carries from one stage to the next: its position in the job lists and the
// first job of the coming stage with the B tiles of that job's first two k-blocks in flight.
struct Stage {
    lds_jobs_t cl;      // this wavefront's job list of the coming stage (LDS)
    JobU j;             // its first job (count = list length; an idle wavefront has one nt = 0 job)
    f32x4 b0[3], b1[3]; // packed B tiles of k-blocks 0 and 1 of that job's (up to) three tiles
};

#ifdef HINT_ABLATE_WLOAD    // diagnostic: every weight fetch hits the same few KiB (L1-resident)
#define HINT_WTILE(T) ((size_t)((T) & 3))
#else
#define HINT_WTILE(T) ((size_t)(T))
#endif

// Always six loads (exact vmcnt bookkeeping for the compiler): a job with fewer tiles or a
// single k-block re-reads a tile it fetches anyway, which the vector L1 serves.
__device__ __forceinline__ void fetch_first(f32x4 (&b0)[3], f32x4 (&b1)[3], const JobU& j,
                                            const float* __restrict__ packed, int lane) {
    const f32x4* wp = (const f32x4*)packed + lane;
    const int k1 = j.nb > 1 ? 1 : 0;
    const int base = j.nb > 0 ? j.wtile : 0;          // K = 0 jobs and outer-product tiles have no weights
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int tt = (j.nb > 0 && t < j.nt) ? t : 0;
        const f32x4* q = wp + HINT_WTILE(base + tt * j.tstride) * 64;
        b0[t] = q[0];
        b1[t] = q[64 * k1];
    }
}

// desc = offset | stride << 16: wavefront w's list starts at jl + offset + w*stride
__device__ __forceinline__ lds_jobs_t stage_list(lds_jobs_t jl, int desc, int wave) {
    return jl + (desc & 0xffff) + wave * (desc >> 16);
}

// Prime S for a stage from scratch (first stage of a kernel; everything after that is primed by
// the stage_run() in front of it).
__device__ __forceinline__ void stage_begin(Stage& S, lds_jobs_t cl, const float* __restrict__ packed, int lane) {
    S.cl = cl;
#ifdef HINT_SKIP_GEMM          // diagnostic: no GEMM stage work at all
    return;
#endif
    S.j = decode_job(*(const LDS_AS i32x4*)cl);
    fetch_first(S.b0, S.b1, S.j, packed, lane);
}

// One job: NT tiles x j.nb k-blocks.  b0/b1 hold k-blocks 0 and 1 on entry and k-blocks 0 and
// 1 of job `jn` (fetched from packed_n) on exit.
template <int EPI, int NT>
__device__ __forceinline__ void run_job(const JobU& j, f32x4 (&b0)[3], f32x4 (&b1)[3], const JobU& jn,
                                        const float* __restrict__ packed, const float* __restrict__ packed_n,
                                        const float* bias_lds, const float* arow, float* O, const float* Mk,
                                        int ldo, int slab_stride, int lane) {
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nl = lane & 15;
    float bias[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        bias[t] = 0.f;
        if (EPI == EPI_RELU || EPI == EPI_LINEAR) bias[t] = bias_lds[j.ocol + 16 * t + nl];   // zero in padding columns
    }
    STAMP_SUB(3)
    if (j.nb > 0) {
        const int NB = j.nb;
        const float* ap = arow + j.acol;
        const f32x4* wp = (const f32x4*)packed + HINT_WTILE(j.wtile) * 64 + lane;
        const int ts = j.tstride;
        f32x4 b2[NT], a0, a1;
#ifdef HINT_ABLATE_AREAD
        a0 = f32x4{1.f, 2.f, 3.f, 4.f}; a1 = a0;
#else
        a0 = *(const f32x4*)ap;
#endif
        int kb = 0;
        // (forward epilogues only: in the backward kernel, which sits at the register limit, the extra
        // path costs 36 more bytes of scratch per lane and 17 us)
        if ((EPI == EPI_RELU || EPI == EPI_LINEAR) && NB == 1) {
            // a thin job (K <= 16: the first layers): no ring, no loads, one way out
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = mfma4(a0.x, b0[t].x, acc[t]);
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = mfma4(a0.y, b0[t].y, acc[t]);
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = mfma4(a0.z, b0[t].z, acc[t]);
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = mfma4(a0.w, b0[t].w, acc[t]);
        } else
        // loads past the job's last k-block re-read that block (L1 hit, never used)
#ifdef HINT_ABLATE_AREAD
#define HINT_AREAD(AN, KA)
#else
#define HINT_AREAD(AN, KA) AN = *(const f32x4*)(ap + 16 * (KA));
#endif
#define HINT_STEP(AC, AN, BC, BN2)                                                                     \
        {                                                                                              \
            const int kn = kb + 2 < NB ? kb + 2 : NB - 1;                                              \
            _Pragma("unroll") for (int t = 0; t < NT; ++t) BN2[t] = wp[(size_t)(t * ts + kn) * 64];    \
            const int ka = kb + 1 < NB ? kb + 1 : NB - 1;                                              \
            HINT_AREAD(AN, ka)                                                                         \
            _Pragma("unroll") for (int t = 0; t < NT; ++t) acc[t] = mfma4(AC.x, BC[t].x, acc[t]);      \
            _Pragma("unroll") for (int t = 0; t < NT; ++t) acc[t] = mfma4(AC.y, BC[t].y, acc[t]);      \
            _Pragma("unroll") for (int t = 0; t < NT; ++t) acc[t] = mfma4(AC.z, BC[t].z, acc[t]);      \
            _Pragma("unroll") for (int t = 0; t < NT; ++t) acc[t] = mfma4(AC.w, BC[t].w, acc[t]);      \
            ++kb;                                                                                      \
        }
        while (true) {
            HINT_STEP(a0, a1, b0, b2) if (kb >= NB) break;
            HINT_STEP(a1, a0, b1, b0) if (kb >= NB) break;
            HINT_STEP(a0, a1, b2, b1) if (kb >= NB) break;
            HINT_STEP(a1, a0, b0, b2) if (kb >= NB) break;
            HINT_STEP(a0, a1, b1, b0) if (kb >= NB) break;
            HINT_STEP(a1, a0, b2, b1) if (kb >= NB) break;
        }
#undef HINT_STEP
#undef HINT_AREAD
    }
    STAMP_SUB(5)
    // the successor's first weights go out before this job's epilogue
    fetch_first(b0, b1, jn, packed_n, lane);
    STAMP_SUB(6)
#ifdef HINT_ABLATE_EPI
    if (jn.nt > 100)
#endif
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const bool ok = (t < NT - 1) || nl < j.nvalid;       // only the job's last tile can be ragged
        const int oo = j.slab * slab_stride + (4 * (lane >> 4)) * ldo + j.ocol + 16 * t + nl;
        float* o = O + oo;
        const float* mk = Mk + oo;                         // EPI_MASK: relu'() of the forward activation
        const float bi = j.slab == 0 ? bias[t] : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float v;
            if (EPI == EPI_RELU) v = ok ? fmaxf(acc[t][i] + bi, 0.f) : 0.f;
            else if (EPI == EPI_LINEAR) v = ok ? acc[t][i] + bi : 0.f;
            else if (EPI == EPI_MASK) v = (ok && mk[i * ldo] > 0.f) ? acc[t][i] : 0.f;
            else v = ok ? acc[t][i] : 0.f;
            o[i * ldo] = v;
        }
    }
}

// One 16x16 outer-product tile of a thin weight gradient (TJOB_OUTER record): the reduction runs
// over the 16 rows of the row tile; the partial goes to this row tile's own slab of the
// workspace with plain stores (every slab element is written exactly once) and part B sums the
// slabs.  Atomics straight into the gradient buffer would have all 256 workgroups hammer the
// same few KiB at once: measured 65 us of a 115 us kernel.
__device__ __forceinline__ void run_outer(const JobU& j, f32x4 (&b0)[3], f32x4 (&b1)[3], const JobU& jn,
                                          const float* __restrict__ packed_n, const float* Abuf, int lda,
                                          const float* Bbuf, int ldb, float* __restrict__ g,
                                          float* __restrict__ trash, int lane) {
    const int nl = lane & 15, kq = lane >> 4;
    // a run of adjacent tiles per record, walking along n (dir 0) or m (dir 1), three at a time: a
    // record's fixed cost (decode, baton, LDS round trips) is ten times a tile's.  The body is branch
    // free - a wavefront issues one instruction per four cycles whatever its kind, and the compiler's
    // version of "store if valid" was fifty branches per round: elements outside the matrix (and the
    // tiles a short last round computes again) go to a 64-float dump behind the slabs instead.
    const int dir = j.slab & 1, cnt = j.slab >> 1;
    const int astep = dir ? 16 : 0, bstep = dir ? 0 : 16;
    const float* ap = Abuf + kq * lda + nl + j.acol;       // rows kq + 4*i of the two operand tiles
    const float* bp = Bbuf + kq * ldb + nl + j.ocol;
    const int mlast = (j.nvalid & 15) + 1, nlast = (j.nvalid >> 4) + 1;
    const int gstep = dir ? 16 * j.tstride : 16;
    float* o0 = g + j.wtile + (4 * kq) * j.tstride + nl;
    float* dump = trash + lane;
    fetch_first(b0, b1, jn, packed_n, lane);
    for (int t0 = 0; t0 < cnt; t0 += 3) {
        const int c3 = cnt - t0;                           // tiles left (this round does min(c3, 3))
        float av[3][4], bv[3][4];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int tt = t0 + (t < c3 ? t : 0);          // (a short round re-reads its first tile)
#pragma unroll
            for (int i = 0; i < 4; ++i) { av[t][i] = ap[4 * i * lda + tt * astep]; bv[t][i] = bp[4 * i * ldb + tt * bstep]; }
        }
        f32x4 acc[3];      // (three tiles interleave: consecutive MFMAs on one accumulator are three apart)
#pragma unroll
        for (int t = 0; t < 3; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int t = 0; t < 3; ++t) acc[t] = mfma4(av[t][i], bv[t][i], acc[t]);
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const f32x4 r = acc[t];
            const bool lastt = t0 + t + 1 >= cnt;
            const int mvalid = t >= c3 ? 0 : ((dir && !lastt) ? 16 : mlast), nvalid = (!dir && !lastt) ? 16 : nlast;
            float* o = o0 + (t0 + t) * gstep;
            const int mrem = nl < nvalid ? mvalid - 4 * kq : 0;      // valid rows among this lane's four
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#ifdef HINT_ABLATE_OUTER_STORE
                asm volatile("" ::"v"(r[i]), "v"(o));
#else
                float* p = i < mrem ? o + i * j.tstride : dump;
                *p = r[i];
#endif
            }
        }
    }
}

// Run the stage S is primed for; on return S is primed for the stage whose list (of this
// wavefront) is `next` with weights in `packed_n`.  Mk: EPI_MASK's relu mask source (laid out
// like O; may be O itself).  OUTER: the lists also hold outer-product tiles A_o^T B_o -> g_o.
template <int EPI, bool OUTER = false>
__device__ __forceinline__ void stage_run(Stage& S, lds_jobs_t next, const float* __restrict__ packed,
                                          const float* __restrict__ packed_n, const float* bias_lds,
                                          const float* A, int lda, float* O, const float* Mk, int ldo,
                                          int slab_stride, int lane, const float* A_o = nullptr, int lda_o = 0,
                                          const float* B_o = nullptr, int ldb_o = 0,
                                          float* __restrict__ g_o = nullptr, float* __restrict__ trash = nullptr) {
#ifdef HINT_SKIP_GEMM
    S.cl = next;
    return;
#endif
    const float* arow = A + (lane & 15) * lda + 4 * (lane >> 4);
    JobU j = S.j;
    const int n = j.count;
    for (int ji = 0;; ++ji) {
        const bool last = ji + 1 >= n;
        STAMP_JOB(ji)
        STAMP_SUB(1)
        const JobU jn = decode_job(*(const LDS_AS i32x4*)(last ? next : S.cl + ji + 1));
        const float* pn = last ? packed_n : packed;
#ifdef HINT_ABLATE_OUTER
        if (OUTER && j.nt == TJOB_OUTER) fetch_first(S.b0, S.b1, jn, pn, lane);
#else
        if (OUTER && j.nt == TJOB_OUTER) run_outer(j, S.b0, S.b1, jn, pn, A_o, lda_o, B_o, ldb_o, g_o, trash, lane);
#endif
        else if (j.nt >= 3) run_job<EPI, 3>(j, S.b0, S.b1, jn, packed, pn, bias_lds, arow, O, Mk, ldo, slab_stride, lane);
        else if (j.nt == 2) run_job<EPI, 2>(j, S.b0, S.b1, jn, packed, pn, bias_lds, arow, O, Mk, ldo, slab_stride, lane);
        else if (j.nt == 1) run_job<EPI, 1>(j, S.b0, S.b1, jn, packed, pn, bias_lds, arow, O, Mk, ldo, slab_stride, lane);
        else fetch_first(S.b0, S.b1, jn, pn, lane);      // idle wavefront: only hand the baton on
        STAMP_SUB(7)
        j = jn;
        if (last) break;
    }
    S.j = j;
    S.cl = next;
}

// bias gradients: column sums over the 16 rows of an LDS buffer, one thread per column; the map
// (compact thin-gradient index per column, -1 for padding) sits in LDS next to the biases
__device__ __forceinline__ void colsum_store(const int32_t* map_lds, int ncols, const float* buf, int ld,
                                             float* __restrict__ g, int tid) {
#ifdef HINT_SKIP_COLSUM
    return;
#endif
    for (int col = tid; col < ncols; col += NTHREADS) {
        const int off = map_lds[col];
        float v[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) v[r] = buf[r * ld + col];
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < ROWS; r += 4) s += (v[r] + v[r + 1]) + (v[r + 2] + v[r + 3]);
#ifdef HINT_ABLATE_OSTORE
        asm volatile("" ::"v"(s));
#else
        if (off >= 0) g[off] = s;
#endif
    }
}

struct EntU { int xcol, scol, tcol; };
__device__ __forceinline__ EntU load_ent(const LDS_AS Ent* ents, int e) {
    const LDS_AS int32_t* p = (const LDS_AS int32_t*)(ents + e);           // 8 bytes, per-lane address
    const unsigned w0 = (unsigned)p[0], w1 = (unsigned)p[1];
    EntU u;
    u.xcol = (int)(w0 & 0xffffu); u.scol = (int)(w0 >> 16); u.tcol = (int)(w1 & 0xffffu);
    return u;
}

// All fields of a group descriptor in one burst of seven 128-bit LDS reads (wave-uniform).
struct GroupU {
    int node_begin, node_end, jl_begin, jl_count, l1_off, l2_off, l3_off, g2_off, g1_off, dv_off, o3_off, o3_cnt,
        o1_off, o1_cnt, ent_begin, ent_cnt, bmap_begin, bmap3_begin, aw, vw, sw, l3_slabs, dv_slabs, wcol0, level,
        level_last, vmap_begin, cont;     // cont: 0 whole nodes, 1 / 2 the t / s unit of a node whose nets run one at a time
};
__device__ __forceinline__ GroupU load_group(const LDS_AS DGroup* g) {
    const LDS_AS i32x4* p = (const LDS_AS i32x4*)g;
    const i32x4 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3], q4 = p[4], q5 = p[5], q6 = p[6];
#define RFL(V) __builtin_amdgcn_readfirstlane(V)
    GroupU u;
    u.node_begin = RFL(q0.x); u.node_end = RFL(q0.y); u.jl_begin = RFL(q0.z); u.jl_count = RFL(q0.w);
    u.l1_off = RFL(q1.x); u.l2_off = RFL(q1.y); u.l3_off = RFL(q1.z); u.g2_off = RFL(q1.w);
    u.g1_off = RFL(q2.x); u.dv_off = RFL(q2.y); u.o3_off = RFL(q2.z); u.o3_cnt = RFL(q2.w);
    u.o1_off = RFL(q3.x); u.o1_cnt = RFL(q3.y); u.ent_begin = RFL(q3.z); u.ent_cnt = RFL(q3.w);
    u.bmap_begin = RFL(q4.x); u.bmap3_begin = RFL(q4.y); u.aw = RFL(q4.z); u.vw = RFL(q4.w);
    u.sw = RFL(q5.x); u.l3_slabs = RFL(q5.y); u.dv_slabs = RFL(q5.z); u.wcol0 = RFL(q5.w);
    u.level = RFL(q6.x); u.level_last = RFL(q6.y); u.vmap_begin = RFL(q6.z); u.cont = RFL(q6.w);
#undef RFL
    return u;
}

__device__ __forceinline__ float row16_sum(float v) {
    // deterministic butterfly over the 16 lanes that share a batch row
    v += __shfl_xor(v, 8, 16);
    v += __shfl_xor(v, 4, 16);
    v += __shfl_xor(v, 2, 16);
    v += __shfl_xor(v, 1, 16);
    return v;
}

// v = [u | c] for every node of the group (hint.py:76), zero padded: one pass over the group's
// v columns through a per-column source map (>= 0: lane column, -1: zero, <= -2: condition column)
__device__ __forceinline__ void stage_build_v(const KArgs& a, const LDS_AS int16_t* vmap, int vw,
                                              const float* xs, const float* cs, float* vb, int tid) {
    const float inv = frcp(vw);
    for (int i = tid; i < ROWS * vw; i += NTHREADS) {
        const int r = fdiv(i, inv), j = i - r * vw;
        const int m = vmap[j];
        float v = 0.f;
        if (m >= 0) v = xs[r * a.xld + m];
        else if (m <= -2) v = cs[r * a.cld + (-2 - m)];
        vb[r * a.vld + j] = v;
    }
}

__device__ __forceinline__ void load_tile(float* dst, int ld, const float* __restrict__ src,
                                          int width, int row0, int B, int tid) {
    // a 16-row tile of a row-major [B,width] tensor is one contiguous run of 16*width floats
    if (src == nullptr) {
        const float inv0 = frcp(width);
        for (int i = tid; i < ROWS * width; i += NTHREADS) { const int r = fdiv(i, inv0); dst[r * ld + (i - r * width)] = 0.f; }
        return;
    }
    const float* p = src + (size_t)row0 * width;
    const int nvalid = (B - row0 < ROWS ? B - row0 : ROWS) * width;
    const float inv = frcp(width);
    for (int i = tid; i < ROWS * width; i += NTHREADS) {
        const int r = fdiv(i, inv);
        dst[r * ld + (i - r * width)] = (i < nvalid) ? p[i] : 0.f;
    }
}

// A [16, width] tile held in registers between its global load and its LDS commit, so that the
// load latency hides behind other work (width <= TILE_REGS * NTHREADS / 16 floats per row).
constexpr int TILE_REGS = 4;
struct TilePrefetch { float r[TILE_REGS]; };
__device__ __forceinline__ void tile_issue(TilePrefetch& tp, const float* __restrict__ src, int width, int row0,
                                           int B, int tid) {
    const float* p = src + (size_t)row0 * width;
    const int nvalid = (B - row0 < ROWS ? B - row0 : ROWS) * width;
#pragma unroll
    for (int k = 0; k < TILE_REGS; ++k) {
        const int i = tid + k * NTHREADS;
        tp.r[k] = (i < nvalid) ? p[i] : 0.f;
    }
}
__device__ __forceinline__ void tile_commit(const TilePrefetch& tp, float* dst, int ld, int width, int tid) {
    const float inv = frcp(width);
#pragma unroll
    for (int k = 0; k < TILE_REGS; ++k) {
        const int i = tid + k * NTHREADS;
        if (i < ROWS * width) { const int r = fdiv(i, inv); dst[r * ld + (i - r * width)] = tp.r[k]; }
    }
}

__device__ __forceinline__ void store_tile(float* __restrict__ dst, const float* src, int ld,
                                           int width, int row0, int B, int tid, int nthreads = NTHREADS) {
    float* p = dst + (size_t)row0 * width;
    const int nvalid = (B - row0 < ROWS ? B - row0 : ROWS) * width;
    const float inv = frcp(width);
    for (int i = tid; i < nvalid; i += nthreads) {
        const int r = fdiv(i, inv);
        p[i] = src[r * ld + (i - r * width)];
    }
}

// 16 rows x `width` columns of an LDS buffer to a [rows][dld] global array (hidden activations)
// NT: non-temporal stores for rows that are read much later (the forward's tape: tens of MB on,
// by another kernel), plain ones for rows the next kernel reads (g2 -> part B).
template <bool NT>
__device__ __forceinline__ void copy_rows_out(float* __restrict__ dst, int dld, int dcol,
                                              const float* src, int sld, int width, int row0,
                                              int tid, int nthreads = NTHREADS) {
    // width is a multiple of 16, dcol/dld multiples of 4 -> 128-bit rows
    const int w4 = width >> 2;
#ifdef HINT_SKIP_WSCOPY
    if (width > 0) return;
#endif
    const float inv = frcp(w4);
    for (int i = tid; i < ROWS * w4; i += nthreads) {
        const int r = fdiv(i, inv), j = (i - r * w4) << 2;
        const f32x4 v = *(const f32x4*)(src + r * sld + j);
        f32x4* p = (f32x4*)(dst + (size_t)(row0 + r) * dld + dcol + j);
        if (NT) __builtin_nontemporal_store(v, p);
        else *p = v;
    }
}

// The job list of the NEXT group travels global -> registers at the start of a group and
// registers -> LDS near its end, so its L2 latency hides behind the group's GEMM stages.
// (The group's biases, a few KiB, ride along: [b1 | b2 | b3] in LDS column order.)
struct JobPrefetch { i32x4 r0, r1, m0; f32x4 b0; int count, nbias4; };

// BMAP: also fetch the group's bias-gradient map (backward kernel; the compact thin-gradient
// index of every bias column, or -1), which lands behind the biases in the LDS bias buffer.
template <bool BMAP>
__device__ __forceinline__ void jobs_issue(JobPrefetch& jp, const GJob* __restrict__ gjobs, int jl_begin,
                                           int jl_count, const float* __restrict__ bias_src,
                                           const int32_t* __restrict__ bmap_src, int nbias, int tid) {
    jp.count = jl_count;
    jp.nbias4 = nbias >> 2;                       // bias blocks are multiples of 16 floats, <= 4*NTHREADS
    const i32x4* src = (const i32x4*)(gjobs + jl_begin);
    if (tid < jp.count) jp.r0 = src[tid];
    if (tid + NTHREADS < jp.count) jp.r1 = src[tid + NTHREADS];
    if (tid < jp.nbias4) {
        jp.b0 = ((const f32x4*)bias_src)[tid];
        if (BMAP) jp.m0 = ((const i32x4*)bmap_src)[tid];
    }
}
template <bool BMAP>
__device__ __forceinline__ void jobs_commit(const JobPrefetch& jp, LDS_AS GJob* jbuf, float* bias_dst, int bmax,
                                            int tid) {
    if (tid < jp.count) ((LDS_AS i32x4*)jbuf)[tid] = jp.r0;
    if (tid + NTHREADS < jp.count) ((LDS_AS i32x4*)jbuf)[tid + NTHREADS] = jp.r1;
    if (tid < jp.nbias4) {
        ((f32x4*)bias_dst)[tid] = jp.b0;
        if (BMAP) ((i32x4*)(bias_dst + bmax))[tid] = jp.m0;
    }
}

// LDS carve-up shared by both block kernels:
//   [meta: groups | vmap | ents][job buffer 0 | 1][bias+bmap buffer 0 | 1][float buffers ...]
// The meta copy is split into issue (global -> registers) and commit (registers -> LDS) so
// that it shares ONE memory round trip with the first job lists and the first lane tile.
struct MetaPrefetch { i32x4 r0, r1; };
__device__ __forceinline__ void meta_issue(MetaPrefetch& mp, const KArgs& a, int tid) {
    const int n16 = a.meta_bytes >> 4;
    if (tid < n16) mp.r0 = ((const i32x4*)a.meta)[tid];
    if (tid + NTHREADS < n16) mp.r1 = ((const i32x4*)a.meta)[tid + NTHREADS];
}
__device__ __forceinline__ void meta_commit(const MetaPrefetch& mp, const KArgs& a, LDS_AS char* mbase, int tid) {
    const int n16 = a.meta_bytes >> 4;
    if (tid < n16) ((LDS_AS i32x4*)mbase)[tid] = mp.r0;
    if (tid + NTHREADS < n16) ((LDS_AS i32x4*)mbase)[tid + NTHREADS] = mp.r1;
    for (int i = tid + 2 * NTHREADS; i < n16; i += NTHREADS) ((LDS_AS i32x4*)mbase)[i] = ((const i32x4*)a.meta)[i];
}
#define HINT_LDS_TABLES()                                                                          \
    LDS_AS char* mbase = (LDS_AS char*)lds;                                                        \
    const LDS_AS DGroup* groups = (const LDS_AS DGroup*)mbase;                                     \
    const LDS_AS int16_t* vmap = (const LDS_AS int16_t*)(mbase + a.vmap_off);                      \
    const LDS_AS Ent* ents = (const LDS_AS Ent*)(mbase + a.ents_off);                              \
    LDS_AS GJob* jbuf0 = (LDS_AS GJob*)(mbase + a.meta_bytes);                                     \
    float* bias0 = lds + ((a.meta_bytes + 2 * a.jmax * (int)sizeof(GJob)) >> 2);                   \
    float* fbase = bias0 + 4 * a.bmax;                                                             \
    MetaPrefetch mp_;                                                                              \
    meta_issue(mp_, a, tid);

// =======================================================================================
// forward (REV=false) / inverse (REV=true): x, J -> z   — one launch per block
// =======================================================================================
// Pointers that reach a kernel inside a by-value struct or a device table are generic ("flat") to
// the compiler; a flat load counts against vmcnt AND lgkmcnt and so serialises with the LDS
// traffic.  Laundering them through address space 1 tells it they are global memory.
#define GLOBAL_AS __attribute__((address_space(1)))
struct GBlock {      // ChainBlock with the pointers typed as global memory
    const GLOBAL_AS float* params;
    const GLOBAL_AS float* packed;
    const GLOBAL_AS float* perm;
    GLOBAL_AS float* tape;
    GLOBAL_AS float* wsA1;
    GLOBAL_AS float* wsG2;
    GLOBAL_AS float* wsT;
    GLOBAL_AS float* gparams;
};
__device__ __forceinline__ GBlock chain_block(const ChainBlock* __restrict__ chain, const ChainBlock& one, int i) {
    const ChainBlock b = (chain != nullptr) ? chain[i] : one;
    GBlock g;
    g.params = (const GLOBAL_AS float*)b.params;
    g.packed = (const GLOBAL_AS float*)b.packed;
    g.perm = (const GLOBAL_AS float*)b.perm;
    g.tape = (GLOBAL_AS float*)b.tape;
    g.wsA1 = (GLOBAL_AS float*)b.wsA1;
    g.wsG2 = (GLOBAL_AS float*)b.wsG2;
    g.wsT = (GLOBAL_AS float*)b.wsT;
    g.gparams = (GLOBAL_AS float*)b.gparams;
    return g;
}

// sum_k row[k] * w[k * stride]: the fixed d x d permutation matrices are read straight from
// global memory (L1/L2 resident); eight loads are issued before the first FMA, otherwise the loop
// is one L1 round trip per term (13 us per block at d = 43).
__device__ __forceinline__ float perm_dot(const float* row, const float* __restrict__ w, int stride, int d) {
    float acc = 0.f;
    int k = 0;
    for (; k + 8 <= d; k += 8) {
        float wv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) wv[u] = w[(size_t)(k + u) * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = fmaf(row[k + u], wv[u], acc);
    }
    for (; k < d; ++k) acc = fmaf(row[k], w[(size_t)k * stride], acc);
    return acc;
}

// Philox4x32-10 (Salmon et al., SC'11) + Box-Muller: four standard normals per (key, counter).
// Used for the dequantisation noise of a training step (train_unconditional.py:121,
// x += 0.01*randn_like(x)) so that it costs no extra launch and no HBM round trip.
__device__ __forceinline__ void philox_normal4(unsigned long long seed, unsigned long long step, unsigned idx,
                                               float (&out)[4]) {
    unsigned c0 = idx, c1 = (unsigned)step, c2 = (unsigned)(step >> 32), c3 = 0x48494e54u;
    unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    const float u0 = ((float)c0 + 1.0f) * 2.3283064365386963e-10f;     // (0, 1]
    const float u1 = (float)c1 * 2.3283064365386963e-10f;
    const float u2 = ((float)c2 + 1.0f) * 2.3283064365386963e-10f;
    const float u3 = (float)c3 * 2.3283064365386963e-10f;
    // hardware transcendentals: v_log_f32 is log2, v_sin/v_cos take their argument in turns
    const float r0 = __builtin_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(fminf(u0, 1.0f)));
    const float r1 = __builtin_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(fminf(u2, 1.0f)));
    const float s0 = __builtin_amdgcn_sinf(u1), cs0 = __builtin_amdgcn_cosf(u1);
    const float s1 = __builtin_amdgcn_sinf(u3), cs1 = __builtin_amdgcn_cosf(u3);
    out[0] = r0 * cs0; out[1] = r0 * s0; out[2] = r1 * cs1; out[3] = r1 * s1;
}

template <bool REV>
__global__ __launch_bounds__(NTHREADS) __attribute__((amdgpu_waves_per_eu(NWAVES / 4, NWAVES / 4))) void hint_block_apply_kernel(
    KArgs a, ChainBlock one, const ChainBlock* __restrict__ chain, int n_chain,
    const float* __restrict__ x, const float* __restrict__ c, float* __restrict__ z,
    float* __restrict__ J, const float* __restrict__ J_in, float* __restrict__ loss_acc,
    float noise, const unsigned long long* __restrict__ rng_state, float* __restrict__ x_noisy) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float inv_d = frcp(a.d);
    STAMP_INIT()
    STAMP(0)
    HINT_LDS_TABLES()
    float* t0 = fbase;                        // two lane tiles: a fused permutation ping-pongs between them
    float* t1 = t0 + ROWS * a.xld;
    float* cs = t0 + 2 * ROWS * a.xld;
    float* vb = cs + ROWS * a.cld;
    float* a1 = vb + ROWS * a.vld;
    float* a2 = a1 + ROWS * a.ald;
    float* st = a2 + ROWS * a.ald;           // [s3][ROWS][sld]
    float* jac = st + a.s3 * ROWS * a.sld;
    const int sstride = ROWS * a.sld;
    const int ntiles = (a.B + ROWS - 1) / ROWS;
    const int fo = REV ? 1 : 0;               // which "first group" record of the kernel arguments
#define HINT_CB(I) chain_block(chain, one, I)
    GBlock blk = HINT_CB(0);
    {   // the first group's job lists and biases: issued together with the meta copy
        JobPrefetch jp0;
        jobs_issue<false>(jp0, a.jobs, a.first[fo][0], a.first[fo][1], (const float*)blk.packed + a.bias_off + a.first[fo][2], nullptr, a.first[fo][3], tid);
        meta_commit(mp_, a, mbase, tid);
        jobs_commit<false>(jp0, jbuf0, bias0, a.bmax, tid);
    }
    // The chain's fixed d x d permutation matrices, once per workgroup (when the launch found LDS for
    // them): read from global memory at every block boundary they cost an exposed L2 round trip there.
    float* ptab = lds + a.perm_lds;
    const int pdd = a.d * a.d;
    if (a.perm_lds > 0) {
        for (int i = tid; i < n_chain * pdd; i += NTHREADS) {
            const int cbi = fdiv(i, frcp(pdd));
            const float* pp = (chain != nullptr) ? chain[cbi].perm : one.perm;
            ptab[i] = pp != nullptr ? ((const GLOBAL_AS float*)pp)[i - cbi * pdd] : 0.f;
        }
    }

    // One Stage object travels through all GEMM stages: every stage_run() leaves it primed for
    // the stage that follows (first job decoded, its first weights in flight) - also across
    // groups, blocks of a chain and row tiles.
    int jb = 0;
    Stage S;
    bool first = true;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int row0 = tile * ROWS;
        const bool more_tiles = tile + (int)gridDim.x < ntiles;
        // the current lane tile is t0 + xcur (an integer offset, not a swapped pointer: the
        // compiler must keep seeing LDS addresses or it falls back to flat loads)
        int xcur = 0;
        const int xflip = ROWS * a.xld;
        (void)t1;
#define xs (t0 + xcur)
#define xo (t0 + (xflip - xcur))
        load_tile(xs, a.xld, x, a.d, row0, a.B, tid);
        if (a.dc > 0) load_tile(cs, a.cld, c, a.dc, row0, a.B, tid);
        if (tid < ROWS) jac[tid] = 0.f;
        __syncthreads();                      // meta, first job lists and the lane tile visible
        if (!REV && rng_state != nullptr) {
            // x += noise * N(0,1), four values per Philox call, keyed by (seed, step, element group)
            const unsigned long long seed = rng_state[0], step = rng_state[1];
            const int nvalid = (a.B - row0 < ROWS ? a.B - row0 : ROWS) * a.d;
            for (int q = tid; 4 * q < nvalid; q += NTHREADS) {
                float nz[4];
                philox_normal4(seed, step, (unsigned)(((size_t)row0 * a.d) / 4 + (size_t)q), nz);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = 4 * q + e;
                    if (i < nvalid) { const int r = fdiv(i, inv_d); xs[r * a.xld + (i - r * a.d)] += noise * nz[e]; }
                }
            }
            __syncthreads();
            if (x_noisy != nullptr) store_tile(x_noisy, xs, a.xld, a.d, row0, a.B, tid);   // what the backward pass starts from
        }
        STAMP(1)

        // ---- the blocks of the chain (one for the plain per-block entry points): the lane tile
        //      stays in LDS from block to block ----
        for (int cb = 0; cb < n_chain; ++cb) {
            const GBlock nblk = HINT_CB(cb + 1 < n_chain ? cb + 1 : 0);   // whose weights get prefetched next
            const float* packed = (const float*)blk.packed;
            const float* perm = (const float*)blk.perm;
            float* tape = (float*)blk.tape;
            float* actA1 = (float*)blk.wsA1;
            const bool train = !REV && actA1 != nullptr;
            if (!REV && perm != nullptr) {
                // fused fixed inter-block permutation (power_hint_8.py:59-62): x' = x W
                if (a.perm_lds > 0) {
                    for (int i = tid; i < ROWS * a.d; i += NTHREADS) {
                        const int r = fdiv(i, inv_d), j = i - r * a.d;
                        xo[r * a.xld + j] = perm_dot(xs + r * a.xld, ptab + cb * pdd + j, a.d, a.d);
                    }
                    xcur = xflip - xcur;
                    lds_barrier();            // (no global load to wait for: the weight prefetch stays in flight)
                } else {
                    for (int i = tid; i < ROWS * a.d; i += NTHREADS) {
                        const int r = fdiv(i, inv_d), j = i - r * a.d;
                        xo[r * a.xld + j] = perm_dot(xs + r * a.xld, perm + j, a.d, a.d);
                    }
                    xcur = xflip - xcur;
                    __syncthreads();
                }
                if (tape != nullptr)      // the permuted input is what the backward pass starts from
                    store_tile(tape + (size_t)(a.n_levels - 1) * a.B * a.d, xs, a.xld, a.d, row0, a.B, tid);
            } else if (!REV && cb > 0 && tape != nullptr) {
                // inner block of a chain without a permutation: its input exists nowhere else
                store_tile(tape + (size_t)(a.n_levels - 1) * a.B * a.d, xs, a.xld, a.d, row0, a.B, tid);
            }
            GroupU g = load_group(groups + (REV ? a.n_groups - 1 : 0));
            if (first) stage_begin(S, stage_list(jbuf0 + jb * a.jmax, g.l1_off, wave), packed, lane);
            first = false;

            for (int gi = 0; gi < a.n_groups; ++gi) {
                const bool last_group = gi + 1 >= a.n_groups;
                const bool has_next = !last_group || (cb + 1 < n_chain) || more_tiles;
                const float* packed_n = last_group ? (const float*)nblk.packed : packed;     // weights of the next group
                const int gnext = last_group ? 0 : gi + 1;
                const GroupU gn = load_group(groups + (REV ? (a.n_groups - 1 - gnext) : gnext));
                JobPrefetch jp;
                jp.count = 0;
                jp.nbias4 = 0;
                if (has_next)
                    jobs_issue<false>(jp, a.jobs, gn.jl_begin, gn.jl_count, packed_n + a.bias_off + gn.bmap_begin, nullptr, 2 * gn.aw + gn.sw, tid);
                lds_jobs_t jl = jbuf0 + jb * a.jmax;
                LDS_AS GJob* jl_next = jbuf0 + (jb ^ 1) * a.jmax;
                const float* bias_g = bias0 + jb * 2 * a.bmax;     // [b1 | b2 | b3] of this group

                stage_build_v(a, vmap + g.vmap_begin, g.vw, xs, cs, vb, tid);
                STAMP(2 + 12 * gi)
                lds_barrier();
                STAMP(3 + 12 * gi)
                STAMP_JOBS(gi < 3 ? 128 + (gi * 3 + 0) * 12 : 0)
                stage_run<EPI_RELU>(S, stage_list(jl, g.l2_off, wave), packed, packed, bias_g, vb, a.vld, a1, nullptr, a.ald, 0, lane);
                if (has_next) jobs_commit<false>(jp, jl_next, bias0 + (jb ^ 1) * 2 * a.bmax, a.bmax, tid);
                STAMP(4 + 12 * gi)
                lds_barrier();
                STAMP(5 + 12 * gi)
                STAMP_JOBS(gi < 3 ? 128 + (gi * 3 + 1) * 12 : 0)
                stage_run<EPI_RELU>(S, stage_list(jl, g.l3_off, wave), packed, packed, bias_g + g.aw, a1, a.ald, a2, nullptr, a.ald, 0, lane);
                STAMP(6 + 12 * gi)
                lds_barrier();
                STAMP(7 + 12 * gi)
                // training: both hidden activations go to the tape, [Bp][WT] row-major - the backward pass
                // reloads them instead of recomputing, and part B reads a1 from there.  Both in this phase
                // (a1 is still intact): stores in front of a stage's weight loads hold up the in-order vmcnt
                // waits of its k-loops, and the third layer's are the shortest (measured: -7 us per step
                // against storing a1 in the second layer's phase, -9 us against the coupling phase).
                // (Leaving all tape stores to one wavefront that takes no GEMM jobs was measured too:
                // +33 us - seven wavefronts balance the tiles worse, and nothing was gained back.)
                if (train) {
                    copy_rows_out<true>(actA1, a.WT, g.wcol0, a1, a.ald, g.aw, row0, tid);
                    copy_rows_out<true>(actA1 + a.act_stride, a.WT, g.wcol0, a2, a.ald, g.aw, row0, tid);
                }
                // the stage after this one: L1 of the next group (next block, next row tile); nothing
                // follows the very last one, which re-primes its own group's L1 (never run)
                STAMP_JOBS(gi < 3 ? 128 + (gi * 3 + 2) * 12 : 0)
                stage_run<EPI_LINEAR>(S, has_next ? stage_list(jl_next, gn.l1_off, wave) : stage_list(jl, g.l1_off, wave),
                                      packed, has_next ? packed_n : packed, bias_g + 2 * g.aw, a2, a.ald, st, nullptr, a.sld, sstride, lane);
                STAMP(8 + 12 * gi)
                lds_barrier();
                STAMP(9 + 12 * gi)
                {   // element-wise affine coupling + log-det partial sums (hint.py:79-83)
                    const int sub = tid & 15, row = tid >> 4;
                    float part = 0.f;
                    // (a split node couples in the unit that comes second: s unit forward, t unit inverse)
                    const bool couple_here = g.cont == 0 || g.cont == (REV ? 1 : 2);
#ifdef HINT_SKIP_COUPLE
                    if (false) {
#else
                    if (row < ROWS && couple_here) {
#endif
                        for (int e = sub; e < g.ent_cnt; e += 16) {
                            const EntU en = load_ent(ents, g.ent_begin + e);
                            float s = 0.f, t = 0.f;
                            for (int sl = 0; sl < g.l3_slabs; ++sl) {
                                s += st[sl * sstride + row * a.sld + en.scol];
                                t += st[sl * sstride + row * a.sld + en.tcol];
                            }
                            const float aa = a.alpha * atanf(s);
                            float* px = xs + row * a.xld + en.xcol;
                            // training: s goes to the tape ([n_levels + level][B][d], indexed by the lane it
                            // scales), so that the backward pass needs no third-layer recompute
                            if (!REV && tape != nullptr && row0 + row < a.B)
                                tape[((size_t)(a.n_levels + g.level) * a.B + row0 + row) * a.d + en.xcol] = s;
                            if (!REV) { *px = expf(aa) * (*px) + t; part += aa; }
                            else      { *px = ((*px) - t) / expf(aa); part -= aa; }
                        }
                    }
                    part = row16_sum(part);
                    if (sub == 0 && row < ROWS) jac[row] += part;
                }
                STAMP(10 + 12 * gi)
                lds_barrier();
                STAMP(11 + 12 * gi)
                // training: keep the lane tile as it stands after each level except the root's, so
                // that the backward pass sees bit-identical subnet inputs (tape[level][B][d])
                if (!REV && tape != nullptr && g.level_last && g.level < a.n_levels - 1)
                    store_tile(tape + (size_t)g.level * a.B * a.d, xs, a.xld, a.d, row0, a.B, tid);
                jb ^= 1;
                g = gn;
            }
            if (REV && perm != nullptr) {     // inverse of the fused permutation: x = x' W^T
                if (a.perm_lds > 0) {
                    for (int i = tid; i < ROWS * a.d; i += NTHREADS) {
                        const int r = fdiv(i, inv_d), j = i - r * a.d;
                        xo[r * a.xld + j] = perm_dot(xs + r * a.xld, ptab + cb * pdd + j * a.d, 1, a.d);
                    }
                    xcur = xflip - xcur;
                    lds_barrier();
                } else {
                    for (int i = tid; i < ROWS * a.d; i += NTHREADS) {
                        const int r = fdiv(i, inv_d), j = i - r * a.d;
                        xo[r * a.xld + j] = perm_dot(xs + r * a.xld, perm + (size_t)j * a.d, 1, a.d);
                    }
                    xcur = xflip - xcur;
                    __syncthreads();
                }
            }
            blk = nblk;
        }
        store_tile(z, xs, a.xld, a.d, row0, a.B, tid);
        if (tid < ROWS && row0 + tid < a.B) J[row0 + tid] = jac[tid] + (J_in != nullptr ? J_in[row0 + tid] : 0.f);
        if (loss_acc != nullptr) {
            // per-workgroup partial sums of the two loss terms (train_unconditional.py:128-129):
            // slot[0] += sum_rows 0.5*|z|^2, slot[1] += sum_rows J_total
            float zz = 0.f;
            const int nvalid = (a.B - row0 < ROWS ? a.B - row0 : ROWS);
            for (int i = tid; i < nvalid * a.d; i += NTHREADS) { const int r = fdiv(i, inv_d); const float v = xs[r * a.xld + (i - r * a.d)]; zz += v * v; }
            for (int o = 32; o > 0; o >>= 1) zz += __shfl_xor(zz, o, 64);
            if (lane == 0) vb[wave] = zz;      // vb is free at this point
            __syncthreads();
            if (tid == 0) {
                float t = 0.f;
                for (int w = 0; w < NWAVES; ++w) t += vb[w];
                float js = 0.f;
                for (int r = 0; r < nvalid; ++r) js += jac[r] + (J_in != nullptr ? J_in[row0 + r] : 0.f);
                // 64 slots of {sum 0.5|z|^2, sum J}: spreads the atomics of the 256 workgroups
                float* slot = loss_acc + 2 * (blockIdx.x & 63);
                atomicAdd(slot, 0.5f * t);
                atomicAdd(slot + 1, js);
            }
        }
        STAMP(120)
        lds_barrier();
#undef xs
#undef xo
    }
    STAMP_FLUSH()
#undef HINT_CB
}

// =======================================================================================
// backward, part A (row parallel): walk the levels root first; per level reload the lane tile
// the forward pass recorded (x for the deepest level, tape[level-1] otherwise), recompute each
// node's activations from it (same code, same inputs -> bit-identical ReLU masks), and
// back-propagate through coupling and subnets to get g_x / g_c.  The thin weight gradients
// (dW1, dW3, all biases: O(h) floats per node) are reduced over the 16 rows here and added to
// the flat gradient buffer with float atomics; only a1 and g2, the operands of the h x h
// gradient dW2 = g2^T a1, go to the workspace for part B.
//   g_t = g_l' ; g_a = g_l'*exp(a)*l + g_J ; g_l = g_l'*exp(a) ; g_s = g_a*alpha/(1+s^2)
// =======================================================================================
// The reverse: 16 rows x `width` columns of a [rows][sld] global array into an LDS buffer, issue
// and commit split like TilePrefetch.  6 float4 per thread cover width <= 768 (two nets of h <= 384).
// RR float4 per thread cover width <= 128*RR columns; the backward kernel is instantiated for
// RR = 3, 4, 6 (widest group <= 384, 512, 768 columns) because these registers are live across
// GEMM stages and every spare one there costs scratch traffic.
template <int RR> struct RowsPrefetch { f32x4 r[RR]; };
template <int RR>
__device__ __forceinline__ void rows_issue(RowsPrefetch<RR>& rp, const float* __restrict__ src, int sld, int scol,
                                           int width, int row0, int tid) {
    const int w4 = width >> 2, n4 = ROWS * w4;
    const float inv = frcp(w4);
#pragma unroll
    for (int k = 0; k < RR; ++k) {
        if (k * NTHREADS < n4) {                       // wave-uniform
            const int i = min(tid + k * NTHREADS, n4 - 1);
            const int r = fdiv(i, inv), j = (i - r * w4) << 2;
            rp.r[k] = *(const f32x4*)(src + (size_t)(row0 + r) * sld + scol + j);
        }
    }
}
template <int RR>
__device__ __forceinline__ void rows_commit(const RowsPrefetch<RR>& rp, float* dst, int dld, int width, int tid) {
    const int w4 = width >> 2, n4 = ROWS * w4;
    const float inv = frcp(w4);
#pragma unroll
    for (int k = 0; k < RR; ++k) {
        const int i = tid + k * NTHREADS;
        if (i < n4) {
            const int r = fdiv(i, inv), j = (i - r * w4) << 2;
            *(f32x4*)(dst + r * dld + j) = rp.r[k];
        }
    }
}

template <int RR>
__global__ __launch_bounds__(NTHREADS) __attribute__((amdgpu_waves_per_eu(NWAVES / 4, NWAVES / 4))) void hint_block_bwd_kernel(
    KArgs a, ChainBlock one, const ChainBlock* __restrict__ chain, int n_chain,
    const float* __restrict__ x, const float* __restrict__ c,
    const float* __restrict__ g_z, const float* __restrict__ g_J, float* __restrict__ g_x,
    float* __restrict__ g_c, float gz_scale, float gJ_const) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float inv_d = frcp(a.d);
    STAMP_INIT()
    STAMP(0)
    HINT_LDS_TABLES()
    float* xs = fbase;
    float* gs = xs + ROWS * a.xld;
    float* cs = gs + ROWS * a.xld;
    float* gcs = cs + ROWS * a.cld;
    float* vb = gcs + ROWS * a.cld;
    float* gv = vb + ROWS * a.vld;           // [sv][ROWS][vld]
    float* a1 = gv + a.sv * ROWS * a.vld;
    float* a2 = a1 + ROWS * a.ald;
    // g2 gets its own buffer so that the dW3 tiles (which read a2) can run in the g2 stage's phase;
    // plans that cannot afford it (a.split_o3) write g2 over a2 and run dW3 as a phase of its own
    float* a3 = a2 + (a.split_o3 ? 0 : ROWS * a.ald);
    float* sb = a3 + ROWS * a.ald;           // [ROWS][xld]: s of the group's level as the forward pass computed it (tape)
    float* gst = sb + ROWS * a.xld;
    float* gj = gst + ROWS * a.sld;
    const int vstride = ROWS * a.vld;
    const int ntiles = (a.B + ROWS - 1) / ROWS;
#define HINT_CB(I) chain_block(chain, one, I)
    GBlock blk = HINT_CB(n_chain - 1);         // the chain is walked from its last block to its first
    {
        JobPrefetch jp0;
        jobs_issue<true>(jp0, a.jobs, a.first[1][0], a.first[1][1], (const float*)blk.packed + a.bias_off + a.first[1][2], a.bmap + a.first[1][2], a.first[1][3], tid);
        meta_commit(mp_, a, mbase, tid);
        jobs_commit<true>(jp0, jbuf0, bias0, a.bmax, tid);
    }

    int jb = 0;
    Stage S;              // walks the GEMM stages g2, g1, dv of every group (see forward kernel)
    bool first_tile = true;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int row0 = tile * ROWS;
        const bool more_tiles = tile + (int)gridDim.x < ntiles;
        load_tile(gs, a.xld, g_z, a.d, row0, a.B, tid);
        if (gz_scale != 1.f)                   // loss gradient fused: g_z = z / B given z
            for (int i = tid; i < ROWS * a.d; i += NTHREADS) { const int r = fdiv(i, inv_d); gs[r * a.xld + (i - r * a.d)] *= gz_scale; }
        if (a.dc > 0) {
            load_tile(cs, a.cld, c, a.dc, row0, a.B, tid);
            load_tile(gcs, a.cld, nullptr, a.dc, row0, a.B, tid);
        }
        if (tid < ROWS) gj[tid] = (row0 + tid < a.B) ? (g_J != nullptr ? g_J[row0 + tid] : gJ_const) : 0.f;
        // lane tile of the first (root) level: same memory round trip as everything above.  The
        // input of a block is x for the first block of a call without a fused permutation, and the
        // top tape slot otherwise (hint_block_apply_kernel stored it there).
#define BLOCK_LEVEL_SRC(TAPE, TOP, LV) ((LV) == 0 ? ((TOP) ? (TAPE) + (size_t)(a.n_levels - 1) * a.B * a.d : x) \
                                                  : (TAPE) + (size_t)((LV) - 1) * a.B * a.d)
#define LEVEL_SRC(LV) BLOCK_LEVEL_SRC(tape, (perm != nullptr || cb > 0), LV)
        {
            const float* tape = (const float*)blk.tape;
            const float* perm = (const float*)blk.perm;
            const int cb = n_chain - 1;
            load_tile(xs, a.xld, LEVEL_SRC(a.n_levels - 1), a.d, row0, a.B, tid);
        }
        __syncthreads();
        STAMP(1)
        // v = [upper lanes | c] of a group's nodes and a cleared g_st: for the first group here,
        // for every later one in the scatter phase of its predecessor
#define HINT_BUILD_V(G)                                                              \
        {                                                                            \
            stage_build_v(a, vmap + (G).vmap_begin, (G).vw, xs, cs, vb, tid);        \
            for (int i = tid; i < ROWS * (G).sw && (G).cont != 1; i += NTHREADS) {   \
                const int r = fdiv(i, frcp((G).sw));                                           \
                gst[r * a.sld + (i - r * (G).sw)] = 0.f;                             \
            }                                                                        \
        }
        // s of a level and the group's hidden activations a2 come from the tape: fetched one group
        // ahead (here: for the first group), committed to LDS at the top of the group
        TilePrefetch stile;
        RowsPrefetch<RR> a2t;
#define HINT_FIRST_DESC(G) (a.split_o3 ? (G).o3_off : (G).g2_off)     /* the first GEMM stage of a group */
        {
            const GroupU g0 = load_group(groups + (a.n_groups - 1));
            HINT_BUILD_V(g0)
            tile_issue(stile, (const float*)blk.tape + (size_t)(a.n_levels + g0.level) * a.B * a.d, a.d, row0, a.B, tid);
            rows_issue(a2t, (const float*)blk.wsA1 + a.act_stride, a.WT, g0.wcol0, g0.aw, row0, tid);
        }
        lds_barrier();
      for (int cb = n_chain - 1; cb >= 0; --cb) {
        const GBlock nblk = HINT_CB(cb > 0 ? cb - 1 : n_chain - 1);    // the block worked on after this one
        const float* packed = (const float*)blk.packed;
        const float* perm = (const float*)blk.perm;
        const float* tape = (const float*)blk.tape;
        const float* actA1 = (const float*)blk.wsA1;
        float* wsG2 = (float*)blk.wsG2;
        float* gparams = (float*)blk.wsT + (size_t)tile * a.thin_total;   // this row tile's thin-gradient slab
        float* dump = (float*)blk.wsT + (size_t)ntiles * a.thin_total + (size_t)tile * 64;   // this row tile's 64 floats behind the slabs
        GroupU g = load_group(groups + (a.n_groups - 1));
        if (first_tile) stage_begin(S, stage_list(jbuf0 + jb * a.jmax, HINT_FIRST_DESC(g), wave), packed, lane);
        first_tile = false;

        for (int gi = a.n_groups - 1; gi >= 0; --gi) {
            const bool has_next = (gi > 0) || (cb > 0) || more_tiles;
            const float* packed_n = (gi > 0) ? packed : (const float*)nblk.packed;   // weights of the next group
            const GroupU gn = load_group(groups + (gi > 0 ? gi - 1 : a.n_groups - 1));
            JobPrefetch jp;
            jp.count = 0;
            jp.nbias4 = 0;
            if (has_next)
                jobs_issue<true>(jp, a.jobs, gn.jl_begin, gn.jl_count, packed_n + a.bias_off + gn.bmap_begin, a.bmap + gn.bmap_begin, 2 * gn.aw + gn.sw, tid);
            lds_jobs_t jl = jbuf0 + jb * a.jmax;
            LDS_AS GJob* jl_next = jbuf0 + (jb ^ 1) * a.jmax;
            const float* bias_g = bias0 + jb * 2 * a.bmax;     // [b1 | b2 | b3] of this group, then their gradient map
            const int32_t* bmap_g = (const int32_t*)(bias_g + a.bmax);
            const int sbase = 2 + 20 * (a.n_groups - 1 - gi);
            (void)sbase;

            // ---- the lanes as the forward pass saw them when it entered the NEXT level: fetched
            //      into registers now, committed to LDS once this group no longer reads xs ----
            const bool block_switch = (gi == 0) && (cb > 0);       // next: the root level of the block before
            const bool level_switch = ((gi > 0) && (gn.level != g.level)) || block_switch;
            TilePrefetch xnext;
            if (level_switch) {
                const float* src = block_switch
                    ? BLOCK_LEVEL_SRC((const float*)nblk.tape, (nblk.perm != nullptr || cb > 1), a.n_levels - 1)
                    : LEVEL_SRC(gn.level);
                tile_issue(xnext, src, a.d, row0, a.B, tid);
            }
            // ---- nothing is recomputed: s and a2 of the group (bit-identical to the forward's) land
            //      in LDS now, a1 during the g2 stage ----
            tile_commit(stile, sb, a.xld, a.d, tid);
            rows_commit(a2t, a2, a.ald, g.aw, tid);
            RowsPrefetch<RR> a1t;                                         // a1: fetched across the coupling and the g2 stage
#ifndef HINT_VAR_LATE_A1
            rows_issue(a1t, actA1, a.WT, g.wcol0, g.aw, row0, tid);
#endif
            STAMP(sbase + 4)
            lds_barrier();
            STAMP(sbase + 5)
            {   // ---- coupling backward (a split node: in its s unit, the first of the two here) ----
                const int sub = tid & 15, row = tid >> 4;
                if (row < ROWS && g.cont != 1) {
                    const float gJr = gj[row];
                    for (int e = sub; e < g.ent_cnt; e += 16) {
                        const EntU en = load_ent(ents, g.ent_begin + e);
                        const float s = sb[row * a.xld + en.xcol];
                        const float aa = a.alpha * atanf(s);
                        const float ea = expf(aa);
                        const float l = xs[row * a.xld + en.xcol];       // lower input of the node
                        float* pg = gs + row * a.xld + en.xcol;
                        const float glp = *pg;                            // grad wrt l' = exp(a)*l + t
                        *pg = glp * ea;                                   // g_l
                        const float ga = glp * ea * l + gJr;              // g_a (a feeds both l' and J)
                        gst[row * a.sld + en.scol] = ga * a.alpha / (1.f + s * s);   // g_s
                        gst[row * a.sld + en.tcol] = glp;                              // g_t
                    }
                }
            }
            if (has_next) jobs_commit<true>(jp, jl_next, bias0 + (jb ^ 1) * 2 * a.bmax, a.bmax, tid);
            STAMP(sbase + 8)
            lds_barrier();
            STAMP(sbase + 9)
            if (level_switch) tile_commit(xnext, xs, a.xld, a.d, tid);   // xs is not read again in this group
            // ---- g2 = (g_st * W3) .* relu'(a2) -> a3;  dW3 += g_st^T a2 (outer-product tiles in the
            //      same lists);  db3 += colsum(g_st) ----
            colsum_store(bmap_g + 2 * g.aw, g.sw, gst, a.sld, gparams, tid);
            if (a.split_o3) {                  // outer-product tiles only; then g2 may overwrite a2
                STAMP_JOBS(gi < 3 ? 128 + (gi * 3 + 0) * 12 : 0)
                stage_run<EPI_PLAIN, true>(S, stage_list(jl, g.g2_off, wave), packed, packed, bias_g, gst, a.sld, a3, nullptr,
                                           a.ald, 0, lane, gst, a.sld, a2, a.ald, gparams, dump);
                lds_barrier();
            }
            STAMP_JOBS(gi < 3 ? 128 + (gi * 3 + 0) * 12 : 0)
            stage_run<EPI_MASK, true>(S, stage_list(jl, g.g1_off, wave), packed, packed, bias_g, gst, a.sld, a3, a2,
                                      a.ald, 0, lane, gst, a.sld, a2, a.ald, gparams, dump);
#ifdef HINT_VAR_LATE_A1
            STAMP(104 + 4 * (a.n_groups - 1 - gi))
            rows_issue(a1t, actA1, a.WT, g.wcol0, g.aw, row0, tid);
#endif
            rows_commit(a1t, a1, a.ald, g.aw, tid);       // (a1 has been free since the dv stage of the group before)
            STAMP(sbase + 12)
            lds_barrier();
            STAMP(sbase + 13)
            // ---- g1 = (g2 * W2) .* relu'(a1), in place over a1;  db2 += colsum(g2) ----
            copy_rows_out<false>(wsG2, a.WT, g.wcol0, a3, a.ald, g.aw, row0, tid);
            STAMP(105 + 4 * (a.n_groups - 1 - gi))
            colsum_store(bmap_g + g.aw, g.aw, a3, a.ald, gparams, tid);
            STAMP(106 + 4 * (a.n_groups - 1 - gi))
            STAMP_JOBS(gi < 3 ? 128 + (gi * 3 + 1) * 12 : 0)
            stage_run<EPI_MASK>(S, stage_list(jl, g.dv_off, wave), packed, packed, bias_g, a3, a.ald, a1, a1, a.ald, 0, lane);
            STAMP(sbase + 14)
            lds_barrier();
            STAMP(sbase + 15)
            // ---- g_v = [g1_s | g1_t] * [W1_s ; W1_t];  dW1 += g1^T v (outer-product tiles);
            //      db1 += colsum(g1) ----
            colsum_store(bmap_g, g.aw, a1, a.ald, gparams, tid);
            if ((gi > 0) || (cb > 0)) {
                const float* tape_n = block_switch ? (const float*)nblk.tape : tape;
                const float* act_n = block_switch ? (const float*)nblk.wsA1 : actA1;
                tile_issue(stile, tape_n + (size_t)(a.n_levels + gn.level) * a.B * a.d, a.d, row0, a.B, tid);
                rows_issue(a2t, act_n + a.act_stride, a.WT, gn.wcol0, gn.aw, row0, tid);
            }
            STAMP_JOBS(gi < 3 ? 128 + (gi * 3 + 2) * 12 : 0)
            stage_run<EPI_PLAIN, true>(S, has_next ? stage_list(jl_next, HINT_FIRST_DESC(gn), wave)
                                                   : stage_list(jl, HINT_FIRST_DESC(g), wave),
                                       packed, has_next ? packed_n : packed, bias_g, a1, a.ald, gv, nullptr, a.vld, vstride,
                                       lane, a1, a.ald, vb, a.vld, gparams, dump);
            STAMP(sbase + 16)
            lds_barrier();
            STAMP(sbase + 17)
            // ---- scatter g_v: upper-lane columns to g (each lane has one v column per group),
            //      condition columns to g_c (every node of the group contributes) ----
            {
                const LDS_AS int16_t* vm = vmap + g.vmap_begin;
                for (int i = tid; i < ROWS * g.vw; i += NTHREADS) {
                    const int r = fdiv(i, frcp(g.vw)), j = i - r * g.vw;
                    const int m = vm[j];
                    if (m >= 0) {
                        float acc = gs[r * a.xld + m];
                        for (int sl = 0; sl < g.dv_slabs; ++sl) acc += gv[sl * vstride + r * a.vld + j];
                        gs[r * a.xld + m] = acc;
                    }
                }
                if (a.dc > 0) {
                    for (int i = tid; i < ROWS * a.dc; i += NTHREADS) {
                        const int r = fdiv(i, frcp(a.dc)), cc = i - r * a.dc;
                        float acc = gcs[r * a.cld + cc];
                        for (int j = 0; j < g.vw; ++j)
                            if (vm[j] == -2 - cc)
                                for (int sl = 0; sl < g.dv_slabs; ++sl) acc += gv[sl * vstride + r * a.vld + j];
                        gcs[r * a.cld + cc] = acc;
                    }
                }
            }
            if ((gi > 0) || (cb > 0)) {
                HINT_BUILD_V(gn)                // xs holds the next group's level since the g2 phase
            }
            STAMP(sbase + 18)
            lds_barrier();
            STAMP(sbase + 19)
            jb ^= 1;
            g = gn;
        }
        if (perm != nullptr) {                 // chain rule through x' = x W:  g_x = g_x' W^T
            // d <= 128 (plan check): at most four elements per thread, held in registers across
            // the barrier so that the product can go back into gs
            float pacc[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = tid + q * NTHREADS;
                float acc = 0.f;
                if (i < ROWS * a.d) {
                    const int r = fdiv(i, inv_d), j = i - r * a.d;
                    acc = perm_dot(gs + r * a.xld, perm + (size_t)j * a.d, 1, a.d);
                }
                pacc[q] = acc;
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = tid + q * NTHREADS;
                if (i < ROWS * a.d) { const int r = fdiv(i, inv_d); gs[r * a.xld + (i - r * a.d)] = pacc[q]; }
            }
            __syncthreads();
        }
        blk = nblk;
      }
        store_tile(g_x, gs, a.xld, a.d, row0, a.B, tid);
        if (a.dc > 0 && g_c != nullptr) store_tile(g_c, gcs, a.cld, a.dc, row0, a.B, tid);
        STAMP(120)
        lds_barrier();
    }
    STAMP_FLUSH()
#undef HINT_CB
}

// =======================================================================================
// backward, part B: dW2[m][n] = sum_b G2[b][col+m] * A1[b][col+n] for every (node, net): a
// GEMM whose reduction runs over the batch, so the OUTPUT is tiled (48x48 per workgroup) and
// the batch is split over `splits` workgroups and the 8 wavefronts of each; wavefront partials
// are combined in LDS, workgroup partials with float atomics.  Block ids are mapped so that
// all tiles of one batch split run on the same XCD (blocks b, b+8, .. share an XCD): the
// G2/A1 rows of a split are fetched from HBM / Infinity Cache once and re-read from that
// XCD's L2 by the tiles that share them.
// =======================================================================================
constexpr int DW_WAVES = 8;

__global__ __launch_bounds__(DW_WAVES * 64) void hint_block_dw_kernel(
    const DWJob* __restrict__ jobs, int n_jobs, int splits, ChainBlock one, const ChainBlock* __restrict__ chain,
    int grid_pb, int grid_used, int WT, int Bp, int rows_per_wg,
    const int32_t* __restrict__ tmap, int thin_total, int ntiles, int tsplit) {
    __shared__ float red[DW_WAVES][9][64][4];   // 72 KiB

    // a chained launch holds grid_pb (a multiple of 8, so that the XCD mapping below holds for
    // every block) workgroups per block of the chain, of which the first grid_used have work
    const int cbi = (int)blockIdx.x / grid_pb;
    const int bid = (int)blockIdx.x - cbi * grid_pb;
    if (bid >= grid_used) return;
    const GBlock blk = chain_block(chain, one, cbi);
    const float* wsA1 = (const float*)blk.wsA1;
    const float* wsG2 = (const float*)blk.wsG2;
    const float* wsT = (const float*)blk.wsT;
    float* gparams = (float*)blk.gparams;

    const int n_dw_blocks = n_jobs * splits;
    if (bid >= n_dw_blocks) {
        // ---- thin gradients (dW1, dW3, biases): sum the per-row-tile slabs part A wrote ----
        const int id = bid - n_dw_blocks;
        const int chunk = id / tsplit, part = id - chunk * tsplit;
        const int idx = chunk * (DW_WAVES * 64) + (int)threadIdx.x;
        if (idx >= thin_total) return;
        const int per = (ntiles + tsplit - 1) / tsplit;
        const int t0 = part * per, t1 = min(ntiles, t0 + per);
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int t = t0;
        for (; t + 4 <= t1; t += 4) {
            s0 += wsT[(size_t)(t + 0) * thin_total + idx];
            s1 += wsT[(size_t)(t + 1) * thin_total + idx];
            s2 += wsT[(size_t)(t + 2) * thin_total + idx];
            s3 += wsT[(size_t)(t + 3) * thin_total + idx];
        }
        for (; t < t1; ++t) s0 += wsT[(size_t)t * thin_total + idx];
        if (t1 > t0) atomicAdd(gparams + tmap[idx], (s0 + s1) + (s2 + s3));
        return;
    }
    int jidx, split;
    {
        const int id = bid;
        if ((splits & 7) == 0) {          // XCD-aware: split s lives on XCD s % 8
            const int xcd = id & 7, t = id >> 3;
            split = xcd + 8 * (t / n_jobs);
            jidx = t % n_jobs;
        } else {
            split = id / n_jobs;
            jidx = id % n_jobs;
        }
    }
    const DWJob job = jobs[jidx];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nl = lane & 15, kq = lane >> 4;

    const int ntm = job.mw, ntn = job.nw;
    const int b_begin = split * rows_per_wg;
    const int b_end = min(Bp, b_begin + rows_per_wg);

    f32x4 acc[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // one 12-byte load per operand and k-step: lane nl holds columns m0 + mw*nl + {0,1,2}
    // (a narrower group over-reads into the following columns; those products are never used)
    typedef float f32x3u __attribute__((ext_vector_type(3), aligned(4)));
    const float* gp = wsG2 + job.col + job.m0 + ntm * nl;
    const float* xp = wsA1 + job.col + job.n0 + ntn * nl;

    // double-buffered over 16-row blocks: the loads of block j+1 are in flight during the MFMAs
    // of block j (issuing four blocks up front measured slower: no load/MFMA overlap)
    f32x3u av[2][4], bv[2][4];
#define DW_LOAD(BUF, BB)                                                     \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                           \
        const size_t row_ = (size_t)((BB) + 4 * i + kq) * WT;                 \
        av[BUF][i] = *(const f32x3u*)(gp + row_);                             \
        bv[BUF][i] = *(const f32x3u*)(xp + row_);                             \
    }
#define DW_MMA(BUF)                                                           \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                             \
        _Pragma("unroll") for (int tm = 0; tm < 3; ++tm)                      \
            if (tm < ntm)                                                     \
                _Pragma("unroll") for (int tn = 0; tn < 3; ++tn)              \
                    if (tn < ntn) acc[tm][tn] = mfma4(av[BUF][i][tm], bv[BUF][i][tn], acc[tm][tn]);

    const int step = 16 * DW_WAVES;
    int bb = b_begin + wave * 16;
    if (bb < b_end) {
        DW_LOAD(0, bb)
        while (true) {
            const int nb1 = bb + step;
            const int l1 = nb1 < b_end ? nb1 : bb;      // clamp: re-load the same rows at the end
            DW_LOAD(1, l1)
            DW_MMA(0)
            if (nb1 >= b_end) break;
            const int nb2 = nb1 + step;
            const int l2 = nb2 < b_end ? nb2 : nb1;
            DW_LOAD(0, l2)
            DW_MMA(1)
            if (nb2 >= b_end) break;
            bb = nb2;
        }
    }
#undef DW_LOAD
#undef DW_MMA
    // combine the wavefronts
#pragma unroll
    for (int tm = 0; tm < 3; ++tm)
#pragma unroll
        for (int tn = 0; tn < 3; ++tn) *(f32x4*)&red[wave][tm * 3 + tn][lane][0] = acc[tm][tn];
    __syncthreads();
    for (int idx = tid; idx < 9 * 64; idx += DW_WAVES * 64) {
        const int t = idx >> 6, l = idx & 63;
        const int tm = t / 3, tn = t - 3 * tm;
        if (tm >= ntm || tn >= ntn) continue;
        f32x4 v = *(f32x4*)&red[0][t][l][0];
#pragma unroll
        for (int w = 1; w < DW_WAVES; ++w) v += *(f32x4*)&red[w][t][l][0];
        const int n = job.n0 + ntn * (l & 15) + tn;          // undo the column permutation of the loads
        if (n >= job.H) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = job.m0 + ntm * (4 * (l >> 4) + i) + tm;
            if (m < job.H) atomicAdd(gparams + job.wofs + (size_t)m * job.H + n, v[i]);
        }
    }
}

// ---- launchers (called from hint_plan.cpp) ----------------------------------------------
namespace hint {

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

hipError_t launch_apply(bool rev, const KArgs& a, int lds_bytes, int grid, const ChainBlock& one,
                        const ChainBlock* chain, int n_chain, const float* x, const float* c, float* z, float* J,
                        const float* J_in, float* loss_acc, float noise, const unsigned long long* rng_state,
                        float* x_noisy, hipStream_t stream) {
    if (rev)
        hipLaunchKernelGGL(hint_block_apply_kernel<true>, dim3(grid), dim3(NTHREADS), lds_bytes, stream, a, one,
                           chain, n_chain, x, c, z, J, J_in, (float*)nullptr, 0.f, (const unsigned long long*)nullptr,
                           (float*)nullptr);
    else
        hipLaunchKernelGGL(hint_block_apply_kernel<false>, dim3(grid), dim3(NTHREADS), lds_bytes, stream, a, one,
                           chain, n_chain, x, c, z, J, J_in, loss_acc, noise, rng_state, x_noisy);
    return hipGetLastError();
}

hipError_t launch_bwd(const KArgs& a, int lds_bytes, int grid, const ChainBlock& one, const ChainBlock* chain,
                      int n_chain, const float* x, const float* c, const float* g_z, const float* g_J,
                      float* g_x, float* g_c, float gz_scale, float gJ_const, hipStream_t stream) {
    const int max_aw = a.max_aw;       // widest group of the plan
#define HINT_LAUNCH_BWD(RR)                                                                                      \
    hipLaunchKernelGGL(hint_block_bwd_kernel<RR>, dim3(grid), dim3(NTHREADS), lds_bytes, stream, a, one, chain,   \
                       n_chain, x, c, g_z, g_J, g_x, g_c, gz_scale, gJ_const)
    if (max_aw <= 128 * 3) HINT_LAUNCH_BWD(3);
    else if (max_aw <= 128 * 4) HINT_LAUNCH_BWD(4);
    else HINT_LAUNCH_BWD(6);
#undef HINT_LAUNCH_BWD
    return hipGetLastError();
}

hipError_t launch_dw(const DWJob* jobs, int n_jobs, int splits, const ChainBlock& one, const ChainBlock* chain,
                     int n_chain, int WT, int Bp, int rows_per_wg, const int32_t* tmap, int thin_total, int ntiles,
                     hipStream_t stream) {
    // thin-gradient reduction rides in the same launch: enough parts that a thread sums <= 32 slabs
    int tsplit = (ntiles + 31) / 32;
    if (tsplit < 1) tsplit = 1;
    const int thin_blocks = ((thin_total + DW_WAVES * 64 - 1) / (DW_WAVES * 64)) * tsplit;
    const int used = n_jobs * splits + thin_blocks;
    const int grid_pb = n_chain > 1 ? (used + 7) / 8 * 8 : used;
    if (used > 0)
        hipLaunchKernelGGL(hint_block_dw_kernel, dim3(grid_pb * n_chain), dim3(DW_WAVES * 64), 0, stream, jobs,
                           n_jobs, splits, one, chain, grid_pb, used, WT, Bp, rows_per_wg, tmap, thin_total, ntiles,
                           tsplit);
    return hipGetLastError();
}

hipError_t set_stamp_buffer(unsigned long long* p) {
#ifdef HINT_STAMPS
    return hipMemcpyToSymbol(HIP_SYMBOL(g_hint_stamps), &p, sizeof(p));
#else
    (void)p;
    return hipErrorNotSupported;
#endif
}

hipError_t set_max_lds(int fwd_bytes, int bwd_bytes) {
    hipError_t e;
    e = hipFuncSetAttribute((const void*)hint_block_apply_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, fwd_bytes);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void*)hint_block_apply_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, fwd_bytes);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void*)hint_block_bwd_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, bwd_bytes);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void*)hint_block_bwd_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, bwd_bytes);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute((const void*)hint_block_bwd_kernel<6>, hipFuncAttributeMaxDynamicSharedMemorySize, bwd_bytes);
}

}  // namespace hint
