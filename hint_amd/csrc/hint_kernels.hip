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
// the block inside one launch; the lane tile, the conditioning input, both hidden
// activations and s/t live in LDS for the whole pass, HBM sees x once in and z, J once out.
// The eight wavefronts split the 16-wide output tiles of the s- and t-subnets of every node
// of a level; weights stream from L2 as pre-packed MFMA B-fragments (one coalesced 1 KiB load
// per 16x16 tile), one chunk of four tiles ahead of the MFMAs that consume them.  fp32 MFMA
// is an exact fp32 FMA chain, so results differ from the CPU reference only by summation
// order.
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
__device__ unsigned long long* g_hint_stamps = nullptr;
#define STAMP(ID)                                                                              \
    if (g_hint_stamps != nullptr && blockIdx.x == 0 && (threadIdx.x & 63) == 0) {               \
        unsigned long long t_;                                                                 \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");             \
        g_hint_stamps[(threadIdx.x >> 6) * 128 + (ID)] = t_;                                    \
    }
#else
#define STAMP(ID)
#endif

// Workgroup barrier that orders LDS traffic only.  __syncthreads() makes hipcc drain vmcnt(0)
// first, which would retire the packed-weight prefetch of the next stage at every stage
// boundary; here only LDS operations (lgkmcnt) are waited for, global loads stay in flight
// across the barrier (cdna_hip_programming.md §5 "Pipelining across barriers").  Global
// stores issued before it are never read back inside the kernel.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

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
__global__ __launch_bounds__(256) void hint_pack_kernel(const PackSeg* __restrict__ segs,
                                                        const int2* __restrict__ ptiles, int n_tiles,
                                                        const int32_t* __restrict__ bmap, int n_bias,
                                                        long bias_off, const float* __restrict__ P,
                                                        float* __restrict__ packed) {
    if ((int)blockIdx.x >= n_tiles) {
        // trailing workgroups: biases laid out per group in LDS column order (zero for padding)
        const int i = ((int)blockIdx.x - n_tiles) * 256 + (int)threadIdx.x;
        if (i < n_bias) { const int off = bmap[i]; packed[bias_off + i] = off >= 0 ? P[off] : 0.f; }
        return;
    }
    const int2 pt = ptiles[blockIdx.x];
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

__global__ __launch_bounds__(256) void hint_zero_kernel(float* __restrict__ p, long n4, long n) {
    const long stride = (long)gridDim.x * blockDim.x;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) ((f32x4*)p)[i] = z;
    const long t = n4 * 4 + (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) p[t] = 0.f;
}

// =======================================================================================
// Generic GEMM stage.  Every wavefront walks its jobs (16-column output tiles, wave-strided)
// as one stream of chunks of <= 4 k-blocks.  Three register sets rotate so that the packed B
// fragments of the next two chunks are in flight while the MFMAs of the current one issue
// (loads are unconditional so the compiler can count them exactly), across job boundaries and,
// through stage_begin(), across the workgroup barrier in front of the stage: weights do not
// depend on the previous stage, only the A operand in LDS does.
//   MFMA lane map (16x16x4 f32): lane l supplies A[m = l&15][kslot = l>>4] and
//   B[kslot][n = l&15]; result reg i = C[4*(l>>4)+i][l&15].  Slot kq of step i of a 16-wide
//   k-block is k = 16*kb + 4*kq + i on both operands, so each lane reads 4 consecutive k with
//   one 128-bit access (LDS for A, global for packed B).
// Job descriptors live in LDS (staged per group, one group ahead): a global or scalar load
// in front of every job would put an L2 round trip (~700 cycles here) on the critical path.
// =======================================================================================
enum { EPI_RELU = 0, EPI_LINEAR = 1, EPI_MASK = 2, EPI_PLAIN = 3 };

struct JobU { int wtile, acol, ocol, nblk, nvalid, slab; };

__device__ __forceinline__ JobU load_job(lds_jobs_t jl, int idx) {
    const i32x4 raw = *(const LDS_AS i32x4*)(jl + idx);    // one ds_read_b128 at a wave-uniform address
    JobU u;
    u.wtile = __builtin_amdgcn_readfirstlane(raw.x);
    const unsigned z = (unsigned)__builtin_amdgcn_readfirstlane(raw.z);
    const unsigned w = (unsigned)__builtin_amdgcn_readfirstlane(raw.w);
    u.acol = (int)(z & 0xffffu);
    u.ocol = (int)(z >> 16);
    u.nblk = (int)(w & 0xffu);
    u.nvalid = (int)((w >> 8) & 0xffu);
    u.slab = (int)((w >> 16) & 0xffu);
    return u;
}

struct Stage {
    lds_jobs_t jl;
    int njobs, rot;
    bool active;
    int pi, pkb;            // prefetch cursor: logical job index, k-block
    JobU pjb;
    int ci, ckb;            // compute cursor
    JobU cjb;
    f32x4 a0, a1, a2, a3, b0, b1, b2, b3;
};

#ifdef HINT_ABLATE_WLOAD    // diagnostic: every weight fetch hits the same 4 KiB (L1-resident)
#define HINT_WTILE(T) ((size_t)((T) & 3))
#else
#define HINT_WTILE(T) ((size_t)(T))
#endif
#define HINT_JIDX(S, I) (((I) + (S).rot) >= (S).njobs ? ((I) + (S).rot) - (S).njobs : ((I) + (S).rot))

#define HINT_FETCH(S, R0, R1, R2, R3)                                                              \
    {                                                                                              \
        const f32x4* wp_ = (const f32x4*)packed + HINT_WTILE((S).pjb.wtile) * 64 + lane;           \
        const int last_ = (S).pjb.nblk > 0 ? (S).pjb.nblk - 1 : 0;                                 \
        R0 = wp_[((S).pkb + 0 < last_ ? (S).pkb + 0 : last_) * 64];                                \
        R1 = wp_[((S).pkb + 1 < last_ ? (S).pkb + 1 : last_) * 64];                                \
        R2 = wp_[((S).pkb + 2 < last_ ? (S).pkb + 2 : last_) * 64];                                \
        R3 = wp_[((S).pkb + 3 < last_ ? (S).pkb + 3 : last_) * 64];                                \
        (S).pkb += 4;                                                                              \
        if ((S).pkb >= (S).pjb.nblk) {                                                             \
            if ((S).pi + NWAVES < (S).njobs) {                                                     \
                (S).pi += NWAVES;                                                                  \
                (S).pjb = load_job((S).jl, HINT_JIDX(S, (S).pi));                                  \
                (S).pkb = 0;                                                                       \
            } else (S).pkb -= 4; /* end of stream: harmlessly re-fetch the last chunk */           \
        }                                                                                          \
    }

// Issue everything of a stage that does not depend on the preceding barrier: first job
// descriptor, biases, the first two chunks of packed weights.
__device__ __forceinline__ void stage_begin(Stage& S, lds_jobs_t jl, int njobs,
                                            const float* __restrict__ packed, int wave, int lane) {
    S.jl = jl;
    S.njobs = njobs;
    S.active = wave < njobs;
    if (!S.active) return;
    // Workgroups run in near lock-step; rotate the job order per workgroup so that the CUs of
    // an XCD do not all ask the same L2 channel for the same tile at the same moment.
    S.rot = (int)((blockIdx.x * 5u) % (unsigned)njobs);
    S.pi = wave;
    S.pkb = 0;
    S.pjb = load_job(jl, HINT_JIDX(S, wave));
    S.ci = wave;
    S.ckb = 0;
    S.cjb = S.pjb;
    HINT_FETCH(S, S.a0, S.a1, S.a2, S.a3)
    HINT_FETCH(S, S.b0, S.b1, S.b2, S.b3)
}

template <int EPI>
__device__ __forceinline__ void stage_epilogue(const JobU& jb, f32x4 acc, const float* bias_lds, float* O,
                                               int ldo, int slab_stride, int lane) {
    const int nl = lane & 15;
    const bool ok = nl < jb.nvalid;
    // biases of the group sit in LDS in output-column order (zero in the padding columns);
    // only slab 0 of a K-split stage adds them
    float bias = 0.f;
    if ((EPI == EPI_RELU || EPI == EPI_LINEAR) && jb.slab == 0) bias = bias_lds[jb.ocol + nl];
    float* o = O + jb.slab * slab_stride + (4 * (lane >> 4)) * ldo + jb.ocol + nl;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float v;
        if (EPI == EPI_RELU) v = ok ? fmaxf(acc[i] + bias, 0.f) : 0.f;
        else if (EPI == EPI_LINEAR) v = ok ? acc[i] + bias : 0.f;
        else if (EPI == EPI_MASK) v = (ok && o[i * ldo] > 0.f) ? acc[i] : 0.f;
        else v = ok ? acc[i] : 0.f;
        o[i * ldo] = v;
    }
}

template <int EPI>
__device__ __forceinline__ void stage_run(Stage& S, const float* __restrict__ packed,
                                          const float* bias_lds, const float* A, int lda, float* O, int ldo,
                                          int slab_stride, int lane) {
    if (!S.active) return;
    const int nl = lane & 15;
    const float* arow = A + nl * lda + 4 * (lane >> 4);
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    f32x4 c0, c1, c2, c3;

#define HINT_MMA1(R, I, ACC)                                                                       \
    if (S.ckb + I < S.cjb.nblk) {                                                                  \
        const f32x4 a_ = *(const f32x4*)(ap_ + I * 16);                                            \
        ACC = mfma4(a_.x, R.x, ACC);                                                               \
        ACC = mfma4(a_.y, R.y, ACC);                                                               \
        ACC = mfma4(a_.z, R.z, ACC);                                                               \
        ACC = mfma4(a_.w, R.w, ACC);                                                               \
    }
#define HINT_COMPUTE(R0, R1, R2, R3, DONE)                                                         \
    {                                                                                              \
        const float* ap_ = arow + S.cjb.acol + S.ckb * 16;                                         \
        HINT_MMA1(R0, 0, acc0) HINT_MMA1(R1, 1, acc1) HINT_MMA1(R2, 2, acc0) HINT_MMA1(R3, 3, acc1) \
        S.ckb += 4;                                                                                \
        if (S.ckb >= S.cjb.nblk) {                                                                 \
            stage_epilogue<EPI>(S.cjb, acc0 + acc1, bias_lds, O, ldo, slab_stride, lane);          \
            acc0 = f32x4{0.f, 0.f, 0.f, 0.f};                                                      \
            acc1 = f32x4{0.f, 0.f, 0.f, 0.f};                                                      \
            S.ci += NWAVES;                                                                        \
            if (S.ci >= S.njobs) { DONE = true; }                                                  \
            else {                                                                                 \
                S.cjb = load_job(S.jl, HINT_JIDX(S, S.ci));                                        \
                S.ckb = 0;                                                                         \
            }                                                                                      \
        }                                                                                          \
    }

    bool done = false;
    while (true) {
        HINT_FETCH(S, c0, c1, c2, c3)
        HINT_COMPUTE(S.a0, S.a1, S.a2, S.a3, done)
        if (done) break;
        HINT_FETCH(S, S.a0, S.a1, S.a2, S.a3)
        HINT_COMPUTE(S.b0, S.b1, S.b2, S.b3, done)
        if (done) break;
        HINT_FETCH(S, S.b0, S.b1, S.b2, S.b3)
        HINT_COMPUTE(c0, c1, c2, c3, done)
        if (done) break;
    }
#undef HINT_MMA1
#undef HINT_COMPUTE
}

// Small outer-product tiles done inside the backward kernel (dW1, dW3): the reduction runs
// over the 16 rows of the tile, results go to the flat gradient buffer with float atomics.
__device__ __forceinline__ void run_ojobs(lds_jobs_t jobs, int njobs, const float* Abuf, int lda,
                                          const float* Bbuf, int ldb, float* __restrict__ g, int wave,
                                          int lane) {
    const int nl = lane & 15, kq = lane >> 4;
    for (int j = wave; j < njobs; j += NWAVES) {
        const i32x4 raw = *(const LDS_AS i32x4*)(jobs + j);
        const int goff = __builtin_amdgcn_readfirstlane(raw.x);
        const unsigned y = (unsigned)__builtin_amdgcn_readfirstlane(raw.y);
        const unsigned zz = (unsigned)__builtin_amdgcn_readfirstlane(raw.z);
        const int acol = (int)(y & 0xffffu), bcol = (int)(y >> 16);
        const int ldg = (int)(zz & 0xffffu), mvalid = (int)((zz >> 16) & 0xffu), nvalid = (int)(zz >> 24);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = kq + 4 * i;
            acc = mfma4(Abuf[row * lda + acol + nl], Bbuf[row * ldb + bcol + nl], acc);
        }
        if (nl < nvalid) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = 4 * kq + i;
                if (m < mvalid) atomicAdd(g + goff + (int64_t)m * ldg + nl, acc[i]);
            }
        }
    }
}

// bias gradients: column sums over the 16 rows of an LDS buffer, one thread per column
__device__ __forceinline__ void colsum_atomic(const int32_t* __restrict__ map, int ncols, const float* buf,
                                              int ld, float* __restrict__ g, int tid) {
    for (int col = tid; col < ncols; col += NTHREADS) {
        const int off = map[col];
        if (off < 0) continue;
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < ROWS; ++r) s += buf[r * ld + col];
        atomicAdd(g + off, s);
    }
}

struct VNodeU { int off, k, cin, cinp, vcol; };
__device__ __forceinline__ VNodeU load_vnode(const LDS_AS VNode* vnodes, int ni) {
    const LDS_AS int32_t* p = (const LDS_AS int32_t*)(vnodes + ni);       // 12 bytes = 3 dwords, wave-uniform
    const unsigned w0 = (unsigned)__builtin_amdgcn_readfirstlane(p[0]);
    const unsigned w1 = (unsigned)__builtin_amdgcn_readfirstlane(p[1]);
    const unsigned w2 = (unsigned)__builtin_amdgcn_readfirstlane(p[2]);
    VNodeU u;
    u.off = (int)(w0 & 0xffffu); u.k = (int)(w0 >> 16);
    u.cin = (int)(w1 & 0xffffu); u.cinp = (int)(w1 >> 16);
    u.vcol = (int)(w2 & 0xffffu);
    return u;
}
struct EntU { int xcol, scol, tcol; };
__device__ __forceinline__ EntU load_ent(const LDS_AS Ent* ents, int e) {
    const LDS_AS int32_t* p = (const LDS_AS int32_t*)(ents + e);           // 8 bytes, per-lane address
    const unsigned w0 = (unsigned)p[0], w1 = (unsigned)p[1];
    EntU u;
    u.xcol = (int)(w0 & 0xffffu); u.scol = (int)(w0 >> 16); u.tcol = (int)(w1 & 0xffffu);
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

// v = [u | c] for every node of the group (hint.py:76), zero padded to cinp columns.
__device__ __forceinline__ void stage_build_v(const KArgs& a, int node_begin, int node_end,
                                              const LDS_AS VNode* vnodes, const float* xs, const float* cs,
                                              float* vb, int tid) {
    for (int ni = node_begin; ni < node_end; ++ni) {
        const VNodeU nd = load_vnode(vnodes, ni);
        const int cinp = nd.cinp, k = nd.k, cin = nd.cin, off = nd.off, vcol = nd.vcol;
        for (int i = tid; i < ROWS * cinp; i += NTHREADS) {
            const int r = i / cinp, j = i - r * cinp;
            float v = 0.f;
            if (j < k) v = xs[r * a.xld + off + j];
            else if (j < cin) v = cs[r * a.cld + (j - k)];
            vb[r * a.vld + vcol + j] = v;
        }
    }
}

__device__ __forceinline__ void load_tile(float* dst, int ld, const float* __restrict__ src,
                                          int width, int row0, int B, int tid) {
    // a 16-row tile of a row-major [B,width] tensor is one contiguous run of 16*width floats
    if (src == nullptr) {
        for (int i = tid; i < ROWS * width; i += NTHREADS) { const int r = i / width; dst[r * ld + (i - r * width)] = 0.f; }
        return;
    }
    const float* p = src + (size_t)row0 * width;
    const int nvalid = (B - row0 < ROWS ? B - row0 : ROWS) * width;
    for (int i = tid; i < ROWS * width; i += NTHREADS) {
        const int r = i / width;
        dst[r * ld + (i - r * width)] = (i < nvalid) ? p[i] : 0.f;
    }
}

__device__ __forceinline__ void store_tile(float* __restrict__ dst, const float* src, int ld,
                                           int width, int row0, int B, int tid) {
    float* p = dst + (size_t)row0 * width;
    const int nvalid = (B - row0 < ROWS ? B - row0 : ROWS) * width;
    for (int i = tid; i < nvalid; i += NTHREADS) {
        const int r = i / width;
        p[i] = src[r * ld + (i - r * width)];
    }
}

// The job list of the NEXT group travels global -> registers at the start of a group and
// registers -> LDS near its end, so its L2 latency hides behind the group's GEMM stages.
// (The group's biases, a few KiB, ride along: [b1 | b2 | b3] in LDS column order.)
struct JobPrefetch { i32x4 r0, r1; f32x4 b0; int count, nbias4; };

__device__ __forceinline__ void jobs_issue(JobPrefetch& jp, const GJob* __restrict__ gjobs, int jl_begin,
                                           int jl_count, const float* __restrict__ bias_src, int nbias,
                                           int tid) {
    jp.count = jl_count;
    jp.nbias4 = nbias >> 2;                       // bias blocks are multiples of 16 floats, <= 4*NTHREADS
    const i32x4* src = (const i32x4*)(gjobs + jl_begin);
    if (tid < jp.count) jp.r0 = src[tid];
    if (tid + NTHREADS < jp.count) jp.r1 = src[tid + NTHREADS];
    if (tid < jp.nbias4) jp.b0 = ((const f32x4*)bias_src)[tid];
}
__device__ __forceinline__ void jobs_commit(const JobPrefetch& jp, LDS_AS GJob* jbuf, float* bias_dst, int tid) {
    if (tid < jp.count) ((LDS_AS i32x4*)jbuf)[tid] = jp.r0;
    if (tid + NTHREADS < jp.count) ((LDS_AS i32x4*)jbuf)[tid + NTHREADS] = jp.r1;
    if (tid < jp.nbias4) ((f32x4*)bias_dst)[tid] = jp.b0;
}

// LDS carve-up shared by both block kernels: [meta | job buffer 0 | job buffer 1 | floats...]
__device__ __forceinline__ void lds_copy_meta(const KArgs& a, LDS_AS char* mbase, int tid) {
    const int n16 = a.meta_bytes >> 4;
    for (int i = tid; i < n16; i += NTHREADS) ((LDS_AS i32x4*)mbase)[i] = ((const i32x4*)a.meta)[i];
}

// =======================================================================================
// forward (REV=false) / inverse (REV=true): x, J -> z   — one launch per block
// =======================================================================================
#define HINT_LDS_TABLES()                                                                          \
    LDS_AS char* mbase = (LDS_AS char*)lds;                                                        \
    const LDS_AS DGroup* groups = (const LDS_AS DGroup*)mbase;                                     \
    const LDS_AS VNode* vnodes = (const LDS_AS VNode*)(mbase + a.vnodes_off);                      \
    const LDS_AS Ent* ents = (const LDS_AS Ent*)(mbase + a.ents_off);                              \
    LDS_AS GJob* jbuf0 = (LDS_AS GJob*)(mbase + a.meta_bytes);                                     \
    float* bias0 = lds + ((a.meta_bytes + 2 * a.jmax * (int)sizeof(GJob)) >> 2);                   \
    float* fbase = bias0 + 2 * a.bmax;                                                             \
    lds_copy_meta(a, mbase, tid);

template <bool REV>
__global__ __launch_bounds__(NTHREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void hint_block_apply_kernel(
    KArgs a, const float* __restrict__ params, const float* __restrict__ packed,
    const float* __restrict__ x, const float* __restrict__ c, float* __restrict__ z,
    float* __restrict__ J, float* __restrict__ tape) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    STAMP(0)
    HINT_LDS_TABLES()
    float* xs = fbase;
    float* cs = xs + ROWS * a.xld;
    float* vb = cs + ROWS * a.cld;
    float* a1 = vb + ROWS * a.vld;
    float* a2 = a1 + ROWS * a.ald;
    float* st = a2 + ROWS * a.ald;           // [s3][ROWS][sld]
    float* jac = st + a.s3 * ROWS * a.sld;
    const int sstride = ROWS * a.sld;
    const int ntiles = (a.B + ROWS - 1) / ROWS;
    __syncthreads();                          // meta visible

    int jb = 0;
    Stage S, N;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int row0 = tile * ROWS;
        const bool first_tile = (tile == (int)blockIdx.x);
        const LDS_AS DGroup* gfirst = groups + (REV ? a.n_groups - 1 : 0);
        if (first_tile) {                     // later tiles get the first group's jobs through the prefetch below
            JobPrefetch jp0;
            jobs_issue(jp0, a.jobs, GF(gfirst, jl_begin), GF(gfirst, jl_count), packed + a.bias_off + GF(gfirst, bmap_begin),
                       2 * GF(gfirst, aw) + GF(gfirst, sw), tid);
            jobs_commit(jp0, jbuf0 + jb * a.jmax, bias0 + jb * a.bmax, tid);
        }
        load_tile(xs, a.xld, x, a.d, row0, a.B, tid);
        if (a.dc > 0) load_tile(cs, a.cld, c, a.dc, row0, a.B, tid);
        if (tid < ROWS) jac[tid] = 0.f;
        __syncthreads();
        STAMP(1)
        if (first_tile)
            stage_begin(S, jbuf0 + jb * a.jmax + GF(gfirst, l1_off), GF(gfirst, l1_cnt), packed, wave, lane);

        for (int gi = 0; gi < a.n_groups; ++gi) {
            const LDS_AS DGroup* g = groups + (REV ? (a.n_groups - 1 - gi) : gi);
            const bool more_tiles = tile + (int)gridDim.x < ntiles;
            const bool has_next = (gi + 1 < a.n_groups) || more_tiles;
            const int gnext = (gi + 1 < a.n_groups) ? gi + 1 : 0;
            const LDS_AS DGroup* gn = groups + (REV ? (a.n_groups - 1 - gnext) : gnext);
            JobPrefetch jp;
            jp.count = 0;
            jp.nbias4 = 0;
            if (has_next)
                jobs_issue(jp, a.jobs, GF(gn, jl_begin), GF(gn, jl_count), packed + a.bias_off + GF(gn, bmap_begin),
                           2 * GF(gn, aw) + GF(gn, sw), tid);
            lds_jobs_t jl = jbuf0 + jb * a.jmax;
            LDS_AS GJob* jl_next = jbuf0 + (jb ^ 1) * a.jmax;
            const int l3_slabs = GF(g, l3_slabs), ent_begin = GF(g, ent_begin), ent_cnt = GF(g, ent_cnt);
            const int g_aw = GF(g, aw);
            const float* bias_g = bias0 + jb * a.bmax;     // [b1 | b2 | b3] of this group

            stage_build_v(a, GF(g, node_begin), GF(g, node_end), vnodes, xs, cs, vb, tid);
            STAMP(2 + 12 * gi)
            lds_barrier();
            STAMP(3 + 12 * gi)
            stage_begin(N, jl + GF(g, l2_off), GF(g, l2_cnt), packed, wave, lane);
            stage_run<EPI_RELU>(S, packed, bias_g, vb, a.vld, a1, a.ald, 0, lane);
            STAMP(4 + 12 * gi)
            lds_barrier();
            STAMP(5 + 12 * gi)
            stage_begin(S, jl + GF(g, l3_off), GF(g, l3_cnt), packed, wave, lane);
            stage_run<EPI_RELU>(N, packed, bias_g + g_aw, a1, a.ald, a2, a.ald, 0, lane);
            if (has_next) jobs_commit(jp, jl_next, bias0 + (jb ^ 1) * a.bmax, tid);
            STAMP(6 + 12 * gi)
            lds_barrier();
            STAMP(7 + 12 * gi)
            stage_run<EPI_LINEAR>(S, packed, bias_g + 2 * g_aw, a2, a.ald, st, a.sld, sstride, lane);
            STAMP(8 + 12 * gi)
            lds_barrier();
            STAMP(9 + 12 * gi)
            if (has_next) stage_begin(S, jl_next + GF(gn, l1_off), GF(gn, l1_cnt), packed, wave, lane);
            {   // element-wise affine coupling + log-det partial sums (hint.py:79-83)
                const int sub = tid & 15, row = tid >> 4;
                float part = 0.f;
                if (row < ROWS) {
                    for (int e = sub; e < ent_cnt; e += 16) {
                        const EntU en = load_ent(ents, ent_begin + e);
                        float s = 0.f, t = 0.f;
                        for (int sl = 0; sl < l3_slabs; ++sl) {
                            s += st[sl * sstride + row * a.sld + en.scol];
                            t += st[sl * sstride + row * a.sld + en.tcol];
                        }
                        const float aa = a.alpha * atanf(s);
                        float* px = xs + row * a.xld + en.xcol;
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
            if (!REV && tape != nullptr && GF(g, level_last) && GF(g, level) < a.n_levels - 1)
                store_tile(tape + (size_t)GF(g, level) * a.B * a.d, xs, a.xld, a.d, row0, a.B, tid);
            jb ^= 1;
        }
        store_tile(z, xs, a.xld, a.d, row0, a.B, tid);
        if (tid < ROWS && row0 + tid < a.B) J[row0 + tid] = jac[tid];
        STAMP(120)
        lds_barrier();
    }
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
__device__ __forceinline__ void copy_rows_out(float* __restrict__ dst, int dld, int dcol,
                                              const float* src, int sld, int width, int row0,
                                              int tid) {
    // width is a multiple of 16, dcol/dld multiples of 4 -> 128-bit rows
    const int w4 = width >> 2;
    for (int i = tid; i < ROWS * w4; i += NTHREADS) {
        const int r = i / w4, j = (i - r * w4) << 2;
        *(f32x4*)(dst + (size_t)(row0 + r) * dld + dcol + j) = *(const f32x4*)(src + r * sld + j);
    }
}

__global__ __launch_bounds__(NTHREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void hint_block_bwd_kernel(
    KArgs a, const float* __restrict__ params, const float* __restrict__ packed,
    const float* __restrict__ x, const float* __restrict__ tape, const float* __restrict__ c,
    const float* __restrict__ g_z, const float* __restrict__ g_J, float* __restrict__ g_x,
    float* __restrict__ g_c, float* __restrict__ gparams, float* __restrict__ wsA1,
    float* __restrict__ wsG2) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
    float* st = a2 + ROWS * a.ald;           // [s3][ROWS][sld]
    float* gst = st + a.s3 * ROWS * a.sld;
    float* gj = gst + ROWS * a.sld;
    const int sstride = ROWS * a.sld, vstride = ROWS * a.vld;
    const int ntiles = (a.B + ROWS - 1) / ROWS;
    __syncthreads();                          // meta visible

    int jb = 0;
    Stage S, N;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int row0 = tile * ROWS;
        const bool first_tile = (tile == (int)blockIdx.x);
        const LDS_AS DGroup* gfirst = groups + (a.n_groups - 1);
        if (first_tile) {
            JobPrefetch jp0;
            jobs_issue(jp0, a.jobs, GF(gfirst, jl_begin), GF(gfirst, jl_count), packed + a.bias_off + GF(gfirst, bmap_begin),
                       2 * GF(gfirst, aw) + GF(gfirst, sw), tid);
            jobs_commit(jp0, jbuf0 + jb * a.jmax, bias0 + jb * a.bmax, tid);
        }
        load_tile(gs, a.xld, g_z, a.d, row0, a.B, tid);
        if (a.dc > 0) {
            load_tile(cs, a.cld, c, a.dc, row0, a.B, tid);
            load_tile(gcs, a.cld, nullptr, a.dc, row0, a.B, tid);
        }
        if (tid < ROWS) gj[tid] = (g_J != nullptr && row0 + tid < a.B) ? g_J[row0 + tid] : 0.f;
        __syncthreads();
        STAMP(1)
        if (first_tile)
            stage_begin(S, jbuf0 + jb * a.jmax + GF(gfirst, l1_off), GF(gfirst, l1_cnt), packed, wave, lane);

        for (int gi = a.n_groups - 1; gi >= 0; --gi) {
            const LDS_AS DGroup* g = groups + gi;
            const bool more_tiles = tile + (int)gridDim.x < ntiles;
            const bool has_next = (gi > 0) || more_tiles;
            const LDS_AS DGroup* gn = groups + (gi > 0 ? gi - 1 : a.n_groups - 1);
            JobPrefetch jp;
            jp.count = 0;
            jp.nbias4 = 0;
            if (has_next)
                jobs_issue(jp, a.jobs, GF(gn, jl_begin), GF(gn, jl_count), packed + a.bias_off + GF(gn, bmap_begin),
                           2 * GF(gn, aw) + GF(gn, sw), tid);
            lds_jobs_t jl = jbuf0 + jb * a.jmax;
            LDS_AS GJob* jl_next = jbuf0 + (jb ^ 1) * a.jmax;
            const int sbase = 2 + 20 * (a.n_groups - 1 - gi);
            (void)sbase;
            const int g_aw = GF(g, aw), g_sw = GF(g, sw), g_wcol0 = GF(g, wcol0);
            const float* bias_g = bias0 + jb * a.bmax;     // [b1 | b2 | b3] of this group
            const int node_begin = GF(g, node_begin), node_end = GF(g, node_end);
            const int l3_slabs = GF(g, l3_slabs), dv_slabs = GF(g, dv_slabs);
            const int ent_begin = GF(g, ent_begin), ent_cnt = GF(g, ent_cnt);
            const int bmap_begin = GF(g, bmap_begin), bmap3_begin = GF(g, bmap3_begin);

            // ---- the lanes as the forward pass saw them when it entered this level ----
            if (GF(g, level_last)) {
                const int level = GF(g, level);
                const float* src = (level == 0) ? x : tape + (size_t)(level - 1) * a.B * a.d;
                load_tile(xs, a.xld, src, a.d, row0, a.B, tid);
                lds_barrier();
            }
            // ---- recompute s, t of every node of the group (bit-identical to the forward) ----
            stage_build_v(a, node_begin, node_end, vnodes, xs, cs, vb, tid);
            for (int i = tid; i < ROWS * g_sw; i += NTHREADS) {
                const int r = i / g_sw;
                gst[r * a.sld + (i - r * g_sw)] = 0.f;
            }
            STAMP(sbase + 0)
            lds_barrier();
            STAMP(sbase + 1)
            stage_begin(N, jl + GF(g, l2_off), GF(g, l2_cnt), packed, wave, lane);
            stage_run<EPI_RELU>(S, packed, bias_g, vb, a.vld, a1, a.ald, 0, lane);
            STAMP(sbase + 2)
            lds_barrier();
            STAMP(sbase + 3)
            stage_begin(S, jl + GF(g, l3_off), GF(g, l3_cnt), packed, wave, lane);
            copy_rows_out(wsA1, a.WT, g_wcol0, a1, a.ald, g_aw, row0, tid);
            stage_run<EPI_RELU>(N, packed, bias_g + g_aw, a1, a.ald, a2, a.ald, 0, lane);
            STAMP(sbase + 4)
            lds_barrier();
            STAMP(sbase + 5)
            stage_begin(N, jl + GF(g, g2_off), GF(g, g2_cnt), packed, wave, lane);
            stage_run<EPI_LINEAR>(S, packed, bias_g + 2 * g_aw, a2, a.ald, st, a.sld, sstride, lane);
            STAMP(sbase + 6)
            lds_barrier();
            STAMP(sbase + 7)
            {   // ---- coupling backward ----
                const int sub = tid & 15, row = tid >> 4;
                if (row < ROWS) {
                    const float gJr = gj[row];
                    for (int e = sub; e < ent_cnt; e += 16) {
                        const EntU en = load_ent(ents, ent_begin + e);
                        float s = 0.f;
                        for (int sl = 0; sl < l3_slabs; ++sl) s += st[sl * sstride + row * a.sld + en.scol];
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
            STAMP(sbase + 8)
            lds_barrier();
            STAMP(sbase + 9)
            // ---- dW3 += g_st^T a2, db3 += colsum(g_st)  (a2 still holds the activations) ----
            run_ojobs(jl + GF(g, o3_off), GF(g, o3_cnt), gst, a.sld, a2, a.ald, gparams, wave, lane);
            colsum_atomic(a.bmap + bmap3_begin, g_sw, gst, a.sld, gparams, tid);
            STAMP(sbase + 10)
            lds_barrier();
            STAMP(sbase + 11)
            // ---- g2 = (g_st * W3) .* relu'(a2), in place over a2 ----
            stage_begin(S, jl + GF(g, g1_off), GF(g, g1_cnt), packed, wave, lane);
            stage_run<EPI_MASK>(N, packed, bias_g, gst, a.sld, a2, a.ald, 0, lane);
            STAMP(sbase + 12)
            lds_barrier();
            STAMP(sbase + 13)
            // ---- g1 = (g2 * W2) .* relu'(a1), in place over a1;  db2 += colsum(g2) ----
            stage_begin(N, jl + GF(g, dv_off), GF(g, dv_cnt), packed, wave, lane);
            copy_rows_out(wsG2, a.WT, g_wcol0, a2, a.ald, g_aw, row0, tid);
            colsum_atomic(a.bmap + bmap_begin + g_aw, g_aw, a2, a.ald, gparams, tid);
            stage_run<EPI_MASK>(S, packed, bias_g, a2, a.ald, a1, a.ald, 0, lane);
            STAMP(sbase + 14)
            lds_barrier();
            STAMP(sbase + 15)
            // ---- g_v = [g1_s | g1_t] * [W1_s ; W1_t];  dW1 += g1^T v;  db1 += colsum(g1) ----
            stage_run<EPI_PLAIN>(N, packed, bias_g, a1, a.ald, gv, a.vld, vstride, lane);
            run_ojobs(jl + GF(g, o1_off), GF(g, o1_cnt), a1, a.ald, vb, a.vld, gparams, wave, lane);
            colsum_atomic(a.bmap + bmap_begin, g_aw, a1, a.ald, gparams, tid);
            if (has_next) jobs_commit(jp, jl_next, bias0 + (jb ^ 1) * a.bmax, tid);
            STAMP(sbase + 16)
            lds_barrier();
            STAMP(sbase + 17)
            if (has_next) stage_begin(S, jl_next + GF(gn, l1_off), GF(gn, l1_cnt), packed, wave, lane);
            // ---- scatter g_v: first k columns to the upper lanes, the rest to g_c ----
            for (int ni = node_begin; ni < node_end; ++ni) {
                const VNodeU nd = load_vnode(vnodes, ni);
                const int k = nd.k;
                for (int i = tid; i < ROWS * k; i += NTHREADS) {
                    const int r = i / k, j = i - r * k;
                    float acc = gs[r * a.xld + nd.off + j];
                    for (int sl = 0; sl < dv_slabs; ++sl) acc += gv[sl * vstride + r * a.vld + nd.vcol + j];
                    gs[r * a.xld + nd.off + j] = acc;
                }
            }
            if (a.dc > 0) {
                for (int i = tid; i < ROWS * a.dc; i += NTHREADS) {
                    const int r = i / a.dc, j = i - r * a.dc;
                    float acc = gcs[r * a.cld + j];
                    for (int ni = node_begin; ni < node_end; ++ni) {
                        const VNodeU nd = load_vnode(vnodes, ni);
                        for (int sl = 0; sl < dv_slabs; ++sl) acc += gv[sl * vstride + r * a.vld + nd.vcol + nd.k + j];
                    }
                    gcs[r * a.cld + j] = acc;
                }
            }
            STAMP(sbase + 18)
            lds_barrier();
            STAMP(sbase + 19)
            jb ^= 1;
        }
        store_tile(g_x, gs, a.xld, a.d, row0, a.B, tid);
        if (a.dc > 0 && g_c != nullptr) store_tile(g_c, gcs, a.cld, a.dc, row0, a.B, tid);
        STAMP(120)
        lds_barrier();
    }
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
    const DWJob* __restrict__ jobs, int n_jobs, int splits, const float* __restrict__ wsA1,
    const float* __restrict__ wsG2, int WT, int Bp, int rows_per_wg, float* __restrict__ gparams) {
    __shared__ float red[DW_WAVES][9][64][4];   // 72 KiB

    int jidx, split;
    {
        const int id = blockIdx.x;
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

    const int ntm = min(3, (job.H - job.m0 + 15) >> 4);
    const int ntn = min(3, (job.H - job.n0 + 15) >> 4);
    const int b_begin = split * rows_per_wg;
    const int b_end = min(Bp, b_begin + rows_per_wg);

    f32x4 acc[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // columns beyond the (node, net)'s padded extent are clamped to a valid tile; their
    // products are discarded below, this only keeps every load in bounds and unconditional
    const float* gp[3];
    const float* xp[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        gp[t] = wsG2 + job.col + job.m0 + 16 * (t < ntm ? t : 0) + nl;
        xp[t] = wsA1 + job.col + job.n0 + 16 * (t < ntn ? t : 0) + nl;
    }

    float av[2][4][3], bv[2][4][3];
#define DW_LOAD(BUF, BB)                                                     \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                           \
        const size_t row_ = (size_t)((BB) + 4 * i + kq) * WT;                 \
        _Pragma("unroll") for (int t = 0; t < 3; ++t) {                       \
            av[BUF][i][t] = gp[t][row_];                                      \
            bv[BUF][i][t] = xp[t][row_];                                      \
        }                                                                     \
    }
#define DW_MMA(BUF)                                                           \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                             \
        _Pragma("unroll") for (int tm = 0; tm < 3; ++tm)                      \
            _Pragma("unroll") for (int tn = 0; tn < 3; ++tn)                  \
                acc[tm][tn] = mfma4(av[BUF][i][tm], bv[BUF][i][tn], acc[tm][tn]);

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
        const int n = job.n0 + 16 * tn + (l & 15);
        if (n >= job.H) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = job.m0 + 16 * tm + 4 * (l >> 4) + i;
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

hipError_t launch_zero(float* p, long n, int num_cu, hipStream_t stream) {
    const long n4 = n / 4;
    long blocks = (n4 + 255) / 256;
    blocks = blocks < 1 ? 1 : (blocks > (long)num_cu * 4 ? (long)num_cu * 4 : blocks);
    hipLaunchKernelGGL(hint_zero_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, p, n4, n);
    return hipGetLastError();
}

hipError_t launch_apply(bool rev, const KArgs& a, int lds_bytes, int grid, const float* params,
                        const float* packed, const float* x, const float* c, float* z, float* J,
                        float* tape, hipStream_t stream) {
    if (rev)
        hipLaunchKernelGGL(hint_block_apply_kernel<true>, dim3(grid), dim3(NTHREADS), lds_bytes, stream, a,
                           params, packed, x, c, z, J, (float*)nullptr);
    else
        hipLaunchKernelGGL(hint_block_apply_kernel<false>, dim3(grid), dim3(NTHREADS), lds_bytes, stream, a,
                           params, packed, x, c, z, J, tape);
    return hipGetLastError();
}

hipError_t launch_bwd(const KArgs& a, int lds_bytes, int grid, const float* params,
                      const float* packed, const float* x, const float* tape, const float* c,
                      const float* g_z, const float* g_J, float* g_x, float* g_c, float* gparams,
                      float* wsA1, float* wsG2, hipStream_t stream) {
    hipLaunchKernelGGL(hint_block_bwd_kernel, dim3(grid), dim3(NTHREADS), lds_bytes, stream, a, params,
                       packed, x, tape, c, g_z, g_J, g_x, g_c, gparams, wsA1, wsG2);
    return hipGetLastError();
}

hipError_t launch_dw(const DWJob* jobs, int n_jobs, int splits, const float* wsA1, const float* wsG2, int WT,
                     int Bp, int rows_per_wg, float* gparams, hipStream_t stream) {
    if (n_jobs > 0)
        hipLaunchKernelGGL(hint_block_dw_kernel, dim3(n_jobs * splits), dim3(DW_WAVES * 64), 0, stream, jobs,
                           n_jobs, splits, wsA1, wsG2, WT, Bp, rows_per_wg, gparams);
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
    return hipFuncSetAttribute((const void*)hint_block_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bwd_bytes);
}

}  // namespace hint
