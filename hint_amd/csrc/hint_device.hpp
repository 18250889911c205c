// Device-side helpers shared by the block kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include "hint_dev.h"

namespace hint {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

// Explicit address spaces.  A pointer that loses its address space is read with flat_load, which
// counts against vmcnt AND lgkmcnt and so serialises LDS traffic with the weight prefetch.
#define LDS_AS __attribute__((address_space(3)))
#define GLOBAL_AS __attribute__((address_space(1)))

__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }

// Workgroup barrier that orders LDS traffic only: __syncthreads() makes hipcc drain vmcnt(0) first,
// which would retire the weight prefetch at every phase boundary.  Global stores issued before it
// are never read back inside the kernel.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// floor(i / d) for 0 <= i < 2^20 through the float unit (three instructions instead of ~25)
__device__ __forceinline__ float frcp(int d) { return __builtin_amdgcn_rcpf((float)d); }
__device__ __forceinline__ int fdiv(int i, float inv) { return (int)(((float)i + 0.5f) * inv); }

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 zero4() { return f32x4{0.f, 0.f, 0.f, 0.f}; }

__device__ __forceinline__ float row16_sum(float v) {      // deterministic butterfly over 16 adjacent lanes
    v += __shfl_xor(v, 8, 16);
    v += __shfl_xor(v, 4, 16);
    v += __shfl_xor(v, 2, 16);
    v += __shfl_xor(v, 1, 16);
    return v;
}

// ChainBlock with the pointers typed as global memory (pointers that arrive inside a by-value struct
// or a device table are generic to the compiler).
struct GBlock {
    const GLOBAL_AS float* params;
    const GLOBAL_AS float* packed;
    const GLOBAL_AS float* perm;
    GLOBAL_AS float* tape;
    GLOBAL_AS float* actA1;
    GLOBAL_AS float* wsG1;
    GLOBAL_AS float* wsGST;
    GLOBAL_AS float* wsSlab;
    GLOBAL_AS float* gparams;
    const GLOBAL_AS float* x_in;
    const GLOBAL_AS float* c_in;
    const GLOBAL_AS float* g_add;
};
template <int MODE = 0>        // 0: whichever is there; 1: the table (chained launch); 2: the by-value block
__device__ __forceinline__ GBlock chain_block(const ChainBlock* __restrict__ chain, const ChainBlock& one, int i) {
    const ChainBlock b = MODE == 1 ? chain[i] : MODE == 2 ? one : (chain != nullptr) ? chain[i] : one;
    GBlock g;
    g.params = (const GLOBAL_AS float*)b.params;
    g.packed = (const GLOBAL_AS float*)b.packed;
    g.perm = (const GLOBAL_AS float*)b.perm;
    g.tape = (GLOBAL_AS float*)b.tape;
    g.actA1 = (GLOBAL_AS float*)b.actA1;
    g.wsG1 = (GLOBAL_AS float*)b.wsG1;
    g.wsGST = (GLOBAL_AS float*)b.wsGST;
    g.wsSlab = (GLOBAL_AS float*)b.wsSlab;
    g.gparams = (GLOBAL_AS float*)b.gparams;
    g.x_in = (const GLOBAL_AS float*)b.x_in;
    g.c_in = (const GLOBAL_AS float*)b.c_in;
    g.g_add = (const GLOBAL_AS float*)b.g_add;
    return g;
}

// ---- wave-uniform reads of the plan tables staged in LDS ----
struct UnitU {
    int w1v, f2, f3, w3v, b2, b1, bias1, bias2, bias3, wcol, tile0, gcol, NT, KB1, RT, cin, ku, r, xoff, h, sl_off, sl_n,
        gv_off, lcol;
};
__device__ __forceinline__ UnitU load_unit(const LDS_AS Unit* u) {
    const LDS_AS i32x4* p = (const LDS_AS i32x4*)u;
    const i32x4 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3], q4 = p[4], q5 = p[5];
    UnitU r;
    r.w1v = rfl(q0.x); r.f2 = rfl(q0.y); r.f3 = rfl(q0.z); r.w3v = rfl(q0.w);
    r.b2 = rfl(q1.x); r.b1 = rfl(q1.y); r.bias1 = rfl(q1.z); r.bias2 = rfl(q1.w);
    r.bias3 = rfl(q2.x); r.wcol = rfl(q2.y); r.tile0 = rfl(q2.z); r.gcol = rfl(q2.w);
    r.NT = rfl(q3.x); r.KB1 = rfl(q3.y); r.RT = rfl(q3.z); r.cin = rfl(q3.w);
    r.ku = rfl(q4.x); r.r = rfl(q4.y); r.xoff = rfl(q4.z); r.h = rfl(q4.w);
    r.sl_off = rfl(q5.x); r.sl_n = rfl(q5.y); r.gv_off = rfl(q5.z); r.lcol = rfl(q5.w);
    return r;
}
struct GroupU {
    int unit_begin, unit_end, ntiles, row_begin, ent_begin, ent_cnt, rng_begin, level, level_last, gcol0, gcols,
        lop_begin, level_first, tile_begin, wcol0, lean, staged, sub,
        nothin;     // lean, or lean-wide (Group::lean bit 3): the group's a1 / g2 tiles are not stored - part B rebuilds them
};
__device__ __forceinline__ GroupU load_group(const LDS_AS Group* g) {
    const LDS_AS i32x4* p = (const LDS_AS i32x4*)g;
    const i32x4 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3];
    GroupU r;
    r.unit_begin = rfl(q0.x); r.unit_end = rfl(q0.y); r.ntiles = rfl(q0.z); r.row_begin = rfl(q0.w);
    r.ent_begin = rfl(q1.x); r.ent_cnt = rfl(q1.y); r.rng_begin = rfl(q1.z); r.level = rfl(q1.w);
    r.level_last = rfl(q2.x); r.gcol0 = rfl(q2.y); r.gcols = rfl(q2.z); r.lop_begin = rfl(q2.w);
    r.level_first = rfl(q3.x); r.tile_begin = rfl(q3.y); r.wcol0 = rfl(q3.z); r.lean = rfl(q3.w) & 1; r.staged = (rfl(q3.w) >> 1) & 1; r.sub = (rfl(q3.w) >> 2) & 1;
    r.nothin = (rfl(q3.w) & 9) != 0 ? 1 : 0;
    return r;
}
__device__ __forceinline__ int lds_i32(const LDS_AS int32_t* p) { return rfl(*p); }
__device__ __forceinline__ int lds_u16(const LDS_AS uint16_t* p) { return rfl((int)*p); }

// ---- [16, width] tiles of row-major [B, width] tensors: one contiguous run of 16*width floats ----
__device__ __forceinline__ void load_tile(float* dst, int ld, const float* __restrict__ src, int width, int row0,
                                          int B, int tid, int nthreads) {
    const float inv = frcp(width);
    if (src == nullptr) {
        for (int i = tid; i < ROWS * width; i += nthreads) { const int r = fdiv(i, inv); dst[r * ld + (i - r * width)] = 0.f; }
        return;
    }
    const float* p = src + (size_t)row0 * width;
    const int nvalid = (B - row0 < ROWS ? B - row0 : ROWS) * width;
    // (two elements per thread and turn, both loads in flight: d = 43 on 512 threads is one turn instead of two latencies)
    const int n = ROWS * width;
    for (int i = tid; i < n; i += 2 * nthreads) {
        const int i2 = i + nthreads;
        const bool ok2 = i2 < n;
        const float v = (i < nvalid) ? p[i] : 0.f, v2 = (ok2 && i2 < nvalid) ? p[i2] : 0.f;
        const int r = fdiv(i, inv), r2 = fdiv(ok2 ? i2 : i, inv);
        dst[r * ld + (i - r * width)] = v;
        if (ok2) dst[r2 * ld + (i2 - r2 * width)] = v2;
    }
}
__device__ __forceinline__ void store_tile(float* __restrict__ dst, const float* src, int ld, int width, int row0,
                                           int B, int tid, int nthreads) {
    float* p = dst + (size_t)row0 * width;
    const int nvalid = (B - row0 < ROWS ? B - row0 : ROWS) * width;
    const float inv = frcp(width);
    for (int i = tid; i < nvalid; i += 2 * nthreads) {
        const int i2 = i + nthreads;
        const bool ok2 = i2 < nvalid;
        const int r = fdiv(i, inv), r2 = fdiv(ok2 ? i2 : i, inv);
        const float v = src[r * ld + (i - r * width)], v2 = src[r2 * ld + ((ok2 ? i2 : i) - r2 * width)];
        p[i] = v;
        if (ok2) p[i2] = v2;
    }
}

// The fixed d x d matrix between two blocks on the matrix pipe: out[r][j] = sum_k in[r][k] * M(k, j) for the 16 rows of a
// lane tile, M(k, j) = w[k*d + j] (x' = x W) or, TRANS, w[j*d + k] (x = x' W^T; g_x = g_x' W^T).  Transposed like every
// product here: out^T[16 columns x 16 rows] = M^T * in^T, k in steps of four; wavefront `wave` of nw takes the 16-column
// tiles wave, wave + nw (d <= 128 on >= 4 wavefronts: at most PERM_TQ of them).  Eight steps' operands are fetched at a
// time.  (As scalar code - 16 d outputs of d FMAs with two loads each, perm_dot below - the product cost a d = 43 block 7 k
// cycles per kernel; k ascends in both, so the sums agree.)
constexpr int PERM_TQ = 2;
// (WP: the matrix through an LDS or a global pointer - a pointer that may be either is read with flat_load, whose waits drain both counters)
template <bool TRANS, typename WP>
__device__ __forceinline__ void perm_mfma(f32x4 (&acc)[PERM_TQ], const float* in, int ld, WP w, int d,
                                          int wave, int nw, int lane) {
    const int m = lane & 15, kq = lane >> 4;
    const int nt = (d + 15) >> 4, ns = (d + 3) >> 2;
#pragma unroll
    for (int q = 0; q < PERM_TQ; ++q) {
        acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int t = wave + q * nw;
        if (t >= nt) continue;
        const int j = 16 * t + m;
        const bool jok = j < d;
        const int jc = jok ? j : d - 1;
        constexpr int U = 8;
        for (int s0 = 0; s0 < ns; s0 += U) {
            float av[U], bv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int k = 4 * (s0 + u) + kq;
                const int kc = k < d ? k : d - 1;
                av[u] = TRANS ? w[(size_t)jc * d + kc] : w[(size_t)kc * d + jc];
                bv[u] = in[m * ld + kc];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool ok = 4 * (s0 + u) + kq < d;
                acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32((ok && jok) ? av[u] : 0.f, ok ? bv[u] : 0.f, acc[q], 0, 0, 0);
            }
        }
    }
}
__device__ __forceinline__ void perm_store(const f32x4 (&acc)[PERM_TQ], float* out, int ld, int d, int wave, int nw, int lane) {
    const int m = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int q = 0; q < PERM_TQ; ++q) {
        const int j0 = 16 * (wave + q * nw) + 4 * kq;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (j0 + i < d) out[m * ld + j0 + i] = acc[q][i];
    }
}

// sum_k row[k] * w[k * stride]: a row of the lane tile times a column (or row) of a fixed d x d matrix
// (WP: the matrix pointer WITH its address space - LDS_AS when the chain's matrices ride in LDS, GLOBAL_AS otherwise: one
//  pointer that may be either is read with flat_load, whose wait drains the vector-memory AND the LDS counter - at every block
//  boundary of the wave-local kernels that was the weight stream's hand-over)
template <typename WP>
__device__ __forceinline__ float perm_dot(const float* row, WP w, int stride, int d) {
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
// Dequantisation noise of a training step (train_unconditional.py:121, x += 0.01*randn_like(x)),
// drawn inside the forward kernel: no extra launch, no HBM round trip.
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

// Diagnostic build only (-DHINT_STAMPS): shader-clock stamps of workgroup 0 at phase boundaries, one row
// of STAMP_IDS per wavefront, collected in LDS (a global store per stamp would sit in the vmcnt queue in
// front of the weight loads and distort what it measures) and flushed when the kernel ends.
#ifdef HINT_STAMPS
constexpr int STAMP_IDS = 512;
static __shared__ unsigned long long hint_stamp_lds[MAX_NW * STAMP_IDS];
#define STAMP_DECL() { for (int i_ = threadIdx.x; i_ < MAX_NW * STAMP_IDS; i_ += blockDim.x) hint_stamp_lds[i_] = 0ull; __syncthreads(); }
#define STAMP(ID) { if (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && (ID) < STAMP_IDS) \
    hint_stamp_lds[(threadIdx.x >> 6) * STAMP_IDS + (ID)] = __builtin_amdgcn_s_memtime(); }
#define STAMP_FLUSH(BUF) { __syncthreads(); if ((BUF) != nullptr && blockIdx.x == 0) \
    for (int i_ = threadIdx.x; i_ < MAX_NW * STAMP_IDS; i_ += blockDim.x) (BUF)[i_] = hint_stamp_lds[i_]; }
#else
#define STAMP_DECL() {}
#define STAMP(ID) {}
#define STAMP_FLUSH(BUF) {}
#endif

// The fragment tiles of a finished group, LDS -> the [Bp][WT] array in global memory, by all threads: 16 rows x
// (16 ntiles) columns starting at column wcol0, whole 128-byte lines per batch row (the stores of a row phase
// would retire slowly - partial lines - and everything a wavefront waits for queues behind its own stores).
__device__ __forceinline__ void stream_tiles(float* __restrict__ dst, const float* tiles, int ntiles, int wcol0, int WT,
                                             int row0, int tid, int nthreads) {
    const int w4 = ntiles * 4;                          // float4 columns of the group
    const float inv = frcp(w4);
    const int n = ROWS * w4;
    for (int i0 = tid; i0 < n; i0 += 4 * nthreads) {    // four elements per thread in flight: reads first, then the stores
        f32x4 v[4];
        int r[4], c4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = min(i0 + u * nthreads, n - 1);
            r[u] = fdiv(idx, inv); c4[u] = idx - r[u] * w4;
            v[u] = *(const f32x4*)(tiles + (c4[u] >> 2) * 256 + ((c4[u] & 3) * 16 + r[u]) * 4);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i0 + u * nthreads < n) *(f32x4*)(dst + (size_t)(row0 + r[u]) * WT + wcol0 + 4 * c4[u]) = v[u];
    }
}

// LDS carve-up shared by the block kernels: [meta blob][float buffers ...]
struct Tables {
    const LDS_AS Group* groups;
    const LDS_AS Unit* units;
    const LDS_AS uint16_t* tmap;
    const LDS_AS Ent* ents;
    const LDS_AS int32_t* rng;
    const LDS_AS LaneOp* lops;
};
__device__ __forceinline__ Tables make_tables(const KArgs& a, float* lds) {
    LDS_AS char* mbase = (LDS_AS char*)lds;
    Tables t;
    t.groups = (const LDS_AS Group*)mbase;
    t.units = (const LDS_AS Unit*)(mbase + a.units_off);
    t.tmap = (const LDS_AS uint16_t*)(mbase + a.tmap_off);
    t.ents = (const LDS_AS Ent*)(mbase + a.ents_off);
    t.rng = (const LDS_AS int32_t*)(mbase + a.rng_off);
    t.lops = (const LDS_AS LaneOp*)(mbase + (a.lops_off >= 0 ? a.lops_off : 0));
    return t;
}

// ---- L2 warm-up of the packed weights, a phase or two ahead of their use ----
// Every workgroup carries its 16 rows through the same blocks and groups at about the same time, so the first touch of a group's
// weights is an L2 miss for all 32 workgroups of an XCD at once (eight XCDs with an L2 each; workgroups are dealt round robin:
// XCD = blockIdx & 7), and a block's tape traffic has pushed the weights out of the L2 - often out of the memory-side cache - since
// the launch before.  ONE wavefront of every workgroup therefore touches what comes two consumers later: the workgroups of an XCD
// share the range's 128-byte lines out among themselves, one line per lane, loaded straight into a 256-byte sink in LDS nobody
// reads (`global_load_lds_dword`: no register waits for it; the wavefront's later loads do - vector-memory operations retire in
// order - so the caller is a wavefront with slack in front of a barrier).
// The consumers of a block in packed order: position 0 = [thin blobs | tiles of the subtree groups] + the biases at the buffer's
// end ("head": what block_stage and the subtree phase read), position 1 + i = the fragment tiles of general group i (forward
// order) or of general group ngen - 1 - i (backward / inverse order: root first).  MINIBOONE x 10: forward 341 -> 309 us,
// backward 452 -> 426; the d = 100 flows 745 -> 670 / 898 -> 880; h = 512: 1215 -> 989 / 1327 -> 1250 with whole blocks a block ahead.
__device__ __forceinline__ void prefetch_range(const GLOBAL_AS float* base, int l0, int l1, float* sink, int lane) {
    const int wgs = (int)(gridDim.x >> 3);
    const int per_xcd = wgs < 32 ? (wgs > 0 ? wgs : 1) : 32;
    const int share = (int)(blockIdx.x >> 3);
    if (share >= per_xcd) return;
    for (int b0 = l0; b0 < l1; b0 += per_xcd * 64) {
        const int line = b0 + share + per_xcd * lane;
        if (line < l1)
            __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)((const GLOBAL_AS char*)base + (size_t)line * 128), (LDS_AS void*)sink, 4, 0, 0);
    }
}
template <bool FWD_ORDER>
__device__ __forceinline__ void prefetch_consumer(const KArgs& a, const Tables& T, const GLOBAL_AS float* packed, int pos, float* sink, int lane) {
    auto first_tile = [&](int g) -> int {        // first packed tile of group g's units (Unit::f2 of its first unit); behind the last group: the tiles' end
        if (g >= a.n_groups) return a.packed_tiles;
        const int ub = rfl(*((const LDS_AS int32_t*)(T.groups + g)));
        return rfl(*((const LDS_AS int32_t*)(T.units + ub) + 1));
    };
    if (pos == 0) {
        prefetch_range(packed, 0, 8 * first_tile(a.n_sub), sink, lane);
        prefetch_range(packed, 8 * a.packed_tiles, a.packed_lines, sink, lane);
    } else {
        const int g = FWD_ORDER ? a.n_sub + pos - 1 : a.n_groups - pos;
        prefetch_range(packed, 8 * first_tile(g), 8 * first_tile(g + 1), sink, lane);
    }
}
__device__ __forceinline__ void copy_meta(const KArgs& a, float* lds, int tid, int nthreads) {
    const int n16 = a.meta_bytes >> 4;
    LDS_AS i32x4* dst = (LDS_AS i32x4*)lds;
    const i32x4* src = (const i32x4*)a.meta;
    for (int i = tid; i < n16; i += nthreads) dst[i] = src[i];
}

}  // namespace hint
