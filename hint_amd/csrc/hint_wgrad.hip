// Backward, part B (gfx950 / CDNA4 only): every weight and bias gradient of a block is a reduction
// over the batch of per-row factors that part A (hint_bwd.hip) and the forward tape left in HBM:
//   dW1[f][k] = sum_b g1[b][f] v[b][k]      v = [lanes of the node's level | condition]
//   dW2[m][n] = sum_b g2[b][m] a1[b][n]
//   dW3[j][f] = sum_b g_st[b][j] a2[b][f]
//   db1 = colsum g1, db2 = colsum g2, db3 = colsum g_st
// i.e. GEMMs whose K dimension is the batch.  Lean plans (hint_dev.h) keep neither a1 nor g2 in HBM: their dW2 jobs
// rebuild both operands per 16-row step with ONE extra MFMA per 16-column tile - a1 = relu(v W1^T + b1) and
// g2 = relu'(a2) (g_st W3), K <= 4 each - whose result layout (lane l: rows 4(l>>4)+i of column l&15) is exactly
// the A / B operand layout of the products over the row subsets {4 kq + i}; the inputs are one float of v, one of
// g_st and one word of a2 sign bytes per lane and step.  The OUTPUT is tiled (up to 48x48 per workgroup: 3x3
// MFMA tiles fed by one 12-byte load per operand and k-step) and the batch is split over `splits`
// workgroups and the 8 wavefronts of each.  Block ids are mapped so that all tiles of one batch split
// run on the same XCD (blocks b, b+8, .. share one): the rows of a split are fetched from HBM /
// Infinity Cache once and re-read from that XCD's L2 by the tiles that share them.
// Deterministic: every (tile, split) writes its partial to the split's own slab with plain stores and
// hint_wreduce_kernel adds the slabs in a fixed order - no float atomics anywhere.
#include "hint_device.hpp"
#include "hint_adam.hpp"

using namespace hint;

// wavefronts per workgroup: they split the rows of a (job, split) and combine in LDS.  Two since round 4 (eight before): a
// wavefront's fixed costs - records, parameters, the partials' trip through LDS, the stores - are paid once per 16 row steps
// instead of once per 4, and eight small workgroups per CU are at different points of their lives where two large ones went
// through their phases together (cfg 2: 82 -> 75 us, the d = 100 flows 658 -> 519; 4: 75 / 551; 1: 77 / 510, MINIBOONE 201 against 180)
#ifndef HINT_DW_WAVES
#define HINT_DW_WAVES 2
#endif
constexpr int DW_WAVES = HINT_DW_WAVES;

#ifndef HINT_DW_BOUND
#define HINT_DW_BOUND , 4
#endif

// Diagnostic build only (-DHINT_STAMPS, `make stamps`): every wavefront of part B leaves DW_STAMP_IDS shader-clock stamps of its
// phase boundaries in a global buffer [workgroup][wavefront][id] (hint_debug_set_dw_stamp_buffer; tools/stamps_dw.py reads them)
#ifdef HINT_STAMPS
constexpr int DW_STAMP_IDS = 8;
static unsigned long long* h_dw_stamps = nullptr;
namespace hint { void set_dw_stamps(unsigned long long* p) { h_dw_stamps = p; } }
#define DW_STAMP(ID) { if (dw_stamps != nullptr) dw_st[ID] = __builtin_amdgcn_s_memtime(); }
#define DW_STAMP_WAIT(ID) { if (dw_stamps != nullptr) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); dw_st[ID] = __builtin_amdgcn_s_memtime(); } }
#define DW_STAMP_FLUSH() { if (dw_stamps != nullptr && (threadIdx.x & 63) == 0) { \
    unsigned long long* o_ = dw_stamps + ((size_t)blockIdx.x * DW_WAVES + (threadIdx.x >> 6)) * DW_STAMP_IDS; \
    for (int i_ = 0; i_ < DW_STAMP_IDS; ++i_) o_[i_] = dw_st[i_]; } }
#else
#define DW_STAMP(ID) {}
#define DW_STAMP_WAIT(ID) {}
#define DW_STAMP_FLUSH() {}
#endif

struct SrcRef { const GLOBAL_AS float* p; int ld; int rows; };     // (global address space kept: a generic pointer is read with flat_load, whose waits drain both counters)

__device__ __forceinline__ SrcRef wsrc(int src, int level, const GBlock& blk, bool top, const GLOBAL_AS float* __restrict__ x,
                                       const GLOBAL_AS float* __restrict__ c, int WT, int ST, int d, int dc, int n_levels, int B, int Bp,
                                       int64_t act_stride, int64_t a2_off) {
    SrcRef r;
    r.rows = Bp;
    switch (src) {
        case WSRC_G1: r.p = blk.wsG1; r.ld = WT; break;
        case WSRC_G2: r.p = blk.wsG1 + act_stride; r.ld = WT; break;
        case WSRC_GST: r.p = blk.wsGST; r.ld = ST; break;
        case WSRC_A1: r.p = blk.actA1; r.ld = WT; break;
        case WSRC_A2: r.p = blk.actA1 + a2_off; r.ld = WT; break;
        case WSRC_X: {
            const GLOBAL_AS float* tape = blk.tape;
            const size_t lvl = (size_t)B * d;
            r.p = level == 0 ? (top ? tape + (size_t)(n_levels - 1) * lvl : (blk.x_in != nullptr ? blk.x_in : x)) : tape + (size_t)(level - 1) * lvl;
            r.ld = d; r.rows = B;
            break;
        }
        default: r.p = blk.c_in != nullptr ? blk.c_in : c; r.ld = dc; r.rows = B; break;
    }
    return r;
}

// ---- the row loops of a (job, split), one instance per tile shape ----
// On gfx950 vector-ALU instructions do NOT hide under the matrix pipe: a SIMD's time is 32 cycles per MFMA PLUS ~4-5 per VALU
// instruction, whatever the number of wavefronts (tools/mfma_valu_bench.hip).  Round 3's loops issued 136 VALU instructions per
// step beside 25-42 MFMAs - address arithmetic of the loads, relu, masks, column sums: 45 % of the kernel's time.  So every load
// here goes through a BUFFER descriptor that the scalar unit re-bases per 16-row step (the per-lane offset is loop invariant,
// rows behind the operand's last read as zero by the descriptor's bound), and the element-wise work is cut to what the
// arithmetic needs: 4 v_max per a1 tile, 8 (bit-field extract + and) per g2 tile, 4 adds per tile for the bias gradient.
#ifndef HINT_DW_LEAN_RING
#define HINT_DW_LEAN_RING 3
#endif
constexpr int DW_LEAN_RING = HINT_DW_LEAN_RING;
constexpr int BUF_FLAGS = 0x00020000;           // raw buffer, 32-bit data format (gfx9 family)
constexpr int BUF_OOB = 0x7ffffff0;             // a per-lane offset past any descriptor's bound: the load returns zero
typedef unsigned u32x3 __attribute__((ext_vector_type(3)));
struct DwCtx { int nl, kq, step, bb0, b_end; };
struct LeanSrc { const GLOBAL_AS float* prm; SrcRef vs; const GLOBAL_AS float* gst; const GLOBAL_AS uint8_t* bits; int ntiles, ST; };
struct GenSrc { SrcRef ps, qs; int pc0, qc0, pcj[3], qcj[3]; };

// descriptor of rows [bb, bb + 16) of a row-major operand, `valid` of them inside the operand (the others read as zero).  Only the
// last 16-row block of a batch can be partial, so `valid` is 16 or a value computed ONCE per loop (rows_valid): a clamp inside the
// loop becomes a v_med3 - there is no scalar one - and with it the descriptor a per-lane value that hipcc wraps in a waterfall loop
__device__ __forceinline__ int rows_valid(const SrcRef& sr, int bb) {
    return rfl(max(0, min(sr.rows - bb, 16)));
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rows_rsrc(const SrcRef& sr, int bb, int valid) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)(sr.p + (size_t)bb * sr.ld), 0, valid * sr.ld * 4, BUF_FLAGS);
}
// relu in ONE instruction: fmaxf() is canonicalise + max for the sake of signalling NaNs; the matrix pipe's outputs need none
__device__ __forceinline__ float relu1(float x) {
#pragma clang fp reassociate(on)
    return __builtin_amdgcn_fmed3f(x, 0.f, 3.0e38f);
}
// (by-value helpers: __builtin_bit_cast applied to a vector ELEMENT expression - v.z, v[i] - reads element 0 with this clang)
__device__ __forceinline__ float as_f32(unsigned u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ int as_i32(float f) { return __builtin_bit_cast(int, f); }
__device__ __forceinline__ float buf_f32(__amdgpu_buffer_rsrc_t r, int voff) {
    return as_f32(__builtin_amdgcn_raw_buffer_load_b32(r, voff, 0, 0));
}

// lean dW2 (hint_dev.h: WSRC_G2R x WSRC_A1R): per 16-row step one float of v, one of g_st and NTM words of a2 sign bytes per lane;
// a1 = relu(v W1^T + b1) and g2 = relu'(a2) (g_st W3) are one K <= 4 MFMA per tile, whose result layout is the operand layout of
// the products over the row subsets {4 kq + i}
template <int NTM, int NTN>
__device__ __forceinline__ void dw_lean(f32x4 (&acc)[3][3], float (&psum)[3], const WJob& job, const LeanSrc& ls, const DwCtx& cx) {
    const int nl = cx.nl, kq = cx.kq;
    float w1b[NTN], w3b[NTM];
    f32x4 b1v[NTN];
#pragma unroll
    for (int t = 0; t < NTN; ++t) {
        const int fq = job.qcol - job.r_wcol + 16 * t + nl;
        w1b[t] = (kq < job.r_cin && fq < job.r_h) ? ls.prm[job.r_w1 + (size_t)fq * job.r_cin + kq] : 0.f;
        const float b1 = fq < job.r_h ? ls.prm[job.r_b1 + fq] : 0.f;
        b1v[t] = f32x4{b1, b1, b1, b1};
    }
#pragma unroll
    for (int t = 0; t < NTM; ++t) {
        const int fp = job.pcol - job.r_wcol + 16 * t + nl;
        w3b[t] = (kq < job.r_r && fp < job.r_h) ? ls.prm[job.r_w3 + (size_t)kq * job.r_h + fp] : 0.f;
    }
    // loop-invariant per-lane offsets: the lanes without an input (kq >= cin / r) read past the bound = zero
    const int voff_v = kq < job.r_cin ? (nl * ls.vs.ld + job.r_xoff + kq) * 4 : BUF_OOB;
    const int voff_g = kq < job.r_r ? (nl * ls.ST + job.r_gcol + kq) * 4 : BUF_OOB;
    const int voff_b = 16 * (nl >> 2) + 4 * kq;
    int bit[4];                                   // where the lane's four mask bits sit in its word of sign bytes
#pragma unroll
    for (int i = 0; i < 4; ++i) bit[i] = (nl & 3) + 8 * i;
    SrcRef gsr; gsr.p = ls.gst; gsr.ld = ls.ST; gsr.rows = 0x7fffffff;
    const int nsteps = (cx.b_end - cx.bb0 + cx.step - 1) / cx.step;
    if (nsteps <= 0) return;
    const int last_bb = cx.bb0 + (nsteps - 1) * cx.step;
    const int v_last = rows_valid(ls.vs, last_bb);
    struct In { float v, g; unsigned s[NTM]; };
    auto load_in = [&](int BB) {
        In in;
        in.v = buf_f32(rows_rsrc(ls.vs, BB, BB == last_bb ? v_last : 16), voff_v);
        in.g = buf_f32(rows_rsrc(gsr, BB, 16), voff_g);
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(ls.bits + ((size_t)(BB >> 4) * ls.ntiles + (job.pcol >> 4)) * 64), 0, NTM * 64, BUF_FLAGS);
#pragma unroll
        for (int t = 0; t < NTM; ++t) in.s[t] = __builtin_amdgcn_raw_buffer_load_b32(rb, voff_b + 64 * t, 0, 0);
        return in;
    };
    auto mma = [&](const In& in) {
        f32x4 p[NTM], q[NTN];
#pragma unroll
        for (int t = 0; t < NTN; ++t) q[t] = mfma4(in.v, w1b[t], b1v[t]);
#pragma unroll
        for (int t = 0; t < NTM; ++t) p[t] = mfma4(in.g, w3b[t], zero4());
#pragma unroll
        for (int t = 0; t < NTN; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) q[t][i] = relu1(q[t][i]);
#pragma unroll
        for (int t = 0; t < NTM; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                // relu'(a2): the mask bit sign-extended to 0 / -1 and and-ed onto the float (two instructions per element)
                const int m = __builtin_amdgcn_sbfe((int)in.s[t], bit[i], 1);
                p[t][i] = as_f32((unsigned)(as_i32(p[t][i]) & m));
            }
#pragma unroll
        for (int t = 0; t < NTM; ++t) psum[t] += (p[t].x + p[t].y) + (p[t].z + p[t].w);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int tm = 0; tm < NTM; ++tm)
#pragma unroll
                for (int tn = 0; tn < NTN; ++tn) acc[tm][tn] = mfma4(p[tm][i], q[tn][i], acc[tm][tn]);
    };
    // a ring of DW_LEAN_RING input sets (2 + NTM registers each): the loads of step j + RING - 1 are issued in front of step j's
    // products (never under a branch: past the end of the split the last block is re-loaded), so that a set has RING - 1 steps to arrive
    constexpr int RING = DW_LEAN_RING;
    In ring[RING];
#pragma unroll
    for (int r = 0; r < RING - 1; ++r) ring[r] = load_in(min(cx.bb0 + r * cx.step, last_bb));
    int bb = cx.bb0 + (RING - 1) * cx.step;
    for (int j = 0; j < nsteps; j += RING) {
#pragma unroll
        for (int r = 0; r < RING; ++r) {
            if (j + r < nsteps) {
                ring[(r + RING - 1) % RING] = load_in(min(bb, last_bb));
                bb += cx.step;
                mma(ring[r]);
            }
        }
    }
}

// lean-WIDE dW2 (round 6; Group::lean bit 3): the unit's thin layers have up to LEANW_MAX inputs / outputs, so a1 and g2 are
// KBV = ceil(cin / 4) resp. KBG = ceil(r / 4) chained K = 4 MFMAs per tile instead of one.  The weights of those products -
// lane l of step kb: W1[feature l&15 of tile t][4 kb + (l>>4)] and W3[4 kb + (l>>4)][feature] - are staged once per job in LDS in
// lane order (`lw`: [t][kb][64] of W1, then of W3: conflict-free b32 reads), the inputs are KBV + KBG floats and NTM sign words
// per lane and 16-row step, double buffered.  What it replaces: reading a1 and g2 ([Bp][WT] columns the forward / backward
// kernels no longer write) - at cfg 5 315 MB of 1.52 GB per step.
constexpr int LWK = (LEANW_MAX + 3) / 4;        // k-blocks of the widest thin layer
// KBM: compile-time bound of both k-block counts (3 or LWK: the input registers are arrays of that size)
template <int NTM, int NTN, int KBM>
__device__ __forceinline__ void dw_leanw(f32x4 (&acc)[3][3], float (&psum)[3], const WJob& job, const LeanSrc& ls, const DwCtx& cx,
                                         const LDS_AS float* lw, int lane) {
    const int nl = cx.nl, kq = cx.kq;
    const int KBV = (job.r_cin + 3) >> 2, KBG = (job.r_r + 3) >> 2;
    const LDS_AS float* lw1 = lw + lane;
    const LDS_AS float* lw3 = lw + NTN * KBV * 64 + lane;
    float b1s[NTN];                           // (one register per tile: the bias vector is rebuilt in front of every chain)
#pragma unroll
    for (int t = 0; t < NTN; ++t) {
        const int fq = job.qcol - job.r_wcol + 16 * t + nl;
        b1s[t] = fq < job.r_h ? ls.prm[job.r_b1 + fq] : 0.f;
    }
    // per-lane offsets of a full k-block (every lane has an input) and of the last one (the lanes past the last input read
    // beyond the bound = zero); k-block kb adds 16 kb bytes through the scalar offset
    const int voff_v = (nl * ls.vs.ld + job.r_xoff + kq) * 4, voff_g = (nl * ls.ST + job.r_gcol + kq) * 4;
    const int voff_vl = 4 * (KBV - 1) + kq < job.r_cin ? voff_v : BUF_OOB, voff_gl = 4 * (KBG - 1) + kq < job.r_r ? voff_g : BUF_OOB;
    const int voff_b = 16 * (nl >> 2) + 4 * kq;
    const int bit0 = nl & 3;                  // where the lane's four mask bits sit in its word of sign bytes: bit0 + 8 i
    SrcRef gsr; gsr.p = ls.gst; gsr.ld = ls.ST; gsr.rows = 0x7fffffff;
    const int nsteps = (cx.b_end - cx.bb0 + cx.step - 1) / cx.step;
    if (nsteps <= 0) return;
    const int last_bb = cx.bb0 + (nsteps - 1) * cx.step;
    const int v_last = rows_valid(ls.vs, last_bb);
    struct In { float v[KBM], g[KBM]; unsigned s[NTM]; };
    auto load_in = [&](int BB) {
        In in;
        const __amdgpu_buffer_rsrc_t rv = rows_rsrc(ls.vs, BB, BB == last_bb ? v_last : 16), rg = rows_rsrc(gsr, BB, 16);
#pragma unroll
        for (int kb = 0; kb < KBM; ++kb) {
            in.v[kb] = kb < KBV ? as_f32(__builtin_amdgcn_raw_buffer_load_b32(rv, kb == KBV - 1 ? voff_vl : voff_v, 16 * kb, 0)) : 0.f;
            in.g[kb] = kb < KBG ? as_f32(__builtin_amdgcn_raw_buffer_load_b32(rg, kb == KBG - 1 ? voff_gl : voff_g, 16 * kb, 0)) : 0.f;
        }
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(ls.bits + ((size_t)(BB >> 4) * ls.ntiles + (job.pcol >> 4)) * 64), 0, NTM * 64, BUF_FLAGS);
#pragma unroll
        for (int t = 0; t < NTM; ++t) in.s[t] = __builtin_amdgcn_raw_buffer_load_b32(rb, voff_b + 64 * t, 0, 0);
        return in;
    };
    // one 16-row step: the operand tiles from `in` (KBV + KBG chained MFMAs per tile, the tiles' chains interleaved), then - the
    // inputs are spent - the NEXT step's loads, in flight across the NTM x NTN x 4 products
    In in = load_in(cx.bb0);
    int bb = cx.bb0;
    for (int j = 0; j < nsteps; ++j) {
        f32x4 p[NTM], q[NTN];
#pragma unroll
        for (int t = 0; t < NTN; ++t) q[t] = f32x4{b1s[t], b1s[t], b1s[t], b1s[t]};
#pragma unroll
        for (int t = 0; t < NTM; ++t) p[t] = zero4();
#pragma unroll
        for (int kb = 0; kb < KBM; ++kb) {
            if (kb < KBV) {
#pragma unroll
                for (int t = 0; t < NTN; ++t) q[t] = mfma4(in.v[kb], lw1[(t * KBV + kb) * 64], q[t]);
            }
            if (kb < KBG) {
#pragma unroll
                for (int t = 0; t < NTM; ++t) p[t] = mfma4(in.g[kb], lw3[(t * KBG + kb) * 64], p[t]);
            }
        }
#pragma unroll
        for (int t = 0; t < NTN; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) q[t][i] = relu1(q[t][i]);
#pragma unroll
        for (int t = 0; t < NTM; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = __builtin_amdgcn_sbfe((int)in.s[t], bit0 + 8 * i, 1);
                p[t][i] = as_f32((unsigned)(as_i32(p[t][i]) & m));
            }
        bb += cx.step;
        in = load_in(min(bb, last_bb));
#pragma unroll
        for (int t = 0; t < NTM; ++t) psum[t] += (p[t].x + p[t].y) + (p[t].z + p[t].w);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int tm = 0; tm < NTM; ++tm)
#pragma unroll
                for (int tn = 0; tn < NTN; ++tn) acc[tm][tn] = mfma4(p[tm][i], q[tn][i], acc[tm][tn]);
    }
}
// the staged weights of dw_leanw: NTN x KBV + NTM x KBG vectors of 64 floats; `nw` wavefronts (1: a solo job's own) share the work
__device__ __forceinline__ void stage_leanw(LDS_AS float* lw, const WJob& job, const GLOBAL_AS float* prm, int ntm, int ntn, int lane,
                                            int w, int nw) {
    const int nl = lane & 15, kq = lane >> 4;
    const int KBV = (job.r_cin + 3) >> 2, KBG = (job.r_r + 3) >> 2;
    const int n1 = ntn * KBV, n3 = ntm * KBG;
    for (int idx = w; idx < n1 + n3; idx += nw) {
        float val = 0.f;
        if (idx < n1) {
            const int t = idx / KBV, kb = idx - t * KBV, k = 4 * kb + kq;
            const int fq = job.qcol - job.r_wcol + 16 * t + nl;
            if (k < job.r_cin && fq < job.r_h) val = prm[job.r_w1 + (size_t)fq * job.r_cin + k];
        } else {
            const int i3 = idx - n1, t = i3 / KBG, kb = i3 - t * KBG, k = 4 * kb + kq;
            const int fp = job.pcol - job.r_wcol + 16 * t + nl;
            if (k < job.r_r && fp < job.r_h) val = prm[job.r_w3 + (size_t)k * job.r_h + fp];
        }
        lw[idx * 64 + lane] = val;
    }
}

// the general products: P columns [pcol, pcol + 16 NTM) x Q columns [qcol, qcol + 16 NTN) over the rows of the split (NTN = 0: the
// column sums of P only); PVEC / QVEC: a full 48-column group inside its array - one 12-byte load per lane and row
template <int NTM, int NTN, bool PVEC, bool QVEC>
__device__ __forceinline__ void dw_gen(f32x4 (&acc)[3][3], float (&psum)[3], const GenSrc& gs, const DwCtx& cx) {
    constexpr int NQ = NTN > 0 ? NTN : 1;
    struct Op { float a[4][NTM], b[4][NQ]; };
    const int kq = cx.kq;
    // loop-invariant per-lane byte offsets inside the 16-row window of a step (row 4 i + kq, the lane's column(s))
    int poff[4][PVEC ? 1 : NTM], qoff[4][QVEC ? 1 : NQ];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 4 * i + kq;
#pragma unroll
        for (int j = 0; j < (PVEC ? 1 : NTM); ++j) poff[i][j] = (row * gs.ps.ld + (PVEC ? gs.pc0 : gs.pcj[j])) * 4;
#pragma unroll
        for (int j = 0; j < (QVEC ? 1 : NQ); ++j) qoff[i][j] = (row * gs.qs.ld + (QVEC ? gs.qc0 : gs.qcj[j])) * 4;
    }
    const int nsteps = (cx.b_end - cx.bb0 + cx.step - 1) / cx.step;
    if (nsteps <= 0) return;
    const int last_bb = cx.bb0 + (nsteps - 1) * cx.step;
    const int p_last = rows_valid(gs.ps, last_bb), q_last = rows_valid(gs.qs, last_bb);
    auto load_op = [&](int BB) {
        Op o;
        const __amdgpu_buffer_rsrc_t rp = rows_rsrc(gs.ps, BB, BB == last_bb ? p_last : 16);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (PVEC) {
                const u32x3 t = __builtin_amdgcn_raw_buffer_load_b96(rp, poff[i][0], 0, 0);
                o.a[i][0] = as_f32(t.x);
                if (NTM > 1) o.a[i][1 % NTM] = as_f32(t.y);
                if (NTM > 2) o.a[i][2 % NTM] = as_f32(t.z);
            } else {
#pragma unroll
                for (int j = 0; j < NTM; ++j) o.a[i][j] = buf_f32(rp, poff[i][PVEC ? 0 : j]);
            }
        }
        if (NTN > 0) {
            const __amdgpu_buffer_rsrc_t rq = rows_rsrc(gs.qs, BB, BB == last_bb ? q_last : 16);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (QVEC) {
                    const u32x3 t = __builtin_amdgcn_raw_buffer_load_b96(rq, qoff[i][0], 0, 0);
                    o.b[i][0] = as_f32(t.x);
                    if (NQ > 1) o.b[i][1 % NQ] = as_f32(t.y);
                    if (NQ > 2) o.b[i][2 % NQ] = as_f32(t.z);
                } else {
#pragma unroll
                    for (int j = 0; j < NTN; ++j) o.b[i][j] = buf_f32(rq, qoff[i][QVEC ? 0 : j]);
                }
            }
        }
        return o;
    };
    auto mma = [&](const Op& o) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int tm = 0; tm < NTM; ++tm) {
                psum[tm] += o.a[i][tm];
#pragma unroll
                for (int tn = 0; tn < NTN; ++tn) acc[tm][tn] = mfma4(o.a[i][tm], o.b[i][tn], acc[tm][tn]);
            }
    };
    int bb = cx.bb0;
    if (bb >= cx.b_end) return;
    Op A = load_op(bb);
    while (true) {
        const int nb1 = bb + cx.step;
        const bool last1 = nb1 >= cx.b_end;
        const Op Bn = load_op(last1 ? bb : nb1);
        mma(A);
        if (last1) break;
        const int nb2 = nb1 + cx.step;
        const bool last2 = nb2 >= cx.b_end;
        A = load_op(last2 ? nb1 : nb2);
        mma(Bn);
        if (last2) break;
        bb = nb2;
    }
}

// A single-tile job on one wavefront is a chain of memory latencies (four MFMAs per 16-row block): eight blocks per step,
// all their loads in flight at once, an accumulator each (acc[3][3] has nine), added up in a fixed order behind the loop.
// What is left of the split (fewer than eight blocks) runs through dw_gen: cx.bb0 is advanced to it.
__device__ __forceinline__ void dw_solo8(f32x4 (&acc)[3][3], float (&psum)[3], const GenSrc& gs, DwCtx& cx, int ntn) {
    constexpr int NS = 8;
    const int pc = gs.pcj[0], qc = gs.qcj[0], kq = cx.kq;
    float ps8[NS];
#pragma unroll
    for (int t = 0; t < NS; ++t) ps8[t] = 0.f;
    int bb = cx.bb0;
    while (bb + 16 * NS <= cx.b_end) {
        float sa[NS][4], sq[NS][4];
#pragma unroll
        for (int t = 0; t < NS; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row_ = bb + 16 * t + 4 * i + kq;
                sa[t][i] = gs.ps.p[(size_t)min(row_, gs.ps.rows - 1) * gs.ps.ld + pc];
                sq[t][i] = ntn > 0 ? gs.qs.p[(size_t)min(row_, gs.qs.rows - 1) * gs.qs.ld + qc] : 0.f;
            }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int t = 0; t < NS; ++t) {
                ps8[t] += sa[t][i];
                acc[t / 3][t % 3] = mfma4(sa[t][i], sq[t][i], acc[t / 3][t % 3]);
            }
        bb += 16 * NS;
    }
#pragma unroll
    for (int t = 1; t < NS; ++t) {
        acc[0][0] += acc[t / 3][t % 3];
        acc[t / 3][t % 3] = zero4();
        ps8[0] += ps8[t];
    }
    psum[0] += ps8[0];
    cx.bb0 = bb;
}


// SMALL: the job list ends in single-tile jobs that share workgroups (n_small > 0); plans without them (the wave-local ones)
// run the instance that holds none of that code
// WIDE: the plan has lean-wide groups (dw_leanw); the other plans run the instances without that code
template <bool SMALL, bool WIDE>
__global__ __launch_bounds__(DW_WAVES * 64 HINT_DW_BOUND) void hint_wgrad_kernel(
    const WJob* __restrict__ jobs, int n_jobs, int n_small, int splits, ChainBlock one, const ChainBlock* __restrict__ chain,
    int grid_pb, int cb0, int WT, int ST, int d, int dc, int n_levels, int B, int Bp, int rows_per_wg, int64_t act_stride,
    int64_t a2_off, int64_t bits_a2_off, int64_t param_floats, const float* __restrict__ x, const float* __restrict__ c
#ifdef HINT_STAMPS
    , unsigned long long* dw_stamps
#endif
    ) {
#ifdef HINT_STAMPS
    unsigned long long dw_st[DW_STAMP_IDS] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    DW_STAMP(0)
    __shared__ float red[DW_WAVES][9][64][4];   // 72 KiB
    __shared__ float bred[DW_WAVES][3][16];

    // a chained launch holds grid_pb (a multiple of 8, so that the XCD mapping below holds for every
    // block) workgroups per block of the chain, of which the first n_jobs * splits have work
    // The last n_small jobs of the list are single-tile jobs (deep levels of narrow trees: an 8 x 8 matrix): eight of them share a
    // workgroup, one (job, split) per WAVEFRONT over all rows of the split - no combine, no barrier: a workgroup per such job
    // spends its time in the fixed costs (MINIBOONE d = 43: 12 000 workgroups for 60 MFLOP)
    const int n_big = n_jobs - n_small, items = n_big + (n_small + DW_WAVES - 1) / DW_WAVES;
    // grid_pb > 0: the chain's blocks one after the other; < 0 (jobs sorted longest first: narrow trees): job j of ALL blocks
    // next to each other, so that the order is longest-first over the whole launch
    int cbi, bid;
    if (grid_pb > 0) {
        cbi = (int)blockIdx.x / grid_pb;
        bid = (int)blockIdx.x - cbi * grid_pb;
    } else {
        const int n_chain = (int)gridDim.x / -grid_pb;
        const int q = (int)blockIdx.x >> 3;
        cbi = q % n_chain;
        bid = (q / n_chain) * 8 + ((int)blockIdx.x & 7);
    }
    if (bid >= items * splits) return;
    const GBlock blk = chain_block(chain, one, cbi);
    const GLOBAL_AS float* xg = (const GLOBAL_AS float*)x;
    const GLOBAL_AS float* cg = (const GLOBAL_AS float*)c;
    // (cb0: position of the launch's first block in its chain; a block with an input of its own - hint_chain_set_block_io - ran as a
    //  launch of its own: its forward left the top tape slice only in front of a fused permutation)
    const bool top = blk.perm != nullptr || (blk.x_in == nullptr && cbi + cb0 > 0);

    int item, split;
    if ((splits & 7) == 0) {          // XCD-aware: split s lives on XCD s % 8
        const int xcd = bid & 7, t = bid >> 3;
        split = xcd + 8 * (t / items);
        item = t % items;
    } else {
        split = bid / items;
        item = bid % items;
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = rfl(tid >> 6);
    const bool solo = SMALL && item >= n_big;
    const int jidx = solo ? n_big + (item - n_big) * DW_WAVES + wave : item;
    if (jidx >= n_jobs) return;         // (a spare wavefront of the last shared workgroup: that path has no barrier)
    const WJob job = jobs[jidx];
    DW_STAMP_WAIT(1)        // block pointers and job record have arrived
    const int nl = lane & 15, kq = lane >> 4;
    const int ntm = job.mw, ntn = job.nw;
    const int b_begin = split * rows_per_wg;
    const int b_end = min(Bp, b_begin + rows_per_wg);

    f32x4 acc[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = zero4();
    float psum[3] = {0.f, 0.f, 0.f};
    const bool natural = job.psrc == WSRC_G2R;       // columns in natural order (no 12-byte loads to serve)
    DwCtx cx;
    cx.nl = nl; cx.kq = kq;
    cx.step = solo ? 16 : 16 * DW_WAVES;
    cx.bb0 = b_begin + (solo ? 0 : wave * 16);
    cx.b_end = b_end;

    // The loops are instantiated per tile shape (ntm x ntn known at compile time: every MFMA of a step in ONE basic block, the
    // next step's loads in flight across it with counted waits).  With the shape a run-time value every MFMA sat under its
    // own scalar branch and behind a vmcnt(0): the step's loads were waited for in front of its first product (round 3:
    // matrix pipe busy 49 % of the kernel).
    if (natural) {
        // ---- lean dW2: both operands rebuilt from the thin layers' inputs ----
        LeanSrc ls;
        ls.prm = blk.params;
        ls.vs = wsrc(WSRC_X, job.qlevel, blk, top, xg, cg, WT, ST, d, dc, n_levels, B, Bp, act_stride, a2_off);
        ls.gst = blk.wsGST;
        ls.bits = (const GLOBAL_AS uint8_t*)blk.actA1 + bits_a2_off;           // a2 sign bytes [row tile][tile][64]
        ls.ntiles = WT >> 4; ls.ST = ST;
        if (WIDE && (job.r_cin > 4 || job.r_r > 4)) {
            // the thin layers' weights -> LDS (the combine buffer `red` is free until the rows are done): one copy per workgroup,
            // or per wavefront for the single-tile jobs that share a workgroup (no workgroup barrier on that path)
            LDS_AS float* lw = (LDS_AS float*)&red[0][0][0][0] + (solo ? wave * (9 * 64 * 4) : 0);
            stage_leanw(lw, job, blk.params, ntm, ntn, lane, solo ? 0 : wave, solo ? 1 : DW_WAVES);
            if (!solo) __syncthreads();
            const bool few = job.r_cin <= 12 && job.r_r <= 12;       // (at most three k-blocks either way: the instances with small input arrays)
            switch (ntm * 4 + ntn) {
#define DW_CASE(M_, N_) case M_ * 4 + N_: if (few) dw_leanw<M_, N_, 3>(acc, psum, job, ls, cx, lw, lane); else dw_leanw<M_, N_, LWK>(acc, psum, job, ls, cx, lw, lane); break;
                DW_CASE(1, 1) DW_CASE(1, 2) DW_CASE(1, 3) DW_CASE(2, 1) DW_CASE(2, 2) DW_CASE(2, 3) DW_CASE(3, 1) DW_CASE(3, 2)
                default: if (few) dw_leanw<3, 3, 3>(acc, psum, job, ls, cx, lw, lane); else dw_leanw<3, 3, LWK>(acc, psum, job, ls, cx, lw, lane); break;
#undef DW_CASE
            }
            if (!solo) __syncthreads();       // (the staged weights are read to the last step: `red` takes the partials below)
        } else
        switch (ntm * 4 + ntn) {
#define DW_CASE(M_, N_) case M_ * 4 + N_: dw_lean<M_, N_>(acc, psum, job, ls, cx); break;
            DW_CASE(1, 1) DW_CASE(1, 2) DW_CASE(1, 3) DW_CASE(2, 1) DW_CASE(2, 2) DW_CASE(2, 3) DW_CASE(3, 1) DW_CASE(3, 2)
            default: dw_lean<3, 3>(acc, psum, job, ls, cx); break;
#undef DW_CASE
        }
    } else {
        GenSrc gs;
        gs.ps = wsrc(job.psrc, 0, blk, top, xg, cg, WT, ST, d, dc, n_levels, B, Bp, act_stride, a2_off);
        gs.qs = wsrc(job.qsrc, job.qlevel, blk, top, xg, cg, WT, ST, d, dc, n_levels, B, Bp, act_stride, a2_off);
        // lane nl holds columns col0 + w*nl + {0..w-1} of its operand (the permutation of columns inside
        // the up-to-48-wide group is undone at write-out).  Full 48-column groups inside the array: one
        // 12-byte load per k-step; otherwise element loads clamped to the operand's last column (products
        // of clamped columns are never written).
        const int pc0 = job.pcol + ntm * nl, qc0 = job.qcol + ntn * nl;
        const bool pvec = ntm == 3 && job.pcol + 47 <= job.pmax;
        const bool qvec = ntn == 3 && job.qcol + 47 <= job.qmax;
        gs.pc0 = pc0; gs.qc0 = qc0;
#pragma unroll
        for (int j = 0; j < 3; ++j) { gs.pcj[j] = min(pc0 + j, job.pmax); gs.qcj[j] = min(qc0 + j, max(job.qmax, 0)); }
        if (SMALL && solo && ntm == 1 && ntn <= 1) dw_solo8(acc, psum, gs, cx, ntn);
        const int shape = ntm * 4 + ntn;
        if (pvec && qvec) dw_gen<3, 3, true, true>(acc, psum, gs, cx);
        else if (pvec) {
            switch (ntn) {
                case 0: dw_gen<3, 0, true, false>(acc, psum, gs, cx); break;
                case 1: dw_gen<3, 1, true, false>(acc, psum, gs, cx); break;
                case 2: dw_gen<3, 2, true, false>(acc, psum, gs, cx); break;
                default: dw_gen<3, 3, true, false>(acc, psum, gs, cx); break;
            }
        } else if (qvec) {
            switch (ntm) {
                case 1: dw_gen<1, 3, false, true>(acc, psum, gs, cx); break;
                case 2: dw_gen<2, 3, false, true>(acc, psum, gs, cx); break;
                default: dw_gen<3, 3, false, true>(acc, psum, gs, cx); break;
            }
        } else {
            switch (shape) {
#define DW_CASE(M_, N_) case M_ * 4 + N_: dw_gen<M_, N_, false, false>(acc, psum, gs, cx); break;
                DW_CASE(1, 0) DW_CASE(1, 1) DW_CASE(1, 2) DW_CASE(1, 3) DW_CASE(2, 0) DW_CASE(2, 1) DW_CASE(2, 2) DW_CASE(2, 3)
                DW_CASE(3, 0) DW_CASE(3, 1) DW_CASE(3, 2)
                default: dw_gen<3, 3, false, false>(acc, psum, gs, cx); break;
#undef DW_CASE
            }
        }
    }
    DW_STAMP_WAIT(2)        // the rows of the split are done
    if (solo) {
        // one (job, split) per wavefront: straight to the split's slab
        GLOBAL_AS float* slab = blk.wsSlab + (size_t)split * param_floats;
#pragma unroll
        for (int tm = 0; tm < 3; ++tm)
#pragma unroll
            for (int tn = 0; tn < 3; ++tn) {
                if (tm >= ntm || tn >= ntn) continue;
                const int n = natural ? 16 * tn + nl : ntn * nl + tn;              // undo the column permutation of the loads
                if (n >= job.N) continue;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int m = natural ? 16 * tm + 4 * kq + i : ntm * (4 * kq + i) + tm;
                    if (m < job.M) slab[job.wofs + (size_t)m * job.ldo + n] = acc[tm][tn][i];
                }
            }
        if (job.bofs >= 0) {
#pragma unroll
            for (int tm = 0; tm < 3; ++tm) {
                if (tm >= ntm) continue;
                float v = psum[tm];
                v += __shfl_xor(v, 16, 64);
                v += __shfl_xor(v, 32, 64);
                const int m = natural ? 16 * tm + nl : ntm * nl + tm;
                if (kq == 0 && m < job.M) slab[job.bofs + m] = v;
            }
        }
        DW_STAMP_WAIT(5)
        DW_STAMP_FLUSH()
        return;
    }
    // combine the wavefronts (fixed order)
#pragma unroll
    for (int tm = 0; tm < 3; ++tm)
#pragma unroll
        for (int tn = 0; tn < 3; ++tn) *(f32x4*)&red[wave][tm * 3 + tn][lane][0] = acc[tm][tn];
    if (job.bofs >= 0) {
#pragma unroll
        for (int tm = 0; tm < 3; ++tm) {
            float v = psum[tm];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (kq == 0) bred[wave][tm][nl] = v;
        }
    }
    DW_STAMP(3)
    __syncthreads();
    DW_STAMP(4)             // every wavefront of the workgroup has left its partials
    GLOBAL_AS float* slab = blk.wsSlab + (size_t)split * param_floats;
    for (int idx = tid; idx < 9 * 64; idx += DW_WAVES * 64) {
        const int t = idx >> 6, l = idx & 63;
        const int tm = t / 3, tn = t - 3 * tm;
        if (tm >= ntm || tn >= ntn) continue;
        f32x4 v = *(f32x4*)&red[0][t][l][0];
#pragma unroll
        for (int w = 1; w < DW_WAVES; ++w) v += *(f32x4*)&red[w][t][l][0];
        const int n = natural ? 16 * tn + (l & 15) : ntn * (l & 15) + tn;          // undo the column permutation of the loads
        if (n >= job.N) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = natural ? 16 * tm + 4 * (l >> 4) + i : ntm * (4 * (l >> 4) + i) + tm;
            if (m < job.M) slab[job.wofs + (size_t)m * job.ldo + n] = v[i];
        }
    }
    if (job.bofs >= 0 && tid < 48) {
        const int tm = tid >> 4, l = tid & 15;
        const int m = natural ? 16 * tm + l : ntm * l + tm;
        if (tm < ntm && m < job.M) {
            float v = bred[0][tm][l];
#pragma unroll
            for (int w = 1; w < DW_WAVES; ++w) v += bred[w][tm][l];
            slab[job.bofs + m] = v;
        }
    }
    DW_STAMP_WAIT(5)
    DW_STAMP_FLUSH()
}

// g[i] (+)= sum over the splits' slabs, in split order, for every real parameter element i (the
// padding between tensors stays untouched when accumulating and is cleared otherwise)
__global__ __launch_bounds__(256) void hint_wreduce_kernel(ChainBlock one, const ChainBlock* __restrict__ chain,
                                                           const uint8_t* __restrict__ real, int64_t param_floats,
                                                           int splits, int accumulate, int blocks_pb,
                                                           const int32_t* __restrict__ twmap, int tw_floats,
                                                           int64_t thin_slab_off, int thin_slabs, int thin_blocks, AdamFuse ad) {
    // ad.p != nullptr (hint_chain_backward_adam): the gradient goes straight into the clamp + Adam step of its parameter - the
    // gradient arena is neither read nor written (it stays zero, as hint_adam_step's zero_grads leaves it)
    const int per_block = blocks_pb + thin_blocks;
    const int cbi = (int)blockIdx.x / per_block;
    const int bid = (int)blockIdx.x - cbi * per_block;
    const GBlock blk = chain_block(chain, one, cbi);
    const float* slab = (const float*)blk.wsSlab;
    float* g = (float*)blk.gparams;
    const bool fuse = ad.p != nullptr;
    const int64_t aoff = fuse ? (const float*)blk.params - ad.p : 0;     // the block's slice of the arenas
    float lr_t = 0.f, bc2 = 0.f;
    if (fuse) { lr_t = ad.st[3]; bc2 = ad.st[4]; }
    if (bid >= blocks_pb) {
        // first-layer gradients of a lean plan: one slab per workgroup of the backward kernel.  32 slab elements per
        // block, 8 threads each: thread q adds slabs q, q+8, .. (eight loads in flight), the eight partials are
        // added in order - a fixed summation order whatever the launch
        __shared__ float part[8][32];
        const int tl = (int)threadIdx.x & 31, q = (int)threadIdx.x >> 5;
        const float* tw = slab + thin_slab_off;
        for (int t0 = (bid - blocks_pb) * 32; t0 < tw_floats; t0 += thin_blocks * 32) {
            const int t = t0 + tl;
            float s = 0.f;
            if (t < tw_floats) {
                int w = q;
                for (; w + 56 < thin_slabs; w += 64) {
                    float v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = tw[(size_t)(w + 8 * u) * tw_floats + t];
#pragma unroll
                    for (int u = 0; u < 8; ++u) s += v[u];
                }
                for (; w < thin_slabs; w += 8) s += tw[(size_t)w * tw_floats + t];
            }
            part[q][tl] = s;
            __syncthreads();
            if (q == 0 && t < tw_floats) {
                const int dst = twmap[t];
                if (dst >= 0) {
                    float r = part[0][tl];
#pragma unroll
                    for (int u = 1; u < 8; ++u) r += part[u][tl];
                    if (fuse) adam_update(ad.p[aoff + dst], ad.m[aoff + dst], ad.v[aoff + dst], r, lr_t, ad.b1, ad.b2, bc2, ad.eps, ad.wd,
                                          ad.gscale, ad.gclamp);
                    else g[dst] = accumulate ? g[dst] + r : r;
                }
            }
            __syncthreads();
        }
        return;
    }
    const int64_t n4 = param_floats >> 2;        // param_floats is a multiple of 4 (hint_plan_param_floats)
    for (int64_t i4 = (int64_t)bid * 256 + threadIdx.x; i4 < n4; i4 += (int64_t)blocks_pb * 256) {
        const uchar4 rl = ((const uchar4*)real)[i4];      // 1: from part B's slabs; 2: from the backward kernel's (above); 0: padding
        // (a plain loop: eight slabs' loads in flight at a time - round 6 - gained 0.8 us in the stand-alone reduction and LOST 1.5 us of
        //  the captured cfg 2 step with the optimizer folded in, three A/B pairs: NOTES.md section 10)
        f32x4 s = zero4();
        for (int sp = 0; sp < splits; ++sp) s += ((const f32x4*)(slab + (size_t)sp * param_floats))[i4];
        if (fuse) {
            f32x4* pp = (f32x4*)(ad.p + aoff) + i4; f32x4* pm = (f32x4*)(ad.m + aoff) + i4; f32x4* pv = (f32x4*)(ad.v + aoff) + i4;
            f32x4 p4 = *pp, m4 = *pm, v4 = *pv;
            const unsigned char r4[4] = {rl.x, rl.y, rl.z, rl.w};
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (r4[j] == 1) {
                    float pj = p4[j], mj = m4[j], vj = v4[j];
                    adam_update(pj, mj, vj, 0.f + s[j], lr_t, ad.b1, ad.b2, bc2, ad.eps, ad.wd, ad.gscale, ad.gclamp);
                    p4[j] = pj; m4[j] = mj; v4[j] = vj;
                }
            if (rl.x != 2 && rl.y != 2 && rl.z != 2 && rl.w != 2) {
                *pp = p4; *pm = m4; *pv = v4;
            } else {                                    // (elements of the other path are not touched here)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (r4[j] == 1) { ((float*)pp)[j] = p4[j]; ((float*)pm)[j] = m4[j]; ((float*)pv)[j] = v4[j]; }
            }
            continue;
        }
        f32x4 o = accumulate ? ((const f32x4*)g)[i4] : zero4();
        if (rl.x == 1) o.x += s.x;
        if (rl.y == 1) o.y += s.y;
        if (rl.z == 1) o.z += s.z;
        if (rl.w == 1) o.w += s.w;
        if (rl.x != 2 && rl.y != 2 && rl.z != 2 && rl.w != 2) {
            ((f32x4*)g)[i4] = o;
        } else {                                        // (elements of the other path are not touched here)
            float* gp = g + 4 * i4;
            if (rl.x != 2) gp[0] = o.x;
            if (rl.y != 2) gp[1] = o.y;
            if (rl.z != 2) gp[2] = o.z;
            if (rl.w != 2) gp[3] = o.w;
        }
    }
}

namespace hint {

hipError_t launch_wgrad(const WJob* jobs, int n_jobs, int n_small, int splits, const ChainBlock& one, const ChainBlock* chain,
                        int n_chain, int cb0, int WT, int ST, int d, int dc, int n_levels, int B, int Bp, int rows_per_wg,
                        int64_t act_stride, int64_t a2_off, int64_t bits_a2_off, int64_t param_floats, const float* x,
                        const float* c, const uint8_t* real, int accumulate, const int32_t* twmap, int tw_floats,
                        int64_t thin_slab_off, int thin_slabs, int num_cu, const AdamFuse* adam, bool wide, hipStream_t stream) {
    const bool interleave = n_small < 0;       // (flag in the sign: the planner sorted the jobs)
    if (interleave) n_small = -n_small - 1;
#ifdef HINT_DW_SOLO_ALL
    n_small = n_jobs;                          // (experiment: one (job, split) per wavefront for every job)
#endif
    const int used = (n_jobs - n_small + (n_small + DW_WAVES - 1) / DW_WAVES) * splits;
    const int grid_pb = n_chain > 1 ? (used + 7) / 8 * 8 : used;
    if (used > 0) {
        const int gpb = (interleave && n_chain > 1 && (splits & 7) == 0) ? -grid_pb : grid_pb;
#ifdef HINT_STAMPS
#define HINT_DW_LAUNCH(S_, W_) hipLaunchKernelGGL((hint_wgrad_kernel<S_, W_>), dim3(grid_pb * n_chain), dim3(DW_WAVES * 64), 0, stream, jobs, n_jobs, n_small, splits, \
                               one, chain, gpb, cb0, WT, ST, d, dc, n_levels, B, Bp, rows_per_wg, act_stride, a2_off, bits_a2_off, param_floats, x, c, h_dw_stamps)
#else
#define HINT_DW_LAUNCH(S_, W_) hipLaunchKernelGGL((hint_wgrad_kernel<S_, W_>), dim3(grid_pb * n_chain), dim3(DW_WAVES * 64), 0, stream, jobs, n_jobs, n_small, splits, \
                               one, chain, gpb, cb0, WT, ST, d, dc, n_levels, B, Bp, rows_per_wg, act_stride, a2_off, bits_a2_off, param_floats, x, c)
#endif
        if (wide) { if (n_small > 0) HINT_DW_LAUNCH(true, true); else HINT_DW_LAUNCH(false, true); }
        else if (n_small > 0) HINT_DW_LAUNCH(true, false);
        else HINT_DW_LAUNCH(false, false);
#undef HINT_DW_LAUNCH
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    int blocks_pb = (int)((param_floats / 4 + 255) / 256);
    const int cap = num_cu * 4 / (n_chain > 0 ? n_chain : 1);
    blocks_pb = blocks_pb < 1 ? 1 : (blocks_pb > cap && cap >= 1 ? cap : blocks_pb);
    const int thin_blocks = twmap != nullptr ? (tw_floats + 31) / 32 : 0;
    hipLaunchKernelGGL(hint_wreduce_kernel, dim3((blocks_pb + thin_blocks) * n_chain), dim3(256), 0, stream, one, chain, real,
                       param_floats, splits, accumulate, blocks_pb, twmap, tw_floats, thin_slab_off, thin_slabs, thin_blocks,
                       adam ? *adam : AdamFuse{});
    return hipGetLastError();
}

}  // namespace hint
