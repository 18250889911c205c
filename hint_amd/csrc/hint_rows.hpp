// The GEMM engine of the block kernels: one wavefront walks a contiguous range of a group's
// fragment tiles (a fragment tile = 16 output features x 16 batch rows of one unit) and, per tile,
// streams that tile's "row" of packed weight tiles from L2 through a register ring:
//
//   phase    main steps (A = weight tile kb)            B operand                tail steps
//   K_L1     KB1 tiles of W1   [h x cin], interleaved   lane tile / condition    -
//   K_L2     NT  tiles of W2   [h x h],   blocked       a1 fragment tiles (LDS)  RT tiles of W3 (B = the a2 tile just
//                                                                                 finished, in registers) [+ RT bias tiles]
//                                                                                 accumulated in the wavefront's LDS slab
//   K_G2     RT  tiles of W3^T [h x r],   interleaved   coupling gradients (LDS) -
//   K_G1     NT  tiles of W2^T [h x h],   blocked       g2 fragment tiles (LDS)  CT tiles of W1^T (B = the g1 tile just finished)
//
// Every product is transposed (out^T = W * act^T), so the accumulator of the main steps - lane l
// holds features 4*(l>>4)+i of batch row l&15 - is exactly the B operand of the tail steps, and
// what a wavefront stores as a fragment tile (ds_write_b128 at lane*16) is what another one reads
// as its B operand (ds_read_b128): no transposes, no per-layer barrier inside a unit.
// The ring holds the next RING weight tiles (1 KiB, one global_load_dwordx4 per lane, fully
// coalesced); a second iterator runs RING steps ahead of the consuming one, across tiles and units.
#pragma once
#include "hint_device.hpp"

namespace hint {

enum { K_L1 = 0, K_L2 = 1, K_G2 = 2, K_G1 = 3 };
#ifndef HINT_RING
#define HINT_RING 4
#endif
constexpr int RING = HINT_RING;

// wave-uniform position in the stream of weight tiles
struct RowIt {
    int t, kb;                  // fragment tile, step inside its row
    int n1, n2, n3;             // main / tail / bias-tile steps of the row
    int base1, base2, base3, stride2;   // packed tile index of step kb: see row_tile()
    int tile0, NT;              // fragment tiles [tile0, tile0 + NT) belong to the iterator's current unit
    int ui;                     // that unit
};
__device__ __forceinline__ int row_tile(const RowIt& it) {
    if (it.kb < it.n1) return it.base1 + it.kb;
    if (it.kb < it.n1 + it.n2) return it.base2 + (it.kb - it.n1) * it.stride2;
    return it.base3 + (it.kb - it.n1 - it.n2);
}

template <int KIND>
__device__ __forceinline__ void row_setup(RowIt& it, const Tables& T, const GroupU& g, int t, bool force) {
    it.t = t;
    it.kb = 0;
    if (force || t >= it.tile0 + it.NT) {
        it.ui = g.unit_begin + lds_u16(T.tmap + g.tmap_begin + t);
        const LDS_AS int32_t* u = (const LDS_AS int32_t*)(T.units + it.ui);
        it.tile0 = lds_i32(u + 10);
        it.NT = lds_i32(u + 12);
    }
    const LDS_AS int32_t* u = (const LDS_AS int32_t*)(T.units + it.ui);
    const int nt = t - it.tile0;
    if (KIND == K_L1) {
        const int KB1 = lds_i32(u + 13);
        it.n1 = KB1; it.n2 = 0; it.n3 = 0;
        it.base1 = lds_i32(u + 0) + nt * KB1;
        it.base2 = it.base3 = it.stride2 = 0;
    } else if (KIND == K_L2) {
        const int RT = lds_i32(u + 14);
        it.n1 = it.NT; it.n2 = RT; it.n3 = nt == 0 ? RT : 0;
        it.base1 = lds_i32(u + 1) + nt * it.NT;
        it.base2 = lds_i32(u + 2) + nt; it.stride2 = it.NT;
        it.base3 = lds_i32(u + 2) + RT * it.NT;          // the unit's b3 "bias tiles" follow its W3 tiles
    } else if (KIND == K_G2) {
        const int RT = lds_i32(u + 14);
        it.n1 = RT; it.n2 = 0; it.n3 = 0;
        it.base1 = lds_i32(u + 3) + nt * RT;
        it.base2 = it.base3 = it.stride2 = 0;
    } else {
        const int CT = lds_i32(u + 13);
        it.n1 = it.NT; it.n2 = CT; it.n3 = 0;
        it.base1 = lds_i32(u + 4) + nt * it.NT;
        it.base2 = lds_i32(u + 5) + nt; it.stride2 = it.NT;
        it.base3 = 0;
    }
}

// advance to the next step; returns false when the range [.., t1) is exhausted
template <int KIND>
__device__ __forceinline__ bool row_advance(RowIt& it, const Tables& T, const GroupU& g, int t1) {
    ++it.kb;
    if (it.kb < it.n1 + it.n2 + it.n3) return true;
    if (it.t + 1 >= t1) { it.t = t1; return false; }
    row_setup<KIND>(it, T, g, it.t + 1, false);
    return true;
}

// What the phases need besides the tables: pointers and strides of the workgroup's LDS buffers and
// of the block's global arrays.
struct PhaseCtx {
    const float* packed;        // packed weights of the block (global)
    float* abuf;                // fragment tiles of the group (LDS): a1 (forward), g2 (backward)
    const float* xs;            // lane tile [16][xld] (LDS)
    const float* cs;            // condition tile [16][cld] (LDS)
    const float* gst;           // coupling gradients [16][gld] (LDS, backward)
    float* slab;                // partial-sum slabs of the tail products (LDS)
    float* out1;                // global [Bp][WT]: a1 (K_L1, forward training), g2 (K_G2), g1 (K_G1); a2 for K_L2
    const float* mask;          // global [Bp][WT]: a2 (K_G2) / a1 (K_G1): relu'() of the forward activation
    int xld, cld, gld, WT, row0;
    bool store;                 // write out1 (training forward / always in the backward pass)
};

// One phase of one wavefront: fragment tiles [t0, t1) of group g; `slabp` is where the wavefront's
// first tail slab goes (K_L2 / K_G1).
template <int KIND>
__device__ __forceinline__ void run_phase(const PhaseCtx& c, const Tables& T, const GroupU& g, int t0, int t1,
                                          float* slabp, int lane) {
    if (t0 >= t1) return;
    const int m = lane & 15, kq = lane >> 4;
    const f32x4* wp4 = (const f32x4*)c.packed + lane;
    const f32x4* abuf4 = (const f32x4*)c.abuf + lane;

    RowIt pit;
    pit.tile0 = 0; pit.NT = 0; pit.ui = 0;
    row_setup<KIND>(pit, T, g, t0, true);
    RowIt cit = pit;
    UnitU U = load_unit(T.units + cit.ui);
    bool pmore = true;
    f32x4 ring[RING];
#pragma unroll
    for (int j = 0; j < RING; ++j) {
        ring[j] = zero4();
        if (pmore) {
            ring[j] = wp4[(size_t)row_tile(pit) * 64];
            pmore = row_advance<KIND>(pit, T, g, t1);
        }
    }
    f32x4 acc0 = zero4(), acc1 = zero4();
    bool slice_first = true;    // no row of the wavefront's current slice has reached its slab yet
    f32x4 aux = zero4();        // bias of the row (K_L1, K_L2) / the forward activation whose sign masks it (K_G2, K_G1)
    f32x4 act = zero4();        // the finished main tile (B operand of the tail steps)
    f32x4 bnext = zero4();      // next B fragment (K_L2, K_G1)

#define HINT_ROW_BEGIN()                                                                                      \
    {                                                                                                         \
        const int nt_ = cit.t - U.tile0;                                                                      \
        const size_t go_ = (size_t)(c.row0 + m) * c.WT + U.wcol + 16 * nt_ + 4 * kq;                           \
        if (KIND == K_L1) aux = *(const f32x4*)(c.packed + U.bias1 + 16 * nt_ + 4 * kq);                       \
        else if (KIND == K_L2) aux = *(const f32x4*)(c.packed + U.bias2 + 16 * nt_ + 4 * kq);                  \
        else aux = *(const f32x4*)(c.mask + go_);                                                             \
        if (KIND == K_L2 || KIND == K_G1) bnext = abuf4[(size_t)U.tile0 * 64];                                \
        acc0 = zero4(); acc1 = zero4();                                                                       \
    }
    HINT_ROW_BEGIN()

    bool more = true;
    while (more) {
#pragma unroll
        for (int j = 0; j < RING; ++j) {
            if (more) {
                const f32x4 w4 = ring[j];
                const int kb = cit.kb;
                if (kb < cit.n1) {
                    // ---- main step ----
                    if (KIND == K_L2 || KIND == K_G1) {
                        const f32x4 b4 = bnext;
                        const int kn = kb + 1 < cit.n1 ? kb + 1 : kb;
                        bnext = abuf4[(size_t)(U.tile0 + kn) * 64];
                        acc0 = mfma4(w4.x, b4.x, acc0);
                        acc1 = mfma4(w4.y, b4.y, acc1);
                        acc0 = mfma4(w4.z, b4.z, acc0);
                        acc1 = mfma4(w4.w, b4.w, acc1);
                    } else {
                        // B read element-wise (interleaved k order: MFMA i covers k = 16 kb + 4 i + kq)
                        const int K = KIND == K_L1 ? U.cin : U.r;
                        const int rem = K - 16 * kb;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            if (4 * i < rem) {
                                const int k = 16 * kb + 4 * i + kq;
                                float b = 0.f;
                                if (KIND == K_L1) {
                                    if (k < U.ku) b = c.xs[m * c.xld + U.xoff + k];
                                    else if (k < U.cin) b = c.cs[m * c.cld + (k - U.ku)];
                                } else {
                                    if (k < U.r) b = c.gst[m * c.gld + U.lcol + k];
                                }
                                if (i & 1) acc1 = mfma4(w4[i], b, acc1);
                                else acc0 = mfma4(w4[i], b, acc0);
                            }
                        }
                    }
                } else {
                    // ---- tail step q: slab[q] (+)= Wtail(q, this tile) * act, the K-split partial of the thin
                    //      product, kept in the wavefront's own LDS slab (compact: quad q4 = 4 q + kq holds
                    //      features 4 q4 .. +3 of rows m); behind the tail steps of a unit's first tile come
                    //      the last layer's bias tiles, added the same way ----
                    const bool is_bias = kb >= cit.n1 + cit.n2;
                    const int q = kb - cit.n1 - (is_bias ? cit.n2 : 0);
                    f32x4 s = w4;
                    if (!is_bias) {
                        s = mfma4(w4.x, act.x, zero4());
                        s = mfma4(w4.y, act.y, s);
                        s = mfma4(w4.z, act.z, s);
                        s = mfma4(w4.w, act.w, s);
                    }
                    const int W = KIND == K_L2 ? U.r : U.cin;
                    const int q4 = 4 * q + kq;
                    if (q4 < ((W + 3) >> 2)) {
                        f32x4* sp = (f32x4*)(slabp + (q4 * 16 + m) * 4);
                        if (is_bias || !slice_first) s += *sp;
                        *sp = s;
                    }
                }
                // ---- refill the slot RING steps ahead ----
                if (pmore) {
                    ring[j] = wp4[(size_t)row_tile(pit) * 64];
                    pmore = row_advance<KIND>(pit, T, g, t1);
                }
                // ---- consumer bookkeeping ----
                ++cit.kb;
                if (cit.kb == cit.n1) {
                    // main tile finished: epilogue
                    const int nt = cit.t - U.tile0;
                    f32x4 v = acc0 + acc1;
                    if (KIND == K_L1 || KIND == K_L2) {
                        v += aux;
                        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                    } else {
                        v.x = aux.x > 0.f ? v.x : 0.f; v.y = aux.y > 0.f ? v.y : 0.f;
                        v.z = aux.z > 0.f ? v.z : 0.f; v.w = aux.w > 0.f ? v.w : 0.f;
                    }
                    act = v;
                    if (KIND == K_L1 || KIND == K_G2) ((f32x4*)c.abuf)[(size_t)cit.t * 64 + lane] = v;
                    if (c.store) {
                        float* o = c.out1 + (size_t)(c.row0 + m) * c.WT + U.wcol + 16 * nt + 4 * kq;
                        if (KIND == K_L1 || KIND == K_L2) __builtin_nontemporal_store(v, (f32x4*)o);
                        else *(f32x4*)o = v;
                    }
                }
                if (cit.kb == cit.n1 + cit.n2 + cit.n3) {
                    // row finished
                    const int tn = cit.t + 1;
                    const bool unit_end = tn >= U.tile0 + U.NT;
                    if (KIND == K_L2 || KIND == K_G1) {
                        slice_first = false;
                        if (tn >= t1 || unit_end) {      // the wavefront's slice of this unit ends: next slab
                            const int W = KIND == K_L2 ? U.r : U.cin;
                            slabp += 64 * ((W + 3) >> 2);
                            slice_first = true;
                        }
                    }
                    if (tn >= t1) {
                        more = false;
                    } else {
                        row_setup<KIND>(cit, T, g, tn, false);
                        if (unit_end) U = load_unit(T.units + cit.ui);
                        HINT_ROW_BEGIN()
                    }
                }
            }
        }
    }
#undef HINT_ROW_BEGIN
}

}  // namespace hint
