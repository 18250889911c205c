// Wave-local ("WL") block kernels for narrow trees: every unit of the block has 1..4 inputs, at most 4 outputs and no
// condition (all of POWER d = 6 and GAS d = 8, i.e. BASELINE configs 1-3).  Same rows, same packed weights, same tape
// and workspace as the general kernels (hint_fwd.hip / hint_bwd.hip), but everything that is not the h x h product
// has moved INTO the wavefronts, so that a tree level costs ONE workgroup barrier instead of three:
//
//   * the thin layers are never a phase of their own: the B operand of k-block kb - a1 = relu(W1 v + b1) forward,
//     g2 = relu'(a2) (W3^T g_st) backward, four features per lane - is computed by the wavefront that needs it, on the
//     vector ALU, right in front of the k-block's MFMAs (<= 4 inputs: 16 FMAs), from the block's thin vectors that the
//     workgroup keeps in LDS (double buffered: the next block's are fetched while this one runs);
//   * the thin product BEHIND the h x h layer (W3 a2 forward, W1^T g1 backward: <= 4 outputs) is 16 FMAs per finished
//     tile and lane plus one xor-butterfly over the four lane groups per (unit, wavefront); its K-split partial goes to
//     the (unit, wavefront) slab in LDS - the only thing that crosses wavefronts, hence the one barrier;
//   * the element-wise coupling, the log-det sum, the fixed permutation between blocks and (backward) the scatter of
//     the g_v partials are done by EVERY wavefront for itself on a private copy of the 16 x d lane tile in LDS (the
//     transformed lanes shared out over the four lane groups of the wavefront): no second and third barrier, no
//     wavefront waiting for another one's atanf;
//   * the matrix pipe sees the h x h products only (and, backward, the 4 MFMAs per tile of the first-layer weight
//     gradient, whose operand - the g1 tile - exists nowhere else).
// The weight stream (fragment tiles from L2, register ring of hint_rows.hpp) is handed from a wavefront's last row of
// a group to its first row of the next group and of the next block.
#pragma once
#include "hint_rows.hpp"

namespace hint {

__device__ __forceinline__ f32x4 fma4(const f32x4 w, float s, const f32x4 a) {
    return f32x4{fmaf(w.x, s, a.x), fmaf(w.y, s, a.y), fmaf(w.z, s, a.z), fmaf(w.w, s, a.w)};
}
__device__ __forceinline__ float dot4(const f32x4 w, const f32x4 v, float a) {
    return fmaf(w.w, v.w, fmaf(w.z, v.z, fmaf(w.y, v.y, fmaf(w.x, v.x, a))));
}
__device__ __forceinline__ f32x4 relu4(const f32x4 v) { return f32x4{fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)}; }
// relu of a matrix-pipe result in ONE instruction per element: fmaxf() of a value the compiler cannot prove canonical is
// canonicalise + max (signalling NaNs); v_med3_f32(x, 0, 3e38) needs neither
__device__ __forceinline__ f32x4 relu4m(const f32x4 v) {
    return f32x4{__builtin_amdgcn_fmed3f(v.x, 0.f, 3.0e38f), __builtin_amdgcn_fmed3f(v.y, 0.f, 3.0e38f),
                 __builtin_amdgcn_fmed3f(v.z, 0.f, 3.0e38f), __builtin_amdgcn_fmed3f(v.w, 0.f, 3.0e38f)};
}
// sum over the four lane groups (lanes l, l^16, l^32, l^48): every lane ends with the same bits
__device__ __forceinline__ float kq_sum(float v) {
    // v_permlane32_swap: upper half of the first <-> lower half of the second; v_permlane16_swap: odd rows of the first <->
    // even rows of the second (gfx950): two vector-ALU steps instead of two trips through the LDS crossbar
    const unsigned u = (unsigned)__float_as_int(v);
    const auto h = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    const float a = __int_as_float((int)h[0]) + __int_as_float((int)h[1]);
    const unsigned ua = (unsigned)__float_as_int(a);
    const auto q = __builtin_amdgcn_permlane16_swap(ua, ua, false, false);
    return __int_as_float((int)q[0]) + __int_as_float((int)q[1]);
}

// The fixed d x d matrix between two blocks (d <= 16: one 16-column tile) on a wavefront's PRIVATE lane tile, on the matrix pipe:
// out[r][j] = sum_k in[r][k] M(k, j), M(k, j) = w[k d + j] (x' = x W) or, TRANS, w[j d + k] (x = x' W^T, g_x = g_x' W^T), transposed
// like every product here (out^T = M^T in^T): lane (m, kq) supplies M(4 t + kq, m) and in[m][4 t + kq] to step t and ends with
// columns 4 kq .. + 3 of row m.  All operand reads are in flight together, then ceil(d / 4) dependent MFMAs: as scalar code
// (perm_dot: per element d dependent FMAs with two LDS reads each, two passes over the 16 x d tile) this cost every block of the
// wave-local kernels ~1.5 k cycles in every wavefront - round 5 stamps.  k ascends inside an MFMA and across the steps, as it did.
template <bool TRANS, typename WP>
__device__ __forceinline__ void wl_perm(float* dst, const float* src, int ld, WP w, int d, int lane) {
    const int m = lane & 15, kq = lane >> 4;
    float av[4], bv[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int k = 4 * t + kq;
        const int kc = k < d ? k : 0, mc = m < d ? m : 0;
        av[t] = TRANS ? w[mc * d + kc] : w[kc * d + mc];
        bv[t] = src[m * ld + kc];
    }
    f32x4 acc = zero4();
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        if (4 * t < d) {        // (wave-uniform)
            const bool ok = 4 * t + kq < d;
            acc = mfma4((ok && m < d) ? av[t] : 0.f, ok ? bv[t] : 0.f, acc);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (4 * kq + i < d) dst[m * ld + 4 * kq + i] = acc[i];
}

// A wavefront's share of a [16, d] tile -> a row-major [B, d] array: every wavefront holds the whole tile (its private copy), so
// each stores 1 / nw of it - one pass of one store instruction per wavefront instead of two passes by ONE wavefront that the
// other seven then wait for at the level's barrier (round 5 stamps: + 0.9 k cycles on the writing wavefront).
__device__ __forceinline__ void wl_store_share(float* __restrict__ dst, const float* tile, int ld, int d, float inv_d, int nvalid,
                                               int wave, int nw, int lane) {
    const int per = (ROWS * d + nw - 1) / nw;        // <= 64 for d <= 16 on >= 4 wavefronts
    const int i = wave * per + lane;
    if (lane < per && i < nvalid) { const int r = fdiv(i, inv_d); dst[i] = tile[r * ld + (i - r * d)]; }
}

typedef unsigned wl_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 wl_as_f32x4(wl_u32x4 v) { return __builtin_bit_cast(f32x4, v); }
// relu'(.) from four sign bits: bit i sign-extended to 0 / -1 and and-ed onto element i (two instructions per element)
__device__ __forceinline__ void wl_mask_by_bits(f32x4& v, int bits) {
    const float e[4] = {v.x, v.y, v.z, v.w};
    float o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = __int_as_float(__float_as_int(e[i]) & __builtin_amdgcn_sbfe(bits, i, 1));
    v = f32x4{o[0], o[1], o[2], o[3]};
}

// The first-layer pre-activation of features 4 kq .. +3 of a tile (vector layout: five float4 per tile and lane group,
// inputs 0..3 then the bias; inputs beyond cin are zero vectors): bias first, then the inputs in order - the order of
// the MFMA chain with which part B rebuilds a1 (hint_wgrad.hip), and the same expression forward and backward, so
// that the ReLU decisions agree bit for bit.
__device__ __forceinline__ f32x4 wl_layer1(const LDS_AS f32x4* q, const float (&vin)[4]) {
    const f32x4 qb = q[16], q0 = q[0], q1 = q[4], q2 = q[8], q3 = q[12];
    return fma4(q3, vin[3], fma4(q2, vin[2], fma4(q1, vin[1], fma4(q0, vin[0], qb))));
}

// NR = 1 or 2 adjacent 16-row tiles per workgroup ("row pair"): both ride the SAME weight stream - every fragment tile
// fetched feeds two MFMAs instead of one (the stream from L2 / L1, not the matrix pipe, is what bounds the k-loop of a
// single tile), and the latency chains of the two tiles' element-wise phases interleave in every wavefront.  Used when
// the batch has more row tiles than the chip has CUs.
struct WlCtx {
    const GLOBAL_AS float* pk;           // packed weights of the block
    const GLOBAL_AS float* pk_next;      // ... of the block worked on next (the last row of a block hands the ring over)
    const GLOBAL_AS uint8_t* bits[2];    // backward: a2 sign bytes of the row tiles (this block, the next one)
    const GLOBAL_AS uint8_t* bits_next[2];
    const LDS_AS int32_t* lrecs;         // this direction's row records, staged in LDS at kernel start (16 ints each)
    const LDS_AS float* par;             // the block's staged thin vectors and biases
    LDS_AS float* slab;                  // the group's slab set of row tile 0 (tile 1: + slab_h)
    const LDS_AS float* xs[2];           // private lane tiles: the inputs of the level's first layers
    const LDS_AS float* gst[2];          // backward: private coupling gradients [16][gld]
    LDS_AS float* scratch;               // backward: private fragment tile (first-layer weight gradient)
    GLOBAL_AS float* a2[2];              // training forward: the row tiles' rows of the [Bp][WT] array
    GLOBAL_AS uint8_t* bits_out[2];      // training forward: their a2 sign bytes
    GLOBAL_AS float* tw;                 // backward: this workgroup's first-layer gradient slab
    int xld, gld, WT, slab_h;
    int sid, sid0;                       // diagnostic builds: stamp id base of the wavefront's current row / of the group
    bool train, first_tile;
};

// (through buffer descriptors: the tile index is wave-uniform, so the scalar unit does the address arithmetic - offset operand of
//  the load - and the per-lane part, lane * 16, is loop invariant.  Vector-ALU instructions do not hide under the matrix pipe on
//  gfx950 - tools/mfma_valu_bench.hip - and a 64-bit per-lane address cost one per tile and step)
template <int KIND, int NR>
__device__ __forceinline__ void wl_load(f32x4 (&dst)[NEL], const GLOBAL_AS float* pk, const GLOBAL_AS uint8_t* const (&bits)[2],
                                        const RowU& r, int kb, const LaneOff& lo) {
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)pk, 0, -1, 0x00020000);
#pragma unroll
    for (int j = 0; j < NTT; ++j) {
        const int jj = j < r.ntt ? j : r.ntt - 1;
        dst[j] = wl_as_f32x4(__builtin_amdgcn_raw_buffer_load_b128(rw, (int)lo.w, (r.base1 + jj * r.n1 + kb) * 1024, 0));
    }
    if (KIND == K_BWD) {
        const int kc = kb < r.n1 ? kb : r.n1 - 1;
        const int o = ((r.wcol >> 4) + kc) * 64;
        const __amdgpu_buffer_rsrc_t rb0 = __builtin_amdgcn_make_buffer_rsrc((void*)bits[0], 0, -1, 0x00020000);
        dst[NTT].x = __int_as_float((int)__builtin_amdgcn_raw_buffer_load_b8(rb0, (int)lo.l, o, 0));
        if (NR == 2) {
            const __amdgpu_buffer_rsrc_t rb1 = __builtin_amdgcn_make_buffer_rsrc((void*)bits[1], 0, -1, 0x00020000);
            dst[NTT].y = __int_as_float((int)__builtin_amdgcn_raw_buffer_load_b8(rb1, (int)lo.l, o, 0));
        }
    }
}

// ---- bookkeeping without memory round trips (round 5) ----
// Round 5's stamps: of a block's 33 k cycles in the forward kernel 2.7 k went into FINDING the wavefront's rows - per group a chain
// of dependent LDS reads through the group and range tables, then a scalar load of the first row's record with its wait (which
// drains the LDS counter as well) - and the plan is the same for every block of the chain.  Now:
//  * per wavefront three tables in VGPR LANES, filled once per kernel: lane e = the e-th group in execution order; a group costs
//    v_readlane instructions instead of LDS reads and waits;
//  * the row records of the direction are staged in LDS once and travel in ONE VGPR (lane i holds int i of the 16), two rows ahead
//    and ACROSS groups and blocks: the record that follows a group's last row is the next group's first - it is there when that
//    group starts, and it costs a vector register, not sixteen scalar ones (they spill); fields are moved to scalar registers
//    (v_readlane) where they are used.
struct RecV { int v; };        // lane i (mod 16) holds int i of the record: ONE register
__device__ __forceinline__ RecV wl_rec_fetch(const LDS_AS int32_t* lrecs, int idx, int lane) {
    RecV r;
    r.v = lrecs[idx * 16 + (lane & 15)];
    return r;
}
__device__ __forceinline__ RowU wl_rec_decode(const RecV& v) {
    i32x16 r;
#pragma unroll
    for (int i = 0; i < 16; ++i) r[i] = __builtin_amdgcn_readlane(v.v, i);
    return decode_rec(r);
}
__device__ __forceinline__ void wl_stage_recs(LDS_AS int32_t* lrecs, const void* recs_dir, int total_rows, int tid, int nthreads) {
    const GLOBAL_AS i32x4* src = (const GLOBAL_AS i32x4*)recs_dir;
    for (int i = tid; i < 4 * total_rows; i += nthreads) ((LDS_AS i32x4*)lrecs)[i] = src[i];
}
__device__ __forceinline__ int wl_lane_get(int table, int e) { return __builtin_amdgcn_readlane(table, e); }

// One row: ntt (1..3) adjacent 16-feature tiles of one unit, for NR row tiles.  Ring slot 0 holds k-block 0 on entry; on
// exit k-block 0 of row `nr` (fetched from pkn / bitsn: the next block's when the row is the wavefront's last of this
// block).  `part`: the wavefront's running partial of the unit's thin product behind the layer (forward: s | t of the
// node, backward: g_v), per row tile.
// ONE code path for every tile count: the ring always carries three tiles (a narrower row's last tile again: L1 hits) and
// only the MFMA groups and the per-tile epilogues sit under (wave-uniform) branches - a path per tile count would define the
// ring in three places, and the merge costs sixteen register copies behind a vmcnt(0) at the end of every row.
template <int KIND, int NR>
__device__ __forceinline__ void wl_row(const WlCtx& c, const RowU& cr, const RowU& nr, const GLOBAL_AS float* pkn,
                                       const GLOBAL_AS uint8_t* const (&bitsn)[2], f32x4 (&ring)[RING][NEL], f32x4 (&part)[NR],
                                       const LaneOff& lo, int lane) {
    static_assert(KIND < 0 || (RING == 2 && NTT == 3), "the WL rows alternate two ring slots of three tiles");     // (checked where instantiated: the general kernels include this header for its helpers with NTT = 4)
    const int m = lane & 15, kq = lane >> 4;
    const int n1 = cr.n1, ntt = cr.ntt;
    STAMP(c.sid + 0)
    // the unit's thin-layer inputs of this lane's batch row: forward the lanes feeding the subnet, backward g_s | g_t
    // ... as the B operand of the thin layer's MFMA: lane (m, kq) supplies input kq of batch row m
    float vk[NR];
#pragma unroll
    for (int h = 0; h < NR; ++h) {
        const int K = cr.thin_k & 0xff;
        const LDS_AS float* src = KIND == K_FWD ? c.xs[h] + m * c.xld + (cr.thin_k >> 16) : c.gst[h] + m * c.gld + (cr.thin_k >> 16);
        const float v = src[kq < K ? kq : 0];
        vk[h] = kq < K ? v : 0.f;
    }
    // forward: W1 vectors of k-block kb at tf[(5 kb + k) 16 + feature] (bias: k = 4); backward: W3^T vectors at tf[(4 kb + j) 16 + feature]
    const LDS_AS float* tf = c.par + cr.thin_w;
    f32x4 acc[NR][NTT];
#pragma unroll
    for (int h = 0; h < NR; ++h)
#pragma unroll
        for (int j = 0; j < NTT; ++j) acc[h][j] = zero4();

    // The B operand of k-block kb - a1 = relu(W1 v + b1) forward, g2 (before its mask) = W3^T g_st backward, four features per lane -
    // is ONE MFMA of its own since round 4 (K = the unit's <= 4 inputs: A = the thin layer's vectors of the k-block, one float per
    // lane; B = vk; C = the bias quad): 16 FMAs on the vector ALU cost 72 cycles of the SIMD beside the matrix pipe, the MFMA 32,
    // and the LDS reads shrink from five float4 to a float and a float4.  The MFMA adds bias, input 0, .., 3 in that order, as the
    // FMA chain did and as part B's rebuild and the backward's wl_layer1 do: the ReLU decisions still agree bit for bit.
    // Software-pipelined two steps deep: step kb issues the LDS reads of k-block kb + 2, turns the operands of kb + 1 (read a step
    // ago) into the B fragment of the next step, and multiplies with the fragment computed a step ago (two buffers: no copies).
    constexpr int KV = KIND == K_FWD ? 5 : 4;
    float qa = 0.f;
    f32x4 qb = zero4();
    auto q_load = [&](int kb) {
        qa = tf[(KV * kb + kq) * 16 + m];
        if (KIND == K_FWD) qb = *(const LDS_AS f32x4*)(tf + (KV * kb + 4) * 16 + 4 * kq);
    };
    auto q_frag = [&](int h) -> f32x4 {
        const f32x4 r = mfma4(qa, vk[h], qb);
        return KIND == K_FWD ? relu4m(r) : r;
    };
    f32x4 bq[2][NR];
    q_load(0);
#pragma unroll
    for (int h = 0; h < NR; ++h) bq[0][h] = q_frag(h);
    q_load(1);
    STAMP(c.sid + 1)
    // Program order of a step is pinned with empty asm statements that "use" the accumulators and clobber memory: the
    // prefetch of the next step's tiles goes out behind the MFMAs of the step before (its ring slot is dead then: no
    // register copies) and in front of this step's MFMAs (a full step of flight time).  Left to itself the compiler
    // either hoists the loads over the previous step (copies + vmcnt(0) at the back edge) or sinks them to their use.
#define WL_PIN()                                                                                        \
    {                                                                                                   \
        if constexpr (NR == 1) asm volatile("" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[0][2]) : : "memory"); \
        else asm volatile("" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[0][2]), "+v"(acc[NR - 1][0]), "+v"(acc[NR - 1][1]), "+v"(acc[NR - 1][2]) : : "memory"); \
    }
#define WL_MFMA(J)                                                                                      \
    _Pragma("unroll") for (int h_ = 0; h_ < NR; ++h_) acc[h_][J] = mfma4(ring[S_][J][i_], bq[S_][h_][i_], acc[h_][J]);
#define WL_STEP(KB, S, LIVE, PK, BITS, NR_, NKB, PREFETCH)                                              \
    {                                                                                                   \
        constexpr int S_ = (S);                                                                         \
        if (PREFETCH) wl_load<KIND, NR>(ring[(S_ + 1) & 1], PK, BITS, NR_, NKB, lo);                    \
        WL_PIN()                                                                                        \
        if (LIVE) {                                                                                     \
            _Pragma("unroll") for (int h = 0; h < NR; ++h) {                                            \
                if (KIND == K_BWD) wl_mask_by_bits(bq[S_][h], __float_as_int(h == 0 ? ring[S_][NTT].x : ring[S_][NTT].y)); \
                bq[S_ ^ 1][h] = q_frag(h);          /* k-block KB + 1 (behind the row's last one: never used) */ \
            }                                                                                           \
            q_load((KB) + 2);                                                                           \
            if (ntt >= 3) {                                                                             \
                _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) { WL_MFMA(0) WL_MFMA(1) WL_MFMA(2) }    \
            } else if (ntt == 2) {                                                                      \
                _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) { WL_MFMA(0) WL_MFMA(1) }               \
            } else {                                                                                    \
                _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) { WL_MFMA(0) }                          \
            }                                                                                           \
        }                                                                                               \
        WL_PIN()                                                                                        \
        STAMP(c.sid + 8 + ((KB) & 15))                                                                  \
    }
    // Steps in pairs (slots 0, 1).  The second step of a pair prefetches k-block k0 + 2 into slot 0 - behind the row's last
    // k-block that is k-block 0 of the next row (the hand-over).  An odd row ends with a single step on slot 0, which can
    // only be re-loaded behind its MFMAs: the hand-over then flies across the epilogue and whatever follows the row.
    // (an odd row pays one dummy step; a single last step that re-loads its own slot behind its MFMAs measured slower)
    const int n1p = (n1 + 1) & ~1;
    int k0 = 0;
    for (; k0 + 2 < n1p; k0 += 2) {
        WL_STEP(k0, 0, true, c.pk, c.bits, cr, k0 + 1, true)
        WL_STEP(k0 + 1, 1, true, c.pk, c.bits, cr, k0 + 2, true)
    }
    WL_STEP(k0, 0, true, c.pk, c.bits, cr, k0 + 1, true)              // (an odd row's step n1: a dummy load of the tile behind)
    WL_STEP(k0 + 1, 1, k0 + 1 < n1, pkn, bitsn, nr, 0, true)          // hands the ring to the next row
#undef WL_STEP
#undef WL_MFMA
#undef WL_PIN

    STAMP(c.sid + 2)
    // ---- the tiles are finished: activation, tape, the thin product behind the layer ----
    if (KIND == K_FWD) {
        const LDS_AS f32x4* w3 = (const LDS_AS f32x4*)(c.par + cr.thin_b) + kq;     // W3^T vector o of the row's tile t: w3[16 t + 4 o]
        const LDS_AS f32x4* b2 = (const LDS_AS f32x4*)(c.par + cr.aux) + kq;
#pragma unroll
        for (int j = 0; j < NTT; ++j) {
            if (j < ntt) {
                const f32x4 bias = b2[4 * j];
                f32x4 wv[4];
#pragma unroll
                for (int o = 0; o < 4; ++o) wv[o] = w3[16 * j + 4 * o];
#pragma unroll
                for (int h = 0; h < NR; ++h) {
                    // (the bias last, as the reference's addmm adds it: fewer rows land on the other side of a ReLU kink)
                    const f32x4 v = relu4(acc[h][j] + bias);
                    if (c.a2[h] != nullptr) {          // (training, and the row tile exists)
                        c.bits_out[h][((cr.ocol >> 4) + j) * 64 + lane] = (uint8_t)sign_bits(v);
                        *(GLOBAL_AS f32x4*)(c.a2[h] + (m * c.WT + cr.ocol + 16 * j + 4 * kq)) = v;
                    }
#pragma unroll
                    for (int o = 0; o < 4; ++o) part[h][o] = dot4(wv[o], v, part[h][o]);
                }
            }
        }
    } else {
        // g1 = acc .* relu'(a1), a1 recomputed from the level's lanes; g_v partial = W1^T g1; dW1 | db1 of the tile
        float xin[NR][4];
        const int cin = cr.p2 & 0xff, xoff = (cr.p2 >> 8) & 0xff, kcp = cin < 4 ? 4 : 8, hw = cr.p2 >> 16;
#pragma unroll
        for (int h = 0; h < NR; ++h)
#pragma unroll
            for (int k = 0; k < 4; ++k) { const float v = c.xs[h][m * c.xld + xoff + (k < cin ? k : 0)]; xin[h][k] = k < cin ? v : 0.f; }
        const LDS_AS f32x4* w1 = (const LDS_AS f32x4*)(c.par + cr.thin_b) + kq;     // W1 vector k of the row's tile t: w1[20 t + 4 k]
        const int nl = m;
#pragma unroll
        for (int j = 0; j < NTT; ++j) {
            if (j < ntt) {
                f32x4 dw = zero4();
#pragma unroll
                for (int h = 0; h < NR; ++h) {
                    const f32x4 pre = wl_layer1(w1 + 20 * j, xin[h]);
                    f32x4 g = acc[h][j];
                    g.x = pre.x > 0.f ? g.x : 0.f; g.y = pre.y > 0.f ? g.y : 0.f; g.z = pre.z > 0.f ? g.z : 0.f; g.w = pre.w > 0.f ? g.w : 0.f;
#pragma unroll
                    for (int o = 0; o < 4; ++o) part[h][o] = dot4(w1[20 * j + 4 * o], g, part[h][o]);
                    // dW1[f][k] = sum_rows g1[row][f] v[row][k], db1[f] = sum_rows g1[row][f]: the tile transposed through the
                    // private scratch tile, then four 16x16x4 MFMAs over the 16 rows (out^T[k][f]: a lane ends with four inputs
                    // of one feature); the second row tile adds into the same accumulator
                    ((LDS_AS f32x4*)c.scratch)[lane] = g;
                    const LDS_AS float* g1p = (const LDS_AS float*)c.scratch + (kq + 16 * (nl >> 2)) * 4 + (nl & 3);
                    const LDS_AS float* vp = c.xs[h] + kq * c.xld + xoff + (nl < cin ? nl : 0);
                    const float one = nl == cin ? 1.f : 0.f;
                    float av[4], bv[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) { av[i] = g1p[16 * i]; bv[i] = vp[4 * i * c.xld]; }      // rows 4 i + kq
#pragma unroll
                    for (int i = 0; i < 4; ++i) dw = mfma4(nl < cin ? bv[i] : one, av[i], dw);
                }
                const int nvalid = hw - 16 * (cr.tb + j);
                if (nl < nvalid && 4 * kq < kcp) {
                    GLOBAL_AS f32x4* dst = (GLOBAL_AS f32x4*)(c.tw + cr.p1 + (16 * j + nl) * kcp + 4 * kq);
                    if (c.first_tile) *dst = dw; else *dst = *dst + dw;
                }
            }
        }
    }
    STAMP(c.sid + 3)
    // (forward: b3 rides with the row that holds the unit's tile 0 - in one lane group, the fold below sums the four)
    if (KIND == K_FWD && cr.first && kq == 0) {
        const f32x4 b3 = *(const LDS_AS f32x4*)(c.par + cr.bias3);
#pragma unroll
        for (int h = 0; h < NR; ++h) part[h] += b3;
    }
    if (cr.ulast) {
        // the wavefront's last row of the unit: fold the four lane groups -> slab
#pragma unroll
        for (int h = 0; h < NR; ++h) {
            f32x4 s;
#pragma unroll
            for (int o = 0; o < 4; ++o) s[o] = kq_sum(part[h][o]);
            if (kq == 0) ((LDS_AS f32x4*)(c.slab + h * c.slab_h + cr.slab))[m] = s;
            part[h] = zero4();
        }
    }
    STAMP(c.sid + 4)
}

// The wavefront's rows [r0, r1) of a group.  rnext: the record the last row hands the ring to (the wavefront's first row of the next group, of the next block when
// other_block) or -1.
struct WlCarry {
    int primed;      // the record whose k-block 0 sits in ring slot 0 (index, or -1)
    int held;        // the record `nrec` holds (index, or -1)
    RecV nrec;       // ... fetched two rows ahead, carried across groups and blocks
};
template <int KIND, int NR>
__device__ __forceinline__ void wl_rows(const WlCtx& c, f32x4 (&ring)[RING][NEL], WlCarry& cy, int r0, int r1, int rnext,
                                        bool other_block, int lane) {
    if (r0 >= r1) return;
    LaneOff lo;
    lo.w = (unsigned)lane * 16u; lo.b = (unsigned)(lane >> 4) * 16u; lo.l = (unsigned)lane;
    if (cy.held != r0) cy.nrec = wl_rec_fetch(c.lrecs, r0, lane);      // (only a wavefront's first rows of a launch, or behind a group it had no rows in)
    RowU cr = wl_rec_decode(cy.nrec);
    if (cy.primed != r0) wl_load<KIND, NR>(ring[0], c.pk, c.bits, cr, 0, lo);
    const int rlast = rnext >= 0 ? rnext : r1 - 1;
    cy.nrec = wl_rec_fetch(c.lrecs, r0 + 1 < r1 ? r0 + 1 : rlast, lane);
    f32x4 part[NR];
#pragma unroll
    for (int h = 0; h < NR; ++h) part[h] = zero4();
    for (int t = r0; t < r1; ++t) {
        const RowU nr = wl_rec_decode(cy.nrec);
        cy.nrec = wl_rec_fetch(c.lrecs, t + 2 < r1 ? t + 2 : rlast, lane);
        const bool hand = t + 1 == r1 && other_block;
#ifdef HINT_STAMPS
        const_cast<WlCtx&>(c).sid = 256 + ((c.sid0 >> 3) & 3) * 64 + (t - r0 < 2 ? t - r0 : 1) * 32;
#endif
        const GLOBAL_AS float* pkn = hand ? c.pk_next : c.pk;
        const GLOBAL_AS uint8_t* const bitsn[2] = {hand ? c.bits_next[0] : c.bits[0], hand ? c.bits_next[1] : c.bits[1]};
        wl_row<KIND, NR>(c, cr, nr, pkn, bitsn, ring, part, lo, lane);
        cr = nr;
    }
    cy.primed = rnext;
    cy.held = rlast;
}

// The block's small parameters: 4 * par_f4 floats = [thin blobs: the start of the packed buffer | biases: at bias_src]
template <int PR>
__device__ __forceinline__ void wl_par_issue(f32x4 (&pf)[PR], const GLOBAL_AS float* packed, const WlArgs& w, int tid, int nthreads) {
#pragma unroll
    for (int q = 0; q < PR; ++q) {
        int i = tid + q * nthreads;
        i = i < w.par_f4 ? i : w.par_f4 - 1;
        const int f = 4 * i;
        pf[q] = *(const GLOBAL_AS f32x4*)(packed + (f < w.par_bias ? f : f - w.par_bias + w.bias_src));
    }
}
// The NEXT block's parameters straight into the other LDS buffer by LDS-DMA (global_load_lds_dwordx4: no VGPR destination, 1 KiB
// per wavefront instruction at M0 + 16 lane; the source address is per lane, so the [blobs | biases] gather is free): round 5's
// stamps had the register-staged copy - six 16-byte loads per thread held across the block's first group, then six
// ds_write_b128 by all eight wavefronts at once - at ~850 cycles per block on the LDS write path, behind the first group's
// coupling, and 24 registers.  hipcc does not count an asm load: its own vmcnt waits only become stricter (the DMA is older than
// every load the rows wait for); wl_par_dma_wait() drains it before the barrier that publishes the buffer.
__device__ __forceinline__ void wl_par_dma(const GLOBAL_AS float* packed, const WlArgs& w, const LDS_AS float* dst, int wave, int nw, int lane) {
    const int nchunk = (w.par_f4 + 63) >> 6;
    const unsigned base = (unsigned)(unsigned long long)dst;
    for (int c = wave; c < nchunk; c += nw) {
        int i = c * 64 + lane;
        i = i < w.par_f4 ? i : w.par_f4 - 1;       // (the buffer is padded to whole KiB: the tail lanes write a copy of the last float4 there)
        const int f = 4 * i;
        const GLOBAL_AS float* src = packed + (f < w.par_bias ? f : f - w.par_bias + w.bias_src);
        const unsigned ldsb = (unsigned)rfl((int)(base + (unsigned)c * 1024u));
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(ldsb) : "memory");
    }
}
__device__ __forceinline__ void wl_par_dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
template <int PR>
__device__ __forceinline__ void wl_par_commit(const f32x4 (&pf)[PR], float* dst, const WlArgs& w, int tid, int nthreads) {
#pragma unroll
    for (int q = 0; q < PR; ++q) {
        const int i = tid + q * nthreads;
        if (i < w.par_f4) ((f32x4*)dst)[i] = pf[q];
    }

}

}  // namespace hint
