// The GEMM engine of the block kernels.  One wavefront executes its list of ROWS of a group; a row
// is up to NTT (three or four) adjacent fragment tiles (16 output features x 16 batch rows each) of one unit
// that share every B fragment:
//
//   direction  thin layer (VALU, per unit)      main steps (A = weight tiles)   B operand      aux step            tail steps
//   forward    a1 = relu(W1 v + b1)             n1 = NT k-blocks of W2          a1 tiles (LDS)  + b2, ReLU -> a2    RT x W3 tiles (B = a2, registers) [+ b3]
//   backward   g2' = W3^T g_st (unmasked)       n1 = NT k-blocks of W2^T        g2' tiles (LDS) .* relu'(a1) -> g1  CT x W1^T tiles (B = g1)
//                                                                               .* relu'(a2 tile kb), which rides in the ring
//
// Every product is transposed (out^T = W * act^T), so the accumulator of the main steps - lane l
// holds features 4*(l>>4)+i of batch row l&15 - is exactly the B operand of the tail steps, and
// what a wavefront stores as a fragment tile (ds_write_b128 at lane*16) is what it (or another
// wavefront) reads as a B operand (ds_read_b128): no transposes.  The thin layers of a group (K = the
// few lanes feeding a subnet, or its r outputs) are a short phase of their own: the wavefronts share
// the group's tiles, a barrier, then the rows.  The K-split partial of a row's tail product goes to
// the LDS slab the wavefront keeps for the unit.
//
// The weight stream runs through a register ring of RING slots x 3 elements: every step consumes one
// slot and first issues the 16-byte-per-lane global loads of the step RING - 1 ahead into the slot the
// step before used (no copies of live values; a row's body is compiled for its tile count; the hand-over
// to the next row always loads NTT elements, a narrower row's last tile again: an L1 hit) - weight
// tiles and the sign bytes of the forward activations that mask the backward tiles travel the same way.  That regularity is what lets hipcc keep the
// loads in flight (counted vmcnt waits): a load under a branch makes its wait-count analysis assume
// the worst on every path.  A row's stream is [main steps, padded with dummies to a multiple of
// RING][extra steps: aux, tail, last-layer bias]; position p lives in slot p % RING, so the loops
// are plain unrolled-by-RING loops with static slots.  Per-row bookkeeping is one 64-byte record,
// fetched with a scalar load one row ahead.
#pragma once
#include "hint_device.hpp"

namespace hint {

enum { K_FWD = 0, K_BWD = 1 };
#ifndef HINT_RING
#define HINT_RING 2
#endif
constexpr int RING = HINT_RING;
static_assert(RING % 2 == 0, "the B-fragment double buffer alternates with the slot parity");
constexpr int DIST = RING - 1;      // how many steps ahead of its use a ring element is loaded
constexpr int NEL = NTT + 1;    // ring elements per slot: the row's weight tiles + (backward) the a2 tile that masks the step's B fragment

typedef int i32x16 __attribute__((ext_vector_type(16)));
#define CONST_AS __attribute__((address_space(4)))
__device__ __forceinline__ i32x16 load_rec(const void* recs, int idx) {
    const CONST_AS i32x16* p = (const CONST_AS i32x16*)(unsigned long long)recs;
    return p[idx];
}
struct RowU {           // decoded record (all wave-uniform)
    int base1, base2, n1, n2, n3, ntt, aux, ocol, tile0, nquad, slab, bias3;
    int thin_w, thin_b, thin_k, NT, thin, first, wcol, tb;
    int ulast, p1, p2;      // (wave-local kernels, hint_wl.hpp)
    int rowdw;              // (general backward kernel: dW1 | db1 of the row's tiles computed by the row)
};
__device__ __forceinline__ RowU decode_rec(const i32x16 r) {
    RowU u;
    u.base1 = r[0]; u.base2 = r[1];
    u.n1 = r[2] & 0xff; u.n2 = (r[2] >> 8) & 0xff; u.n3 = (r[2] >> 16) & 0xff; u.ntt = (r[2] >> 24) & 0xff;
    u.aux = r[3]; u.ocol = r[4];
    u.tile0 = r[5] & 0xffff; u.nquad = (r[5] >> 16) & 0xff;
    u.slab = r[6]; u.bias3 = r[7];
    u.thin_w = r[8]; u.thin_b = r[9]; u.thin_k = r[10];
    u.NT = r[11] & 0xff; u.thin = (r[11] >> 8) & 1; u.first = (r[11] >> 9) & 1;
    u.wcol = r[12]; u.tb = r[13];
    u.ulast = (r[11] >> 10) & 1; u.p1 = r[14]; u.p2 = r[15]; u.rowdw = (r[11] >> 11) & 1;
    return u;
}

// What the phases need besides the records: pointers and strides of the workgroup's LDS buffers and
// of the block's global arrays.  (Explicit address spaces: a generic pointer is read with flat_load,
// which counts against vmcnt AND lgkmcnt and turns every wait into vmcnt(0) lgkmcnt(0).)
struct PhaseCtx {
    const GLOBAL_AS float* packed;   // packed weights of the block
    const void* recs;                // row records of the plan (global, constant), this direction's
    LDS_AS float* abuf;              // fragment tiles of the group: a1 (forward), g2 (backward)
    LDS_AS float* slab;              // the rows' slabs
    LDS_AS float* obuf;              // fragment tiles of the group's outputs on their way to global memory: a2 (training forward), g1 (backward);
                                     // nullptr: no LDS for them, the rows store to out_thin / out_main themselves
    GLOBAL_AS float* out_thin;       // [Bp][WT]: a1 (training forward) / g2 (backward)
    GLOBAL_AS float* out_main;       // [Bp][WT]: a2 (training forward) / g1 (backward)
    const LDS_AS float* xs;          // lane tile [16][xld]
    const LDS_AS float* cs;          // condition tile [16][cld]
    const LDS_AS float* gst;         // coupling gradients [16][gld] (backward)
    const LDS_AS float* thin_l;      // the direction's thin blob staged in LDS, or nullptr
    const GLOBAL_AS float* thin_g;   // ... in the packed buffer
    // relu'() of the forward activations as one byte per lane and fragment tile (bit i: element i of the lane's
    // float4 is positive), [tile of the block][64], of this workgroup's rows: written by the training forward next
    // to the activation tiles, all the backward kernel reads of them (1/16 of the bytes)
    GLOBAL_AS uint8_t* bits_a1;
    GLOBAL_AS uint8_t* bits_a2;
    int xld, cld, gld, WT, row0;
    int wcol0;                       // column of the group's first tile in the [Bp][WT] arrays
    int sid;                         // diagnostic builds: stamp id base of the phase
    LDS_AS float* scratch;           // backward: one fragment tile of this wavefront (rows with the rowdw flag)
    GLOBAL_AS float* tw;             // backward: this workgroup's first-layer gradient slab
    bool first_tile;                 // backward: the workgroup's first row tile (plain stores into tw; later tiles add)
    bool store;                      // keep the outputs (training forward; always in the backward pass): the kernels stream them out of LDS after the phase
    bool fly;                        // forward, lean group with its thin vectors in LDS: no thin phase - the rows make their B fragments (a1 tiles) themselves
};

#ifdef HINT_NO_ROWDW          // (hint_bwd3.hip: the instance for plans without such rows)
#define HINT_ROWDW_ON false
#else
#define HINT_ROWDW_ON true
#endif
#ifdef HINT_ABLATE_STORE      // diagnostic: no activation / gradient rows leave the kernel
#define HINT_STORE_ON false
#else
#define HINT_STORE_ON c.store
#endif

// Byte offsets of a lane inside the three kinds of stream elements.
struct LaneOff { unsigned w, b, l; };    // lane*16 (weight tiles), kq*16 (bias vectors), lane (sign bytes)

__device__ __forceinline__ int sign_bits(const f32x4& v) {
    return (v.x > 0.f ? 1 : 0) | (v.y > 0.f ? 2 : 0) | (v.z > 0.f ? 4 : 0) | (v.w > 0.f ? 8 : 0);
}
__device__ __forceinline__ void mask_by_bits(f32x4& v, int bits) {
    v.x = (bits & 1) ? v.x : 0.f; v.y = (bits & 2) ? v.y : 0.f; v.z = (bits & 4) ? v.z : 0.f; v.w = (bits & 8) ? v.w : 0.f;
}

// the first N weight elements of main step kb of row r (tile j of a narrower row: its last tile again); backward:
// also the sign byte of the a2 tile of k-block kb (element NTT, in .x), which masks the step's B fragment
// (through buffer descriptors since round 4: the tile index is wave-uniform, so the scalar unit does the address arithmetic - the
//  offset operand of the load - and the per-lane part, lane * 16, is loop invariant; vector-ALU instructions do not hide under the
//  matrix pipe on gfx950, and a 64-bit per-lane address cost one or two per tile and step)
typedef unsigned rows_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 rows_as_f32x4(rows_u32x4 v) { return __builtin_bit_cast(f32x4, v); }
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rows_rsrc_of(const GLOBAL_AS void* p) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, -1, 0x00020000);
}
template <int KIND, int N>
__device__ __forceinline__ void load_main3(f32x4 (&dst)[NEL], const PhaseCtx& c, const RowU& r, int kb, const LaneOff& lo) {
    const __amdgpu_buffer_rsrc_t rw = rows_rsrc_of(c.packed);
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const int jj = j < r.ntt ? j : r.ntt - 1;
        dst[j] = rows_as_f32x4(__builtin_amdgcn_raw_buffer_load_b128(rw, (int)lo.w, (r.base1 + jj * r.n1 + kb) * 1024, 0));
    }
    if (KIND == K_BWD) {
        const int kc = kb < r.n1 ? kb : r.n1 - 1;
        dst[NTT].x = __int_as_float((int)__builtin_amdgcn_raw_buffer_load_b8(rows_rsrc_of(c.bits_a2), (int)lo.l, ((r.wcol >> 4) + kc) * 64, 0));
    }
}
// the first N elements of extra step e: 0 = aux (bias vectors / forward activation tiles), 1 .. n2 = tail
// weight tiles, then n3 bias vectors of the last layer; anything beyond re-loads the aux elements
template <int KIND, int N>
__device__ __forceinline__ void load_extra3(f32x4 (&dst)[NTT], const PhaseCtx& c, const RowU& r, int e, const LaneOff& lo) {
    int base, off, step;            // byte offsets inside the packed buffer: base + jj * step (scalar), off (per lane)
    if (e >= 1 && e <= r.n2) {
        base = (r.base2 + (e - 1) * r.n1) * 1024; off = (int)lo.w; step = 1024;
    } else if (e > r.n2 && e <= r.n2 + r.n3) {
        base = (r.bias3 + 16 * (e - 1 - r.n2)) * 4; off = (int)lo.b; step = 0;
    } else if (KIND == K_FWD) {
        base = r.aux * 4; off = (int)lo.b; step = 64;
    } else {            // backward aux: the sign bytes of the row's a1 tiles (in .x)
        const __amdgpu_buffer_rsrc_t rb = rows_rsrc_of(c.bits_a1);
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const int jj = j < r.ntt ? j : r.ntt - 1;
            dst[j].x = __int_as_float((int)__builtin_amdgcn_raw_buffer_load_b8(rb, (int)lo.l, ((r.ocol >> 4) + jj) * 64, 0));
        }
        return;
    }
    const __amdgpu_buffer_rsrc_t rw = rows_rsrc_of(c.packed);
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const int jj = j < r.ntt ? j : r.ntt - 1;
        dst[j] = rows_as_f32x4(__builtin_amdgcn_raw_buffer_load_b128(rw, off, base + jj * step, 0));
    }
}

// The thin layers of a group on the vector ALU: every wavefront computes its share [t0, t1) of the group's
// fragment tiles into LDS (a workgroup barrier follows before the rows read them as B operands):
//   forward   a1[f]  = relu(b1[f] + sum_k W1[f][k] v[k]),  v = [lanes xoff .. xoff+ku | condition]
//   backward  g2'[f] = sum_j W3[j][f] g_st[j]              (relu'(a2) is applied where the tiles are consumed)
// Lane l owns features 4*(l>>4)..+3 of row l&15 of a tile; the weights come as one float4 per input from the
// direction's thin blob (vector layout, inputs padded to four with zero vectors: no branches for the usual
// K <= 4), staged in LDS once per block when it is small enough (STAGED).
typedef int i32x4c __attribute__((ext_vector_type(4)));
// One group's thin vectors -> the LDS buffer (blobs too large to stage whole): the group's tiles are [tile_begin, tile_begin + ntiles)
// of this direction's records, their vectors one contiguous slice of the blob.  Returns the buffer's base for the records'
// blob-relative offsets (a workgroup barrier follows), or nullptr when the slice is larger than the buffer (buf_floats).
__device__ __forceinline__ const LDS_AS float* thin_group_stage(float* buf, int buf_floats, const GLOBAL_AS float* blob, const void* thins,
                                                                int tile_begin, int ntiles, int total_tiles, int blob_floats, int tid, int nthreads) {
    const CONST_AS i32x4c* recs = (const CONST_AS i32x4c*)(unsigned long long)thins;
    const int v0 = recs[tile_begin].x;
    const int v1 = tile_begin + ntiles < total_tiles ? recs[tile_begin + ntiles].x : blob_floats;
    if (v1 - v0 > buf_floats) return nullptr;
    const GLOBAL_AS f32x4* src = (const GLOBAL_AS f32x4*)(blob + v0);
    const int n4 = (v1 - v0) >> 2;
    for (int base = tid; base < n4; base += 8 * nthreads) {      // (eight loads in flight before the first LDS store: hint_sub.hpp block_stage)
        f32x4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) if (base + k * nthreads < n4) v[k] = src[base + k * nthreads];
#pragma unroll
        for (int k = 0; k < 8; ++k) if (base + k * nthreads < n4) ((f32x4*)buf)[base + k * nthreads] = v[k];
    }
    return (const LDS_AS float*)buf - v0;
}
#ifndef HINT_THIN_RUN
#define HINT_THIN_RUN 4
#endif
constexpr int THIN_RUN = HINT_THIN_RUN;       // tiles of a thin layer worked on together (4 or 6)
// NTI tiles with the same inputs (tiles of one unit, or of the s and the t subnet of a node) in lockstep: the inputs are read
// once and the tiles' dependent chains - record, vectors, FMAs / MFMAs, store - overlap instead of following each other
// (a tile is 1.5-3 k cycles of latency and a few hundred of work)
template <int KIND, bool STAGED, int NTI>
__device__ __forceinline__ void thin_tiles(const PhaseCtx& c, const i32x4c (&rec)[NTI], int lane) {
    const int m = lane & 15, kq = lane >> 4;
    const int K = rec[0].y & 0xff, kp = rec[0].w & 0xff;
    auto vec = [&](int n, int k) -> f32x4 {                  // vector k of this lane's features of tile n: float offset rec.x + 4 kq + 16 k
        if (STAGED) return *(const LDS_AS f32x4*)(c.thin_l + rec[n].x + 4 * kq + 16 * k);
        return *(const GLOBAL_AS f32x4*)(c.thin_g + rec[n].x + 4 * kq + 16 * k);
    };
    auto input = [&](int k) -> float {
        if (KIND == K_FWD) {
            const int ku = (rec[0].y >> 8) & 0xff, xoff = rec[0].y >> 16;
            return k < ku ? c.xs[m * c.xld + xoff + k] : c.cs[m * c.cld + (k - ku)];
        }
        return c.gst[m * c.gld + (rec[0].y >> 16) + k];
    };
    f32x4 acc[NTI];
    if ((rec[0].w >> 8) != 0) {
        // wide layer: on the matrix pipe, out^T = W * in^T from fragment tiles of W (k-block kb = 1 KiB, lane l: W[16nt + (l&15)]
        // [16kb + 4(l>>4) + i]); the B operand is gathered from the inputs (row l&15, input 16kb + 4(l>>4) + i; zero beyond K)
        const GLOBAL_AS f32x4* wp[NTI];
        f32x4 w[NTI];
        const int KB = (K + 15) >> 4;
#pragma unroll
        for (int n = 0; n < NTI; ++n) {
            wp[n] = (const GLOBAL_AS f32x4*)(c.packed + (size_t)((rec[n].w >> 8) - 1) * 256) + lane;
            acc[n] = KIND == K_FWD ? vec(n, kp) : zero4();
            w[n] = wp[n][0];
        }
        for (int kb = 0; kb < KB; ++kb) {
            f32x4 wn[NTI];
#pragma unroll
            for (int n = 0; n < NTI; ++n) wn[n] = wp[n][(kb + 1 < KB ? kb + 1 : kb) * 64];
            float b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { const int k = 16 * kb + 4 * kq + i; b[i] = input(k < K ? k : K - 1); b[i] = k < K ? b[i] : 0.f; }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int n = 0; n < NTI; ++n) acc[n] = mfma4(w[n][i], b[i], acc[n]);
#pragma unroll
            for (int n = 0; n < NTI; ++n) w[n] = wn[n];
        }
    } else {
        f32x4 w[NTI][4];
        float vin[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            vin[k] = k < K ? input(k) : 0.f;
#pragma unroll
            for (int n = 0; n < NTI; ++n) w[n][k] = vec(n, k);
        }
#pragma unroll
        for (int n = 0; n < NTI; ++n) {
            acc[n] = KIND == K_FWD ? vec(n, kp) : zero4();
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[n] += w[n][k] * vin[k];
        }
        // (inputs beyond four, at most eight: more go to the matrix pipe; their reads in flight together - one at a time is a
        //  dependent round trip per input; the padding terms add w * 0)
        constexpr int TL = 4;
        for (int k = 4; k < K; k += TL) {
            f32x4 wv[NTI][TL];
            float iv[TL];
#pragma unroll
            for (int u = 0; u < TL; ++u) {
                const int kk = k + u < K ? k + u : K - 1;
                iv[u] = input(kk);
#pragma unroll
                for (int n = 0; n < NTI; ++n) wv[n][u] = vec(n, kk);
            }
            __builtin_amdgcn_sched_barrier(0);      // (all reads first: hipcc otherwise waits for them pair by pair)
#pragma unroll
            for (int u = 0; u < TL; ++u)
#pragma unroll
                for (int n = 0; n < NTI; ++n) acc[n] += wv[n][u] * (k + u < K ? iv[u] : 0.f);
        }
    }
#pragma unroll
    for (int n = 0; n < NTI; ++n) {
        f32x4 v = acc[n];
        if (KIND == K_FWD) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        ((LDS_AS f32x4*)c.abuf)[rec[n].z * 64 + lane] = v;
        if (KIND == K_FWD && HINT_STORE_ON) {
            c.bits_a1[((c.wcol0 >> 4) + rec[n].z) * 64 + lane] = (uint8_t)sign_bits(v);
            if (c.obuf == nullptr && c.out_thin != nullptr)      // (no LDS staging: the tile goes to the tape from here)
                *(GLOBAL_AS f32x4*)(c.out_thin + ((size_t)c.row0 * c.WT + c.wcol0 + 16 * rec[n].z) + (m * c.WT + 4 * kq)) = v;
        }
    }
}
template <int KIND, bool STAGED>
__device__ __forceinline__ void thin_phase(const PhaseCtx& c, const void* thins, int t0, int t1, int lane) {
    const CONST_AS i32x4c* recs = (const CONST_AS i32x4c*)(unsigned long long)thins;
    int t = t0;
    while (t < t1) {
        // up to THIN_RUN consecutive tiles with the same inputs at a time
        // (no look-ahead beyond them: a scalar load in flight would be waited for by every lgkmcnt wait of their LDS reads)
        i32x4c r[THIN_RUN];
#pragma unroll
        for (int n = 0; n < THIN_RUN; ++n) r[n] = recs[t + n < t1 ? t + n : t];
        int run = 1;
#pragma unroll
        for (int n = 1; n < THIN_RUN; ++n)
            if (run == n && t + n < t1 && r[n].y == r[0].y && (r[n].w & 0xff) == (r[0].w & 0xff) && ((r[n].w >> 8) != 0) == ((r[0].w >> 8) != 0)) run = n + 1;
#define HINT_THIN_CASE(N)                                                          \
        if (run == (N)) {                                                          \
            i32x4c rr[N];                                                          \
            _Pragma("unroll") for (int n = 0; n < (N); ++n) rr[n] = r[n];          \
            thin_tiles<KIND, STAGED, (N)>(c, rr, lane);                            \
        }
        HINT_THIN_CASE(1) else HINT_THIN_CASE(2) else HINT_THIN_CASE(3) else HINT_THIN_CASE(4)
#if HINT_THIN_RUN >= 6
        else HINT_THIN_CASE(5) else HINT_THIN_CASE(6)
#endif
#undef HINT_THIN_CASE
        t += run;
    }
}

// One row with NA tiles (compile-time: a run-time tile count would put a branch around every MFMA).  On entry the
// ring holds the row's first RING main steps; on exit the first RING main steps of row `nr`.  The row's first
// extra elements - bias vectors / forward activation tiles, the tail product's first weight tiles, the first
// last-layer bias vector - are fetched when the row starts and wait in registers of their own; further tail
// steps (r or cin beyond 16) fetch theirs when they run.
// FLY (PhaseCtx::fly; backward: the hint_bwd_fly.hip instance, B fragment = g2' = W3^T g_st): the row's B fragments - the a1 tiles of its unit, 1..4 inputs - are not read from LDS but made on
// the spot, one K <= 4 MFMA each from the unit's thin vectors staged in LDS (the wave-local kernels' first layer, hint_wl.hpp
// wl_row: A = W1[feature l&15][input l>>4] = the tile's four 16-float vectors read at + lane, B = the row's inputs, C = b1);
// the unit's first row leaves their sign bytes.  The d = 100 trees' lean groups spent 6-7 k of their 25 k cycles in the thin
// phase and its barrier for 64 tiles of 16 FMAs per lane.
template <int KIND, int NA, bool FLY = false>
__device__ __forceinline__ void row_body(const PhaseCtx& c, const RowU& cr, const RowU& nr, f32x4 (&ring)[RING][NEL],
                                         const LaneOff& lo, int lane) {
    const int m = lane & 15, kq = lane >> 4;
    const LDS_AS f32x4* abuf4 = (const LDS_AS f32x4*)c.abuf + lane;
    const int n1 = cr.n1;
    const int n1p = (n1 + RING - 1) / RING * RING;
    f32x4 xaux[NTT], xtail[NTT], xb3[NTT];
    f32x4 acc[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) acc[j] = zero4();
    // (FLY: tile kb's vectors at thin_w + 80 kb - four weight vectors and the bias vector of 16 floats)
    const LDS_AS float* tf = c.thin_l + cr.thin_w;
    float vin = 0.f;
    if (FLY && KIND == K_FWD) { const int cin = cr.thin_k & 0xff, xoff = cr.thin_k >> 16; const float v = c.xs[m * c.xld + xoff + (kq < cin ? kq : 0)]; vin = kq < cin ? v : 0.f; }
    if (FLY && KIND == K_BWD) { const int r = cr.thin_k & 0xff, lcol = cr.thin_k >> 16; const float v = c.gst[m * c.gld + lcol + (kq < r ? kq : 0)]; vin = kq < r ? v : 0.f; }
    auto bfrag = [&](int kb) -> f32x4 {
        if (!FLY) return abuf4[(cr.tile0 + kb) * 64];
        // (backward: g2' = W3^T g_st of the tile, unmasked - four 16-float vectors per tile, no bias vector)
        if (KIND == K_BWD) return mfma4(tf[64 * kb + lane], vin, zero4());
        const float qa = tf[80 * kb + lane];
        const f32x4 qb = *(const LDS_AS f32x4*)(tf + 80 * kb + 64 + 4 * kq);
        f32x4 v = mfma4(qa, vin, qb);
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        // (the sign bytes go to the group's - otherwise unused - a1 region in LDS; the element-wise phase streams them out: a global
        //  store among the ring's loads would hold every later load back until it is acknowledged)
        if (HINT_STORE_ON && cr.first) ((LDS_AS uint8_t*)c.abuf)[(cr.tile0 + kb) * 64 + lane] = (uint8_t)sign_bits(v);
        return v;
    };
    f32x4 bb[2];
    bb[0] = bfrag(0);
    bb[1] = zero4();

    // One main step of static slot S: k-block KB with the slot's weight fragments (backward: element NTT = the sign
    // byte of the a2 tile of the k-block).  First the loads of the step DIST = RING - 1 ahead - of row NR (this row,
    // or the next one at the end of the last chunk) - into slot (S + DIST) % RING, whose last use was the step
    // before: no copy of a live value, nothing at the loop's back edge that waits for a load just issued.
    const bool stg2 = KIND == K_BWD && HINT_STORE_ON && cr.first && c.out_thin != nullptr;      // this row leaves the unit's masked g2 tiles in LDS (streamed out later)
#define HINT_MAIN_STEP(KB, S, LIVE, NR, NKB, NLOAD)                                                     \
    {                                                                                                   \
        load_main3<KIND, NLOAD>(ring[((S) + DIST) % RING], c, NR, NKB, lo);                             \
        f32x4 b4 = bb[(S) & 1];                                                                         \
        const int kn = (KB) + 1 < n1 ? (KB) + 1 : (KB);                                                 \
        bb[((S) + 1) & 1] = bfrag(kn < n1 ? kn : n1 - 1);      /* (dummy steps: the last tile again) */ \
        if (LIVE) {                                                                                     \
            if (KIND == K_BWD) {                                                                        \
                mask_by_bits(b4, __float_as_int(ring[S][NTT].x));                                       \
                if (stg2) {                                                                             \
                    if (c.obuf != nullptr) ((LDS_AS f32x4*)c.abuf)[(cr.tile0 + (KB)) * 64 + lane] = b4;   /* (idempotent for the other readers) */ \
                    else *(GLOBAL_AS f32x4*)(c.out_thin + ((size_t)c.row0 * c.WT + cr.wcol + 16 * (KB)) + (m * c.WT + 4 * kq)) = b4; \
                }                                                                                       \
            }                                                                                           \
            _Pragma("unroll") for (int i = 0; i < 4; ++i)                                               \
                _Pragma("unroll") for (int j = 0; j < NA; ++j) acc[j] = mfma4(ring[S][j][i], b4[i], acc[j]); \
        }                                                                                               \
    }

    // ---- main steps: all chunks but the last (no dummies in them) ----
    int k0 = 0;
    for (; k0 + RING < n1p; k0 += RING) {
#pragma unroll
        for (int s = 0; s < RING; ++s) {
            HINT_MAIN_STEP(k0 + s, s, true, cr, k0 + s + DIST, NA)
            STAMP(384 + ((c.sid >> 4) & 7) * 16 + ((k0 + s) & 15))
        }
    }
    STAMP(c.sid + 10)
    // the row's first extra elements: issued here, behind the loop, so that what waits for them further down can
    // count the loads in between (a value loaded in front of a loop of unknown trip count is waited for with
    // vmcnt(0) - i.e. for the hand-over loads below as well)
    load_extra3<KIND, NA>(xaux, c, cr, 0, lo);
    load_extra3<KIND, NA>(xtail, c, cr, 1, lo);
    load_extra3<KIND, 1>(xb3, c, cr, 1 + cr.n2, lo);
    // ---- last main chunk: its last DIST steps fetch the next row's first DIST main steps (all NTT elements: the
    //      next row may be wider) ----
#pragma unroll
    for (int s = 0; s < RING; ++s) {
        const bool live = k0 + s < n1;
        if (s + DIST < RING) { HINT_MAIN_STEP(k0 + s, s, live, cr, k0 + s + DIST, NA) }
        else { HINT_MAIN_STEP(k0 + s, s, live, nr, s + DIST - RING, NTT) }
    }
#undef HINT_MAIN_STEP

    STAMP(c.sid + 11)
    // ---- aux step: the main tiles are finished ----
    f32x4 act[NA];           // (B operands of the tail steps)
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        f32x4 v = acc[j];
        if (KIND == K_FWD) {
            v += xaux[j];                                                            // bias
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        } else {                                                                     // relu'() of the forward activation
            mask_by_bits(v, __float_as_int(xaux[j].x));
        }
        act[j] = v;
        if (KIND == K_FWD && HINT_STORE_ON) c.bits_a2[((cr.ocol >> 4) + j) * 64 + lane] = (uint8_t)sign_bits(v);
        if (HINT_STORE_ON && !(KIND == K_BWD && HINT_ROWDW_ON && cr.rowdw)) {       // (a row that computes dW1 | db1 itself keeps its g1 on chip)
            if (c.obuf != nullptr) ((LDS_AS f32x4*)c.obuf)[(cr.tile0 + cr.tb + j) * 64 + lane] = v;
            else    // (groups too large to stage: straight to the tape, non-temporal - it is read again a kernel later, by part B; -2 % at d = 100)
                __builtin_nontemporal_store(v, (GLOBAL_AS f32x4*)(c.out_main + ((size_t)c.row0 * c.WT + cr.ocol + 16 * j) + (m * c.WT + 4 * kq)));
        }
    }
    // ---- tail steps: slab[q] = sum over the row's tiles of Wtail(q, tile) * act, the K-split partial of the thin
    //      product (compact: quad q4 = 4 q + kq holds features 4 q4 .. +3 of rows m); then, with a unit's first row,
    //      the last layer's bias vectors are added to the same slab ----
    LDS_AS float* slabp = c.slab + cr.slab;
    for (int q = 0; q < cr.n2; ++q) {
        if (q > 0) load_extra3<KIND, NA>(xtail, c, cr, 1 + q, lo);
        f32x4 sv = zero4();
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NA; ++j) sv = mfma4(xtail[j][i], act[j][i], sv);
        const int q4 = 4 * q + kq;
        if (q4 < cr.nquad) {
            LDS_AS f32x4* sp = (LDS_AS f32x4*)(slabp + (q4 * 16 + m) * 4);
            if (cr.thin) *sp = sv; else *sp = *sp + sv;             // (the wavefront's first row of the unit starts the slab)
        }
    }
    for (int q = 0; q < cr.n3; ++q) {
        if (q > 0) load_extra3<KIND, 1>(xb3, c, cr, 1 + cr.n2 + q, lo);
        const int q4 = 4 * q + kq;
        if (q4 < cr.nquad) {
            LDS_AS f32x4* sp = (LDS_AS f32x4*)(slabp + (q4 * 16 + m) * 4);
            *sp = *sp + xb3[0];
        }
    }
    if (KIND == K_BWD && HINT_ROWDW_ON && cr.rowdw) {
        // dW1[f][k] = sum_rows g1[row][f] v[row][k], db1[f] = sum_rows g1[row][f] of the row's tiles (lean group whose outputs are not
        // staged in LDS: g1 would have to travel to part B otherwise): each tile transposed through the wavefront's scratch tile,
        // four 16x16x4 MFMAs over the 16 rows (out^T[k][f]: a lane ends with four inputs of one feature), into the workgroup's slab
        const int cin = cr.p2 & 0xff, xoff = (cr.p2 >> 8) & 0xff, hw = cr.p2 >> 16, kcp = cin < 4 ? 4 : 8;
        const LDS_AS float* g1p = c.scratch + (kq + 16 * (m >> 2)) * 4 + (m & 3);
        const LDS_AS float* vp = c.xs + kq * c.xld + xoff + (m < cin ? m : 0);
        const float one = m == cin ? 1.f : 0.f;
        float bv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) bv[i] = vp[4 * i * c.xld];                  // rows 4 i + kq
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            ((LDS_AS f32x4*)c.scratch)[lane] = act[j];
            asm volatile("" ::: "memory");              // (the wavefront's own LDS traffic is in order)
            float av[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) av[i] = g1p[16 * i];
            f32x4 dw = zero4();
#pragma unroll
            for (int i = 0; i < 4; ++i) dw = mfma4(m < cin ? bv[i] : one, av[i], dw);
            const int nvalid = hw - 16 * (cr.tb + j);
            if (m < nvalid && 4 * kq < kcp) {
                GLOBAL_AS f32x4* dst = (GLOBAL_AS f32x4*)(c.tw + cr.p1 + (16 * j + m) * kcp + 4 * kq);
                if (c.first_tile) *dst = dw; else *dst = *dst + dw;
            }
            asm volatile("" ::: "memory");
        }
    }
    STAMP(c.sid + 12)
}

// Row records as ONE vector register (round 5, what the wave-local kernels do - hint_wl.hpp): lane i (mod 16) holds int i of
// the 64-byte record, fetched with one dword load per lane two rows ahead and carried ACROSS groups (the record behind a group's
// last row is the next group's first); fields go to scalar registers by v_readlane where they are used.  A scalar load per record
// was sixteen scalar registers in flight per row and, at every group's start, a wait that drains the LDS counter as well.
struct RecCarry { int held; int v; };        // the record index `v` holds (or -1)
__device__ __forceinline__ int recv_fetch(const void* recs, int idx, int lane) {
    return ((const GLOBAL_AS int*)(unsigned long long)recs)[(size_t)idx * 16 + (lane & 15)];
}
__device__ __forceinline__ RowU recv_decode(int v) {
    i32x16 r;
#pragma unroll
    for (int i = 0; i < 16; ++i) r[i] = __builtin_amdgcn_readlane(v, i);
    return decode_rec(r);
}

// One GEMM phase of one wavefront: records [r0, r1) of this direction's record list, in two calls so that the
// first row's weight stream can be started a phase early (its loads do not depend on the element-wise phase
// in between): rows_begin() primes the ring, rows_run() executes the rows.
template <int KIND>
__device__ __forceinline__ void rows_begin(const PhaseCtx& c, f32x4 (&ring)[RING][NEL], RecCarry& rc, int r0, int r1, int lane) {
    if (r0 >= r1) return;
    LaneOff lo;
    lo.w = (unsigned)lane * 16u; lo.b = (unsigned)(lane >> 4) * 16u; lo.l = (unsigned)lane;
    RowU cr;
    if constexpr (KIND == K_BWD) {
        if (rc.held != r0) { rc.v = recv_fetch(c.recs, r0, lane); rc.held = r0; }
        cr = recv_decode(rc.v);
    } else {
        cr = decode_rec(load_rec(c.recs, r0));
    }
#pragma unroll
    for (int s = 0; s < DIST; ++s) load_main3<KIND, NTT>(ring[s], c, cr, s, lo);     // (a padded row's first RING positions are main steps)
}
// rnext: the record whose first main steps the LAST row hands the ring over to - the wavefront's first row of the
// next group of the block - or -1 (then it re-loads its own: never used)
template <int KIND, bool FLYK = false>
__device__ __forceinline__ void rows_run(const PhaseCtx& c, f32x4 (&ring)[RING][NEL], RecCarry& rc, int r0, int r1, int rnext, int lane) {
    if (r0 >= r1) {
        if (rnext >= 0) rows_begin<KIND>(c, ring, rc, rnext, rnext + 1, lane);     // nothing to do here, but the next group has work
        return;
    }
    const int kq = lane >> 4;
    LaneOff lo;
    lo.w = (unsigned)lane * 16u; lo.b = (unsigned)kq * 16u; lo.l = (unsigned)lane;

    STAMP(c.sid + 7)
    // (vector records in the backward instances only: MINIBOONE's backward - 3.6 %, d = 100's - 0.8 %; the forward kernels lost 1-1.5 %
    //  with them and keep the scalar loads)
    constexpr bool VREC = KIND == K_BWD;
    RowU cr;
    if constexpr (VREC) { if (rc.held != r0) rc.v = recv_fetch(c.recs, r0, lane); cr = recv_decode(rc.v); }
    else cr = decode_rec(load_rec(c.recs, r0));
    const int rlast = rnext >= 0 ? rnext : r1 - 1;                            // what follows the last row
    int nrecv = 0;
    i32x16 nrecs = {};
    if constexpr (VREC) nrecv = recv_fetch(c.recs, r0 + 1 < r1 ? r0 + 1 : rlast, lane);       // next row's record, one row ahead
    else nrecs = load_rec(c.recs, r0 + 1 < r1 ? r0 + 1 : rlast);
    STAMP(c.sid + 8 + (cr.n1 > 1000 ? 1 : 0))
    for (int t = r0; t < r1; ++t) {
        RowU nr;
        if constexpr (VREC) { nr = recv_decode(nrecv); nrecv = recv_fetch(c.recs, t + 2 < r1 ? t + 2 : rlast, lane); }
        else { nr = decode_rec(nrecs); nrecs = load_rec(c.recs, t + 2 < r1 ? t + 2 : rlast); }
        if (t == r0) { STAMP(c.sid + 9) }
        if constexpr (NTT >= 4 && FLYK) {
            if (c.fly) {
                if (cr.ntt >= 4) row_body<KIND, 4, true>(c, cr, nr, ring, lo, lane);
                else if (cr.ntt == 3) row_body<KIND, 3, true>(c, cr, nr, ring, lo, lane);
                else if (cr.ntt == 2) row_body<KIND, 2, true>(c, cr, nr, ring, lo, lane);
                else row_body<KIND, 1, true>(c, cr, nr, ring, lo, lane);
            } else {
                if (cr.ntt >= 4) row_body<KIND, 4>(c, cr, nr, ring, lo, lane);
                else if (cr.ntt == 3) row_body<KIND, 3>(c, cr, nr, ring, lo, lane);
                else if (cr.ntt == 2) row_body<KIND, 2>(c, cr, nr, ring, lo, lane);
                else row_body<KIND, 1>(c, cr, nr, ring, lo, lane);
            }
        } else if constexpr (NTT >= 4) {
            if (cr.ntt >= 4) row_body<KIND, 4>(c, cr, nr, ring, lo, lane);
            else if (cr.ntt == 3) row_body<KIND, 3>(c, cr, nr, ring, lo, lane);
            else if (cr.ntt == 2) row_body<KIND, 2>(c, cr, nr, ring, lo, lane);
            else row_body<KIND, 1>(c, cr, nr, ring, lo, lane);
        } else if constexpr (NTT == 3) {
            if (cr.ntt >= 3) row_body<KIND, 3>(c, cr, nr, ring, lo, lane);
            else if (cr.ntt == 2) row_body<KIND, 2>(c, cr, nr, ring, lo, lane);
            else row_body<KIND, 1>(c, cr, nr, ring, lo, lane);
        } else if constexpr (NTT == 2) {
            if (cr.ntt >= 2) row_body<KIND, 2>(c, cr, nr, ring, lo, lane);
            else row_body<KIND, 1>(c, cr, nr, ring, lo, lane);
        } else {
            row_body<KIND, 1>(c, cr, nr, ring, lo, lane);
        }
        cr = nr;
    }
    if constexpr (VREC) { rc.v = nrecv; rc.held = rlast; }          // (the record behind the last row: the next group's first, when there is one)
    STAMP(c.sid + 13)
    STAMP(c.sid + 14)
}

}  // namespace hint
