// Subtree groups of the general block kernels (hint_fwd.hip / hint_bwd.hip): the deepest levels of a wide tree, where
// every subnet has 1..4 inputs, at most 4 outputs and one 16-feature tile (MINIBOONE d = 43: the 24 nodes of depth 3 and
// 4, hidden width 8), cost a general group's three phases and barriers per level although a level is a few hundred FMAs -
// and below some depth the subtrees never exchange anything (hint.py:70-73,85-88: the children of a node are
// independent).  So from that depth down (Group::lean bit 2, hint_plan.cpp) every wavefront takes whole subtrees and runs
// their nodes back to back with WAVE-local synchronisation only, register resident: per node both subnets in lockstep
// (first layer by FMAs from the staged thin vectors, h x h on the matrix pipe with the W2 tile fetched while the node before
// ran, the thin product behind it by FMAs + lane-group fold: hint_wl.hpp's expressions), then the coupling of the node's
// lanes on the wavefront's own columns of the workgroup's lane tile - nothing between lane tile and lane tile touches LDS
// slabs or row records.  One workgroup barrier where the subtrees rejoin the general groups.  Tape, workspace and packed
// weights are unchanged (part B does not know).  MINIBOONE x 10 blocks, 4096 rows: forward 386 -> 338 us, backward
// 500 -> 445 us (the two levels cost the general kernels 39 k cycles per block and kernel, here 15 k / 20 k).
#pragma once
#include "hint_wl.hpp"

namespace hint {

// LDS traffic of one wavefront is processed in order: what it wrote it can read back; the compiler must not reorder
__device__ __forceinline__ void wave_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// What a block's phases read out of LDS instead of L2, global -> LDS by the whole workgroup at the top of the block: the subtree
// units' thin vectors and biases [forward vectors | backward vectors | biases] (what the rows' records index; n_sub > 0) and the
// block's thin-layer vectors (thin_n4 float4 from thin_src; 0: none).  One index space, up to STAGE_INFLIGHT loads per thread
// issued before the first LDS store: a load -> store loop pays the L2 latency once per iteration (MINIBOONE: 5 + 2 iterations,
// 11 k cycles per block and kernel; batched 3 k).
constexpr int STAGE_INFLIGHT = 8;
// (sub / thin_n4 = 0: without that part - the two are wanted at different times: forward, the subtree phase comes first and the thin
//  vectors are staged behind it, in front of its barrier, where the early wavefronts wait anyway; backward the other way round)
__device__ __forceinline__ void block_stage(const KArgs& a, const GLOBAL_AS float* packed, float* lds, bool sub, int thin_n4, int tid, int nthreads) {
    const int n_sub4 = (sub && a.n_sub > 0) ? a.sub_par_f4 : 0;
    const int total = n_sub4 + thin_n4;
    // (no branches: the threads past the end repeat the last element - the same value to the same address)
    for (int base = 0; base < total; base += STAGE_INFLIGHT * nthreads) {
        f32x4 v[STAGE_INFLIGHT];
        int dst[STAGE_INFLIGHT];
#pragma unroll
        for (int k = 0; k < STAGE_INFLIGHT; ++k) {
            int i = base + k * nthreads + tid;
            i = i < total ? i : total - 1;
            const int f = 4 * i, ft = 4 * (i - n_sub4);
            const int src = i >= n_sub4 ? a.thin_off + ft
                          : f < a.sub_pf ? f : f < a.sub_pf + a.sub_pb ? f - a.sub_pf + a.sub_bsrc : f - a.sub_pf - a.sub_pb + a.sub_bias_src;
            dst[k] = i >= n_sub4 ? a.thin_lds + ft : a.sub_par + f;
            v[k] = *(const GLOBAL_AS f32x4*)(packed + src);
        }
#pragma unroll
        for (int k = 0; k < STAGE_INFLIGHT; ++k) *(f32x4*)(lds + dst[k]) = v[k];
    }
}

// a global load the compiler's wait-count bookkeeping does not see, and the wait that belongs to it
__device__ __forceinline__ void sub_load(f32x4& dst, const GLOBAL_AS f32x4* p) {
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}
template <int N>
__device__ __forceinline__ void sub_wait() { asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory"); }

// Both subnets (s, t) of a subtree node, forward, in lockstep - every stage of the two independent chains is issued back to
// back, so that their LDS and MFMA latencies overlap: one 16-feature tile, one k-block each (h <= 16).  w: the W2 fragment
// tiles (in registers since the node before); v: the lane's four hidden activations a2 (features 4 kq .. +3 of batch row m);
// out: the third layer's outputs of row m (every lane of the row holds all four).  The expressions are hint_wl.hpp's
// (wl_layer1, bias last, dot4 + fold): the backward pass recomputes relu'(a1) from them.
__device__ __forceinline__ void sub_node_fwd(const LDS_AS f32x4* par4, const KArgs& a, const UnitU (&u)[2], const f32x4 (&w)[2],
                                             const float (&vin)[4], int kq, f32x4 (&v)[2], f32x4 (&out)[2]) {
    const int pb = (a.sub_pf + a.sub_pb - a.sub_bias_src) >> 2;          // (float4 index of packed-buffer bias offset 0 inside the staged parameters)
    // everything the node reads from the staged parameters, up front: none of it depends on what is computed
    f32x4 q1[2][5], b2[2], w3[2][4], b3[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const LDS_AS f32x4* q = par4 + (u[n].w1v >> 2) + kq;
#pragma unroll
        for (int k = 0; k < 5; ++k) q1[n][k] = q[4 * k];
        b2[n] = par4[pb + (u[n].bias2 >> 2) + kq];
        const LDS_AS f32x4* w3p = par4 + ((a.sub_pf + u[n].w3v) >> 2) + kq;
#pragma unroll
        for (int o = 0; o < 4; ++o) w3[n][o] = w3p[4 * o];
        b3[n] = par4[pb + (u[n].bias3 >> 2)];
    }
    f32x4 a1[2], acc[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) {      // (wl_layer1's expression: bias first, then the inputs in order)
        a1[n] = relu4(fma4(q1[n][3], vin[3], fma4(q1[n][2], vin[2], fma4(q1[n][1], vin[1], fma4(q1[n][0], vin[0], q1[n][4])))));
        acc[n] = zero4();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[n] = mfma4(w[n][i], a1[n][i], acc[n]);
    f32x4 p[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        v[n] = relu4(acc[n] + b2[n]);
#pragma unroll
        for (int o = 0; o < 4; ++o) p[n][o] = dot4(w3[n][o], v[n], 0.f);
        if (kq == 0) p[n] += b3[n];
    }
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int n = 0; n < 2; ++n) out[n][o] = kq_sum(p[n][o]);
}

// Forward (children first) / inverse (parents first) through the subtree groups; xs: the workgroup's lane tile.  Per node:
// both subnets (their W2 tiles were fetched while the node before ran), then the coupling (hint.py:79-83) of the node's r
// lanes by the lane groups kq < r - everything between the lane tile and the lane tile stays in registers.  Leaves the
// wavefront's share of the log-det of batch row l & 15 in LDS (a.sub_misc + 16 wave + row); the caller adds them up behind
// its barrier.
// TRAIN (the tape is written) is a template parameter, the prefetch and its wait are unconditional, and every store the wait
// count relies on is issued whatever the lanes' predicates: the hand-counted waits are then provable on the generated assembly
// path by path (tools/check_untracked_loads.py, `make asm`) - a count that holds only because two `if (train)` agree, or
// because a predicated store happens to have an active lane, is not.
template <bool REV, bool TRAIN>
__device__ __forceinline__ void sub_apply_t(const KArgs& a, const Tables& T, float* lds, const GBlock& blk, float* xs,
                                            int row0, int wave, int lane, int sid) {
    constexpr bool train = TRAIN;
    (void)sid;
    const int m = lane & 15, kq = lane >> 4;
    const LDS_AS f32x4* par4 = (const LDS_AS f32x4*)(lds + a.sub_par);
    float* tape = (float*)blk.tape;
    const size_t lvl = (size_t)a.B * a.d;
    GLOBAL_AS float* a2 = train ? blk.actA1 + a.a2_off + (size_t)row0 * a.WT : nullptr;
    GLOBAL_AS uint8_t* bits = train ? (GLOBAL_AS uint8_t*)(blk.actA1 + a.bits_off) + a.bits_stride + (size_t)(row0 >> 4) * (a.WT >> 4) * 64 : nullptr;
    const GLOBAL_AS f32x4* wt = (const GLOBAL_AS f32x4*)blk.packed + lane;        // fragment tile t: wt[64 t]
    const int c0 = lds_i32(T.rng + a.sub_cols + 4 * wave + 2), c1 = lds_i32(T.rng + a.sub_cols + 4 * wave + 3);     // the lanes it stores to the tape
    const int wc = c1 - c0;
    const float inv_wc = frcp(wc > 0 ? wc : 1);
    float jpart = 0.f;
    // the wavefront's units of subtree group gidx: [ub, ue) (one row per unit: the row ranges are unit ranges)
    auto range = [&](int gidx, int& ub, int& ue, int& level) {
        const LDS_AS int32_t* gp = (const LDS_AS int32_t*)(T.groups + gidx);
        const int unit_begin = lds_i32(gp + 0), rngb = lds_i32(gp + 6);
        level = lds_i32(gp + 7);
        ub = unit_begin + lds_i32(T.rng + rngb + wave); ue = unit_begin + lds_i32(T.rng + rngb + wave + 1);
    };
    auto tile_of = [&](int u) -> int { return lds_i32((const LDS_AS int32_t*)(T.units + u) + 1); };     // Unit::f2
    int ub, ue, level;
    range(REV ? a.n_sub - 1 : 0, ub, ue, level);
    // The W2 tiles of a node are fetched while the node before runs - with loads the compiler does not track (sub_load): a
    // tracked load that is carried around the loop is waited for with vmcnt(0), i.e. for the tape stores of the node in
    // between as well (2-3 k cycles per node); the waits are placed by hand (sub_wait: at least four stores follow a
    // training node's loads, none an inference node's).
    f32x4 ws = zero4(), wtt = zero4();
    int have = -1;                      // the node whose tiles are in ws / wtt
    for (int q = 0; q < a.n_sub; ++q) {
        int nb = 0, ne = 0, nlevel = 0;
        if (q + 1 < a.n_sub) range(REV ? a.n_sub - 2 - q : q + 1, nb, ne, nlevel);
        for (int u = ub; u < ue; u += 2) {
            if (have != u) {            // (the wavefront's first node, or the first one behind a level without nodes of its own)
                sub_load(ws, wt + (size_t)tile_of(u) * 64); sub_load(wtt, wt + (size_t)tile_of(u + 1) * 64);
                sub_wait<0>();
            }
            // the next node (of the next level behind this level's last node; behind the last one of all: this node's tiles again -
            // the loads are never under a branch)
            const bool has_next = u + 2 < ue || nb < ne;
            const int nu = u + 2 < ue ? u + 2 : (nb < ne ? nb : u);
            f32x4 nws, nwt;             // (written by the untracked loads only: no value of the compiler's may share their registers before sub_wait)
            sub_load(nws, wt + (size_t)tile_of(nu) * 64); sub_load(nwt, wt + (size_t)tile_of(nu + 1) * 64);
            STAMP(sid + 3)
            const UnitU us = load_unit(T.units + u), ut = load_unit(T.units + u + 1);
            STAMP(sid + 4)
            float vin[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { const float v = xs[m * a.xld + us.xoff + (k < us.cin ? k : 0)]; vin[k] = k < us.cin ? v : 0.f; }
            const int xcol = us.xoff + us.ku + (kq < us.r ? kq : 0);
            const float xold = xs[m * a.xld + xcol];
            const UnitU uu[2] = {us, ut};
            const f32x4 ww[2] = {ws, wtt};
            f32x4 vv[2], oo[2];
            sub_node_fwd(par4, a, uu, ww, vin, kq, vv, oo);
            const f32x4 vs = vv[0], vt = vv[1], so = oo[0], to = oo[1];
            STAMP(sid + 5)
            if (train) {
                bits[(us.wcol >> 4) * 64 + lane] = (uint8_t)sign_bits(vs);
                bits[(ut.wcol >> 4) * 64 + lane] = (uint8_t)sign_bits(vt);
                *(GLOBAL_AS f32x4*)(a2 + (m * a.WT + us.wcol + 4 * kq)) = vs;
                *(GLOBAL_AS f32x4*)(a2 + (m * a.WT + ut.wcol + 4 * kq)) = vt;
            }
            STAMP(sid + 6)
            const float sv = kq == 0 ? so.x : kq == 1 ? so.y : kq == 2 ? so.z : so.w;
            const float tv = kq == 0 ? to.x : kq == 1 ? to.y : kq == 2 ? to.z : to.w;
            if (kq < us.r) {
                const float aa = a.alpha * atanf(sv);
                // training: s goes to the tape ([n_levels + level][B][d], indexed by the lane it scales)
                if (!REV && tape != nullptr && row0 + m < a.B) tape[(size_t)(a.n_levels + level) * lvl + (size_t)(row0 + m) * a.d + xcol] = sv;
                float xn;
                if (!REV) { xn = expf(aa) * xold + tv; jpart += aa; }
                else      { xn = (xold - tv) * __builtin_amdgcn_rcpf(expf(aa)); jpart -= aa; }
                xs[m * a.xld + xcol] = xn;
            }
            STAMP(sid + 7)
            // (training: the node's four tape stores - two sign-byte rows, two a2 tiles, none of them predicated - are younger than the prefetch)
            sub_wait<TRAIN ? 4 : 0>();
            if (has_next) { ws = nws; wtt = nwt; have = nu; }
        }
        wave_sync();
        // training: the wavefront's lanes as they stand after the level (tape[level][B][d])
        if (!REV && tape != nullptr && level < a.n_levels - 1) {
            float* dst = tape + (size_t)level * lvl + (size_t)row0 * a.d + c0;
            for (int i = lane; i < ROWS * wc; i += 64) {
                const int r = fdiv(i, inv_wc), j = i - r * wc;
                if (row0 + r < a.B) dst[(size_t)r * a.d + j] = xs[r * a.xld + c0 + j];
            }
        }
        STAMP(sid + 8)
        ub = nb; ue = ne; level = nlevel;
    }
    const float js = kq_sum(jpart);
    if (kq == 0) (lds + a.sub_misc)[wave * ROWS + m] = js;
}

constexpr int SUB_LV = 4;       // floats per lane of a wavefront's columns of a level tile: 16 x (its lanes) <= 64 SUB_LV (hint_plan.cpp checks)

__device__ __forceinline__ void sub_load_byte(int& dst, const GLOBAL_AS uint8_t* p) {
    asm volatile("global_load_ubyte %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}

template <bool REV>
__device__ __forceinline__ void sub_apply(const KArgs& a, const Tables& T, float* lds, const GBlock& blk, float* xs, bool train,
                                          int row0, int wave, int lane, int sid) {
    if (!REV && train) sub_apply_t<REV, !REV>(a, T, lds, blk, xs, row0, wave, lane, sid);
    else sub_apply_t<REV, false>(a, T, lds, blk, xs, row0, wave, lane, sid);
}

// Both subnets of a subtree node, backward, in lockstep (hint_wl.hpp's expressions): vin = the coupling gradients of their r
// outputs, w = the W2^T fragment tiles, bits = the sign bytes of their a2 tiles, xin = the lanes their first layers saw.
// g1: the lane's g1 (features 4 kq .. +3 of batch row m); gv: g_v = W1^T g1 of row m (every lane of the row holds all four).
__device__ __forceinline__ void sub_node_bwd(const LDS_AS f32x4* par4, const KArgs& a, const UnitU (&u)[2], const f32x4 (&w)[2],
                                             const int (&bits)[2], const float (&vin)[2][4], const float (&xin)[4], int kq,
                                             f32x4 (&g1)[2], f32x4 (&gv)[2]) {
    f32x4 g2[2], acc[2], pre[2];
    {
        f32x4 w3[2][4];
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const LDS_AS f32x4* w3p = par4 + ((a.sub_pf + u[n].w3v) >> 2) + kq;
#pragma unroll
            for (int o = 0; o < 4; ++o) w3[n][o] = w3p[4 * o];
        }
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            g2[n] = fma4(w3[n][3], vin[n][3], fma4(w3[n][2], vin[n][2], fma4(w3[n][1], vin[n][1], fma4(w3[n][0], vin[n][0], zero4()))));
            mask_by_bits(g2[n], bits[n]);
            acc[n] = zero4();
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[n] = mfma4(w[n][i], g2[n][i], acc[n]);
    __builtin_amdgcn_sched_barrier(0);          // (the first layer's vectors behind the MFMAs: 40 registers that need not live through them)
    f32x4 w1[2][5];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const LDS_AS f32x4* w1p = par4 + (u[n].w1v >> 2) + kq;
#pragma unroll
        for (int k = 0; k < 5; ++k) w1[n][k] = w1p[4 * k];
    }
#pragma unroll
    for (int n = 0; n < 2; ++n)
        pre[n] = fma4(w1[n][3], xin[3], fma4(w1[n][2], xin[2], fma4(w1[n][1], xin[1], fma4(w1[n][0], xin[0], w1[n][4]))));   // (wl_layer1)
    f32x4 p[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        f32x4 g = acc[n];
        g.x = pre[n].x > 0.f ? g.x : 0.f; g.y = pre[n].y > 0.f ? g.y : 0.f; g.z = pre[n].z > 0.f ? g.z : 0.f; g.w = pre[n].w > 0.f ? g.w : 0.f;
        g1[n] = g;
#pragma unroll
        for (int o = 0; o < 4; ++o) p[n][o] = dot4(w1[n][o], g, 0.f);
    }
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int n = 0; n < 2; ++n) gv[n][o] = kq_sum(p[n][o]);
}

// dW1 | db1 of both subnets' tiles from their g1 (hint_wl.hpp: transposed through the wavefront's two scratch tiles, four
// MFMAs over the 16 rows each) into the workgroup's first-layer gradient slab; both in lockstep
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 sub_as_f32x4(u32x4 v) { return __builtin_bit_cast(f32x4, v); }      // (whole vectors by value: see hint_wgrad.hip as_f32)
__device__ __forceinline__ u32x4 sub_as_u32x4(f32x4 v) { return __builtin_bit_cast(u32x4, v); }
__device__ __forceinline__ void sub_dw1(const f32x4 (&g)[2], float* scratch, const float* xs, int xld, const UnitU (&u)[2], GLOBAL_AS float* tw,
                                        int tw_floats, bool first_tile, int lane) {
    const int nl = lane & 15, kq = lane >> 4;
    ((f32x4*)scratch)[lane] = g[0];
    ((f32x4*)scratch)[64 + lane] = g[1];
    asm volatile("" ::: "memory");          // (the wavefront's own LDS traffic is in order)
    const float* g1p = scratch + (kq + 16 * (nl >> 2)) * 4 + (nl & 3);
    const float* vp = xs + kq * xld + u[0].xoff + (nl < u[0].cin ? nl : 0);       // (both subnets read the same lanes)
    const float one = nl == u[0].cin ? 1.f : 0.f;
    float av[2][4], bv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { av[0][i] = g1p[16 * i]; av[1][i] = g1p[256 + 16 * i]; bv[i] = vp[4 * i * xld]; }      // rows 4 i + kq
    f32x4 dw[2] = {zero4(), zero4()};
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int n = 0; n < 2; ++n) dw[n] = mfma4(nl < u[0].cin ? bv[i] : one, av[n][i], dw[n]);
    // through a buffer descriptor of the workgroup's slab: the lanes without an element get an offset past its bound (their store is
    // dropped, their load reads zero), so that the two stores are ISSUED for every node - sub_bwd's hand-counted wait relies on them
    const int kcp = u[0].cin < 4 ? 4 : 8;
    const bool mine = nl < u[0].h && 4 * kq < kcp;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)tw, 0, tw_floats * 4, 0x00020000);
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int off = mine ? (u[n].bias1 + nl * kcp + 4 * kq) * 4 : 0x7ffffff0;
        f32x4 v = dw[n];
        if (!first_tile) {
            const u32x4 o = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
            v += sub_as_f32x4(o);
        }
        __builtin_amdgcn_raw_buffer_store_b128(sub_as_u32x4(v), rs, off, 0, 0);
    }
}

// Backward through the subtree groups, parents first.  On entry the boundary in front of the top subtree level has been
// worked off by the workgroup (hint_bwd.hip: the g_v partials of the general group above scattered, the level's coupling
// backward done: its coupling gradients wait in gst, xs / sb hold its level).  Per node: coupling backward (below the top
// level), both subnets (g2 on the fly, W2^T on the matrix pipe, relu'(a1) recomputed), their dW1 | db1, and the g_v of the
// node's inputs added to the wavefront's columns of the gradient tile gs.
__device__ __forceinline__ void sub_bwd(const KArgs& a, const Tables& T, float* lds, const GBlock& blk, const float* x, bool top,
                                        float* xs, float* sb, float* gs, float* gst, const float* gj, int row0, bool first_tile,
                                        int wave, int lane, int sid) {
    (void)sid;
    const int m = lane & 15, kq = lane >> 4;
    const LDS_AS f32x4* par4 = (const LDS_AS f32x4*)(lds + a.sub_par);
    float* scratch = lds + a.sub_misc + wave * 512;             // two fragment tiles (the transposition of the subnets' g1)
    const float* tape = (const float*)blk.tape;
    float* wsGST = (float*)blk.wsGST;
    const size_t lvl = (size_t)a.B * a.d;
    GLOBAL_AS float* tw = blk.wsSlab + a.thin_slab_off + (size_t)blockIdx.x * a.tw_floats;
    const GLOBAL_AS uint8_t* bits = (const GLOBAL_AS uint8_t*)(blk.actA1 + a.bits_off) + a.bits_stride + (size_t)(row0 >> 4) * (a.WT >> 4) * 64;
    const GLOBAL_AS f32x4* wt = (const GLOBAL_AS f32x4*)blk.packed + lane;
    const int c0 = lds_i32(T.rng + a.sub_cols + 4 * wave), c1 = lds_i32(T.rng + a.sub_cols + 4 * wave + 1);     // the lanes of its subtrees
    const int wc = c1 - c0;
    const float inv_wc = frcp(wc > 0 ? wc : 1);
    auto range = [&](int gidx, int& ub, int& ue, int& level) {
        const LDS_AS int32_t* gp = (const LDS_AS int32_t*)(T.groups + gidx);
        const int unit_begin = lds_i32(gp + 0), rngb = lds_i32(gp + 6);
        level = lds_i32(gp + 7);
        ub = unit_begin + lds_i32(T.rng + rngb + wave); ue = unit_begin + lds_i32(T.rng + rngb + wave + 1);
    };
    auto tile_of = [&](int u) -> int { return lds_i32((const LDS_AS int32_t*)(T.units + u) + 4); };     // Unit::b2 (W2^T)
    auto byte_of = [&](int u) -> const GLOBAL_AS uint8_t* { return bits + (lds_i32((const LDS_AS int32_t*)(T.units + u) + 9) >> 4) * 64 + lane; };   // Unit::wcol
    // ---- the wavefront's columns of the next level down (lanes as the forward saw them, s): tape -> registers while the level
    //      above runs, -> xs / sb when the level starts ----
    float lx[SUB_LV], ls[SUB_LV];
    auto level_fetch = [&](int gidx) {
        const int lv = lds_i32((const LDS_AS int32_t*)(T.groups + gidx) + 7);
        const float* xsrc = lv == 0 ? (top ? tape + (size_t)(a.n_levels - 1) * lvl : x) : tape + (size_t)(lv - 1) * lvl;
        const float* ssrc = tape + (size_t)(a.n_levels + lv) * lvl;
#pragma unroll
        for (int k = 0; k < SUB_LV; ++k) {
            const int i = lane + 64 * k;
            const int r = fdiv(i < ROWS * wc ? i : 0, inv_wc), j = (i < ROWS * wc ? i : 0) - r * wc;
            const size_t o = (size_t)(row0 + r < a.B ? row0 + r : row0) * a.d + c0 + j;
            lx[k] = xsrc[o]; ls[k] = ssrc[o];
        }
    };
    if (a.n_sub > 1) level_fetch(a.n_sub - 2);
    int ub, ue, level;
    range(a.n_sub - 1, ub, ue, level);
    f32x4 ws = zero4(), wtt = zero4();
    int bs = 0, bt = 0;
    int have = -1;                      // the node whose tiles and sign bytes are in ws / wtt / bs / bt
    for (int q = a.n_sub - 1; q >= 0; --q) {
        const bool is_top = q == a.n_sub - 1;
        int nb = 0, ne = 0, nlevel = 0;
        if (q > 0) range(q - 1, nb, ne, nlevel);
        if (!is_top) {
            // this level's lanes and s into the wavefront's columns of the tiles
#pragma unroll
            for (int k = 0; k < SUB_LV; ++k) {
                const int i = lane + 64 * k;
                if (i < ROWS * wc) {
                    const int r = fdiv(i, inv_wc), j = i - r * wc;
                    const bool ok = row0 + r < a.B;
                    xs[r * a.xld + c0 + j] = ok ? lx[k] : 0.f;
                    sb[r * a.xld + c0 + j] = ok ? ls[k] : 0.f;
                }
            }
            wave_sync();
            if (q > 0) level_fetch(q - 1);
        }
        for (int u = ub; u < ue; u += 2) {
            if (have != u) {
                sub_load(ws, wt + (size_t)tile_of(u) * 64); sub_load(wtt, wt + (size_t)tile_of(u + 1) * 64);
                sub_load_byte(bs, byte_of(u)); sub_load_byte(bt, byte_of(u + 1));
                sub_wait<0>();
            }
            STAMP(sid + 5)
            const bool has_next = u + 2 < ue || nb < ne;
            const int nu = u + 2 < ue ? u + 2 : (nb < ne ? nb : u);      // (behind the last node of all: this node's again - never under a branch)
            f32x4 nws, nwt;             // (written by the untracked loads only: no value of the compiler's may share their registers before sub_wait)
            int nbs, nbt;
            sub_load(nws, wt + (size_t)tile_of(nu) * 64); sub_load(nwt, wt + (size_t)tile_of(nu + 1) * 64);
            sub_load_byte(nbs, byte_of(nu)); sub_load_byte(nbt, byte_of(nu + 1));
            const UnitU us = load_unit(T.units + u), ut = load_unit(T.units + u + 1);
            STAMP(sid + 6)
            if (!is_top) {
                // coupling backward of the node: lane group kq < r takes the transformed lane xoff + ku + kq of batch row m
                if (kq < us.r) {
                    const int xcol = us.xoff + us.ku + kq;
                    const float gval = gs[m * a.xld + xcol];
                    const float sv = sb[m * a.xld + xcol];
                    const float aa = a.alpha * atanf(sv);
                    const float ea = expf(aa);
                    const float l = xs[m * a.xld + xcol];                 // lower input of the node
                    const float ga = gval * ea * l + gj[m];               // g_a (a feeds both l' and J)
                    const float gsv = ga * a.alpha * __builtin_amdgcn_rcpf(1.f + sv * sv);     // g_s (v_rcp_f32, 1 ulp)
                    gst[m * a.gld + us.lcol + kq] = gsv;
                    gst[m * a.gld + ut.lcol + kq] = gval;                 // g_t = g_l'
                    float* go = wsGST + (size_t)(row0 + m) * a.ST;
                    go[us.gcol + kq] = gsv;
                    go[ut.gcol + kq] = gval;
                    gs[m * a.xld + xcol] = gval * ea;                     // g_l
                }
                wave_sync();
            }
            STAMP(sid + 7)
            float vst[2][4], xin[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int jr = k < us.r ? k : 0, kc = k < us.cin ? k : 0;
                const float a0 = gst[m * a.gld + us.lcol + jr], a1 = gst[m * a.gld + ut.lcol + jr], a2 = xs[m * a.xld + us.xoff + kc];
                vst[0][k] = k < us.r ? a0 : 0.f; vst[1][k] = k < us.r ? a1 : 0.f; xin[k] = k < us.cin ? a2 : 0.f;
            }
            const UnitU uu[2] = {us, ut};
            const f32x4 ww[2] = {ws, wtt};
            const int bb[2] = {bs, bt};
            f32x4 g1[2], gvv[2];
            sub_node_bwd(par4, a, uu, ww, bb, vst, xin, kq, g1, gvv);
            const f32x4 gvs = gvv[0], gvt = gvv[1];
            STAMP(sid + 8)
            sub_dw1(g1, scratch, xs, a.xld, uu, tw, a.tw_floats, first_tile, lane);
            // g_v of the node's inputs: lane group kq < cin adds input kq of batch row m
            const float gsum = kq == 0 ? gvs.x + gvt.x : kq == 1 ? gvs.y + gvt.y : kq == 2 ? gvs.z + gvt.z : gvs.w + gvt.w;
            if (kq < us.cin) gs[m * a.xld + us.xoff + kq] += gsum;
            STAMP(sid + 9)
            sub_wait<2>();              // (the node's two slab stores - buffer stores, issued whatever the lanes' predicates - are younger than the prefetch)
            if (has_next) { ws = nws; wtt = nwt; bs = nbs; bt = nbt; have = nu; }
        }
        STAMP(sid + 10)
        wave_sync();
        ub = nb; ue = ne; level = nlevel;
    }
}

}  // namespace hint
