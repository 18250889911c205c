// Host side of the C ABI (include/hint_amd.h): turns the node list of one coupling tree
// (the structure /root/reference/hint.py:25-54 builds recursively) into a static level
// schedule in device memory, and launches the kernels of hint_kernels.hip / hint_optim.hip.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/hint_amd.h"
#include "hint_dev.h"

namespace hint {
hipError_t launch_pack(const PackSeg* segs, const int2* ptiles, int n_tiles, const int32_t* bmap, int n_bias,
                       long bias_off, const float* params, float* packed, hipStream_t stream);
hipError_t launch_pack_many(const PackItem* items, int n_items, int grid, float* zero_buf, int zero_floats,
                            unsigned long long* rng_state, float* opt_state, hipStream_t stream);
hipError_t launch_zero(float* p, long n, int num_cu, hipStream_t stream);
hipError_t launch_apply(bool rev, const KArgs& a, int lds_bytes, int grid, const ChainBlock& one,
                        const ChainBlock* chain, int n_chain, const float* x, const float* c, float* z, float* J,
                        const float* J_in, float* loss_acc, float noise, const unsigned long long* rng_state,
                        float* x_noisy, hipStream_t stream);
hipError_t launch_bwd(const KArgs& a, int lds_bytes, int grid, const ChainBlock& one, const ChainBlock* chain,
                      int n_chain, const float* x, const float* c, const float* g_z, const float* g_J,
                      float* g_x, float* g_c, float gz_scale, float gJ_const, hipStream_t stream);
hipError_t launch_dw(const DWJob* jobs, int n_jobs, int splits, const ChainBlock& one, const ChainBlock* chain,
                     int n_chain, int WT, int Bp, int rows_per_wg, const int32_t* tmap, int thin_total, int ntiles,
                     hipStream_t stream);
hipError_t set_max_lds(int fwd_bytes, int bwd_bytes);
hipError_t set_stamp_buffer(unsigned long long* p);
hipError_t launch_adam(float* p, float* g, float* m, float* v, long n, float lr_t, float b1, float b2,
                       float inv_sqrt_bc2, float eps, float wd, float gscale, float gclamp, int zero_grads,
                       int num_cu, const float* dev_state, hipStream_t stream);
}  // namespace hint

using namespace hint;

static thread_local std::string g_err;

static int fail(const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return 1;
}

#define HIP_TRY(expr)                                                              \
    do {                                                                           \
        hipError_t e_ = (expr);                                                    \
        if (e_ != hipSuccess) return fail("%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

static inline int pad16(int v) { return (v + 15) & ~15; }

// LDS row stride for a buffer read with ds_read_b128 by lanes (row = l&15, 16-byte column
// slot = l>>4): stride = 8 (mod 64) floats makes each 16-lane group of the instruction hit 64
// distinct banks (MI355X_MICROARCH.md §LDS).
static inline int lds_stride(int width) {
    int w = std::max(width, 16);
    int ld = ((w + 63) / 64) * 64 + 8;
    if (ld - 64 >= w) ld -= 64;
    return ld;
}

struct hint_plan {
    int device = -1;
    int d = 0, dc = 0, n_nodes = 0, n_groups = 0, n_levels = 0, n_dwjobs = 0, n_ptiles = 0;
    float alpha = 0.f;
    int64_t param_floats = 0, packed_floats = 0;
    int WT = 0;
    int xld = 0, cld = 0, ald = 0, vld = 0, sld = 0, max_aw = 0;
    int s3 = 1, sv = 1;   // max K-split slabs of the layer-3 / dv stages
    int lds_fwd = 0, lds_bwd = 0;
    int num_cu = 256;
    int meta_bytes = 0, vmap_off = 0, ents_off = 0, jmax = 0, bmax = 0, n_bias = 0, split_o3 = 0;
    int first[2][4] = {{0}};
    int thin_total = 0;
    int32_t* d_tmap = nullptr;
    int32_t* d_tbmap = nullptr;
    void* d_meta = nullptr;      // [groups | vnodes | ents]
    GJob* d_jobs = nullptr;      // per-group job lists (GJob and OJob records, 16 bytes each)
    int32_t* d_bmap = nullptr;
    DWJob* d_dwjobs = nullptr;
    PackSeg* d_segs = nullptr;
    int2* d_ptiles = nullptr;
};

static constexpr int LDS_LIMIT = 160 * 1024;
#ifndef HINT_JOB_OVERHEAD
#define HINT_JOB_OVERHEAD 1200
#endif
static constexpr int PERM_LDS_MAX = 16 * 1024;   // the chain's permutation matrices ride in LDS up to this size
static constexpr int WS_SLACK = 64;
static thread_local bool g_host_only = false;   // hint_plan_check: build and verify the plan, touch no device   // floats of slack at the end of every workspace array
static int g_bwd_stages = 3;          // profiling aid: bit0 = row-parallel part A, bit1 = weight-gradient part B

static int fwd_lds_bytes(int xld, int cld, int vld, int ald, int sld, int s3) {
    return 4 * ROWS * (2 * xld + cld + vld + 2 * ald + s3 * sld + 1);
}
static int bwd_lds_bytes(int xld, int cld, int vld, int ald, int sld, int s3, int sv, int n_abuf) {
    (void)s3;
    return 4 * ROWS * (3 * xld + 2 * cld + (1 + sv) * vld + n_abuf * ald + sld + 1);
}
static constexpr int JOBS_PER_GROUP_MAX = 2 * NTHREADS;   // what the in-kernel job prefetch moves
static constexpr int SPLIT_HP = 384;   // nodes with pad16(h) beyond this are planned one net at a time

// how many K-split slabs a thin stage gets: enough jobs to occupy the 8 wavefronts, each slab
// at least 2 k-blocks deep
static int pick_slabs(int n_tile_jobs, int min_nblk, int max_slabs) {
    if (n_tile_jobs <= 0) return 1;
    int s = NWAVES / n_tile_jobs;
    s = std::min(s, min_nblk / 2);
    return std::max(1, std::min(s, max_slabs));
}

static int build_plan(const hint_node_desc* nodes, int32_t n_nodes, int32_t d, int32_t dc, float clamp,
                      int max_slabs, int cap_scale, bool use_a3, hint_plan** out, bool* retry_smaller) {
    *retry_smaller = false;
    int max_depth = 0;
    for (int i = 0; i < n_nodes; ++i) max_depth = std::max(max_depth, nodes[i].depth);

    hint_plan* P = new hint_plan();
    P->d = d;
    P->dc = dc;
    P->n_nodes = n_nodes;
    P->n_levels = max_depth + 1;
    P->alpha = (float)((double)clamp * 0.636);   // hint.py:57,60 (python float product, then fp32)
    P->xld = pad16(d) + 4;
    P->cld = dc > 0 ? pad16(dc) + 4 : 0;

    // ---- forward order: deepest level first (children before parents, hint.py:70-73) ----
    std::vector<int> order;
    for (int dep = max_depth; dep >= 0; --dep)
        for (int i = 0; i < n_nodes; ++i)
            if (nodes[i].depth == dep) order.push_back(i);

    // Every single node must fit the 160 KiB LDS of the backward kernel (that fixes minimum
    // strides); groups then grow up to soft caps so that shallow-but-wide and deep-but-narrow
    // levels end up with similar footprints.
    int min_aw = 0, min_vw = 0, min_sw = 0;
    for (int i = 0; i < n_nodes; ++i) {
        min_aw = std::max(min_aw, (pad16(nodes[i].h) > SPLIT_HP ? 1 : 2) * pad16(nodes[i].h));
        min_vw = std::max(min_vw, pad16(nodes[i].k + dc));
        min_sw = std::max(min_sw, 2 * pad16(nodes[i].r));
    }
    const int cap_aw = std::max(min_aw, 512 / cap_scale), cap_vw = std::max(min_vw, 128 / cap_scale),
              cap_sw = std::max(min_sw, 256 / cap_scale);
    // rough size of what rides along in LDS besides the float buffers (group/node/lane tables)
    const int meta_guess = 16 * n_nodes + 8 * d + 160 * (max_depth + 2);
    auto bwd_bytes = [&](int aw_, int vw_, int sw_) {
        return meta_guess + bwd_lds_bytes(P->xld, P->cld, lds_stride(vw_), lds_stride(aw_), lds_stride(sw_), 1, 1, 2);
    };

    // Host-side working copy of a node - or of ONE of its two nets: a node whose two nets do not fit
    // the LDS side by side (h > 384) is planned as two units in two consecutive single-unit groups,
    // the t net first (forward order), then the s net, which carries the coupling.  Both units keep
    // the node's full [s | t] column layout in the s/t and g_st buffers (those are not cleared
    // between the two groups), everything else is per net.
    struct DNode {
        int off, k, r, h, cin, hp, rp, cinp, acol, vcol, scol, wcol;
        int net0, nn;         // nets of this unit: [net0, net0 + nn)
        bool couples;         // this unit runs the node's coupling (false for the t-only unit)
    };
    auto has_net = [](const DNode& q, int net) { return net >= q.net0 && net < q.net0 + q.nn; };
    auto acol_of = [](const DNode& q, int net) { return q.acol + (net - q.net0) * q.hp; };
    std::vector<DNode> dn;
    std::vector<const hint_node_desc*> src;   // parallel to dn
    std::vector<DGroup> dg;
    std::vector<GJob> jobs;                   // all groups' job lists (GJob and OJob records)
    std::vector<Ent> ents;
    std::vector<int16_t> vmap;
    std::vector<int32_t> tmap;     // compact thin-gradient index -> offset in the flat gradient buffer
    std::vector<int32_t> tbmap;    // like bmap, but compact thin indices (bias gradients)
    std::vector<int32_t> bmap;
    std::vector<DWJob> dwj;
    std::vector<PackSeg> segs;
    std::vector<int2> ptiles;
    int wcol = 0;
    int max_aw = 0, max_vw = 0, max_sw = 0;
    int64_t pmax = 0, packed = 0;

    auto add_seg = [&](int N, int K, int NB, int ld, int mode, int64_t s0, int64_t s1, int hp, int h) -> int64_t {
        PackSeg sg{};
        sg.dst = packed; sg.src0 = s0; sg.src1 = s1; sg.N = N; sg.K = K; sg.NB = NB; sg.ld = ld; sg.mode = mode;
        sg.hp = hp; sg.h = h; sg.tile_begin = (int)ptiles.size();
        const int NT = (N + 15) / 16;
        for (int nt = 0; nt < NT; ++nt) ptiles.push_back(int2{(int)segs.size(), nt});
        segs.push_back(sg);
        packed += (int64_t)NT * NB * 256;
        return sg.dst;
    };

    size_t pos = 0;
    bool split_pending = false;       // the t unit of order[pos] is planned, its s unit comes next
    while (pos < order.size()) {
        DGroup g{};
        g.node_begin = (int)dn.size();
        g.wcol0 = wcol;
        int aw = 0, vw = 0, sw = 0;
        const int depth = nodes[order[pos]].depth;
        while (pos < order.size() && nodes[order[pos]].depth == depth) {
            const hint_node_desc& n = nodes[order[pos]];
            const int hp = pad16(n.h), rp = pad16(n.r), cin = n.k + dc, cinp = pad16(cin);
            const bool split = hp > SPLIT_HP;
            const int unit_aw = split ? hp : 2 * hp;
            if (bwd_bytes(unit_aw, cinp, 2 * rp) > LDS_LIMIT) {
                delete P;
                return fail("hint_plan_create: a node with h=%d, cin=%d, r=%d does not fit the 160 KiB LDS", n.h, cin, n.r);
            }
            // start a new group at the same depth when this node would overflow the budget; the two
            // units of a split node are single-unit groups
            if (aw > 0 && (split || aw + unit_aw > cap_aw || vw + cinp > cap_vw || sw + 2 * rp > cap_sw ||
                           bwd_bytes(std::max(min_aw, aw + unit_aw), std::max(min_vw, vw + cinp),
                                     std::max(min_sw, sw + 2 * rp)) > LDS_LIMIT))
                break;
            DNode q{};
            q.off = n.off; q.k = n.k; q.r = n.r; q.h = n.h; q.cin = cin;
            q.hp = hp; q.rp = rp; q.cinp = cinp;
            q.acol = aw; q.vcol = vw; q.scol = sw; q.wcol = wcol;
            q.net0 = 0; q.nn = 2; q.couples = true;
            if (split) {                     // t unit now, s unit (with the coupling) as the next group
                q.net0 = split_pending ? 0 : 1; q.nn = 1; q.couples = split_pending;
            }
            const int64_t sizes[6] = {(int64_t)n.h * cin, n.h, (int64_t)n.h * n.h, n.h, (int64_t)n.r * n.h, n.r};
            for (int t = 0; t < 12; ++t) pmax = std::max(pmax, n.p_off[t] + sizes[t % 6]);
            aw += unit_aw; vw += cinp; sw += 2 * rp; wcol += unit_aw;
            dn.push_back(q);
            src.push_back(&n);
            if (split) {
                if (!split_pending) { split_pending = true; break; }     // same node again: its s unit
                split_pending = false;
                ++pos;
                break;
            }
            ++pos;
        }
        g.node_end = (int)dn.size();
        g.aw = aw; g.vw = vw; g.sw = sw;
        g.level = max_depth - depth;
        g.level_last = (!split_pending && (pos >= order.size() || nodes[order[pos]].depth != depth)) ? 1 : 0;
        // Units of a split node: the coupling needs s and t, so it runs in whichever unit comes second
        // (forward: the s unit, inverse: the t unit); the backward pass, which only needs s, couples in
        // the s unit (first in its order) and the t unit after it must keep the g_st columns written there.
        g.pad = dn.back().nn == 2 ? 0 : (dn.back().net0 == 1 ? 1 : 2);      // 0 whole nodes, 1 t unit, 2 s unit
        max_aw = std::max(max_aw, aw); max_vw = std::max(max_vw, vw); max_sw = std::max(max_sw, sw);

        // ---- K-split factors of the thin stages of this group ----
        int l3_tiles = 0, dv_tiles = 0, min_hb = 1 << 30;
        for (int ni = g.node_begin; ni < g.node_end; ++ni) {
            l3_tiles += 2 * (dn[ni].rp / 16);        // (a split node's units use the slab count of the whole node)
            dv_tiles += dn[ni].cinp / 16;
            min_hb = std::min(min_hb, dn[ni].hp / 16);
        }
        g.l3_slabs = pick_slabs(l3_tiles, min_hb, max_slabs);
        g.dv_slabs = pick_slabs(dv_tiles, (dn[g.node_begin].nn == 1 ? 1 : 2) * min_hb, max_slabs);
        P->s3 = std::max(P->s3, g.l3_slabs);
        P->sv = std::max(P->sv, g.dv_slabs);

        // ---- packed weight segments + GEMM tile jobs ----
        struct NodePack { int64_t f1[2], f2[2], f3[2], b3[2], b2[2], bdv; };
        std::vector<NodePack> np(g.node_end - g.node_begin);
        for (int ni = g.node_begin; ni < g.node_end; ++ni) {
            const DNode& q = dn[ni];
            const hint_node_desc& n = *src[ni];
            NodePack& k = np[ni - g.node_begin];
            for (int net = 0; net < 2; ++net) {
                if (!has_net(q, net)) continue;
                const int64_t* po = n.p_off + net * 6;
                k.f1[net] = add_seg(q.h, q.cin, q.cinp / 16, q.cin, 0, po[0], 0, q.hp, q.h);   // v  -> a1
                k.f2[net] = add_seg(q.h, q.h, q.hp / 16, q.h, 0, po[2], 0, q.hp, q.h);         // a1 -> a2
                k.f3[net] = add_seg(q.r, q.h, q.hp / 16, q.h, 0, po[4], 0, q.hp, q.h);         // a2 -> s|t
                k.b3[net] = add_seg(q.h, q.r, q.rp / 16, q.h, 1, po[4], 0, q.hp, q.h);         // g_st -> g2
                k.b2[net] = add_seg(q.h, q.h, q.hp / 16, q.h, 1, po[2], 0, q.hp, q.h);         // g2 -> g1
            }
            if (q.nn == 2)
                k.bdv = add_seg(q.cin, 2 * q.hp, 2 * q.hp / 16, q.cin, 2, n.p_off[0], n.p_off[6], q.hp, q.h);   // g1 -> g_v
            else        // one net: the "stack" is that net's W1 alone
                k.bdv = add_seg(q.cin, q.hp, q.hp / 16, q.cin, 2, n.p_off[6 * q.net0], n.p_off[6 * q.net0], q.hp, q.h);
        }
        // ---- thin weight gradients done inside the row-parallel backward kernel: 16x16 outer-product
        //      tiles  T[m][n] = sum_rows A[row][acol+m] * B[row][bcol+n]  of dW3 = g_st^T a2 (with the g2
        //      stage) and dW1 = g1^T v (with the dv stage), stored at slab[goff + m*N + n] ----
        struct OuterTile { int32_t goff; int acol, bcol, mvalid, nvalid, N, cnt, dir; };   // cnt tiles along dir (0: n, 1: m)
        struct OuterMat { int base, acol, bcol, M, N; };       // one thin gradient matrix [M x N]
        std::vector<OuterMat> outer3, outer1;
        auto thin_alloc = [&](int64_t param_off, int count) {   // contiguous compact range mirroring a tensor
            const int base = (int)tmap.size();
            for (int i = 0; i < count; ++i) tmap.push_back((int32_t)(param_off + i));
            return base;
        };
        // Records of adjacent tiles along a matrix's longer side (they share the other side's operand and
        // one decode / baton hand-over): `per` evenly sized records per row of tiles.
        auto outer_records = [&](const std::vector<OuterMat>& mats, int per, std::vector<OuterTile>& out) {
            out.clear();
            for (const OuterMat& m : mats) {
                const int MT = (m.M + 15) / 16, NTl = (m.N + 15) / 16;
                const int dir = MT > NTl ? 1 : 0, L = dir ? MT : NTl, O = dir ? NTl : MT;
                const int nrec = std::max(std::min(L, per), (L + 126) / 127);
                for (int o = 0; o < O; ++o)
                    for (int rc = 0, l0 = 0; rc < nrec; ++rc) {
                        const int cnt = (L - l0 + (nrec - rc) - 1) / (nrec - rc);
                        const int mt = dir ? l0 : o, nt = dir ? o : l0;
                        const int mlast = dir ? l0 + cnt - 1 : o, nlast = dir ? o : l0 + cnt - 1;
                        out.push_back(OuterTile{m.base + 16 * mt * m.N + 16 * nt, m.acol + 16 * mt, m.bcol + 16 * nt,
                                                std::min(16, m.M - 16 * mlast), std::min(16, m.N - 16 * nlast), m.N, cnt, dir});
                        l0 += cnt;
                    }
            }
        };
        for (int ni = g.node_begin; ni < g.node_end; ++ni)
            for (int net = 0; net < 2; ++net) {            // dW3[r][h] = g_st^T a2
                const DNode& q = dn[ni];
                if (!has_net(q, net)) continue;
                outer3.push_back(OuterMat{thin_alloc(src[ni]->p_off[net * 6 + 4], q.r * q.h), q.scol + net * q.rp,
                                          acol_of(q, net), q.r, q.h});
            }
        for (int ni = g.node_begin; ni < g.node_end; ++ni)
            for (int net = 0; net < 2; ++net) {            // dW1[h][cin] = g1^T v
                const DNode& q = dn[ni];
                if (q.cin == 0 || !has_net(q, net)) continue;
                outer1.push_back(OuterMat{thin_alloc(src[ni]->p_off[net * 6 + 0], q.h * q.cin), acol_of(q, net), q.vcol,
                                          q.h, q.cin});
            }
        // one (slab, node, net) of a stage: NT adjacent output tiles over the same k-blocks
        struct Segment { int64_t wtile; int NT, nb, tstride, acol, ocol, N, slab; };
        auto emit_stage = [&](int which) -> int {
            // which: 1 = L1, 2 = L2, 3 = L3, 4 = g2, 5 = g1, 6 = dv
            std::vector<Segment> segs;
            const int slabs = which == 3 ? g.l3_slabs : (which == 6 ? g.dv_slabs : 1);
            for (int sl = 0; sl < slabs && which != 7; ++sl)
                for (int ni = g.node_begin; ni < g.node_end; ++ni) {
                    const DNode& q = dn[ni];
                    const NodePack& k = np[ni - g.node_begin];
                    const int nets = which == 6 ? 1 : 2;
                    for (int net = 0; net < nets; ++net) {
                        if (which != 6 && !has_net(q, net)) continue;
                        int N, NB, acol, ocol0; int64_t wbase;
                        const int ac = which == 6 ? q.acol : acol_of(q, net);
                        switch (which) {
                            case 1: N = q.h; NB = q.cinp / 16; acol = q.vcol; ocol0 = ac; wbase = k.f1[net]; break;
                            case 2: N = q.h; NB = q.hp / 16; acol = ac; ocol0 = ac; wbase = k.f2[net]; break;
                            case 3: N = q.r; NB = q.hp / 16; acol = ac; ocol0 = q.scol + net * q.rp; wbase = k.f3[net]; break;
                            case 4: N = q.h; NB = q.rp / 16; acol = q.scol + net * q.rp; ocol0 = ac; wbase = k.b3[net]; break;
                            case 5: N = q.h; NB = q.hp / 16; acol = ac; ocol0 = ac; wbase = k.b2[net]; break;
                            default: N = q.cin; NB = q.nn * q.hp / 16; acol = q.acol; ocol0 = q.vcol; wbase = k.bdv; break;
                        }
                        // this slab's share of the k-blocks
                        const int kb0 = (int)((int64_t)NB * sl / slabs), kb1 = (int)((int64_t)NB * (sl + 1) / slabs);
                        Segment sg{};
                        sg.NT = (N + 15) / 16;
                        if (sg.NT == 0) continue;
                        sg.N = N; sg.ocol = ocol0; sg.slab = sl; sg.tstride = std::max(NB, 1);
                        if (kb1 > kb0) { sg.wtile = wbase / 256 + kb0; sg.nb = kb1 - kb0; sg.acol = acol + kb0 * 16; }
                        else { sg.wtile = 0; sg.nb = 0; sg.acol = 0; sg.tstride = 0; }   // K = 0 (cin = 0) or empty slab
                        segs.push_back(sg);
                    }
                }
            // Cut the segments into jobs of <= 3 tiles and deal them to the wavefronts.  More, smaller
            // jobs balance better, fewer, wider ones share more A reads and pay fewer prologues: try
            // every total job count from the minimum up and keep the cheapest estimated makespan.
            // Cost model (cycles, from in-kernel stamps): 128 per tile and k-block on the SIMD's matrix pipe,
            // which the two wavefronts of a SIMD (w, w+4) share, plus a per-job prologue/epilogue - decode,
            // dispatch, LDS round trips, the baton hand-over - that the partner hides only in part and that
            // dwarfs the pipe time of a thin job; an outer-product record costs about as much plus 150 per tile.
            const long JOB_OVERHEAD = HINT_JOB_OVERHEAD, OUTER_OVERHEAD = HINT_JOB_OVERHEAD, OUTER_TILE = 150;
            struct Cut { int seg, t0, nt; long cost, overhead; };   // seg < 0: outer tile -1-seg
            // which == 7: the dW3 tiles as a stage of their own (plans without LDS for the g2 buffer)
            const std::vector<OuterMat>* outer_mats = (which == 4 && use_a3) || which == 7 ? &outer3 : (which == 6 ? &outer1 : nullptr);
            std::vector<OuterTile> orec, best_orec;
            auto cut_segments = [&](int extra, std::vector<Cut>& cuts) {
                // segment s gets ceil(NT/3) jobs plus a share of `extra` (largest work per job first)
                std::vector<int> cnt(segs.size());
                for (size_t i = 0; i < segs.size(); ++i) cnt[i] = (segs[i].NT + 2) / 3;
                for (int e = 0; e < extra; ++e) {
                    int best = -1; double bw = 0;
                    for (size_t i = 0; i < segs.size(); ++i) {
                        if (cnt[i] >= segs[i].NT) continue;
                        const double w = (double)segs[i].NT * std::max(segs[i].nb, 1) / cnt[i];
                        if (w > bw) { bw = w; best = (int)i; }
                    }
                    if (best < 0) break;
                    ++cnt[best];
                }
                cuts.clear();
                for (size_t i = 0; i < segs.size(); ++i) {
                    int t0 = 0;
                    for (int c = 0; c < cnt[i]; ++c) {
                        const int nt = (segs[i].NT - t0 + (cnt[i] - c) - 1) / (cnt[i] - c);
                        cuts.push_back(Cut{(int)i, t0, nt, 128L * nt * std::max(segs[i].nb, 1), JOB_OVERHEAD});
                        t0 += nt;
                    }
                }
                for (size_t i = 0; i < orec.size(); ++i) cuts.push_back(Cut{-1 - (int)i, 0, 1, OUTER_TILE * orec[i].cnt, OUTER_OVERHEAD});
            };
            auto deal = [&](const std::vector<Cut>& cuts, std::vector<std::vector<int>>& per_wave) -> long {
                std::vector<int> idx(cuts.size());
                for (size_t i = 0; i < idx.size(); ++i) idx[i] = (int)i;
                std::stable_sort(idx.begin(), idx.end(), [&](int x, int y) {
                    return cuts[x].cost + cuts[x].overhead > cuts[y].cost + cuts[y].overhead; });
                per_wave.assign(NWAVES, {});
                long wload[NWAVES] = {0}, sload[4] = {0};
                for (int i : idx) {
                    int w = 0;
                    for (int v = 1; v < NWAVES; ++v) {
                        const long sv = sload[v & 3], sw = sload[w & 3];
                        if (sv < sw || (sv == sw && wload[v] < wload[w])) w = v;
                    }
                    per_wave[w].push_back(i);
                    wload[w] += cuts[i].cost + cuts[i].overhead;
                    sload[w & 3] += cuts[i].cost + cuts[i].overhead / 2;
                }
                long worst = 0;
                for (int w = 0; w < NWAVES; ++w) worst = std::max(worst, std::max(wload[w], sload[w & 3]));
                return worst;
            };
            std::vector<Cut> cuts, best_cuts;
            std::vector<std::vector<int>> per_wave, best_pw(NWAVES);
            long best_cost = -1;
            int total_tiles = 0;
            for (const Segment& sg : segs) total_tiles += sg.NT;
            int max_run = 1;      // longest row of tiles of the stage's thin gradient matrices
            if (outer_mats)
                for (const OuterMat& m : *outer_mats) max_run = std::max(max_run, (std::max(m.M, m.N) + 15) / 16);
            for (int per = 1; per <= std::min(max_run, 16); ++per) {     // records per row of outer-product tiles
                if (outer_mats) outer_records(*outer_mats, per, orec);
                for (int extra = 0; extra <= 2 * NWAVES; ++extra) {
                    cut_segments(extra, cuts);
                    const long c = deal(cuts, per_wave);
                    if (best_cost < 0 || c < best_cost) { best_cost = c; best_cuts = cuts; best_pw = per_wave; best_orec = orec; }
                    if ((int)cuts.size() >= total_tiles + (int)orec.size()) break;
                }
            }
            const int hdr = (int)jobs.size() - g.jl_begin;
            std::vector<std::vector<TJob>> lists(NWAVES);
            size_t longest = 1;
            for (int w = 0; w < NWAVES; ++w) {
                for (int i : best_pw[w]) {
                    const Cut& c = best_cuts[i];
                    TJob t{};
                    if (c.seg < 0) {                       // outer-product tile (see TJob)
                        const OuterTile& o = best_orec[-1 - c.seg];
                        if (o.N > 0xffff) return -1;
                        t.wtile = o.goff;
                        t.acol = (uint16_t)o.acol; t.ocol = (uint16_t)o.bcol;
                        t.nb = 0; t.nt = TJOB_OUTER;
                        t.slab = (uint8_t)(o.dir | (o.cnt << 1));
                        t.nvalid = (uint8_t)((o.mvalid - 1) | ((o.nvalid - 1) << 4));
                        t.tstride = (uint16_t)o.N;
                        lists[w].push_back(t);
                        continue;
                    }
                    const Segment& sg = segs[c.seg];
                    const int64_t wt = sg.wtile + (int64_t)c.t0 * sg.tstride;
                    if (wt > 0x7fffffff || sg.tstride > 0xffff) return -1;
                    t.wtile = (int32_t)wt;
                    t.acol = (uint16_t)sg.acol;
                    t.ocol = (uint16_t)(sg.ocol + 16 * c.t0);
                    t.nb = (uint8_t)sg.nb;
                    t.nt = (uint8_t)c.nt;
                    t.nvalid = (uint8_t)std::min(16, sg.N - 16 * (c.t0 + c.nt - 1));
                    t.slab = (uint8_t)sg.slab;
                    t.tstride = (uint16_t)sg.tstride;
                    lists[w].push_back(t);
                }
                // an idle wavefront still has one record (nt = 0): it only hands the prefetch baton on
                if (lists[w].empty()) lists[w].push_back(TJob{});
                longest = std::max(longest, lists[w].size());
            }
            // ---- self-check (always on; also what hint_plan_check exists for): read the records back the
            //      way the kernels do and make sure every output tile of the stage is produced exactly
            //      once, by the right k-blocks, and every outer-product tile exactly once ----
            {
                std::vector<std::vector<int>> seen(segs.size());
                for (size_t i = 0; i < segs.size(); ++i) seen[i].assign(segs[i].NT, 0);
                std::vector<int64_t> outer_seen;
                for (int w = 0; w < NWAVES; ++w)
                    for (const TJob& t : lists[w]) {
                        if (t.nt == 0) continue;
                        if (t.nt == TJOB_OUTER) {
                            const int dir = t.slab & 1, cnt = t.slab >> 1;
                            if (cnt < 1) return -2;
                            for (int k = 0; k < cnt; ++k) outer_seen.push_back((int64_t)t.wtile + (int64_t)k * (dir ? 16 * t.tstride : 16));
                            continue;
                        }
                        bool found = false;
                        for (size_t i = 0; i < segs.size() && !found; ++i) {
                            const Segment& sg = segs[i];
                            if (sg.slab != t.slab || t.ocol < sg.ocol || t.ocol >= sg.ocol + 16 * sg.NT || sg.nb != t.nb) continue;
                            const int t0 = (t.ocol - sg.ocol) / 16;
                            if ((t.ocol - sg.ocol) % 16 || t0 + t.nt > sg.NT || t.nt > 3) return -2;
                            if (sg.nb > 0 && (t.wtile != sg.wtile + (int64_t)t0 * sg.tstride || t.acol != sg.acol || t.tstride != sg.tstride)) continue;
                            const int want_valid = std::min(16, sg.N - 16 * (t0 + t.nt - 1));
                            if (t.nvalid != want_valid) return -2;
                            for (int k = 0; k < t.nt; ++k) ++seen[i][t0 + k];
                            found = true;
                        }
                        if (!found) return -2;
                    }
                for (size_t i = 0; i < segs.size(); ++i)
                    for (int c : seen[i]) if (c != 1) return -2;
                std::vector<int64_t> outer_want;
                if (outer_mats)
                    for (const OuterMat& m : *outer_mats)
                        for (int mt = 0; mt * 16 < m.M; ++mt)
                            for (int nt = 0; nt * 16 < m.N; ++nt) outer_want.push_back((int64_t)m.base + 16 * mt * m.N + 16 * nt);
                std::sort(outer_seen.begin(), outer_seen.end());
                std::sort(outer_want.begin(), outer_want.end());
                if (outer_seen != outer_want) return -2;
            }
            const int stride = (int)longest;
            for (int w = 0; w < NWAVES; ++w) {
                if (lists[w].size() > 0xffff) return -1;
                lists[w][0].count = (uint16_t)lists[w].size();
                lists[w].resize(stride, TJob{});
                for (const TJob& c : lists[w]) jobs.push_back(c);
            }
            if (hdr > 0xffff || stride > 0x7fff) return -1;
            return STAGE_DESC(hdr, stride);
        };
        g.jl_begin = (int)jobs.size();
        g.l1_off = emit_stage(1);
        g.l2_off = emit_stage(2);
        g.l3_off = emit_stage(3);
        g.g2_off = emit_stage(4);
        g.g1_off = emit_stage(5);
        g.dv_off = emit_stage(6);
        g.o3_off = use_a3 ? 0 : emit_stage(7);
        if (g.o3_off == -2 || g.l1_off == -2 || g.l2_off == -2 || g.l3_off == -2 || g.g2_off == -2 || g.g1_off == -2 ||
            g.dv_off == -2) {
            delete P;
            return fail("hint_plan_create: internal error, a stage's job lists do not cover its tiles exactly once");
        }
        if (g.o3_off < 0 || g.l1_off < 0 || g.l2_off < 0 || g.l3_off < 0 || g.g2_off < 0 || g.g1_off < 0 || g.dv_off < 0) {
            delete P;
            if (cap_scale < 16) { *retry_smaller = true; return 1; }
            return fail("hint_plan_create: a group's job lists exceed the 16-bit stage descriptor");
        }
        g.o3_cnt = g.o1_off = g.o1_cnt = 0;      // outer-product tiles ride in the g2 / dv stage lists (or o3_off's)
        g.jl_count = (int)jobs.size() - g.jl_begin;
        if (g.jl_count > JOBS_PER_GROUP_MAX) {
            delete P;
            if (cap_scale < 16) { *retry_smaller = true; return 1; }
            return fail("hint_plan_create: a group needs %d tile jobs (max %d)", g.jl_count, JOBS_PER_GROUP_MAX);
        }
        P->jmax = std::max(P->jmax, g.jl_count);
        P->bmax = std::max(P->bmax, 2 * aw + sw);
        g.bmap_begin = (int)bmap.size();
        for (int layer = 0; layer < 2; ++layer)          // b1 map, then b2 map, aw entries each
            for (int ni = g.node_begin; ni < g.node_end; ++ni)
                for (int net = 0; net < 2; ++net)
                {
                    if (!has_net(dn[ni], net)) continue;
                    const int tb = thin_alloc(src[ni]->p_off[net * 6 + (layer ? 3 : 1)], dn[ni].h);
                    for (int j = 0; j < dn[ni].hp; ++j) {
                        bmap.push_back(j < dn[ni].h ? (int32_t)(src[ni]->p_off[net * 6 + (layer ? 3 : 1)] + j) : -1);
                        tbmap.push_back(j < dn[ni].h ? tb + j : -1);
                    }
                }
        g.bmap3_begin = (int)bmap.size();
        for (int ni = g.node_begin; ni < g.node_end; ++ni)
            for (int net = 0; net < 2; ++net)
            {
                // (the [s | t] column layout is kept by both units of a split node; the absent net's
                // columns carry no bias and no gradient slot)
                const bool here = has_net(dn[ni], net);
                const int tb = here ? thin_alloc(src[ni]->p_off[net * 6 + 5], dn[ni].r) : 0;
                for (int j = 0; j < dn[ni].rp; ++j) {
                    bmap.push_back(here && j < dn[ni].r ? (int32_t)(src[ni]->p_off[net * 6 + 5] + j) : -1);
                    tbmap.push_back(here && j < dn[ni].r ? tb + j : -1);
                }
            }

        g.vmap_begin = (int)vmap.size();
        for (int ni = g.node_begin; ni < g.node_end; ++ni)
            for (int j = 0; j < dn[ni].cinp; ++j)
                vmap.push_back(j < dn[ni].k ? (int16_t)(dn[ni].off + j)
                                             : (j < dn[ni].cin ? (int16_t)(-2 - (j - dn[ni].k)) : (int16_t)-1));
        g.ent_begin = (int)ents.size();
        for (int ni = g.node_begin; ni < g.node_end; ++ni)
            for (int j = 0; j < dn[ni].r; ++j)       // (both units of a split node list the lanes: which one couples depends on the direction)
                ents.push_back(Ent{(int16_t)(dn[ni].off + dn[ni].k + j), (int16_t)(dn[ni].scol + j),
                                   (int16_t)(dn[ni].scol + dn[ni].rp + j), 0});
        g.ent_cnt = (int)ents.size() - g.ent_begin;
        dg.push_back(g);
    }
    if (pmax >= (int64_t)1 << 31 || packed >= (int64_t)1 << 31) {
        delete P;
        return fail("hint_plan_create: block too large (parameter offsets must fit 31 bits)");
    }
    P->n_groups = (int)dg.size();
    P->WT = wcol;
    P->param_floats = pmax;
    P->packed_floats = packed;
    P->ald = lds_stride(max_aw);
    P->max_aw = max_aw;
    P->vld = lds_stride(max_vw);
    P->sld = lds_stride(max_sw);
    // ---- meta blob staged in LDS by the kernels: [groups | vmap | ents] ----
    auto up16 = [](size_t v) { return (v + 15) & ~(size_t)15; };
    const size_t groups_bytes = up16(dg.size() * sizeof(DGroup));
    const size_t vmap_bytes = up16(vmap.size() * sizeof(int16_t));
    const size_t ents_bytes = up16(ents.size() * sizeof(Ent));
    P->vmap_off = (int)groups_bytes;
    P->ents_off = (int)(groups_bytes + vmap_bytes);
    P->meta_bytes = (int)(groups_bytes + vmap_bytes + ents_bytes);
    std::vector<char> meta(P->meta_bytes, 0);
    std::memcpy(meta.data(), dg.data(), dg.size() * sizeof(DGroup));
    if (!vmap.empty()) std::memcpy(meta.data() + P->vmap_off, vmap.data(), vmap.size() * sizeof(int16_t));
    if (!ents.empty()) std::memcpy(meta.data() + P->ents_off, ents.data(), ents.size() * sizeof(Ent));
    for (int o = 0; o < 2; ++o) {
        const DGroup& fg = o == 0 ? dg.front() : dg.back();
        P->first[o][0] = fg.jl_begin; P->first[o][1] = fg.jl_count;
        P->first[o][2] = fg.bmap_begin; P->first[o][3] = 2 * fg.aw + fg.sw;
    }
    P->n_bias = (int)bmap.size();
    P->split_o3 = use_a3 ? 0 : 1;
    const int fixed = P->meta_bytes + 2 * P->jmax * (int)sizeof(GJob) + 4 * P->bmax * 4;   // [biases | bias-gradient map] x 2
    P->lds_fwd = fixed + fwd_lds_bytes(P->xld, P->cld, P->vld, P->ald, P->sld, P->s3);
    P->lds_bwd = fixed + bwd_lds_bytes(P->xld, P->cld, P->vld, P->ald, P->sld, P->s3, P->sv, use_a3 ? 3 : 2);
    if (P->lds_bwd > LDS_LIMIT || P->lds_fwd > LDS_LIMIT || d + 16 > 32000 || max_aw > 60000 ||
        P->bmax > 4 * NTHREADS) {
        const int need = std::max(P->lds_bwd, P->lds_fwd);
        const bool can_retry = use_a3 || max_slabs > 1 || cap_scale < 16;
        delete P;
        if (can_retry) { *retry_smaller = true; return 1; }   // rebuild with fewer slabs / smaller groups
        return fail("hint_plan_create: block needs %d bytes of LDS (> %d); d/dc/h too large", need, LDS_LIMIT);
    }

    // ---- weight-gradient jobs: 48x48 output tiles of dW2 of every (node, net) ----
    for (size_t ni = 0; ni < dn.size(); ++ni)
        for (int net = 0; net < 2; ++net) {
            if (!has_net(dn[ni], net)) continue;
            const int T = dn[ni].hp / 16;          // padded extent in 16-wide tiles, cut into groups of <= 3
            for (int mt = 0; mt < T; mt += 3)
                for (int nt = 0; nt < T; nt += 3)
                    dwj.push_back(DWJob{dn[ni].wcol + (net - dn[ni].net0) * dn[ni].hp, dn[ni].h, mt * 16, nt * 16, std::min(3, T - mt),
                                        std::min(3, T - nt), src[ni]->p_off[net * 6 + 2]});
        }
    P->n_dwjobs = (int)dwj.size();
    P->n_ptiles = (int)ptiles.size();

    P->thin_total = (int)tmap.size();
    if (g_host_only) {           // hint_plan_check: everything above ran (and checked itself); no device
        P->num_cu = 256;
        *out = P;
        return 0;
    }
    // ---- upload ----
    HIP_TRY(hipGetDevice(&P->device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, P->device));
    P->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    auto upload = [](void** dst, const void* srcp, size_t bytes) -> hipError_t {
        hipError_t e = hipMalloc(dst, std::max<size_t>(bytes, 16));
        if (e != hipSuccess) return e;
        return bytes ? hipMemcpy(*dst, srcp, bytes, hipMemcpyHostToDevice) : hipSuccess;
    };
    hipError_t e = hipSuccess;
    if (e == hipSuccess) e = upload((void**)&P->d_meta, meta.data(), meta.size());
    if (e == hipSuccess) e = upload((void**)&P->d_jobs, jobs.data(), jobs.size() * sizeof(GJob));
    P->thin_total = (int)tmap.size();
    if (e == hipSuccess) e = upload((void**)&P->d_tmap, tmap.data(), tmap.size() * sizeof(int32_t));
    if (e == hipSuccess) e = upload((void**)&P->d_tbmap, tbmap.data(), tbmap.size() * sizeof(int32_t));
    if (e == hipSuccess) e = upload((void**)&P->d_bmap, bmap.data(), bmap.size() * sizeof(int32_t));
    if (e == hipSuccess) e = upload((void**)&P->d_dwjobs, dwj.data(), dwj.size() * sizeof(DWJob));
    if (e == hipSuccess) e = upload((void**)&P->d_segs, segs.data(), segs.size() * sizeof(PackSeg));
    if (e == hipSuccess) e = upload((void**)&P->d_ptiles, ptiles.data(), ptiles.size() * sizeof(int2));
    if (e == hipSuccess) e = set_max_lds(std::min(LDS_LIMIT, P->lds_fwd + PERM_LDS_MAX), std::min(LDS_LIMIT, P->lds_bwd + PERM_LDS_MAX));
    if (e != hipSuccess) {
        hint_plan_destroy(P);
        return fail("hint_plan_create: device setup failed: %s", hipGetErrorString(e));
    }
    *out = P;
    return 0;
}

extern "C" {

int hint_abi_version(void) { return HINT_AMD_ABI_VERSION; }
const char* hint_last_error(void) { return g_err.c_str(); }

int hint_plan_create(const hint_node_desc* nodes, int32_t n_nodes, int32_t d, int32_t dc, float clamp,
                     hint_plan** out) {
    if (!nodes || n_nodes <= 0 || d <= 0 || dc < 0 || !out) return fail("hint_plan_create: bad arguments");
    *out = nullptr;
    if (d > 4 * NTHREADS / ROWS)      // TilePrefetch of the backward kernel holds 4 floats per thread
        return fail("hint_plan_create: d = %d lanes exceeds the supported maximum %d", d, 4 * NTHREADS / ROWS);
    // ---- validate the tree: lane ranges inside [0,d), same-depth nodes disjoint ----
    for (int i = 0; i < n_nodes; ++i) {
        const hint_node_desc& n = nodes[i];
        // (hint.py:41 always splits at D/2; the conditional-lane couplings use other splits, e.g. k = 0)
        if (n.D < 1 || n.k < 0 || n.r < 1 || n.r != n.D - n.k || n.off < 0 || n.off + n.D > d || n.h < 1 || n.depth < 0)
            return fail("hint_plan_create: node %d is malformed (off=%d D=%d k=%d r=%d h=%d depth=%d)", i,
                        n.off, n.D, n.k, n.r, n.h, n.depth);
        for (int t = 0; t < 12; ++t)
            if (n.p_off[t] < 0) return fail("hint_plan_create: negative parameter offset");
    }
    for (int i = 0; i < n_nodes; ++i)
        for (int j = i + 1; j < n_nodes; ++j)
            if (nodes[i].depth == nodes[j].depth && nodes[i].off < nodes[j].off + nodes[j].D &&
                nodes[j].off < nodes[i].off + nodes[i].D)
                return fail("hint_plan_create: nodes %d and %d of depth %d overlap", i, j, nodes[i].depth);
    // first choice: full K-split and large groups; fall back to fewer slabs, then smaller groups
    for (int cap_scale = 1; cap_scale <= 16; cap_scale *= 2)
        for (int max_slabs = MAX_SLABS; max_slabs >= 1; max_slabs /= 2)
            for (int use_a3 = 1; use_a3 >= 0; --use_a3) {
            bool retry = false;
            const int st = build_plan(nodes, n_nodes, d, dc, clamp, max_slabs, cap_scale, use_a3 != 0, out, &retry);
            if (st == 0 || !retry) return st;
        }
    return fail("hint_plan_create: could not fit the block into LDS");
}

int hint_plan_check(const hint_node_desc* nodes, int32_t n_nodes, int32_t d, int32_t dc, float clamp, int64_t* stats) {
    hint_plan* P = nullptr;
    g_host_only = true;
    const int st = hint_plan_create(nodes, n_nodes, d, dc, clamp, &P);
    g_host_only = false;
    if (st != 0) return st;
    if (stats) {
        stats[0] = P->n_groups; stats[1] = P->n_levels; stats[2] = P->WT; stats[3] = P->thin_total;
        stats[4] = P->lds_fwd; stats[5] = P->lds_bwd; stats[6] = P->jmax; stats[7] = P->n_dwjobs;
        stats[8] = P->param_floats; stats[9] = P->packed_floats; stats[10] = P->split_o3; stats[11] = P->max_aw;
    }
    delete P;                   // (host-only plans own no device memory)
    return 0;
}

void hint_plan_destroy(hint_plan* P) {
    if (!P) return;
    (void)hipFree(P->d_meta);
    (void)hipFree(P->d_jobs);
    (void)hipFree(P->d_bmap);
    (void)hipFree(P->d_tmap);
    (void)hipFree(P->d_tbmap);
    (void)hipFree(P->d_dwjobs);
    (void)hipFree(P->d_segs);
    (void)hipFree(P->d_ptiles);
    delete P;
}

int64_t hint_plan_param_floats(const hint_plan* P) { return P ? P->param_floats : -1; }
// +3 KiB of slack: the weight prefetch of the GEMM stages never reads past a job's last
// k-block, but keeping a margin makes that robust against future tuning
int64_t hint_plan_packed_floats(const hint_plan* P) { return P ? P->packed_floats + P->n_bias + 3 * 256 : -1; }

static inline int rows_padded(int B) { return (B + ROWS - 1) / ROWS * ROWS; }

// Tape layout (floats): [lane tiles: L x B x d][s: L x B x d][pad to 4][a1: Bp x WT + slack][a2: same]
static inline int64_t tape_act_off(const hint_plan* P, int B) {
    return (2 * (int64_t)P->n_levels * B * P->d + 3) / 4 * 4;
}
static inline int64_t tape_act_stride(const hint_plan* P, int B) {
    return (int64_t)rows_padded(B) * P->WT + WS_SLACK;
}

int64_t hint_plan_tape_floats(const hint_plan* P, int32_t B) {
    if (!P || B < 0) return -1;
    // per level: the lane tile as the level saw it (the last slice: the block's permuted input) and
    // the s values of the level's couplings, both [B, d]; then both hidden activations of every
    // (node, net), [Bp, WT] each (the operand a1 of part B's dW2 = g2^T a1 lives here as well)
    return tape_act_off(P, B) + 2 * tape_act_stride(P, B);
}

size_t hint_plan_workspace_bytes(const hint_plan* P, int32_t B) {
    if (!P || B <= 0) return 0;
    const size_t Bp = rows_padded(B);
    // [g2 rows][slack][thin-gradient slabs, one per row tile][one 64-float dump per row tile: where the
    // backward kernel's branch-free outer-product stores put the elements that fall outside a matrix]
    const size_t floats = (Bp * (size_t)P->WT + WS_SLACK) + (Bp / ROWS) * (size_t)P->thin_total + (Bp / ROWS) * (size_t)64 + WS_SLACK;
    return floats * sizeof(float);
}

int32_t hint_plan_lds_bytes(const hint_plan* P, int32_t backward) {
    return P ? (backward ? P->lds_bwd : P->lds_fwd) : -1;
}

// LDS bytes of the launch: the plan's, plus the chain's permutation matrices when they fit behind it
static int lds_with_perms(const hint_plan* P, int lds_plan, int n_blocks, bool any_perm, KArgs* a) {
    const long extra = (long)n_blocks * P->d * P->d * (long)sizeof(float);
    a->perm_lds = 0;
    if (!any_perm || extra > PERM_LDS_MAX || lds_plan + extra > LDS_LIMIT) return lds_plan;
    a->perm_lds = lds_plan / (int)sizeof(float);
    return lds_plan + (int)extra;
}
static KArgs make_args(const hint_plan* P, int B) {
    KArgs a{};
    a.meta = P->d_meta; a.jobs = P->d_jobs; a.bmap = P->d_tbmap; a.thin_total = P->thin_total;
    a.meta_bytes = P->meta_bytes; a.vmap_off = P->vmap_off; a.ents_off = P->ents_off; a.jmax = P->jmax;
    std::memcpy(a.first, P->first, sizeof a.first);
    a.bmax = P->bmax; a.bias_off = P->packed_floats; a.act_stride = tape_act_stride(P, B);
    a.s3 = P->s3; a.sv = P->sv;
    a.n_groups = P->n_groups; a.n_levels = P->n_levels; a.d = P->d; a.dc = P->dc;
    a.xld = P->xld; a.cld = P->cld; a.ald = P->ald; a.vld = P->vld; a.sld = P->sld;
    a.WT = P->WT;
    a.alpha = P->alpha; a.B = B; a.split_o3 = P->split_o3; a.max_aw = P->max_aw;
    return a;
}

int hint_block_pack(const hint_plan* P, const float* params, float* packed, void* stream) {
    if (!P || !params || !packed) return fail("hint_block_pack: null argument");
    HIP_TRY(launch_pack(P->d_segs, P->d_ptiles, P->n_ptiles, P->d_bmap, P->n_bias, (long)P->packed_floats, params,
                        packed, (hipStream_t)stream));
    return 0;
}

struct hint_pack_group {
    PackItem* d_items = nullptr;
    int n = 0, grid = 0;
};

int hint_pack_group_create(const hint_plan* const* plans, const float* const* params, float* const* packed,
                           int32_t n, hint_pack_group** out) {
    if (!plans || !params || !packed || n <= 0 || !out) return fail("hint_pack_group_create: bad arguments");
    std::vector<PackItem> items(n);
    int grid = 0;
    for (int i = 0; i < n; ++i) {
        const hint_plan* P = plans[i];
        if (!P || !params[i] || !packed[i]) return fail("hint_pack_group_create: null entry %d", i);
        PackItem& q = items[i];
        q.segs = P->d_segs; q.ptiles = P->d_ptiles; q.bmap = P->d_bmap; q.params = params[i]; q.packed = packed[i];
        q.bias_off = P->packed_floats; q.n_tiles = P->n_ptiles; q.n_bias = P->n_bias; q.grid_begin = grid; q.pad = 0;
        grid += P->n_ptiles + (P->n_bias + 255) / 256;
    }
    hint_pack_group* G = new hint_pack_group();
    G->n = n; G->grid = grid;
    hipError_t e = hipMalloc((void**)&G->d_items, items.size() * sizeof(PackItem));
    if (e == hipSuccess) e = hipMemcpy(G->d_items, items.data(), items.size() * sizeof(PackItem), hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(G->d_items); delete G; return fail("hint_pack_group_create: %s", hipGetErrorString(e)); }
    *out = G;
    return 0;
}

int hint_pack_group_run(const hint_pack_group* G, void* stream) {
    return hint_pack_group_run_ex(G, nullptr, 0, nullptr, nullptr, stream);
}

int hint_pack_group_run_ex(const hint_pack_group* G, float* zero_buf, int32_t zero_floats, uint64_t* rng_state,
                           float* opt_state, void* stream) {
    if (!G) return fail("hint_pack_group_run: null group");
    if (zero_floats < 0 || (zero_floats > 0 && !zero_buf)) return fail("hint_pack_group_run_ex: bad zero buffer");
    if (opt_state && !rng_state) return fail("hint_pack_group_run_ex: opt_state needs the step counter of rng_state");
    HIP_TRY(launch_pack_many(G->d_items, G->n, G->grid, zero_buf, zero_floats, (unsigned long long*)rng_state,
                             opt_state, (hipStream_t)stream));
    return 0;
}

void hint_pack_group_destroy(hint_pack_group* G) {
    if (!G) return;
    (void)hipFree(G->d_items);
    delete G;
}

static void split_workspace(const hint_plan* P, int B, void* workspace, ChainBlock* b) {
    const size_t Bp = rows_padded(B);
    b->wsG2 = (float*)workspace;
    b->wsT = b->wsG2 + Bp * P->WT + WS_SLACK;       // [row tile][thin_total] partial thin gradients
}

// the hidden activations live inside the tape (the training forward writes them)
static void bind_tape(const hint_plan* P, int B, float* tape, ChainBlock* b) {
    b->tape = tape;
    b->wsA1 = tape ? tape + tape_act_off(P, B) : nullptr;
}

// part A (row-parallel) + part B (weight gradients) of the backward pass of one block or a chain
static int run_backward(const hint_plan* P, const ChainBlock& one, const ChainBlock* chain, int n_chain,
                        const float* x, const float* c, const float* g_z, const float* g_J, float* g_x, float* g_c,
                        float gz_scale, float gJ_const, int B, hipStream_t s, bool any_perm) {
    const size_t Bp = rows_padded(B);
    const int ntiles = (B + ROWS - 1) / ROWS;
    const int grid = std::min(ntiles, P->num_cu * 8);
    const int stages = g_bwd_stages;
    if (stages & 1) {
        // (the permutation matrices stay in global memory here: the LDS table was measured +5 us in this
        // kernel, which has no register to spare)
        (void)any_perm;
        HIP_TRY(launch_bwd(make_args(P, B), P->lds_bwd, grid, one, chain, n_chain, x, c, g_z, g_J, g_x, g_c,
                           gz_scale, gJ_const, s));
    }
    if (!(stages & 2)) return 0;
    // batch split of the dW2 GEMMs: a multiple of 8 splits (one XCD each), enough workgroups
    // to cover the chip, every workgroup reducing at least 128 rows
    int splits = 8;
    while ((long)splits * P->n_dwjobs * n_chain < (long)P->num_cu && (long)Bp / (splits * 2) >= 128) splits *= 2;
    int rows_per_wg = (int)(((long)Bp + splits - 1) / splits);
    rows_per_wg = (rows_per_wg + 15) / 16 * 16;
    if ((long)rows_per_wg * (splits - 1) >= (long)Bp)   // tiny batches: fewer, non-empty splits
        splits = (int)((Bp + rows_per_wg - 1) / rows_per_wg);
    HIP_TRY(launch_dw(P->d_dwjobs, P->n_dwjobs, splits, one, chain, n_chain, P->WT, (int)Bp, rows_per_wg, P->d_tmap,
                      P->thin_total, ntiles, s));
    return 0;
}

static int apply(const hint_plan* P, bool rev, const float* params, const float* packed, const float* x,
                 const float* c, float* z, float* J, float* tape, const float* perm, const float* J_in,
                 float* loss_acc, int32_t B, void* stream) {
    const char* what = rev ? "inverse" : "forward";
    if (!P || !params || !packed || !x || !z || !J) return fail("hint_block_%s: null argument", what);
    if (P->dc > 0 && !c) return fail("hint_block_%s: plan has dc=%d but c is NULL", what, P->dc);
    if (B < 0) return fail("negative batch");
    if (B == 0) return 0;
    const int ntiles = (B + ROWS - 1) / ROWS;
    const int grid = std::min(ntiles, P->num_cu * 8);
    ChainBlock one{};
    one.params = params; one.packed = packed; one.perm = perm;
    bind_tape(P, B, rev ? nullptr : tape, &one);
    KArgs a = make_args(P, B);
    const int lds = lds_with_perms(P, P->lds_fwd, 1, perm != nullptr, &a);
    HIP_TRY(launch_apply(rev, a, lds, grid, one, nullptr, 1, x, c, z, J, J_in, loss_acc, 0.f,
                         nullptr, nullptr, (hipStream_t)stream));
    return 0;
}

int hint_block_forward(const hint_plan* P, const float* params, const float* packed, const float* x,
                       const float* c, float* z, float* J, float* tape, int32_t B, void* stream) {
    return apply(P, false, params, packed, x, c, z, J, tape, nullptr, nullptr, nullptr, B, stream);
}

int hint_block_forward_ex(const hint_plan* P, const float* params, const float* packed, const float* x,
                          const float* c, float* z, float* J, float* tape, const float* perm,
                          const float* J_in, float* loss_acc, int32_t B, void* stream) {
    return apply(P, false, params, packed, x, c, z, J, tape, perm, J_in, loss_acc, B, stream);
}

int hint_block_inverse(const hint_plan* P, const float* params, const float* packed, const float* z,
                       const float* c, float* x, float* J, int32_t B, void* stream) {
    return apply(P, true, params, packed, z, c, x, J, nullptr, nullptr, nullptr, nullptr, B, stream);
}

int hint_block_inverse_ex(const hint_plan* P, const float* params, const float* packed, const float* z,
                          const float* c, float* x, float* J, const float* perm, const float* J_in,
                          int32_t B, void* stream) {
    return apply(P, true, params, packed, z, c, x, J, nullptr, perm, J_in, nullptr, B, stream);
}

int hint_block_backward(const hint_plan* P, const float* params, const float* packed, const float* x,
                        const float* tape, const float* c, const float* g_z, const float* g_J, float* g_x,
                        float* g_c, float* g_params, int32_t accumulate, void* workspace,
                        size_t workspace_bytes, int32_t B, void* stream) {
    return hint_block_backward_ex(P, params, packed, x, tape, c, g_z, g_J, g_x, g_c, g_params, accumulate,
                                  workspace, workspace_bytes, nullptr, 1.0f, 0.0f, B, stream);
}

int hint_block_backward_ex(const hint_plan* P, const float* params, const float* packed, const float* x,
                           const float* tape, const float* c, const float* g_z, const float* g_J, float* g_x,
                           float* g_c, float* g_params, int32_t accumulate, void* workspace,
                           size_t workspace_bytes, const float* perm, float gz_scale, float gJ_const,
                           int32_t B, void* stream) {
    if (!P || !params || !packed || (!x && !perm) || !g_x || !g_params) return fail("hint_block_backward: null argument");
    if (perm && !tape) return fail("hint_block_backward_ex: a fused permutation needs the tape of hint_block_forward_ex");
    if (P->dc > 0 && !c) return fail("hint_block_backward: plan has dc=%d but c is NULL", P->dc);
    if (!tape && B > 0) return fail("hint_block_backward: tape is NULL (the backward pass reads the forward's lane tiles and s values from it)");
    if (B < 0) return fail("negative batch");
    hipStream_t s = (hipStream_t)stream;
    if (!accumulate) HIP_TRY(launch_zero(g_params, (long)P->param_floats, P->num_cu, s));
    if (B == 0) return 0;
    if (!workspace || workspace_bytes < hint_plan_workspace_bytes(P, B))
        return fail("hint_block_backward: workspace too small (%zu < %zu)", workspace_bytes,
                    hint_plan_workspace_bytes(P, B));
    if (((uintptr_t)workspace & 15) != 0) return fail("hint_block_backward: workspace must be 16-byte aligned");
    ChainBlock one{};
    one.params = params; one.packed = packed; one.perm = perm;
    bind_tape(P, B, const_cast<float*>(tape), &one);
    one.gparams = g_params;
    split_workspace(P, B, workspace, &one);
    return run_backward(P, one, nullptr, 1, x, c, g_z, g_J, g_x, g_c, gz_scale, gJ_const, B, s, one.perm != nullptr);
}

// ---------------------------------------------------------------------------------------
// chained launches: the blocks of a flow (same plan, own parameters) in one kernel each for the
// forward pass, backward part A and backward part B
// ---------------------------------------------------------------------------------------
struct hint_chain {
    const hint_plan* plan = nullptr;
    int n = 0, B = 0;
    bool committed = false;
    std::vector<ChainBlock> host;
    std::vector<char> set;
    ChainBlock* d_table = nullptr;
};

static bool chain_any_perm(const hint_chain* C) {
    for (const ChainBlock& b : C->host) if (b.perm != nullptr) return true;
    return false;
}

int hint_chain_create(const hint_plan* P, int32_t n_blocks, int32_t B, hint_chain** out) {
    if (!P || !out) return fail("hint_chain_create: null argument");
    if (n_blocks < 1 || B < 1) return fail("hint_chain_create: n_blocks and B must be >= 1");
    hint_chain* C = new hint_chain();
    C->plan = P; C->n = n_blocks; C->B = B;
    C->host.assign(n_blocks, ChainBlock{});
    C->set.assign(n_blocks, 0);
    if (hipMalloc((void**)&C->d_table, sizeof(ChainBlock) * (size_t)n_blocks) != hipSuccess) {
        delete C;
        return fail("hint_chain_create: hipMalloc failed");
    }
    *out = C;
    return 0;
}

int hint_chain_set_block(hint_chain* C, int32_t i, const float* params, const float* packed, const float* perm,
                         float* tape, void* workspace, size_t workspace_bytes, float* g_params) {
    if (!C || !params || !packed) return fail("hint_chain_set_block: null argument");
    if (i < 0 || i >= C->n) return fail("hint_chain_set_block: block %d out of range (chain has %d)", i, C->n);
    const hint_plan* P = C->plan;
    if (!tape && workspace)
        return fail("hint_chain_set_block: a trainable chain block needs a tape");
    ChainBlock b{};
    b.params = params; b.packed = packed; b.perm = perm; b.gparams = g_params;
    bind_tape(P, C->B, tape, &b);
    if (workspace) {
        if (!g_params) return fail("hint_chain_set_block: workspace without g_params");
        if (workspace_bytes < hint_plan_workspace_bytes(P, C->B))
            return fail("hint_chain_set_block: workspace too small (%zu < %zu)", workspace_bytes,
                        hint_plan_workspace_bytes(P, C->B));
        if (((uintptr_t)workspace & 15) != 0) return fail("hint_chain_set_block: workspace must be 16-byte aligned");
        split_workspace(P, C->B, workspace, &b);
    }
    C->host[i] = b;
    C->set[i] = 1;
    C->committed = false;
    return 0;
}

int hint_chain_commit(hint_chain* C) {
    if (!C) return fail("hint_chain_commit: null argument");
    for (int i = 0; i < C->n; ++i)
        if (!C->set[i]) return fail("hint_chain_commit: block %d was never set", i);
    HIP_TRY(hipMemcpy(C->d_table, C->host.data(), sizeof(ChainBlock) * (size_t)C->n, hipMemcpyHostToDevice));
    C->committed = true;
    return 0;
}

int hint_chain_forward(const hint_chain* C, const float* x, const float* c, float* z, float* J, const float* J_in,
                       float* loss_acc, void* stream) {
    return hint_chain_forward_noisy(C, x, c, z, J, J_in, loss_acc, 0.f, nullptr, nullptr, stream);
}

int hint_chain_forward_noisy(const hint_chain* C, const float* x, const float* c, float* z, float* J,
                             const float* J_in, float* loss_acc, float noise, const uint64_t* rng_state,
                             float* x_noisy, void* stream) {
    if (!C || !x || !z || !J) return fail("hint_chain_forward: null argument");
    if (!C->committed) return fail("hint_chain_forward: hint_chain_commit() has not been called");
    const hint_plan* P = C->plan;
    if (P->dc > 0 && !c) return fail("hint_chain_forward: plan has dc=%d but c is NULL", P->dc);
    const int ntiles = (C->B + ROWS - 1) / ROWS;
    const int grid = std::min(ntiles, P->num_cu * 8);
    KArgs a = make_args(P, C->B);
    const int lds = lds_with_perms(P, P->lds_fwd, C->n, chain_any_perm(C), &a);
    HIP_TRY(launch_apply(false, a, lds, grid, C->host[0], C->d_table, C->n, x, c, z, J,
                         J_in, loss_acc, noise, (const unsigned long long*)rng_state, x_noisy, (hipStream_t)stream));
    return 0;
}

int hint_chain_backward(const hint_chain* C, const float* x, const float* c, const float* g_z, const float* g_J,
                        float* g_x, float* g_c, float gz_scale, float gJ_const, int32_t accumulate, void* stream) {
    if (!C || !g_z || !g_x) return fail("hint_chain_backward: null argument");
    if (!C->committed) return fail("hint_chain_backward: hint_chain_commit() has not been called");
    const hint_plan* P = C->plan;
    if (P->dc > 0 && !c) return fail("hint_chain_backward: plan has dc=%d but c is NULL", P->dc);
    if (!x && !C->host[0].perm) return fail("hint_chain_backward: x is NULL but the first block has no fused permutation");
    for (int i = 0; i < C->n; ++i)
        if (!C->host[i].wsG2 || !C->host[i].wsA1 || !C->host[i].gparams)
            return fail("hint_chain_backward: block %d was set without workspace / g_params", i);
    hipStream_t s = (hipStream_t)stream;
    if (!accumulate)
        for (int i = 0; i < C->n; ++i) HIP_TRY(launch_zero(C->host[i].gparams, (long)P->param_floats, P->num_cu, s));
    return run_backward(P, C->host[0], C->d_table, C->n, x, c, g_z, g_J, g_x, g_c, gz_scale, gJ_const, C->B, s, chain_any_perm(C));
}

void hint_chain_destroy(hint_chain* C) {
    if (!C) return;
    (void)hipFree(C->d_table);
    delete C;
}

void hint_debug_set_backward_stages(int32_t mask) { g_bwd_stages = mask & 3; }

int hint_debug_set_stamp_buffer(void* device_buffer) {
    HIP_TRY(set_stamp_buffer((unsigned long long*)device_buffer));
    return 0;
}

static int adam_num_cu() {
    static int num_cu = 0;
    if (num_cu == 0) {
        int dev = 0; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) num_cu = prop.multiProcessorCount;
        if (num_cu <= 0) num_cu = 256;
    }
    return num_cu;
}

int hint_adam_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, int32_t step,
                   float lr, float beta1, float beta2, float eps, float weight_decay, float grad_scale,
                   float grad_clamp, int32_t zero_grads, void* stream) {
    if (!params || !grads || !exp_avg || !exp_avg_sq) return fail("hint_adam_step: null argument");
    if (n < 0 || step < 1) return fail("hint_adam_step: n must be >= 0 and step >= 1");
    if (n == 0) return 0;
    if ((((uintptr_t)params | (uintptr_t)grads | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) != 0)
        return fail("hint_adam_step: buffers must be 16-byte aligned");
    // bias corrections in double like torch.optim.Adam's python scalars
    const double bc1 = 1.0 - std::pow((double)beta1, (double)step);
    const double bc2 = 1.0 - std::pow((double)beta2, (double)step);
    HIP_TRY(launch_adam(params, grads, exp_avg, exp_avg_sq, (long)n, (float)((double)lr / bc1), beta1, beta2,
                        (float)(1.0 / std::sqrt(bc2)), eps, weight_decay, grad_scale,
                        grad_clamp > 0.f ? grad_clamp : 3.0e38f, zero_grads ? 1 : 0, adam_num_cu(), nullptr,
                        (hipStream_t)stream));
    return 0;
}

int hint_adam_step_dev(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                       const float* opt_state, float beta1, float beta2, float eps, float weight_decay,
                       float grad_scale, float grad_clamp, int32_t zero_grads, void* stream) {
    if (!params || !grads || !exp_avg || !exp_avg_sq || !opt_state) return fail("hint_adam_step_dev: null argument");
    if (n < 0) return fail("hint_adam_step_dev: n must be >= 0");
    if (n == 0) return 0;
    if ((((uintptr_t)params | (uintptr_t)grads | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) != 0)
        return fail("hint_adam_step_dev: buffers must be 16-byte aligned");
    HIP_TRY(launch_adam(params, grads, exp_avg, exp_avg_sq, (long)n, 0.f, beta1, beta2, 0.f, eps, weight_decay,
                        grad_scale, grad_clamp > 0.f ? grad_clamp : 3.0e38f, zero_grads ? 1 : 0, adam_num_cu(),
                        opt_state, (hipStream_t)stream));
    return 0;
}

}  // extern "C"
