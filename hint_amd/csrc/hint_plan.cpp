// The planner (host side of the C ABI, include/hint_amd.h): turns the node list of one coupling tree (the
// structure /root/reference/hint.py:25-54 builds recursively) into a static schedule in device
// memory - groups of same-depth nodes, their units (one subnet of one node each), the split of every
// group's fragment tiles over the wavefronts, the packed-weight layout, the weight-gradient jobs.
// The launches live in hint_abi.cpp / hint_chain.cpp / hint_invgrad.cpp.
#include "hint_host.hpp"

using namespace hint;

#ifndef HINT_THIN_MFMA_MIN
#define HINT_THIN_MFMA_MIN 8
#endif
static constexpr int THIN_MFMA_MIN = HINT_THIN_MFMA_MIN;          // thin layers with more inputs than this use the matrix pipe (fragment tiles of W1 / W3^T)
static constexpr int THIN_LDS_MAX = 24 * 1024;   // a block's thin-layer vectors are staged in LDS up to this size
static constexpr int MAX_TAIL = 8;               // hint_rows.hpp: tail accumulators
static thread_local bool g_host_only = false;    // hint_plan_check: build and verify the plan, touch no device

// A row of a group's GEMM phase: up to NTT adjacent fragment tiles [tb, tb + ntt) of one unit
struct Row { int unit, tb, ntt, slab3, slabv; long cost; };

// The rows of a group: per unit at least ceil(NT / NTT) of them; more (narrower ones) while the group has fewer rows
// than wavefronts.
static constexpr int GEN_NTT = 4;       // tiles per row of the general kernels (hint_fwd.hip / hint_bwd.hip are compiled with HINT_NTT 4); NTT (3): the wave-local ones
static std::vector<Row> split_rows(const std::vector<Unit>& units, const Group& g, int nw, int ntt_max) {
    std::vector<int> nrows(g.unit_end - g.unit_begin);
    int total = 0;
    for (int ui = g.unit_begin; ui < g.unit_end; ++ui) { nrows[ui - g.unit_begin] = cdiv(units[ui].NT, ntt_max); total += nrows[ui - g.unit_begin]; }
    while (total < nw) {
        int best = -1; double bw = 0;
        for (int ui = g.unit_begin; ui < g.unit_end; ++ui) {       // split where the rows are widest
            const int nr = nrows[ui - g.unit_begin];
            if (nr >= units[ui].NT) continue;
            const double wdt = (double)units[ui].NT / nr * units[ui].NT;
            if (wdt > bw) { bw = wdt; best = ui; }
        }
        if (best < 0) break;
        ++nrows[best - g.unit_begin]; ++total;
    }
    std::vector<Row> rows;
    for (int ui = g.unit_begin; ui < g.unit_end; ++ui) {
        const Unit& u = units[ui];
        const int nr = nrows[ui - g.unit_begin];
        int tb = 0;
        for (int ri = 0; ri < nr; ++ri) {
            const int ntt = (u.NT - tb + (nr - ri) - 1) / (nr - ri);
            // matrix-pipe time of the row (main + tail steps) plus what its bookkeeping costs in the same unit
            const long cost = (long)u.NT * ntt * 4 + std::max(u.RT, u.KB1) * ntt * 4 + 24;
            rows.push_back(Row{ui, tb, ntt, 0, 0, cost});
            tb += ntt;
        }
    }
    return rows;
}

// Deal a group's rows to the wavefronts: longest first, to the wavefront whose SIMD (wavefronts w and w + 4 share one:
// its matrix pipe and its issue slots) is least loaded; at most unit_waves wavefronts share the rows of one unit (each
// of them keeps an LDS slab for it).  Returns the row indices of every wavefront in (unit, tile) order.
static std::vector<std::vector<int>> deal_rows(const std::vector<Row>& rows, int nw, int unit_waves) {
    std::vector<std::vector<int>> wave_rows(nw);
    std::vector<int> idx(rows.size());
    for (size_t i = 0; i < idx.size(); ++i) idx[i] = (int)i;
    std::stable_sort(idx.begin(), idx.end(), [&](int x, int y) { return rows[x].cost > rows[y].cost; });
    std::vector<long> wload(nw, 0), sload(4, 0);
    auto holds = [&](int w, int unit) {
        for (int r : wave_rows[w]) if (rows[r].unit == unit) return true;
        return false;
    };
    for (int i : idx) {
        int best = 0, holders = 0;
        long best_s = -1, best_w = -1;
        for (int w = 0; w < nw; ++w) holders += holds(w, rows[i].unit) ? 1 : 0;
        for (int w = 0; w < nw; ++w) {
            if (!holds(w, rows[i].unit) && holders >= unit_waves) continue;           // (no further slab for this unit)
            // (the two wavefronts of a SIMD interleave: what one wavefront runs back to back counts as well)
            const long sl = sload[w & 3] + wload[w] + 2 * rows[i].cost, wl = wload[w] + rows[i].cost;
            if (best_s < 0 || sl < best_s || (sl == best_s && wl < best_w)) { best = w; best_s = sl; best_w = wl; }
        }
        wave_rows[best].push_back(i);
        wload[best] += rows[i].cost; sload[best & 3] += rows[i].cost;
    }
    for (int w = 0; w < nw; ++w) std::sort(wave_rows[w].begin(), wave_rows[w].end());   // unit order, then tile order
    return wave_rows;
}

// LDS slabs for the K-split partials of the tail products: one per (unit, wavefront that has rows of it), a unit's slabs
// adjacent (forward: 16 rows x pad4(r); backward: 16 rows x pad4(cin)); *off3 / *offv: floats used so far
static void assign_slabs(std::vector<Unit>& units, const Group& g, std::vector<Row>& rows,
                         const std::vector<std::vector<int>>& wave_rows, int* off3, int* offv) {
    for (int ui = g.unit_begin; ui < g.unit_end; ++ui) {
        Unit& u = units[ui];
        u.sl_off = *off3; u.gv_off = *offv; u.sl_n = 0;
        for (const std::vector<int>& mine : wave_rows) {
            bool has = false;
            for (int ri : mine)
                if (rows[ri].unit == ui) { rows[ri].slab3 = *off3; rows[ri].slabv = *offv; has = true; }
            if (has) { *off3 += 64 * cdiv(u.r, 4); *offv += 64 * cdiv(u.cin, 4); ++u.sl_n; }
        }
    }
}

// The row records of a group in wavefront order (hint_dev.h: RowRec), both directions, and the wavefronts' ranges in
// that list (nw + 1 offsets appended to rng).  false: a unit does not fit the records' bit fields.
static bool emit_row_records(const std::vector<Unit>& units, const std::vector<Row>& rows,
                             const std::vector<std::vector<int>>& wave_rows, int row_begin, std::vector<RowRec>* recs_f,
                             std::vector<RowRec>* recs_b, std::vector<int32_t>* rng, std::vector<int>* rec_unit) {
    for (const std::vector<int>& mine : wave_rows) {
        rng->push_back((int)recs_f->size() - row_begin);
        int last_unit = -1;
        for (size_t mi = 0; mi < mine.size(); ++mi) {
            const int ri = mine[mi];
            const Row& rw = rows[ri];
            const int ulast = (mi + 1 == mine.size() || rows[mine[mi + 1]].unit != rw.unit) ? 1 : 0;     // the wavefront's last row of the unit
            rec_unit->push_back(rw.unit);
            const Unit& u = units[rw.unit];
            if (u.NT > 255 || u.cin > 255 || u.ku > 255 || u.r > 255 || u.tile0 > 0xffff) return false;
            const int thin = rw.unit != last_unit ? 1 : 0, first = rw.tb == 0 ? 1 : 0;     // (thin: the wavefront's first row of the unit starts its slab)
            last_unit = rw.unit;
            RowRec r{};
            r.ocol = u.wcol + 16 * rw.tb; r.bias3 = u.bias3; r.wcol = u.wcol; r.tb = rw.tb;
            r.flags = u.NT | (thin << 8) | (first << 9) | (ulast << 10);
            // forward: second layer + third layer partials (+ b3 with the unit's first row)
            r.base1 = u.f2 + rw.tb * u.NT; r.base2 = u.f3 + rw.tb;
            r.counts = u.NT | (u.RT << 8) | ((first ? u.RT : 0) << 16) | (rw.ntt << 24);
            r.aux = u.bias2 + 16 * rw.tb; r.tile = u.tile0 | (cdiv(u.r, 4) << 16); r.slab = rw.slab3;
            r.thin_w = u.w1v; r.thin_b = 0; r.thin_k = u.cin | (u.ku << 8) | (u.xoff << 16);
            recs_f->push_back(r);
            // backward: g1 + g_v partials
            r.base1 = u.b2 + rw.tb * u.NT; r.base2 = u.b1 + rw.tb;
            r.counts = u.NT | (u.KB1 << 8) | (rw.ntt << 24);
            r.aux = 0; r.tile = u.tile0 | (cdiv(u.cin, 4) << 16); r.slab = rw.slabv;
            r.thin_w = u.w3v; r.thin_b = 0; r.thin_k = u.r | (u.lcol << 16);
            recs_b->push_back(r);
        }
    }
    rng->push_back((int)recs_f->size() - row_begin);
    return true;
}

// Coupling entries of a group: one per transformed lane (where its s and t partials wait in the slabs)
static void emit_coupling_entries(const std::vector<Unit>& units, const Group& g, std::vector<Ent>* ents) {
    for (int ui = g.unit_begin; ui < g.unit_end; ui += 2) {
        const Unit& us = units[ui];
        const Unit& ut = units[ui + 1];
        for (int j = 0; j < us.r; ++j) {
            Ent e{};
            e.xcol = (int16_t)(us.xoff + us.ku + j);
            e.nquad = (int16_t)cdiv(us.r, 4);
            e.sl_ns = (int16_t)us.sl_n; e.sl_nt = (int16_t)ut.sl_n;
            e.s_off = us.sl_off + (j / 4) * 64 + (j % 4);
            e.t_off = ut.sl_off + (j / 4) * 64 + (j % 4);
            ents->push_back(e);
        }
    }
}

// Backward lane tables: per boundary (in front of group gi; slot n_groups: behind group 0) and lane, what to add from
// the group before (the g_v partials of the node whose input this lane is) and where the coupling gradients of the coming
// group's node that transforms this lane go (hint_dev.h: LaneOp).  Sets every group's lop_begin.
static std::vector<LaneOp> build_lane_ops(std::vector<Group>& groups, const std::vector<Unit>& units, int d) {
    const int n_groups = (int)groups.size();
    std::vector<LaneOp> lops((size_t)(n_groups + 1) * d);
    for (int b = 0; b <= n_groups; ++b) {
        const int cur = b < n_groups ? b : -1;                 // the group about to run (none for the last slot)
        const int prev = b < n_groups ? (b + 1 < n_groups ? b + 1 : -1) : 0;   // the group that ran just before
        for (int col = 0; col < d; ++col) {
            LaneOp op{};
            op.sc_unit = -1; op.sc_k = 0; op.cp_ls = -1; op.cp_lt = 0; op.cp_gs = 0; op.cp_gt = 0;
            if (prev >= 0)
                for (int ui = groups[prev].unit_begin; ui < groups[prev].unit_end; ui += 2) {
                    const Unit& u = units[ui];
                    if (col >= u.xoff && col < u.xoff + u.ku) { op.sc_unit = (int16_t)ui; op.sc_k = (int16_t)(col - u.xoff); }
                }
            if (cur >= 0)
                for (int ui = groups[cur].unit_begin; ui < groups[cur].unit_end; ui += 2) {
                    const Unit& us = units[ui];
                    const Unit& ut = units[ui + 1];
                    const int j = col - us.xoff - us.ku;
                    if (j >= 0 && j < us.r) {
                        op.cp_ls = (int16_t)(us.lcol + j); op.cp_lt = (int16_t)(ut.lcol + j);
                        op.cp_gs = (int16_t)(us.gcol + j); op.cp_gt = (int16_t)(ut.gcol + j);
                    }
                }
            lops[(size_t)b * d + col] = op;
        }
        // the boundary's ACTIVE lanes (something to add or a coupling gradient to form), compacted: entry k's `pad` holds the k-th
        // active lane | count << 16.  The kernels read the SLOT table built from this one further down (order: lanes that are both,
        // pairs of a coupling-only and a scatter-only lane, the rest); this per-lane table is the planner's working copy, and its
        // count is what HINT_PLAN_DUMP prints beside the slots.
        int cnt = 0;
        for (int pass = 0; pass < 2; ++pass)
            for (int col = 0; col < d; ++col) {
                const LaneOp& o = lops[(size_t)b * d + col];
                if (pass == 0 ? o.cp_ls >= 0 : (o.cp_ls < 0 && o.sc_unit >= 0)) lops[(size_t)b * d + cnt++].pad = col;
            }
        for (int k = 0; k < d; ++k) lops[(size_t)b * d + k].pad = (k < cnt ? lops[(size_t)b * d + k].pad : 0) | (cnt << 16);
        if (b < n_groups) groups[b].lop_begin = b * d;
    }
    return lops;
}

// Self-check: the record lists of every group cover every fragment tile of every unit exactly once, in both
// directions, and exactly one row per unit is its first.  Returns what is wrong, or nullptr.
static const char* check_records(const std::vector<Group>& groups, const std::vector<Unit>& units,
                                 const std::vector<RowRec>& recs_f, const std::vector<RowRec>& recs_b,
                                 const std::vector<int32_t>& rng, int nw) {
    for (const Group& g : groups) {
        const int32_t* r = rng.data() + g.rng_begin;
        const int nrows = r[nw];
        for (int dir = 0; dir < 2; ++dir) {
            const std::vector<RowRec>& rc = dir ? recs_b : recs_f;
            std::vector<int> seen(g.ntiles, 0), firsts(g.unit_end - g.unit_begin, 0);
            for (int w = 0; w < nw; ++w) {
                if (r[w] > r[w + 1] || r[w + 1] > nrows) return "record ranges";
                for (int i = r[w]; i < r[w + 1]; ++i) {
                    const RowRec& q = rc[g.row_begin + i];
                    const int tile0 = q.tile & 0xffff, ntt = (q.counts >> 24) & 0xff, NT = q.flags & 0xff;
                    const int tb = (q.ocol - q.wcol) / 16;
                    int ui = -1;
                    for (int u = g.unit_begin; u < g.unit_end; ++u) if (units[u].tile0 == tile0) ui = u;
                    if (ui < 0 || units[ui].NT != NT || ntt < 1 || ntt > GEN_NTT || tb < 0 || tb + ntt > NT) return "row record";
                    if ((q.flags >> 9) & 1) ++firsts[ui - g.unit_begin];
                    for (int j = 0; j < ntt; ++j) ++seen[tile0 + tb + j];
                }
            }
            for (int c : seen) if (c != 1) return "tiles not covered exactly once";
            for (int c : firsts) if (c != 1) return "first rows";
        }
    }
    return nullptr;
}

// The weight-gradient jobs of part B (hint_dev.h: WJob) - per unit dW2, dW3 (+ db2, db3) and, unless the backward kernel
// computes them itself (fuse_dw1), dW1 (+ db1) - and the map of real parameter elements (1: summed from part B's slabs,
// 2: from the backward kernel's, 0: padding between tensors)
static void make_wgrad_jobs(const hint_plan* P, const hint_node_desc* nodes, const std::vector<Unit>& units,
                            const std::vector<int>& unit_node, const std::vector<char>& unit_lean,
                            const std::vector<char>& unit_fused, int max_depth, std::vector<WJob>* wjobs,
                            std::vector<uint8_t>* real) {
    const int d = P->d, dc = P->dc;
    for (size_t ui = 0; ui < units.size(); ++ui) {
        const Unit& u = units[ui];
        const hint_node_desc& n = nodes[unit_node[ui]];
        const bool lean = unit_lean[ui] != 0, fused = unit_fused[ui] != 0;     // (operands rebuilt - lean or lean-wide; dW1 / db1 from the backward kernel)
        const int net = (int)(ui & 1);
        const int64_t* po = n.p_off + net * 6;
        const int level = max_depth - n.depth;
        const int64_t sizes[6] = {(int64_t)n.h * u.cin, n.h, (int64_t)n.h * n.h, n.h, (int64_t)n.r * n.h, n.r};
        for (int t = 0; t < 6; ++t)
            for (int64_t i = 0; i < sizes[t]; ++i) (*real)[(size_t)(po[t] + i)] = (fused && t < 2) ? 2 : 1;      // 2: summed from the backward kernel's slabs
        auto add_jobs = [&](int psrc, int pcol, int M, int pmaxc, int qsrc, int qcol, int N, int qmaxc, int qlevel, int ldo,
                            int64_t wofs, int64_t bofs) {
            // tiles of up to 48 x 48 outputs; the bias gradient rides with the first column group
            const int MT = cdiv(M, 16), NTn = std::max(1, cdiv(N, 16));
            for (int mt = 0; mt < MT; mt += 3)
                for (int nt = 0; nt < NTn; nt += 3) {
                    WJob j{};
                    j.psrc = psrc; j.pcol = pcol + 16 * mt; j.M = std::min(48, M - 16 * mt); j.mw = std::min(3, MT - mt);
                    j.qsrc = qsrc; j.qcol = qcol + 16 * nt; j.N = N > 0 ? std::min(48, N - 16 * nt) : 0;
                    j.nw = N > 0 ? std::min(3, NTn - nt) : 0;
                    j.qlevel = qlevel; j.ldo = ldo; j.pmax = pmaxc; j.qmax = qmaxc;
                    j.wofs = wofs + (int64_t)16 * mt * ldo + 16 * nt;
                    j.bofs = (bofs >= 0 && nt == 0) ? bofs + 16 * mt : -1;
                    j.r_w1 = (int32_t)po[HINT_W1]; j.r_b1 = (int32_t)po[HINT_B1]; j.r_w3 = (int32_t)po[HINT_W3];
                    j.r_cin = u.cin; j.r_xoff = u.xoff; j.r_r = n.r; j.r_gcol = u.gcol; j.r_h = n.h; j.r_wcol = u.wcol;
                    wjobs->push_back(j);
                }
        };
        add_jobs(lean ? WSRC_G2R : WSRC_G2, u.wcol, n.h, P->WT - 1, lean ? WSRC_A1R : WSRC_A1, u.wcol, n.h, P->WT - 1,
                 lean ? level : 0, n.h, po[HINT_W2], po[HINT_B2]);
        add_jobs(WSRC_GST, u.gcol, n.r, P->ST - 1, WSRC_A2, u.wcol, n.h, P->WT - 1, 0, n.h, po[HINT_W3], po[HINT_B3]);
        if (u.ku > 0 && !fused)
            add_jobs(WSRC_G1, u.wcol, n.h, P->WT - 1, WSRC_X, u.xoff, u.ku, d - 1, level, u.cin, po[HINT_W1], po[HINT_B1]);
        if (dc > 0)
            add_jobs(WSRC_G1, u.wcol, n.h, P->WT - 1, WSRC_C, 0, dc, dc - 1, 0, u.cin, po[HINT_W1] + u.ku, u.ku > 0 ? -1 : po[HINT_B1]);
        if (u.cin == 0)
            add_jobs(WSRC_G1, u.wcol, n.h, P->WT - 1, WSRC_A1, u.wcol, 0, P->WT - 1, 0, 1, po[HINT_W1], po[HINT_B1]);
    }
}

// tile_cap: fragment tiles per group (1 KiB of LDS each), unless one node needs more
// unit_waves: how many wavefronts may share the rows of one unit (each of them keeps a slab for it)
// returns 0, 1 (error) or 2 (the block does not fit the LDS with these two settings; *retry_smaller: smaller groups exist)
static int build_plan(const hint_node_desc* nodes, int32_t n_nodes, int32_t d, int32_t dc, float clamp, int nw,
                      int tile_cap, int unit_waves, hint_plan** out, bool* retry_smaller) {
    *retry_smaller = false;
    int max_depth = 0;
    for (int i = 0; i < n_nodes; ++i) max_depth = std::max(max_depth, nodes[i].depth);

    hint_plan* P = new hint_plan();
    P->d = d;
    P->dc = dc;
    P->n_nodes = n_nodes;
    P->n_levels = max_depth + 1;
    P->nw = nw;
    P->alpha = (float)((double)clamp * 0.636);   // hint.py:57,60 (python float product, then fp32)
    P->xld = d | 1;                              // odd strides: 16 rows hit 16 different LDS banks
    P->cld = dc > 0 ? (dc | 1) : 0;

    // ---- forward order: deepest level first (children before parents, hint.py:70-73) ----
    std::vector<int> order;
    for (int dep = max_depth; dep >= 0; --dep)
        for (int i = 0; i < n_nodes; ++i)
            if (nodes[i].depth == dep) order.push_back(i);

    // ---- subtree groups (hint_sub.hpp): from depth sub_depth down every subnet is lean (1..4 inputs, at most 4 outputs, no
    //      condition) and at most two tiles wide, every level is one group, and the level sub_depth has enough nodes to
    //      occupy half the wavefronts: those levels run one subtree per wavefront without workgroup barriers.  Not for
    //      trees the wave-local kernels take whole.  node_wave: the wavefront of a node of these levels.  (HINT_SUB=0: never) ----
    int sub_depth = max_depth + 1;
    std::vector<int> node_wave(n_nodes, -1);
    std::vector<int32_t> sub_cols_v;
    {
        bool all_lean = dc == 0;
        for (int i = 0; i < n_nodes; ++i)
            if (nodes[i].k < 1 || nodes[i].k > 4 || nodes[i].r < 1 || nodes[i].r > 4) all_lean = false;
        bool on = dc == 0 && !(all_lean && d <= 4 * WL_LV);
        on = on && knobs().sub && knobs().lean && knobs().fuse_dw1;
        for (int dep = max_depth; on && dep >= 1; --dep) {
            int cnt = 0, tiles = 0, last_off = -1;
            bool ok = true;
            for (int i = 0; i < n_nodes; ++i) {
                if (nodes[i].depth != dep) continue;
                const hint_node_desc& n = nodes[i];
                if (n.k < 1 || n.k > 4 || n.r < 1 || n.r > 4 || n.h > 16 || n.off <= last_off || n.off > 255) ok = false;
                last_off = n.off; ++cnt; tiles += 2 * cdiv(n.h, 16);
            }
            if (!ok || tiles > tile_cap || 2 * cnt < nw) break;
            sub_depth = dep;
        }
        if (sub_depth <= max_depth) {
            std::vector<int> roots;
            for (int i = 0; i < n_nodes; ++i) if (nodes[i].depth == sub_depth) roots.push_back(i);
            const int nsub = (int)roots.size();
            for (int i = 0; i < n_nodes; ++i) {
                if (nodes[i].depth < sub_depth) continue;
                for (int j = 0; j < nsub; ++j) {
                    const hint_node_desc& rt = nodes[roots[j]];
                    if (nodes[i].off >= rt.off && nodes[i].off < rt.off + rt.D) node_wave[i] = nsub >= nw ? (int)((long)j * nw / nsub) : j;
                }
                if (node_wave[i] < 0) { sub_depth = max_depth + 1; break; }
            }
            // per wavefront four lane bounds: [0], [1] = the lanes of its subtrees (what it reads of a level's tape slices: at most
            // 16 lanes, hint_sub.hpp SUB_LV); [2], [3] = the lanes it stores to the tape after a level - the subtrees' lanes and
            // the lanes up to the next wavefront's (a partition of all d lanes: a tape slice is the whole lane tile)
            std::vector<int> lo(nw, d), hi(nw, 0);
            for (int j = 0; j < nsub; ++j) {
                const int w = node_wave[roots[j]];
                lo[w] = std::min(lo[w], nodes[roots[j]].off); hi[w] = std::max(hi[w], nodes[roots[j]].off + nodes[roots[j]].D);
            }
            int prev = 0;
            for (int w = 0; w < nw && sub_depth <= max_depth; ++w) {
                const bool has = lo[w] < hi[w];
                if (has && hi[w] - lo[w] > 16) { sub_depth = max_depth + 1; break; }
                int next = d;                  // where the next wavefront with subtrees starts
                for (int v = w + 1; v < nw; ++v) if (lo[v] < hi[v]) { next = lo[v]; break; }
                sub_cols_v.push_back(has ? lo[w] : 0); sub_cols_v.push_back(has ? hi[w] : 0);
                sub_cols_v.push_back(has ? prev : 0); sub_cols_v.push_back(has ? next : 0);
                if (has) prev = next;
            }
        }
    }
    // rows of three tiles for the trees the wave-local kernels may take, of four for the general kernels
    int ntt_max = GEN_NTT;
    {
        bool narrow = dc == 0 && d <= 4 * WL_LV;
        for (int i = 0; i < n_nodes; ++i)
            if (nodes[i].k < 1 || nodes[i].k > 4 || nodes[i].r < 1 || nodes[i].r > 4) narrow = false;
        if (narrow) ntt_max = NTT;
    }
    int sub_off3 = 0, sub_offv = 0;       // the subtree groups' slabs: one area for all of them (the wavefronts are at different levels at any time)
    bool sub_closed = false;             // the first group above the subtree levels has been seen

    std::vector<Group> groups;
    std::vector<Unit> units;
    std::vector<int> unit_node;       // node index (into `nodes`) of every unit
    std::vector<RowRec> recs_f, recs_b;   // row records in (group, wavefront, unit) order
    std::vector<int> rec_unit;            // ... and the unit of each
    std::vector<ThinRec> thin_f, thin_b;  // thin records in (group, unit, tile) order
    std::vector<int> grp_slab_f, grp_slab_b;  // per group: floats of its slabs (forward / backward)
    std::vector<int> unit_f1, unit_b3;    // per unit: first fragment tile of W1 / W3^T for a wide thin layer, or -1
    std::vector<Ent> ents;
    std::vector<int32_t> rng;
    std::vector<PackSeg> segs;
    std::vector<int2> ptiles;
    std::vector<int32_t> bmap;
    std::vector<WJob> wjobs;
    int64_t pmax = 0, packed = 0;
    int wcol = 0, gcol = 0;

    // the thin layers' vectors form two contiguous blobs at the start of the packed buffer (forward: W1 and b1 of
    // every unit; backward: W3^T), each padded to whole 256-float tiles; the fragment tiles follow
    int64_t blob_f = 0, blob_b = 0;
    for (int i = 0; i < n_nodes; ++i) {
        const int NT = cdiv(nodes[i].h, 16);
        blob_f += 2 * (int64_t)NT * (std::max(4, nodes[i].k + dc) + 1) * 16;
        blob_b += 2 * (int64_t)NT * std::max(4, nodes[i].r) * 16;
    }
    const int64_t blob_f_pad = (blob_f + 255) / 256 * 256, blob_b_pad = (blob_b + 255) / 256 * 256;
    int64_t cur_f = 0, cur_b = blob_f_pad;
    packed = blob_f_pad + blob_b_pad;
    // fragment-layout segment: returns its first packed tile (offset / 256)
    auto add_seg = [&](int N, int K, int NB, int ld, int trans, int64_t src) -> int {
        PackSeg sg{};
        sg.dst = packed; sg.src = src; sg.src2 = -1; sg.N = N; sg.K = K; sg.NB = NB; sg.ld = ld; sg.trans = trans; sg.kmap = 0;
        sg.tile_begin = (int)ptiles.size();
        const int NTn = std::max(1, cdiv(N, 16));
        for (int nt = 0; nt < NTn; ++nt) ptiles.push_back(int2{(int)segs.size(), nt});
        segs.push_back(sg);
        const int first = (int)(packed / 256);
        packed += (int64_t)NTn * NB * 256;
        return first;
    };
    // vector-layout segment (thin layers) inside a blob: returns its float offset inside that blob
    auto add_vec = [&](int64_t& cur, int64_t blob0, int N, int K, int ld, int trans, int64_t src, int64_t src2) -> int {
        PackSeg sg{};
        const int Kp = std::max(4, K);      // inputs padded to four with zero vectors: the usual K <= 4 runs branch free
        sg.dst = cur; sg.src = src; sg.src2 = src2; sg.N = N; sg.K = K; sg.NB = Kp; sg.ld = ld; sg.trans = trans; sg.kmap = 2;
        sg.tile_begin = (int)ptiles.size();
        const int NTn = std::max(1, cdiv(N, 16));
        for (int nt = 0; nt < NTn; ++nt) ptiles.push_back(int2{(int)segs.size(), nt});
        segs.push_back(sg);
        const int first = (int)(cur - blob0);
        cur += (int64_t)NTn * (Kp + (src2 >= 0 ? 1 : 0)) * 16;
        return first;
    };

    // ---- groups, units, packed segments ----
    size_t pos = 0;
    while (pos < order.size()) {
        Group g{};
        g.unit_begin = (int)units.size();
        g.gcol0 = gcol;
        g.wcol0 = wcol;
        const int depth = nodes[order[pos]].depth;
        const bool sub = depth >= sub_depth;
        if (!sub && !sub_closed) {
            // what the subtree groups read from LDS: the vectors and biases of their units open the two blobs and the bias region
            sub_closed = true;
            P->n_sub = (int)groups.size();
            P->sub_pf = (int)cur_f; P->sub_pb = (int)(cur_b - blob_f_pad); P->sub_pbias = (int)bmap.size(); P->sub_bsrc = (int)blob_f_pad;
        }
        int tiles = 0;
        const size_t first_pos = pos;
        while (pos < order.size() && nodes[order[pos]].depth == depth) {
            const hint_node_desc& n = nodes[order[pos]];
            const int NT = cdiv(n.h, 16);
            if (tiles > 0 && tiles + 2 * NT > tile_cap) break;
            const int cin = n.k + dc, KB1 = std::max(1, cdiv(cin, 16)), RT = cdiv(n.r, 16);
            if (RT > MAX_TAIL || KB1 > MAX_TAIL) {
                delete P;
                return fail("hint_plan_create: a node with r=%d outputs / cin=%d inputs exceeds the kernels' limit of %d",
                            n.r, cin, 16 * MAX_TAIL);
            }
            const int64_t sizes[6] = {(int64_t)n.h * cin, n.h, (int64_t)n.h * n.h, n.h, (int64_t)n.r * n.h, n.r};
            for (int t = 0; t < 12; ++t) pmax = std::max(pmax, n.p_off[t] + sizes[t % 6]);
            for (int net = 0; net < 2; ++net) {
                const int64_t* po = n.p_off + net * 6;
                Unit u{};
                u.w1v = add_vec(cur_f, 0, n.h, cin, cin, 0, po[HINT_W1], po[HINT_B1]);       // v  -> a1 (vector ALU; b1 as vector cin)
                u.f2 = add_seg(n.h, n.h, NT, n.h, 0, po[HINT_W2]);           // a1 -> a2
                u.f3 = add_seg(n.r, n.h, NT, n.h, 0, po[HINT_W3]);           // a2 -> s | t
                u.w3v = add_vec(cur_b, blob_f_pad, n.h, n.r, n.h, 1, po[HINT_W3], -1);       // g_st -> g2 (vector ALU)
                u.b2 = add_seg(n.h, n.h, NT, n.h, 1, po[HINT_W2]);           // g2 -> g1
                u.b1 = add_seg(cin, n.h, NT, cin, 1, po[HINT_W1]);           // g1 -> g_v
                // wide thin layers (more than THIN_MFMA_MIN inputs) run on the matrix pipe from fragment tiles instead
                unit_f1.push_back(cin > THIN_MFMA_MIN ? add_seg(n.h, cin, KB1, cin, 0, po[HINT_W1]) : -1);          // v -> a1
                unit_b3.push_back(n.r > THIN_MFMA_MIN ? add_seg(n.h, n.r, RT, n.h, 1, po[HINT_W3]) : -1);          // g_st -> g2
                u.bias1 = (int)bmap.size();                                     // (made absolute below)
                for (int j = 0; j < 16 * NT; ++j) bmap.push_back(j < n.h ? (int32_t)(po[HINT_B1] + j) : -1);
                u.bias2 = (int)bmap.size();
                for (int j = 0; j < 16 * NT; ++j) bmap.push_back(j < n.h ? (int32_t)(po[HINT_B2] + j) : -1);
                u.bias3 = (int)bmap.size();
                for (int j = 0; j < 16 * RT; ++j) bmap.push_back(j < n.r ? (int32_t)(po[HINT_B3] + j) : -1);
                u.wcol = wcol; u.tile0 = tiles; u.gcol = gcol;
                u.NT = NT; u.KB1 = KB1; u.RT = RT; u.cin = cin;
                u.ku = n.k; u.r = n.r; u.xoff = n.off; u.h = n.h;
                u.lcol = sub ? gcol : gcol - g.gcol0;       // (subtree groups: one coupling-gradient buffer for all of them)
                units.push_back(u);
                unit_node.push_back(order[pos]);
                wcol += 16 * NT; gcol += pad4(n.r); tiles += NT;
            }
            ++pos;
        }
        g.unit_end = (int)units.size();
        g.ntiles = tiles;
        g.gcols = gcol - g.gcol0;
        g.level = max_depth - depth;
        g.level_first = (first_pos == 0 || nodes[order[first_pos - 1]].depth != depth) ? 1 : 0;
        g.level_last = (pos >= order.size() || nodes[order[pos]].depth != depth) ? 1 : 0;
        P->abuf_tiles = std::max(P->abuf_tiles, tiles);
        P->gld = std::max(P->gld, (sub ? gcol : g.gcols) | 1);      // (subtree groups: the columns of all of them side by side)

        std::vector<Row> rows = split_rows(units, g, nw, ntt_max);
        for (const Row& rw : rows) P->row_ntt = std::max(P->row_ntt, rw.ntt);
        int off3 = 0, offv = 0;
        g.tile_begin = (int)thin_f.size();
        for (int ui = g.unit_begin; ui < g.unit_end; ++ui) {
            Unit& u = units[ui];
            const int kpf = std::max(4, u.cin), kpb = std::max(4, u.r);
            for (int nt = 0; nt < u.NT; ++nt) {
                // (kp | (first fragment tile of the tile's k-blocks + 1) << 8: the matrix-pipe variant)
                const int f1 = unit_f1[ui], b3 = unit_b3[ui];
                thin_f.push_back(ThinRec{u.w1v + nt * (kpf + 1) * 16, u.cin | (u.ku << 8) | (u.xoff << 16), u.tile0 + nt,
                                         kpf | ((f1 >= 0 ? f1 + nt * u.KB1 + 1 : 0) << 8)});
                thin_b.push_back(ThinRec{u.w3v + nt * kpb * 16, u.r | (u.lcol << 16), u.tile0 + nt,
                                         kpb | ((b3 >= 0 ? b3 + nt * u.RT + 1 : 0) << 8)});
            }
        }
        std::vector<std::vector<int>> wave_rows;
        if (sub) {
            // every row of a node goes to the wavefront that owns the node's subtree
            wave_rows.assign(nw, std::vector<int>());
            for (size_t ri = 0; ri < rows.size(); ++ri) wave_rows[node_wave[unit_node[rows[ri].unit]]].push_back((int)ri);
            off3 = sub_off3; offv = sub_offv;
            assign_slabs(units, g, rows, wave_rows, &off3, &offv);
            sub_off3 = off3; sub_offv = offv;
            grp_slab_f.push_back(0); grp_slab_b.push_back(0);
            g.lean |= 4;                   // (bits 0, 1 are set further down)
        } else {
            wave_rows = deal_rows(rows, nw, unit_waves);
            assign_slabs(units, g, rows, wave_rows, &off3, &offv);
            grp_slab_f.push_back(off3); grp_slab_b.push_back(offv);
            P->slab_fwd = std::max(P->slab_fwd, off3);
            P->slab_bwd = std::max(P->slab_bwd, offv);
        }
        if (knobs().plan_dump) {
            std::fprintf(stderr, "[hint plan] group %d level %d: %d units, %d tiles, %d rows\n", (int)groups.size(), g.level,
                         g.unit_end - g.unit_begin, tiles, (int)rows.size());
            for (int w = 0; w < nw; ++w) {
                std::fprintf(stderr, "   wave %d:", w);
                for (int ri : wave_rows[w]) std::fprintf(stderr, " (u%d t%d+%d)", rows[ri].unit - g.unit_begin, rows[ri].tb, rows[ri].ntt);
                std::fprintf(stderr, "\n");
            }
        }
        // ---- the wavefronts' record lists, forward and backward; the thin layers' tiles are shared out evenly ----
        g.row_begin = (int)recs_f.size();
        g.rng_begin = (int)rng.size();
        if (!emit_row_records(units, rows, wave_rows, g.row_begin, &recs_f, &recs_b, &rng, &rec_unit)) {
            delete P;
            return fail("hint_plan_create: a node is too wide for the row records (h <= 4080, cin <= 255)");
        }
        if (sub) {
            // the wavefronts' ranges in the group's entry list (entries are emitted in unit order = wavefront order)
            int acc = 0;
            for (int w = 0; w < nw; ++w) {
                rng.push_back(acc);
                for (int ui = g.unit_begin; ui < g.unit_end; ui += 2)
                    if (node_wave[unit_node[ui]] == w) acc += units[ui].r;
            }
            rng.push_back(acc);
        } else {
            for (int w = 0; w <= nw; ++w) rng.push_back((int)((long)tiles * w / nw));
        }

        g.ent_begin = (int)ents.size();
        emit_coupling_entries(units, g, &ents);
        g.ent_cnt = (int)ents.size() - g.ent_begin;
        groups.push_back(g);
    }
    if (P->n_sub > 0) {
        P->sub_cols = (int)rng.size();
        rng.insert(rng.end(), sub_cols_v.begin(), sub_cols_v.end());
        P->sub_slab_f = 0; P->sub_slab_b = 0;       // (the nodes' partial sums stay in registers: no slabs in LDS; the offsets above are unused)
    }
    // the backward lane tables (one boundary per group + the tail's); every boundary's slot count rides in the ranges table (set where the slot table is built)
    std::vector<LaneOp> lops = build_lane_ops(groups, units, d);
    P->lop_cnt = (int)rng.size();
    for (size_t b = 0; b <= groups.size(); ++b) rng.push_back(lops[b * (size_t)d].pad >> 16);
    for (const Unit& u : units) P->max_h = std::max(P->max_h, (int)u.h);
    P->n_groups = (int)groups.size();
    P->n_units = (int)units.size();
    P->WT = wcol;
    P->ST = gcol;
    P->param_floats = (pmax + 3) / 4 * 4;
    P->packed_floats = packed;
    P->n_bias = (int)bmap.size();
    if (pmax >= (int64_t)1 << 31 || packed + (int64_t)bmap.size() >= (int64_t)1 << 31 || units.size() > 32000 ||
        gcol > 32000 || d > 32000) {
        delete P;
        return fail("hint_plan_create: block too large (offsets must fit 31 / 15 bits)");
    }
    for (Unit& u : units) { u.bias1 += (int)packed; u.bias2 += (int)packed; u.bias3 += (int)packed; }   // the bias region follows the weight tiles
    // lean groups: thin layers narrow enough that part B rebuilds a1 and g2 instead of reading them - every unit of the
    // group has 1..4 inputs, at most 4 outputs and no condition (HINT_LEAN=0: never).  P->lean: all groups are, and the
    // a1 / g2 arrays do not exist at all.
    const bool lean_on = dc == 0 && knobs().lean;
    std::vector<char> unit_lean(units.size(), 0);
    P->lean = 1;
    for (Group& g : groups) {
        const int sub_bit = g.lean & 4;
        g.lean = lean_on ? 1 : 0;
        for (int ui = g.unit_begin; ui < g.unit_end; ++ui) {
            const hint_node_desc& n = nodes[unit_node[ui]];
            if (units[ui].cin < 1 || units[ui].cin > 4 || n.r < 1 || n.r > 4) g.lean = 0;
        }
        // lean-wide (round 6): not lean, yet thin enough that part B rebuilds a1 / g2 per 16-row step (ceil(cin / 4) + ceil(r / 4)
        // K = 4 MFMAs per tile) instead of reading them: the forward stores no a1, the backward no g2 (HINT_LEANW=0: never)
        bool leanw = !g.lean && lean_on && knobs().leanw;
        for (int ui = g.unit_begin; ui < g.unit_end; ++ui) {
            const hint_node_desc& n = nodes[unit_node[ui]];
            if (units[ui].cin < 1 || units[ui].cin > knobs().leanw_max || n.r < 1 || n.r > knobs().leanw_max) leanw = false;
        }
        for (int ui = g.unit_begin; ui < g.unit_end; ++ui) unit_lean[ui] = (char)(g.lean ? 1 : leanw ? 2 : 0);
        if (!g.lean) P->lean = 0;
        if (g.lean && !sub_bit) P->has_fly = 1;
        if (leanw) { g.lean |= 8; P->has_leanw = 1; }
        g.lean |= sub_bit;
    }
    // ---- wave-local plans (hint_wl.hpp): every group lean, narrow lane tile, the block's thin vectors and biases small
    //      enough to ride in LDS twice (HINT_WL=0: never) ----
    const int par_bias = (int)(blob_f_pad + blob_b_pad);
    const int par_f4 = (par_bias + (int)bmap.size() + 3) / 4;
    bool wl = P->lean && dc == 0 && d <= 4 * WL_LV && par_f4 <= WL_PAR_REGS * 64 * nw;
    for (const Unit& u : units) if (u.xoff > 255 || u.h > 32767) wl = false;
    // (the wave-local kernels read the lane table from LDS only and keep per-group tables in the 64 lanes of a register)
    if ((groups.size() + 1) * (size_t)d * sizeof(LaneOp) > 16 * 1024 || groups.size() + 1 > 64 || ents.size() > 65535) wl = false;
    if (!knobs().wl) wl = false;
    // ---- LDS: the meta blob's size, then per group the region [its tiles | its output tiles, staged for the element-wise
    //      phase to stream out, when there is room | its slabs]; the launch reserves the largest group's ----
    auto up16 = [](size_t v) { return (v + 15) & ~(size_t)15; };
    const size_t groups_bytes = up16(groups.size() * sizeof(Group));
    const size_t units_bytes = up16(units.size() * sizeof(Unit));
    const size_t ents_bytes = up16(ents.size() * sizeof(Ent));
    const size_t rng_bytes = up16(rng.size() * sizeof(int32_t));
    const size_t lops_count = (size_t)(groups.size() + 1) * d;
    const bool lops_lds = lops_count * sizeof(LaneOp) <= 16 * 1024;     // larger tables stay in global memory
    const size_t lops_bytes = lops_lds ? up16(lops_count * sizeof(LaneOp)) : 0;
    P->units_off = (int)groups_bytes;
    P->tmap_off = P->units_off + (int)units_bytes;
    P->ents_off = P->tmap_off;
    P->rng_off = P->ents_off + (int)ents_bytes;
    P->lops_off = lops_lds ? P->rng_off + (int)rng_bytes : -1;
    P->meta_bytes = P->rng_off + (int)rng_bytes + (int)lops_bytes;
    // (subtree groups: their slabs, their staged parameters, nw x 16 log-det partials / nw scratch tiles)
    const int sub_par_floats = P->sub_pf + P->sub_pb + P->sub_pbias;
    const int sub_f_bytes = P->n_sub > 0 ? 4 * (pad4(P->sub_slab_f) + sub_par_floats + nw * 16) : 0;
    const int sub_b_bytes = P->n_sub > 0 ? 4 * (pad4(P->sub_slab_b) + sub_par_floats + nw * 512) : 0;
    const int fixed_f = P->meta_bytes + 4 * (2 * ROWS * P->xld + ROWS * P->cld + 2 * ROWS + MAX_NW) + sub_f_bytes;     // (2 x ROWS: the log-det sums of the coupling phase's two halves)
    const int fixed_b = P->meta_bytes + 4 * (3 * ROWS * P->xld + 2 * ROWS * P->cld + ROWS * P->gld + ROWS + ROWS * P->xld) + sub_b_bytes;   // (+ the lanes of the level before: first-layer gradients)
    // Lean groups get their dW1 | db1 from the backward kernel: staged ones in a pass of their own (the g1 tiles wait in LDS),
    // the others - too large to stage - row by row (RowRec flag rowdw: one scratch tile per wavefront), so that g1 never travels.  The scratch tiles count against the LDS the staging decision sees: two passes.
    const bool rowdw_on = !wl && knobs().fuse_dw1;
    int rowdw_bytes = 0;
    std::vector<char> unit_fused(units.size(), 0), unit_rowdw(units.size(), 0);
    for (int pass = 0; pass < 2; ++pass) {
    P->stage_out = 0; P->region_fwd = 0; P->region_bwd = 0;
    std::fill(unit_fused.begin(), unit_fused.end(), 0); std::fill(unit_rowdw.begin(), unit_rowdw.end(), 0);
    bool any_rowdw = false;
    for (size_t gi = 0; gi < groups.size(); ++gi) {
        Group& g = groups[gi];
        const long tiles = (long)g.ntiles * 256;
        if (g.lean & 4) {          // subtree group: no fragment tiles in LDS at all; dW1 / db1 come from the backward kernel
            for (int ui = g.unit_begin; ui < g.unit_end; ++ui) unit_fused[ui] = 1;
            continue;
        }
        const bool staged = fixed_f + 4 * (2 * tiles + grp_slab_f[gi]) <= LDS_LIMIT && fixed_b + rowdw_bytes + 4 * (2 * tiles + grp_slab_b[gi]) <= LDS_LIMIT;
        g.lean &= ~2;
        if (staged) { g.lean |= 2; P->stage_out = 1; }
        P->region_fwd = std::max(P->region_fwd, (int)(tiles * (staged ? 2 : 1) + grp_slab_f[gi]));
        P->region_bwd = std::max(P->region_bwd, (int)(tiles * (staged ? 2 : 1) + grp_slab_b[gi]));
        if (staged && (g.lean & 1))
            for (int ui = g.unit_begin; ui < g.unit_end; ++ui) unit_fused[ui] = 1;
        if (!staged && (g.lean & 1) && rowdw_on) {
            bool ok = true;
            for (int ui = g.unit_begin; ui < g.unit_end; ++ui) if (units[ui].xoff > 255 || units[ui].h > 32767) ok = false;
            if (ok) { any_rowdw = true; for (int ui = g.unit_begin; ui < g.unit_end; ++ui) { unit_fused[ui] = 1; unit_rowdw[ui] = 1; } }
        }
    }
    if (pass == 0 && any_rowdw) { rowdw_bytes = 1024 * nw; continue; }      // (again, with the scratch tiles in the budget)
    if (!any_rowdw) rowdw_bytes = 0;
    break;
    }
    if (!knobs().fuse_dw1) std::fill(unit_fused.begin(), unit_fused.end(), 0);
    if (wl) {
        // LDS of the wave-local kernels (float offsets): [meta | 2 x staged parameters | 2 x nr slab sets | per wavefront: its own
        // tiles of nr row tiles | 32 floats shared] (+ the chain's permutation matrices behind, when the launch finds room)
        auto r4 = [](int v) { return (v + 3) & ~3; };
        for (int nr = 1; nr <= 2; ++nr) {
            WlArgs wf{}, wb{};
            wf.nr = wb.nr = nr;
            wf.par_f4 = wb.par_f4 = par_f4; wf.par_bias = wb.par_bias = par_bias;
            wf.off_par = wb.off_par = P->meta_bytes / 4;
            wf.off_slab = wb.off_slab = wf.off_par + 2 * ((4 * par_f4 + 255) & ~255);      // (whole KiB: the next block's copy arrives by LDS-DMA, 1 KiB per wavefront instruction)
            wf.slab_floats = r4(P->slab_fwd); wb.slab_floats = r4(P->slab_bwd);
            wf.off_priv = wf.off_slab + 2 * nr * wf.slab_floats; wb.off_priv = wb.off_slab + 2 * nr * wb.slab_floats;
            wf.priv_tile = r4(2 * ROWS * P->xld); wf.priv_stride = nr * wf.priv_tile;
            wb.priv_tile = r4(4 * ROWS * P->xld + r4(ROWS * P->gld)); wb.priv_stride = nr * wb.priv_tile + 256;
            wf.off_misc = wf.off_priv + nw * wf.priv_stride; wb.off_misc = wb.off_priv + nw * wb.priv_stride;
            wf.off_recs = wf.off_misc + 32; wb.off_recs = wb.off_misc + 32;
            wf.off_perm = wf.off_recs + 16 * (int)recs_f.size(); wb.off_perm = wb.off_recs + 16 * (int)recs_b.size();
            const bool fits = 4 * wf.off_perm <= LDS_LIMIT && 4 * wb.off_perm <= LDS_LIMIT && (nr == 1 || par_f4 <= WL_PAR_REGS2 * 64 * nw);
            if (nr == 1 && !fits) { wl = false; break; }
            if (fits) { P->wl_f[nr - 1] = wf; P->wl_b[nr - 1] = wb; if (nr == 2) P->wl_nr2 = 1; }
        }
    }
    P->wl = wl ? 1 : 0;
    if (knobs().plan_dump)
        std::fprintf(stderr, "[hint plan] nw %d: wave-local %d (lean %d, par_f4 %d of %d, LDS fwd %d bwd %d bytes; row pairs %d: %d / %d)\n", nw, P->wl,
                     P->lean, par_f4, WL_PAR_REGS * 64 * nw, 4 * P->wl_f[0].off_perm, 4 * P->wl_b[0].off_perm, P->wl_nr2,
                     4 * P->wl_f[1].off_perm, 4 * P->wl_b[1].off_perm);
    if (knobs().plan_dump && P->n_sub > 0)
        std::fprintf(stderr, "[hint plan] nw %d: %d subtree groups (depth >= %d): parameters %d + %d + %d floats, slabs %d / %d floats\n", nw,
                     P->n_sub, sub_depth, P->sub_pf, P->sub_pb, P->sub_pbias, P->sub_slab_f, P->sub_slab_b);
    if (wl) std::fill(unit_fused.begin(), unit_fused.end(), 1);      // dW1 / db1 always come from the backward kernel
    // the first-layer gradients of the fused units: [h][4 or 8] per unit (cin input gradients, then the bias gradient) in a slab per
    // workgroup of the backward kernel; Unit::bias1 (not needed by the kernels otherwise) = the unit's offset in it
    std::vector<int32_t> twmap;
    for (size_t ui = 0; ui < units.size(); ++ui) {
        Unit& u = units[ui];
        u.bias1 = -1;
        if (!unit_fused[ui]) continue;
        const hint_node_desc& n = nodes[unit_node[ui]];
        const int64_t* po = n.p_off + (int)(ui & 1) * 6;
        u.bias1 = (int)twmap.size();
        const int kcp = u.cin < 4 ? 4 : 8;          // slab row of a feature: its cin input gradients, the bias gradient, padding
        for (int f = 0; f < n.h; ++f)
            for (int k = 0; k < kcp; ++k)
                twmap.push_back(k < u.cin ? (int32_t)(po[HINT_W1] + (int64_t)f * u.cin + k) : k == u.cin ? (int32_t)(po[HINT_B1] + f) : -1);
    }
    P->tw_floats = (int)twmap.size();
    // ... and one record per fragment tile, in the thin records' order, for the backward kernel's first-layer-gradient
    // pass: {slab offset of the tile's first feature row, cin | xoff << 8 | valid features << 16, tile inside the group}
    std::vector<ThinRec> thin_w;
    for (const Group& g : groups)
        for (int ui = g.unit_begin; ui < g.unit_end; ++ui) {
            const Unit& u = units[ui];
            for (int nt = 0; nt < u.NT; ++nt)
                thin_w.push_back(ThinRec{u.bias1 < 0 ? 0 : u.bias1 + nt * 16 * (u.cin < 4 ? 4 : 8),
                                         u.cin | (u.xoff << 8) | (std::min(16, u.h - 16 * nt) << 16), u.tile0 + nt, 0});
        }
    for (RowRec& r : recs_f) { r.aux += (int)packed; r.bias3 += (int)packed; }
    if (wl) {
        // the wave-local kernels read thin vectors and biases from the staged parameter buffer [forward blob | backward
        // blob | biases]: offsets relative to it
        for (int q = 0; q < 2; ++q) P->wl_f[q].bias_src = P->wl_b[q].bias_src = (int)packed;
        for (size_t i = 0; i < recs_f.size(); ++i) {
            const Unit& u = units[rec_unit[i]];
            RowRec& f = recs_f[i];
            RowRec& b = recs_b[i];
            f.aux = par_bias + (f.aux - (int)packed); f.bias3 = par_bias + (f.bias3 - (int)packed);
            f.thin_b = (int)blob_f_pad + u.w3v + f.tb * 64;                 // W3^T vectors of the row's first tile
            b.thin_w = (int)blob_f_pad + u.w3v;
            b.thin_b = u.w1v + b.tb * 80;                                    // W1 | b1 vectors of the row's first tile
            b.p1 = u.bias1 + b.tb * 16 * (u.cin < 4 ? 4 : 8);
            b.p2 = u.cin | (u.xoff << 8) | (u.h << 16);
        }
    }

    if (P->n_sub > 0 && !wl) {
        // the subtree groups' rows run on the wave-local row engine: thin vectors and biases relative to the staged
        // parameters [forward vectors | backward vectors | biases] of their units
        const int nrec = groups[P->n_sub].row_begin;
        for (int i = 0; i < nrec; ++i) {
            const Unit& u = units[rec_unit[i]];
            RowRec& f = recs_f[i];
            RowRec& b = recs_b[i];
            f.aux = P->sub_pf + P->sub_pb + (f.aux - (int)packed); f.bias3 = P->sub_pf + P->sub_pb + (f.bias3 - (int)packed);
            f.thin_b = P->sub_pf + u.w3v + f.tb * 64;
            b.thin_w = P->sub_pf + u.w3v;
            b.thin_b = u.w1v + b.tb * 80;
            b.p1 = u.bias1 + b.tb * 16 * (u.cin < 4 ? 4 : 8);
            b.p2 = u.cin | (u.xoff << 8) | (u.h << 16);
        }
    }

    if (rowdw_bytes > 0) {
        for (size_t i = 0; i < recs_b.size(); ++i) {
            const Unit& u = units[rec_unit[i]];
            if (!unit_rowdw[rec_unit[i]]) continue;
            RowRec& b = recs_b[i];
            b.flags |= 1 << 11;
            b.p1 = u.bias1 + b.tb * 16 * (u.cin < 4 ? 4 : 8);
            b.p2 = u.cin | (u.xoff << 8) | (u.h << 16);
        }
    }


    // What the backward kernels walk per boundary (LDS copy, or KArgs::lopsc for the large trees):
    // SLOTS of one thread's work each.  A lane can be a scatter target (an input of the finished group's nodes), a transformed lane
    // of the group about to run, or both; the two halves of a slot's work are independent unless they meet in one lane, and a
    // wavefront runs through both halves whatever its lanes need - so a coupling-only lane and a scatter-only lane share a slot:
    // first the lanes that are both, then the pairs, then what is left of the longer list.  pad = column of the coupling (or only)
    // lane | column of the scatter lane << 16; the slot counts replace the active-lane counts in the ranges table.
    // (MINIBOONE d = 43: 22 / 22 / 23 / 24 / 27 slots for 22 / 32 / 33 / 32 / 35 active lanes - 16 x 33 elements were two passes of the 512
    //  threads; d = 100: up to 52 slots for 76 active lanes.  HINT_PLAN_DUMP=1 prints the list)
    std::vector<LaneOp> lc(lops.size());
    {
        const int nb = (int)(lops.size() / (size_t)d);
        for (int b = 0; b < nb; ++b) {
            const LaneOp* src = lops.data() + (size_t)b * d;
            std::vector<int> both, conly, sonly;
            for (int col = 0; col < d; ++col) {
                const bool sc = src[col].sc_unit >= 0, cp = src[col].cp_ls >= 0;
                if (sc && cp) both.push_back(col); else if (cp) conly.push_back(col); else if (sc) sonly.push_back(col);
            }
            int k = 0;
            for (int col : both) { LaneOp op = src[col]; op.pad = col | (col << 16); lc[(size_t)b * d + k++] = op; }
            const size_t np = std::max(conly.size(), sonly.size());
            for (size_t i = 0; i < np; ++i) {
                LaneOp op{};
                op.sc_unit = -1; op.cp_ls = -1;
                int ca = -1, cb = -1;
                if (i < conly.size()) { const LaneOp& o = src[conly[i]]; ca = conly[i]; op.cp_ls = o.cp_ls; op.cp_lt = o.cp_lt; op.cp_gs = o.cp_gs; op.cp_gt = o.cp_gt; }
                if (i < sonly.size()) { const LaneOp& o = src[sonly[i]]; cb = sonly[i]; op.sc_unit = o.sc_unit; op.sc_k = o.sc_k; }
                if (ca < 0) ca = cb;
                if (cb < 0) cb = ca;
                op.pad = ca | (cb << 16);
                lc[(size_t)b * d + k++] = op;
            }
            rng[P->lop_cnt + b] = k;
            for (; k < d; ++k) { LaneOp op{}; op.sc_unit = -1; op.cp_ls = -1; lc[(size_t)b * d + k] = op; }
        }
    }

    {   // self-check: per boundary every scatter target is the scatter lane of exactly one slot, every transformed lane the coupling
        // lane of exactly one, with its own fields, and no slot names a lane that has neither
        const int nb = (int)(lops.size() / (size_t)d);
        for (int b = 0; b < nb; ++b) {
            const int cnt = rng[P->lop_cnt + b];
            std::vector<int> sc_seen(d, 0), cp_seen(d, 0);
            bool ok = cnt >= 0 && cnt <= d;
            for (int k = 0; ok && k < cnt; ++k) {
                const LaneOp& o = lc[(size_t)b * d + k];
                const int ca = o.pad & 0xffff, cb = (int)((uint32_t)o.pad >> 16);
                if (ca >= d || cb >= d || (o.sc_unit < 0 && o.cp_ls < 0)) { ok = false; break; }
                const LaneOp& pa = lops[(size_t)b * d + ca];
                const LaneOp& pb = lops[(size_t)b * d + cb];
                if (o.cp_ls >= 0) { ++cp_seen[ca]; ok = ok && pa.cp_ls == o.cp_ls && pa.cp_lt == o.cp_lt && pa.cp_gs == o.cp_gs && pa.cp_gt == o.cp_gt; }
                if (o.sc_unit >= 0) { ++sc_seen[cb]; ok = ok && pb.sc_unit == o.sc_unit && pb.sc_k == o.sc_k; }
                if (ca != cb) ok = ok && pa.sc_unit < 0 && pb.cp_ls < 0 && o.cp_ls >= 0 && o.sc_unit >= 0;      // a pair: a coupling-only and a scatter-only lane
            }
            for (int col = 0; ok && col < d; ++col) {
                const LaneOp& pl = lops[(size_t)b * d + col];
                ok = cp_seen[col] == (pl.cp_ls >= 0 ? 1 : 0) && sc_seen[col] == (pl.sc_unit >= 0 ? 1 : 0);
            }
            if (!ok) {
                delete P;
                return fail("hint_plan_create: internal error (boundary slots)");
            }
            P->max_slots = std::max(P->max_slots, cnt);
        }
        if (knobs().plan_dump) {
            std::fprintf(stderr, "[hint plan] backward boundary slots (active lanes):");
            for (int b = 0; b < nb; ++b) std::fprintf(stderr, " %d (%d)", rng[P->lop_cnt + b], lops[(size_t)b * d].pad >> 16);
            std::fprintf(stderr, "\n");
        }
    }

    // ---- meta blob staged in LDS by the kernels ----
    std::vector<char> meta(P->meta_bytes, 0);
    std::memcpy(meta.data(), groups.data(), groups.size() * sizeof(Group));
    std::memcpy(meta.data() + P->units_off, units.data(), units.size() * sizeof(Unit));
    if (!ents.empty()) std::memcpy(meta.data() + P->ents_off, ents.data(), ents.size() * sizeof(Ent));
    std::memcpy(meta.data() + P->rng_off, rng.data(), rng.size() * sizeof(int32_t));
    if (lops_lds) std::memcpy(meta.data() + P->lops_off, lc.data(), lc.size() * sizeof(LaneOp));
    P->lds_fwd = (fixed_f - sub_f_bytes + 4 * P->region_fwd + 15) / 16 * 16;
    P->lds_bwd = (fixed_b - sub_b_bytes + 4 * P->region_bwd + 15) / 16 * 16;
    if (rowdw_bytes > 0) { P->rowdw_lds = P->lds_bwd / 4; P->lds_bwd += rowdw_bytes; }
    if (P->n_sub > 0) {
        P->sub_lds_f[0] = P->lds_fwd / 4; P->sub_lds_f[1] = P->sub_lds_f[0] + pad4(P->sub_slab_f); P->sub_lds_f[2] = P->sub_lds_f[1] + sub_par_floats;
        P->sub_lds_b[0] = P->lds_bwd / 4; P->sub_lds_b[1] = P->sub_lds_b[0] + pad4(P->sub_slab_b); P->sub_lds_b[2] = P->sub_lds_b[1] + sub_par_floats;
        P->lds_fwd += sub_f_bytes; P->lds_bwd += sub_b_bytes;
    }
    // the thin blobs ride in LDS (staged once per block) when they are small
    P->thin_f_off = 0; P->thin_f_floats = (int)((blob_f + 3) / 4 * 4);
    P->thin_b_off = (int)blob_f_pad; P->thin_b_floats = (int)((blob_b + 3) / 4 * 4);
    // (small ones always; larger ones when the block's LDS already rules out two workgroups per CU, or still allows them)
    auto stage_thin = [&](int lds, int blob) {
        return lds + blob <= LDS_LIMIT && (blob <= THIN_LDS_MAX || lds > LDS_LIMIT / 2 || lds + blob <= LDS_LIMIT / 2);
    };
    if (stage_thin(P->lds_fwd, P->thin_f_floats * 4)) {
        P->thin_lds_f = P->lds_fwd / 4;
        P->lds_fwd += P->thin_f_floats * 4;
    }
    if (stage_thin(P->lds_bwd, P->thin_b_floats * 4)) {
        P->thin_lds_b = P->lds_bwd / 4;
        P->lds_bwd += P->thin_b_floats * 4;
    }
    // blobs too large for that (the d = 100 trees: 300 KB): one group's vectors at a time - a run of thin tiles reading them from L2
    // costs 5 k cycles, from LDS 2 k
    {
        // (the buffer: the largest group slice that fits; larger groups - the roots, whose wide thin layers run on the matrix pipe
        //  and read one bias vector per tile - keep reading from L2)
        const bool on = true;
        auto avail = [&](int lds) { return lds > LDS_LIMIT / 2 ? LDS_LIMIT - lds : LDS_LIMIT / 2 - lds; };
        const int av_f = avail(P->lds_fwd) / 4, av_b = avail(P->lds_bwd) / 4;
        int gmax_f = 0, gmax_b = 0;
        for (size_t gi = 0; gi < groups.size(); ++gi) {
            const int t0 = groups[gi].tile_begin, t1 = t0 + groups[gi].ntiles;
            if (t0 >= t1) continue;
            const int f1 = t1 < (int)thin_f.size() ? thin_f[t1].voff : P->thin_f_floats, b1 = t1 < (int)thin_b.size() ? thin_b[t1].voff : P->thin_b_floats;
            const int sf = (f1 - thin_f[t0].voff + 3) / 4 * 4, sb = (b1 - thin_b[t0].voff + 3) / 4 * 4;
            if (sf <= av_f) gmax_f = std::max(gmax_f, sf);
            if (sb <= av_b) gmax_b = std::max(gmax_b, sb);
        }
        if (on && P->thin_lds_f == 0 && gmax_f > 0) { P->thin_lds_f = P->lds_fwd / 4; P->thin_grp_f = gmax_f; P->lds_fwd += 4 * gmax_f; }
        if (on && P->thin_lds_b == 0 && gmax_b > 0) { P->thin_lds_b = P->lds_bwd / 4; P->thin_grp_b = gmax_b; P->lds_bwd += 4 * gmax_b; }
    }
    if (P->lds_bwd > LDS_LIMIT || P->lds_fwd > LDS_LIMIT) {
        const int need = std::max(P->lds_bwd, P->lds_fwd);
        const bool could_shrink = P->abuf_tiles > 0 && P->n_groups < (int)order.size();
        delete P;
        *retry_smaller = tile_cap > 8 && could_shrink;
        fail("hint_plan_create: block needs %d bytes of LDS (> %d); d/dc/h too large", need, LDS_LIMIT);
        return 2;
    }

    if (P->n_sub > 0) {
        // subtree groups: one row per unit, the wavefronts' units in unit order, the tape lanes a partition of [0, d)
        const char* bad = nullptr;
        for (int gi = 0; gi < P->n_sub && !bad; ++gi) {
            const Group& g = groups[gi];
            if (!(g.lean & 4) || rng[g.rng_begin + nw] != g.unit_end - g.unit_begin) bad = "rows of a subtree group";
            int lastw = 0;
            for (int ui = g.unit_begin; ui < g.unit_end && !bad; ++ui) {
                const int w = node_wave[unit_node[ui]];
                if (w < lastw || units[ui].NT != 1 || units[ui].cin > 4 || units[ui].r > 4) bad = "units of a subtree group";
                lastw = w;
            }
        }
        for (int gi = P->n_sub; gi < (int)groups.size() && !bad; ++gi) if (groups[gi].lean & 4) bad = "subtree groups are not the deepest";
        int covered = 0;
        for (int w = 0; w < nw && !bad; ++w) {
            const int32_t* c = rng.data() + P->sub_cols + 4 * w;
            if (c[1] - c[0] > 16 || (c[1] > c[0] && (c[2] > c[0] || c[3] < c[1]))) bad = "lanes of a subtree wavefront";
            if (c[3] > c[2]) { if (c[2] != covered) bad = "tape lanes of the subtree wavefronts"; covered = c[3]; }
        }
        if (!bad && covered != d) bad = "tape lanes of the subtree wavefronts do not cover the block";
        if (bad) {
            delete P;
            return fail("hint_plan_create: internal error (%s)", bad);
        }
    }
    if (const char* what = check_records(groups, units, recs_f, recs_b, rng, nw)) {
        delete P;
        return fail("hint_plan_create: internal error (%s)", what);
    }

    // ---- weight-gradient jobs (part B) and the map of real parameter elements ----
    P->fuse_dw1 = P->tw_floats > 0 ? 1 : 0;       // (some lean, staged groups: their dW1 / db1 come from the backward kernel)
    std::vector<uint8_t> real((size_t)P->param_floats, 0);
    make_wgrad_jobs(P, nodes, units, unit_node, unit_lean, unit_fused, max_depth, &wjobs, &real);
    {   // single-tile jobs last: eight of them share a workgroup (hint_wgrad.hip) - for the trees with subtree groups, whose part B
        // is hundreds of 8 x 8 jobs (MINIBOONE: 272 -> 239 us); the d = 100 trees' single-tile jobs are column remainders of wide
        // arrays, and one wavefront walking 512 rows of a 9600-column array alone is slower (+6 %).  HINT_DW_SMALL=0 / 1 overrides (experiments).
        const bool small_on = knobs().dw_small >= 0 ? knobs().dw_small != 0 : P->n_sub > 0;
        // (single-tile jobs that rebuild their operands read no wide array: for every tree on the general kernels - the d = 100 flows' part B
        //  -4 %; the narrow trees of the wave-local kernels have a handful of them with long batches per split: GAS +5 %, left alone)
        const bool nat_on = !P->wl;
        auto is_small = [&](const WJob& j) { return j.mw <= 1 && j.nw <= 1 && (small_on || (nat_on && j.psrc == WSRC_G2R)); };
        std::stable_partition(wjobs.begin(), wjobs.end(), [&](const WJob& j) { return !is_small(j); });
        if (P->wl) {
            // the workgroups' jobs longest first (tile products per 16-row step): the launch's last wave of workgroups is the short ones
            // (narrow trees: POWER -7 %, GAS -4 %; the d = 100 trees +12 % - there a unit's jobs next to each other share their
            //  operands' rows in L2, which is worth more: not sorted)
            const auto nbig = std::count_if(wjobs.begin(), wjobs.end(), [&](const WJob& j) { return !is_small(j); });
            std::stable_sort(wjobs.begin(), wjobs.begin() + nbig, [](const WJob& x, const WJob& y) {
                return x.mw * std::max(1, x.nw) > y.mw * std::max(1, y.nw); });
            P->wsorted = 1;
        }
        P->n_wsmall = (int)std::count_if(wjobs.begin(), wjobs.end(), is_small);
    }
    P->n_wjobs = (int)wjobs.size();
    P->total_rows = (int)recs_f.size();
    P->total_tiles = (int)thin_f.size();
    P->n_ptiles = (int)ptiles.size();

    if (g_host_only) {           // hint_plan_check: everything above ran (and checked itself); no device
        P->num_cu = 256;
        *out = P;
        return 0;
    }
    // ---- upload ----
    hipDeviceProp_t prop;
    {
        hipError_t e0 = hipGetDevice(&P->device);
        if (e0 == hipSuccess) e0 = hipGetDeviceProperties(&prop, P->device);
        if (e0 != hipSuccess) {
            delete P;
            return fail("hint_plan_create: no device: %s", hipGetErrorString(e0));
        }
    }
    P->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    auto upload = [](void** dst, const void* srcp, size_t bytes) -> hipError_t {
        hipError_t e = hipMalloc(dst, std::max<size_t>(bytes, 16));
        if (e != hipSuccess) return e;
        return bytes ? hipMemcpy(*dst, srcp, bytes, hipMemcpyHostToDevice) : hipSuccess;
    };
    hipError_t e = hipSuccess;
    if (e == hipSuccess) e = upload((void**)&P->d_meta, meta.data(), meta.size());
    if (e == hipSuccess) e = upload((void**)&P->d_lopsc, lc.data(), lc.size() * sizeof(LaneOp));
    if (e == hipSuccess) {
        std::vector<ThinRec> both(thin_f);
        both.insert(both.end(), thin_b.begin(), thin_b.end());
        both.insert(both.end(), thin_w.begin(), thin_w.end());
        e = upload((void**)&P->d_thins, both.data(), both.size() * sizeof(ThinRec));
    }
    if (e == hipSuccess) {
        std::vector<RowRec> both(recs_f);
        both.insert(both.end(), recs_b.begin(), recs_b.end());
        e = upload((void**)&P->d_recs, both.data(), both.size() * sizeof(RowRec));
    }
    if (e == hipSuccess) e = upload((void**)&P->d_bmap, bmap.data(), bmap.size() * sizeof(int32_t));
    if (e == hipSuccess) e = upload((void**)&P->d_real, real.data(), real.size());
    if (e == hipSuccess) e = upload((void**)&P->d_wjobs, wjobs.data(), wjobs.size() * sizeof(WJob));
    if (e == hipSuccess && !twmap.empty()) e = upload((void**)&P->d_twmap, twmap.data(), twmap.size() * sizeof(int32_t));
    if (e == hipSuccess) e = upload((void**)&P->d_segs, segs.data(), segs.size() * sizeof(PackSeg));
    if (e == hipSuccess) e = upload((void**)&P->d_ptiles, ptiles.data(), ptiles.size() * sizeof(int2));
    // the kernels' dynamic-LDS ceiling is a per-kernel attribute: always the hardware limit, so that plans
    // of different sizes created in any order (or on several devices) cannot lower it for each other
    if (e == hipSuccess) e = set_max_lds_apply(LDS_ATTR);
    if (e == hipSuccess) e = set_max_lds_bwd(LDS_ATTR);
    if (e == hipSuccess) e = set_max_lds_bwd_n3(LDS_ATTR);
    if (e == hipSuccess) e = set_max_lds_bwd_fly(LDS_ATTR);
    if (e == hipSuccess) e = set_max_lds_wl_apply(LDS_ATTR);
    if (e == hipSuccess) e = set_max_lds_wl_bwd(LDS_ATTR);
    if (e != hipSuccess) {
        hint_plan_destroy(P);
        return fail("hint_plan_create: device setup failed: %s", hipGetErrorString(e));
    }
    *out = P;
    return 0;
}

// wavefronts per workgroup: HINT_NW (4 or 8) overrides (experiments); otherwise 8 (4 only for blocks narrow enough
// for the backward kernel's register-held [16, d] tile)
static int pick_nw(int d) {
    int nw = 8;
    if (knobs().nw == 4 || knobs().nw == 8 || knobs().nw == 16) nw = std::min(knobs().nw, MAX_NW);
    while (nw < MAX_NW && ROWS * d > LV_REGS * 64 * nw) nw *= 2;
    return nw;
}

// large groups first (fewer phases per block); smaller ones when the block does not fit the LDS, and when the smallest
// groups do not fit either, fewer wavefronts per unit (fewer slabs)
static int plan_for(const hint_node_desc* nodes, int32_t n_nodes, int32_t d, int32_t dc, float clamp, int nw, hint_plan** out) {
    for (int unit_waves = nw; unit_waves >= 1; unit_waves /= 2)
        // (64 tiles = 16 rows of four: two per wavefront; 72 gave 18 rows - three for two of the wavefronts, everybody waits for them:
        //  the d = 100 flows +4 %)
        for (int tile_cap = 64; tile_cap >= 8; tile_cap -= 16) {
            bool retry = false;
            const int st = build_plan(nodes, n_nodes, d, dc, clamp, nw, tile_cap, unit_waves, out, &retry);
            if (st != 2) return st;
            if (!retry) break;
        }
    return 1;          // (the last attempt's message stands)
}

extern "C" {

int hint_plan_create(const hint_node_desc* nodes, int32_t n_nodes, int32_t d, int32_t dc, float clamp,
                     hint_plan** out) {
    if (!nodes || n_nodes <= 0 || d <= 0 || dc < 0 || !out) return fail("hint_plan_create: bad arguments");
    *out = nullptr;
    if (ROWS * d > LV_REGS * 64 * MAX_NW)
        return fail("hint_plan_create: d = %d lanes exceeds the supported maximum %d", d, LV_REGS * 64 * MAX_NW / ROWS);
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
    const int nw = pick_nw(d);
    const int st = plan_for(nodes, n_nodes, d, dc, clamp, nw, out);
    if (st != 0) return st;
    // batches of more row tiles than CUs run two 4-wavefront workgroups per CU instead of one 8-wavefront workgroup
    // after the other (when the block fits twice and the backward kernel's register-held lane tile allows 256 threads)
    if (nw == 8 && knobs().nw == 0 && ROWS * d <= LV_REGS * 64 * 4) {
        const std::string keep = last_error_ref();
        hint_plan* alt = nullptr;
        if (plan_for(nodes, n_nodes, d, dc, clamp, 4, &alt) == 0) {
            const hint_plan* P = *out;
            const bool same_layout = alt->packed_floats == P->packed_floats && alt->n_bias == P->n_bias && alt->WT == P->WT &&
                                     alt->ST == P->ST && alt->param_floats == P->param_floats && alt->lean == P->lean;
            // (a few KiB of margin for the chain's permutation matrices behind a wave-local plan's LDS)
            const int alt_lds = std::max(plan_lds(alt, false), plan_lds(alt, true)) + (alt->wl ? 4096 : 0);
            if (same_layout && alt->wl == P->wl && alt_lds <= LDS_LIMIT / 2) (*out)->alt4 = alt;
            else if (g_host_only) delete alt;
            else hint_plan_destroy(alt);
        }
        last_error_ref() = keep;
    }
    (*out)->nodes.assign(nodes, nodes + n_nodes);
    (*out)->clamp = clamp;
    return 0;
}

int hint_plan_check(const hint_node_desc* nodes, int32_t n_nodes, int32_t d, int32_t dc, float clamp, int64_t* stats) {
    hint_plan* P = nullptr;
    g_host_only = true;
    const int st = hint_plan_create(nodes, n_nodes, d, dc, clamp, &P);
    g_host_only = false;
    if (st != 0) return st;
    if (stats) {
        stats[0] = P->n_groups; stats[1] = P->n_levels; stats[2] = P->WT; stats[3] = P->ST;
        stats[4] = P->lds_fwd; stats[5] = P->lds_bwd; stats[6] = P->nw; stats[7] = P->n_wjobs;
        stats[8] = P->param_floats; stats[9] = P->packed_floats; stats[10] = P->n_units; stats[11] = P->abuf_tiles;
        stats[12] = P->n_sub; stats[13] = P->wl; stats[14] = P->n_wsmall; stats[15] = P->max_slots;
    }
    delete P->alt4;
    delete P;                   // (host-only plans own no device memory)
    return 0;
}

void hint_plan_destroy(hint_plan* P) {
    if (!P) return;
    hint_plan_destroy(P->alt4);
    for (hint_plan* L : P->inv_levels) hint_plan_destroy(L);
    (void)hipFree(P->d_inv_lower);
    (void)hipFree(P->d_meta);
    (void)hipFree(P->d_lopsc);
    (void)hipFree(P->d_recs);
    (void)hipFree(P->d_thins);
    (void)hipFree(P->d_bmap);
    (void)hipFree(P->d_real);
    (void)hipFree(P->d_wjobs);
    (void)hipFree(P->d_twmap);
    (void)hipFree(P->d_segs);
    (void)hipFree(P->d_ptiles);
    delete P;
}

}  // extern "C"
