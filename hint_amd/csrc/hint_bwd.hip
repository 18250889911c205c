// Backward kernel, part A (row parallel; gfx950 / CDNA4 only): gradients with respect to the lanes
// and the condition, and the per-row factors of every weight gradient.
//
// What autograd derives from /root/reference/hint.py:62-101 when the training loop calls
// loss.backward() (train_unconditional.py:137), per node, root first:
//   g_t = g_l' ; g_a = g_l'*exp(a)*l + g_J ; g_l = g_l'*exp(a) ; g_s = g_a*alpha/(1+s^2)
//   g2 = (W3^T g_st) .* relu'(a2) ; g1 = (W2^T g2) .* relu'(a1) ; g_v = W1^T g1 ; g_u += g_v[:k], g_c += g_v[k:]
// Nothing is recomputed: s, the lane tiles of every level and both hidden activations come from the
// forward's tape (bit-identical ReLU masks by construction).  One workgroup carries the gradient
// tile of 16 batch rows through all blocks of the chain, last to first.  Per group two phases:
//   Q1  element-wise: add the g_v partials of the group before, coupling backward -> g_st (LDS + global)
//   Q3  every wavefront runs its rows of the group: the unit's g2 tiles (masked by a2) on the vector
//       ALU -> LDS + global, g1 tiles (masked by a1) on the matrix pipe -> global, and straight from
//       the accumulator the K-split partial of g_v into the row's slab            (run_rows<K_BWD>)
// All weight gradients are batch reductions of (g1, g2, g_st) against (v, a1, a2): part B
// (hint_wgrad.hip) computes them from the arrays written here.
// rows of up to FOUR tiles in the general kernels (the wave-local ones keep three): a row's steps wait for L2 however few MFMAs
// they hold, so fewer, wider rows - h = 56 as one row instead of 2 + 2 - halve what a unit costs (hint_plan.cpp: GEN_NTT)
#ifndef HINT_NTT
#define HINT_NTT 4
#endif
#ifndef HINT_PF_DIST
#define HINT_PF_DIST 2
#endif
// (g2 / g1 tiles of groups that are not lean through LDS and out in whole lines at the next boundary: slower than the rows' own stores -
//  cfg 5 backward 417 -> 396 us and part B, which reads them, 179 -> 172; d = 100 763 -> 754.  Lean staged groups keep g1 in LDS: their dW1 pass reads it.)
#ifndef HINT_BWD_STAGE
#define HINT_BWD_STAGE false
#endif
#include "hint_sub.hpp"

using namespace hint;

#ifdef HINT_BWD_FLY       // (hint_bwd_fly.hip: the instance for plans with lean general groups - their rows make g2' themselves, no thin phase)
constexpr bool BWD_FLYK = true;
#else
constexpr bool BWD_FLYK = false;
#endif

struct LevelPrefetch { float x[LV_REGS], s[LV_REGS]; int nvalid; };
// (the loads only: nothing here may look at the loaded values - a select on them would be a wait for HBM in the
//  middle of the group; rows beyond the batch are zeroed when the tile is committed)
__device__ __forceinline__ void level_issue(LevelPrefetch& p, const float* __restrict__ xsrc, const float* __restrict__ ssrc,
                                            int d, int row0, int B, int tid, int nthreads) {
    const int nvalid = (B - row0 < ROWS ? B - row0 : ROWS) * d;
    const size_t base = (size_t)row0 * d;
    p.nvalid = nvalid;
#pragma unroll
    for (int k = 0; k < LV_REGS; ++k) {
        const int i = tid + k * nthreads;
        const int ic = i < nvalid ? i : 0;
        p.x[k] = xsrc[base + ic];
        p.s[k] = ssrc[base + ic];
    }
}
__device__ __forceinline__ void level_commit(const LevelPrefetch& p, float* xs, float* sb, int ld, int d, int tid, int nthreads) {
    const float inv = frcp(d);
#pragma unroll
    for (int k = 0; k < LV_REGS; ++k) {
        const int i = tid + k * nthreads;
        if (i < ROWS * d) {
            const int r = fdiv(i, inv), j = i - r * d;
            xs[r * ld + j] = i < p.nvalid ? p.x[k] : 0.f;
            sb[r * ld + j] = i < p.nvalid ? p.s[k] : 0.f;
        }
    }
}

__global__ __launch_bounds__(64 * MAX_NW) void hint_bwd_kernel(
    KArgs a, ChainBlock one, const ChainBlock* __restrict__ chain, int n_chain,
    const float* __restrict__ x, const float* __restrict__ c,
    const float* __restrict__ g_z, const float* __restrict__ g_J, float* __restrict__ g_x,
    float* __restrict__ g_c, float gz_scale, float gJ_const) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, nthreads = blockDim.x;
    const int lane = tid & 63;
    const int wave = rfl(tid >> 6);
    const float inv_d = frcp(a.d);
    const Tables T = make_tables(a, lds);
    float* xs = lds + (a.meta_bytes >> 2);    // lane tile of the current level, as the forward pass saw it
    float* gs = xs + ROWS * a.xld;            // gradient tile
    float* sb = gs + ROWS * a.xld;            // s of the current level (tape)
    float* cs = sb + ROWS * a.xld;
    float* gcs = cs + ROWS * a.cld;
    float* gst = gcs + ROWS * a.cld;          // [16][gld] coupling gradients of the group
    float* abuf = gst + ROWS * a.gld;         // per group: [g2 tiles | g1 tiles on their way out (staged groups) | g_v partials (slabs)]
    float* gj = abuf + a.region_floats;
    float* xo = gj + ROWS;                    // the lane tile of the level before (first-layer gradients of the group just finished)
    float* thinb = lds + a.thin_lds;          // the block's thin-layer vectors (when the launch found LDS for them)
    const int ntiles = (a.B + ROWS - 1) / ROWS;
    const size_t lvl = (size_t)a.B * a.d;     // floats of one [B, d] tape slice
    STAMP_DECL()
    copy_meta(a, lds, tid, nthreads);
#define HINT_CB(I) chain_block(chain, one, I)
    // input lanes of level LV of a block: x for the deepest level of the first block of a call without
    // a fused permutation, the top tape slice (hint_apply_kernel stored it) for other deepest levels,
    // tape[LV-1] otherwise
#define LEVEL_SRC(TAPE, TOP, LV) ((LV) == 0 ? ((TOP) ? (TAPE) + (size_t)(a.n_levels - 1) * lvl : x) : (TAPE) + (size_t)((LV) - 1) * lvl)

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int row0 = tile * ROWS;
        load_tile(gs, a.xld, g_z, a.d, row0, a.B, tid, nthreads);
        if (a.dc > 0) {
            load_tile(cs, a.cld, c, a.dc, row0, a.B, tid, nthreads);
            load_tile(gcs, a.cld, nullptr, a.dc, row0, a.B, tid, nthreads);
        }
        if (tid < ROWS) gj[tid] = (row0 + tid < a.B) ? (g_J != nullptr ? g_J[row0 + tid] : gJ_const) : 0.f;
        {
            const GBlock lb = HINT_CB(n_chain - 1);
            const float* tape = (const float*)lb.tape;
            const bool top = lb.perm != nullptr || n_chain > 1;
            load_tile(xs, a.xld, LEVEL_SRC(tape, top, a.n_levels - 1), a.d, row0, a.B, tid, nthreads);
            load_tile(sb, a.xld, tape + (size_t)(a.n_levels + a.n_levels - 1) * lvl, a.d, row0, a.B, tid, nthreads);
        }
        __syncthreads();
        if (gz_scale != 1.f) {                 // loss gradient fused: g_z = z / B given z
            for (int i = tid; i < ROWS * a.d; i += nthreads) { const int r = fdiv(i, inv_d); gs[r * a.xld + (i - r * a.d)] *= gz_scale; }
            __syncthreads();
        }

        for (int cb = n_chain - 1; cb >= 0; --cb) {
            const GBlock blk = HINT_CB(cb);
            const GBlock nblk = HINT_CB(cb > 0 ? cb - 1 : 0);          // the block worked on after this one
            const float* perm = (const float*)blk.perm;
            const float* tape = (const float*)blk.tape;
            const bool top = perm != nullptr || cb > 0;
            float* wsGST = (float*)blk.wsGST;
            {   // the subtree groups' parameters and the block's thin-layer vectors -> LDS (every wavefront re-reads them for its units,
                // and all workgroups asking L2 for the same few lines at once is what made them slow)
                const bool thin_blk = a.thin_lds > 0 && a.thin_grp == 0;
                // (the subtree parameters as well, although their phase is the block's last: staged in front of it, behind the boundary's
                //  element-wise work, they cost MINIBOONE's backward 5 us more)
                block_stage(a, blk.packed, lds, true, thin_blk ? a.thin_floats >> 2 : 0, tid, nthreads);
            }
            if ((a.thin_lds > 0 && a.thin_grp == 0) || a.n_sub > 0) __syncthreads();
            PhaseCtx pc;
            pc.packed = blk.packed;
            pc.thin_l = a.thin_lds > 0 ? (const LDS_AS float*)thinb : nullptr;
            pc.thin_g = blk.packed + a.thin_off;
            pc.recs = (const char*)a.recs + (size_t)a.total_rows * sizeof(RowRec);
            pc.abuf = (LDS_AS float*)abuf; pc.obuf = nullptr; pc.slab = nullptr;      // (obuf, slab: per group)
            pc.out_thin = nullptr; pc.out_main = blk.wsG1; pc.wcol0 = 0;      // (out_thin: per group - lean groups keep no g2)
            pc.xs = (const LDS_AS float*)xs; pc.cs = (const LDS_AS float*)cs; pc.gst = (const LDS_AS float*)gst;
            pc.bits_a1 = (GLOBAL_AS uint8_t*)(blk.actA1 + a.bits_off) + (size_t)(row0 >> 4) * (a.WT >> 4) * 64;
            pc.bits_a2 = pc.bits_a1 + a.bits_stride;
            pc.xld = a.xld; pc.cld = a.cld; pc.gld = a.gld; pc.WT = a.WT; pc.row0 = row0;
            pc.store = true; pc.fly = false;
            pc.scratch = (LDS_AS float*)(lds + a.rowdw_lds + wave * 256);
            pc.tw = blk.wsSlab + a.thin_slab_off + (size_t)blockIdx.x * a.tw_floats;
            pc.first_tile = tile == (int)blockIdx.x;

            f32x4 ring[RING][NEL];
            RecCarry rcar; rcar.held = -1; rcar.v = 0;
            {   // the weight stream (and the a2 tiles) of the root group's first row: started before its coupling phase
                const GroupU g0 = load_group(T.groups + (a.n_groups - 1));
                const LDS_AS int32_t* rng0 = T.rng + g0.rng_begin;
                rows_begin<K_BWD>(pc, ring, rcar, g0.row_begin + lds_i32(rng0 + wave), g0.row_begin + lds_i32(rng0 + wave + 1), lane);
            }
            for (int gi = a.n_groups; gi >= 0; --gi) {
                // gi == 0 .. n_groups-1: the boundary in front of group gi (root first), then its GEMM phases;
                // slot n_groups is used for the boundary BEHIND group 0 (scatter only), visited last
                const int slot = gi > 0 ? gi - 1 : a.n_groups;
                const bool tail_only = gi == 0;
                const GroupU g = load_group(T.groups + (tail_only ? 0 : slot));
                const int lop0 = tail_only ? a.n_groups * a.d : g.lop_begin;
                // units of the group just finished (whose g_v partials wait in the slabs)
                const bool has_prev = gi < a.n_groups;
                const GroupU gp = load_group(T.groups + (has_prev ? (tail_only ? 0 : slot + 1) : 0));

                const int sid = ((n_chain - 1 - cb) * (a.n_groups + 1) + (a.n_groups - gi)) * 16;
                (void)sid;
                pc.sid = sid;
                STAMP(sid + 0)
                // the finished group's g2 (masked) and g1 tiles: out of LDS to the workspace, whole lines per batch row
                // (by the wavefronts the element-wise work below does not need)
                int qthreads = nthreads;                            // threads of the element-wise phase
                // (the finished group's carving of the region: its g1 tiles and its slabs)
                float* obuf = abuf + gp.ntiles * 256;
                const bool gp_staged = gp.staged && (gp.lean || HINT_BWD_STAGE);      // (lean staged groups keep g1 in LDS for their dW1 pass)
                float* slab = obuf + (gp_staged ? gp.ntiles * 256 : 0);
                // (slots of the boundary below: its threads; the others move the finished group's tiles meanwhile)
                const int nact = lds_i32(T.rng + a.lop_cnt + (tail_only ? a.n_groups : slot));
                if (has_prev && gp_staged) {
                    const int need = (ROWS * (nact > a.dc ? nact : a.dc) + 63) & ~63;
                    const int soff = need < nthreads ? need : 0;
                    if (soff > 0) qthreads = soff;
                    if (tid >= soff) {
                        if (!gp.nothin) stream_tiles((float*)blk.wsG1 + a.act_stride, abuf, gp.ntiles, gp.wcol0, a.WT, row0, tid - soff, nthreads - soff);
                        if (!(a.fuse_dw1 && gp.lean)) {
                            stream_tiles((float*)blk.wsG1, obuf, gp.ntiles, gp.wcol0, a.WT, row0, tid - soff, nthreads - soff);
                        } else {
                            // dW1[f][k] = sum_rows g1[row][f] v[row][k], db1[f] = sum_rows g1[row][f] of the finished group's units,
                            // from the g1 tiles in LDS and the lanes its level saw: into this workgroup's slab (plain stores;
                            // hint_wreduce_kernel adds the workgroups' slabs in order).  g1 never leaves the chip.
                            float* tw = (float*)blk.wsSlab + a.thin_slab_off + (size_t)blockIdx.x * a.tw_floats;
                            const bool first_tile = tile == (int)blockIdx.x;
                            // One 16-feature tile per wavefront and turn, on the matrix pipe: out[f][k] = sum over the 16 rows
                            // as four 16x16x4 MFMAs (A = g1^T from the fragment tile: lane l gives feature l&15 of row 4i + (l>>4);
                            // B = the lanes, column cin = 1 for the bias gradient).
                            const int ws = rfl((tid - soff) >> 6), nws = (nthreads - soff) >> 6;
                            const int nl = lane & 15, kq = lane >> 4;
                            const CONST_AS i32x4c* wrec = (const CONST_AS i32x4c*)(unsigned long long)((const char*)a.thins +
                                                          (size_t)2 * a.total_tiles * sizeof(ThinRec)) + gp.tile_begin;
                            // (transposed: out^T[k][f], so that a lane ends up with four consecutive inputs of ONE feature - a row of the
                            //  slab, padded to 4 or 8 floats: one 16-byte store per lane, 256 contiguous bytes per tile)
                            i32x4c nrec = wrec[ws < gp.ntiles ? ws : 0];                      // (records one tile ahead: a scalar load takes as long as the tile)
                            for (int t = ws; t < gp.ntiles; t += nws) {
                                const i32x4c rec = nrec;
                                nrec = wrec[t + nws < gp.ntiles ? t + nws : t];
                                const int cin = rec.y & 0xff, xoff = (rec.y >> 8) & 0xff, nvalid = rec.y >> 16, kcp = cin < 4 ? 4 : 8;
                                const LDS_AS float* g1p = (const LDS_AS float*)obuf + (rec.z * 64 + kq + 16 * (nl >> 2)) * 4 + (nl & 3);
                                const LDS_AS float* vp = (const LDS_AS float*)xo + kq * a.xld + xoff + (nl < cin ? nl : 0);
                                const float one = nl == cin ? 1.f : 0.f;
                                f32x4 acc = zero4();
                                float av[4], bv[4];                       // (all eight LDS reads in flight: no read under a lane mask)
#pragma unroll
                                for (int i = 0; i < 4; ++i) { av[i] = g1p[16 * i]; bv[i] = vp[4 * i * a.xld]; }     // rows 4i + kq
#pragma unroll
                                for (int i = 0; i < 4; ++i) acc = mfma4(nl < cin ? bv[i] : one, av[i], acc);
                                if (nl < nvalid && 4 * kq < kcp) {
                                    f32x4* dst = (f32x4*)(tw + rec.x + nl * kcp + 4 * kq);     // feature nl of the tile, inputs 4 kq .. 4 kq + 3
                                    if (first_tile) *dst = acc; else *dst = *dst + acc;
                                }
                            }
                        }
                    }
                }
                // (this group's thin vectors -> LDS when the block's are too many for it: read behind the barrier below)
                bool thin_staged = a.thin_lds > 0;
                if (a.thin_grp > 0 && !tail_only && slot >= a.n_sub) {
                    pc.thin_l = thin_group_stage(thinb, a.thin_grp, blk.packed + a.thin_off, (const char*)a.thins + (size_t)a.total_tiles * sizeof(ThinRec),
                                                 g.tile_begin, g.ntiles, a.total_tiles, a.thin_floats, tid, nthreads);
                    thin_staged = pc.thin_l != nullptr;
                }
                // ---- Q1: scatter of the previous group's g_v + coupling backward of this one ----
                // the boundary's SLOTS (hint_plan.cpp): a thread adds the finished group's g_v partials onto one lane (`colb`) and forms the
                // coupling gradients of one transformed lane (`col`) - the same lane, or a scatter-only and a coupling-only lane that share the
                // slot (a wavefront runs through both halves whatever its lanes need); from the compacted table in LDS, or in global memory for
                // the large trees.  16 x 22-27 slots at d = 43 (32-35 active lanes of 43: two passes of the workgroup), up to 16 x 52 at d = 100 (76 lanes).
                const bool lop_g = a.lops_off < 0;
                const float inv_n = frcp(nact > 0 ? nact : 1);
                for (int idx = tid; idx < ROWS * nact && tid < qthreads; idx += qthreads) {
                    const int row = fdiv(idx, inv_n);
                    i32x4 lq;
                    if (lop_g) lq = ((const GLOBAL_AS i32x4*)a.lopsc)[lop0 + idx - row * nact];
                    else lq = *(const LDS_AS i32x4*)(T.lops + lop0 + idx - row * nact);
                    const unsigned w0 = (unsigned)lq.x, w1 = (unsigned)lq.y, w2 = (unsigned)lq.z;
                    const int col = lq.w & 0xffff, colb = (int)((unsigned)lq.w >> 16);
                    const int sc_unit = (int)(int16_t)(w0 & 0xffffu), sc_k = (int)(w0 >> 16);
                    const int cp_ls = (int)(int16_t)(w1 & 0xffffu), cp_lt = (int)(w1 >> 16);
                    const int cp_gs = (int)(w2 & 0xffffu), cp_gt = (int)(w2 >> 16);
                    float gval = gs[row * a.xld + col];
                    if (sc_unit >= 0) {
                        float gb = gs[row * a.xld + colb];
#pragma unroll
                        for (int net = 0; net < 2; ++net) {
                            const LDS_AS int32_t* up = (const LDS_AS int32_t*)(T.units + sc_unit + net);
                            const int cin = up[15], sl_n = up[21], gv_off = up[22];
                            const int stride = 64 * ((cin + 3) >> 2);
                            const float* sp = slab + gv_off + ((sc_k >> 2) * 16 + row) * 4 + (sc_k & 3);
                            for (int sl = 0; sl < sl_n; ++sl) gb += sp[sl * stride];
                        }
                        if (colb == col) gval = gb;
                        else gs[row * a.xld + colb] = gb;
                    }
                    if (!tail_only && cp_ls >= 0) {
                        const float s = sb[row * a.xld + col];
                        const float aa = a.alpha * atanf(s);
                        const float ea = expf(aa);
                        const float l = xs[row * a.xld + col];            // lower input of the node
                        const float ga = gval * ea * l + gj[row];           // g_a (a feeds both l' and J)
                        const float gsv = ga * a.alpha * __builtin_amdgcn_rcpf(1.f + s * s);     // g_s (v_rcp_f32, 1 ulp: an IEEE division is ten vector instructions)
                        gst[row * a.gld + cp_ls] = gsv;
                        gst[row * a.gld + cp_lt] = gval;                    // g_t = g_l'
                        float* go = wsGST + (size_t)(row0 + row) * a.ST;
                        go[cp_gs] = gsv;
                        go[cp_gt] = gval;
                        gval *= ea;                                         // g_l
                    }
                    gs[row * a.xld + col] = gval;
                }
                if (a.dc > 0 && has_prev) {
                    // condition columns: every unit of the previous group contributes (fixed order: deterministic)
                    for (int idx = tid; idx < ROWS * a.dc && tid < qthreads; idx += qthreads) {
                        const int row = idx / a.dc, cc = idx - row * a.dc;
                        float acc = gcs[row * a.cld + cc];
                        for (int u = gp.unit_begin; u < gp.unit_end; ++u) {
                            const LDS_AS int32_t* up = (const LDS_AS int32_t*)(T.units + u);
                            const int cin = up[15], ku = up[16], sl_n = up[21], gv_off = up[22];
                            const int k = ku + cc;
                            const int stride = 64 * ((cin + 3) >> 2);
                            const float* sp = slab + gv_off + ((k >> 2) * 16 + row) * 4 + (k & 3);
                            for (int sl = 0; sl < sl_n; ++sl) acc += sp[sl * stride];
                        }
                        gcs[row * a.cld + cc] = acc;
                    }
                }
                STAMP(sid + 1)
                lds_barrier();
                STAMP(sid + 2)
                if (tail_only) break;
                if (slot < a.n_sub) {
                    // ---- the subtree groups: from here down every wavefront runs its own subtrees, wave-local synchronisation
                    //      only (hint_sub.hpp); the root level of the block before is fetched meanwhile ----
                    sub_bwd(a, T, lds, blk, x, top, xs, sb, gs, gst, gj, row0, tile == (int)blockIdx.x, wave, lane, sid);
                    STAMP(sid + 3)
                    LevelPrefetch lq;           // (issued behind the subtree phase: nine registers less in it; the barrier's wait hides most of it)
                    if (cb > 0) {
                        const float* ntape = (const float*)nblk.tape;
                        const bool ntop = nblk.perm != nullptr || cb > 1;
                        level_issue(lq, LEVEL_SRC(ntape, ntop, a.n_levels - 1), ntape + (size_t)(2 * a.n_levels - 1) * lvl, a.d, row0, a.B, tid, nthreads);
                    }
                    lds_barrier();
                    STAMP(sid + 4)
                    if (cb > 0) { level_commit(lq, xs, sb, a.xld, a.d, tid, nthreads); lds_barrier(); }
                    break;
                }

                // ---- lane tile and s of the level the NEXT boundary needs: global -> registers now,
                //      registers -> LDS at the end of Q3 (xs / sb are not read by the GEMM phases) ----
                LevelPrefetch lp;
                bool lp_pending = false;
                {
                    const bool block_switch = slot == 0;                     // next: root level of the block before
                    int nlevel = a.n_levels - 1;
                    if (!block_switch) nlevel = lds_i32((const LDS_AS int32_t*)(T.groups + slot - 1) + 7);
                    if (block_switch ? cb > 0 : nlevel != g.level) {
                        const float* ntape = block_switch ? (const float*)nblk.tape : tape;
                        const bool ntop = block_switch ? (nblk.perm != nullptr || cb > 1) : top;
                        level_issue(lp, LEVEL_SRC(ntape, ntop, nlevel), ntape + (size_t)(a.n_levels + nlevel) * lvl, a.d, row0, a.B,
                                    tid, nthreads);
                        lp_pending = true;
                    }
                }
                const LDS_AS int32_t* rng = T.rng + g.rng_begin;
                STAMP(sid + 3)
                // ---- Q2: g2' = W3^T g_st of every unit of the group on the vector ALU, the tiles shared out - not for a lean group with
                //      its thin vectors in LDS in the instance whose rows make those fragments themselves (hint_rows.hpp row_body FLY) ----
                pc.fly = BWD_FLYK && g.lean && thin_staged;
                if (!pc.fly) {
                    const char* thb = (const char*)a.thins + (size_t)a.total_tiles * sizeof(ThinRec);
                    const int t0 = g.tile_begin + lds_i32(rng + a.nw + 1 + wave), t1 = g.tile_begin + lds_i32(rng + a.nw + 2 + wave);
                    if (thin_staged) thin_phase<K_BWD, true>(pc, thb, t0, t1, lane);
                    else thin_phase<K_BWD, false>(pc, thb, t0, t1, lane);
                    lds_barrier();
                }
                STAMP(sid + 4)
                // ---- Q3: g1 = (W2^T (g2' .* relu'(a2))) .* relu'(a1);  g_v partial = W1^T g1 ----
                pc.out_thin = g.nothin ? nullptr : (GLOBAL_AS float*)(blk.wsG1 + a.act_stride);       // (lean and lean-wide groups keep no g2)
                const bool g_staged = g.staged && (g.lean || HINT_BWD_STAGE);
                pc.obuf = g_staged ? (LDS_AS float*)(abuf + g.ntiles * 256) : nullptr;
                pc.slab = (LDS_AS float*)(abuf + g.ntiles * 256 * (1 + g_staged));
                {
                    int rnext = -1;         // the last row hands the weight ring to the wavefront's first row of the next group
                    if (slot > a.n_sub) {
                        const GroupU gn = load_group(T.groups + (slot - 1));
                        const LDS_AS int32_t* rngn = T.rng + gn.rng_begin;
                        const int n0 = lds_i32(rngn + wave), n1 = lds_i32(rngn + wave + 1);
                        if (n0 < n1) rnext = gn.row_begin + n0;
                    }
#if defined(HINT_BWD_FLY) || defined(HINT_BWD_PRIO_ALL)
                    // (the SIMD's younger wavefront first while the rows run - see hint_wl_bwd.hip; measured: the lean-groups instance -2.2 % at d = 100,
                    //  the MINIBOONE instance +-0, the forward / inverse kernels +1-3 %)
                    if (wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif
                    rows_run<K_BWD, BWD_FLYK>(pc, ring, rcar, g.row_begin + lds_i32(rng + wave), g.row_begin + lds_i32(rng + wave + 1), rnext, lane);
#if defined(HINT_BWD_FLY) || defined(HINT_BWD_PRIO_ALL)
                    __builtin_amdgcn_s_setprio(0);
#endif
                }
                STAMP(sid + 15)
                STAMP(sid + 5)
                if (a.sink_lds > 0 && wave == a.nw - 1) {       // (L2 warm-up two consumers ahead: hint_device.hpp prefetch_consumer; backward order: root first)
                    const int ngen = a.n_groups - a.n_sub, pos = a.n_groups - slot;
                    auto pf = [&](int q) {
                        const GLOBAL_AS float* pk = blk.packed;
                        if (q > ngen) {
                            if (cb == 0) return;
                            q -= ngen + 1;
                            pk = nblk.packed;
                        }
                        prefetch_consumer<false>(a, T, pk, q, lds + a.sink_lds, lane);
                    };
                    pf(pos + HINT_PF_DIST);
                    if (pos == 1) for (int q = 2; q <= HINT_PF_DIST; ++q) pf(q);      // (the head has no phase of its own)
                    if (slot == a.n_sub && perm != nullptr) prefetch_range((const GLOBAL_AS float*)perm, 0, (a.d * a.d * 4 + 127) >> 7, lds + a.sink_lds, lane);   // (the matrix behind the block)
                }
                if (a.fuse_dw1)          // the lanes this group's first layers read: kept for their weight gradients (computed across the next boundary)
                    for (int i = tid; i < ROWS * a.d; i += nthreads) { const int r = fdiv(i, inv_d), j = i - r * a.d; xo[r * a.xld + j] = xs[r * a.xld + j]; }
                // (rows that compute dW1 | db1 themselves read the level's lanes to their end: nobody overwrites them before all are through)
                if (lp_pending && a.rowdw_lds > 0 && g.lean && !g.staged) lds_barrier();
                if (lp_pending) level_commit(lp, xs, sb, a.xld, a.d, tid, nthreads);
                lds_barrier();
                STAMP(sid + 6)
            }
            if (blk.g_add != nullptr) {            // the second consumer of the block's (permuted) input: ChainBlock::g_add
                const int nv = (a.B - row0 < ROWS ? a.B - row0 : ROWS) * a.d;
                for (int i = tid; i < nv; i += nthreads) { const int r = fdiv(i, inv_d); gs[r * a.xld + (i - r * a.d)] += blk.g_add[(size_t)row0 * a.d + i]; }
                __syncthreads();
            }
            if (perm != nullptr) {                 // chain rule through x' = x W:  g_x = g_x' W^T
                // (in place: the products wait in registers for the barrier)
                f32x4 pacc[PERM_TQ];
                perm_mfma<true>(pacc, gs, a.xld, (const GLOBAL_AS float*)perm, a.d, wave, a.nw, lane);
                __syncthreads();
                perm_store(pacc, gs, a.xld, a.d, wave, a.nw, lane);
                __syncthreads();
            }
        }
        store_tile(g_x, gs, a.xld, a.d, row0, a.B, tid, nthreads);
        if (a.dc > 0 && g_c != nullptr) store_tile(g_c, gcs, a.cld, a.dc, row0, a.B, tid, nthreads);
        __syncthreads();
    }
    STAMP_FLUSH(a.stamps)
#undef HINT_CB
#undef LEVEL_SRC
}

namespace hint {

hipError_t launch_bwd(const KArgs& a, int lds_bytes, int grid, const ChainBlock& one, const ChainBlock* chain,
                      int n_chain, const float* x, const float* c, const float* g_z, const float* g_J,
                      float* g_x, float* g_c, float gz_scale, float gJ_const, hipStream_t stream) {
    hipLaunchKernelGGL(hint_bwd_kernel, dim3(grid), dim3(64 * a.nw), lds_bytes, stream, a, one, chain, n_chain, x, c,
                       g_z, g_J, g_x, g_c, gz_scale, gJ_const);
    return hipGetLastError();
}

hipError_t set_max_lds_bwd(int bytes) {
    return hipFuncSetAttribute((const void*)hint_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

}  // namespace hint
