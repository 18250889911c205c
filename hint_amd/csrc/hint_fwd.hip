// Forward / inverse kernel (gfx950 / CDNA4 only) of HINT's recursive affine-coupling block.
//
// Arithmetic reproduced (reference, read-only): /root/reference/hint.py:62-101
//   per node:  v = [u | c];  s = mlp_s(v), t = mlp_t(v)            (hint.py:76-77, :10-13)
//              a = alpha*atan(s), alpha = clamp*0.636               (hint.py:56-60)
//   forward    l' = exp(a)*l + t ;  J += sum a   (children first)   (hint.py:70-80,97-99)
//   inverse    l  = (l' - t)/exp(a); J -= sum a  (root first)       (hint.py:82-88)
//
// One workgroup owns a tile of 16 batch rows and carries it through ALL tree levels of ALL blocks
// of a flow inside one launch; the lane tile and the condition stay in LDS, HBM sees x once in and
// z, J once out (plus the training tape).  Per group of same-depth nodes two phases:
//   P2  every wavefront runs its rows of the group: the unit's first layer on the vector ALU (a1
//       fragment tiles in LDS), the second layer on the matrix pipe and, straight from its
//       accumulator, the third layer's K-split partial (+ bias) into the row's slab   (run_rows<K_FWD>)
//   P3  coupling: s, t = sum of the slabs; l' = exp(a) l + t; log-det
// rows of up to FOUR tiles in the general kernels (the wave-local ones keep three): a row's steps wait for L2 however few MFMAs
// they hold, so fewer, wider rows - h = 56 as one row instead of 2 + 2 - halve what a unit costs (hint_plan.cpp: GEN_NTT)
#define HINT_NTT 4
#ifndef HINT_PF_DIST
#define HINT_PF_DIST 2
#endif
#ifndef HINT_FWD_STAGE
#define HINT_FWD_STAGE false
#endif
#ifndef HINT_FLY_ON
#define HINT_FLY_ON true
#endif
#include "hint_sub.hpp"

using namespace hint;

// FLYK: the instance for plans with lean general groups (hint_plan::has_fly) - their rows make the first layer themselves, no
// thin phase (hint_rows.hpp row_body FLY); plans without such groups keep the instance without that code (its mere presence cost
// MINIBOONE's forward 20 us of 300)
template <bool REV, bool FLYK>
__global__ __launch_bounds__(64 * MAX_NW) void hint_apply_kernel(
    KArgs a, ChainBlock one, const ChainBlock* __restrict__ chain, int n_chain,
    const float* __restrict__ x, const float* __restrict__ c, float* __restrict__ z,
    float* __restrict__ J, const float* __restrict__ J_in, float* __restrict__ loss_acc,
    float noise, const unsigned long long* __restrict__ rng_state, float* __restrict__ x_noisy) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, nthreads = blockDim.x;
    const int lane = tid & 63;
    const int wave = rfl(tid >> 6);
    const float inv_d = frcp(a.d);
    const Tables T = make_tables(a, lds);
    float* t0 = lds + (a.meta_bytes >> 2);    // two lane tiles: a fused permutation ping-pongs between them
    float* cs = t0 + 2 * ROWS * a.xld;
    float* abuf = cs + ROWS * a.cld;          // per group: [a1 tiles | a2 tiles on their way to the tape (staged groups) | slabs]
    float* jac = abuf + a.region_floats;
    float* jac2 = jac + ROWS;                 // ... of the coupling phase's second half (P3)
    float* red = jac2 + ROWS;                 // MAX_NW floats: loss partials
    float* thinb = lds + a.thin_lds;          // the block's thin-layer vectors (when the launch found LDS for them)
    const int ntiles = (a.B + ROWS - 1) / ROWS;
    STAMP_DECL()
    copy_meta(a, lds, tid, nthreads);
    // the chain's fixed d x d permutation matrices, once per workgroup (when the launch found LDS for them)
    float* ptab = lds + a.perm_lds;
    const int pdd = a.d * a.d;
    if (a.perm_lds > 0) {
        for (int i = tid; i < n_chain * pdd; i += nthreads) {
            const int cbi = fdiv(i, frcp(pdd));
            const float* pp = (chain != nullptr) ? chain[cbi].perm : one.perm;
            ptab[i] = pp != nullptr ? ((const GLOBAL_AS float*)pp)[i - cbi * pdd] : 0.f;
        }
    }
#define HINT_CB(I) chain_block(chain, one, I)

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int row0 = tile * ROWS;
        // the current lane tile is t0 + xcur (an integer offset, not a swapped pointer: the compiler
        // must keep seeing LDS addresses or it falls back to flat loads)
        int xcur = 0;
        const int xflip = ROWS * a.xld;
#define XS (t0 + xcur)
#define XO (t0 + (xflip - xcur))
        load_tile(XS, a.xld, x, a.d, row0, a.B, tid, nthreads);
        if (a.dc > 0) load_tile(cs, a.cld, c, a.dc, row0, a.B, tid, nthreads);
        if (tid < 2 * ROWS) jac[tid] = 0.f;
        __syncthreads();                      // meta and the lane tile visible
        if (!REV && rng_state != nullptr) {
            // x += noise * N(0,1), four values per Philox call, keyed by (seed, step, element group)
            const unsigned long long seed = rng_state[0], step = rng_state[1];
            const int nvalid = (a.B - row0 < ROWS ? a.B - row0 : ROWS) * a.d;
            for (int q = tid; 4 * q < nvalid; q += nthreads) {
                float nz[4];
                philox_normal4(seed, step, (unsigned)(((size_t)row0 * a.d) / 4 + (size_t)q), nz);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = 4 * q + e;
                    if (i < nvalid) { const int r = fdiv(i, inv_d); XS[r * a.xld + (i - r * a.d)] += noise * nz[e]; }
                }
            }
            __syncthreads();
            if (x_noisy != nullptr) store_tile(x_noisy, XS, a.xld, a.d, row0, a.B, tid, nthreads);   // what the backward pass starts from
        }

        for (int cb = 0; cb < n_chain; ++cb) {
            const int bi = REV ? n_chain - 1 - cb : cb;       // the inverse walks the chain last block first
            const GBlock blk = HINT_CB(bi);
            const float* perm = (const float*)blk.perm;
            float* tape = (float*)blk.tape;
            float* actA1 = (float*)blk.actA1;
            const bool train = !REV && actA1 != nullptr;
            STAMP((cb * a.n_groups + a.n_groups - a.n_sub) * 16 + 9)
            if (!REV && perm != nullptr) {
                // fused fixed inter-block permutation (power_hint_8.py:59-62): x' = x W
                f32x4 pacc[PERM_TQ];
                if (a.perm_lds > 0) perm_mfma<false>(pacc, XS, a.xld, (const LDS_AS float*)(ptab + bi * pdd), a.d, wave, a.nw, lane);
                else perm_mfma<false>(pacc, XS, a.xld, (const GLOBAL_AS float*)perm, a.d, wave, a.nw, lane);
                STAMP((cb * a.n_groups + a.n_groups - a.n_sub) * 16 + 10)
                perm_store(pacc, XO, a.xld, a.d, wave, a.nw, lane);
                xcur = xflip - xcur;
                __syncthreads();
                STAMP((cb * a.n_groups + a.n_groups - a.n_sub) * 16 + 11)
                if (tape != nullptr)      // the permuted input is what the backward pass starts from
                    store_tile(tape + (size_t)(a.n_levels - 1) * a.B * a.d, XS, a.xld, a.d, row0, a.B, tid, nthreads);
            } else if (!REV && cb > 0 && tape != nullptr) {
                // inner block of a chain without a permutation: its input exists nowhere else
                store_tile(tape + (size_t)(a.n_levels - 1) * a.B * a.d, XS, a.xld, a.d, row0, a.B, tid, nthreads);
            }
            STAMP((cb * a.n_groups + a.n_groups - a.n_sub) * 16 + 12)
            {   // the subtree groups' parameters and the block's thin-layer vectors -> LDS (every wavefront re-reads them for its units,
                // and all workgroups asking L2 for the same few lines at once is what made them slow)
                const bool thin_blk = a.thin_lds > 0 && a.thin_grp == 0;
                const bool thin_late = !REV && a.n_sub > 0;          // (behind the subtree phase: below)
                block_stage(a, blk.packed, lds, true, thin_blk && !thin_late ? a.thin_floats >> 2 : 0, tid, nthreads);
            }
            STAMP((cb * a.n_groups + a.n_groups - a.n_sub) * 16 + 13)
            if ((a.thin_lds > 0 && a.thin_grp == 0) || a.n_sub > 0) __syncthreads();
            STAMP((cb * a.n_groups + a.n_groups - a.n_sub) * 16 + 14)
            PhaseCtx pc;
            pc.packed = blk.packed;
            pc.thin_l = a.thin_lds > 0 ? (const LDS_AS float*)thinb : nullptr;
            pc.thin_g = blk.packed + a.thin_off;
            pc.recs = a.recs; pc.abuf = (LDS_AS float*)abuf; pc.obuf = nullptr; pc.slab = nullptr;      // (obuf, slab: per group)
            pc.out_thin = nullptr; pc.out_main = train ? blk.actA1 + a.a2_off : nullptr;      // (out_thin: per group - lean groups keep no a1)
            pc.bits_a1 = train ? (GLOBAL_AS uint8_t*)(blk.actA1 + a.bits_off) + (size_t)(row0 >> 4) * (a.WT >> 4) * 64 : nullptr;
            pc.bits_a2 = train ? pc.bits_a1 + a.bits_stride : nullptr;
            pc.cs = (const LDS_AS float*)cs; pc.gst = nullptr;
            pc.xld = a.xld; pc.cld = a.cld; pc.gld = 0; pc.WT = a.WT; pc.row0 = row0;
            pc.store = train; pc.fly = false;
            pc.scratch = nullptr; pc.tw = nullptr; pc.first_tile = false;

            // the weight stream of a group's first row starts one phase early: before the block's first group
            // here, for the later ones right behind the rows of the group before (i.e. across its coupling phase)
            f32x4 ring[RING][NEL];
            RecCarry rcar; rcar.held = -1; rcar.v = 0;
            const int ngen = a.n_groups - a.n_sub;        // the general groups: [n_sub, n_groups)
            auto pf = [&](int pos) {                      // consumer `pos` of this block's sequence [head | general groups in this direction's order], or of the next block's
                const GLOBAL_AS float* pk = blk.packed;
                if (pos > ngen) {
                    if (cb + 1 >= n_chain) return;
                    pos -= ngen + 1;
                    const GBlock nb = HINT_CB(REV ? bi - 1 : bi + 1);
                    pk = nb.packed;
                    // (with the next block's head: the d x d matrix in front of it, when it is read from global memory)
                    if (!REV && pos == 0 && a.perm_lds == 0 && nb.perm != nullptr) prefetch_range(nb.perm, 0, (pdd * 4 + 127) >> 7, lds + a.sink_lds, lane);
                }
                prefetch_consumer<!REV>(a, T, pk, pos, lds + a.sink_lds, lane);
            };
            if (!REV && a.n_sub > 0) {
                // the deepest levels: one subtree per wavefront, no workgroup barrier until they rejoin (hint_sub.hpp)
                STAMP((cb * a.n_groups + ngen) * 16 + 0)
                sub_apply<false>(a, T, lds, blk, XS, train, row0, wave, lane, (cb * a.n_groups + ngen) * 16);
                if (a.sink_lds > 0 && wave == 0) pf(HINT_PF_DIST);       // (the first wavefronts are through their subtrees 2-4 k cycles before the last)
                if (a.thin_lds > 0 && a.thin_grp == 0) block_stage(a, blk.packed, lds, false, a.thin_floats >> 2, tid, nthreads);
                STAMP((cb * a.n_groups + ngen) * 16 + 1)
                lds_barrier();
                STAMP((cb * a.n_groups + ngen) * 16 + 2)
                if (tid < ROWS) { float t = 0.f; for (int w = 0; w < a.nw; ++w) t += (lds + a.sub_misc)[w * ROWS + tid]; jac[tid] += t; }
            }
            {
                const GroupU g0 = load_group(T.groups + (REV ? a.n_groups - 1 : a.n_sub));
                const LDS_AS int32_t* rng0 = T.rng + g0.rng_begin;
                rows_begin<K_FWD>(pc, ring, rcar, g0.row_begin + lds_i32(rng0 + wave), g0.row_begin + lds_i32(rng0 + wave + 1), lane);
            }
            for (int gi = 0; gi < ngen; ++gi) {
                const GroupU g = load_group(T.groups + (REV ? a.n_groups - 1 - gi : a.n_sub + gi));
                const LDS_AS int32_t* rng = T.rng + g.rng_begin;
                pc.xs = (const LDS_AS float*)(XS);
                pc.wcol0 = g.wcol0;
                float* obuf = abuf + g.ntiles * 256;
                // (output tiles through LDS and out in whole lines by the element-wise phase's idle wavefronts - what the planner's "staged"
                //  meant for the forward too until round 4 - is slower than the rows' own non-temporal stores: cfg 5 forward 300 -> 280 us,
                //  d = 100 632 -> 604, h = 512 913 -> 923; off at compile time, the code path costs 10 us by being there)
                const bool gstaged = g.staged && HINT_FWD_STAGE;
                float* slab = obuf + (gstaged ? g.ntiles * 256 : 0);
                pc.obuf = gstaged ? (LDS_AS float*)obuf : nullptr; pc.slab = (LDS_AS float*)slab;
                pc.out_thin = (train && !g.nothin) ? (GLOBAL_AS float*)blk.actA1 : nullptr;       // (lean and lean-wide groups keep no a1)
                const int sid = (cb * a.n_groups + gi) * 16;
                (void)sid;
                pc.sid = sid;
                STAMP(sid + 0)
                bool thin_staged = a.thin_lds > 0;
                if (a.thin_grp > 0) {      // (the group's thin vectors: the block's are too many for the LDS)
                    pc.thin_l = thin_group_stage(thinb, a.thin_grp, blk.packed + a.thin_off, a.thins, g.tile_begin, g.ntiles, a.total_tiles,
                                                 a.thin_floats, tid, nthreads);
                    thin_staged = pc.thin_l != nullptr;
                    if (thin_staged) lds_barrier();
                }
                // ---- P1: first layer of every unit of the group on the vector ALU, the tiles shared out - not for a lean group with its
                //      thin vectors in LDS: its rows make their a1 fragments themselves (hint_rows.hpp row_body FLY) ----
                pc.fly = FLYK && g.lean && thin_staged && HINT_FLY_ON;
                if (!pc.fly) {
                    const int t0 = g.tile_begin + lds_i32(rng + a.nw + 1 + wave), t1 = g.tile_begin + lds_i32(rng + a.nw + 2 + wave);
                    if (thin_staged) thin_phase<K_FWD, true>(pc, a.thins, t0, t1, lane);
                    else thin_phase<K_FWD, false>(pc, a.thins, t0, t1, lane);
                    STAMP(sid + 1)
                    lds_barrier();
                }
                STAMP(sid + 2)
                // ---- P2: second layer, third layer partials; the last row hands the weight ring to the wavefront's
                //      first row of the next group (its loads fly across the coupling and thin phases) ----
                {
                    int rnext = -1;
                    if (gi + 1 < ngen) {
                        const GroupU gn = load_group(T.groups + (REV ? a.n_groups - 2 - gi : a.n_sub + gi + 1));
                        const LDS_AS int32_t* rngn = T.rng + gn.rng_begin;
                        const int n0 = lds_i32(rngn + wave), n1 = lds_i32(rngn + wave + 1);
                        if (n0 < n1) rnext = gn.row_begin + n0;
                    }
                    rows_run<K_FWD, FLYK>(pc, ring, rcar, g.row_begin + lds_i32(rng + wave), g.row_begin + lds_i32(rng + wave + 1), rnext, lane);
                }
                STAMP(sid + 15)
                STAMP(sid + 3)
                lds_barrier();
                STAMP(sid + 4)
                if (a.sink_lds > 0 && wave == a.nw - 1) {       // (L2 warm-up two consumers ahead: hint_device.hpp prefetch_consumer)
                    pf(gi + 1 + HINT_PF_DIST);
                    if (gi == 0 && (REV || a.n_sub == 0)) pf(HINT_PF_DIST);        // (the head has no phase of its own here)
                    if (gi == 0) for (int q = 2; q < HINT_PF_DIST; ++q) pf(q);
                }
                // ---- P3: element-wise affine coupling + log-det partial sums (hint.py:79-83) on the first wavefront(s):
                //      16 rows x nsub lanes; the others meanwhile send both hidden activations of the group to the
                //      tape (training), out of LDS, whole lines per batch row ----
                const int nsub = g.ent_cnt <= 4 ? 4 : 16;
                const int ncpl = ROWS * nsub;                     // threads of the coupling
                if (train && gstaged) {
                    const int soff = nthreads > ncpl ? ncpl : 0;   // (a workgroup of one coupling's size does both in turn)
                    if (tid >= soff) {
                        if (!g.nothin) stream_tiles(actA1, abuf, g.ntiles, g.wcol0, a.WT, row0, tid - soff, nthreads - soff);
                        stream_tiles(actA1 + a.a2_off, obuf, g.ntiles, g.wcol0, a.WT, row0, tid - soff, nthreads - soff);
                    }
                }
                if (FLYK && train && pc.fly) {
                    // the sign bytes of the group's a1 tiles, left in its a1 region by the rows: 64 bytes per tile, consecutive tiles
                    const int soff = nthreads > ncpl ? ncpl : 0;
                    GLOBAL_AS i32x4* dst = (GLOBAL_AS i32x4*)(pc.bits_a1 + (size_t)(g.wcol0 >> 4) * 64);
                    if (tid >= soff)
                        for (int i = tid - soff; i < g.ntiles * 4; i += nthreads - soff) dst[i] = ((const LDS_AS i32x4*)abuf)[i];
                }
                // (round 5: a group with 17 .. 24 transformed lanes - every general level of MINIBOONE's tree - splits them over two
                //  sets of wavefronts that work at the same time: entries 0 .. 15 on wavefronts 0-3 as before, entries 16 .. on wavefronts
                //  4-5, eight lanes per batch row, with log-det sums of their own (jac2); the second pass of the one set cost 0.4-0.9 k
                //  cycles per group while four wavefronts waited.  The last wavefront keeps out: it issues the L2 warm-up.)
                // (only in the instance without the lean-row code: the d = 100 plans' instance lost 1 % by the code's presence)
                const bool halves = !FLYK && nsub == 16 && g.ent_cnt > 16 && g.ent_cnt <= 24 && nthreads >= 6 * 64;
                if (tid < ncpl || (halves && tid < ncpl + 128)) {
                    const bool second = halves && tid >= ncpl;
                    const int t2 = tid - ncpl;
                    const int ns = second ? 8 : nsub;
                    const int sub = second ? (t2 & 7) : (tid & (nsub - 1));
                    const int row = second ? (t2 >> 3) : (nsub == 4 ? tid >> 2 : tid >> 4);
                    const int e0 = second ? 16 + sub : sub, e1 = (halves && !second) ? 16 : g.ent_cnt;
                    float part = 0.f;
                    for (int e = e0; e < e1; e += ns) {
                        const LDS_AS int32_t* ep = (const LDS_AS int32_t*)(T.ents + g.ent_begin + e);
                        const unsigned w0 = (unsigned)ep[0], w1 = (unsigned)ep[1];
                        const int xcol = (int)(w0 & 0xffffu), sl_ns = (int)(w1 & 0xffffu), sl_nt = (int)(w1 >> 16);
                        const int stride = 64 * (int)(w0 >> 16);             // floats per slice slab: 16 rows x pad4(r)
                        const int s_off = ep[2], t_off = ep[3];
                        float s = 0.f, t = 0.f;
                        for (int sl = 0; sl < sl_ns; ++sl) s += slab[s_off + sl * stride + row * 4];
                        for (int sl = 0; sl < sl_nt; ++sl) t += slab[t_off + sl * stride + row * 4];
                        const float aa = a.alpha * atanf(s);
                        float* px = XS + row * a.xld + xcol;
                        // training: s goes to the tape ([n_levels + level][B][d], indexed by the lane it
                        // scales): the backward pass needs no third-layer recompute
                        if (!REV && tape != nullptr && row0 + row < a.B)
                            tape[((size_t)(a.n_levels + g.level) * a.B + row0 + row) * a.d + xcol] = s;
                        if (!REV) { *px = expf(aa) * (*px) + t; part += aa; }
                        else      { *px = ((*px) - t) * __builtin_amdgcn_rcpf(expf(aa)); part -= aa; }      // (v_rcp_f32, 1 ulp, instead of a ten-instruction division)
                    }
                    // deterministic butterfly over the adjacent lanes that share a batch row
                    if (ns == 16) part += __shfl_xor(part, 8, 16);
                    if (ns >= 8) part += __shfl_xor(part, 4, 16);
                    part += __shfl_xor(part, 2, 16);
                    part += __shfl_xor(part, 1, 16);
                    if (sub == 0) (second ? jac2 : jac)[row] += part;
                }
                STAMP(sid + 5)
                lds_barrier();
                STAMP(sid + 6)
                // training: keep the lane tile as it stands after each level except the root's, so that
                // the backward pass sees bit-identical subnet inputs (tape[level][B][d])
                if (!REV && tape != nullptr && g.level_last && g.level < a.n_levels - 1)
                    store_tile(tape + (size_t)g.level * a.B * a.d, XS, a.xld, a.d, row0, a.B, tid, nthreads);
            }
            if (REV && a.n_sub > 0) {
                sub_apply<true>(a, T, lds, blk, XS, false, row0, wave, lane, (cb * a.n_groups + ngen) * 16);
                lds_barrier();
                if (tid < ROWS) { float t = 0.f; for (int w = 0; w < a.nw; ++w) t += (lds + a.sub_misc)[w * ROWS + tid]; jac[tid] += t; }
            }
            if (REV && perm != nullptr) {     // inverse of the fused permutation: x = x' W^T
                f32x4 pacc[PERM_TQ];
                if (a.perm_lds > 0) perm_mfma<true>(pacc, XS, a.xld, (const LDS_AS float*)(ptab + bi * pdd), a.d, wave, a.nw, lane);
                else perm_mfma<true>(pacc, XS, a.xld, (const GLOBAL_AS float*)perm, a.d, wave, a.nw, lane);
                perm_store(pacc, XO, a.xld, a.d, wave, a.nw, lane);
                xcur = xflip - xcur;
                __syncthreads();
            }
        }
        store_tile(z, XS, a.xld, a.d, row0, a.B, tid, nthreads);
        if (tid < ROWS && row0 + tid < a.B) J[row0 + tid] = (jac[tid] + jac2[tid]) + (J_in != nullptr ? J_in[row0 + tid] : 0.f);
        if (loss_acc != nullptr) {
            // per-workgroup partial sums of the two loss terms (train_unconditional.py:128-129):
            // slot[0] += sum_rows 0.5*|z|^2, slot[1] += sum_rows J_total
            float zz = 0.f;
            const int nvalid = (a.B - row0 < ROWS ? a.B - row0 : ROWS);
            for (int i = tid; i < nvalid * a.d; i += nthreads) { const int r = fdiv(i, inv_d); const float v = XS[r * a.xld + (i - r * a.d)]; zz += v * v; }
            for (int o = 32; o > 0; o >>= 1) zz += __shfl_xor(zz, o, 64);
            if (lane == 0) red[wave] = zz;
            __syncthreads();
            if (tid == 0) {
                float t = 0.f;
                for (int w = 0; w < a.nw; ++w) t += red[w];
                float js = 0.f;
                for (int r = 0; r < nvalid; ++r) js += (jac[r] + jac2[r]) + (J_in != nullptr ? J_in[row0 + r] : 0.f);
                // 64 slots of {sum 0.5|z|^2, sum J}: spreads the atomics of the workgroups
                float* slot = loss_acc + 2 * (blockIdx.x & 63);
                atomicAdd(slot, 0.5f * t);
                atomicAdd(slot + 1, js);
            }
        }
        __syncthreads();
#undef XS
#undef XO
    }
    STAMP_FLUSH(a.stamps)
#undef HINT_CB
}

namespace hint {

hipError_t launch_apply(bool rev, bool fly, const KArgs& a, int lds_bytes, int grid, const ChainBlock& one,
                        const ChainBlock* chain, int n_chain, const float* x, const float* c, float* z, float* J,
                        const float* J_in, float* loss_acc, float noise, const unsigned long long* rng_state,
                        float* x_noisy, hipStream_t stream) {
#define HINT_LAUNCH_APPLY(REV, FLYK, ...) hipLaunchKernelGGL((hint_apply_kernel<REV, FLYK>), dim3(grid), dim3(64 * a.nw), lds_bytes, stream, a, one, chain, n_chain, __VA_ARGS__)
    if (rev) {
        if (fly) HINT_LAUNCH_APPLY(true, true, x, c, z, J, J_in, (float*)nullptr, 0.f, (const unsigned long long*)nullptr, (float*)nullptr);
        else HINT_LAUNCH_APPLY(true, false, x, c, z, J, J_in, (float*)nullptr, 0.f, (const unsigned long long*)nullptr, (float*)nullptr);
    } else {
        if (fly) HINT_LAUNCH_APPLY(false, true, x, c, z, J, J_in, loss_acc, noise, rng_state, x_noisy);
        else HINT_LAUNCH_APPLY(false, false, x, c, z, J, J_in, loss_acc, noise, rng_state, x_noisy);
    }
#undef HINT_LAUNCH_APPLY
    return hipGetLastError();
}

hipError_t set_max_lds_apply(int bytes) {
    hipError_t e = hipFuncSetAttribute((const void*)hint_apply_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)hint_apply_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)hint_apply_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)hint_apply_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    return e;
}

}  // namespace hint
