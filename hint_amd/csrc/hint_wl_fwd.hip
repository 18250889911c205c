// Forward / inverse kernel of the wave-local plans (hint_wl.hpp; gfx950 / CDNA4 only): HINT's recursive
// affine-coupling block for trees whose every subnet has 1..4 inputs and at most 4 outputs.
//
// Arithmetic reproduced (reference, read-only): /root/reference/hint.py:62-101
//   per node:  s = mlp_s(u), t = mlp_t(u)                           (hint.py:76-77, :10-13)
//              a = alpha*atan(s), alpha = clamp*0.636               (hint.py:56-60)
//   forward    l' = exp(a)*l + t ;  J += sum a   (children first)   (hint.py:70-80,97-99)
//   inverse    l  = (l' - t)/exp(a); J -= sum a  (root first)       (hint.py:82-88)
//
// One workgroup carries 16 batch rows through all tree levels of all blocks of a flow.  Per level: every
// wavefront runs its rows (first layer on the fly, h x h layer on the matrix pipe, third layer's K-split partial
// to its slab), ONE workgroup barrier, then every wavefront sums the slabs and applies the coupling to its own
// copy of the lane tile.  Tape and packed-weight layouts are those of hint_fwd.hip (hint_plan.cpp).
#include "hint_wl.hpp"

using namespace hint;

// CH: a chained launch (the blocks' pointers come from the device table `chain`) or a single block (`one`, by value): as one
// kernel the select between the two kept the 18 scalar registers of `one` alive through every chained launch.
template <bool REV, int NR, bool CH>
__global__ __launch_bounds__(64 * MAX_NW) __attribute__((amdgpu_waves_per_eu(2, 2))) void hint_wl_apply_kernel(
    KArgs a, WlArgs w, ChainBlock one, const ChainBlock* __restrict__ chain, int n_chain,
    const float* __restrict__ x, float* __restrict__ z, float* __restrict__ J, const float* __restrict__ J_in,
    float* __restrict__ loss_acc, float noise, const unsigned long long* __restrict__ rng_state,
    float* __restrict__ x_noisy) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, nthreads = blockDim.x;
    const int lane = tid & 63, m = lane & 15, kq = lane >> 4;
    const int wave = rfl(tid >> 6);
    const float inv_d = frcp(a.d);
    const Tables T = make_tables(a, lds);
    float* par = lds + w.off_par;               // [2][4 * par_f4]
    float* slabs = lds + w.off_slab;            // [2][NR][slab_floats]
    float* ptab = lds + w.off_perm;
    float* priv = lds + w.off_priv + wave * w.priv_stride;      // this wavefront's lane tiles: per row tile two (a fused permutation ping-pongs)
    const int par_floats = (4 * w.par_f4 + 255) & ~255;
    const int ntiles = (a.B + ROWS - 1) / ROWS;
    const int ngroups = (ntiles + NR - 1) / NR;                  // workgroup-sized pieces of the batch: NR adjacent row tiles
    const int pdd = a.d * a.d;
    const size_t lvl = (size_t)a.B * a.d;
    STAMP_DECL()
    copy_meta(a, lds, tid, nthreads);
    LDS_AS int32_t* lrecs = (LDS_AS int32_t*)(lds + w.off_recs);
    wl_stage_recs(lrecs, a.recs, a.total_rows, tid, nthreads);
    if (a.perm_lds > 0) {
        for (int i = tid; i < n_chain * pdd; i += nthreads) {
            const int cbi = fdiv(i, frcp(pdd));
            const float* pp = CH ? chain[cbi].perm : one.perm;
            ptab[i] = pp != nullptr ? ((const GLOBAL_AS float*)pp)[i - cbi * pdd] : 0.f;
        }
    }
#define HINT_CB(I) chain_block<CH ? 1 : 2>(chain, one, I)
    // the wavefront's tables (hint_wl.hpp): lane e = the e-th group in execution order.  tr0 / tr1: its rows [r0, r1) of the group
    // (absolute record indices); tgi: ent_begin | ent_cnt << 16 | level << 24 | level_last << 30
    int tr0 = 0, tr1 = 0, tgi = 0;
    bool tabs_ready = false;

    for (int tg = blockIdx.x; tg < ngroups; tg += gridDim.x) {
        int row0[NR], nvalid[NR];
#pragma unroll
        for (int h = 0; h < NR; ++h) {
            row0[h] = (tg * NR + h) * ROWS;             // (a pair's second tile may lie behind the batch: no valid rows)
            const int rows = a.B - row0[h];
            nvalid[h] = (rows < 0 ? 0 : rows < ROWS ? rows : ROWS) * a.d;
        }
        int xcur = 0;
        const int xflip = ROWS * a.xld;
#define XS(H) (priv + (H) * w.priv_tile + xcur)
#define XO(H) (priv + (H) * w.priv_tile + (xflip - xcur))
        // every wavefront its own copy of the lane tiles
#pragma unroll
        for (int h = 0; h < NR; ++h)
            for (int i = lane; i < ROWS * a.d; i += 64) {
                const int r = fdiv(i, inv_d);
                XS(h)[r * a.xld + (i - r * a.d)] = i < nvalid[h] ? x[(size_t)row0[h] * a.d + i] : 0.f;
            }
        if (!REV && rng_state != nullptr) {
            // x += noise * N(0,1), four values per Philox call, keyed by (seed, step, element group): the same numbers in
            // every wavefront (and as hint_fwd.hip draws them)
            const unsigned long long seed = rng_state[0], step = rng_state[1];
#pragma unroll
            for (int h = 0; h < NR; ++h) {
                for (int q = lane; 4 * q < nvalid[h]; q += 64) {
                    float nz[4];
                    philox_normal4(seed, step, (unsigned)(((size_t)row0[h] * a.d) / 4 + (size_t)q), nz);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int i = 4 * q + e;
                        if (i < nvalid[h]) { const int r = fdiv(i, inv_d); XS(h)[r * a.xld + (i - r * a.d)] += noise * nz[e]; }
                    }
                }
                if (x_noisy != nullptr && wave == 0)
                    for (int i = lane; i < nvalid[h]; i += 64) { const int r = fdiv(i, inv_d); x_noisy[(size_t)row0[h] * a.d + i] = XS(h)[r * a.xld + (i - r * a.d)]; }
            }
        }
        {   // the first block's thin vectors and biases
            const GBlock b0 = HINT_CB(REV ? n_chain - 1 : 0);
            f32x4 pf[NR == 2 ? WL_PAR_REGS2 : WL_PAR_REGS];
            wl_par_issue(pf, b0.packed, w, tid, nthreads);
            wl_par_commit(pf, par, w, tid, nthreads);
        }
        __syncthreads();                          // meta, permutations, parameters visible
        if (!tabs_ready) {
            const int e = lane < a.n_groups ? lane : 0;
            const LDS_AS int32_t* gp = (const LDS_AS int32_t*)(T.groups + (REV ? a.n_groups - 1 - e : e));
            const int row_begin = gp[3], rngb = gp[6];
            tr0 = row_begin + T.rng[rngb + wave];
            tr1 = row_begin + T.rng[rngb + wave + 1];
            tgi = gp[4] | (gp[5] << 16) | (gp[7] << 24) | (gp[8] << 30);
            tabs_ready = true;
        }
        float jpart[NR];                          // this lane's share of the log-det of batch row m
#pragma unroll
        for (int h = 0; h < NR; ++h) jpart[h] = 0.f;
        int phase = 0;                            // slab set: alternates per group, across blocks
        WlCarry primed; primed.primed = -1; primed.held = -1;
        f32x4 ring[RING][NEL];

        for (int cb = 0; cb < n_chain; ++cb) {
            const int bi = REV ? n_chain - 1 - cb : cb;
            const bool has_next = cb + 1 < n_chain;
            const GBlock blk = HINT_CB(bi);
            const GBlock nblk = HINT_CB(has_next ? (REV ? bi - 1 : bi + 1) : bi);
            const float* perm = (const float*)blk.perm;
            float* tape = (float*)blk.tape;
            const bool train = !REV && blk.actA1 != nullptr;
            const int tsel = cb & (a.nw - 1);          // the wavefront that writes this block's small tape entries
            if (!REV && perm != nullptr) {
                // fused fixed permutation in front of the block (power_hint_8.py:59-62): x' = x W
#pragma unroll
                for (int h = 0; h < NR; ++h) {
                    if (a.perm_lds > 0) wl_perm<false>(XO(h), XS(h), a.xld, (const LDS_AS float*)(ptab + bi * pdd), a.d, lane);
                    else wl_perm<false>(XO(h), XS(h), a.xld, (const GLOBAL_AS float*)perm, a.d, lane);
                }
                xcur = xflip - xcur;
            }
            if (!REV && tape != nullptr && (perm != nullptr || cb > 0)) {
                // the block's input exists nowhere else: the top tape slice is what the backward pass starts from
                // (every wavefront stores its share of its own copy)
#pragma unroll
                for (int h = 0; h < NR; ++h)
                    wl_store_share(tape + (size_t)(a.n_levels - 1) * lvl + (size_t)row0[h] * a.d, XS(h), a.xld, a.d, inv_d, nvalid[h], wave, a.nw, lane);
            }
            // the next block's thin vectors and biases: in flight across this block's first group
            // (fetched behind the first group's rows instead - 20-24 registers fewer across its k-loops - the commit waits for them: +1.4 us)
            // (by LDS-DMA instead - hint_wl.hpp wl_par_dma, what the row-pair backward kernel does: no gain here, GAS + 5 %)
            f32x4 pf[NR == 2 ? WL_PAR_REGS2 : WL_PAR_REGS];
            wl_par_issue(pf, nblk.packed, w, tid, nthreads);

            WlCtx c;
            c.pk = blk.packed; c.pk_next = nblk.packed;
            c.lrecs = lrecs;
            c.par = (const LDS_AS float*)(par + (cb & 1) * par_floats);
            c.scratch = nullptr; c.tw = nullptr;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int hh = h < NR ? h : 0;
                c.bits[h] = nullptr; c.bits_next[h] = nullptr; c.gst[h] = nullptr;
                // (a pair's second tile may lie behind the batch: its rows of the padded arrays do not exist - nothing of it is kept)
                const bool tile_ok = train && h < NR && row0[hh] < ntiles * ROWS;
                c.a2[h] = tile_ok ? blk.actA1 + a.a2_off + (size_t)row0[hh] * a.WT : nullptr;
                c.bits_out[h] = tile_ok ? (GLOBAL_AS uint8_t*)(blk.actA1 + a.bits_off) + a.bits_stride + (size_t)(row0[hh] >> 4) * (a.WT >> 4) * 64 : nullptr;
            }
            c.xld = a.xld; c.gld = 0; c.WT = a.WT; c.slab_h = w.slab_floats; c.train = train; c.first_tile = false;

            for (int gi = 0; gi < a.n_groups; ++gi) {
                const int ginfo = wl_lane_get(tgi, gi);
                struct { int ent_begin, ent_cnt, level, level_last; } g;
                g.ent_begin = ginfo & 0xffff; g.ent_cnt = (ginfo >> 16) & 0xff; g.level = (ginfo >> 24) & 0x3f; g.level_last = (ginfo >> 30) & 1;
                float* slab = slabs + (phase & 1) * NR * w.slab_floats;
                ++phase;
#pragma unroll
                for (int h = 0; h < 2; ++h) c.xs[h] = (const LDS_AS float*)XS(h < NR ? h : 0);
                c.slab = (LDS_AS float*)slab;
                const int sid = (cb * a.n_groups + gi) * 8;
                (void)sid;
                c.sid0 = sid; c.sid = 256;
                STAMP(sid + 0)
                {
                    // the wavefront's first row of the next group - of the next block behind the block's last group
                    int rnext = -1;
                    const bool wrap = gi + 1 == a.n_groups;
                    if (!wrap || has_next) {
                        const int en = wrap ? 0 : gi + 1;
                        const int n0 = wl_lane_get(tr0, en), n1 = wl_lane_get(tr1, en);
                        if (n0 < n1) rnext = n0;
                    }
                    // (two wavefronts share a SIMD and the hardware serves the older one first: while the rows run the YOUNGER one goes first
                    //  instead - s_setprio; measured per instance: forward / inverse of single row tiles -2 %, at 512 rows -5 %, row pairs +3 %: not there)
                    if (NR == 1 && wave >= 4) __builtin_amdgcn_s_setprio(1);
                    wl_rows<K_FWD, NR>(c, ring, primed, wl_lane_get(tr0, gi), wl_lane_get(tr1, gi), rnext, wrap, lane);
                    if (NR == 1) __builtin_amdgcn_s_setprio(0);
                }
                STAMP(sid + 1)
                // ---- coupling (hint.py:79-83) on this wavefront's copy of the lane tiles: lane group kq takes the
                //      transformed lanes kq, kq + 4, ..; batch row m.  What does not depend on the other wavefronts - the
                //      entry's record and the lanes' old values - is fetched in front of the barrier. ----
                const bool has_ent = kq < g.ent_cnt;
                i32x4 ent = *(const LDS_AS i32x4*)(T.ents + g.ent_begin + (has_ent ? kq : 0));
                float xold[NR];
#pragma unroll
                for (int h = 0; h < NR; ++h) xold[h] = XS(h)[m * a.xld + (ent.x & 0xffff)];
                STAMP(sid + 2)
                lds_barrier();
                STAMP(sid + 3)
                // (passes of four entries, wave-uniform count; the lanes without an entry compute on a copy of entry 0 and store nothing:
                //  a loop per lane group - `for (e = kq; e < ent_cnt; e += 4)` - and slab loops with per-lane trip counts were exec-masked
                //  loops with a dozen scalar instructions of control each)
                const int npass = (g.ent_cnt + 3) >> 2;
                for (int ps = 0; ps < npass; ++ps) {
                    const bool act = 4 * ps + kq < g.ent_cnt;
                    const int xcol = ent.x & 0xffff, sl_ns = ent.y & 0xffff, sl_nt = (int)((unsigned)ent.y >> 16);
                    const bool more = __builtin_amdgcn_ballot_w64(sl_ns > 4 || sl_nt > 4) != 0;     // (a unit shared by more than four wavefronts: h > 192)
#pragma unroll
                    for (int h = 0; h < NR; ++h) {
                        const float* sp = slab + h * w.slab_floats + ent.z + m * 4;
                        const float* tp = slab + h * w.slab_floats + ent.w + m * 4;
                        // (the slabs of the wavefronts that share the unit: the first four of either net in flight together, added in slab order)
                        float vs[4], vt[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) { vs[u] = sp[(u < sl_ns ? u : sl_ns - 1) * 64]; vt[u] = tp[(u < sl_nt ? u : sl_nt - 1) * 64]; }
                        float s = 0.f, t = 0.f;
#pragma unroll
                        for (int u = 0; u < 4; ++u) { s += u < sl_ns ? vs[u] : 0.f; t += u < sl_nt ? vt[u] : 0.f; }
                        if (more) {
                            for (int sl = 4; sl < sl_ns; ++sl) s += sp[sl * 64];
                            for (int sl = 4; sl < sl_nt; ++sl) t += tp[sl * 64];
                        }
                        const float aa = a.alpha * atanf(s);
                        // training: s goes to the tape ([n_levels + level][B][d], indexed by the lane it scales)
                        if (act && train && tape != nullptr && wave == tsel && row0[h] + m < a.B)
                            tape[(size_t)(a.n_levels + g.level) * lvl + (size_t)(row0[h] + m) * a.d + xcol] = s;
                        float xn;
                        if (!REV) { xn = expf(aa) * xold[h] + t; jpart[h] += act ? aa : 0.f; }
                        else      { xn = (xold[h] - t) * __builtin_amdgcn_rcpf(expf(aa)); jpart[h] -= act ? aa : 0.f; }      // (v_rcp_f32, 1 ulp, instead of a ten-instruction division)
                        if (act) XS(h)[m * a.xld + xcol] = xn;
                    }
                    if (ps + 1 < npass) {        // (more than four transformed lanes in the group: the next entries)
                        const int en = 4 * (ps + 1) + kq;
                        ent = *(const LDS_AS i32x4*)(T.ents + g.ent_begin + (en < g.ent_cnt ? en : 0));
#pragma unroll
                        for (int h = 0; h < NR; ++h) xold[h] = XS(h)[m * a.xld + (ent.x & 0xffff)];
                    }
                }
                STAMP(sid + 4)
                // the next block's parameters -> the other buffer (loaded long ago: behind the rows, the barrier and the coupling
                // nothing waits here); visible behind the next group's barrier (a one-group block: its own)
                // (in front of the group's barrier instead, where the early wavefronts would copy while they wait: GAS + 3 %)
                if (gi == 0) {
                    wl_par_commit(pf, par + ((cb + 1) & 1) * par_floats, w, tid, nthreads);
                    if (a.n_groups == 1) lds_barrier();
                }
                // training: the lane tile as it stands after each level except the root's (tape[level][B][d])
                if (!REV && tape != nullptr && g.level_last && g.level < a.n_levels - 1) {
#pragma unroll
                    for (int h = 0; h < NR; ++h)
                        wl_store_share(tape + (size_t)g.level * lvl + (size_t)row0[h] * a.d, XS(h), a.xld, a.d, inv_d, nvalid[h], wave, a.nw, lane);
                }
                STAMP(sid + 5)
            }
            if (REV && perm != nullptr) {     // inverse of the fused permutation: x = x' W^T
#pragma unroll
                for (int h = 0; h < NR; ++h) {
                    if (a.perm_lds > 0) wl_perm<true>(XO(h), XS(h), a.xld, (const LDS_AS float*)(ptab + bi * pdd), a.d, lane);
                    else wl_perm<true>(XO(h), XS(h), a.xld, (const GLOBAL_AS float*)perm, a.d, lane);
                }
                xcur = xflip - xcur;
            }
        }
        // ---- results: every wavefront holds them; the first one writes ----
        float zz = 0.f, js = 0.f;
#pragma unroll
        for (int h = 0; h < NR; ++h) {
            const bool rok = row0[h] + m < a.B;
            const float jrow = kq_sum(jpart[h]) + ((J_in != nullptr && rok) ? J_in[row0[h] + m] : 0.f);
            if (wave == 0) {
                for (int i = lane; i < nvalid[h]; i += 64) {
                    const int r = fdiv(i, inv_d);
                    const float v = XS(h)[r * a.xld + (i - r * a.d)];
                    z[(size_t)row0[h] * a.d + i] = v;
                    zz += v * v;
                }
                if (kq == 0 && rok) { J[row0[h] + m] = jrow; js += jrow; }
            }
        }
        if (wave == 0 && loss_acc != nullptr) {
            // partial sums of the two loss terms (train_unconditional.py:128-129): slot[0] += sum 0.5*|z|^2, slot[1] += sum J
            for (int o = 32; o > 0; o >>= 1) { zz += __shfl_xor(zz, o, 64); js += __shfl_xor(js, o, 64); }
            if (lane == 0) {
                float* slot = loss_acc + 2 * (blockIdx.x & 63);      // 64 slots spread the atomics of the workgroups
                atomicAdd(slot, 0.5f * zz);
                atomicAdd(slot + 1, js);
            }
        }
        __syncthreads();          // (the parameter buffers and slabs are re-used by the next row tiles)
#undef XS
#undef XO
    }
    STAMP_FLUSH(a.stamps)
#undef HINT_CB
}

namespace hint {

hipError_t launch_wl_apply(bool rev, const KArgs& a, const WlArgs& w, int lds_bytes, int grid, const ChainBlock& one,
                           const ChainBlock* chain, int n_chain, const float* x, float* z, float* J, const float* J_in,
                           float* loss_acc, float noise, const unsigned long long* rng_state, float* x_noisy,
                           hipStream_t stream) {
    const unsigned long long* no_rng = nullptr;
    float* no_f = nullptr;
#define HINT_LAUNCH(REV, NRV, CHV, LOSS, NOISE, RNG, XN)                                                             \
    hipLaunchKernelGGL((hint_wl_apply_kernel<REV, NRV, CHV>), dim3(grid), dim3(64 * a.nw), lds_bytes, stream, a, w, one, chain, \
                       n_chain, x, z, J, J_in, LOSS, NOISE, RNG, XN)
#define HINT_LAUNCH_CH(REV, NRV, LOSS, NOISE, RNG, XN)                                                               \
    { if (chain != nullptr) HINT_LAUNCH(REV, NRV, true, LOSS, NOISE, RNG, XN); else HINT_LAUNCH(REV, NRV, false, LOSS, NOISE, RNG, XN); }
    if (rev) { if (w.nr == 2) HINT_LAUNCH_CH(true, 2, no_f, 0.f, no_rng, no_f) else HINT_LAUNCH_CH(true, 1, no_f, 0.f, no_rng, no_f) }
    else { if (w.nr == 2) HINT_LAUNCH_CH(false, 2, loss_acc, noise, rng_state, x_noisy) else HINT_LAUNCH_CH(false, 1, loss_acc, noise, rng_state, x_noisy) }
#undef HINT_LAUNCH_CH
#undef HINT_LAUNCH
    return hipGetLastError();
}

hipError_t set_max_lds_wl_apply(int bytes) {
    const void* fns[8] = {(const void*)hint_wl_apply_kernel<false, 1, true>, (const void*)hint_wl_apply_kernel<true, 1, true>,
                          (const void*)hint_wl_apply_kernel<false, 2, true>, (const void*)hint_wl_apply_kernel<true, 2, true>,
                          (const void*)hint_wl_apply_kernel<false, 1, false>, (const void*)hint_wl_apply_kernel<true, 1, false>,
                          (const void*)hint_wl_apply_kernel<false, 2, false>, (const void*)hint_wl_apply_kernel<true, 2, false>};
    for (const void* f : fns) {
        hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace hint
