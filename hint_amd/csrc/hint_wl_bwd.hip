// Backward kernel, part A, of the wave-local plans (hint_wl.hpp; gfx950 / CDNA4 only): gradients with respect to
// the lanes, the coupling gradients g_st that part B (hint_wgrad.hip) reduces into dW2 / dW3 / db2 / db3, and the
// first-layer weight gradients dW1 / db1 themselves (per-workgroup slabs, added up by hint_wreduce_kernel).
//
// What autograd derives from /root/reference/hint.py:62-101 when the training loop calls loss.backward()
// (train_unconditional.py:137), per node, root first:
//   g_t = g_l' ; g_a = g_l'*exp(a)*l + g_J ; g_l = g_l'*exp(a) ; g_s = g_a*alpha/(1+s^2)
//   g2 = (W3^T g_st) .* relu'(a2) ; g1 = (W2^T g2) .* relu'(a1) ; g_u += W1^T g1
// From the forward's tape come s, the lane tiles of every level and the a2 sign bytes; relu'(a1) is recomputed from the
// level's lanes with the forward's own expression (bit-identical decisions).  One workgroup carries the gradient tile
// of 16 batch rows through all blocks of the chain, last to first.  Per tree level, root first: every wavefront adds
// the g_v partials of the level before and runs the coupling backward on its own copy of the tiles, runs its rows
// (g2 on the fly, W2^T on the matrix pipe, g_v partial to its slab), then ONE workgroup barrier.
#include "hint_wl.hpp"

using namespace hint;

struct WlLevel { float x[WL_LV], s[WL_LV]; };
// (loads only: a select on the loaded values would be a wait for HBM in the middle of the level)
__device__ __forceinline__ void wl_level_issue(WlLevel& p, const float* __restrict__ xsrc, const float* __restrict__ ssrc,
                                               int d, int row0, int nvalid, int lane) {
    const size_t base = (size_t)row0 * d;
#pragma unroll
    for (int k = 0; k < WL_LV; ++k) {
        const int i = lane + 64 * k;
        const int ic = i < nvalid ? i : 0;
        p.x[k] = xsrc[base + ic];
        p.s[k] = ssrc[base + ic];
    }
}
__device__ __forceinline__ void wl_level_commit(const WlLevel& p, float* xs, float* sb, int ld, int d, int nvalid, int lane) {
    const float inv = frcp(d);
#pragma unroll
    for (int k = 0; k < WL_LV; ++k) {
        const int i = lane + 64 * k;
        if (i < ROWS * d) {
            const int r = fdiv(i, inv), j = i - r * d;
            xs[r * ld + j] = i < nvalid ? p.x[k] : 0.f;
            sb[r * ld + j] = i < nvalid ? p.s[k] : 0.f;
        }
    }
}

template <int NR, bool CH>        // (CH: chained launch or single block - hint_wl_fwd.hip)
__global__ __launch_bounds__(64 * MAX_NW) __attribute__((amdgpu_waves_per_eu(2, 2))) void hint_wl_bwd_kernel(
    KArgs a, WlArgs w, ChainBlock one, const ChainBlock* __restrict__ chain, int n_chain,
    const float* __restrict__ x, const float* __restrict__ g_z, const float* __restrict__ g_J,
    float* __restrict__ g_x, float gz_scale, float gJ_const) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, nthreads = blockDim.x;
    const int lane = tid & 63;
    const int wave = rfl(tid >> 6);
    const float inv_d = frcp(a.d);
    const Tables T = make_tables(a, lds);
    float* par = lds + w.off_par;
    float* slabs = lds + w.off_slab;            // [2][NR][slab_floats]
    float* ptab = lds + w.off_perm;
    float* gj = lds + w.off_misc;               // g_J of the tiles' rows [NR][16]
    // this wavefront's own tiles, per row tile: lanes of the level (as the forward saw them), s of the level, two gradient
    // tiles (a fused permutation ping-pongs), coupling gradients; behind them one fragment tile of scratch
    float* priv = lds + w.off_priv + wave * w.priv_stride;
    const int tl = ROWS * a.xld;
#define XSP(H) (priv + (H) * w.priv_tile)
#define SBP(H) (priv + (H) * w.priv_tile + tl)
#define G0P(H) (priv + (H) * w.priv_tile + 2 * tl)
#define GSTP(H) (priv + (H) * w.priv_tile + 4 * tl)
    float* scratch = priv + NR * w.priv_tile;
    const int par_floats = (4 * w.par_f4 + 255) & ~255;
    const int ntiles = (a.B + ROWS - 1) / ROWS;
    const int ngroups = (ntiles + NR - 1) / NR;
    const int pdd = a.d * a.d;
    const size_t lvl = (size_t)a.B * a.d;
    STAMP_DECL()
    copy_meta(a, lds, tid, nthreads);
    LDS_AS int32_t* lrecs = (LDS_AS int32_t*)(lds + w.off_recs);
    wl_stage_recs(lrecs, (const char*)a.recs + (size_t)a.total_rows * sizeof(RowRec), a.total_rows, tid, nthreads);
    if (a.perm_lds > 0) {
        for (int i = tid; i < n_chain * pdd; i += nthreads) {
            const int cbi = fdiv(i, frcp(pdd));
            const float* pp = CH ? chain[cbi].perm : one.perm;
            ptab[i] = pp != nullptr ? ((const GLOBAL_AS float*)pp)[i - cbi * pdd] : 0.f;
        }
    }
#define HINT_CB(I) chain_block<CH ? 1 : 2>(chain, one, I)
    // the wavefront's tables (hint_wl.hpp), lane = boundary slot b (0 .. n_groups: the boundary in front of group b; n_groups: the
    // one behind group 0): tr0 / tr1 its rows [r0, r1) of group b; tgi = level of group b | active lanes of boundary b << 8
    int tr0 = 0, tr1 = 0, tgi = 0;
    bool tabs_ready = false;
#define LEVEL_SRC(TAPE, TOP, LV) ((LV) == 0 ? ((TOP) ? (TAPE) + (size_t)(a.n_levels - 1) * lvl : x) : (TAPE) + (size_t)((LV) - 1) * lvl)
#define BITS_A2(BLK, R0) ((const GLOBAL_AS uint8_t*)((BLK).actA1 + a.bits_off) + a.bits_stride + (size_t)((R0) >> 4) * (a.WT >> 4) * 64)

    for (int tg = blockIdx.x; tg < ngroups; tg += gridDim.x) {
        int row0[NR], nvalid[NR], rowt[NR];
#pragma unroll
        for (int h = 0; h < NR; ++h) {
            row0[h] = (tg * NR + h) * ROWS;
            const int rows = a.B - row0[h];
            nvalid[h] = (rows < 0 ? 0 : rows < ROWS ? rows : ROWS) * a.d;
            rowt[h] = row0[h] < ntiles * ROWS ? row0[h] : row0[0];    // (a pair's second tile behind the batch reads the first one's tape: all its gradients are zero)
        }
        int gcur = 0;
#define GS(H) (G0P(H) + gcur)
#define GO(H) (G0P(H) + (tl - gcur))
#pragma unroll
        for (int h = 0; h < NR; ++h)
            for (int i = lane; i < ROWS * a.d; i += 64) {
                const int r = fdiv(i, inv_d);
                GS(h)[r * a.xld + (i - r * a.d)] = i < nvalid[h] ? g_z[(size_t)row0[h] * a.d + i] * gz_scale : 0.f;      // (gz_scale: the loss gradient g_z = z / B fused)
            }
        if (tid < ROWS * NR) {
            const int h = tid >> 4, r = tid & 15;
            gj[tid] = ((tg * NR + h) * ROWS + r < a.B) ? (g_J != nullptr ? g_J[(tg * NR + h) * ROWS + r] : gJ_const) : 0.f;
        }
        {
            const GBlock lb = HINT_CB(n_chain - 1);
            const float* tape = (const float*)lb.tape;
            const bool top = lb.perm != nullptr || n_chain > 1;
            WlLevel lp[NR];
#pragma unroll
            for (int h = 0; h < NR; ++h)
                wl_level_issue(lp[h], LEVEL_SRC(tape, top, a.n_levels - 1), tape + (size_t)(2 * a.n_levels - 1) * lvl, a.d, rowt[h], nvalid[h], lane);
            f32x4 pf[NR == 2 ? WL_PAR_REGS2 : WL_PAR_REGS];
            wl_par_issue(pf, lb.packed, w, tid, nthreads);
#pragma unroll
            for (int h = 0; h < NR; ++h) wl_level_commit(lp[h], XSP(h), SBP(h), a.xld, a.d, nvalid[h], lane);
            wl_par_commit(pf, par, w, tid, nthreads);
        }
        __syncthreads();
        if (!tabs_ready) {
            const int b = lane <= a.n_groups ? lane : 0;
            const LDS_AS int32_t* gp = (const LDS_AS int32_t*)(T.groups + (b < a.n_groups ? b : 0));
            const int row_begin = gp[3], rngb = gp[6];
            tr0 = row_begin + T.rng[rngb + wave];
            tr1 = row_begin + T.rng[rngb + wave + 1];
            tgi = gp[7] | (T.rng[a.lop_cnt + b] << 8);
            tabs_ready = true;
        }
        int phase = 0;
        WlCarry primed; primed.primed = -1; primed.held = -1;
        f32x4 ring[RING][NEL];

        for (int cb = n_chain - 1; cb >= 0; --cb) {
            const int wi = n_chain - 1 - cb;                           // position in the walk
            const bool has_next = cb > 0;
            const GBlock blk = HINT_CB(cb);
            const GBlock nblk = HINT_CB(has_next ? cb - 1 : cb);       // the block worked on after this one
            const float* perm = (const float*)blk.perm;
            const float* tape = (const float*)blk.tape;
            const bool top = perm != nullptr || cb > 0;
            float* wsGST = (float*)blk.wsGST;
            const int tsel = wi & (a.nw - 1);                          // the wavefront that writes this block's g_st rows
            // the next block's parameters -> the other buffer.  Row pairs: by LDS-DMA (hint_wl.hpp wl_par_dma) - the kernel is at the
            // register limit and the staged copy's 20 registers across the root's k-loops were spills (GAS backward 223 -> 217 us);
            // one row tile: through registers (the DMA measured + 1.6 us there: the first k-step's wait completes it in order)
            constexpr bool PAR_DMA = NR == 2;
            f32x4 pf[PAR_DMA ? 1 : WL_PAR_REGS];
            if constexpr (PAR_DMA) { if (has_next) wl_par_dma(nblk.packed, w, (const LDS_AS float*)(par + ((wi + 1) & 1) * par_floats), wave, a.nw, lane); }
            else wl_par_issue(pf, nblk.packed, w, tid, nthreads);

            WlCtx c;
            c.pk = blk.packed; c.pk_next = nblk.packed;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int hh = h < NR ? h : 0;
                c.bits[h] = BITS_A2(blk, rowt[hh]); c.bits_next[h] = BITS_A2(nblk, rowt[hh]);
                c.xs[h] = (const LDS_AS float*)XSP(hh); c.gst[h] = (const LDS_AS float*)GSTP(hh);
                c.a2[h] = nullptr; c.bits_out[h] = nullptr;
            }
            c.lrecs = lrecs;
            c.par = (const LDS_AS float*)(par + (wi & 1) * par_floats);
            c.scratch = (LDS_AS float*)scratch;
            c.tw = blk.wsSlab + a.thin_slab_off + (size_t)blockIdx.x * a.tw_floats;
            c.xld = a.xld; c.gld = a.gld; c.WT = a.WT; c.slab_h = w.slab_floats; c.train = true; c.first_tile = tg == (int)blockIdx.x;

            for (int gi = a.n_groups; gi >= 0; --gi) {
                // gi == n_groups .. 1: the boundary in front of group gi - 1 (root first), then its rows; gi == 0: the
                // boundary BEHIND group 0 (scatter only)
                const int slot = gi > 0 ? gi - 1 : a.n_groups;
                const bool tail_only = gi == 0;
                const int ginfo = wl_lane_get(tgi, slot);
                struct { int level; } g;
                g.level = ginfo & 0xff;
                const int lop0 = slot * a.d;
                const float* slab_prev = slabs + ((phase + 1) & 1) * NR * w.slab_floats;     // the g_v partials of the group just finished
                const int sid = (wi * (a.n_groups + 1) + (a.n_groups - gi)) * 8;
                (void)sid;
                STAMP(sid + 0)
                // ---- scatter of the previous group's g_v + coupling backward of this one, on the wavefront's own tiles ----
                // (the boundary's SLOTS, hint_plan.cpp: a thread adds the finished group's g_v partials onto one lane - `colb` - and forms the
                //  coupling gradients of one transformed lane - `col`: the same lane, or a scatter-only and a coupling-only lane sharing the slot,
                //  since a wavefront runs through both halves whatever its lanes need.  16 rows x nact slots: at d = 6 the three boundaries of a
                //  block have 3, 4 and 2 slots for 3, 5 and 2 active lanes of 6 - each ONE pass of four slots x 16 rows)
                const int nact = ginfo >> 8;
                for (int idx = lane; idx < ROWS * nact; idx += 64) {
                    const int row = idx & 15, kk = idx >> 4;
                    const i32x4 lq = *(const LDS_AS i32x4*)(T.lops + lop0 + kk);        // (the planner keeps a wave-local plan's table in LDS)
                    const unsigned w0 = (unsigned)lq.x, w1 = (unsigned)lq.y, w2 = (unsigned)lq.z;
                    const int col = lq.w & 0xffff, colb = (int)((unsigned)lq.w >> 16);
                    const int sc_unit = (int)(int16_t)(w0 & 0xffffu), sc_k = (int)(w0 >> 16);
                    const int cp_ls = (int)(int16_t)(w1 & 0xffffu), cp_lt = (int)(w1 >> 16);
                    const int cp_gs = (int)(w2 & 0xffffu), cp_gt = (int)(w2 >> 16);
                    float gval[NR];
#pragma unroll
                    for (int h = 0; h < NR; ++h) gval[h] = GS(h)[row * a.xld + col];
                    if (sc_unit >= 0) {
                        float gb[NR];
#pragma unroll
                        for (int h = 0; h < NR; ++h) gb[h] = GS(h)[row * a.xld + colb];
#pragma unroll
                        for (int net = 0; net < 2; ++net) {
                            const LDS_AS int32_t* up = (const LDS_AS int32_t*)(T.units + sc_unit + net);
                            const int sl_n = up[21], gv_off = up[22];
                            const bool more = __builtin_amdgcn_ballot_w64(sl_n > 4) != 0;       // (wave-uniform: a per-lane trip count is an exec-masked loop)
#pragma unroll
                            for (int h = 0; h < NR; ++h) {
                                const float* sp = slab_prev + h * w.slab_floats + gv_off + row * 4 + sc_k;
                                float v[4];             // (the first four slabs in flight together, added in slab order)
#pragma unroll
                                for (int u = 0; u < 4; ++u) v[u] = sp[(u < sl_n ? u : sl_n - 1) * 64];
#pragma unroll
                                for (int u = 0; u < 4; ++u) gb[h] += u < sl_n ? v[u] : 0.f;
                                if (more) for (int sl = 4; sl < sl_n; ++sl) gb[h] += sp[sl * 64];
                            }
                        }
#pragma unroll
                        for (int h = 0; h < NR; ++h) {
                            if (colb == col) gval[h] = gb[h];
                            else GS(h)[row * a.xld + colb] = gb[h];
                        }
                    }
                    if (!tail_only && cp_ls >= 0) {
#pragma unroll
                        for (int h = 0; h < NR; ++h) {
                            const float s = SBP(h)[row * a.xld + col];
                            const float aa = a.alpha * atanf(s);
                            const float ea = expf(aa);
                            const float l = XSP(h)[row * a.xld + col];          // lower input of the node
                            const float ga = gval[h] * ea * l + gj[h * ROWS + row];   // g_a (a feeds both l' and J)
                            const float gsv = ga * a.alpha * __builtin_amdgcn_rcpf(1.f + s * s);     // g_s (v_rcp_f32, 1 ulp: an IEEE division is ten vector instructions)
                            GSTP(h)[row * a.gld + cp_ls] = gsv;
                            GSTP(h)[row * a.gld + cp_lt] = gval[h];             // g_t = g_l'
                            if (wave == tsel && row0[h] < ntiles * ROWS) {
                                float* go = wsGST + (size_t)(row0[h] + row) * a.ST;
                                go[cp_gs] = gsv;
                                go[cp_gt] = gval[h];
                            }
                            gval[h] *= ea;                                      // g_l
                        }
                    }
#pragma unroll
                    for (int h = 0; h < NR; ++h) GS(h)[row * a.xld + col] = gval[h];
                }
                c.sid0 = sid; c.sid = 256;
                STAMP(sid + 1)
                if (tail_only) break;

                // ---- lane tile and s of the level the NEXT boundary needs: global -> registers now, -> LDS behind the rows ----
                WlLevel lp[NR];
                bool lp_pending = false;
                {
                    const bool block_switch = slot == 0;                     // next: root level of the block before
                    int nlevel = a.n_levels - 1;
                    if (!block_switch) nlevel = wl_lane_get(tgi, slot - 1) & 0xff;
                    if (block_switch ? cb > 0 : nlevel != g.level) {
                        const float* ntape = block_switch ? (const float*)nblk.tape : tape;
                        const bool ntop = block_switch ? (nblk.perm != nullptr || cb > 1) : top;
#pragma unroll
                        for (int h = 0; h < NR; ++h)
                            wl_level_issue(lp[h], LEVEL_SRC(ntape, ntop, nlevel), ntape + (size_t)(a.n_levels + nlevel) * lvl, a.d, rowt[h], nvalid[h], lane);
                        lp_pending = true;
                    }
                }
                c.slab = (LDS_AS float*)(slabs + (phase & 1) * NR * w.slab_floats);
                ++phase;
                {
                    int rnext = -1;         // the wavefront's first row of the next group - of the next block's root behind group 0
                    const bool wrap = slot == 0;
                    if (!wrap || has_next) {
                        const int gn = wrap ? a.n_groups - 1 : slot - 1;
                        const int n0 = wl_lane_get(tr0, gn), n1 = wl_lane_get(tr1, gn);
                        if (n0 < n1) rnext = n0;
                    }
                    // (two wavefronts share a SIMD and the hardware serves the older one first: while the rows run the YOUNGER one goes first
                    //  instead - s_setprio; measured: backward -1.2 % at cfg 2, -1.4 % with row pairs (GAS), -3.5 % at 512 rows)
                    if (wave >= 4) __builtin_amdgcn_s_setprio(1);
                    wl_rows<K_BWD, NR>(c, ring, primed, wl_lane_get(tr0, slot), wl_lane_get(tr1, slot), rnext, wrap, lane);
                    __builtin_amdgcn_s_setprio(0);
                }
                STAMP(sid + 2)
                // the next block's parameters -> the other buffer, in front of the first group's barrier (the wavefronts arrive there far
                // apart: the early ones copy while they would wait; hint_wl_fwd.hip); visible behind that barrier
                if (gi == a.n_groups) {
                    if constexpr (PAR_DMA) wl_par_dma_wait();
                    else wl_par_commit(pf, par + ((wi + 1) & 1) * par_floats, w, tid, nthreads);
                }
                if (lp_pending) {
#pragma unroll
                    for (int h = 0; h < NR; ++h) wl_level_commit(lp[h], XSP(h), SBP(h), a.xld, a.d, nvalid[h], lane);
                }
                STAMP(sid + 3)
                lds_barrier();
                STAMP(sid + 4)
            }
            if (blk.g_add != nullptr) {            // the second consumer of the block's (permuted) input: ChainBlock::g_add
#pragma unroll
                for (int h = 0; h < NR; ++h)
                    for (int i = lane; i < nvalid[h]; i += 64) {
                        const int r = fdiv(i, inv_d);
                        GS(h)[r * a.xld + (i - r * a.d)] += blk.g_add[(size_t)row0[h] * a.d + i];
                    }
            }
            if (perm != nullptr) {                 // chain rule through x' = x W:  g_x = g_x' W^T
#pragma unroll
                for (int h = 0; h < NR; ++h) {
                    if (a.perm_lds > 0) wl_perm<true>(GO(h), GS(h), a.xld, (const LDS_AS float*)(ptab + cb * pdd), a.d, lane);
                    else wl_perm<true>(GO(h), GS(h), a.xld, (const GLOBAL_AS float*)perm, a.d, lane);
                }
                gcur = tl - gcur;
            }
        }
        if (wave == 0) {
#pragma unroll
            for (int h = 0; h < NR; ++h)
                for (int i = lane; i < nvalid[h]; i += 64) { const int r = fdiv(i, inv_d); g_x[(size_t)row0[h] * a.d + i] = GS(h)[r * a.xld + (i - r * a.d)]; }
        }
        __syncthreads();
#undef GS
#undef GO
    }
    STAMP_FLUSH(a.stamps)
#undef HINT_CB
#undef LEVEL_SRC
#undef BITS_A2
#undef XSP
#undef SBP
#undef G0P
#undef GSTP
}

namespace hint {

hipError_t launch_wl_bwd(const KArgs& a, const WlArgs& w, int lds_bytes, int grid, const ChainBlock& one,
                         const ChainBlock* chain, int n_chain, const float* x, const float* g_z, const float* g_J,
                         float* g_x, float gz_scale, float gJ_const, hipStream_t stream) {
#define HINT_LAUNCH(NRV, CHV)                                                                                        \
    hipLaunchKernelGGL((hint_wl_bwd_kernel<NRV, CHV>), dim3(grid), dim3(64 * a.nw), lds_bytes, stream, a, w, one, chain, n_chain, x, \
                       g_z, g_J, g_x, gz_scale, gJ_const)
    if (w.nr == 2) { if (chain != nullptr) HINT_LAUNCH(2, true); else HINT_LAUNCH(2, false); }
    else { if (chain != nullptr) HINT_LAUNCH(1, true); else HINT_LAUNCH(1, false); }
#undef HINT_LAUNCH
    return hipGetLastError();
}

hipError_t set_max_lds_wl_bwd(int bytes) {
    const void* fns[4] = {(const void*)hint_wl_bwd_kernel<1, true>, (const void*)hint_wl_bwd_kernel<2, true>,
                          (const void*)hint_wl_bwd_kernel<1, false>, (const void*)hint_wl_bwd_kernel<2, false>};
    for (const void* f : fns) {
        hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace hint
