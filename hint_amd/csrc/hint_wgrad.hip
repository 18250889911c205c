// Backward, part B (gfx950 / CDNA4 only): every weight and bias gradient of a block is a reduction
// over the batch of per-row factors that part A (hint_bwd.hip) and the forward tape left in HBM:
//   dW1[f][k] = sum_b g1[b][f] v[b][k]      v = [lanes of the node's level | condition]
//   dW2[m][n] = sum_b g2[b][m] a1[b][n]
//   dW3[j][f] = sum_b g_st[b][j] a2[b][f]
//   db1 = colsum g1, db2 = colsum g2, db3 = colsum g_st
// i.e. GEMMs whose K dimension is the batch.  Lean plans (hint_dev.h) keep neither a1 nor g2 in HBM: their dW2 jobs
// rebuild both operands per 16-row step with ONE extra MFMA per 16-column tile - a1 = relu(v W1^T + b1) and
// g2 = relu'(a2) (g_st W3), K <= 4 each - whose result layout (lane l: rows 4(l>>4)+i of column l&15) is exactly
// the A / B operand layout of the products over the row subsets {4 kq + i}; the inputs are one float of v, one of
// g_st and one word of a2 sign bytes per lane and step.  The OUTPUT is tiled (up to 48x48 per workgroup: 3x3
// MFMA tiles fed by one 12-byte load per operand and k-step) and the batch is split over `splits`
// workgroups and the 8 wavefronts of each.  Block ids are mapped so that all tiles of one batch split
// run on the same XCD (blocks b, b+8, .. share one): the rows of a split are fetched from HBM /
// Infinity Cache once and re-read from that XCD's L2 by the tiles that share them.
// Deterministic: every (tile, split) writes its partial to the split's own slab with plain stores and
// hint_wreduce_kernel adds the slabs in a fixed order - no float atomics anywhere.
#include "hint_device.hpp"
#include "hint_adam.hpp"

using namespace hint;

#ifndef HINT_DW_WAVES
#define HINT_DW_WAVES 8
#endif
constexpr int DW_WAVES = HINT_DW_WAVES;

struct SrcRef { const float* p; int ld; int rows; };

__device__ __forceinline__ SrcRef wsrc(int src, int level, const GBlock& blk, bool top, const float* __restrict__ x,
                                       const float* __restrict__ c, int WT, int ST, int d, int dc, int n_levels, int B, int Bp,
                                       int64_t act_stride, int64_t a2_off) {
    SrcRef r;
    r.rows = Bp;
    switch (src) {
        case WSRC_G1: r.p = (const float*)blk.wsG1; r.ld = WT; break;
        case WSRC_G2: r.p = (const float*)blk.wsG1 + act_stride; r.ld = WT; break;
        case WSRC_GST: r.p = (const float*)blk.wsGST; r.ld = ST; break;
        case WSRC_A1: r.p = (const float*)blk.actA1; r.ld = WT; break;
        case WSRC_A2: r.p = (const float*)blk.actA1 + a2_off; r.ld = WT; break;
        case WSRC_X: {
            const float* tape = (const float*)blk.tape;
            const size_t lvl = (size_t)B * d;
            r.p = level == 0 ? (top ? tape + (size_t)(n_levels - 1) * lvl : x) : tape + (size_t)(level - 1) * lvl;
            r.ld = d; r.rows = B;
            break;
        }
        default: r.p = c; r.ld = dc; r.rows = B; break;
    }
    return r;
}

// SMALL: the job list ends in single-tile jobs that share workgroups (n_small > 0); plans without them (the wave-local ones)
// run the instance that holds none of that code
template <bool SMALL>
__global__ __launch_bounds__(DW_WAVES * 64) void hint_wgrad_kernel(
    const WJob* __restrict__ jobs, int n_jobs, int n_small, int splits, ChainBlock one, const ChainBlock* __restrict__ chain,
    int grid_pb, int cb0, int WT, int ST, int d, int dc, int n_levels, int B, int Bp, int rows_per_wg, int64_t act_stride,
    int64_t a2_off, int64_t bits_a2_off, int64_t param_floats, const float* __restrict__ x, const float* __restrict__ c) {
    __shared__ float red[DW_WAVES][9][64][4];   // 72 KiB
    __shared__ float bred[DW_WAVES][3][16];

    // a chained launch holds grid_pb (a multiple of 8, so that the XCD mapping below holds for every
    // block) workgroups per block of the chain, of which the first n_jobs * splits have work
    // The last n_small jobs of the list are single-tile jobs (deep levels of narrow trees: an 8 x 8 matrix): eight of them share a
    // workgroup, one (job, split) per WAVEFRONT over all rows of the split - no combine, no barrier: a workgroup per such job
    // spends its time in the fixed costs (MINIBOONE d = 43: 12 000 workgroups for 60 MFLOP)
    const int n_big = n_jobs - n_small, items = n_big + (n_small + DW_WAVES - 1) / DW_WAVES;
    // grid_pb > 0: the chain's blocks one after the other; < 0 (jobs sorted longest first: narrow trees): job j of ALL blocks
    // next to each other, so that the order is longest-first over the whole launch
    int cbi, bid;
    if (grid_pb > 0) {
        cbi = (int)blockIdx.x / grid_pb;
        bid = (int)blockIdx.x - cbi * grid_pb;
    } else {
        const int n_chain = (int)gridDim.x / -grid_pb;
        const int q = (int)blockIdx.x >> 3;
        cbi = q % n_chain;
        bid = (q / n_chain) * 8 + ((int)blockIdx.x & 7);
    }
    if (bid >= items * splits) return;
    const GBlock blk = chain_block(chain, one, cbi);
    const bool top = blk.perm != nullptr || cbi + cb0 > 0;        // (cb0: position of the launch's first block in its chain)

    int item, split;
    if ((splits & 7) == 0) {          // XCD-aware: split s lives on XCD s % 8
        const int xcd = bid & 7, t = bid >> 3;
        split = xcd + 8 * (t / items);
        item = t % items;
    } else {
        split = bid / items;
        item = bid % items;
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = rfl(tid >> 6);
    const bool solo = SMALL && item >= n_big;
    const int jidx = solo ? n_big + (item - n_big) * DW_WAVES + wave : item;
    if (jidx >= n_jobs) return;         // (a spare wavefront of the last shared workgroup: that path has no barrier)
    const WJob job = jobs[jidx];
    const int nl = lane & 15, kq = lane >> 4;
    const int ntm = job.mw, ntn = job.nw;
    const int b_begin = split * rows_per_wg;
    const int b_end = min(Bp, b_begin + rows_per_wg);

    f32x4 acc[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = zero4();
    float psum[3] = {0.f, 0.f, 0.f};
    const int step = solo ? 16 : 16 * DW_WAVES;
    const int bb0 = b_begin + (solo ? 0 : wave * 16);
    const bool natural = job.psrc == WSRC_G2R;       // columns in natural order (no 12-byte loads to serve)

    if (natural) {
        // ---- lean dW2: both operands rebuilt from the thin layers' inputs ----
        const float* prm = (const float*)blk.params;
        const SrcRef vs = wsrc(WSRC_X, job.qlevel, blk, top, x, c, WT, ST, d, dc, n_levels, B, Bp, act_stride, a2_off);
        const float* gst = (const float*)blk.wsGST;
        const uint8_t* bits = (const uint8_t*)blk.actA1 + bits_a2_off;           // a2 sign bytes [row tile][tile][64]
        const int ntiles = WT >> 4;
        float w1b[3], b1c[3], w3b[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int fq = job.qcol - job.r_wcol + 16 * t + nl, fp = job.pcol - job.r_wcol + 16 * t + nl;
            w1b[t] = (kq < job.r_cin && fq < job.r_h) ? prm[job.r_w1 + (size_t)fq * job.r_cin + kq] : 0.f;
            b1c[t] = fq < job.r_h ? prm[job.r_b1 + fq] : 0.f;
            w3b[t] = (kq < job.r_r && fp < job.r_h) ? prm[job.r_w3 + (size_t)kq * job.r_h + fp] : 0.f;
        }
        const int vcol = job.r_xoff + min(kq, job.r_cin - 1), gcol = job.r_gcol + min(kq, job.r_r - 1);
        const unsigned bofs = 16 * (nl >> 2) + 4 * kq, bsh = nl & 3;
        struct In { float v, g; unsigned s[3]; };
        auto load_in = [&](int BB) {
            In in;
            in.v = vs.p[(size_t)min(BB + nl, vs.rows - 1) * vs.ld + vcol];
            in.g = gst[(size_t)(BB + nl) * ST + gcol];
            const uint8_t* bp = bits + ((size_t)(BB >> 4) * ntiles + (job.pcol >> 4)) * 64 + bofs;
#pragma unroll
            for (int t = 0; t < 3; ++t) in.s[t] = t < ntm ? *(const unsigned*)(bp + 64 * t) : 0u;
            return in;
        };
        auto mma = [&](const In& in) {
            const float va = kq < job.r_cin ? in.v : 0.f, ga = kq < job.r_r ? in.g : 0.f;
            f32x4 p[3], q[3];
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                if (t < ntn) {
                    f32x4 b = {b1c[t], b1c[t], b1c[t], b1c[t]};
                    q[t] = mfma4(va, w1b[t], b);
                    q[t].x = fmaxf(q[t].x, 0.f); q[t].y = fmaxf(q[t].y, 0.f); q[t].z = fmaxf(q[t].z, 0.f); q[t].w = fmaxf(q[t].w, 0.f);
                }
                if (t < ntm) {
                    p[t] = mfma4(ga, w3b[t], zero4());
                    const unsigned sw = in.s[t] >> bsh;
                    p[t].x = (sw & 1u) ? p[t].x : 0.f; p[t].y = (sw & 0x100u) ? p[t].y : 0.f;
                    p[t].z = (sw & 0x10000u) ? p[t].z : 0.f; p[t].w = (sw & 0x1000000u) ? p[t].w : 0.f;
                    psum[t] += (p[t].x + p[t].y) + (p[t].z + p[t].w);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int tm = 0; tm < 3; ++tm)
                    if (tm < ntm)
#pragma unroll
                        for (int tn = 0; tn < 3; ++tn)
                            if (tn < ntn) acc[tm][tn] = mfma4(p[tm][i], q[tn][i], acc[tm][tn]);
        };
        int bb = bb0;
        if (bb < b_end) {
            In cur = load_in(bb);
            while (true) {
                const int nb = bb + step;
                const bool last = nb >= b_end;
                const In nxt = load_in(last ? bb : nb);
                mma(cur);
                if (last) break;
                cur = nxt; bb = nb;
            }
        }
    } else {
    const SrcRef ps = wsrc(job.psrc, 0, blk, top, x, c, WT, ST, d, dc, n_levels, B, Bp, act_stride, a2_off);
    const SrcRef qs = wsrc(job.qsrc, job.qlevel, blk, top, x, c, WT, ST, d, dc, n_levels, B, Bp, act_stride, a2_off);

    // lane nl holds columns col0 + w*nl + {0..w-1} of its operand (the permutation of columns inside
    // the up-to-48-wide group is undone at write-out).  Full 48-column groups inside the array: one
    // 12-byte load per k-step; otherwise element loads clamped to the operand's last column (products
    // of clamped columns are never written).
    typedef float f32x3u __attribute__((ext_vector_type(3), aligned(4)));
    const int pc0 = job.pcol + ntm * nl, qc0 = job.qcol + ntn * nl;
    const bool pvec = ntm == 3 && job.pcol + 47 <= job.pmax;
    const bool qvec = ntn == 3 && job.qcol + 47 <= job.qmax;
    int pcj[3], qcj[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) { pcj[j] = min(pc0 + j, job.pmax); qcj[j] = min(qc0 + j, max(job.qmax, 0)); }
    const int prow_max = ps.rows - 1, qrow_max = qs.rows - 1;

    f32x3u av[2][4], bv[2][4];
#define DW_LOAD(BUF, BB)                                                                  \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                        \
        const int row_ = (BB) + 4 * i + kq;                                                \
        const size_t pr_ = (size_t)min(row_, prow_max) * ps.ld;                            \
        const size_t qr_ = (size_t)min(row_, qrow_max) * qs.ld;                            \
        if (pvec) av[BUF][i] = *(const f32x3u*)(ps.p + pr_ + pc0);                         \
        else { av[BUF][i].x = ps.p[pr_ + pcj[0]]; av[BUF][i].y = ps.p[pr_ + pcj[1]]; av[BUF][i].z = ps.p[pr_ + pcj[2]]; } \
        if (ntn > 0) {                                                                     \
            if (qvec) bv[BUF][i] = *(const f32x3u*)(qs.p + qr_ + qc0);                     \
            else { bv[BUF][i].x = qs.p[qr_ + qcj[0]]; bv[BUF][i].y = qs.p[qr_ + qcj[1]]; bv[BUF][i].z = qs.p[qr_ + qcj[2]]; } \
        }                                                                                  \
    }
#define DW_MMA(BUF)                                                                       \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                        \
        _Pragma("unroll") for (int tm = 0; tm < 3; ++tm) {                                 \
            psum[tm] += av[BUF][i][tm];                                                    \
            if (tm < ntm)                                                                  \
                _Pragma("unroll") for (int tn = 0; tn < 3; ++tn)                           \
                    if (tn < ntn) acc[tm][tn] = mfma4(av[BUF][i][tm], bv[BUF][i][tn], acc[tm][tn]); \
        }                                                                                  \
    }

    int bb = bb0;
    if (SMALL && solo && ntm == 1) {
        // A single-tile job on one wavefront is a chain of memory latencies (four MFMAs per 16-row block): eight blocks per step,
        // all their loads in flight at once, an accumulator each (acc[3][3] has nine), added up in a fixed order behind the loop.
        // What is left of the split (fewer than eight blocks) runs through the loop below.
        constexpr int NS = 8;
        const int pc = pcj[0], qc = qcj[0];
        float ps8[NS];
#pragma unroll
        for (int t = 0; t < NS; ++t) ps8[t] = 0.f;
        while (bb + 16 * NS <= b_end) {
            float sa[NS][4], sq[NS][4];
#pragma unroll
            for (int t = 0; t < NS; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row_ = bb + 16 * t + 4 * i + kq;
                    sa[t][i] = ps.p[(size_t)min(row_, prow_max) * ps.ld + pc];
                    sq[t][i] = ntn > 0 ? qs.p[(size_t)min(row_, qrow_max) * qs.ld + qc] : 0.f;
                }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int t = 0; t < NS; ++t) {
                    ps8[t] += sa[t][i];
                    acc[t / 3][t % 3] = mfma4(sa[t][i], sq[t][i], acc[t / 3][t % 3]);
                }
            bb += 16 * NS;
        }
#pragma unroll
        for (int t = 1; t < NS; ++t) {
            acc[0][0] += acc[t / 3][t % 3];
            acc[t / 3][t % 3] = zero4();
            ps8[0] += ps8[t];
        }
        psum[0] += ps8[0];
    }
    // double-buffered over 16-row blocks: the loads of block j+1 are in flight during the MFMAs of block j
    if (bb < b_end) {
        DW_LOAD(0, bb)
        while (true) {
            const int nb1 = bb + step;
            const bool last1 = nb1 >= b_end;
            if (!last1) { DW_LOAD(1, nb1) }
            DW_MMA(0)
            if (last1) break;
            const int nb2 = nb1 + step;
            const bool last2 = nb2 >= b_end;
            if (!last2) { DW_LOAD(0, nb2) }
            DW_MMA(1)
            if (last2) break;
            bb = nb2;
        }
    }
#undef DW_LOAD
#undef DW_MMA
    }
    if (solo) {
        // one tile, one wavefront: straight to the split's slab
        float* slab = (float*)blk.wsSlab + (size_t)split * param_floats;
        const int n = nl;
        if (ntn > 0 && n < job.N) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = 4 * kq + i;
                if (m < job.M) slab[job.wofs + (size_t)m * job.ldo + n] = acc[0][0][i];
            }
        }
        if (job.bofs >= 0) {
            float v = psum[0];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (kq == 0 && nl < job.M) slab[job.bofs + nl] = v;
        }
        return;
    }
    // combine the wavefronts (fixed order)
#pragma unroll
    for (int tm = 0; tm < 3; ++tm)
#pragma unroll
        for (int tn = 0; tn < 3; ++tn) *(f32x4*)&red[wave][tm * 3 + tn][lane][0] = acc[tm][tn];
    if (job.bofs >= 0) {
#pragma unroll
        for (int tm = 0; tm < 3; ++tm) {
            float v = psum[tm];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (kq == 0) bred[wave][tm][nl] = v;
        }
    }
    __syncthreads();
    float* slab = (float*)blk.wsSlab + (size_t)split * param_floats;
    for (int idx = tid; idx < 9 * 64; idx += DW_WAVES * 64) {
        const int t = idx >> 6, l = idx & 63;
        const int tm = t / 3, tn = t - 3 * tm;
        if (tm >= ntm || tn >= ntn) continue;
        f32x4 v = *(f32x4*)&red[0][t][l][0];
#pragma unroll
        for (int w = 1; w < DW_WAVES; ++w) v += *(f32x4*)&red[w][t][l][0];
        const int n = natural ? 16 * tn + (l & 15) : ntn * (l & 15) + tn;          // undo the column permutation of the loads
        if (n >= job.N) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = natural ? 16 * tm + 4 * (l >> 4) + i : ntm * (4 * (l >> 4) + i) + tm;
            if (m < job.M) slab[job.wofs + (size_t)m * job.ldo + n] = v[i];
        }
    }
    if (job.bofs >= 0 && tid < 48) {
        const int tm = tid >> 4, l = tid & 15;
        const int m = natural ? 16 * tm + l : ntm * l + tm;
        if (tm < ntm && m < job.M) {
            float v = bred[0][tm][l];
#pragma unroll
            for (int w = 1; w < DW_WAVES; ++w) v += bred[w][tm][l];
            slab[job.bofs + m] = v;
        }
    }
}

// g[i] (+)= sum over the splits' slabs, in split order, for every real parameter element i (the
// padding between tensors stays untouched when accumulating and is cleared otherwise)
__global__ __launch_bounds__(256) void hint_wreduce_kernel(ChainBlock one, const ChainBlock* __restrict__ chain,
                                                           const uint8_t* __restrict__ real, int64_t param_floats,
                                                           int splits, int accumulate, int blocks_pb,
                                                           const int32_t* __restrict__ twmap, int tw_floats,
                                                           int64_t thin_slab_off, int thin_slabs, int thin_blocks, AdamFuse ad) {
    // ad.p != nullptr (hint_chain_backward_adam): the gradient goes straight into the clamp + Adam step of its parameter - the
    // gradient arena is neither read nor written (it stays zero, as hint_adam_step's zero_grads leaves it)
    const int per_block = blocks_pb + thin_blocks;
    const int cbi = (int)blockIdx.x / per_block;
    const int bid = (int)blockIdx.x - cbi * per_block;
    const GBlock blk = chain_block(chain, one, cbi);
    const float* slab = (const float*)blk.wsSlab;
    float* g = (float*)blk.gparams;
    const bool fuse = ad.p != nullptr;
    const int64_t aoff = fuse ? (const float*)blk.params - ad.p : 0;     // the block's slice of the arenas
    float lr_t = 0.f, bc2 = 0.f;
    if (fuse) { lr_t = ad.st[3]; bc2 = ad.st[4]; }
    if (bid >= blocks_pb) {
        // first-layer gradients of a lean plan: one slab per workgroup of the backward kernel.  32 slab elements per
        // block, 8 threads each: thread q adds slabs q, q+8, .. (eight loads in flight), the eight partials are
        // added in order - a fixed summation order whatever the launch
        __shared__ float part[8][32];
        const int tl = (int)threadIdx.x & 31, q = (int)threadIdx.x >> 5;
        const float* tw = slab + thin_slab_off;
        for (int t0 = (bid - blocks_pb) * 32; t0 < tw_floats; t0 += thin_blocks * 32) {
            const int t = t0 + tl;
            float s = 0.f;
            if (t < tw_floats) {
                int w = q;
                for (; w + 56 < thin_slabs; w += 64) {
                    float v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = tw[(size_t)(w + 8 * u) * tw_floats + t];
#pragma unroll
                    for (int u = 0; u < 8; ++u) s += v[u];
                }
                for (; w < thin_slabs; w += 8) s += tw[(size_t)w * tw_floats + t];
            }
            part[q][tl] = s;
            __syncthreads();
            if (q == 0 && t < tw_floats) {
                const int dst = twmap[t];
                if (dst >= 0) {
                    float r = part[0][tl];
#pragma unroll
                    for (int u = 1; u < 8; ++u) r += part[u][tl];
                    if (fuse) adam_update(ad.p[aoff + dst], ad.m[aoff + dst], ad.v[aoff + dst], r, lr_t, ad.b1, ad.b2, bc2, ad.eps, ad.wd,
                                          ad.gscale, ad.gclamp);
                    else g[dst] = accumulate ? g[dst] + r : r;
                }
            }
            __syncthreads();
        }
        return;
    }
    const int64_t n4 = param_floats >> 2;        // param_floats is a multiple of 4 (hint_plan_param_floats)
    for (int64_t i4 = (int64_t)bid * 256 + threadIdx.x; i4 < n4; i4 += (int64_t)blocks_pb * 256) {
        const uchar4 rl = ((const uchar4*)real)[i4];      // 1: from part B's slabs; 2: from the backward kernel's (above); 0: padding
        f32x4 s = zero4();
        for (int sp = 0; sp < splits; ++sp) s += ((const f32x4*)(slab + (size_t)sp * param_floats))[i4];
        if (fuse) {
            f32x4* pp = (f32x4*)(ad.p + aoff) + i4; f32x4* pm = (f32x4*)(ad.m + aoff) + i4; f32x4* pv = (f32x4*)(ad.v + aoff) + i4;
            f32x4 p4 = *pp, m4 = *pm, v4 = *pv;
            const unsigned char r4[4] = {rl.x, rl.y, rl.z, rl.w};
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (r4[j] == 1) {
                    float pj = p4[j], mj = m4[j], vj = v4[j];
                    adam_update(pj, mj, vj, 0.f + s[j], lr_t, ad.b1, ad.b2, bc2, ad.eps, ad.wd, ad.gscale, ad.gclamp);
                    p4[j] = pj; m4[j] = mj; v4[j] = vj;
                }
            if (rl.x != 2 && rl.y != 2 && rl.z != 2 && rl.w != 2) {
                *pp = p4; *pm = m4; *pv = v4;
            } else {                                    // (elements of the other path are not touched here)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (r4[j] == 1) { ((float*)pp)[j] = p4[j]; ((float*)pm)[j] = m4[j]; ((float*)pv)[j] = v4[j]; }
            }
            continue;
        }
        f32x4 o = accumulate ? ((const f32x4*)g)[i4] : zero4();
        if (rl.x == 1) o.x += s.x;
        if (rl.y == 1) o.y += s.y;
        if (rl.z == 1) o.z += s.z;
        if (rl.w == 1) o.w += s.w;
        if (rl.x != 2 && rl.y != 2 && rl.z != 2 && rl.w != 2) {
            ((f32x4*)g)[i4] = o;
        } else {                                        // (elements of the other path are not touched here)
            float* gp = g + 4 * i4;
            if (rl.x != 2) gp[0] = o.x;
            if (rl.y != 2) gp[1] = o.y;
            if (rl.z != 2) gp[2] = o.z;
            if (rl.w != 2) gp[3] = o.w;
        }
    }
}

namespace hint {

hipError_t launch_wgrad(const WJob* jobs, int n_jobs, int n_small, int splits, const ChainBlock& one, const ChainBlock* chain,
                        int n_chain, int cb0, int WT, int ST, int d, int dc, int n_levels, int B, int Bp, int rows_per_wg,
                        int64_t act_stride, int64_t a2_off, int64_t bits_a2_off, int64_t param_floats, const float* x,
                        const float* c, const uint8_t* real, int accumulate, const int32_t* twmap, int tw_floats,
                        int64_t thin_slab_off, int thin_slabs, int num_cu, const AdamFuse* adam, hipStream_t stream) {
    const bool interleave = n_small < 0;       // (flag in the sign: the planner sorted the jobs)
    if (interleave) n_small = -n_small - 1;
    const int used = (n_jobs - n_small + (n_small + DW_WAVES - 1) / DW_WAVES) * splits;
    const int grid_pb = n_chain > 1 ? (used + 7) / 8 * 8 : used;
    if (used > 0) {
        const int gpb = (interleave && n_chain > 1 && (splits & 7) == 0) ? -grid_pb : grid_pb;
        if (n_small > 0)
            hipLaunchKernelGGL(hint_wgrad_kernel<true>, dim3(grid_pb * n_chain), dim3(DW_WAVES * 64), 0, stream, jobs, n_jobs, n_small, splits,
                               one, chain, gpb, cb0, WT, ST, d, dc, n_levels, B, Bp, rows_per_wg, act_stride, a2_off, bits_a2_off,
                               param_floats, x, c);
        else
            hipLaunchKernelGGL(hint_wgrad_kernel<false>, dim3(grid_pb * n_chain), dim3(DW_WAVES * 64), 0, stream, jobs, n_jobs, n_small, splits,
                               one, chain, gpb, cb0, WT, ST, d, dc, n_levels, B, Bp, rows_per_wg, act_stride, a2_off, bits_a2_off,
                               param_floats, x, c);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    int blocks_pb = (int)((param_floats / 4 + 255) / 256);
    const int cap = num_cu * 4 / (n_chain > 0 ? n_chain : 1);
    blocks_pb = blocks_pb < 1 ? 1 : (blocks_pb > cap && cap >= 1 ? cap : blocks_pb);
    const int thin_blocks = twmap != nullptr ? (tw_floats + 31) / 32 : 0;
    hipLaunchKernelGGL(hint_wreduce_kernel, dim3((blocks_pb + thin_blocks) * n_chain), dim3(256), 0, stream, one, chain, real,
                       param_floats, splits, accumulate, blocks_pb, twmap, tw_floats, thin_slab_off, thin_slabs, thin_blocks,
                       adam ? *adam : AdamFuse{});
    return hipGetLastError();
}

}  // namespace hint
