// Internal device-side plan layout shared by the plan builder (hint_plan.cpp) and the
// kernels (hint_kernels.hip).  Not part of the public ABI (that is include/hint_amd.h).
#pragma once
#include <stdint.h>

namespace hint {

constexpr int ROWS = 16;        // batch rows per workgroup tile = one MFMA M-tile
constexpr int NTHREADS = 256;   // 4 wavefronts of 64
constexpr int NWAVES = 4;
constexpr int TILE = 16;        // MFMA 16x16x4 f32 tile edge

// One tree node, device view.  Column bases index the per-group LDS buffers (acol, vcol,
// scol) and the backward workspace (w*): s-net columns first, t-net columns right after.
struct DNode {
    int32_t off, k, r, h, cin;
    int32_t hp, rp, cinp;       // h, r, cin rounded up to a multiple of 16 (zero padded)
    int32_t acol;               // activation buffers: s at acol, t at acol + hp
    int32_t vcol;               // v = [u | c] buffer
    int32_t scol;               // s/t output buffer: s at scol, t at scol + rp
    int32_t wcol, wvcol, wscol; // same three, but global over the whole block (workspace)
    int64_t p[12];              // parameter offsets [net*6 + {W1,b1,W2,b2,W3,b3}]
};

// A group = a set of same-depth nodes processed together by one workgroup pass.
struct DGroup {
    int32_t node_begin, node_end;
    int32_t jobsH_begin, jobsH_cnt;   // one job per (node, net, 16-wide tile of h)
    int32_t jobsR_begin, jobsR_cnt;   // ... tile of r   (layer 3)
    int32_t jobsC_begin, jobsC_cnt;   // ... tile of cin (backward: dv)
    int32_t ent_begin, ent_cnt;       // one entry per transformed lane of the group
    int32_t aw, vw, sw;               // used widths of the act / v / st buffers
    int32_t wcol0, wvcol0, wscol0;    // workspace column of this group's first node
    int32_t level;                    // 0 = deepest tree level ... n_levels-1 = root
    int32_t level_last;               // 1 if this is the last group of its level (forward order)
};

struct Job { int32_t node, net, tile, pad; };
struct Ent { int32_t xcol, scol, tcol, node; };

// Weight-gradient GEMM job: dW[m][n] = sum_b G[b][gcol+m] * X[b][xcol+n], 48x48 tile.
struct DWJob {
    int32_t gsel, gcol, M;      // gsel: 0=G1 1=G2 2=G3
    int32_t xsel, xcol, N;      // xsel: 0=V  1=A1 2=A2
    int32_t m0, n0;
    int64_t wofs, bofs;         // offsets into the flat gradient buffer (bofs used iff n0 == 0)
};

struct KArgs {
    const DNode* nodes;
    const DGroup* groups;
    const Job* jobs;
    const Ent* ents;
    int32_t n_groups, n_levels;
    int32_t d, dc;
    int32_t xld, cld, ald, vld, sld;   // LDS row strides (floats)
    int32_t WT, VT, ST;                // workspace row widths (floats)
    float alpha;
    int32_t B;
};

}  // namespace hint
