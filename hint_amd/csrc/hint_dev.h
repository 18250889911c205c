// Internal device-side plan layout shared by the plan builder (hint_plan.cpp) and the
// kernels (hint_kernels.hip).  Not part of the public ABI (that is include/hint_amd.h).
#pragma once
#include <stdint.h>

namespace hint {

// Tuning knobs (compile-time; the plan builder and the kernels must agree):
//   HINT_NWAVES  wavefronts per workgroup (8 or 16): with one 16-row tile per CU at B = 4096 the
//                only latency hiding is between the wavefronts of the one resident workgroup
#ifndef HINT_NWAVES
#define HINT_NWAVES 8
#endif
constexpr int ROWS = 16;        // batch rows per workgroup tile = one MFMA M-tile
constexpr int NWAVES = HINT_NWAVES;
constexpr int NTHREADS = 64 * NWAVES;
constexpr int TILE = 16;        // MFMA 16x16x4 f32 tile edge
constexpr int MAX_SLABS = 4;    // K-split partial-sum slabs of the thin (N <= 16) layers

// ---------------------------------------------------------------------------------------
// Packed weights.  Every GEMM of the block reads its B operand from a buffer in MFMA
// fragment order, zero padded in both dimensions: the 16x16 tile (n-tile nt, k-block kb) of a
// logical matrix Wlog[N][K] is 256 consecutive floats, element [lane*4 + i] =
// Wlog[nt*16 + (lane&15)][kb*16 + 4*(lane>>4) + i], tiles ordered kb-fastest.  A wavefront
// fetches one tile with a single fully coalesced 1 KiB global_load_dwordx4.
// ---------------------------------------------------------------------------------------
struct PackSeg {
    int64_t dst;        // float offset of the segment in the packed buffer
    int64_t src0, src1; // float offsets into the flat parameter buffer
    int32_t N, K;       // logical (unpadded) extents; tiles = ceil(N/16) x NB
    int32_t NB;         // k-blocks per n-tile
    int32_t ld;         // row stride of the source tensor
    int32_t mode;       // 0: Wlog[n][k] = P[src0 + n*ld + k]           (forward layers)
                        // 1: Wlog[n][k] = P[src0 + k*ld + n]           (transposed, backward)
                        // 2: Wlog[n][k] = P[src(k/hp) + (k%hp)*ld + n], k%hp < h  (stacked s|t)
    int32_t hp, h;
    int32_t tile_begin; // index of this segment's first n-tile in the global n-tile list
};

// Per-block pointers of a chained launch (hint_chain_*): all blocks of a flow share one plan
// shape, so the kernels loop over the blocks with the lane tile (forward) or the gradient tile
// (backward) staying in LDS between blocks.
struct ChainBlock {
    const float* params;
    const float* packed;
    const float* perm;     // [d,d] permutation in front of the block, or nullptr
    float* tape;           // forward tape of this block (or nullptr)
    float* wsA1;           // hidden activations a1 [Bp][WT] inside the tape (a2 follows act_stride later); NULL = inference
    float* wsG2;
    float* wsT;
    float* gparams;        // flat parameter gradient of this block (part B)
};

// one block's share of a multi-block pack launch (hint_pack_group_*)
struct PackItem {
    const PackSeg* segs;
    const void* ptiles;       // int2[n_tiles]
    const int32_t* bmap;
    const float* params;
    float* packed;
    int64_t bias_off;
    int32_t n_tiles, n_bias;
    int32_t grid_begin, pad;
};

// One job of a GEMM stage: nt (1..3) adjacent 16-column output tiles of one (node, net) that
// share their A operand, over nb consecutive 16-wide k-blocks:
//   out[16 rows][16*nt cols] (+)= A[16][16*nb] * Wlog^T.
// The host cuts every stage into jobs and deals them to the 8 wavefronts so that the four SIMDs
// (wavefronts w and w+4 share one) carry equal MFMA work; a wavefront's jobs form a list of
// 16-byte records read from LDS with one ds_read_b128 each.
struct TJob {
    int32_t wtile;      // packed offset, in 256-float tiles, of (first n-tile, first k-block)
    uint16_t acol;      // A column (floats) of the first k-block in the stage's LDS input
    uint16_t ocol;      // first output column of the first tile in the stage's LDS output
    uint8_t nb;         // k-blocks (0 only for K = 0 jobs: the tiles are bias-only)
    uint8_t nt;         // n-tiles; 0 = nothing to do (the single record of an idle wavefront)
    uint8_t nvalid;     // valid output columns of the LAST tile (the others are full; rest is written as 0)
    uint8_t slab;       // K-split slab the partial result goes to
    uint16_t tstride;   // distance, in tiles, between consecutive n-tiles in the packed buffer
    uint16_t count;     // in the FIRST record of a wavefront's list: number of jobs in the list
};
static_assert(sizeof(TJob) == 16, "TJob must be 16 bytes");
// nt == TJOB_OUTER marks an outer-product tile of a thin weight gradient (dW1, dW3) that rides in
// a GEMM stage's lists of the backward kernel:  T[m][n] = sum_rows A[row][acol+m] * B[row][ocol+n],
// stored at slab[wtile + m*tstride + n] for m <= (nvalid & 15), n <= (nvalid >> 4); nb = 0.
// A record covers a run of adjacent tiles (up to 127) that share one operand: slab = direction | tiles << 1, direction
// 0 = along n (ocol += 16 per tile, A shared), 1 = along m (acol += 16, B shared); nvalid describes the
// LAST tile (the others are full in the direction the record walks).
constexpr int TJOB_OUTER = 0xff;
typedef TJob GJob;      // the per-group lists hold TJob and OJob records, 16 B each

// A stage's lists sit at a fixed stride: wavefront w's list starts at record w*stride (the stage
// descriptor packs offset and stride into one int, see STAGE_DESC).
#define STAGE_DESC(OFF, STRIDE) (((OFF) & 0xffff) | ((STRIDE) << 16))

struct Ent { int16_t xcol, scol, tcol, pad; };                  // one transformed lane of a group

// A group = a set of same-depth nodes processed together by one workgroup pass.  28 int32
// fields = 7 x 16 bytes, read from LDS in one burst (see load_group()).  The *_off fields are stage
// descriptors (STAGE_DESC) into the group's job list (16-byte records from jl_begin): L1, L2, L3
// of the forward pass, g2, g1, dv of the backward pass (the thin weight gradients' outer-product
// tiles ride in g2's and dv's lists), o3_off: the dW3 tiles as a stage of their own (plans without
// LDS for a separate g2 buffer, KArgs.split_o3).  o3_cnt, o1_off, o1_cnt are unused.
// pad: 0 = whole nodes; 1 / 2 = the t / s unit of a node whose two nets run one at a time.
struct DGroup {
    int32_t node_begin, node_end, jl_begin, jl_count;
    int32_t l1_off, l2_off, l3_off, g2_off;
    int32_t g1_off, dv_off, o3_off, o3_cnt;
    int32_t o1_off, o1_cnt, ent_begin, ent_cnt;
    int32_t bmap_begin, bmap3_begin, aw, vw;
    int32_t sw, l3_slabs, dv_slabs, wcol0;
    int32_t level, level_last, vmap_begin, pad;
};
static_assert(sizeof(DGroup) == 112, "DGroup must be 7 x 16 bytes");

// dW2 tile job of the weight-gradient kernel: C[m][n] = sum_b G2[b][col+m] * A1[b][col+n]
// The tile spans mw x nw "virtual" 16x16 MFMA tiles (1..3 each way): lane l of virtual tile j
// holds column m0 + mw*(l&15) + j, so one dwordx3 load per operand and k-step feeds up to three
// MFMA tiles (the permutation of columns inside the 48-wide group is undone at write-out).
struct DWJob {
    int32_t col, H;     // workspace column of this (node, net); H = valid extent (h)
    int32_t m0, n0;     // output tile origin
    int32_t mw, nw;     // virtual tiles (= floats per lane and load) along m and n
    int64_t wofs;       // offset of dW2 in the flat gradient buffer (row stride H)
};

struct KArgs {
    const void* meta;              // [groups | vmap | ents] contiguous, copied to LDS at kernel start
    const GJob* jobs;              // all groups' job lists (GJob / OJob, 16 bytes each)
    const int32_t* bmap;           // per LDS column: compact thin-gradient index of its bias, or -1
    int32_t thin_total;            // floats of one row tile's thin-gradient slab
    int32_t meta_bytes;            // multiple of 16
    int32_t vmap_off, ents_off;    // byte offsets inside meta (vmap: int16 per v column)
    int32_t first[2][4];           // {jl_begin, jl_count, bmap_begin, nbias} of the first group: [0] forward order, [1] reverse
    int32_t jmax;                  // capacity (jobs) of one LDS job buffer
    int32_t bmax;                  // capacity (floats) of one LDS bias buffer
    int64_t bias_off;              // float offset of the bias region inside the packed buffer
    int64_t act_stride;            // floats between the a1 and a2 activation arrays of the tape (ChainBlock.wsA1)
    int32_t n_groups, n_levels;
    int32_t d, dc;
    int32_t xld, cld, ald, vld, sld;   // LDS row strides (floats)
    int32_t max_aw;                    // widest group's activation columns (a1 / a2 buffers)
    int32_t perm_lds;                  // float offset in LDS of the chain's d x d permutation matrices ([n_chain][d][d]); 0: read them from global memory
    int32_t s3, sv;                    // slab counts of the st / gv buffers
    int32_t WT;                        // workspace row width (floats)
    int32_t split_o3;                  // backward: no LDS for a separate g2 buffer -> dW3 tiles run as their own phase
    float alpha;
    int32_t B;
};

}  // namespace hint
