// Internal device-side plan layout shared by the plan builder (hint_plan.cpp) and the
// kernels (hint_kernels.hip).  Not part of the public ABI (that is include/hint_amd.h).
#pragma once
#include <stdint.h>

namespace hint {

constexpr int ROWS = 16;        // batch rows per workgroup tile = one MFMA M-tile
constexpr int NTHREADS = 512;   // 8 wavefronts of 64: two per SIMD, so one wave's weight loads
constexpr int NWAVES = 8;       //   land while its SIMD partner issues MFMAs
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

// One 16-column output tile of a GEMM stage:  out[16 rows][16 cols] (+)= A[16][K] * Wlog^T.
// 16 bytes: the job lists of a group are staged in LDS and read with one ds_read_b128.
struct GJob {
    int32_t wtile;      // packed offset, in 256-float tiles, of the first k-block this job reads
    int32_t boff;       // (unused by the kernels: biases are staged in LDS per group)
    uint16_t acol;      // first A column (floats) in the stage's LDS input buffer
    uint16_t ocol;      // first output column in the stage's LDS output buffer
    uint8_t nblk;       // number of 16-wide k-blocks
    uint8_t nvalid;     // valid output columns of this tile (others are written as 0)
    uint8_t slab;       // K-split slab the partial result goes to
    uint8_t pad;
};
static_assert(sizeof(GJob) == 16, "GJob must be 16 bytes");

// Small weight-gradient tile done inside the row-parallel backward kernel:
//   g[goff + m*ldg + n] += sum_rows A[row][acol+m] * B[row][bcol+n]   (atomic)
struct OJob {
    int32_t goff;
    uint16_t acol, bcol;
    uint16_t ldg;
    uint8_t mvalid, nvalid;
    int32_t pad;
};
static_assert(sizeof(OJob) == 16, "OJob must be 16 bytes");

struct Ent { int16_t xcol, scol, tcol, pad; };                  // one transformed lane of a group
struct VNode { int16_t off, k, cin, cinp, vcol, pad; };          // what build_v / scatter need

// A group = a set of same-depth nodes processed together by one workgroup pass.
// All *_off fields index the group's job list (16-byte units from jl_begin).
struct DGroup {
    int32_t node_begin, node_end;
    int32_t jl_begin, jl_count;     // this group's job list in the global job array
    int32_t l1_off, l1_cnt, l2_off, l2_cnt, l3_off, l3_cnt;      // GJob stages
    int32_t g2_off, g2_cnt, g1_off, g1_cnt, dv_off, dv_cnt;
    int32_t o3_off, o3_cnt, o1_off, o1_cnt;                      // OJob lists (dW3, dW1)
    int32_t ent_begin, ent_cnt;
    int32_t bmap_begin;             // per activation column: offsets of b1 / b2 (aw entries each)
    int32_t bmap3_begin;            // per s/t column: offset of b3 (sw entries)
    int32_t aw, vw, sw;
    int32_t l3_slabs, dv_slabs;
    int32_t wcol0;
    int32_t level, level_last;
    int32_t pad0, pad1, pad2, pad3;
};
static_assert(sizeof(DGroup) % 16 == 0, "DGroup must be a multiple of 16 bytes");

// dW2 tile job of the weight-gradient kernel: C[m][n] = sum_b G2[b][col+m] * A1[b][col+n]
struct DWJob {
    int32_t col, H;     // workspace column of this (node, net); H = valid extent (h)
    int32_t m0, n0;     // 48x48 output tile origin
    int64_t wofs;       // offset of dW2 in the flat gradient buffer (row stride H)
};

struct KArgs {
    const void* meta;              // [groups | vnodes | ents] contiguous, copied to LDS at kernel start
    const GJob* jobs;              // all groups' job lists (GJob / OJob, 16 bytes each)
    const int32_t* bmap;
    int32_t meta_bytes;            // multiple of 16
    int32_t vnodes_off, ents_off;  // byte offsets inside meta
    int32_t jmax;                  // capacity (jobs) of one LDS job buffer
    int32_t bmax;                  // capacity (floats) of one LDS bias buffer
    int64_t bias_off;              // float offset of the bias region inside the packed buffer
    int32_t n_groups, n_levels;
    int32_t d, dc;
    int32_t xld, cld, ald, vld, sld;   // LDS row strides (floats)
    int32_t s3, sv;                    // slab counts of the st / gv buffers
    int32_t WT;                        // workspace row width (floats)
    float alpha;
    int32_t B;
};

}  // namespace hint
