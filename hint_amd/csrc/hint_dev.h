// Internal device-side plan layout shared by the plan builder (hint_plan.cpp) and the kernels
// (hint_fwd.hip, hint_bwd.hip, hint_wgrad.hip, hint_pack.hip).  Not part of the public ABI (that
// is include/hint_amd.h).
//
// Design in one paragraph (DESIGN.md has the long form).  A workgroup owns a tile of 16 batch rows
// and carries it through every tree level of every block of a flow.  The wide subnet GEMMs (second
// layer forward, g1 backward) are computed TRANSPOSED, out^T[features x 16 rows] = W[features x K] *
// act^T[K x 16 rows], with v_mfma_f32_16x16x4_f32: the weights are the A operand (pre-packed
// fragments streamed from L2), the activations the B operand.  In that orientation the accumulator of
// one layer (lane l holds features 4*(l>>4)+i of row l&15) IS the B operand of the next layer, so
// the thin last layer (forward) / g_v (backward) is multiplied straight out of the accumulator, and
// activations travel between wavefronts as 1 KiB "fragment tiles" in LDS (element [lane*4+i]),
// written and read with one ds_*_b128 per lane.  The thin first layer (K = a few lanes) and g2
// (K = r) are plain FMAs on the vector ALU in the same lane layout.  The unit of work is a ROW: up to
// NTT (three; four in the general kernels) adjacent 16-feature tiles of one unit (one subnet of one node) that share every B fragment;
// the plan deals a group's rows to the wavefronts (balanced per SIMD) as per-wavefront record lists.
#pragma once
#include <stdint.h>

namespace hint {

constexpr int ROWS = 16;        // batch rows per row tile = one MFMA N-tile of the transposed products
constexpr int TILE = 16;        // MFMA 16x16x4 f32 tile edge
#ifndef HINT_MAX_NW
#define HINT_MAX_NW 8
#endif
#ifndef HINT_NTT
#define HINT_NTT 3
#endif
constexpr int MAX_NW = HINT_MAX_NW;   // wavefronts per workgroup (plan-time choice: 4, 8, ...)
constexpr int NTT = HINT_NTT;         // fragment tiles per row (hint_rows.hpp): 3 by default (planner, wave-local kernels); hint_fwd.hip / hint_bwd.hip define 4
constexpr int MAX_RT = 4;       // 16-wide tiles of a unit's output (r <= 64) and of its input (cin <= 64 + dc)
constexpr int MAX_CT = 12;      // 16-wide tiles of a unit's input v = [u | c] (cin <= 192)
constexpr int LEANW_MAX = 28;   // widest thin layer (inputs / outputs) part B still rebuilds instead of reading a1 / g2 (7 k-blocks of four)
constexpr int LV_REGS = 4;      // hint_bwd.hip holds a prefetched [16, d] tile in LV_REGS floats per thread: 16 * d <= LV_REGS * threads (the planner picks the wavefront count for it)

// ---------------------------------------------------------------------------------------
// Packed weights.  The GEMMs read their weight operand from a buffer in MFMA fragment order, zero
// padded in both dimensions: the 16x16 tile (n-tile nt, k-block kb) of a logical matrix Wlog[N][K] is
// 256 consecutive floats, tiles ordered kb-fastest; element [lane*4 + i] is
//   layout 0 ("fragment"):  Wlog[nt*16 + (lane&15)][kb*16 + 4*(lane>>4) + i]
// and MFMA i of a k-block takes component i of the lane's float4 as its A operand (the order in which
// an accumulator tile of the previous layer supplies the B operand).  The thin layers' weights are
// stored for the vector ALU instead:
//   layout 2 ("vector"):    dst[(nt*KV + k)*16 + f] = Wlog[nt*16 + f][k]   (k < K; zero up to Kp = max(4, K); KV = Kp, or Kp + 1
//                            with the layer's bias as vector Kp)
// so that the lane that owns features 4*(l>>4)..+3 reads the four weights of input k with one float4.  The
// vector-layout segments of a block form two contiguous "thin blobs" (forward, backward) at the start of
// the packed buffer, small enough to be staged in LDS once per block.
// ---------------------------------------------------------------------------------------
struct PackSeg {
    int64_t dst;        // float offset of the segment in the packed buffer
    int64_t src;        // float offset into the flat parameter buffer
    int32_t N, K;       // logical (unpadded) extents; tiles = ceil(N/16) x NB
    int32_t NB;         // k-blocks per n-tile (vector layout: Kp)
    int32_t ld;         // row stride of the source tensor
    int32_t trans;      // 0: Wlog[n][k] = P[src + n*ld + k];  1: Wlog[n][k] = P[src + k*ld + n]
    int32_t kmap;       // layout: 0 fragment, 2 vector
    int32_t tile_begin; // index of this segment's first n-tile in the global n-tile list
    int32_t pad;
    int64_t src2;       // vector layout: float offset of a bias tensor stored as vector K of every tile, or -1
};

// Per-block pointers of a launch (one block: passed by value; a chain: a device table).
struct ChainBlock {
    const float* params;
    const float* packed;
    const float* perm;     // [d,d] permutation in front of the block, or nullptr
    float* tape;           // forward tape of this block (or nullptr): [lane tiles L x B x d][s L x B x d] ...
    float* actA1;          // hidden activations a1 [Bp][WT] inside the tape (a2 follows act_stride later); NULL = inference
    float* wsG1;           // workspace: g1 [Bp][WT]; g2 follows act_stride later
    float* wsGST;          // workspace: coupling gradients [Bp][ST] (g_s | g_t columns of every unit)
    float* wsSlab;         // workspace: [splits][param_floats] partial weight gradients (part B)
    float* gparams;        // flat parameter gradient of this block
    // round 5 (hint_chain_set_block_io): blocks that ran as launches of their OWN and are gathered in a chain only for part B -
    // the modules of the conditional two-lane model - carry their own level-0 input and condition; g_add: a [B, d] gradient
    // added to the block's input gradient before it goes back through the block's fused permutation (the second consumer of the
    // block's permuted input: the y lane of the conditional model is also the x lane's condition)
    const float* x_in;     // or nullptr: the chain's x (first block) / the tape's top slice
    const float* c_in;     // or nullptr: the launch's c
    const float* g_add;    // or nullptr
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

// One unit = one subnet (s: even index, t: odd index) of one node.  24 x int32 = 6 x 16 bytes.
struct Unit {
    int32_t w1v, f2, f3, w3v;       // float offset of W1 (vector layout); first packed tile (offset / 256) of W2 [h x h], W3 [r x h]; float offset of W3^T (vector layout)
    int32_t b2, b1, bias1, bias2;   // first packed tile of W2^T [h x h], W1^T [cin x h]; float offsets of b1, b2 (zero padded to 16) in packed
    int32_t bias3, wcol, tile0, gcol;   // b3; column in the [Bp][WT] arrays; first fragment tile inside the group; column in [Bp][ST]
    int32_t NT, KB1, RT, cin;       // tiles of h, k-blocks of cin, tiles of r; subnet input width
    int32_t ku, r, xoff, h;         // lanes among the inputs (the rest is the condition), outputs, first input lane, hidden width
    int32_t sl_off, sl_n, gv_off, lcol; // first L3 slab (floats inside the slab buffer), rows (= slabs); first g_v slab; gcol - group's gcol0
};
static_assert(sizeof(Unit) == 96, "Unit must be 6 x 16 bytes");

// A group = a set of same-depth nodes processed together.  16 x int32.
struct Group {
    int32_t unit_begin, unit_end, ntiles, row_begin;    // row_begin: the group's first row record (both directions)
    int32_t ent_begin, ent_cnt, rng_begin, level;       // rng: int32[nw+1 | nw+1] = the wavefronts' ranges in the group's row-record list and in its thin-record list
    int32_t level_last, gcol0, gcols, lop_begin;        // first / number of [ST] columns; LaneOp[d] of the boundary in front of the group (backward)
    int32_t level_first, tile_begin, wcol0, lean;       // tile_begin: the group's first thin record (both directions); wcol0: column of its first tile in the [Bp][WT] arrays;
                                                        // lean: bit 0 = every unit of the group has 1..4 inputs, <= 4 outputs, no condition: its a1 / g2 tiles are never stored;
                                                        // bit 1 = the group's output tiles (a2 / g1) are staged in LDS and streamed out by the element-wise phase;
                                                        // bit 2 = subtree group (hint_sub.hpp): rng's second half holds the wavefronts' ranges in the group's entry list
                                                        // bit 3 = lean-wide (round 6): not lean, but every unit has at most LEANW_MAX inputs and outputs and no condition - the thin phases
                                                        //         run as for any group, yet a1 / g2 are not stored: part B rebuilds them with ceil(cin / 4) + ceil(r / 4) K = 4 MFMAs per tile
};
static_assert(sizeof(Group) == 64, "Group must be 4 x 16 bytes");

// One transformed lane of a group (coupling): 16 bytes.
struct Ent {
    int16_t xcol, nquad;    // lane column; ceil(r / 4) of the node: a slice's slab is 16 rows x 4*nquad floats
    int16_t sl_ns, sl_nt;   // slices (slabs to add up) of the node's s / t subnet
    int32_t s_off, t_off;   // float offset inside the slab buffer of element (quad j/4, row 0, j%4) of the first slice
};
static_assert(sizeof(Ent) == 16, "Ent must be 16 bytes");

// Backward, per lane column at the boundary in front of a group (slot n_groups: behind the last one):
// what to add from the group before (the g_v partials of the node whose input this lane is) and
// where the coupling gradients of the coming group's node that transforms this lane go.  16 bytes.
struct LaneOp {
    int16_t sc_unit, sc_k;  // s unit (global index) of the node of the previous group that takes this lane as input k; -1: none
    int16_t cp_ls, cp_lt;   // column of g_s / g_t of this lane in the LDS coupling-gradient buffer; cp_ls = -1: lane not transformed
    int16_t cp_gs, cp_gt;   // the same columns in the global [Bp][ST] array
    int32_t pad;            // slot table (what the kernels read): column of the coupling lane | column of the scatter lane << 16 (equal when one lane is both, or
                            // when the slot holds one lane only); the planner's per-lane working table: the boundary's k-th active lane | their number << 16
};
static_assert(sizeof(LaneOp) == 16, "LaneOp must be 16 bytes");

// Weight-gradient job (part B): out[m][n] = sum_b P[b][pcol+m] * Q[b][qcol+n] for one tile of up to
// 48 x 48 outputs of one parameter matrix; bofs >= 0: the job also sums P's columns (bias gradient).
// Lean plans (every unit has 1..4 inputs, no condition, and at most 4 outputs) keep neither a1 nor g2 in HBM: the
// dW2 jobs rebuild their operands from what feeds the thin layers (WSRC_A1R: relu(W1 v + b1) from the level's lanes;
// WSRC_G2R: relu'(a2) (W3^T g_st) from a2 and the coupling gradients) - a few FMAs per element instead of 4 bytes.
enum { WSRC_G1 = 0, WSRC_G2 = 1, WSRC_GST = 2, WSRC_A1 = 3, WSRC_A2 = 4, WSRC_X = 5, WSRC_C = 6, WSRC_A1R = 7, WSRC_G2R = 8 };
struct WJob {
    int32_t psrc, pcol, M, mw;      // operand array, first column, valid outputs (<= 16*mw), 16-wide tiles (1..3)
    int32_t qsrc, qcol, N, nw;      // N may be 0 (bias only)
    int32_t qlevel, ldo;            // WSRC_X: tree level whose input lanes are read; row stride of the output matrix
    int32_t pmax, qmax;             // last readable column of either operand (loads are clamped to it)
    int64_t wofs;                   // float offset of out[0][0] in the flat gradient layout
    int64_t bofs;                   // float offset of the bias gradient's first element, or -1
    // operands rebuilt on the fly (WSRC_A1R / WSRC_G2R): the unit's thin layers in the flat parameter layout
    int32_t r_w1, r_b1;             // W1 [h][cin], b1 [h]
    int32_t r_w3;                   // W3 [r][h]
    int32_t r_cin, r_xoff;          // inputs: lanes xoff .. xoff+cin of level qlevel
    int32_t r_r, r_gcol;            // outputs: columns gcol .. gcol+r of the coupling gradients [Bp][ST]
    int32_t r_h, r_wcol;            // hidden width (row stride of W3); the unit's first column in the [Bp][WT] arrays
    int32_t pad_[7];
};
static_assert(sizeof(WJob) == 128, "WJob must be 128 bytes");

// Row record: everything a wavefront needs to know about one row - up to NTT adjacent fragment tiles
// [tb, tb+ntt) of one unit - in one direction; 16 x int32, read with one scalar load.  The steps of a
// row, every one `ntt` elements wide: n1 main steps (weight tiles base1 + j*n1 + kb), one aux step (bias
// vectors at packed + aux + 16 j, or the forward activation tiles at column ocol + 16 j), n2 tail steps
// (weight tiles base2 + q*n1 + j), n3 steps of the last layer's bias (packed + bias3 + 16 q).  A record
// with the `thin` flag first runs the unit's thin layer (first layer / g2) for all NT tiles of the unit.
struct RowRec {
    int32_t base1, base2;
    int32_t counts;     // n1 | n2 << 8 | n3 << 16 | ntt << 24
    int32_t aux;        // float offset of the row's first bias vector in the packed buffer (forward)
    int32_t ocol;       // column of the row's first tile in the [Bp][WT] arrays
    int32_t tile;       // first fragment tile of the unit inside the group | ceil(W/4) << 16 (W = r forward, cin backward)
    int32_t slab;       // float offset of the row's slab inside the slab buffer
    int32_t bias3;      // float offset of the unit's b3 (zero padded to 16) in the packed buffer
    int32_t thin_w;     // float offset of the unit's thin-layer vectors inside the direction's thin blob
    int32_t thin_b;     // wave-local plans: offset of the OTHER direction's thin vectors of the row's first tile (forward: W3^T, backward: W1 | b1)
    int32_t thin_k;     // forward: cin | ku << 8 | xoff << 16;  backward: r | lcol << 16
    int32_t flags;      // NT | thin << 8 (run the thin layer before this row) | first << 9 (the row holds the unit's tile 0: it stores the thin layer's tiles and adds b3)
                        // | ulast << 10 (the wavefront's last row of the unit) | rowdw << 11 (backward, general kernels: the row computes dW1 | db1 of its
                        //   tiles itself - lean group whose outputs are not staged in LDS - and does not store g1; p1, p2 as in the wave-local plans)
    int32_t wcol;       // column of the unit's tile 0 in the [Bp][WT] arrays
    int32_t tb;         // first tile of the row inside its unit
    int32_t p1, p2;     // wave-local plans, backward: offset of the row's first feature in the workgroup's first-layer gradient slab; cin | xoff << 8 | h << 16
};
static_assert(sizeof(RowRec) == 64, "RowRec must be 64 bytes");

// Thin record: one fragment tile of a unit's thin layer (first layer forward, g2' backward), computed on the
// vector ALU by whichever wavefront the tile is dealt to; 4 x int32, read with a scalar load.
struct ThinRec {
    int32_t voff;       // float offset of the tile's vectors inside the direction's thin blob
    int32_t k;          // forward: cin | ku << 8 | xoff << 16;  backward: r | lcol << 16
    int32_t tile;       // the tile's index among the group's LDS fragment tiles
    int32_t kp;         // weight vectors per tile: max(4, K) (forward: the bias vector follows them)
};

struct KArgs {
    const void* meta;              // [groups | units | tmap | ents | ranges | laneops] contiguous, copied to LDS at kernel start
    const void* lopsc;             // the backward boundaries' SLOT table, LaneOp[(n_groups + 1) * d], in global memory (read when lops_off < 0: too big for LDS; the LDS copy is the same table)
    const void* thins;             // ThinRec[2][total_tiles]: forward records, then backward records, tiles in (group, unit) order
    int32_t total_tiles;
    const void* recs;              // RowRec[2][total_rows]: forward records, then backward records, rows in (group, wavefront, unit) order
    int32_t total_rows;
    int32_t meta_bytes;            // multiple of 16
    int32_t units_off, tmap_off, ents_off, rng_off, lops_off;   // byte offsets inside meta
    int32_t n_groups, n_levels, n_units, nw;
    int32_t d, dc, xld, cld;       // lanes, condition width, LDS row strides of the lane / condition tiles
    int32_t abuf_tiles;            // fragment tiles of the widest group (LDS activation buffer)
    int32_t slab_floats;           // L3 slab buffer (forward) / g_v slab buffer (backward), floats
    int32_t gld;                   // LDS row stride of the coupling-gradient buffer (backward)
    int32_t WT, ST;                // row widths of the activation / coupling-gradient arrays
    int32_t lean;                  // 1: a1 is not kept on the tape and g2 not in the workspace (the a2 / g1 arrays come first)
    int32_t fuse_dw1;              // 1: the backward kernel computes dW1, db1 itself (slab per workgroup) and g1 is not stored
    int32_t tw_floats, pad_tw;     // floats of one such slab
    int64_t thin_slab_off;         // floats from ChainBlock::wsSlab to the first of them
    int64_t a2_off, bits_off;      // floats from a block's a1 array (ChainBlock::actA1) to its a2 array / to the sign bytes
    int64_t bits_stride;           // bytes between the a1 and the a2 sign bytes of a block's tape
    int32_t region_floats;         // LDS floats of the per-group region [group's tiles | staged output tiles | slabs] (this direction)
    int32_t stage_out;             // 1: the rows leave their output tiles in LDS (obuf) and the element-wise phase streams them out; 0: no LDS for that, they store them themselves
    int32_t thin_off, thin_floats; // the direction's thin blob: float offset in the packed buffer, size (multiple of 4)
    int32_t thin_lds;              // float offset in LDS where the kernel stages it per block; 0: read it from global memory
    int32_t thin_grp;              // > 0: the blob is too large for that - the buffer at thin_lds (thin_grp floats) takes one GROUP's vectors at a time
    int32_t perm_lds;              // float offset in LDS of the chain's d x d permutation matrices; 0: read them from global memory
    int64_t act_stride;            // floats between the a1 and a2 (g1 and g2) arrays
    float alpha;
    int32_t B;
    unsigned long long* stamps;    // diagnostic builds (-DHINT_STAMPS): where workgroup 0 leaves its phase stamps; else unused
    // subtree groups (hint_sub.hpp): the block's deepest n_sub groups (Group::lean bit 2) run one subtree per wavefront,
    // wave-local synchronisation only
    int32_t n_sub;                 // 0: none
    int32_t sub_par, sub_par_f4;   // LDS float offset of their staged parameters [forward vectors | backward vectors | biases]; float4 of it
    int32_t sub_pf, sub_pb;        // floats of the first two segments
    int32_t sub_bsrc, sub_bias_src;    // float offsets of the backward vectors / the biases in the packed buffer (the forward vectors start it)
    int32_t sub_slab;              // LDS float offset of their slabs (this direction)
    int32_t sub_misc;              // LDS float offset: forward nw x 16 log-det partials; backward two 256-float scratch tiles per wavefront
    int32_t sub_cols;              // index in the ranges table of the wavefronts' lane bounds: four per wavefront (hint_plan.cpp)
    int32_t packed_tiles, packed_lines;    // a block's packed buffer: 256-float tiles of [thin blobs | fragment tiles]; 128-byte lines with the biases behind them
    int32_t lop_cnt;               // index in the ranges table of the backward boundaries' slot counts (boundary b = in front of group b; n_groups: the tail's)
    int32_t sink_lds;              // float offset in LDS of a 64-float sink for the L2 prefetch (hint_device.hpp prefetch_consumer); 0: no prefetch
    int32_t rowdw_lds;             // backward: float offset in LDS of one scratch tile (256 floats) per wavefront for the rows that compute dW1 | db1 themselves; 0: none do
};

// ---- wave-local plans (hint_wl.hpp) ----
constexpr int WL_PAR_REGS = 6;      // float4 per thread of the next block's staged parameters in flight
constexpr int WL_PAR_REGS2 = 5;     // ... of the row-pair kernels (GAS d = 8 needs five; six cost the forward 5 and the backward 6 more spilled registers)
constexpr int WL_LV = 4;            // floats per lane of a [16, d] tile held in registers: d <= 16

// where the WL kernels keep things in LDS (float offsets) and how the block's small parameters are staged
struct WlArgs {
    int32_t par_f4;         // float4 of one staged parameter buffer: [forward thin blob | backward thin blob | biases]
    int32_t par_bias;       // float offset of the bias region inside it (= size of the two blobs)
    int32_t bias_src;       // float offset of the bias region inside the packed buffer
    int32_t off_par;        // two buffers of 4 * par_f4 floats
    int32_t off_slab;       // two (alternating per group) x nr slab sets of slab_floats
    int32_t slab_floats;
    int32_t off_perm;       // the chain's d x d matrices (when KArgs::perm_lds != 0)
    int32_t off_priv;       // per wavefront: priv_stride floats = nr x priv_tile (+ backward: one scratch tile)
    int32_t priv_stride, priv_tile;
    int32_t off_misc;       // 32 floats shared: g_J of the tiles' rows
    int32_t off_recs;       // the direction's row records (16 ints each, KArgs::total_rows of them), staged once per kernel
    int32_t nr;             // 16-row tiles per workgroup: 1, or 2 (a row pair on one weight stream)
};

}  // namespace hint
