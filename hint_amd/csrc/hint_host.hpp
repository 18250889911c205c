// Host-side internals shared by the translation units behind the C ABI (include/hint_amd.h): the plan object, the
// kernels' launch wrappers and the helpers every entry point uses.  Not part of the public ABI.
//   hint_plan.cpp     the planner: node list -> static schedule in device memory (hint_plan_create / _check / _destroy)
//   hint_abi.cpp      sizes and layouts, the block-level entry points, re-pack groups, the optimizer, build info
//   hint_chain.cpp    chained launches: the blocks of a flow in one kernel each (hint_chain_*)
//   hint_invgrad.cpp  backward of the inverse direction, level by level on the block kernels
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/hint_amd.h"
#include "hint_dev.h"
#include "hint_adam.hpp"

namespace hint {
hipError_t launch_pack(const PackSeg* segs, const int2* ptiles, int n_tiles, const int32_t* bmap, int n_bias,
                       long bias_off, const float* params, float* packed, hipStream_t stream);
hipError_t launch_pack_many(const PackItem* items, int n_items, int grid, float* zero_buf, int zero_floats,
                            unsigned long long* rng_state, float* opt_state, hipStream_t stream);
hipError_t launch_zero(float* p, long n, int num_cu, hipStream_t stream);
hipError_t launch_inv_lane(int op, float* out, const float* g, const float* a, const float* b, const uint8_t* lower, long n, int d,
                           int num_cu, hipStream_t stream);
hipError_t launch_inv_minus(float* dst, const float* src, long n, int keep, int num_cu, hipStream_t stream);
hipError_t launch_inv_rowmat(const float* x, const float* P, float* y, long n, int d, int num_cu, hipStream_t stream);
hipError_t launch_apply(bool rev, bool fly, const KArgs& a, int lds_bytes, int grid, const ChainBlock& one,
                        const ChainBlock* chain, int n_chain, const float* x, const float* c, float* z, float* J,
                        const float* J_in, float* loss_acc, float noise, const unsigned long long* rng_state,
                        float* x_noisy, hipStream_t stream);
hipError_t launch_bwd(const KArgs& a, int lds_bytes, int grid, const ChainBlock& one, const ChainBlock* chain,
                      int n_chain, const float* x, const float* c, const float* g_z, const float* g_J,
                      float* g_x, float* g_c, float gz_scale, float gJ_const, hipStream_t stream);
hipError_t launch_wgrad(const WJob* jobs, int n_jobs, int n_small, int splits, const ChainBlock& one, const ChainBlock* chain,
                        int n_chain, int cb0, int WT, int ST, int d, int dc, int n_levels, int B, int Bp, int rows_per_wg,
                        int64_t act_stride, int64_t a2_off, int64_t bits_a2_off, int64_t param_floats, const float* x,
                        const float* c, const uint8_t* real, int accumulate, const int32_t* twmap, int tw_floats,
                        int64_t thin_slab_off, int thin_slabs, int num_cu, const AdamFuse* adam, bool wide, hipStream_t stream);
hipError_t launch_bwd_n3(const KArgs& a, int lds_bytes, int grid, const ChainBlock& one, const ChainBlock* chain,
                         int n_chain, const float* x, const float* c, const float* g_z, const float* g_J,
                         float* g_x, float* g_c, float gz_scale, float gJ_const, hipStream_t stream);      // (hint_bwd3.hip: rows of <= 3 tiles)
hipError_t set_max_lds_apply(int bytes);
hipError_t set_max_lds_bwd(int bytes);
hipError_t set_max_lds_bwd_n3(int bytes);
hipError_t launch_bwd_fly(const KArgs& a, int lds_bytes, int grid, const ChainBlock& one, const ChainBlock* chain,
                          int n_chain, const float* x, const float* c, const float* g_z, const float* g_J,
                          float* g_x, float* g_c, float gz_scale, float gJ_const, hipStream_t stream);     // (hint_bwd_fly.hip: plans with lean general groups)
hipError_t set_max_lds_bwd_fly(int bytes);
hipError_t launch_wl_apply(bool rev, const KArgs& a, const WlArgs& w, int lds_bytes, int grid, const ChainBlock& one,
                           const ChainBlock* chain, int n_chain, const float* x, float* z, float* J, const float* J_in,
                           float* loss_acc, float noise, const unsigned long long* rng_state, float* x_noisy,
                           hipStream_t stream);
hipError_t launch_wl_bwd(const KArgs& a, const WlArgs& w, int lds_bytes, int grid, const ChainBlock& one,
                         const ChainBlock* chain, int n_chain, const float* x, const float* g_z, const float* g_J,
                         float* g_x, float gz_scale, float gJ_const, hipStream_t stream);
hipError_t set_max_lds_wl_apply(int bytes);
hipError_t set_max_lds_wl_bwd(int bytes);
hipError_t launch_adam(float* p, float* g, float* m, float* v, long n, float lr_t, float b1, float b2,
                       float inv_sqrt_bc2, float eps, float wd, float gscale, float gclamp, int zero_grads,
                       int num_cu, const float* dev_state, hipStream_t stream);
}  // namespace hint

namespace hint {

int fail(const char* fmt, ...);          // sets the thread's hint_last_error() message; returns 1
std::string& last_error_ref();           // that message (hint_plan_create keeps it across an optional second plan)

#define HIP_TRY(expr)                                                              \
    do {                                                                           \
        hipError_t e_ = (expr);                                                    \
        if (e_ != hipSuccess) return fail("%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

inline int cdiv(int a, int b) { return (a + b - 1) / b; }
inline int pad4(int v) { return (v + 3) & ~3; }

#ifdef HINT_STAMPS
static constexpr int LDS_LIMIT = 160 * 1024 - 32 * 1024 - 512;   // the diagnostic build's static stamp array shares the 160 KiB
#else
static constexpr int LDS_LIMIT = 160 * 1024;
#endif
static constexpr int LDS_ATTR = LDS_LIMIT;
static constexpr int PERM_LDS_MAX = 16 * 1024;   // the chain's permutation matrices ride in LDS up to this size
static constexpr int WS_SLACK = 64;              // floats of slack behind every [Bp][W] array

// Every HINT_* environment variable the LIBRARY reads (the Python host's own are listed in INTEGRATION.md), read ONCE at first use
// (getenv on a launch path is neither cheap nor safe beside a setenv in another thread; a captured hipGraph keeps what was read at
// capture time anyway).  Tests and A/B tools that change the environment inside one process call hint_debug_reload_knobs().
// `non_default`: "HINT_X=v HINT_Y=w" of the variables that are set - appended to hint_build_info(), so that a bench line or a
// counter summary says which kernels its numbers belong to.
struct Knobs {
    bool plan_dump = false;     // HINT_PLAN_DUMP=1   the planner prints groups, rows per wavefront, LDS sizes, kernel decisions, boundary slots
    bool wl = true;             // HINT_WL=0          no wave-local kernels (narrow trees run on the general ones)
    int wl_nr = 0;              // HINT_WL_NR=1|2     wave-local kernels: row tiles per workgroup forced
    bool sub = true;            // HINT_SUB=0         no subtree groups
    int nw = 0;                 // HINT_NW=4|8        wavefronts per workgroup of the general kernels (0: the planner picks)
    bool lean = true;           // HINT_LEAN=0        keep a1 / g2 in HBM (no lean, lean-wide or subtree groups)
    bool leanw = true;          // HINT_LEANW=0       no lean-wide groups (round 6: a1 / g2 of thin layers with 5 .. 28 inputs / outputs rebuilt by part B)
    bool fuse_dw1 = true;       // HINT_FUSE_DW1=0    first-layer weight gradients in part B instead of the backward kernel
    int pf = 1;                 // HINT_PF=0          general kernels: no L2 warm-up of the packed weights (hint_debug_set_prefetch switches it later)
    bool no_bwd_fly = false;    // HINT_NO_BWD_FLY=1  plans with lean general groups on the shared backward kernel
    int dw_splits = 0;          // HINT_DW_SPLITS=n   batch splits of part B
    int dw_small = -1;          // HINT_DW_SMALL=0|1  single-tile part-B jobs one per wavefront of a shared workgroup: never / always (-1: trees with subtree groups)
    int leanw_max = 12;         // HINT_LEANW_MAX=n   widest thin layer (inputs / outputs) of a lean-wide group: 5 .. 28 (default 12: three k-blocks;
                                //                    wider ones cost part B more MFMAs than their a1 / g2 traffic is worth - NOTES.md section 10)
    bool ablation_ok = false;   // HINT_ABLATION_OK=1 lets an ablation build (-DHINT_ABLATE_STORE) run a backward pass (its gradients are wrong: timing only)
    std::string non_default;
};
const Knobs& knobs();

}  // namespace hint

using namespace hint;      // (internal header: the plan object below is a global C-ABI type made of hint:: records)

struct hint_plan {
    int device = -1;
    int d = 0, dc = 0, n_nodes = 0, n_groups = 0, n_levels = 0, n_units = 0, n_wjobs = 0, n_wsmall = 0, wsorted = 0, n_ptiles = 0, nw = 8;   // n_wsmall: single-tile jobs at the end of the job list (hint_wgrad.hip)
    float alpha = 0.f;
    int64_t param_floats = 0, packed_floats = 0;
    int WT = 0, ST = 0;
    int xld = 0, cld = 0, gld = 0, abuf_tiles = 0, slab_fwd = 0, slab_bwd = 0;
    int region_fwd = 0, region_bwd = 0;   // LDS floats of the per-group region [tiles | staged output tiles | slabs]
    int stage_out = 1;
    int max_h = 0;              // widest hidden layer of the block
    int has_fly = 0;            // some general (not subtree) group is lean: the forward kernel's instance whose rows make such groups' first layer themselves
    int lean = 0;               // a1 / g2 are rebuilt by the weight-gradient kernel instead of kept in HBM
    int has_leanw = 0;          // some group is lean-wide (Group::lean bit 3): part B runs the instance that rebuilds wide thin layers
    int fuse_dw1 = 0;           // lean plans with LDS-staged outputs: dW1, db1 come from the backward kernel (per-workgroup slabs), g1 stays on chip
    int tw_floats = 0;          // floats of one such slab
    int32_t* d_twmap = nullptr; // slab index -> offset in the flat parameter layout (or -1)
    int thin_f_off = 0, thin_f_floats = 0, thin_b_off = 0, thin_b_floats = 0, thin_lds_f = 0, thin_lds_b = 0;
    int thin_grp_f = 0, thin_grp_b = 0;     // > 0: LDS takes one group's thin vectors at a time (floats of the largest group's)
    int lds_fwd = 0, lds_bwd = 0;
    // wave-local plans (hint_wl.hpp): narrow trees run on hint_wl_apply_kernel / hint_wl_bwd_kernel
    int wl = 0, wl_nr2 = 0;     // wl_nr2: two 16-row tiles per workgroup on one weight stream fit the LDS as well
    WlArgs wl_f[2]{}, wl_b[2]{};  // [nr - 1]
    // subtree groups (hint_sub.hpp): the deepest n_sub groups run one subtree per wavefront
    int n_sub = 0, sub_pf = 0, sub_pb = 0, sub_pbias = 0, sub_bsrc = 0, sub_cols = 0;
    int sub_slab_f = 0, sub_slab_b = 0;                 // floats of their slabs
    int sub_lds_f[3] = {0, 0, 0}, sub_lds_b[3] = {0, 0, 0};   // LDS float offsets: slabs, staged parameters, misc
    int rowdw_lds = 0;          // backward: LDS float offset of the scratch tiles of the rows that compute dW1 | db1 themselves (0: none)
    int row_ntt = 0;            // tiles of the widest row (<= 3: the backward pass runs on hint_bwd_kernel_n3)
    int num_cu = 256;
    int meta_bytes = 0, units_off = 0, tmap_off = 0, ents_off = 0, rng_off = 0, lops_off = 0, n_bias = 0;
    void* d_meta = nullptr;
    int lop_cnt = 0;                // index in the ranges table of the boundaries' slot counts (n_groups + 1 of them)
    int max_slots = 0;              // slots of the backward's widest boundary (hint_plan_check reports it)
    LaneOp* d_lopsc = nullptr;      // the boundaries' slot table (KArgs::lopsc)
    RowRec* d_recs = nullptr;
    ThinRec* d_thins = nullptr;
    int total_tiles = 0;
    int total_rows = 0;
    int32_t* d_bmap = nullptr;
    uint8_t* d_real = nullptr;
    WJob* d_wjobs = nullptr;
    PackSeg* d_segs = nullptr;
    int2* d_ptiles = nullptr;
    // the same block planned for 4 wavefronts per workgroup (two workgroups per CU: one row tile's serial phases overlap
    // the other's GEMM phases) - used for batches of more row tiles than CUs; owned by this plan; may be absent
    hint_plan* alt4 = nullptr;
    // hint_block_inverse_backward: the node table the plan was made from and, built at the first call, one plan per tree
    // level (its nodes as a forest of depth 0) with the lanes each level transforms
    std::vector<hint_node_desc> nodes;
    float clamp = 0.f;
    std::mutex inv_mu;
    std::vector<hint_plan*> inv_levels;         // deepest level last
    uint8_t* d_inv_lower = nullptr;             // [levels][d]
};

namespace hint {

inline int plan_lds(const hint_plan* P, bool backward, int nr = 1) {
    if (P->wl) return 4 * (backward ? P->wl_b[nr - 1].off_perm : P->wl_f[nr - 1].off_perm);
    return backward ? P->lds_bwd : P->lds_fwd;
}
inline int rows_padded(int B) { return (B + ROWS - 1) / ROWS * ROWS; }

#ifdef HINT_STAMPS
void set_dw_stamps(unsigned long long* p);      // hint_wgrad.hip (diagnostic build)
#endif

// hint_abi.cpp
bool l2_prefetch_on();       // the general kernels' L2 warm-up (HINT_PF / hint_debug_set_prefetch)
const hint_plan* variant(const hint_plan* P, int B);
int wl_nr_for(const hint_plan* Pv, int B);
int64_t tape_act_off(const hint_plan* P, int B);
int64_t act_stride(const hint_plan* P, int B);
int64_t bits_stride(const hint_plan* P, int B);
int grid_for(const hint_plan* P, int B);
int lds_with_perms(const hint_plan* P, int lds_plan, int n_blocks, bool any_perm, KArgs* a, bool backward = false);     // (backward: which slot of hint_debug_last_lds_bytes the launch's size goes to)
KArgs make_args(const hint_plan* P, int B, bool backward);
void split_workspace(const hint_plan* P, int B, void* workspace, ChainBlock* b);
void bind_tape(const hint_plan* P, int B, float* tape, ChainBlock* b);
// part A (row-parallel, bit 0 of `parts`) and part B (weight gradients, bit 1) of the backward pass of one block or of
// blocks [cb0, cb0 + n_chain) of a chain of n_total blocks (the batch split of part B follows n_total: a bucketed
// launch sums in the same order as the whole chain's)
int run_backward(const hint_plan* P, const ChainBlock& one, const ChainBlock* chain, const ChainBlock* chain_host, int n_chain,
                 int cb0, int n_total, const float* x, const float* c, const float* g_z, const float* g_J, float* g_x, float* g_c,
                 float gz_scale, float gJ_const, int B, int accumulate, int parts, hipStream_t s, const AdamFuse* adam = nullptr);
int block_backward(const hint_plan* P, const float* params, const float* packed, const float* x, const float* tape,
                   const float* c, const float* g_z, const float* g_J, float* g_x, float* g_c, float* g_params,
                   int32_t accumulate, void* workspace, size_t workspace_bytes, const float* perm, float gz_scale,
                   float gJ_const, int32_t B, int parts, void* stream);

}  // namespace hint
