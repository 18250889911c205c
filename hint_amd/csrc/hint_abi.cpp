// Sizes and layouts of a plan's buffers, the block-level entry points of the C ABI (include/hint_amd.h), the re-pack
// groups, the optimizer and the build information; launches the kernels of hint_fwd.hip / hint_bwd.hip /
// hint_wl_*.hip / hint_wgrad.hip / hint_pack.hip / hint_optim.hip.
#include "hint_host.hpp"

#include <atomic>

using namespace hint;

static thread_local std::string g_err;
static unsigned long long* g_stamp_buf = nullptr;  // diagnostic builds only (hint_debug_set_stamp_buffer)

namespace hint {

static Knobs read_knobs() {
    Knobs k;
    auto get = [&](const char* name, int* out) {
        const char* e = std::getenv(name);
        if (!e || !*e) return false;
        *out = std::atoi(e);
        k.non_default += (k.non_default.empty() ? "" : " ") + std::string(name) + "=" + e;
        return true;
    };
    int v;
    if (get("HINT_PLAN_DUMP", &v)) k.plan_dump = v != 0;
    if (get("HINT_WL", &v)) k.wl = v != 0;
    if (get("HINT_WL_NR", &v)) k.wl_nr = v;
    if (get("HINT_SUB", &v)) k.sub = v != 0;
    if (get("HINT_NW", &v)) k.nw = v;
    if (get("HINT_LEAN", &v)) k.lean = v != 0;
    if (get("HINT_LEANW", &v)) k.leanw = v != 0;
    if (get("HINT_LEANW_MAX", &v)) k.leanw_max = std::min(std::max(v, 4), LEANW_MAX);
    if (get("HINT_FUSE_DW1", &v)) k.fuse_dw1 = v != 0;
    if (get("HINT_PF", &v)) k.pf = v != 0 ? 1 : 0;
    if (get("HINT_NO_BWD_FLY", &v)) k.no_bwd_fly = v != 0;
    if (get("HINT_DW_SPLITS", &v)) k.dw_splits = v;
    if (get("HINT_DW_SMALL", &v)) k.dw_small = v != 0 ? 1 : 0;
    if (get("HINT_ABLATION_OK", &v)) k.ablation_ok = v != 0;
    return k;
}
static std::mutex g_knobs_mu;
static Knobs* g_knobs = nullptr;          // (replaced, never freed: a reader may hold the old one across hint_debug_reload_knobs)
const Knobs& knobs() {
    std::lock_guard<std::mutex> lk(g_knobs_mu);
    if (!g_knobs) g_knobs = new Knobs(read_knobs());
    return *g_knobs;
}

int fail(const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return 1;
}
std::string& last_error_ref() { return g_err; }

// the plan variant a batch of B rows runs on, and (wave-local plans) how many 16-row tiles a workgroup takes:
//   up to one row tile per CU                  the 8-wavefront plan, one tile per workgroup
//   more                                       wave-local plans: ROW PAIRS - two tiles on one weight stream (hint_wl.hpp) - on the
//                                              8-wavefront plan; beyond two pairs per CU on the 4-wavefront plan (two workgroups
//                                              per CU) when that fits the LDS twice
//                                              other plans: the 4-wavefront plan, one tile per workgroup
const hint_plan* variant(const hint_plan* P, int B) {
    if (!P || !P->alt4) return P;
    const int ntiles = (B + ROWS - 1) / ROWS;
    if (ntiles <= P->num_cu) return P;
    const int nr_forced = knobs().wl_nr;
    if (P->wl && P->wl_nr2 && nr_forced != 1) {
        const hint_plan* A = P->alt4;
        const bool alt_twice = A->wl && A->wl_nr2 && std::max(plan_lds(A, false, 2), plan_lds(A, true, 2)) + 4096 <= LDS_LIMIT / 2;
        return (alt_twice && ntiles > 4 * P->num_cu) ? A : P;
    }
    return P->alt4;
}
int wl_nr_for(const hint_plan* Pv, int B) {       // Pv: the variant already picked
    if (!Pv->wl || !Pv->wl_nr2) return 1;
    const int nr_forced = knobs().wl_nr;
    if (nr_forced == 1 || nr_forced == 2) return nr_forced;
    return (B + ROWS - 1) / ROWS > Pv->num_cu ? 2 : 1;
}

// Tape layout (floats): [lane tiles: L x B x d][s: L x B x d][pad to 4][a1: Bp x WT + slack][a2: same]
//                       [sign bytes of a1: Bp/16 x WT/16 x 64 bytes][of a2: same]
// (lean plans: no a1 array)
int64_t tape_act_off(const hint_plan* P, int B) {
    return (2 * (int64_t)P->n_levels * B * P->d + 3) / 4 * 4;
}
int64_t act_stride(const hint_plan* P, int B) { return (int64_t)rows_padded(B) * P->WT + WS_SLACK; }
int64_t bits_stride(const hint_plan* P, int B) { return (int64_t)rows_padded(B) / ROWS * (P->WT / 16) * 64; }   // bytes

// batch split of part B: a multiple of 8 splits (one XCD each), enough workgroups to cover the chip,
// every workgroup reducing at least 128 rows
void wgrad_splits(const hint_plan* P, int B, int n_chain, int* splits_out, int* rows_out) {
    const long Bp = rows_padded(B);
    int splits = 8;
    while ((long)splits * P->n_wjobs * n_chain < (long)P->num_cu && Bp / (splits * 2) >= 128) splits *= 2;
    const int forced = knobs().dw_splits;          // (experiments)
    if (forced > 0 && Bp / forced >= 16) splits = forced;
    int rows_per_wg = (int)((Bp + splits - 1) / splits);
    rows_per_wg = (rows_per_wg + 15) / 16 * 16;
    if ((long)rows_per_wg * (splits - 1) >= Bp)   // tiny batches: fewer, non-empty splits
        splits = (int)((Bp + rows_per_wg - 1) / rows_per_wg);
    *splits_out = splits;
    *rows_out = rows_per_wg;
}

// Workspace layout (floats): [g1: Bp x WT + slack][g2: same][g_st: Bp x ST + slack, padded to 4][slabs: splits x param_floats]
int64_t ws_gst_off(const hint_plan* P, int B) { return (P->lean ? 1 : 2) * act_stride(P, B); }      // (lean plans: no g2 array)
int64_t ws_slab_off(const hint_plan* P, int B) {
    return ws_gst_off(P, B) + ((int64_t)rows_padded(B) * P->ST + WS_SLACK + 3) / 4 * 4;
}

// the backward kernel's first-layer gradient slabs (fuse_dw1) follow part B's slabs: [workgroup][tw_floats]
int64_t ws_thin_off(const hint_plan* P, int B) {        // floats from the part-B slabs' start
    int splits, rows;
    wgrad_splits(P, B, 1, &splits, &rows);        // (a chain never uses more splits than a single block)
    return (int64_t)splits * P->param_floats;
}

// The L2 warm-up has an explicit setter for the tests and A/B runs (hint_debug_set_prefetch): its initial value comes from
// HINT_PF (0 = off; hint_host.hpp Knobs).
static std::atomic<int> g_l2_prefetch{-1};
static std::atomic<int> g_last_lds[2] = {{0}, {0}};     // dynamic LDS bytes of the last row-kernel launch: [0] forward / inverse, [1] backward part A
bool l2_prefetch_on() {
    int v = g_l2_prefetch.load(std::memory_order_relaxed);
    if (v < 0) {
        v = knobs().pf;
        g_l2_prefetch.store(v, std::memory_order_relaxed);
    }
    return v != 0;
}
static int perm_lds_cap() { return PERM_LDS_MAX; }
static bool plan_dump() { return knobs().plan_dump; }
static bool no_bwd_fly() { return knobs().no_bwd_fly; }

// the 256-byte sink of the general kernels' L2 warm-up (hint_device.hpp prefetch_consumer) behind everything else of the launch
static int add_sink(const hint_plan* P, int total, KArgs* a) {
    a->sink_lds = 0;
    if (!P->wl && l2_prefetch_on() && total + 256 <= LDS_LIMIT) {
        a->sink_lds = total / (int)sizeof(float);
        total += 256;
    }
    return total;
}

// LDS bytes of the launch: the plan's, plus the chain's permutation matrices when they fit behind it
int lds_with_perms(const hint_plan* P, int lds_plan, int n_blocks, bool any_perm, KArgs* a, bool backward) {
    const long extra = (long)n_blocks * P->d * P->d * (long)sizeof(float);
    a->perm_lds = 0;
    const int cap = perm_lds_cap();
    if (plan_dump()) fprintf(stderr, "[hint plan] lds %d + perms %ld (cap %d)\n", lds_plan, extra, cap);
    int total = lds_plan;
    if (any_perm && extra <= cap && lds_plan + extra <= LDS_LIMIT) {
        a->perm_lds = lds_plan / (int)sizeof(float);
        total += (int)extra;
    }
    // the general kernels warm the L2 with the weights of what comes two phases later (hint_device.hpp prefetch_consumer): loads
    // straight into a 256-byte sink behind everything else (hint_debug_set_prefetch(0) / HINT_PF=0: never).  Not the wave-local
    // kernels: the same touch a block ahead is worth 6 us of 109 (forward) and 2 of 133 (backward) at cfg 2, but the call site -
    // wherever it was put - costs their register allocation 10 / 8 us (SGPR spills in the level loop), and from the block's top the
    // issuing wavefront's own loads wait for it (+14 / +9 us); as straight-line code - a `buffer_load ... lds` every wavefront
    // always issues, with an out-of-bound offset for the lanes that are not to touch anything - +2 / +6 us
    total = add_sink(P, total, a);
    if (plan_dump()) fprintf(stderr, "[hint plan] packed lines %d: sink at %d, lds %d\n", a->packed_lines, a->sink_lds, total);
    g_last_lds[backward ? 1 : 0].store(total, std::memory_order_relaxed);
    return total;
}
KArgs make_args(const hint_plan* P, int B, bool backward) {
    KArgs a{};
    a.meta = P->d_meta; a.meta_bytes = P->meta_bytes; a.recs = P->d_recs; a.total_rows = P->total_rows; a.thins = P->d_thins; a.total_tiles = P->total_tiles;
    a.units_off = P->units_off; a.tmap_off = P->tmap_off; a.ents_off = P->ents_off; a.rng_off = P->rng_off;
    a.lops_off = P->lops_off; a.lopsc = P->d_lopsc; a.lop_cnt = P->lop_cnt;
    a.n_groups = P->n_groups; a.n_levels = P->n_levels; a.n_units = P->n_units; a.nw = P->nw;
    a.d = P->d; a.dc = P->dc; a.xld = P->xld; a.cld = P->cld;
    a.abuf_tiles = P->abuf_tiles; a.slab_floats = backward ? P->slab_bwd : P->slab_fwd; a.gld = P->gld;
    a.region_floats = backward ? P->region_bwd : P->region_fwd;
    a.WT = P->WT; a.ST = P->ST; a.perm_lds = 0; a.stage_out = P->stage_out;
    a.thin_off = backward ? P->thin_b_off : P->thin_f_off;
    a.thin_floats = backward ? P->thin_b_floats : P->thin_f_floats;
    a.thin_lds = backward ? P->thin_lds_b : P->thin_lds_f;
    a.thin_grp = backward ? P->thin_grp_b : P->thin_grp_f;
    a.act_stride = act_stride(P, B); a.bits_stride = bits_stride(P, B);
    a.fuse_dw1 = P->fuse_dw1; a.tw_floats = P->tw_floats; a.thin_slab_off = ws_thin_off(P, B);
    a.lean = P->lean; a.a2_off = P->lean ? 0 : a.act_stride; a.bits_off = (P->lean ? 1 : 2) * a.act_stride;
    a.alpha = P->alpha; a.B = B; a.stamps = g_stamp_buf;
    a.rowdw_lds = backward ? P->rowdw_lds : 0;
    a.n_sub = P->n_sub;
    a.packed_tiles = (int)(P->packed_floats / 256);
    a.packed_lines = (int)(((P->packed_floats + P->n_bias) * 4 + 127) / 128);
    if (P->n_sub > 0) {
        const int* o = backward ? P->sub_lds_b : P->sub_lds_f;
        a.sub_slab = o[0]; a.sub_par = o[1]; a.sub_misc = o[2];
        a.sub_pf = P->sub_pf; a.sub_pb = P->sub_pb; a.sub_par_f4 = (P->sub_pf + P->sub_pb + P->sub_pbias) / 4;
        a.sub_bsrc = P->sub_bsrc; a.sub_bias_src = (int)P->packed_floats; a.sub_cols = P->sub_cols;
    }
    return a;
}

void split_workspace(const hint_plan* P, int B, void* workspace, ChainBlock* b) {
    b->wsG1 = (float*)workspace;
    b->wsGST = b->wsG1 + ws_gst_off(P, B);
    b->wsSlab = b->wsG1 + ws_slab_off(P, B);
}

// the hidden activations live inside the tape (the training forward writes them)
void bind_tape(const hint_plan* P, int B, float* tape, ChainBlock* b) {
    b->tape = tape;
    b->actA1 = tape ? tape + tape_act_off(P, B) : nullptr;
}

int grid_for(const hint_plan* P, int B) {
    const int ntiles = (B + ROWS - 1) / ROWS, nr = wl_nr_for(P, B);
    return std::min((ntiles + nr - 1) / nr, P->num_cu * 8);
}

// part A (row-parallel, bit 0 of `parts`) and part B (weight gradients, bit 1) of the backward pass of
// one block or a chain
int run_backward(const hint_plan* P, const ChainBlock& one, const ChainBlock* chain, const ChainBlock* chain_host, int n_chain,
                 int cb0, int n_total, const float* x, const float* c, const float* g_z, const float* g_J, float* g_x, float* g_c,
                 float gz_scale, float gJ_const, int B, int accumulate, int parts, hipStream_t s, const AdamFuse* adam) {
#ifdef HINT_ABLATE_STORE
    // (a diagnostic build whose row kernels keep no activations cannot train: only with HINT_ABLATION_OK=1 in the environment -
    //  the timing scripts set it - does it launch a backward pass at all)
    if (!knobs().ablation_ok) return fail("this is an ablation build (HINT_ABLATE_STORE): its gradients are wrong; set HINT_ABLATION_OK=1 for timing runs");
#endif

    if ((parts & 1) && P->wl) {
        KArgs a = make_args(P, B, true);
        const int nr = wl_nr_for(P, B);
        WlArgs w = P->wl_b[nr - 1];
        bool any_perm = one.perm != nullptr;
        if (chain_host) for (int i = 0; i < n_chain; ++i) any_perm = any_perm || chain_host[i].perm != nullptr;
        const int lds = lds_with_perms(P, plan_lds(P, true, nr), n_chain, any_perm, &a, true);
        w.off_perm = a.perm_lds;
        HIP_TRY(launch_wl_bwd(a, w, lds, grid_for(P, B), one, chain, n_chain, x, g_z, g_J, g_x, gz_scale, gJ_const, s));
    } else if (parts & 1) {
        // (the permutation matrices stay in global memory here: one d x d product per block)
        // (round 5: the backward's L2 warm-up gets its sink too - until then KArgs::sink_lds was 0 here and the block in hint_bwd.hip dead code)
        KArgs a = make_args(P, B, true);
        const int lds = add_sink(P, P->lds_bwd, &a);
        g_last_lds[1].store(lds, std::memory_order_relaxed);
        HIP_TRY((P->has_fly && !no_bwd_fly() ? launch_bwd_fly : P->row_ntt <= 3 && P->rowdw_lds == 0 ? launch_bwd_n3 : launch_bwd)(a, lds, grid_for(P, B), one, chain, n_chain, x, c,
                                                               g_z, g_J, g_x, g_c, gz_scale, gJ_const, s));
    }
    if (!(parts & 2)) return 0;
    int splits, rows_per_wg;
    wgrad_splits(P, B, n_total, &splits, &rows_per_wg);     // (the whole chain's count: a bucketed launch sums in the same order)
    HIP_TRY(launch_wgrad(P->d_wjobs, P->n_wjobs, P->wsorted ? -P->n_wsmall - 1 : P->n_wsmall, splits, one, chain, n_chain, cb0, P->WT, P->ST, P->d, P->dc, P->n_levels, B,
                         rows_padded(B), rows_per_wg, act_stride(P, B), P->lean ? 0 : act_stride(P, B),
                         (P->lean ? 1 : 2) * act_stride(P, B) * 4 + bits_stride(P, B), P->param_floats, x, c, P->d_real,
                         accumulate, P->fuse_dw1 ? P->d_twmap : nullptr, P->tw_floats, ws_thin_off(P, B), grid_for(P, B),
                         P->num_cu, adam, P->has_leanw != 0, s));
    return 0;
}

}  // namespace hint

static int apply(const hint_plan* P, bool rev, const float* params, const float* packed, const float* x,
                 const float* c, float* z, float* J, float* tape, const float* perm, const float* J_in,
                 float* loss_acc, int32_t B, void* stream, float noise = 0.f, const uint64_t* rng_state = nullptr,
                 float* x_noisy = nullptr) {
    const char* what = rev ? "inverse" : "forward";
    if (!P || !params || !packed || !x || !z || !J) return fail("hint_block_%s: null argument", what);
    if (P->dc > 0 && !c) return fail("hint_block_%s: plan has dc=%d but c is NULL", what, P->dc);
    if (B < 0) return fail("negative batch");
    if (B == 0) return 0;
    P = variant(P, B);
    ChainBlock one{};
    one.params = params; one.packed = packed; one.perm = perm;
    bind_tape(P, B, rev ? nullptr : tape, &one);
    KArgs a = make_args(P, B, false);
    const int nr = wl_nr_for(P, B);
    const int lds = lds_with_perms(P, plan_lds(P, false, nr), 1, perm != nullptr, &a);
    if (P->wl) {
        WlArgs w = P->wl_f[nr - 1];
        w.off_perm = a.perm_lds;
        HIP_TRY(launch_wl_apply(rev, a, w, lds, grid_for(P, B), one, nullptr, 1, x, z, J, J_in, loss_acc, noise,
                                (const unsigned long long*)rng_state, x_noisy, (hipStream_t)stream));
        return 0;
    }
    HIP_TRY(launch_apply(rev, P->has_fly != 0, a, lds, grid_for(P, B), one, nullptr, 1, x, c, z, J, J_in, loss_acc, noise,
                         (const unsigned long long*)rng_state, x_noisy, (hipStream_t)stream));
    return 0;
}

extern "C" {

int hint_abi_version(void) { return HINT_AMD_ABI_VERSION; }

const char* hint_build_info(void) {
    // what the shipped binary was compiled with (the box that runs it may carry another HIP runtime: bench.py prints both), its
    // source stamp (-DHINT_SRC_STAMP: a hash of hint_amd/csrc + include, Makefile) and the HINT_* knobs set in this process
    static thread_local std::string info;
    info = std::string("libhint_amd abi ") + std::to_string(HINT_AMD_ABI_VERSION) +
#ifdef HINT_ABLATE_STORE
                                    " ABLATION BUILD (no activation / gradient stores: timing only, results are wrong)" +
#endif
                                    ", gfx950, HIP " +
                                    std::to_string(HIP_VERSION_MAJOR) + "." + std::to_string(HIP_VERSION_MINOR) + "." +
                                    std::to_string(HIP_VERSION_PATCH) + ", clang " + __clang_version__ +
#ifdef HINT_SRC_STAMP
                                    ", src " HINT_SRC_STAMP +
#endif
                                    (knobs().non_default.empty() ? std::string("") : ", knobs: " + knobs().non_default);
    return info.c_str();
}
const char* hint_last_error(void) { return g_err.c_str(); }

int hint_debug_reload_knobs(void) {
    std::lock_guard<std::mutex> lk(g_knobs_mu);
    g_knobs = new Knobs(read_knobs());
    g_l2_prefetch.store(g_knobs->pf, std::memory_order_relaxed);
    return 0;
}

int hint_debug_set_prefetch(int on) {
    hint::l2_prefetch_on();        // (so that the previous value is the environment's, not "unread")
    return g_l2_prefetch.exchange(on ? 1 : 0);
}
int32_t hint_debug_last_lds_bytes(int32_t backward) { return g_last_lds[backward ? 1 : 0].load(std::memory_order_relaxed); }

int64_t hint_plan_param_floats(const hint_plan* P) { return P ? P->param_floats : -1; }
// + 4 KiB of slack behind the bias region (a padded row's dummy steps load up to three tiles past its last)
int64_t hint_plan_packed_floats(const hint_plan* P) { return P ? P->packed_floats + P->n_bias + 1024 : -1; }

static inline int rows_padded(int B) { return (B + ROWS - 1) / ROWS * ROWS; }
int64_t hint_plan_tape_floats(const hint_plan* P, int32_t B) {
    if (!P || B < 0) return -1;
    P = variant(P, B);
    return tape_act_off(P, B) + (P->lean ? 1 : 2) * act_stride(P, B) + 2 * bits_stride(P, B) / 4;
}

size_t hint_plan_workspace_bytes(const hint_plan* P, int32_t B) {
    if (!P || B <= 0) return 0;
    P = variant(P, B);
    const int64_t thin = P->fuse_dw1 ? (int64_t)grid_for(P, B) * P->tw_floats : 0;
    return (size_t)(ws_slab_off(P, B) + ws_thin_off(P, B) + thin) * sizeof(float);
}

int32_t hint_plan_lds_bytes(const hint_plan* P, int32_t backward) {
    return P ? plan_lds(P, backward != 0) : -1;
}

int hint_plan_describe(const hint_plan* P, int32_t B, int32_t* out) {
    if (!P || !out || B < 0) return fail("hint_plan_describe: bad arguments");
    P = variant(P, B);
    out[0] = P->wl; out[1] = wl_nr_for(P, B); out[2] = P->nw; out[3] = P->lean;
    out[4] = P->n_sub; out[5] = P->row_ntt; out[6] = P->rowdw_lds > 0 ? 1 : 0; out[7] = P->has_fly;
    return 0;
}

int hint_block_pack(const hint_plan* P, const float* params, float* packed, void* stream) {
    if (!P || !params || !packed) return fail("hint_block_pack: null argument");
    HIP_TRY(launch_pack(P->d_segs, P->d_ptiles, P->n_ptiles, P->d_bmap, P->n_bias, (long)P->packed_floats, params,
                        packed, (hipStream_t)stream));
    return 0;
}

struct hint_pack_group {
    PackItem* d_items = nullptr;
    int n = 0, grid = 0;
};

int hint_pack_group_create(const hint_plan* const* plans, const float* const* params, float* const* packed,
                           int32_t n, hint_pack_group** out) {
    if (!plans || !params || !packed || n <= 0 || !out) return fail("hint_pack_group_create: bad arguments");
    std::vector<PackItem> items(n);
    int grid = 0;
    for (int i = 0; i < n; ++i) {
        const hint_plan* P = plans[i];
        if (!P || !params[i] || !packed[i]) return fail("hint_pack_group_create: null entry %d", i);
        PackItem& q = items[i];
        q.segs = P->d_segs; q.ptiles = P->d_ptiles; q.bmap = P->d_bmap; q.params = params[i]; q.packed = packed[i];
        q.bias_off = P->packed_floats; q.n_tiles = P->n_ptiles; q.n_bias = P->n_bias; q.grid_begin = grid; q.pad = 0;
        grid += P->n_ptiles + (P->n_bias + 255) / 256;
    }
    hint_pack_group* G = new hint_pack_group();
    G->n = n; G->grid = grid;
    hipError_t e = hipMalloc((void**)&G->d_items, items.size() * sizeof(PackItem));
    if (e == hipSuccess) e = hipMemcpy(G->d_items, items.data(), items.size() * sizeof(PackItem), hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(G->d_items); delete G; return fail("hint_pack_group_create: %s", hipGetErrorString(e)); }
    *out = G;
    return 0;
}

int hint_pack_group_run(const hint_pack_group* G, void* stream) {
    return hint_pack_group_run_ex(G, nullptr, 0, nullptr, nullptr, stream);
}

int hint_pack_group_run_ex(const hint_pack_group* G, float* zero_buf, int32_t zero_floats, uint64_t* rng_state,
                           float* opt_state, void* stream) {
    if (!G) return fail("hint_pack_group_run: null group");
    if (zero_floats < 0 || (zero_floats > 0 && !zero_buf)) return fail("hint_pack_group_run_ex: bad zero buffer");
    if (opt_state && !rng_state) return fail("hint_pack_group_run_ex: opt_state needs the step counter of rng_state");
    HIP_TRY(launch_pack_many(G->d_items, G->n, G->grid, zero_buf, zero_floats, (unsigned long long*)rng_state,
                             opt_state, (hipStream_t)stream));
    return 0;
}

void hint_pack_group_destroy(hint_pack_group* G) {
    if (!G) return;
    (void)hipFree(G->d_items);
    delete G;
}

int hint_block_forward(const hint_plan* P, const float* params, const float* packed, const float* x,
                       const float* c, float* z, float* J, float* tape, int32_t B, void* stream) {
    return apply(P, false, params, packed, x, c, z, J, tape, nullptr, nullptr, nullptr, B, stream);
}

int hint_block_forward_ex(const hint_plan* P, const float* params, const float* packed, const float* x,
                          const float* c, float* z, float* J, float* tape, const float* perm,
                          const float* J_in, float* loss_acc, int32_t B, void* stream) {
    return apply(P, false, params, packed, x, c, z, J, tape, perm, J_in, loss_acc, B, stream);
}

int hint_block_forward_noisy(const hint_plan* P, const float* params, const float* packed, const float* x,
                             const float* c, float* z, float* J, float* tape, const float* perm,
                             const float* J_in, float* loss_acc, float noise, const uint64_t* rng_state,
                             float* x_noisy, int32_t B, void* stream) {
    if (noise != 0.f && (!rng_state || !x_noisy)) return fail("hint_block_forward_noisy: noise needs rng_state and x_noisy");
    if (noise != 0.f && perm) return fail("hint_block_forward_noisy: the noise is added to the block's input, in front of no permutation");
    return apply(P, false, params, packed, x, c, z, J, tape, perm, J_in, loss_acc, B, stream, noise, noise != 0.f ? rng_state : nullptr,
                 noise != 0.f ? x_noisy : nullptr);
}

int hint_block_inverse(const hint_plan* P, const float* params, const float* packed, const float* z,
                       const float* c, float* x, float* J, int32_t B, void* stream) {
    return apply(P, true, params, packed, z, c, x, J, nullptr, nullptr, nullptr, nullptr, B, stream);
}

int hint_block_inverse_ex(const hint_plan* P, const float* params, const float* packed, const float* z,
                          const float* c, float* x, float* J, const float* perm, const float* J_in,
                          int32_t B, void* stream) {
    return apply(P, true, params, packed, z, c, x, J, nullptr, perm, J_in, nullptr, B, stream);
}

int hint_block_backward(const hint_plan* P, const float* params, const float* packed, const float* x,
                        const float* tape, const float* c, const float* g_z, const float* g_J, float* g_x,
                        float* g_c, float* g_params, int32_t accumulate, void* workspace,
                        size_t workspace_bytes, int32_t B, void* stream) {
    return hint_block_backward_ex(P, params, packed, x, tape, c, g_z, g_J, g_x, g_c, g_params, accumulate,
                                  workspace, workspace_bytes, nullptr, 1.0f, 0.0f, B, stream);
}

}  // extern "C"

namespace hint {
// parts: bit 0 the row-parallel kernel, bit 1 the weight gradients (run_backward)
int block_backward(const hint_plan* P, const float* params, const float* packed, const float* x,
                          const float* tape, const float* c, const float* g_z, const float* g_J, float* g_x,
                          float* g_c, float* g_params, int32_t accumulate, void* workspace,
                          size_t workspace_bytes, const float* perm, float gz_scale, float gJ_const,
                          int32_t B, int parts, void* stream) {
    if (!P || !params || !packed || (!x && !perm) || !g_x || (!g_params && (parts & 2))) return fail("hint_block_backward: null argument");
    if (perm && !tape) return fail("hint_block_backward_ex: a fused permutation needs the tape of hint_block_forward_ex");
    if (P->dc > 0 && !c) return fail("hint_block_backward: plan has dc=%d but c is NULL", P->dc);
    if (!tape && B > 0) return fail("hint_block_backward: tape is NULL (the backward pass reads the forward's lane tiles, s values and activations from it)");
    if (B < 0) return fail("negative batch");
    hipStream_t s = (hipStream_t)stream;
    if (B == 0) {
        if (!accumulate && g_params) HIP_TRY(launch_zero(g_params, (long)P->param_floats, P->num_cu, s));
        return 0;
    }
    P = variant(P, B);
    if (!workspace || workspace_bytes < hint_plan_workspace_bytes(P, B))
        return fail("hint_block_backward: workspace too small (%zu < %zu)", workspace_bytes,
                    hint_plan_workspace_bytes(P, B));
    if (((uintptr_t)workspace & 15) != 0) return fail("hint_block_backward: workspace must be 16-byte aligned");
    if (g_params && ((uintptr_t)g_params & 15) != 0) return fail("hint_block_backward: g_params must be 16-byte aligned");
    ChainBlock one{};
    one.params = params; one.packed = packed; one.perm = perm;
    bind_tape(P, B, const_cast<float*>(tape), &one);
    one.gparams = g_params;
    split_workspace(P, B, workspace, &one);
    return run_backward(P, one, nullptr, nullptr, 1, 0, 1, x, c, g_z, g_J, g_x, g_c, gz_scale, gJ_const, B, accumulate ? 1 : 0, parts, s);
}
}  // namespace hint

extern "C" {

int hint_block_backward_ex(const hint_plan* P, const float* params, const float* packed, const float* x,
                           const float* tape, const float* c, const float* g_z, const float* g_J, float* g_x,
                           float* g_c, float* g_params, int32_t accumulate, void* workspace,
                           size_t workspace_bytes, const float* perm, float gz_scale, float gJ_const,
                           int32_t B, void* stream) {
    return block_backward(P, params, packed, x, tape, c, g_z, g_J, g_x, g_c, g_params, accumulate, workspace, workspace_bytes,
                          perm, gz_scale, gJ_const, B, 3, stream);
}

int hint_block_backward_rows(const hint_plan* P, const float* params, const float* packed, const float* x,
                             const float* tape, const float* c, const float* g_z, const float* g_J, float* g_x,
                             float* g_c, void* workspace, size_t workspace_bytes, const float* perm, float gz_scale,
                             float gJ_const, int32_t B, void* stream) {
    // part A alone: g_x, g_c and the per-row factors of the weight gradients (left in `workspace` for a later
    // hint_chain_wgrad_range / hint_chain_wgrad_adam over a chain that holds this block with the same tape and workspace)
    return block_backward(P, params, packed, x, tape, c, g_z, g_J, g_x, g_c, nullptr, 1, workspace, workspace_bytes, perm, gz_scale,
                          gJ_const, B, 1, stream);
}

#ifdef HINT_STAMPS
// diagnostic builds only (make stamps): device buffer of MAX_NW x 256 uint64 that workgroup 0 of the block
// kernels fills with shader-clock stamps of its phase boundaries; not part of the shipped ABI
int hint_debug_set_stamp_buffer(void* device_buffer) { g_stamp_buf = (unsigned long long*)device_buffer; return 0; }
// ... and one of [workgroups of a part-B launch][wavefronts][8] uint64 for hint_wgrad_kernel (tools/stamps_dw.py); NULL: off
int hint_debug_set_dw_stamp_buffer(void* device_buffer) { hint::set_dw_stamps((unsigned long long*)device_buffer); return 0; }
#endif

static int adam_num_cu() {
    static int num_cu = 0;
    if (num_cu == 0) {
        int dev = 0; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) num_cu = prop.multiProcessorCount;
        if (num_cu <= 0) num_cu = 256;
    }
    return num_cu;
}

int hint_adam_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, int32_t step,
                   float lr, float beta1, float beta2, float eps, float weight_decay, float grad_scale,
                   float grad_clamp, int32_t zero_grads, void* stream) {
    if (!params || !grads || !exp_avg || !exp_avg_sq) return fail("hint_adam_step: null argument");
    if (n < 0 || step < 1) return fail("hint_adam_step: n must be >= 0 and step >= 1");
    if (n == 0) return 0;
    if ((((uintptr_t)params | (uintptr_t)grads | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) != 0)
        return fail("hint_adam_step: buffers must be 16-byte aligned");
    // bias corrections in double like torch.optim.Adam's python scalars
    const double bc1 = 1.0 - std::pow((double)beta1, (double)step);
    const double bc2 = 1.0 - std::pow((double)beta2, (double)step);
    HIP_TRY(launch_adam(params, grads, exp_avg, exp_avg_sq, (long)n, (float)((double)lr / bc1), beta1, beta2,
                        (float)(1.0 / std::sqrt(bc2)), eps, weight_decay, grad_scale,
                        grad_clamp > 0.f ? grad_clamp : 3.0e38f, zero_grads ? 1 : 0, adam_num_cu(), nullptr,
                        (hipStream_t)stream));
    return 0;
}

int hint_adam_step_dev(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                       const float* opt_state, float beta1, float beta2, float eps, float weight_decay,
                       float grad_scale, float grad_clamp, int32_t zero_grads, void* stream) {
    if (!params || !grads || !exp_avg || !exp_avg_sq || !opt_state) return fail("hint_adam_step_dev: null argument");
    if (n < 0) return fail("hint_adam_step_dev: n must be >= 0");
    if (n == 0) return 0;
    if ((((uintptr_t)params | (uintptr_t)grads | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) != 0)
        return fail("hint_adam_step_dev: buffers must be 16-byte aligned");
    HIP_TRY(launch_adam(params, grads, exp_avg, exp_avg_sq, (long)n, 0.f, beta1, beta2, 0.f, eps, weight_decay,
                        grad_scale, grad_clamp > 0.f ? grad_clamp : 3.0e38f, zero_grads ? 1 : 0, adam_num_cu(),
                        opt_state, (hipStream_t)stream));
    return 0;
}

}  // extern "C"