// Host side of the C ABI (include/hint_amd.h): turns the node list of one coupling tree
// (the structure /root/reference/hint.py:25-54 builds recursively) into a static level
// schedule in device memory, and launches the kernels of hint_kernels.hip.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/hint_amd.h"
#include "hint_dev.h"

namespace hint {
hipError_t launch_apply(bool rev, const KArgs& a, int lds_bytes, int grid, const float* params,
                        const float* x, const float* c, float* z, float* J, float* tape,
                        hipStream_t stream);
hipError_t launch_bwd(const KArgs& a, int lds_bytes, int grid, const float* params, const float* x,
                      const float* tape, const float* c, const float* g_z, const float* g_J, float* g_x, float* g_c,
                      float* wsV, float* wsA1, float* wsA2, float* wsG1, float* wsG2, float* wsG3,
                      hipStream_t stream);
hipError_t launch_dw(const DWJob* jobs, int n_jobs, int splits, const float* wsV, const float* wsA1,
                     const float* wsA2, const float* wsG1, const float* wsG2, const float* wsG3,
                     int WT, int VT, int ST, int Bp, int rows_per_wg, float* gparams,
                     hipStream_t stream);
hipError_t set_max_lds(int fwd_bytes, int bwd_bytes);
hipError_t launch_adam(float* p, const float* g, float* m, float* v, long n, float lr_t, float b1, float b2,
                       float inv_sqrt_bc2, float eps, float wd, float gscale, float gclamp, int num_cu,
                       hipStream_t stream);
}  // namespace hint

using namespace hint;

static thread_local std::string g_err;

static int fail(const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return 1;
}

#define HIP_TRY(expr)                                                              \
    do {                                                                           \
        hipError_t e_ = (expr);                                                    \
        if (e_ != hipSuccess) return fail("%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

static inline int pad16(int v) { return (v + 15) & ~15; }

// LDS row stride for a buffer read with ds_read_b128 by lanes (row = l&15, 16-byte column
// slot = l>>4): stride = 8 (mod 64) floats makes each 16-lane group of the instruction hit 64
// distinct banks (MI355X_MICROARCH.md §LDS).
static inline int lds_stride(int width) {
    int w = std::max(width, 16);
    int ld = ((w + 63) / 64) * 64 + 8;
    if (ld - 64 >= w) ld -= 64;
    return ld;
}

struct hint_plan {
    int device = -1;
    int d = 0, dc = 0, n_nodes = 0, n_groups = 0, n_levels = 0, n_dwjobs = 0;
    float alpha = 0.f;
    int64_t param_floats = 0;
    int WT = 0, VT = 0, ST = 0;
    int xld = 0, cld = 0, ald = 0, vld = 0, sld = 0;
    int lds_fwd = 0, lds_bwd = 0;
    int num_cu = 256;
    DNode* d_nodes = nullptr;
    DGroup* d_groups = nullptr;
    Job* d_jobs = nullptr;
    Ent* d_ents = nullptr;
    DWJob* d_dwjobs = nullptr;
};

static constexpr int LDS_LIMIT = 160 * 1024;
static int g_bwd_stages = 3;   // profiling aid: bit0 = row-parallel part A, bit1 = weight-gradient part B
static constexpr int WS_SLACK = 64;   // floats of slack at the end of every workspace array

extern "C" {

int hint_abi_version(void) { return HINT_AMD_ABI_VERSION; }
const char* hint_last_error(void) { return g_err.c_str(); }

int hint_plan_create(const hint_node_desc* nodes, int32_t n_nodes, int32_t d, int32_t dc, float clamp,
                     hint_plan** out) {
    if (!nodes || n_nodes <= 0 || d <= 0 || dc < 0 || !out) return fail("hint_plan_create: bad arguments");
    *out = nullptr;
    // ---- validate the tree: lane ranges inside [0,d), same-depth nodes disjoint ----
    int max_depth = 0;
    for (int i = 0; i < n_nodes; ++i) {
        const hint_node_desc& n = nodes[i];
        if (n.D < 1 || n.k != n.D / 2 || n.r != n.D - n.k || n.off < 0 || n.off + n.D > d || n.h < 1 || n.depth < 0)
            return fail("hint_plan_create: node %d is malformed (off=%d D=%d k=%d r=%d h=%d depth=%d)", i,
                        n.off, n.D, n.k, n.r, n.h, n.depth);
        max_depth = std::max(max_depth, n.depth);
    }
    for (int i = 0; i < n_nodes; ++i)
        for (int j = i + 1; j < n_nodes; ++j)
            if (nodes[i].depth == nodes[j].depth && nodes[i].off < nodes[j].off + nodes[j].D &&
                nodes[j].off < nodes[i].off + nodes[i].D)
                return fail("hint_plan_create: nodes %d and %d of depth %d overlap", i, j, nodes[i].depth);

    hint_plan* P = new hint_plan();
    P->d = d;
    P->dc = dc;
    P->n_nodes = n_nodes;
    P->alpha = (float)((double)clamp * 0.636);   // hint.py:57,60 (python float product, then fp32)
    P->xld = pad16(d) + 4;
    P->cld = dc > 0 ? pad16(dc) + 4 : 0;

    // ---- forward order: deepest level first (children before parents, hint.py:70-73) ----
    std::vector<int> order;
    for (int dep = max_depth; dep >= 0; --dep)
        for (int i = 0; i < n_nodes; ++i)
            if (nodes[i].depth == dep) order.push_back(i);

    // The LDS footprint of the backward kernel bounds what one group may hold:
    //   ROWS * (2*xld + 2*cld + 2*vld + 2*ald + 2*sld + 1) floats  <=  160 KiB.
    // Every single node must fit (that fixes minimum strides); groups then grow up to soft
    // caps so that shallow-but-wide and deep-but-narrow levels end up with similar footprints.
    const int WMAX = 768;
    int min_aw = 0, min_vw = 0, min_sw = 0;
    for (int i = 0; i < n_nodes; ++i) {
        min_aw = std::max(min_aw, 2 * pad16(nodes[i].h));
        min_vw = std::max(min_vw, pad16(nodes[i].k + dc));
        min_sw = std::max(min_sw, 2 * pad16(nodes[i].r));
    }
    const int cap_aw = std::max(min_aw, 512), cap_vw = std::max(min_vw, 128), cap_sw = std::max(min_sw, 256);
    auto bwd_bytes = [&](int aw_, int vw_, int sw_) {
        return 4 * ROWS * (2 * P->xld + 2 * P->cld + 2 * lds_stride(vw_) + 2 * lds_stride(aw_) + 2 * lds_stride(sw_) + 1);
    };
    std::vector<DNode> dn;
    std::vector<DGroup> dg;
    std::vector<Job> jobs;
    std::vector<Ent> ents;
    std::vector<DWJob> dwj;
    int wcol = 0, wvcol = 0, wscol = 0;
    int max_aw = 0, max_vw = 0, max_sw = 0;
    int64_t pmax = 0;

    size_t pos = 0;
    while (pos < order.size()) {
        DGroup g{};
        g.node_begin = (int)dn.size();
        g.wcol0 = wcol; g.wvcol0 = wvcol; g.wscol0 = wscol;
        int aw = 0, vw = 0, sw = 0;
        const int depth = nodes[order[pos]].depth;
        while (pos < order.size() && nodes[order[pos]].depth == depth) {
            const hint_node_desc& n = nodes[order[pos]];
            const int hp = pad16(n.h), rp = pad16(n.r), cin = n.k + dc, cinp = pad16(cin);
            if (2 * hp > WMAX) {
                delete P;
                return fail("hint_plan_create: hidden width %d exceeds the supported maximum %d", n.h, WMAX / 2);
            }
            if (bwd_bytes(2 * hp, cinp, 2 * rp) > LDS_LIMIT) {
                delete P;
                return fail("hint_plan_create: a node with h=%d, cin=%d, r=%d does not fit the 160 KiB LDS", n.h, cin, n.r);
            }
            // start a new group at the same depth when this node would overflow the budget
            if (aw > 0 && (aw + 2 * hp > cap_aw || vw + cinp > cap_vw || sw + 2 * rp > cap_sw ||
                           bwd_bytes(std::max(min_aw, aw + 2 * hp), std::max(min_vw, vw + cinp),
                                     std::max(min_sw, sw + 2 * rp)) > LDS_LIMIT))
                break;
            DNode q{};
            q.off = n.off; q.k = n.k; q.r = n.r; q.h = n.h; q.cin = cin;
            q.hp = hp; q.rp = rp; q.cinp = cinp;
            q.acol = aw; q.vcol = vw; q.scol = sw;
            q.wcol = wcol; q.wvcol = wvcol; q.wscol = wscol;
            const int64_t sizes[6] = {(int64_t)n.h * cin, n.h, (int64_t)n.h * n.h, n.h, (int64_t)n.r * n.h, n.r};
            for (int t = 0; t < 12; ++t) {
                q.p[t] = n.p_off[t];
                if (n.p_off[t] < 0) { delete P; return fail("hint_plan_create: negative parameter offset"); }
                pmax = std::max(pmax, n.p_off[t] + sizes[t % 6]);
            }
            aw += 2 * hp; vw += cinp; sw += 2 * rp;
            wcol += 2 * hp; wvcol += cinp; wscol += 2 * rp;
            dn.push_back(q);
            ++pos;
        }
        g.node_end = (int)dn.size();
        g.aw = aw; g.vw = vw; g.sw = sw;
        g.level = max_depth - depth;
        g.level_last = (pos >= order.size() || nodes[order[pos]].depth != depth) ? 1 : 0;
        max_aw = std::max(max_aw, aw); max_vw = std::max(max_vw, vw); max_sw = std::max(max_sw, sw);
        // jobs: one per 16-wide output tile
        g.jobsH_begin = (int)jobs.size();
        for (int ni = g.node_begin; ni < g.node_end; ++ni)
            for (int net = 0; net < 2; ++net)
                for (int t = 0; t < dn[ni].hp / 16; ++t) jobs.push_back(Job{ni, net, t, 0});
        g.jobsH_cnt = (int)jobs.size() - g.jobsH_begin;
        g.jobsR_begin = (int)jobs.size();
        for (int ni = g.node_begin; ni < g.node_end; ++ni)
            for (int net = 0; net < 2; ++net)
                for (int t = 0; t < dn[ni].rp / 16; ++t) jobs.push_back(Job{ni, net, t, 0});
        g.jobsR_cnt = (int)jobs.size() - g.jobsR_begin;
        g.jobsC_begin = (int)jobs.size();
        for (int ni = g.node_begin; ni < g.node_end; ++ni)
            for (int t = 0; t < dn[ni].cinp / 16; ++t) jobs.push_back(Job{ni, 0, t, 0});
        g.jobsC_cnt = (int)jobs.size() - g.jobsC_begin;
        g.ent_begin = (int)ents.size();
        for (int ni = g.node_begin; ni < g.node_end; ++ni)
            for (int j = 0; j < dn[ni].r; ++j)
                ents.push_back(Ent{dn[ni].off + dn[ni].k + j, dn[ni].scol + j, dn[ni].scol + dn[ni].rp + j, ni});
        g.ent_cnt = (int)ents.size() - g.ent_begin;
        dg.push_back(g);
    }
    P->n_groups = (int)dg.size();
    P->n_levels = max_depth + 1;
    P->WT = wcol; P->VT = std::max(wvcol, 16); P->ST = wscol;
    P->param_floats = pmax;
    P->ald = lds_stride(max_aw);
    P->vld = lds_stride(max_vw);
    P->sld = lds_stride(max_sw);
    P->lds_fwd = 4 * ROWS * (P->xld + P->cld + P->vld + 2 * P->ald + P->sld + 1);
    P->lds_bwd = 4 * ROWS * (2 * P->xld + 2 * P->cld + 2 * P->vld + 2 * P->ald + 2 * P->sld + 1);
    if (P->lds_bwd > LDS_LIMIT || P->lds_fwd > LDS_LIMIT) {
        int need = P->lds_bwd;
        delete P;
        return fail("hint_plan_create: block needs %d bytes of LDS (> %d); d/dc/h too large", need, LDS_LIMIT);
    }

    // ---- weight-gradient jobs: 48x48 output tiles of dW1, dW2, dW3 of every (node, net) ----
    for (size_t ni = 0; ni < dn.size(); ++ni) {
        const DNode& q = dn[ni];
        for (int net = 0; net < 2; ++net) {
            struct L { int gsel, gcol, M, xsel, xcol, N; int64_t wofs, bofs; } ls[3] = {
                {0, q.wcol + net * q.hp, q.h, 0, q.wvcol, q.cin, q.p[net * 6 + 0], q.p[net * 6 + 1]},
                {1, q.wcol + net * q.hp, q.h, 1, q.wcol + net * q.hp, q.h, q.p[net * 6 + 2], q.p[net * 6 + 3]},
                {2, q.wscol + net * q.rp, q.r, 2, q.wcol + net * q.hp, q.h, q.p[net * 6 + 4], q.p[net * 6 + 5]},
            };
            for (const L& l : ls)
                for (int m0 = 0; m0 < l.M; m0 += 48) {
                    int n0 = 0;
                    do {
                        dwj.push_back(DWJob{l.gsel, l.gcol, l.M, l.xsel, l.xcol, l.N, m0, n0, l.wofs, l.bofs});
                        n0 += 48;
                    } while (n0 < l.N);
                }
        }
    }
    P->n_dwjobs = (int)dwj.size();

    // ---- upload ----
    HIP_TRY(hipGetDevice(&P->device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, P->device));
    P->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    auto upload = [](void** dst, const void* src, size_t bytes) -> hipError_t {
        hipError_t e = hipMalloc(dst, std::max<size_t>(bytes, 16));
        if (e != hipSuccess) return e;
        return bytes ? hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice) : hipSuccess;
    };
    hipError_t e = hipSuccess;
    if (e == hipSuccess) e = upload((void**)&P->d_nodes, dn.data(), dn.size() * sizeof(DNode));
    if (e == hipSuccess) e = upload((void**)&P->d_groups, dg.data(), dg.size() * sizeof(DGroup));
    if (e == hipSuccess) e = upload((void**)&P->d_jobs, jobs.data(), jobs.size() * sizeof(Job));
    if (e == hipSuccess) e = upload((void**)&P->d_ents, ents.data(), ents.size() * sizeof(Ent));
    if (e == hipSuccess) e = upload((void**)&P->d_dwjobs, dwj.data(), dwj.size() * sizeof(DWJob));
    if (e == hipSuccess) e = set_max_lds(P->lds_fwd, P->lds_bwd);
    if (e != hipSuccess) {
        hint_plan_destroy(P);
        return fail("hint_plan_create: device setup failed: %s", hipGetErrorString(e));
    }
    *out = P;
    return 0;
}

void hint_plan_destroy(hint_plan* P) {
    if (!P) return;
    (void)hipFree(P->d_nodes);
    (void)hipFree(P->d_groups);
    (void)hipFree(P->d_jobs);
    (void)hipFree(P->d_ents);
    (void)hipFree(P->d_dwjobs);
    delete P;
}

int64_t hint_plan_param_floats(const hint_plan* P) { return P ? P->param_floats : -1; }

int64_t hint_plan_tape_floats(const hint_plan* P, int32_t B) {
    if (!P || B < 0) return -1;
    return (int64_t)(P->n_levels - 1) * B * P->d;
}

static inline int rows_padded(int B) { return (B + ROWS - 1) / ROWS * ROWS; }

size_t hint_plan_workspace_bytes(const hint_plan* P, int32_t B) {
    if (!P || B <= 0) return 0;
    const size_t Bp = rows_padded(B);
    const size_t floats = Bp * ((size_t)4 * P->WT + P->VT + P->ST) + 6 * WS_SLACK;
    return floats * sizeof(float);
}

int32_t hint_plan_lds_bytes(const hint_plan* P, int32_t backward) {
    return P ? (backward ? P->lds_bwd : P->lds_fwd) : -1;
}

static KArgs make_args(const hint_plan* P, int B) {
    KArgs a{};
    a.nodes = P->d_nodes; a.groups = P->d_groups; a.jobs = P->d_jobs; a.ents = P->d_ents;
    a.n_groups = P->n_groups; a.n_levels = P->n_levels; a.d = P->d; a.dc = P->dc;
    a.xld = P->xld; a.cld = P->cld; a.ald = P->ald; a.vld = P->vld; a.sld = P->sld;
    a.WT = P->WT; a.VT = P->VT; a.ST = P->ST;
    a.alpha = P->alpha; a.B = B;
    return a;
}

static int apply(const hint_plan* P, bool rev, const float* params, const float* x, const float* c,
                 float* z, float* J, float* tape, int32_t B, void* stream) {
    if (!P || !params || !x || !z || !J) return fail("hint_block_%s: null argument", rev ? "inverse" : "forward");
    if (P->dc > 0 && !c) return fail("hint_block_%s: plan has dc=%d but c is NULL", rev ? "inverse" : "forward", P->dc);
    if (B < 0) return fail("negative batch");
    if (B == 0) return 0;
    const int ntiles = (B + ROWS - 1) / ROWS;
    const int grid = std::min(ntiles, P->num_cu * 8);
    HIP_TRY(launch_apply(rev, make_args(P, B), P->lds_fwd, grid, params, x, c, z, J, tape, (hipStream_t)stream));
    return 0;
}

int hint_block_forward(const hint_plan* P, const float* params, const float* x, const float* c, float* z,
                       float* J, float* tape, int32_t B, void* stream) {
    return apply(P, false, params, x, c, z, J, tape, B, stream);
}

int hint_block_inverse(const hint_plan* P, const float* params, const float* z, const float* c, float* x,
                       float* J, int32_t B, void* stream) {
    return apply(P, true, params, z, c, x, J, nullptr, B, stream);
}

int hint_block_backward(const hint_plan* P, const float* params, const float* x, const float* tape,
                        const float* c, const float* g_z, const float* g_J, float* g_x, float* g_c, float* g_params,
                        void* workspace, size_t workspace_bytes, int32_t B, void* stream) {
    if (!P || !params || !x || !g_x || !g_params) return fail("hint_block_backward: null argument");
    if (P->n_levels > 1 && !tape && B > 0) return fail("hint_block_backward: tape is NULL but the tree has %d levels", P->n_levels);
    if (P->dc > 0 && !c) return fail("hint_block_backward: plan has dc=%d but c is NULL", P->dc);
    if (B < 0) return fail("negative batch");
    hipStream_t s = (hipStream_t)stream;
    if (B == 0) {
        HIP_TRY(hipMemsetAsync(g_params, 0, (size_t)P->param_floats * sizeof(float), s));
        return 0;
    }
    if (!workspace || workspace_bytes < hint_plan_workspace_bytes(P, B))
        return fail("hint_block_backward: workspace too small (%zu < %zu)", workspace_bytes,
                    hint_plan_workspace_bytes(P, B));
    const size_t Bp = rows_padded(B);
    float* w = (float*)workspace;
    float* wsA1 = w; w += Bp * P->WT + WS_SLACK;
    float* wsA2 = w; w += Bp * P->WT + WS_SLACK;
    float* wsG1 = w; w += Bp * P->WT + WS_SLACK;
    float* wsG2 = w; w += Bp * P->WT + WS_SLACK;
    float* wsV = w;  w += Bp * P->VT + WS_SLACK;
    float* wsG3 = w;

    const int ntiles = (B + ROWS - 1) / ROWS;
    const int grid = std::min(ntiles, P->num_cu * 8);
    if (g_bwd_stages & 1)
        HIP_TRY(launch_bwd(make_args(P, B), P->lds_bwd, grid, params, x, tape, c, g_z, g_J, g_x, g_c, wsV, wsA1,
                           wsA2, wsG1, wsG2, wsG3, s));
    if (!(g_bwd_stages & 2)) return 0;
    // batch split of the weight-gradient GEMMs: enough workgroups to cover the chip ~2x,
    // each wavefront reducing at least 32 rows
    int rows_per_wg = 64;
    {
        const long target_wgs = 2L * P->num_cu;
        long splits = std::max<long>(1, target_wgs / std::max(1, P->n_dwjobs));
        long rp = ((long)Bp + splits - 1) / splits;
        rp = std::max<long>(128, (rp + 63) / 64 * 64);
        rows_per_wg = (int)rp;
    }
    const int splits = (int)((Bp + rows_per_wg - 1) / rows_per_wg);
    // always clear: covers the alignment gaps of the arena and the atomics of the split case
    HIP_TRY(hipMemsetAsync(g_params, 0, (size_t)P->param_floats * sizeof(float), s));
    HIP_TRY(launch_dw(P->d_dwjobs, P->n_dwjobs, splits, wsV, wsA1, wsA2, wsG1, wsG2, wsG3, P->WT, P->VT,
                      P->ST, (int)Bp, rows_per_wg, g_params, s));
    return 0;
}

void hint_debug_set_backward_stages(int32_t mask) { g_bwd_stages = mask & 3; }

int hint_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                   int32_t step, float lr, float beta1, float beta2, float eps, float weight_decay,
                   float grad_scale, float grad_clamp, void* stream) {
    if (!params || !grads || !exp_avg || !exp_avg_sq) return fail("hint_adam_step: null argument");
    if (n < 0 || step < 1) return fail("hint_adam_step: n must be >= 0 and step >= 1");
    if (n == 0) return 0;
    if ((((uintptr_t)params | (uintptr_t)grads | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) != 0)
        return fail("hint_adam_step: buffers must be 16-byte aligned");
    // bias corrections in double like torch.optim.Adam's python scalars
    const double bc1 = 1.0 - std::pow((double)beta1, (double)step);
    const double bc2 = 1.0 - std::pow((double)beta2, (double)step);
    static int num_cu = 0;
    if (num_cu == 0) {
        int dev = 0; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) num_cu = prop.multiProcessorCount;
        if (num_cu <= 0) num_cu = 256;
    }
    HIP_TRY(launch_adam(params, grads, exp_avg, exp_avg_sq, (long)n, (float)((double)lr / bc1), beta1, beta2,
                        (float)(1.0 / std::sqrt(bc2)), eps, weight_decay, grad_scale,
                        grad_clamp > 0.f ? grad_clamp : 3.0e38f, num_cu, (hipStream_t)stream));
    return 0;
}

}  // extern "C"
