// hint_block_inverse_backward (include/hint_amd.h): gradients through the inverse direction of a block.
#include "hint_host.hpp"

using namespace hint;

extern "C" {

// ---------------------------------------------------------------------------------------
// backward of the INVERSE direction (hint.py:82-88 under autograd), level by level on the block kernels.
// With y_L = x (times the node permutations) and y_l = level_l(y_{l+1}) the forward direction rebuilds, deepest level
// first, what the inverse saw at every node; the inverse coupling's derivative is the forward coupling's with the roles
// turned round:  g_z2 = g_x2 / e(s);  (g_s, g_t) and with them every subnet, weight and condition gradient of the level
// are MINUS what the forward's backward returns for the upstream pair (g_z2, g_J);  g_z1 = g_x1 - (its gradient on the
// conditioning lanes).  e(s) itself is the lane gradient of a row-parallel backward launch with g = 1 on the level's
// transformed lanes (1 * e: exact).
// ---------------------------------------------------------------------------------------
static int inv_levels(const hint_plan* Pc) {
    hint_plan* P = const_cast<hint_plan*>(Pc);
    std::lock_guard<std::mutex> lock(P->inv_mu);
    if (!P->inv_levels.empty()) return 0;
    int depth = 0;
    for (const hint_node_desc& n : P->nodes) depth = std::max(depth, n.depth + 1);
    std::vector<hint_plan*> levels;
    std::vector<uint8_t> lower;
    for (int lev = 0; lev < depth; ++lev) {
        std::vector<hint_node_desc> sel;
        for (hint_node_desc n : P->nodes)
            if (n.depth == lev) { n.depth = 0; sel.push_back(n); }
        if (sel.empty()) continue;
        hint_plan* L = nullptr;
        if (hint_plan_create(sel.data(), (int32_t)sel.size(), P->d, P->dc, P->clamp, &L) != 0) {
            for (hint_plan* q : levels) hint_plan_destroy(q);
            return 1;                       // (hint_plan_create's message stands)
        }
        levels.push_back(L);
        lower.resize(levels.size() * (size_t)P->d, 0);
        for (const hint_node_desc& n : sel)
            for (int j = n.off + n.k; j < n.off + n.D; ++j) lower[(levels.size() - 1) * (size_t)P->d + j] = 1;
    }
    if (hipMalloc((void**)&P->d_inv_lower, lower.size()) != hipSuccess ||
        hipMemcpy(P->d_inv_lower, lower.data(), lower.size(), hipMemcpyHostToDevice) != hipSuccess) {
        for (hint_plan* q : levels) hint_plan_destroy(q);
        (void)hipFree(P->d_inv_lower);
        P->d_inv_lower = nullptr;
        return fail("hint_block_inverse_backward: device allocation failed");
    }
    P->inv_levels = levels;
    return 0;
}

struct InvWs { size_t y0, y1, g, e, t, gin, J, gc, tape, packed, gneg, bws, bws_bytes, total_bytes; };
static InvWs inv_ws(const hint_plan* P, int B) {
    auto al = [](size_t floats) { return (floats + 3) & ~(size_t)3; };
    const size_t lane = al((size_t)B * P->d);
    InvWs w{};
    size_t off = 0, tape = 0, packed = 0;
    w.y0 = off; off += lane;  w.y1 = off; off += lane;  w.g = off; off += lane;  w.e = off; off += lane;
    w.t = off; off += lane;   w.gin = off; off += lane;
    w.J = off; off += al((size_t)B);
    w.gc = off; off += al((size_t)B * P->dc);
    for (const hint_plan* L : P->inv_levels) {
        tape = std::max(tape, (size_t)hint_plan_tape_floats(L, B));
        packed = std::max(packed, (size_t)hint_plan_packed_floats(L));
        w.bws_bytes = std::max(w.bws_bytes, hint_plan_workspace_bytes(L, B));
    }
    w.tape = off; off += al(tape);
    w.packed = off; off += al(packed);
    w.gneg = off; off += al((size_t)P->param_floats);
    w.bws = off; off += al((w.bws_bytes + 3) / 4);
    w.total_bytes = off * sizeof(float);
    return w;
}

size_t hint_plan_inverse_workspace_bytes(const hint_plan* P, int32_t B) {
    if (!P || B <= 0 || inv_levels(P) != 0) return 0;
    return inv_ws(P, B).total_bytes;
}

int hint_block_inverse_backward(const hint_plan* P, const float* params, const float* x, const float* c, const float* g_x,
                                const float* g_J, float* g_z, float* g_c, float* g_params, int32_t accumulate,
                                void* workspace, size_t workspace_bytes, const float* perm, int32_t B, void* stream) {
    if (!P || !params || !x || !g_z || !g_params) return fail("hint_block_inverse_backward: null argument");
    if (P->dc > 0 && !c) return fail("hint_block_inverse_backward: plan has dc=%d but c is NULL", P->dc);
    if (B < 0) return fail("negative batch");
    if (((uintptr_t)g_params & 15) != 0) return fail("hint_block_inverse_backward: g_params must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    if (B == 0) {
        if (!accumulate) HIP_TRY(launch_zero(g_params, (long)P->param_floats, P->num_cu, s));
        return 0;
    }
    if (inv_levels(P) != 0) return 1;
    const InvWs w = inv_ws(P, B);
    if (!workspace || workspace_bytes < w.total_bytes)
        return fail("hint_block_inverse_backward: workspace too small (%zu < %zu)", workspace_bytes, w.total_bytes);
    if (((uintptr_t)workspace & 15) != 0) return fail("hint_block_inverse_backward: workspace must be 16-byte aligned");
    float* W = (float*)workspace;
    const long n = (long)B * P->d;
    const int cu = P->num_cu;
    float* ybuf[2] = {W + w.y0, W + w.y1};
    float *g = W + w.g, *e = W + w.e, *t = W + w.t, *gin = W + w.gin, *gc = W + w.gc, *gneg = W + w.gneg;
    const float* y = x;
    int cur = 0;
    if (g_x == nullptr) HIP_TRY(launch_zero(g, n, cu, s));
    if (perm != nullptr) {              // x = y P^T behind the inverse (hint.py:93-94): y = x P, g_y = g_x P
        HIP_TRY(launch_inv_rowmat(x, perm, ybuf[0], n, P->d, cu, s));
        y = ybuf[0];
        cur = 1;
        if (g_x != nullptr) HIP_TRY(launch_inv_rowmat(g_x, perm, g, n, P->d, cu, s));
    } else if (g_x != nullptr) {
        HIP_TRY(hipMemcpyAsync(g, g_x, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, s));
    }
    HIP_TRY(launch_zero(gneg, (long)P->param_floats, cu, s));
    if (g_c != nullptr) HIP_TRY(launch_zero(g_c, (long)B * P->dc, cu, s));
    for (size_t li = P->inv_levels.size(); li-- > 0;) {
        const hint_plan* L = P->inv_levels[li];
        const uint8_t* lower = P->d_inv_lower + li * (size_t)P->d;
        float* yup = ybuf[cur];
        if (hint_block_pack(L, params, W + w.packed, stream) != 0) return 1;
        if (hint_block_forward(L, params, W + w.packed, y, c, yup, W + w.J, W + w.tape, B, stream) != 0) return 1;
        HIP_TRY(launch_inv_lane(0, t, nullptr, nullptr, nullptr, lower, n, P->d, cu, s));
        if (block_backward(L, params, W + w.packed, y, W + w.tape, c, t, nullptr, e, nullptr, gneg, 1, W + w.bws, w.bws_bytes,
                           nullptr, 1.f, 0.f, B, 1, stream) != 0) return 1;
        HIP_TRY(launch_inv_lane(1, t, g, e, nullptr, lower, n, P->d, cu, s));
        if (block_backward(L, params, W + w.packed, y, W + w.tape, c, t, g_J, gin, g_c ? gc : nullptr, gneg, 1, W + w.bws,
                           w.bws_bytes, nullptr, 1.f, 0.f, B, 3, stream) != 0) return 1;
        HIP_TRY(launch_inv_lane(2, g, g, t, gin, lower, n, P->d, cu, s));
        if (g_c != nullptr) HIP_TRY(launch_inv_minus(g_c, gc, (long)B * P->dc, 1, cu, s));
        y = yup;
        cur ^= 1;
    }
    HIP_TRY(hipMemcpyAsync(g_z, g, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, s));
    HIP_TRY(launch_inv_minus(g_params, gneg, (long)P->param_floats, accumulate ? 1 : 0, cu, s));
    return 0;
}

}  // extern "C"
