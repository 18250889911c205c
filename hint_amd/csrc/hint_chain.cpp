#include "hint_host.hpp"

using namespace hint;

extern "C" {

// ---------------------------------------------------------------------------------------
// chained launches: the blocks of a flow (same plan, own parameters) in one kernel each for the
// forward pass, backward part A and backward part B (+ its slab reduction)
// ---------------------------------------------------------------------------------------
struct hint_chain {
    const hint_plan* plan = nullptr;
    int n = 0, B = 0;
    bool committed = false;
    std::vector<ChainBlock> host;
    std::vector<char> set;
    ChainBlock* d_table = nullptr;
};

static bool chain_any_perm(const hint_chain* C) {
    for (const ChainBlock& b : C->host) if (b.perm != nullptr) return true;
    return false;
}

int hint_chain_create(const hint_plan* P, int32_t n_blocks, int32_t B, hint_chain** out) {
    if (!P || !out) return fail("hint_chain_create: null argument");
    if (n_blocks < 1 || B < 1) return fail("hint_chain_create: n_blocks and B must be >= 1");
    hint_chain* C = new hint_chain();
    C->plan = variant(P, B); C->n = n_blocks; C->B = B;      // (the variant planned for this many row tiles)
    C->host.assign(n_blocks, ChainBlock{});
    C->set.assign(n_blocks, 0);
    if (hipMalloc((void**)&C->d_table, sizeof(ChainBlock) * (size_t)n_blocks) != hipSuccess) {
        delete C;
        return fail("hint_chain_create: hipMalloc failed");
    }
    *out = C;
    return 0;
}

int hint_chain_set_block(hint_chain* C, int32_t i, const float* params, const float* packed, const float* perm,
                         float* tape, void* workspace, size_t workspace_bytes, float* g_params) {
    if (!C || !params || !packed) return fail("hint_chain_set_block: null argument");
    if (i < 0 || i >= C->n) return fail("hint_chain_set_block: block %d out of range (chain has %d)", i, C->n);
    const hint_plan* P = C->plan;
    if (!tape && workspace)
        return fail("hint_chain_set_block: a trainable chain block needs a tape");
    ChainBlock b{};
    b.params = params; b.packed = packed; b.perm = perm; b.gparams = g_params;
    bind_tape(P, C->B, tape, &b);
    if (workspace) {
        if (!g_params) return fail("hint_chain_set_block: workspace without g_params");
        if (workspace_bytes < hint_plan_workspace_bytes(P, C->B))
            return fail("hint_chain_set_block: workspace too small (%zu < %zu)", workspace_bytes,
                        hint_plan_workspace_bytes(P, C->B));
        if (((uintptr_t)workspace & 15) != 0) return fail("hint_chain_set_block: workspace must be 16-byte aligned");
        if (((uintptr_t)g_params & 15) != 0) return fail("hint_chain_set_block: g_params must be 16-byte aligned");
        split_workspace(P, C->B, workspace, &b);
    }
    C->host[i] = b;
    C->set[i] = 1;
    C->committed = false;
    return 0;
}

int hint_chain_set_block_io(hint_chain* C, int32_t i, const float* x_in, const float* c_in, const float* g_add) {
    if (!C) return fail("hint_chain_set_block_io: null argument");
    if (i < 0 || i >= C->n) return fail("hint_chain_set_block_io: block %d out of range (chain has %d)", i, C->n);
    if (!C->set[i]) return fail("hint_chain_set_block_io: hint_chain_set_block(%d) comes first", i);
    if (c_in && C->plan->dc == 0) return fail("hint_chain_set_block_io: the plan has no condition");
    C->host[i].x_in = x_in; C->host[i].c_in = c_in; C->host[i].g_add = g_add;
    C->committed = false;
    return 0;
}

// blocks gathered for part B only (they ran as launches of their own): no chained forward / inverse / part A over them
static bool chain_gathered(const hint_chain* C) {
    for (const ChainBlock& b : C->host) if (b.x_in != nullptr || b.c_in != nullptr) return true;
    return false;
}

int hint_chain_commit(hint_chain* C) {
    if (!C) return fail("hint_chain_commit: null argument");
    for (int i = 0; i < C->n; ++i)
        if (!C->set[i]) return fail("hint_chain_commit: block %d was never set", i);
    HIP_TRY(hipMemcpy(C->d_table, C->host.data(), sizeof(ChainBlock) * (size_t)C->n, hipMemcpyHostToDevice));
    C->committed = true;
    return 0;
}

int hint_chain_forward(const hint_chain* C, const float* x, const float* c, float* z, float* J, const float* J_in,
                       float* loss_acc, void* stream) {
    return hint_chain_forward_noisy(C, x, c, z, J, J_in, loss_acc, 0.f, nullptr, nullptr, stream);
}

int hint_chain_forward_noisy(const hint_chain* C, const float* x, const float* c, float* z, float* J,
                             const float* J_in, float* loss_acc, float noise, const uint64_t* rng_state,
                             float* x_noisy, void* stream) {
    if (!C || !x || !z || !J) return fail("hint_chain_forward: null argument");
    if (!C->committed) return fail("hint_chain_forward: hint_chain_commit() has not been called");
    if (chain_gathered(C)) return fail("hint_chain_forward: the chain's blocks have inputs of their own (hint_chain_set_block_io): it serves part B only");
    const hint_plan* P = C->plan;
    if (P->dc > 0 && !c) return fail("hint_chain_forward: plan has dc=%d but c is NULL", P->dc);
    KArgs a = make_args(P, C->B, false);
    const int nr = wl_nr_for(P, C->B);
    const int lds = lds_with_perms(P, plan_lds(P, false, nr), C->n, chain_any_perm(C), &a);
    if (P->wl) {
        WlArgs w = P->wl_f[nr - 1];
        w.off_perm = a.perm_lds;
        HIP_TRY(launch_wl_apply(false, a, w, lds, grid_for(P, C->B), C->host[0], C->d_table, C->n, x, z, J, J_in, loss_acc, noise,
                                (const unsigned long long*)rng_state, x_noisy, (hipStream_t)stream));
        return 0;
    }
    HIP_TRY(launch_apply(false, P->has_fly != 0, a, lds, grid_for(P, C->B), C->host[0], C->d_table, C->n, x, c, z, J,
                         J_in, loss_acc, noise, (const unsigned long long*)rng_state, x_noisy, (hipStream_t)stream));
    return 0;
}

int hint_chain_inverse(const hint_chain* C, const float* z, const float* c, float* x, float* J, const float* J_in,
                       void* stream) {
    if (!C || !z || !x || !J) return fail("hint_chain_inverse: null argument");
    if (!C->committed) return fail("hint_chain_inverse: hint_chain_commit() has not been called");
    if (chain_gathered(C)) return fail("hint_chain_inverse: the chain's blocks have inputs of their own (hint_chain_set_block_io): it serves part B only");
    const hint_plan* P = C->plan;
    if (P->dc > 0 && !c) return fail("hint_chain_inverse: plan has dc=%d but c is NULL", P->dc);
    KArgs a = make_args(P, C->B, false);
    const int nr = wl_nr_for(P, C->B);
    const int lds = lds_with_perms(P, plan_lds(P, false, nr), C->n, chain_any_perm(C), &a);
    if (P->wl) {
        WlArgs w = P->wl_f[nr - 1];
        w.off_perm = a.perm_lds;
        HIP_TRY(launch_wl_apply(true, a, w, lds, grid_for(P, C->B), C->host[0], C->d_table, C->n, z, x, J, J_in, nullptr, 0.f,
                                nullptr, nullptr, (hipStream_t)stream));
        return 0;
    }
    HIP_TRY(launch_apply(true, P->has_fly != 0, a, lds, grid_for(P, C->B), C->host[0], C->d_table, C->n, z, c, x, J, J_in, nullptr, 0.f,
                         nullptr, nullptr, (hipStream_t)stream));
    return 0;
}

int hint_chain_backward_parts(const hint_chain* C, const float* x, const float* c, const float* g_z, const float* g_J,
                              float* g_x, float* g_c, float gz_scale, float gJ_const, int32_t accumulate,
                              int32_t parts, void* stream) {
    if (!C) return fail("hint_chain_backward: null argument");
    if (!C->committed) return fail("hint_chain_backward: hint_chain_commit() has not been called");
    const hint_plan* P = C->plan;
    if ((parts & 3) == 0) return fail("hint_chain_backward_parts: parts must select part A (1), part B (2) or both (3)");
    if ((parts & 1) && chain_gathered(C)) return fail("hint_chain_backward: the chain's blocks have inputs of their own (hint_chain_set_block_io): part B only");
    if ((parts & 1) && (!g_z || !g_x)) return fail("hint_chain_backward: null argument");
    for (int i = 0; i < C->n; ++i) {
        if (P->dc > 0 && !c && !C->host[i].c_in) return fail("hint_chain_backward: plan has dc=%d but block %d has no condition", P->dc, i);
        if (!C->host[i].wsG1 || !C->host[i].actA1 || !C->host[i].gparams)
            return fail("hint_chain_backward: block %d was set without workspace / g_params", i);
    }
    if (!x && !C->host[0].perm && !C->host[0].x_in) return fail("hint_chain_backward: x is NULL but the first block has no fused permutation");
    return run_backward(P, C->host[0], C->d_table, C->host.data(), C->n, 0, C->n, x, c, g_z, g_J, g_x, g_c, gz_scale, gJ_const, C->B,
                        accumulate ? 1 : 0, parts & 3, (hipStream_t)stream);
}

int hint_chain_wgrad_range(const hint_chain* C, const float* x, const float* c, int32_t accumulate, int32_t block_begin,
                           int32_t block_end, void* stream) {
    if (!C) return fail("hint_chain_wgrad_range: null argument");
    if (!C->committed) return fail("hint_chain_wgrad_range: hint_chain_commit() has not been called");
    if (block_begin < 0 || block_end > C->n || block_begin >= block_end)
        return fail("hint_chain_wgrad_range: blocks [%d, %d) out of range (chain has %d)", block_begin, block_end, C->n);
    const hint_plan* P = C->plan;
    if (!x && !C->host[0].perm && !C->host[0].x_in) return fail("hint_chain_wgrad_range: x is NULL but the first block has no fused permutation");
    for (int i = block_begin; i < block_end; ++i) {
        if (P->dc > 0 && !c && !C->host[i].c_in) return fail("hint_chain_wgrad_range: plan has dc=%d but block %d has no condition", P->dc, i);
        if (!C->host[i].wsG1 || !C->host[i].actA1 || !C->host[i].gparams)
            return fail("hint_chain_wgrad_range: block %d was set without workspace / g_params", i);
    }
    return run_backward(P, C->host[block_begin], C->d_table + block_begin, C->host.data() + block_begin, block_end - block_begin,
                        block_begin, C->n, x, c, nullptr, nullptr, nullptr, nullptr, 1.f, 0.f, C->B, accumulate ? 1 : 0, 2, (hipStream_t)stream);
}

int hint_chain_backward(const hint_chain* C, const float* x, const float* c, const float* g_z, const float* g_J,
                        float* g_x, float* g_c, float gz_scale, float gJ_const, int32_t accumulate, void* stream) {
    return hint_chain_backward_parts(C, x, c, g_z, g_J, g_x, g_c, gz_scale, gJ_const, accumulate, 3, stream);
}

int hint_chain_backward_adam(const hint_chain* C, const float* x, const float* c, const float* g_z, const float* g_J,
                             float* g_x, float* g_c, float gz_scale, float gJ_const, float* params, float* exp_avg,
                             float* exp_avg_sq, int64_t n, const float* opt_state, float beta1, float beta2, float eps,
                             float weight_decay, float grad_scale, float grad_clamp, void* stream) {
    if (!C || !g_z || !g_x || !params || !exp_avg || !exp_avg_sq || !opt_state) return fail("hint_chain_backward_adam: null argument");
    if (!C->committed) return fail("hint_chain_backward_adam: hint_chain_commit() has not been called");
    const hint_plan* P = C->plan;
    if (P->dc > 0 && !c) return fail("hint_chain_backward_adam: plan has dc=%d but c is NULL", P->dc);
    if (!x && !C->host[0].perm) return fail("hint_chain_backward_adam: x is NULL but the first block has no fused permutation");
    if ((((uintptr_t)params | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) != 0)
        return fail("hint_chain_backward_adam: the arenas must be 16-byte aligned");
    for (int i = 0; i < C->n; ++i) {
        const ChainBlock& b = C->host[i];
        if (!b.wsG1 || !b.actA1 || !b.gparams) return fail("hint_chain_backward_adam: block %d was set without workspace / g_params", i);
        const int64_t off = b.params - params;
        if (off < 0 || off + P->param_floats > n || (off & 3) != 0)
            return fail("hint_chain_backward_adam: block %d's parameters are not a 16-byte aligned slice of the arena [params, params + n)", i);
    }
    AdamFuse ad{params, exp_avg, exp_avg_sq, opt_state, beta1, beta2, eps, weight_decay, grad_scale,
                grad_clamp > 0.f ? grad_clamp : 3.0e38f};
    return run_backward(P, C->host[0], C->d_table, C->host.data(), C->n, 0, C->n, x, c, g_z, g_J, g_x, g_c, gz_scale, gJ_const, C->B, 1, 3,
                        (hipStream_t)stream, &ad);
}

int hint_chain_wgrad_adam(const hint_chain* C, const float* x, const float* c, float* params, float* exp_avg, float* exp_avg_sq,
                          int64_t n, const float* opt_state, float beta1, float beta2, float eps, float weight_decay,
                          float grad_scale, float grad_clamp, void* stream) {
    if (!C || !params || !exp_avg || !exp_avg_sq || !opt_state) return fail("hint_chain_wgrad_adam: null argument");
    if (!C->committed) return fail("hint_chain_wgrad_adam: hint_chain_commit() has not been called");
    const hint_plan* P = C->plan;
    if ((((uintptr_t)params | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) != 0)
        return fail("hint_chain_wgrad_adam: the arenas must be 16-byte aligned");
    if (!x && !C->host[0].perm && !C->host[0].x_in) return fail("hint_chain_wgrad_adam: x is NULL but the first block has no fused permutation");
    for (int i = 0; i < C->n; ++i) {
        const ChainBlock& b = C->host[i];
        if (P->dc > 0 && !c && !b.c_in) return fail("hint_chain_wgrad_adam: plan has dc=%d but block %d has no condition", P->dc, i);
        if (!b.wsG1 || !b.actA1 || !b.gparams) return fail("hint_chain_wgrad_adam: block %d was set without workspace / g_params", i);
        const int64_t off = b.params - params;
        if (off < 0 || off + P->param_floats > n || (off & 3) != 0)
            return fail("hint_chain_wgrad_adam: block %d's parameters are not a 16-byte aligned slice of the arena [params, params + n)", i);
    }
    AdamFuse ad{params, exp_avg, exp_avg_sq, opt_state, beta1, beta2, eps, weight_decay, grad_scale,
                grad_clamp > 0.f ? grad_clamp : 3.0e38f};
    return run_backward(P, C->host[0], C->d_table, C->host.data(), C->n, 0, C->n, x, c, nullptr, nullptr, nullptr, nullptr, 1.f, 0.f, C->B, 1, 2,
                        (hipStream_t)stream, &ad);
}

void hint_chain_destroy(hint_chain* C) {
    if (!C) return;
    (void)hipFree(C->d_table);
    delete C;
}

}  // extern "C"
