/* hint_amd.h — C ABI of the MI355X (gfx950) implementation of HINT's recursive
 * affine-coupling block: forward / inverse / log|det J| and the backward pass.
 *
 * This is the drop-in boundary.  The reference (vislearn/HINT) has no native code; the
 * interface replaced here is the Python module protocol of
 *     /root/reference/hint.py:104-133   HierarchicalAffineCouplingBlock
 *         .forward(x:list, c=[], rev=False) -> [Tensor]      (hint.py:124-126)
 *         .jacobian(x, c=[], rev=False)     -> Tensor[B]     (hint.py:128-129)
 *     /root/reference/hint.py:21-101    HierarchicalAffineCouplingTree (the arithmetic)
 * plus the autograd backward PyTorch derives from it when the training loop calls
 * loss.backward() (/root/reference/train_unconditional.py:137).
 * `hint_amd/hint.py` is the host-side mirror of that module and binds these symbols with
 * ctypes; INTEGRATION.md shows the binding a reference maintainer would add.
 *
 * Conventions
 *   - plain C, no C++/torch types; every function returns 0 on success, non-zero on error
 *     (hint_last_error() gives a thread-local message); no exceptions cross the boundary.
 *   - all tensors are fp32, row-major, caller-allocated DEVICE memory: x,z,g_x,g_z [B,d];
 *     c,g_c [B,dc] (all conditions concatenated, hint.py:76); J,g_J [B].
 *   - calls are asynchronous and ordered on `stream` (a hipStream_t passed as void*).
 *   - a plan is immutable after creation and bound to the HIP device current at creation;
 *     it may be shared by threads.  Parameters are passed per call as ONE flat fp32 buffer
 *     whose layout the caller described at plan creation (p_off, in floats), because the
 *     reference loops rebind p.data / call .to() (train_unconditional.py:165-167).
 */
#ifndef HINT_AMD_H
#define HINT_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HINT_AMD_ABI_VERSION 7

/* index into hint_node_desc.p_off: [net][tensor]; net 0 = s, net 1 = t (hint.py:44-45);
 * tensors in nn.Sequential order (hint.py:11-13): W1 [h,cin], b1 [h], W2 [h,h], b2 [h],
 * W3 [r,h], b3 [r]; all row-major [out,in] like torch.nn.Linear.weight. */
enum { HINT_W1 = 0, HINT_B1 = 1, HINT_W2 = 2, HINT_B2 = 3, HINT_W3 = 4, HINT_B3 = 5 };

/* One node of the coupling tree (hint.py:25-54), in any order.  A node owns lanes
 * [off, off+D); its first k lanes condition the transform of the other r = D-k >= 1
 * (hint.py:41,68 always split at k = D/2; other splits are accepted, e.g. k = 0 for a coupling
 * that transforms all its lanes given the condition only, conditional_hint_4_full.py:76-82).
 * depth = 0 for the root.  Nodes of equal depth own disjoint lanes. */
typedef struct hint_node_desc {
    int32_t off, D, k, r;
    int32_t h;          /* hidden width of both subnets (hint.py:44-45) */
    int32_t depth;
    int64_t p_off[12];  /* offsets in floats into the flat parameter buffer: [net*6 + tensor] */
} hint_node_desc;

typedef struct hint_plan hint_plan;

/* Build the static level schedule for one block.  d = lanes of the block, dc = total
 * condition width (0 if unconditional), clamp as in hint.py:108 (alpha = clamp*0.636,
 * hint.py:57,60).  Replaces HierarchicalAffineCouplingTree.__init__ (hint.py:25-54). */
int hint_plan_create(const hint_node_desc* nodes, int32_t n_nodes, int32_t d, int32_t dc,
                     float clamp, hint_plan** out);

/* Host-only dry run of hint_plan_create (no device needed, nothing uploaded): builds the plan, lets the
 * planner verify its own schedule (every fragment tile of every group in exactly one wavefront's range of
 * either GEMM phase, slices and slabs consistent with the ranges) and reports what it came to:
 * stats[16] = { groups, levels, WT (activation columns), ST (coupling-gradient columns), LDS bytes forward,
 * LDS bytes backward, wavefronts per workgroup, part-B tile jobs, parameter floats, packed floats, units,
 * fragment tiles of the widest group, subtree groups (the deepest levels that run one subtree per wavefront; 0: none),
 * 1 when the block runs on the wave-local kernels, single-tile part-B jobs that share workgroups,
 * slots (threads per batch row) of the backward's widest boundary }.
 * For tests and tools; same return convention as hint_plan_create. */
int hint_plan_check(const hint_node_desc* nodes, int32_t n_nodes, int32_t d, int32_t dc, float clamp,
                    int64_t* stats);
void hint_plan_destroy(hint_plan* plan);

/* floats the flat parameter (and gradient) buffer must hold: max(p_off + tensor size), rounded up to 4. */
/* (A plan may hold two launch variants of the block - 8 wavefronts per workgroup for batches of up to one 16-row tile
 * per CU, 4 for larger ones - and every entry point that takes B picks by B; the sizes below are the picked
 * variant's, so query them with the B you will run.) */
int64_t hint_plan_param_floats(const hint_plan* plan);
/* floats of the packed-weight buffer (both subnets of every node, forward and transposed
 * copies, in MFMA fragment order, zero padded). */
int64_t hint_plan_packed_floats(const hint_plan* plan);
/* floats of the forward "tape" for a batch of B rows, recorded by the training forward and read
 * by the backward pass: per tree level one [B,d] snapshot of the lane tensor as that level saw it
 * (the last slice holds the block's permuted input for the _ex / chain forms) and one [B,d] array
 * of the level's coupling arguments s (indexed by the lane each one scales); the hidden activations
 * a2 of every subnet, [B rounded up to 16, sum of 2*pad16(h)] (and a1, the same size, unless every
 * subnet of the block has 1..4 inputs, at most 4 outputs and no condition; tree levels of that kind -
 * "lean" groups - leave their a1 columns unwritten in any case: a1 is rebuilt where it is needed); and one sign byte per four activations (what the backward kernel reads instead
 * of the activations).  The backward pass recomputes nothing else (what autograd keeps for hint.py:77,
 * minus the pre-activations), and the weight-gradient kernel takes its a2 / lane operands from here. */
int64_t hint_plan_tape_floats(const hint_plan* plan, int32_t B);
/* bytes of scratch hint_block_backward needs for a batch of B rows: the per-row gradient factors g1, g2
 * ([B rounded up to 16, sum of 2*pad16(h)] each; neither for lean plans, whose backward kernel computes the
 * first-layer gradients itself into one small slab per workgroup) and g_s | g_t, and one partial-gradient
 * slab per batch split of the weight-gradient kernel. */
size_t hint_plan_workspace_bytes(const hint_plan* plan, int32_t B);
/* dynamic LDS bytes per workgroup of the forward / backward kernels (informational). */
int32_t hint_plan_lds_bytes(const hint_plan* plan, int32_t backward);
/* Which kernels a batch of B rows runs on (diagnostics, bench labels): out[0] = 1 when the block runs on the wave-local
 * kernels (hint_wl_apply_kernel / hint_wl_bwd_kernel: every subnet has 1..4 inputs and at most 4 outputs), 0 for the
 * general ones (hint_apply_kernel / hint_bwd_kernel); out[1] = 16-row tiles per workgroup (1, or 2: row pairs);
 * out[2] = wavefronts per workgroup; out[3] = 1 when no a1 / g2 arrays exist (part B rebuilds them); out[4] = subtree groups (the
 * deepest levels that run one subtree per wavefront); out[5] = tiles of the widest row (<= 3: the general backward pass runs on
 * hint_bwd_kernel_n3, unless out[6]); out[6] = 1 when rows of the backward kernel compute first-layer weight gradients themselves;
 * out[7] = 1 when some general group is lean: forward and inverse run on hint_apply_kernel<REV, true>, whose rows make such groups'
 * first layer themselves (no thin phase), else on hint_apply_kernel<REV, false>.  out must hold 8 values. */
int hint_plan_describe(const hint_plan* plan, int32_t B, int32_t* out);

/* Re-pack the flat parameters into `packed` (hint_plan_packed_floats floats).  Must be called
 * after every change of the parameters and before the next forward / inverse / backward that
 * should see it; one small launch (the weights of a block are a few hundred KiB). */
int hint_block_pack(const hint_plan* plan, const float* params, float* packed, void* stream);

/* The same for several blocks in ONE launch (a trainer re-packs every block of the flow after
 * each optimizer step): the group records plan / params / packed pointers of n blocks once. */
typedef struct hint_pack_group hint_pack_group;
int hint_pack_group_create(const hint_plan* const* plans, const float* const* params,
                           float* const* packed, int32_t n, hint_pack_group** out);
int hint_pack_group_run(const hint_pack_group* group, void* stream);
/* The same launch as the prologue of a training step: additionally clears zero_floats floats at
 * zero_buf (the loss sums of the step before; NULL/0 = nothing), adds 1 to rng_state[1] (the step
 * counter t of hint_chain_forward_noisy; NULL = no counter) and, if opt_state is given (device
 * float[5] = {lr, beta1, beta2, out, out}; needs rng_state), writes Adam's factors of step t,
 * opt_state[3] = lr / (1 - beta1^t) and opt_state[4] = 1 / sqrt(1 - beta2^t), for
 * hint_adam_step_dev. */
int hint_pack_group_run_ex(const hint_pack_group* group, float* zero_buf, int32_t zero_floats,
                           uint64_t* rng_state, float* opt_state, void* stream);
void hint_pack_group_destroy(hint_pack_group* group);

/* z, J = block(x | c), rev=False (hint.py:62-80,90,97-99).  c may be NULL iff dc == 0.
 * params: flat parameters (biases are read from here); packed: output of hint_block_pack.
 * tape: NULL for inference, else hint_plan_tape_floats(plan, B) floats (training). */
int hint_block_forward(const hint_plan* plan, const float* params, const float* packed,
                       const float* x, const float* c, float* z, float* J, float* tape, int32_t B,
                       void* stream);
/* x, J = block(z | c), rev=True: own coupling undone first, then children
 * (hint.py:82-88); J is the NEGATED log-det like the reference returns it (hint.py:83). */
int hint_block_inverse(const hint_plan* plan, const float* params, const float* packed,
                       const float* z, const float* c, float* x, float* J, int32_t B, void* stream);
/* Backward of hint_block_forward.  Takes the block INPUT x and the tape the forward call
 * recorded (required), upstream g_z [B,d] and
 * g_J [B] (either may be NULL = zeros).  Writes g_x [B,d], g_c [B,dc] (may be NULL) and the
 * flat parameter gradient g_params (same layout as params): overwritten when accumulate == 0,
 * added to when accumulate != 0 (the caller then owns zeroing, e.g. hint_adam_step's
 * zero_grads); 16-byte aligned.  Every gradient is reduced in a fixed order (per-split slabs, then one
 * reduction pass; no float atomics): the same inputs give bit-identical gradients.  workspace:
 * hint_plan_workspace_bytes(plan, B) bytes, 16-byte aligned device scratch. */
int hint_block_backward(const hint_plan* plan, const float* params, const float* packed,
                        const float* x, const float* tape, const float* c, const float* g_z,
                        const float* g_J, float* g_x, float* g_c, float* g_params,
                        int32_t accumulate, void* workspace, size_t workspace_bytes, int32_t B,
                        void* stream);

/* Backward of hint_block_inverse / hint_block_inverse_ex: what autograd derives when `rev=True` runs with gradients
 * (hint.py:82-88 and the recursion of :85-88 are differentiable torch ops; train_unconditional.py:152-153 only samples
 * under no_grad, so no reference loop needs it - it is here for users of the module who do).
 *   x        [B,d] the OUTPUT of the inverse call (with the same perm, if any); c as given to it.
 *   g_x, g_J upstream gradients of the inverse's outputs (either may be NULL = zeros).
 *   g_z [B,d], g_c [B,dc] (may be NULL), g_params (flat, same layout as params; overwritten when accumulate == 0, added
 *   to otherwise; 16-byte aligned) receive the gradients.
 *   perm     the matrix given to hint_block_inverse_ex (x = block^-1(z) @ perm^T), or NULL.
 * Runs level by level on the block kernels, deepest level first: the forward direction rebuilds from x what the
 * inverse saw at every node, and each level's derivative is the forward coupling's with the roles turned round
 * (g_z2 = g_x2 / e(s), all subnet gradients with the opposite sign; derivation: DESIGN.md section 1).  The first call
 * on a plan builds one plan per tree level (device tables; not stream-ordered, like hint_plan_create).
 * workspace: hint_plan_inverse_workspace_bytes(plan, B) bytes of 16-byte aligned device scratch (that call builds the
 * level plans as well; returns 0 on failure, see hint_last_error). */
size_t hint_plan_inverse_workspace_bytes(const hint_plan* plan, int32_t B);
int hint_block_inverse_backward(const hint_plan* plan, const float* params, const float* x, const float* c,
                                const float* g_x, const float* g_J, float* g_z, float* g_c, float* g_params,
                                int32_t accumulate, void* workspace, size_t workspace_bytes, const float* perm,
                                int32_t B, void* stream);

/* Chained forms used by the flow container / trainer (hint_amd/flow.py, hint_amd/train.py): the
 * work FrEIA's graph does between two blocks is folded into the block kernels.
 *   perm     [d,d] row-major fixed orthogonal matrix of the permutation node in front of the
 *            block (power_hint_8.py:59-62): forward computes block(x @ perm), inverse returns
 *            block^-1(z) @ perm^T, backward returns g_x @ perm^T.  NULL = none.  With perm the
 *            forward stores the permuted input in the last [B,d] slice of the tape and backward
 *            reads it from there (x may be NULL).
 *   J_in     [B] log-det accumulated by the preceding blocks, added to this block's J
 *            (ReversibleGraphNet.log_jacobian sums the nodes); NULL = 0.
 *   loss_acc float[64][2], accumulated atomically (workgroup b adds to slot b % 64, the caller
 *            sums the slots): [.][0] += sum_rows 0.5*|z|^2, [.][1] += sum_rows J (the two loss
 *            terms of train_unconditional.py:128-129 before the .mean()); NULL = skip.
 *   gz_scale multiplies g_z on load (pass z and 1/B for the first term's gradient);
 *   gJ_const used for every row when g_J is NULL (-1/B for the second term). */
int hint_block_forward_ex(const hint_plan* plan, const float* params, const float* packed,
                          const float* x, const float* c, float* z, float* J, float* tape,
                          const float* perm, const float* J_in, float* loss_acc, int32_t B,
                          void* stream);
int hint_block_inverse_ex(const hint_plan* plan, const float* params, const float* packed,
                          const float* z, const float* c, float* x, float* J, const float* perm,
                          const float* J_in, int32_t B, void* stream);
int hint_block_backward_ex(const hint_plan* plan, const float* params, const float* packed,
                           const float* x, const float* tape, const float* c, const float* g_z,
                           const float* g_J, float* g_x, float* g_c, float* g_params,
                           int32_t accumulate, void* workspace, size_t workspace_bytes,
                           const float* perm, float gz_scale, float gJ_const, int32_t B,
                           void* stream);

/* Whole-flow launches.  A chain is n_blocks blocks of ONE plan (the configs stack identical
 * blocks, power_hint_8.py:56-67), each with its own parameters, optional permutation in front,
 * tape and backward workspace, laid out for a fixed batch size B.  hint_chain_forward runs
 * all blocks in one kernel (the lane tile of a row stays on chip from block to block; what
 * ReversibleGraphNet.forward does node by node), hint_chain_backward runs the row-parallel
 * part of all blocks in one kernel and all weight gradients in a second one.
 *   hint_chain_set_block: pointers of block i; they are captured, not copied, and must stay
 *     valid.  tape: hint_plan_tape_floats(plan, B) floats (required for training).  workspace /
 *     g_params may be NULL for an inference-only chain.
 *   hint_chain_commit: uploads the table (synchronous); call after the last set_block and
 *     before the first forward/backward (and again after changing a block).
 *   forward: z [B,d], J [B] = sum of the blocks' log-dets (+ J_in); loss_acc as in
 *     hint_block_forward_ex.  x may alias z.
 *   backward: arguments as in hint_block_backward_ex; x is only read when block 0 has no
 *     permutation.  g_c accumulates over the blocks (all blocks see the same c). */
typedef struct hint_chain hint_chain;
int hint_chain_create(const hint_plan* plan, int32_t n_blocks, int32_t B, hint_chain** out);
int hint_chain_set_block(hint_chain* chain, int32_t i, const float* params, const float* packed,
                         const float* perm, float* tape, void* workspace, size_t workspace_bytes,
                         float* g_params);
int hint_chain_commit(hint_chain* chain);
int hint_chain_forward(const hint_chain* chain, const float* x, const float* c, float* z, float* J,
                       const float* J_in, float* loss_acc, void* stream);
/* hint_chain_forward on x + noise * N(0,1) (train_unconditional.py:121), the noise drawn inside
 * the kernel: Philox4x32-10 keyed by rng_state = {seed, step} (device memory, read only here;
 * hint_pack_group_run_ex advances step) and the element index, Box-Muller.  x_noisy [B,d]
 * (may be NULL) receives the perturbed input, which is what hint_chain_backward must be given
 * as x.  rng_state == NULL: no noise. */
int hint_chain_forward_noisy(const hint_chain* chain, const float* x, const float* c, float* z,
                             float* J, const float* J_in, float* loss_acc, float noise,
                             const uint64_t* rng_state, float* x_noisy, void* stream);
int hint_chain_backward(const hint_chain* chain, const float* x, const float* c, const float* g_z,
                        const float* g_J, float* g_x, float* g_c, float gz_scale, float gJ_const,
                        int32_t accumulate, void* stream);
/* The same with the two halves of the backward pass selectable per call (profiling, or overlapping
 * part B with other work): parts bit 0 = the row-parallel kernel (g_x, g_c and the per-row factors in
 * the workspace), bit 1 = the weight-gradient kernels (read what bit 0 left in the workspace). */
int hint_chain_backward_parts(const hint_chain* chain, const float* x, const float* c, const float* g_z,
                              const float* g_J, float* g_x, float* g_c, float gz_scale, float gJ_const,
                              int32_t accumulate, int32_t parts, void* stream);
/* Part B (the weight-gradient kernels) for the blocks [block_begin, block_end) of the chain only: a data-parallel
 * step finishes the gradient of the last blocks first and starts their all-reduce while the rest of part B runs
 * (one bucket per call; the blocks' gradients are whatever g_params slices hint_chain_set_block was given).  Part A
 * (hint_chain_backward_parts, parts = 1) must have run. */
int hint_chain_wgrad_range(const hint_chain* chain, const float* x, const float* c, int32_t accumulate,
                           int32_t block_begin, int32_t block_end, void* stream);
/* Sampling direction (train_unconditional.py:152-153, rev=True through the whole graph): the blocks
 * of the chain last to first in ONE launch, x = chain^-1(z), J = J_in - sum of the blocks' log-dets
 * (the reference's rev=True sign, hint.py:83).  x may alias z. */
int hint_chain_inverse(const hint_chain* chain, const float* z, const float* c, float* x, float* J,
                       const float* J_in, void* stream);
/* hint_chain_backward with the optimizer folded into the weight gradients' final reduction (one process: nothing sits
 * between backward and the step - train_unconditional.py:137-144): every parameter element takes its clamp + Adam step
 * (hint_adam_step_dev's arithmetic, bit for bit) the moment its gradient is summed, and the gradient arena is neither read
 * nor written.  params / exp_avg / exp_avg_sq: arenas of n floats that hold every block's parameter slice (the blocks'
 * params pointers must point into [params, params + n)); elements outside the blocks' slices are not touched. */
int hint_chain_backward_adam(const hint_chain* chain, const float* x, const float* c, const float* g_z, const float* g_J,
                             float* g_x, float* g_c, float gz_scale, float gJ_const, float* params, float* exp_avg,
                             float* exp_avg_sq, int64_t n, const float* opt_state, float beta1, float beta2, float eps,
                             float weight_decay, float grad_scale, float grad_clamp, void* stream);
void hint_chain_destroy(hint_chain* chain);

/* ---- modules that run as launches of their own and share ONE part B (round 5; abi 6) ----
 * The conditional two-lane model (configs/plus_shape/conditional_hint_4_full.py:58-94) is a graph, not a chain: its x lane's
 * blocks take the y lane as their condition, so every module's forward and row-parallel backward is a launch of its own.
 * Their weight gradients need not be: the modules of one plan are gathered in a chain whose blocks carry their OWN level-0
 * input (x_in: what the module's forward was given; NULL when it had a fused permutation - the tape holds the permuted input)
 * and condition (c_in), and hint_chain_wgrad_range / hint_chain_wgrad_adam run part B and its slab reduction for all of them
 * in one launch each.  A chain with such blocks serves part B only.  g_add (any chain): a [B, d] gradient the backward's part A
 * adds to block i's input gradient before it goes back through the block's fused permutation - the block's permuted input had
 * a second consumer (the y lane after its permutation is also the x lane's condition, train_conditional.py:50-55 graph).
 * Call between hint_chain_set_block(i) and hint_chain_commit. */
int hint_chain_set_block_io(hint_chain* chain, int32_t i, const float* x_in, const float* c_in, const float* g_add);
/* part B + slab reduction of every block of the chain with the clamp + Adam step in the reduction (hint_chain_backward_adam
 * without part A): the row-parallel launches (hint_block_backward_rows, hint_chain_backward_parts(.., parts = 1, ..)) have
 * left the per-row factors in the blocks' workspaces. */
int hint_chain_wgrad_adam(const hint_chain* chain, const float* x, const float* c, float* params, float* exp_avg,
                          float* exp_avg_sq, int64_t n, const float* opt_state, float beta1, float beta2, float eps,
                          float weight_decay, float grad_scale, float grad_clamp, void* stream);
/* hint_block_forward_ex with the dequantisation noise of train_conditional.py:121 drawn in the kernel (Philox4x32-10 keyed by
 * rng_state = {seed, step}; x_noisy [B,d] receives the perturbed input the backward pass starts from); noise = 0: plain. */
int hint_block_forward_noisy(const hint_plan* plan, const float* params, const float* packed, const float* x,
                             const float* c, float* z, float* J, float* tape, const float* perm,
                             const float* J_in, float* loss_acc, float noise, const uint64_t* rng_state,
                             float* x_noisy, int32_t B, void* stream);
/* the row-parallel part of hint_block_backward_ex alone: g_x, g_c, and the per-row factors of the weight gradients in
 * `workspace` (hint_plan_workspace_bytes) for a later part B over a chain that holds this block with that workspace. */
int hint_block_backward_rows(const hint_plan* plan, const float* params, const float* packed, const float* x,
                             const float* tape, const float* c, const float* g_z, const float* g_J, float* g_x,
                             float* g_c, void* workspace, size_t workspace_bytes, const float* perm, float gz_scale,
                             float gJ_const, int32_t B, void* stream);

/* Fused gradient clamp + Adam step over a flat fp32 arena of n parameters; replaces
 *   for p in params: p.grad.data.clamp_(-5, 5)        (train_unconditional.py:140-141)
 *   torch.optim.Adam(..., eps, weight_decay).step()    (train_unconditional.py:144,174-176)
 * g' = clamp(grads*grad_scale, +-grad_clamp) + weight_decay*p, then the standard Adam update
 * with bias correction for the 1-based `step`.  grad_scale = 1/world_size turns the summed
 * all-reduce into the mean BEFORE the clamp; grad_clamp <= 0 disables clamping.  With
 * zero_grads != 0 the gradient arena is cleared after it has been consumed.  All four buffers
 * must be 16-byte aligned. */
int hint_adam_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                   int32_t step, float lr, float beta1, float beta2, float eps, float weight_decay,
                   float grad_scale, float grad_clamp, int32_t zero_grads, void* stream);
/* The same step with its step-dependent factors read from device memory (opt_state as written by
 * hint_pack_group_run_ex), so that the launch carries no per-step host argument and can be
 * captured in a graph together with the kernels of the step. */
int hint_adam_step_dev(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                       const float* opt_state, float beta1, float beta2, float eps,
                       float weight_decay, float grad_scale, float grad_clamp, int32_t zero_grads,
                       void* stream);

int hint_abi_version(void);
const char* hint_last_error(void);
/* what the library binary was built with and runs with: "libhint_amd abi N, gfx950, HIP x.y.z, clang ..., src <12 hex digits: hash
 * of the sources it was compiled from>[, knobs: HINT_X=v ...]" - the last part lists the HINT_* environment variables the library
 * found set (they change which kernels run).  Valid until the thread's next call; the HIP runtime of the machine that loads the
 * library may differ - bench.py prints both. */
const char* hint_build_info(void);

/* Diagnostics (process-wide; no effect on any result).  The general kernels touch the packed weights of what runs two
 * phases later into an LDS sink nobody reads (an L2 warm-up; on unless the environment says HINT_PF=0 when the library
 * first launches): hint_debug_set_prefetch(0 / 1) switches it for the launches that follow and returns the previous
 * setting - a launch already captured in a hipGraph keeps what it was captured with.  hint_debug_last_lds_bytes(backward)
 * = dynamic LDS bytes of the process's last forward / inverse (0) or backward part-A (1) launch. */
int hint_debug_set_prefetch(int on);
/* The library reads its HINT_* environment variables once (the list: INTEGRATION.md); this re-reads them - for tests and A/B tools
 * that change the environment inside one process.  Plans already built keep what they were built with. */
int hint_debug_reload_knobs(void);
int32_t hint_debug_last_lds_bytes(int32_t backward);

#ifdef __cplusplus
}
#endif
#endif /* HINT_AMD_H */
