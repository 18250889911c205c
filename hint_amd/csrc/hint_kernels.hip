// HIP kernels (gfx950 / CDNA4 only) for HINT's recursive affine-coupling block.
//
// Arithmetic reproduced (reference, read-only): /root/reference/hint.py:62-101
//   per node:  v = [u | c];  s = mlp_s(v), t = mlp_t(v)            (hint.py:76-77, :10-13)
//              a = alpha*atan(s), alpha = clamp*0.636               (hint.py:56-60)
//   forward    l' = exp(a)*l + t ;  J += sum a   (children first)   (hint.py:70-80,97-99)
//   inverse    l  = (l' - t)/exp(a); J -= sum a  (root first)       (hint.py:82-88)
//
// Design: samples are independent, so one workgroup owns a tile of 16 batch
// rows (= one M-tile of v_mfma_f32_16x16x4_f32) and carries it through ALL tree levels of
// the block inside one launch; the lane tile, the conditioning input, both hidden
// activations and s/t live in LDS for the whole pass, HBM sees x once in and z, J once out.
// The four wavefronts split the hidden units (N-tiles) of the s- and t-subnets of every node
// of a level; weights stream straight from L2 into MFMA B-fragments (each weight is used
// once per row tile, so staging them in LDS would buy nothing).  fp32 MFMA is an exact
// fp32 FMA chain, so results differ from the CPU reference only by summation order.
#include <hip/hip_runtime.h>
#include "hint_dev.h"

using namespace hint;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// ---------------------------------------------------------------------------------------
// 16x16 output tile, A from LDS (16 rows, row stride lda, columns [acol, acol+pad16(K)) zero
// padded), B from global weights.  MFMA lane map (16x16x4 f32): lane l supplies
// A[m = l&15][kslot = l>>4] and B[kslot][n = l&15]; result reg i = C[4*(l>>4)+i][l&15].
// The reduction index is permuted consistently on both operands (slot kq of step i of a
// 16-wide block is k = kb + 4*kq + i) so that each lane fetches 4 consecutive k with one
// 128-bit access.
//
// gemm_nt:  C[b][n] = sum_k A[b][k] * W[n][k]     W row-major [N][K]  (torch Linear.weight)
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ f32x4 gemm_nt(const float* As, int lda, int acol,
                                         const float* __restrict__ W, int ldw, int n0, int N,
                                         int K, int lane) {
    const int nl = lane & 15, kq = lane >> 4;
    const int n = n0 + nl;
    const float* wrow = W + (size_t)(n < N ? n : N - 1) * ldw + 4 * kq;
    const float* arow = As + nl * lda + acol + 4 * kq;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    const int Kfull = K & ~15;
    int kb = 0;
    for (; kb + 32 <= Kfull; kb += 32) {
        const f32x4 a0 = *(const f32x4*)(arow + kb);
        const f32x4 a1 = *(const f32x4*)(arow + kb + 16);
        const f32x4 b0 = *(const f32x4u*)(wrow + kb);
        const f32x4 b1 = *(const f32x4u*)(wrow + kb + 16);
        acc0 = mfma4(a0.x, b0.x, acc0);
        acc1 = mfma4(a1.x, b1.x, acc1);
        acc0 = mfma4(a0.y, b0.y, acc0);
        acc1 = mfma4(a1.y, b1.y, acc1);
        acc0 = mfma4(a0.z, b0.z, acc0);
        acc1 = mfma4(a1.z, b1.z, acc1);
        acc0 = mfma4(a0.w, b0.w, acc0);
        acc1 = mfma4(a1.w, b1.w, acc1);
    }
    if (kb < Kfull) {
        const f32x4 a0 = *(const f32x4*)(arow + kb);
        const f32x4 b0 = *(const f32x4u*)(wrow + kb);
        acc0 = mfma4(a0.x, b0.x, acc0);
        acc0 = mfma4(a0.y, b0.y, acc0);
        acc0 = mfma4(a0.z, b0.z, acc0);
        acc0 = mfma4(a0.w, b0.w, acc0);
        kb += 16;
    }
    if (kb < K) {   // ragged tail: guard every element (A side is zero padded in LDS)
        const f32x4 a0 = *(const f32x4*)(arow + kb);
        const int kk = kb + 4 * kq;
        f32x4 b0;
        b0.x = (kk + 0 < K) ? wrow[kb + 0] : 0.f;
        b0.y = (kk + 1 < K) ? wrow[kb + 1] : 0.f;
        b0.z = (kk + 2 < K) ? wrow[kb + 2] : 0.f;
        b0.w = (kk + 3 < K) ? wrow[kb + 3] : 0.f;
        acc1 = mfma4(a0.x, b0.x, acc1);
        acc1 = mfma4(a0.y, b0.y, acc1);
        acc1 = mfma4(a0.z, b0.z, acc1);
        acc1 = mfma4(a0.w, b0.w, acc1);
    }
    return acc0 + acc1;
}

// gemm_nn:  C[b][n] = sum_kk A[b][kk] * W[kk][n]    W row-major [K][N] (ld = ldw): the
// transposed product of the backward pass (dX = g * W).
__device__ __forceinline__ f32x4 gemm_nn(const float* As, int lda, int acol,
                                         const float* __restrict__ W, int ldw, int n0, int N,
                                         int K, int lane) {
    const int nl = lane & 15, kq = lane >> 4;
    const int n = n0 + nl;
    const float* wcol = W + (n < N ? n : N - 1) + (size_t)(4 * kq) * ldw;
    const float* arow = As + nl * lda + acol + 4 * kq;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    const int Kfull = K & ~15;
    int kb = 0;
    for (; kb + 32 <= Kfull; kb += 32) {
        const f32x4 a0 = *(const f32x4*)(arow + kb);
        const f32x4 a1 = *(const f32x4*)(arow + kb + 16);
        const float* w0 = wcol + (size_t)kb * ldw;
        const float* w1 = w0 + (size_t)16 * ldw;
        const float b00 = w0[0], b01 = w0[ldw], b02 = w0[2 * ldw], b03 = w0[3 * ldw];
        const float b10 = w1[0], b11 = w1[ldw], b12 = w1[2 * ldw], b13 = w1[3 * ldw];
        acc0 = mfma4(a0.x, b00, acc0);
        acc1 = mfma4(a1.x, b10, acc1);
        acc0 = mfma4(a0.y, b01, acc0);
        acc1 = mfma4(a1.y, b11, acc1);
        acc0 = mfma4(a0.z, b02, acc0);
        acc1 = mfma4(a1.z, b12, acc1);
        acc0 = mfma4(a0.w, b03, acc0);
        acc1 = mfma4(a1.w, b13, acc1);
    }
    if (kb < Kfull) {
        const f32x4 a0 = *(const f32x4*)(arow + kb);
        const float* w0 = wcol + (size_t)kb * ldw;
        const float b00 = w0[0], b01 = w0[ldw], b02 = w0[2 * ldw], b03 = w0[3 * ldw];
        acc0 = mfma4(a0.x, b00, acc0);
        acc0 = mfma4(a0.y, b01, acc0);
        acc0 = mfma4(a0.z, b02, acc0);
        acc0 = mfma4(a0.w, b03, acc0);
        kb += 16;
    }
    if (kb < K) {
        const f32x4 a0 = *(const f32x4*)(arow + kb);
        const float* w0 = wcol + (size_t)kb * ldw;
        const int kk = kb + 4 * kq;
        const float b00 = (kk + 0 < K) ? w0[0] : 0.f;
        const float b01 = (kk + 1 < K) ? w0[ldw] : 0.f;
        const float b02 = (kk + 2 < K) ? w0[2 * ldw] : 0.f;
        const float b03 = (kk + 3 < K) ? w0[3 * ldw] : 0.f;
        acc1 = mfma4(a0.x, b00, acc1);
        acc1 = mfma4(a0.y, b01, acc1);
        acc1 = mfma4(a0.z, b02, acc1);
        acc1 = mfma4(a0.w, b03, acc1);
    }
    return acc0 + acc1;
}

__device__ __forceinline__ float row16_sum(float v) {
    // deterministic butterfly over the 16 lanes that share a batch row
    v += __shfl_xor(v, 8, 16);
    v += __shfl_xor(v, 4, 16);
    v += __shfl_xor(v, 2, 16);
    v += __shfl_xor(v, 1, 16);
    return v;
}

// ---- stage helpers shared by forward / inverse / backward ------------------------------

// v = [u | c] for every node of the group (hint.py:76), zero padded to cinp columns.
__device__ __forceinline__ void stage_build_v(const KArgs& a, const DGroup& g, const float* xs,
                                              const float* cs, float* vb, int tid) {
    for (int ni = g.node_begin; ni < g.node_end; ++ni) {
        const DNode& nd = a.nodes[ni];
        const int cinp = nd.cinp, k = nd.k, cin = nd.cin, off = nd.off, vcol = nd.vcol;
        for (int i = tid; i < ROWS * cinp; i += NTHREADS) {
            const int r = i / cinp, j = i - r * cinp;
            float v = 0.f;
            if (j < k) v = xs[r * a.xld + off + j];
            else if (j < cin) v = cs[r * a.cld + (j - k)];
            vb[r * a.vld + vcol + j] = v;
        }
    }
}

// hidden layer: out = relu(A * W^T + b); LAYER 1 reads v (K = cin), LAYER 2 reads a1 (K = h)
template <int LAYER>
__device__ __forceinline__ void stage_hidden(const KArgs& a, const DGroup& g,
                                             const float* __restrict__ params, const float* Ain,
                                             float* Aout, int wave, int lane) {
    for (int j = wave; j < g.jobsH_cnt; j += NWAVES) {
        const Job job = a.jobs[g.jobsH_begin + j];
        const DNode& nd = a.nodes[job.node];
        const int h = nd.h, n0 = job.tile * TILE;
        const float* W = params + nd.p[job.net * 6 + (LAYER == 1 ? 0 : 2)];
        const float* bvec = params + nd.p[job.net * 6 + (LAYER == 1 ? 1 : 3)];
        f32x4 acc;
        if (LAYER == 1) acc = gemm_nt(Ain, a.vld, nd.vcol, W, nd.cin, n0, h, nd.cin, lane);
        else            acc = gemm_nt(Ain, a.ald, nd.acol + job.net * nd.hp, W, h, n0, h, h, lane);
        const int n = n0 + (lane & 15);
        const bool ok = n < h;
        const float bias = ok ? bvec[n] : 0.f;
        float* o = Aout + (4 * (lane >> 4)) * a.ald + nd.acol + job.net * nd.hp + n;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i * a.ald] = ok ? fmaxf(acc[i] + bias, 0.f) : 0.f;
    }
}

// output layer: st = a2 * W3^T + b3  (no activation)
__device__ __forceinline__ void stage_out(const KArgs& a, const DGroup& g,
                                          const float* __restrict__ params, const float* a2,
                                          float* st, int wave, int lane) {
    for (int j = wave; j < g.jobsR_cnt; j += NWAVES) {
        const Job job = a.jobs[g.jobsR_begin + j];
        const DNode& nd = a.nodes[job.node];
        const int h = nd.h, r = nd.r, n0 = job.tile * TILE;
        const float* W = params + nd.p[job.net * 6 + 4];
        const float* bvec = params + nd.p[job.net * 6 + 5];
        const f32x4 acc = gemm_nt(a2, a.ald, nd.acol + job.net * nd.hp, W, h, n0, r, h, lane);
        const int n = n0 + (lane & 15);
        const bool ok = n < r;
        const float bias = ok ? bvec[n] : 0.f;
        float* o = st + (4 * (lane >> 4)) * a.sld + nd.scol + job.net * nd.rp + n;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i * a.sld] = ok ? acc[i] + bias : 0.f;
    }
}

__device__ __forceinline__ void load_tile(float* dst, int ld, const float* __restrict__ src,
                                          int width, int row0, int B, int tid) {
    // a 16-row tile of a row-major [B,width] tensor is one contiguous run of 16*width floats
    if (src == nullptr) {
        for (int i = tid; i < ROWS * width; i += NTHREADS) { const int r = i / width; dst[r * ld + (i - r * width)] = 0.f; }
        return;
    }
    const float* p = src + (size_t)row0 * width;
    const int nvalid = (B - row0 < ROWS ? B - row0 : ROWS) * width;
    for (int i = tid; i < ROWS * width; i += NTHREADS) {
        const int r = i / width;
        dst[r * ld + (i - r * width)] = (i < nvalid) ? p[i] : 0.f;
    }
}

__device__ __forceinline__ void store_tile(float* __restrict__ dst, const float* src, int ld,
                                           int width, int row0, int B, int tid) {
    float* p = dst + (size_t)row0 * width;
    const int nvalid = (B - row0 < ROWS ? B - row0 : ROWS) * width;
    for (int i = tid; i < nvalid; i += NTHREADS) {
        const int r = i / width;
        p[i] = src[r * ld + (i - r * width)];
    }
}

// =======================================================================================
// forward (REV=false) / inverse (REV=true): x, J -> z   — one launch per block
// =======================================================================================
template <bool REV>
__global__ __launch_bounds__(NTHREADS) void hint_block_apply_kernel(
    KArgs a, const float* __restrict__ params, const float* __restrict__ x,
    const float* __restrict__ c, float* __restrict__ z, float* __restrict__ J,
    float* __restrict__ tape) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* xs = lds;
    float* cs = xs + ROWS * a.xld;
    float* vb = cs + ROWS * a.cld;
    float* a1 = vb + ROWS * a.vld;
    float* a2 = a1 + ROWS * a.ald;
    float* st = a2 + ROWS * a.ald;
    float* jac = st + ROWS * a.sld;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntiles = (a.B + ROWS - 1) / ROWS;

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int row0 = tile * ROWS;
        load_tile(xs, a.xld, x, a.d, row0, a.B, tid);
        if (a.dc > 0) load_tile(cs, a.cld, c, a.dc, row0, a.B, tid);
        if (tid < ROWS) jac[tid] = 0.f;
        __syncthreads();

        for (int gi = 0; gi < a.n_groups; ++gi) {
            const DGroup& g = a.groups[REV ? (a.n_groups - 1 - gi) : gi];
            stage_build_v(a, g, xs, cs, vb, tid);
            __syncthreads();
            stage_hidden<1>(a, g, params, vb, a1, wave, lane);
            __syncthreads();
            stage_hidden<2>(a, g, params, a1, a2, wave, lane);
            __syncthreads();
            stage_out(a, g, params, a2, st, wave, lane);
            __syncthreads();
            {   // element-wise affine coupling + log-det partial sums (hint.py:79-83)
                const int sub = tid & 15, row = tid >> 4;
                float part = 0.f;
                for (int e = sub; e < g.ent_cnt; e += 16) {
                    const Ent en = a.ents[g.ent_begin + e];
                    const float s = st[row * a.sld + en.scol];
                    const float t = st[row * a.sld + en.tcol];
                    const float aa = a.alpha * atanf(s);
                    float* px = xs + row * a.xld + en.xcol;
                    if (!REV) { *px = expf(aa) * (*px) + t; part += aa; }
                    else      { *px = ((*px) - t) / expf(aa); part -= aa; }
                }
                part = row16_sum(part);
                if (sub == 0) jac[row] += part;
            }
            __syncthreads();
            // training: keep the lane tile as it stands after each level except the root's, so
            // that the backward pass sees bit-identical subnet inputs (tape[level][B][d])
            if (!REV && tape != nullptr && g.level_last && g.level < a.n_levels - 1)
                store_tile(tape + (size_t)g.level * a.B * a.d, xs, a.xld, a.d, row0, a.B, tid);
        }
        store_tile(z, xs, a.xld, a.d, row0, a.B, tid);
        if (tid < ROWS && row0 + tid < a.B) J[row0 + tid] = jac[tid];
        __syncthreads();
    }
}

// =======================================================================================
// backward, part A (row parallel): walk the levels root first; per level reload the lane tile
// the forward pass recorded (x for the deepest level, tape[level-1] otherwise), recompute each
// node's activations from it (same code, same inputs -> bit-identical ReLU masks),
// back-propagate through coupling and subnets to get g_x / g_c, and leave the per-layer
// activations and pre-activation gradients in the workspace for the weight-gradient GEMMs.
//   g_t = g_l' ; g_a = g_l'*exp(a)*l + g_J ; g_l = g_l'*exp(a) ; g_s = g_a*alpha/(1+s^2)
// =======================================================================================
__device__ __forceinline__ void copy_rows_out(float* __restrict__ dst, int dld, int dcol,
                                              const float* src, int sld, int width, int row0,
                                              int tid) {
    // width is a multiple of 16, dcol/dld multiples of 4 -> 128-bit rows
    const int w4 = width >> 2;
    for (int i = tid; i < ROWS * w4; i += NTHREADS) {
        const int r = i / w4, j = (i - r * w4) << 2;
        *(f32x4*)(dst + (size_t)(row0 + r) * dld + dcol + j) = *(const f32x4*)(src + r * sld + j);
    }
}

__global__ __launch_bounds__(NTHREADS) void hint_block_bwd_kernel(
    KArgs a, const float* __restrict__ params, const float* __restrict__ x,
    const float* __restrict__ tape, const float* __restrict__ c, const float* __restrict__ g_z,
    const float* __restrict__ g_J,
    float* __restrict__ g_x, float* __restrict__ g_c, float* __restrict__ wsV,
    float* __restrict__ wsA1, float* __restrict__ wsA2, float* __restrict__ wsG1,
    float* __restrict__ wsG2, float* __restrict__ wsG3) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* xs = lds;
    float* gs = xs + ROWS * a.xld;
    float* cs = gs + ROWS * a.xld;
    float* gcs = cs + ROWS * a.cld;
    float* vb = gcs + ROWS * a.cld;
    float* gv = vb + ROWS * a.vld;
    float* a1 = gv + ROWS * a.vld;
    float* a2 = a1 + ROWS * a.ald;
    float* st = a2 + ROWS * a.ald;
    float* gst = st + ROWS * a.sld;
    float* gj = gst + ROWS * a.sld;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntiles = (a.B + ROWS - 1) / ROWS;

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int row0 = tile * ROWS;
        load_tile(gs, a.xld, g_z, a.d, row0, a.B, tid);
        if (a.dc > 0) {
            load_tile(cs, a.cld, c, a.dc, row0, a.B, tid);
            load_tile(gcs, a.cld, nullptr, a.dc, row0, a.B, tid);
        }
        if (tid < ROWS) gj[tid] = (g_J != nullptr && row0 + tid < a.B) ? g_J[row0 + tid] : 0.f;
        __syncthreads();

        for (int gi = a.n_groups - 1; gi >= 0; --gi) {
            const DGroup& g = a.groups[gi];
            // ---- the lanes as the forward pass saw them when it entered this level ----
            if (g.level_last) {
                const float* src = (g.level == 0) ? x : tape + (size_t)(g.level - 1) * a.B * a.d;
                load_tile(xs, a.xld, src, a.d, row0, a.B, tid);
                __syncthreads();
            }
            // ---- recompute s, t of every node of the group (bit-identical to the forward) ----
            stage_build_v(a, g, xs, cs, vb, tid);
            for (int i = tid; i < ROWS * g.sw; i += NTHREADS) {
                const int r = i / g.sw;
                gst[r * a.sld + (i - r * g.sw)] = 0.f;
            }
            __syncthreads();
            if (g.vw > 0) copy_rows_out(wsV, a.VT, g.wvcol0, vb, a.vld, g.vw, row0, tid);
            stage_hidden<1>(a, g, params, vb, a1, wave, lane);
            __syncthreads();
            copy_rows_out(wsA1, a.WT, g.wcol0, a1, a.ald, g.aw, row0, tid);
            stage_hidden<2>(a, g, params, a1, a2, wave, lane);
            __syncthreads();
            copy_rows_out(wsA2, a.WT, g.wcol0, a2, a.ald, g.aw, row0, tid);
            stage_out(a, g, params, a2, st, wave, lane);
            __syncthreads();
            {   // ---- coupling backward ----
                const int sub = tid & 15, row = tid >> 4;
                const float gJr = gj[row];
                for (int e = sub; e < g.ent_cnt; e += 16) {
                    const Ent en = a.ents[g.ent_begin + e];
                    const float s = st[row * a.sld + en.scol];
                    const float aa = a.alpha * atanf(s);
                    const float ea = expf(aa);
                    const float l = xs[row * a.xld + en.xcol];       // lower input of the node
                    float* pg = gs + row * a.xld + en.xcol;
                    const float glp = *pg;                            // grad wrt l' = exp(a)*l + t
                    *pg = glp * ea;                                   // g_l
                    const float ga = glp * ea * l + gJr;              // g_a (a feeds both l' and J)
                    gst[row * a.sld + en.scol] = ga * a.alpha / (1.f + s * s);   // g_s
                    gst[row * a.sld + en.tcol] = glp;                              // g_t
                }
            }
            __syncthreads();
            // ---- g2 = (g3 * W3) .* relu'(a2), in place over a2 ----
            copy_rows_out(wsG3, a.ST, g.wscol0, gst, a.sld, g.sw, row0, tid);
            for (int j = wave; j < g.jobsH_cnt; j += NWAVES) {
                const Job job = a.jobs[g.jobsH_begin + j];
                const DNode& nd = a.nodes[job.node];
                const int h = nd.h, n0 = job.tile * TILE;
                const float* W3 = params + nd.p[job.net * 6 + 4];
                const f32x4 acc = gemm_nn(gst, a.sld, nd.scol + job.net * nd.rp, W3, h, n0, h, nd.r, lane);
                const int n = n0 + (lane & 15);
                float* o = a2 + (4 * (lane >> 4)) * a.ald + nd.acol + job.net * nd.hp + n;
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i * a.ald] = (n < h && o[i * a.ald] > 0.f) ? acc[i] : 0.f;
            }
            __syncthreads();
            // ---- g1 = (g2 * W2) .* relu'(a1), in place over a1 ----
            copy_rows_out(wsG2, a.WT, g.wcol0, a2, a.ald, g.aw, row0, tid);
            for (int j = wave; j < g.jobsH_cnt; j += NWAVES) {
                const Job job = a.jobs[g.jobsH_begin + j];
                const DNode& nd = a.nodes[job.node];
                const int h = nd.h, n0 = job.tile * TILE;
                const float* W2 = params + nd.p[job.net * 6 + 2];
                const f32x4 acc = gemm_nn(a2, a.ald, nd.acol + job.net * nd.hp, W2, h, n0, h, h, lane);
                const int n = n0 + (lane & 15);
                float* o = a1 + (4 * (lane >> 4)) * a.ald + nd.acol + job.net * nd.hp + n;
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i * a.ald] = (n < h && o[i * a.ald] > 0.f) ? acc[i] : 0.f;
            }
            __syncthreads();
            // ---- g_v = g1_s * W1_s + g1_t * W1_t  (both nets feed the same v) ----
            copy_rows_out(wsG1, a.WT, g.wcol0, a1, a.ald, g.aw, row0, tid);
            for (int j = wave; j < g.jobsC_cnt; j += NWAVES) {
                const Job job = a.jobs[g.jobsC_begin + j];   // net field unused: one job sums s and t
                const DNode& nd = a.nodes[job.node];
                const int h = nd.h, cin = nd.cin, n0 = job.tile * TILE;
                const f32x4 accs = gemm_nn(a1, a.ald, nd.acol, params + nd.p[0], cin, n0, cin, h, lane);
                const f32x4 acct = gemm_nn(a1, a.ald, nd.acol + nd.hp, params + nd.p[6], cin, n0, cin, h, lane);
                const int n = n0 + (lane & 15);
                float* o = gv + (4 * (lane >> 4)) * a.vld + nd.vcol + n;
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i * a.vld] = (n < cin) ? accs[i] + acct[i] : 0.f;
            }
            __syncthreads();
            // ---- scatter g_v: first k columns to the upper lanes, the rest to g_c ----
            for (int ni = g.node_begin; ni < g.node_end; ++ni) {
                const DNode& nd = a.nodes[ni];
                const int k = nd.k;
                for (int i = tid; i < ROWS * k; i += NTHREADS) {
                    const int r = i / k, j = i - r * k;
                    gs[r * a.xld + nd.off + j] += gv[r * a.vld + nd.vcol + j];
                }
            }
            if (a.dc > 0) {
                for (int i = tid; i < ROWS * a.dc; i += NTHREADS) {
                    const int r = i / a.dc, j = i - r * a.dc;
                    float acc = gcs[r * a.cld + j];
                    for (int ni = g.node_begin; ni < g.node_end; ++ni) {
                        const DNode& nd = a.nodes[ni];
                        acc += gv[r * a.vld + nd.vcol + nd.k + j];
                    }
                    gcs[r * a.cld + j] = acc;
                }
            }
            __syncthreads();
        }
        store_tile(g_x, gs, a.xld, a.d, row0, a.B, tid);
        if (a.dc > 0 && g_c != nullptr) store_tile(g_c, gcs, a.cld, a.dc, row0, a.B, tid);
        __syncthreads();
    }
}

// =======================================================================================
// backward, part B: weight gradients.  dW[m][n] = sum_b G[b][gcol+m] * X[b][xcol+n] is a
// GEMM whose reduction runs over the batch, so here the OUTPUT is tiled (48x48 per
// workgroup) and the batch is split over blockIdx.y and over the 4 wavefronts of a
// workgroup; wavefront partials are combined in LDS, workgroup partials with float atomics.
// The bias gradient (column sums of G) rides along as one extra MFMA against a ones vector.
// =======================================================================================
__global__ __launch_bounds__(NTHREADS) void hint_block_dw_kernel(
    const DWJob* __restrict__ jobs, const float* __restrict__ wsV, const float* __restrict__ wsA1,
    const float* __restrict__ wsA2, const float* __restrict__ wsG1, const float* __restrict__ wsG2,
    const float* __restrict__ wsG3, int WT, int VT, int ST, int Bp, int rows_per_wg,
    int use_atomics, float* __restrict__ gparams) {
    __shared__ float red[NWAVES][12][64][4];   // 48 KB

    const DWJob job = jobs[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nl = lane & 15, kq = lane >> 4;

    const float* G; int gld;
    if (job.gsel == 0) { G = wsG1; gld = WT; } else if (job.gsel == 1) { G = wsG2; gld = WT; } else { G = wsG3; gld = ST; }
    const float* X; int xld;
    if (job.xsel == 0) { X = wsV; xld = VT; } else if (job.xsel == 1) { X = wsA1; xld = WT; } else { X = wsA2; xld = WT; }

    const int ntm = min(3, (job.M - job.m0 + 15) >> 4);
    const int ntn = min(3, (job.N - job.n0 + 15) >> 4);
    const bool bias = (job.n0 == 0);

    const int b_begin = blockIdx.y * rows_per_wg;
    const int b_end = min(Bp, b_begin + rows_per_wg);

    f32x4 acc[3][3], accb[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        accb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    const float* gp = G + job.gcol + job.m0 + nl;
    const float* xp = X + job.xcol + job.n0 + nl;
    for (int bb = b_begin + wave * 16; bb < b_end; bb += 16 * NWAVES) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const size_t row = (size_t)(bb + 4 * i + kq);
            float av[3], bv[3];
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                av[t] = (t < ntm) ? gp[row * gld + 16 * t] : 0.f;
                bv[t] = (t < ntn) ? xp[row * xld + 16 * t] : 0.f;
            }
#pragma unroll
            for (int tm = 0; tm < 3; ++tm) {
                if (tm < ntm) {
#pragma unroll
                    for (int tn = 0; tn < 3; ++tn)
                        if (tn < ntn) acc[tm][tn] = mfma4(av[tm], bv[tn], acc[tm][tn]);
                    if (bias) accb[tm] = mfma4(av[tm], 1.0f, accb[tm]);
                }
            }
        }
    }
    // combine the four wavefronts
#pragma unroll
    for (int tm = 0; tm < 3; ++tm) {
#pragma unroll
        for (int tn = 0; tn < 3; ++tn) *(f32x4*)&red[wave][tm * 3 + tn][lane][0] = acc[tm][tn];
        *(f32x4*)&red[wave][9 + tm][lane][0] = accb[tm];
    }
    __syncthreads();
    for (int idx = tid; idx < 12 * 64; idx += NTHREADS) {
        const int t = idx >> 6, l = idx & 63;
        const int tm = (t < 9) ? t / 3 : t - 9;
        const int tn = (t < 9) ? t - 3 * tm : 0;
        if (tm >= ntm || (t < 9 && tn >= ntn) || (t >= 9 && !bias)) continue;
        const f32x4 v = *(f32x4*)&red[0][t][l][0] + *(f32x4*)&red[1][t][l][0] +
                        *(f32x4*)&red[2][t][l][0] + *(f32x4*)&red[3][t][l][0];
        const int n = job.n0 + 16 * tn + (l & 15);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = job.m0 + 16 * tm + 4 * (l >> 4) + i;
            if (m >= job.M) continue;
            if (t < 9) {
                if (n < job.N) {
                    float* dst = gparams + job.wofs + (size_t)m * job.N + n;
                    if (use_atomics) atomicAdd(dst, v[i]); else *dst = v[i];
                }
            } else if ((l & 15) == 0) {
                float* dst = gparams + job.bofs + m;
                if (use_atomics) atomicAdd(dst, v[i]); else *dst = v[i];
            }
        }
    }
}

// ---- launchers (called from hint_plan.cpp) ----------------------------------------------
namespace hint {

hipError_t launch_apply(bool rev, const KArgs& a, int lds_bytes, int grid, const float* params,
                        const float* x, const float* c, float* z, float* J, float* tape,
                        hipStream_t stream) {
    if (rev)
        hipLaunchKernelGGL(hint_block_apply_kernel<true>, dim3(grid), dim3(NTHREADS), lds_bytes, stream, a, params, x, c, z, J, (float*)nullptr);
    else
        hipLaunchKernelGGL(hint_block_apply_kernel<false>, dim3(grid), dim3(NTHREADS), lds_bytes, stream, a, params, x, c, z, J, tape);
    return hipGetLastError();
}

hipError_t launch_bwd(const KArgs& a, int lds_bytes, int grid, const float* params, const float* x,
                      const float* tape, const float* c, const float* g_z, const float* g_J, float* g_x, float* g_c,
                      float* wsV, float* wsA1, float* wsA2, float* wsG1, float* wsG2, float* wsG3,
                      hipStream_t stream) {
    hipLaunchKernelGGL(hint_block_bwd_kernel, dim3(grid), dim3(NTHREADS), lds_bytes, stream, a, params,
                       x, tape, c, g_z, g_J, g_x, g_c, wsV, wsA1, wsA2, wsG1, wsG2, wsG3);
    return hipGetLastError();
}

hipError_t launch_dw(const DWJob* jobs, int n_jobs, int splits, const float* wsV, const float* wsA1,
                     const float* wsA2, const float* wsG1, const float* wsG2, const float* wsG3,
                     int WT, int VT, int ST, int Bp, int rows_per_wg, float* gparams,
                     hipStream_t stream) {
    hipLaunchKernelGGL(hint_block_dw_kernel, dim3(n_jobs, splits), dim3(NTHREADS), 0, stream, jobs, wsV,
                       wsA1, wsA2, wsG1, wsG2, wsG3, WT, VT, ST, Bp, rows_per_wg, splits > 1 ? 1 : 0,
                       gparams);
    return hipGetLastError();
}

hipError_t set_max_lds(int fwd_bytes, int bwd_bytes) {
    hipError_t e;
    e = hipFuncSetAttribute((const void*)hint_block_apply_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, fwd_bytes);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void*)hint_block_apply_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, fwd_bytes);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute((const void*)hint_block_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bwd_bytes);
}

}  // namespace hint
