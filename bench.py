#!/usr/bin/env python3
"""Benchmark of the HINT coupling-flow training step on MI355X (BASELINE.json metric:
training samples/s + mean NLL, UCI POWER d=6, batch 4096 per GPU, 1/2/4/8 GPUs).

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W [--scaling strong]

A "step" = noise + forward + backward of the 8-block flow on this rank's shard, one all-reduce of the
flat gradient arena, fused clamp + Adam (train_unconditional.py:114-144).  --scaling weak (default):
every rank has the workload's batch (4096 rows); --scaling strong: the workload's batch is the GLOBAL
batch, split over the ranks.  Inputs are synthetic N(0,1) rows already resident in HBM.  Rank 0 prints
ONE JSON line; it also carries the sampling direction (`inverse_samples_per_sec`, the whole chain
x = f^-1(z) in one launch), the roofline of the step's dominant kernel and the CPU baseline.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

WORKLOADS = {
    # BASELINE.json configs[1]: UCI POWER d=6, 8 recursive coupling blocks, batch 4096, 1 GPU
    "power_hint_8": dict(d=6, n_blocks=8, c_internal=[140, 70, 35, 17], batch=4096),
    "power_hint_4": dict(d=6, n_blocks=4, c_internal=[200, 100, 50, 25], batch=512),
    "gas_hint_8": dict(d=8, n_blocks=8, c_internal=[128, 64, 32, 16], batch=8192),
    "miniboone_hint_10": dict(d=43, n_blocks=10, c_internal=[67, 33, 16, 8], batch=4096),
    "plus_hint_4": dict(d=100, n_blocks=4, c_internal=[224, 112, 56], batch=4096),
    # configs/plus_shape/unconditional_hint_4_3_big.py (h = 512: the s and t nets of the wide nodes run one at a time)
    "plus_hint_4_big": dict(d=100, n_blocks=4, c_internal=[512, 256, 128, 64], batch=4096),
}
# BASELINE.json configs[3]: configs/plus_shape/conditional_hint_4_full.py:58-94 - the two-lane conditional model (x lane: recursive
# block d = 100 + ExternalAffineCoupling given y; y lane: AffineCoupling d = 4), 4 blocks, internal width 224, 4096 rows per GPU
CONDITIONAL = {"conditional_hint_4_full": dict(nx=100, ny=4, n_blocks=4, hidden=224, batch=4096)}
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 = fp32 vector rate
PEAK_HBM_GBS = 8000.0


def flops_per_sample_block(d, widths, dc=0):
    """algorithmic forward FLOPs of one block per sample: sum over nodes of both subnets,
    2*(cin*h + h*h + h*r) MACs each."""
    from oracle import hint_oracle as orc   # only for the node list (bench-side bookkeeping)
    nodes = orc.build_nodes(d, [(dc,)] if dc else [], widths)
    return sum(2 * 2 * (n.cin * n.h + n.h * n.h + n.h * n.r) for n in nodes)


def tree_nodes(tree, dc):
    """oracle node list of a hint_amd tree module (any split rule): what the CPU restatement runs on"""
    from oracle import hint_oracle as orc
    out = []

    def rec(node, path, off, depth):
        D, k = node.data_shape[0], node.split_idx
        out.append(orc.ONode(path, off, D, k, D - k, node.s[0].out_features, k + dc, depth, node.leaf))
        if not node.leaf:
            rec(node.upper, path + ".upper", off, depth + 1)
            rec(node.lower, path + ".lower", off + k, depth + 1)

    rec(tree, "tree", 0, 0)
    return out


def conditional_main(args, name, rank, world, dev, use_dist, dist):
    """the conditional two-lane model (BASELINE.json configs[3]) on hint_amd.ConditionalFlowTrainer: one step = noise, both
    lanes forward and backward (train_conditional.py:120-150), one all-reduce of the flat gradient arena, fused clamp + Adam"""
    import hint_amd
    from oracle import hint_oracle as orc
    cfg = CONDITIONAL[name]
    B = args.batch if args.batch > 0 else cfg["batch"]
    if args.scaling == "strong":
        B //= world
    torch.manual_seed(0)
    model = hint_amd.ConditionalHintFlow(cfg["nx"], cfg["ny"], cfg["n_blocks"], cfg["hidden"]).to(dev)
    with torch.no_grad():
        gw = torch.Generator().manual_seed(0)
        for p in model.parameters():
            p.data = (0.005 * torch.randn(p.shape, generator=gw)).to(dev)
    tr = hint_amd.ConditionalFlowTrainer(model, use_graph=not args.no_graph)
    gx = torch.Generator().manual_seed(1000 + rank)
    x = torch.randn(B, cfg["nx"], generator=gx).to(dev)
    y = torch.randn(B, cfg["ny"], generator=gx).to(dev)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        tr.step(x, y)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        l0, l1 = tr.step(x, y)
    barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank != 0:
        return
    # algorithmic FLOPs of a step: forward MACs of every subnet (SURVEY §8d's F, summed over the three modules of a
    # block) x 3 (forward, dX, dW)
    F = 0
    mods = []
    for i in range(cfg["n_blocks"]):
        mods += [(model.hac_x[i], 0), (model.ac_y_to_x[i], cfg["ny"]), (model.ac_y[i], 0)]
    for m, dc in mods:
        F += sum(2 * 2 * (n.cin * n.h + n.h * n.h + n.h * n.r) for n in tree_nodes(m.tree, dc))
    ms = elapsed / args.steps * 1e3
    ach = 3.0 * F * B / (ms * 1e-3) / 1e12
    res = {
        "metric": "train_samples_per_sec", "value": B * world * args.steps / elapsed, "unit": "samples/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": args.scaling,
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{name}: two-lane conditional HINT, x d={cfg['nx']}, y d={cfg['ny']}, {cfg['n_blocks']} blocks, "
                               f"internal width {cfg['hidden']}, batch {B} per GPU (configs/plus_shape/conditional_hint_4_full.py:58-94)",
                   "global_batch": B * world, "parallelism": f"dp{world}", "hip_graph": not args.no_graph},
        "last_step_loss": float(l0) + float(l1),
        "roofline": {"bound": "mfma", "kernel": "whole step (the x lane's hint_apply / hint_bwd / hint_wgrad launches dominate)",
                     "achieved": ach, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_F32_MFMA_TFLOPS, "traffic": None,
                     "algorithmic_flops_per_step": 3.0 * F * B,
                     "timing": "wall clock of the timed steps (graph replays), all launches of a step"},
    }
    if world == 1 and not args.no_cpu_baseline:
        # the same step on the host cores: oracle blocks (plain torch CPU ops + autograd) composed as the two-lane graph, a
        # bounded sample of 256 rows
        Bc = min(B, 256)
        P = [{k: v.detach().cpu().clone().requires_grad_(True) for k, v in m.state_dict().items()} for m, _ in mods]
        nodes = [tree_nodes(m.tree, dc) for m, dc in mods]
        Wy = [model.perm_y[i].W.cpu() if i > 0 else None for i in range(cfg["n_blocks"])]
        Wx = [model.perm_x[i].W.cpu() if i > 0 else None for i in range(cfg["n_blocks"])]
        opt = torch.optim.Adam([p for d_ in P for p in d_.values()], lr=0.01 * 3e-2, betas=(0.9, 0.95), eps=1e-4, weight_decay=1.86e-5)
        xc, yc = x[:Bc].cpu(), y[:Bc].cpu()

        def cpu_step():
            opt.zero_grad()
            xo, yo = xc + 0.01 * torch.randn_like(xc), yc
            J = 0
            for i in range(cfg["n_blocks"]):
                if i > 0:
                    yo = yo @ Wy[i]; xo = xo @ Wx[i]
                xo, j = orc.block_apply(nodes[3 * i], P[3 * i], xo, [], clamp=4.0); J = J + j
                xo, j = orc.block_apply(nodes[3 * i + 1], P[3 * i + 1], xo, [yo], clamp=5.0); J = J + j
                yo, j = orc.block_apply(nodes[3 * i + 2], P[3 * i + 2], yo, [], clamp=5.0); J = J + j
            loss = 0.5 * (torch.cat([xo, yo], -1) ** 2).sum(1).mean() - J.mean()
            loss.backward()
            for d_ in P:
                for p in d_.values():
                    p.grad.clamp_(-5.0, 5.0)
            opt.step()

        ncores = os.cpu_count() or 1
        best_nt, best_t = None, float("inf")
        for nt in sorted({min(ncores, v) for v in (8, 16, 32)}):
            torch.set_num_threads(nt)
            cpu_step()
            t1 = time.perf_counter(); cpu_step(); t = time.perf_counter() - t1
            if t < best_t:
                best_nt, best_t = nt, t
        torch.set_num_threads(best_nt)
        n = int(max(2, min(50, 12.0 / best_t)))
        t1 = time.perf_counter()
        for _ in range(n):
            cpu_step()
        dt = time.perf_counter() - t1
        res["cpu_baseline"] = dict(value=Bc * n / dt, unit="samples/s", cores=best_nt, kind="port",
                                   sample=f"{n} training steps of {Bc} rows ({dt:.1f} s): oracle blocks composed as the two-lane graph "
                                          f"(torch CPU ops + autograd + clamp + Adam), best of 8/16/32 threads = {best_nt}; host has {ncores} cores")
        res["speedup_vs_cpu"] = res["value"] / res["cpu_baseline"]["value"]
    print(json.dumps(res), flush=True)


def cpu_baseline(cfg, budget_s=12.0, max_steps=200):
    """the reference's CPU path (oracle restatement of hint.py in plain torch CPU ops +
    autograd + clamp + Adam), timed on this box's host cores on a bounded number of steps.
    torch's default of one thread per core is pathological for these small matrices on a
    many-core host, so a few thread counts are probed first (2 steps each) and the fastest
    is used; `cores` reports the threads actually used."""
    from oracle import hint_oracle as orc
    ncores = os.cpu_count() or 1
    flow = orc.OracleFlow(cfg["d"], cfg["n_blocks"], cfg["c_internal"], seed=0, init_scale=0.005)
    flow.make_optimizer()
    g = torch.Generator().manual_seed(0)
    x = torch.randn(cfg["batch"], cfg["d"], generator=g)

    def run(n):
        t0 = time.perf_counter()
        for _ in range(n):
            flow.train_step(x + 0.01 * torch.randn_like(x))
        return time.perf_counter() - t0

    best_nt, best_t = None, float("inf")
    for nt in sorted({min(ncores, v) for v in (8, 16, 32, 64)}):
        torch.set_num_threads(nt)
        run(1)
        t = run(2) / 2
        if t < best_t:
            best_nt, best_t = nt, t
    torch.set_num_threads(best_nt)
    n = int(max(3, min(max_steps, budget_s / best_t)))
    dt = run(n)
    return dict(value=cfg["batch"] * n / dt, unit="samples/s", cores=best_nt, kind="port",
                sample=f"{n} training steps of {cfg['batch']} rows ({dt:.1f} s), oracle/hint_oracle.py OracleFlow "
                       f"(torch CPU ops, best of 8/16/32/64 threads = {best_nt}; host has {ncores} cores)")


def kernel_legs(trainer, x, reps=200):
    """average device time of each hot kernel, HIP events on the launch stream, back to back"""
    from hint_amd import _lib
    lib = _lib.load()
    eng = trainer.engines[0]
    B = x.shape[0]
    out = {}

    def timed(fn):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / reps   # us

    z = torch.empty_like(x); J = torch.empty(B, device=x.device)
    tape = torch.empty(max(lib.hint_plan_tape_floats(eng.plan, B), 1), device=x.device)
    stream = torch.cuda.current_stream().cuda_stream
    P, PK = eng.arena.data_ptr(), eng.packed.data_ptr()
    out["hint_pack_kernel"] = timed(lambda: lib.hint_block_pack(eng.plan, P, PK, stream))
    out["hint_apply_kernel<fwd>"] = timed(
        lambda: lib.hint_block_forward(eng.plan, P, PK, x.data_ptr(), None, z.data_ptr(), J.data_ptr(), tape.data_ptr(), B, stream))
    gz = torch.randn_like(x); gJ = torch.full((B,), -1.0 / B, device=x.device)
    gx = torch.empty_like(x); gp = torch.empty(eng.total, device=x.device)
    nb = lib.hint_plan_workspace_bytes(eng.plan, B)
    ws = torch.empty(nb, dtype=torch.uint8, device=x.device)

    def bwd():
        return lib.hint_block_backward(eng.plan, P, PK, x.data_ptr(), tape.data_ptr(), None, gz.data_ptr(), gJ.data_ptr(),
                                       gx.data_ptr(), None, gp.data_ptr(), 1, ws.data_ptr(), nb, B, stream)
    out["backward_total"] = timed(bwd)
    # the launches the timed step actually makes: the whole flow per kernel (hint_chain_*)
    if trainer._chainable:
        chain = trainer._chain_for(B)
        n = len(trainer.engines)
        zc = torch.empty_like(x); Jc = torch.empty(B, device=x.device); gxc = torch.empty_like(x)
        acc = torch.zeros(64, 2, device=x.device)
        out[f"chain{n}:hint_apply_kernel<fwd>"] = timed(
            lambda: lib.hint_chain_forward(chain, x.data_ptr(), None, zc.data_ptr(), Jc.data_ptr(), None,
                                           acc.data_ptr(), stream))

        for parts, name in ((1, "hint_bwd_kernel"), (2, "hint_wgrad_kernel+hint_wreduce_kernel"), (3, "backward_total")):
            out[f"chain{n}:{name}"] = timed(
                lambda: lib.hint_chain_backward_parts(chain, x.data_ptr(), None, zc.data_ptr(), None, gxc.data_ptr(), None,
                                                      1.0 / B, -1.0 / B, 1, parts, stream))
        trainer.G.zero_()                    # the timing launches accumulated into the gradient arena
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="power_hint_8", choices=sorted(WORKLOADS) + sorted(CONDITIONAL))
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--steps-per-graph", type=int, default=1,
                    help="training iterations per hipGraph replay (FlowTrainer.step_many; one process only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--batch", type=int, default=0,
                    help="rows per GPU instead of the workload's (small-batch measurements; the config line says so)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: the workload's batch per rank; strong: the workload's batch is the global batch")
    ap.add_argument("--legs", action="store_true",
                    help="also time every kernel in a hot back-to-back loop (single-block and chained launches)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    use_dist = world > 1 or os.environ.get("HINT_FORCE_DIST") == "1"   # the latter: exercise RCCL on one GPU
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import hint_amd
    if args.workload in CONDITIONAL:
        conditional_main(args, args.workload, rank, world, dev, use_dist, dist)
        if use_dist:
            dist.destroy_process_group()
        return
    cfg = WORKLOADS[args.workload]
    d, B = cfg["d"], cfg["batch"]
    if args.batch > 0:
        B = args.batch
    if args.scaling == "strong":
        if B % world:
            raise SystemExit(f"--scaling strong: the global batch {B} does not divide over {world} ranks")
        B //= world
    torch.manual_seed(0)                                     # identical weights on every rank
    flow = hint_amd.HintFlow(d, cfg["n_blocks"], cfg["c_internal"]).to(dev)
    with torch.no_grad():
        gw = torch.Generator().manual_seed(0)
        for p in flow.parameters():                          # train_unconditional.py:165-167
            p.data = (0.005 * torch.randn(p.shape, generator=gw)).to(dev)
    trainer = hint_amd.FlowTrainer(flow, use_graph=not args.no_graph)
    gx = torch.Generator().manual_seed(1000 + rank)          # each rank its own shard of the global batch
    x = torch.randn(B, d, generator=gx).to(dev)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # the batch is resident in HBM before the timed region starts (the bench contract): it is written
    # once into the captured step's own input buffer, so that no device-to-device copy of it runs per step
    spg = args.steps_per_graph if (args.steps_per_graph > 1 and not use_dist and not args.no_graph) else 1
    x, _ = trainer.input_buffers(x)
    if spg > 1:                              # K resident batches (K different shards of synthetic data) per replay
        xs = torch.randn(spg, B, d, generator=gx).to(dev)
        xs, _ = trainer.input_buffers_many(xs)
    for _ in range(args.warmup):
        trainer.step(x)
    if spg > 1:
        trainer.step_many(xs)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps // spg):
        if spg > 1:
            trainer.step_many(xs)            # spg full iterations (re-pack, forward, backward, clamp+Adam each)
        else:
            trainer.step(x)                  # loss terms are accumulated on the device, read once below
    for _ in range(args.steps % spg):
        trainer.step(x)
    barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    ms_per_step = elapsed / args.steps * 1e3
    value = B * world * args.steps / elapsed
    l0, l1 = trainer.last_losses()
    loss_last = float(l0) + float(l1)
    nll = trainer.nll(x)

    # the step's launches inside real (un-captured) steps, HIP events between them: what the roofline
    # is priced on.  Every rank runs them (the steps contain the gradient all-reduce).
    in_step = {}
    if trainer._chainable:
        for k in range(3 + 20):
            t = trainer.timed_step(x)
            if k >= 3:
                for n_, v_ in t.items():
                    in_step[n_] = in_step.get(n_, 0.0) + v_ / 20

    # the sampling direction: x = f^-1(z) through the whole chain, one launch (hint_chain_inverse), on the same
    # resident rows; HIP events on the launch stream
    inv_us = None
    if trainer._chainable:
        from hint_amd import _lib
        lib = _lib.load()
        chain = trainer._chain_infer(B)
        zi = torch.randn(B, d, device=dev)
        xi, Ji = torch.empty_like(zi), torch.empty(B, device=dev)
        stream = torch.cuda.current_stream().cuda_stream
        run_inv = lambda: _lib.check(lib.hint_chain_inverse(chain, zi.data_ptr(), None, xi.data_ptr(), Ji.data_ptr(), None,
                                                            stream), "hint_chain_inverse")
        for _ in range(5):
            run_inv()
        reps = 50
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        barrier()
        e0.record()
        for _ in range(reps):
            run_inv()
        e1.record()
        torch.cuda.synchronize()
        inv_us = e0.elapsed_time(e1) * 1e3 / reps
        if use_dist:
            t = torch.tensor([inv_us], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            inv_us = float(t.item())

    if rank == 0:
        F = flops_per_sample_block(d, cfg["c_internal"])
        nb = cfg["n_blocks"]                           # blocks one launch processes
        legs = kernel_legs(trainer, x) if args.legs else {}
        res = {
            "metric": "train_samples_per_sec", "value": value, "unit": "samples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: d={d}, {nb} recursive coupling blocks, "
                                   f"c_internal={cfg['c_internal']}, batch {B} per GPU",
                       "global_batch": B * world, "parallelism": f"dp{world}", "hip_graph": not args.no_graph, "steps_per_graph": spg,
                       "input": "batch resident in the captured step's input buffer (no per-step copy)"},
            "mean_nll_nats": nll, "last_step_loss": loss_last,
        }
        # HBM traffic per launch comes from rocprofv3 PMC passes run separately (tools/refresh_profiles.sh); it is NOT measured in
        # this process - the line says where it was read from
        pmc = {}
        pmc_file = "profiles/r03_pmc_summary.json"
        pmc_path = os.path.join(ROOT, pmc_file)
        if os.path.exists(pmc_path) and args.workload == "power_hint_8" and B == 4096:
            try:
                pmc = json.load(open(pmc_path))
            except Exception:
                pmc = {}

        def pmc_entry(kernel):             # rocprof's kernel names carry the template arguments: match on the prefix
            return next((v for k, v in pmc.items() if k.replace(" ", "").startswith(kernel.replace(" ", ""))), None)

        def mfma_roofline(kernel, us, flops):
            ach = flops / (us * 1e-6) / 1e12
            e = pmc_entry(kernel) or {}
            issued = e.get("sq", {}).get("SQ_INSTS_MFMA")
            return {"bound": "mfma", "kernel": kernel, "achieved": ach, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": ach / PEAK_F32_MFMA_TFLOPS, "traffic": e.get("hbm_bytes_per_launch"),
                    "traffic_source": f"{pmc_file} (rocprofv3 --pmc passes of tools/steps.py, builder-run; not measured in this run)" if e else None,
                    "avg_launch_us": us, "timing": "HIP events around the launch inside 20 un-captured training steps (FlowTrainer.timed_step); "
                                                   "`value` is timed on graph replays",
                    "algorithmic_flops_per_launch": flops, "blocks_per_launch": nb,
                    "mfma_issued_over_algorithmic": (issued * 2048.0 / flops) if issued else None}

        if in_step:
            fwd_name, bwd_name = trainer.kernel_names(B)
            # dominant kernel = the row-parallel backward kernel, one launch for all blocks: dX through the three
            # layers of every subnet = the forward's MAC count F per sample and block (the hidden activations' signs come
            # from the forward's tape; the weight gradients are the wgrad kernel's, the first-layer ones - not counted - its own)
            res["roofline"] = mfma_roofline(bwd_name, in_step[bwd_name], F * B * nb)
            res["kernels_in_step_us"] = in_step       # inside real (un-captured) steps, HIP events between the launches
            fwd_us = in_step[fwd_name]
            wg_us = in_step.get("hint_wgrad_kernel+hint_wreduce_kernel")
            res["roofline_other_kernels"] = {fwd_name: mfma_roofline(fwd_name, fwd_us, F * B * nb)}
            if wg_us:
                # weight gradients: every weight matrix once, 2*B MACs per element -> F per sample and block as well
                res["roofline_other_kernels"]["hint_wgrad_kernel+hint_wreduce_kernel"] = mfma_roofline("hint_wgrad_kernel", wg_us, F * B * nb)
            # the element-wise view the north_star asks for: compulsory HBM bytes of the forward,
            # 4*(2d+1) B per sample and block, against 8 TB/s
            hbytes = 4.0 * (2 * d + 1) * B * nb
            hb = hbytes / (fwd_us * 1e-6) / 1e9
            res["hbm_view_fwd_kernel"] = {"achieved": hb, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": hb / PEAK_HBM_GBS,
                                          "bytes_per_launch": hbytes}
        if inv_us is not None:
            inv_name = trainer.kernel_names(B)[0].replace("<false", "<true")
            res["inverse_samples_per_sec"] = B * world / (inv_us * 1e-6)
            res["inverse"] = {"what": f"x = f^-1(z), {nb} blocks in one launch of {inv_name}, {B} rows per GPU, no tape",
                              "avg_launch_us": inv_us, "roofline": mfma_roofline(inv_name, inv_us, F * B * nb)}
        if world > 1:
            res["config"]["rows_per_gpu"] = B
            res["config"]["gradient_allreduce"] = "two buckets (second half of the blocks first, beside the rest of part B), RCCL, captured in the step graph"
        if legs:
            res["kernels_hot_loop_us"] = legs         # back-to-back loops of one kernel (caches hot)
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(cfg)
            # NLL of the trained GPU weights re-evaluated by the CPU oracle on the same rows
            from oracle import hint_oracle as orc
            nodes = orc.build_nodes(d, (), cfg["c_internal"])
            xc = x.cpu()
            Jt = torch.zeros(B)
            h = xc
            with torch.no_grad():
                for i, blk in enumerate(flow.blocks):
                    if flow.has_perm(i):
                        h = h @ flow.perms[i].W.cpu()
                    Pc = {k: v.cpu() for k, v in blk.state_dict().items()}
                    h, Ji = orc.block_apply(nodes, Pc, h, (), rev=False)
                    Jt = Jt + Ji
            const = 0.5 * d * math.log(2 * math.pi)
            nll_cpu = float(0.5 * torch.sum(h.double() ** 2, 1).mean() - Jt.double().mean()) + const
            with torch.no_grad():
                zg = flow(x).double().cpu()
                Jg = flow.log_jacobian(run_forward=False).double().cpu()
            nll_gpu = float(0.5 * torch.sum(zg ** 2, 1).mean() - Jg.mean()) + const
            res["nll_cpu_oracle"] = nll_cpu
            res["nll_gpu_f64_reduction"] = nll_gpu
            res["nll_rel_err"] = abs(nll_gpu - nll_cpu) / abs(nll_cpu)
            res["speedup_vs_cpu"] = value / res["cpu_baseline"]["value"]
        print(json.dumps(res), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
