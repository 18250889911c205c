#!/usr/bin/env python3
"""Benchmark of the HINT coupling-flow training step on MI355X (BASELINE.json metric:
training samples/s + mean NLL, UCI POWER d=6, batch 4096 per GPU, 1/2/4/8 GPUs).

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W [--scaling strong]

A "step" = noise + forward + backward of the 8-block flow on this rank's shard, one all-reduce of the
flat gradient arena, fused clamp + Adam (train_unconditional.py:114-144).  --scaling weak (default):
every rank has the workload's batch (4096 rows); --scaling strong: the workload's batch is the GLOBAL
batch, split over the ranks.  Inputs are synthetic N(0,1) rows already resident in HBM.

Timing: after the warm-up the region "exactly --steps steps between barrier + synchronize" is repeated
--reps times (default 7); `value` / `ms_per_step` are the MEDIAN repetition (max over ranks per
repetition), `rep_min_ms` / `rep_max_ms` / `reps` say how far the others lay.

Rank 0 prints ONE JSON line; it also carries the sampling direction (`inverse_samples_per_sec`, the whole
chain x = f^-1(z) in one launch), the roofline of the step's dominant kernel, the CPU baseline and - on
the default single-GPU run - `other_workloads`: BASELINE.json's other configs (cfg 3 GAS, cfg 5 MINIBOONE,
the d = 100 lane of cfg 4 and cfg 4's two-lane conditional model) timed the same way, each with its
dominant kernel's roofline and an NLL cross-check against the CPU oracle.
"""
import argparse
import glob
import json
import math
import os
import re
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

WORKLOADS = {
    # BASELINE.json configs[1]: UCI POWER d=6, 8 recursive coupling blocks, batch 4096, 1 GPU
    "power_hint_8": dict(d=6, n_blocks=8, c_internal=[140, 70, 35, 17], batch=4096, check_scale=0.06),
    "power_hint_4": dict(d=6, n_blocks=4, c_internal=[200, 100, 50, 25], batch=512, check_scale=0.06),
    "gas_hint_8": dict(d=8, n_blocks=8, c_internal=[128, 64, 32, 16], batch=8192, check_scale=0.06),
    "miniboone_hint_10": dict(d=43, n_blocks=10, c_internal=[67, 33, 16, 8], batch=4096, check_scale=0.06),
    "plus_hint_4": dict(d=100, n_blocks=4, c_internal=[224, 112, 56], batch=4096, check_scale=0.03),
    # configs/plus_shape/unconditional_hint_4_3_big.py (h = 512: the s and t nets of the wide nodes run one at a time)
    "plus_hint_4_big": dict(d=100, n_blocks=4, c_internal=[512, 256, 128, 64], batch=4096, check_scale=0.03),
}
# BASELINE.json configs[3]: configs/plus_shape/conditional_hint_4_full.py:58-94 - the two-lane conditional model (x lane: recursive
# block d = 100 + ExternalAffineCoupling given y; y lane: AffineCoupling d = 4), 4 blocks, internal width 224, 4096 rows per GPU
CONDITIONAL = {"conditional_hint_4_full": dict(nx=100, ny=4, n_blocks=4, hidden=224, batch=4096, check_scale=0.03)}
# what the default single-GPU run times besides the headline (BASELINE.json configs[2], [4], [3]: gas_hint_8.py:29-36,
# miniboone_hint_8.py:29-31 at BASELINE's d = 43 / 10 blocks, conditional_hint_4_full.py:58-94 and its x lane alone)
OTHER_WORKLOADS = ["gas_hint_8", "miniboone_hint_10", "plus_hint_4", "conditional_hint_4_full"]
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 = fp32 vector rate
PEAK_HBM_GBS = 8000.0
CHECK_THREADS = 16             # host threads of the oracle cross-checks (one per core is pathological for these small matrices)


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


# ---------------------------------------------------------------------------------------------------
# environment of the run: what RCCL / the HIP build are, as the process sees them
# ---------------------------------------------------------------------------------------------------
def runtime_info(world, use_dist, dist, trainer=None):
    info = {"rccl_world_size": dist.get_world_size() if use_dist else 1,
            "backend": dist.get_backend() if use_dist else "none (one process)",
            "hip_runtime": getattr(torch.version, "hip", None)}
    try:
        info["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception as e:      # noqa: BLE001
        info["rccl_version"] = f"unavailable ({type(e).__name__})"
    try:
        from hint_amd import _lib
        info["library_build"] = _lib.load().hint_build_info().decode()
    except Exception as e:      # noqa: BLE001
        info["library_build"] = f"unavailable ({type(e).__name__})"
    if trainer is not None and hasattr(trainer, "allreduce_plan"):
        info["gradient_allreduce"] = trainer.allreduce_plan()
    return info


def newest_pmc(workload):
    """the committed counter summary of the newest round for a workload: profiles/rNN_pmc_summary.json is the headline's,
    profiles/rNN_pmc_<workload>.json the others' (tools/refresh_profiles.sh + tools/collect_profiles.sh).  Counters are NOT
    measured in this process: the line names the file (`traffic_source`)."""
    pat = "profiles/r*_pmc_summary.json" if workload == "power_hint_8" else f"profiles/r*_pmc_{workload}.json"
    best, best_r = None, -1
    for f in glob.glob(os.path.join(ROOT, pat)):
        m = re.match(r"r(\d+)_pmc_", os.path.basename(f))
        if m and int(m.group(1)) > best_r:
            best, best_r = f, int(m.group(1))
    if best is None:
        return {}, None
    try:
        return json.load(open(best)), os.path.relpath(best, ROOT)
    except Exception:      # noqa: BLE001
        return {}, None


def newest_kernel_stats(workload):
    """{kernel name without spaces: average ns} from the committed `rocprofv3 --kernel-trace --stats` summary of the newest round
    (profiles/rNN_kernel_stats.csv: the headline's bench run; profiles/rNN_kernel_stats_<workload>.csv the others'), and its path"""
    pat = "profiles/r*_kernel_stats.csv" if workload == "power_hint_8" else f"profiles/r*_kernel_stats_{workload}.csv"
    best, best_r = None, -1
    for f in glob.glob(os.path.join(ROOT, pat)):
        m = re.match(r"r(\d+)_kernel_stats", os.path.basename(f))
        if m and int(m.group(1)) > best_r:
            best, best_r = f, int(m.group(1))
    if best is None:
        return {}, None
    import csv
    out = {}
    try:
        for r in csv.DictReader(open(best)):
            name = r["Name"].split("(")[0].replace("void ", "").replace(" ", "")
            out[name] = float(r["AverageNs"])
    except Exception:      # noqa: BLE001
        return {}, None
    return out, os.path.relpath(best, ROOT)


def live_hbm_traffic(workload, steps=10, timeout=240):
    """HBM bytes per launch of every hint_* kernel of `workload`'s training step, MEASURED IN THIS RUN: two child processes
    `rocprofv3 --kernel-trace --pmc <FETCH_SIZE | WRITE_SIZE> -- python3 tools/steps.py <workload> <steps>` (separate passes, as
    MI355X_MICROARCH.md prescribes; children, started from /tmp - the profiler never wraps this process), corrected the same way
    as tools/pmc_summary.py: (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 tallies 128-byte reads as 64).  Returns
    ({kernel name without spaces: bytes}, note) or ({}, why not)."""
    import csv, shutil, subprocess, tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {}, "rocprofv3 not found"
    per = {}
    env = dict(os.environ, TMPDIR="/tmp")
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="hint_pmc_", dir="/tmp")
        try:
            r = subprocess.run([exe, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "--", "python3",
                                os.path.join(ROOT, "tools", "steps.py"), workload, str(steps)],
                               cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout)
            if b"\nok " not in b"\n" + r.stdout:
                return {}, f"the {counter} pass did not finish ({r.returncode})"
            acc = {}
            for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
                for row in csv.DictReader(open(f)):
                    if row["Counter_Name"] != counter:
                        continue
                    k = row["Kernel_Name"].split("(")[0].replace("void ", "").replace(" ", "")
                    acc.setdefault(k, []).append(float(row["Counter_Value"]))
            for k, v in acc.items():
                per.setdefault(k, {})[counter] = sum(v) / len(v)
        except Exception as e:      # noqa: BLE001
            return {}, f"the {counter} pass failed: {type(e).__name__}"
        finally:
            shutil.rmtree(d, ignore_errors=True)
    out = {k: (2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024.0 for k, v in per.items() if "hint_" in k and len(v) == 2}
    return out, (f"measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE, two child processes of tools/steps.py "
                 f"({steps} un-captured steps each), (2 * FETCH_SIZE + WRITE_SIZE) * 1024 per launch")


def timed_reps(step, barrier, steps, reps, use_dist, dist, dev):
    """`reps` repetitions of: barrier + synchronize, exactly `steps` steps, barrier + synchronize; per repetition the
    max over ranks.  Returns the list of elapsed seconds."""
    out = []
    for _ in range(reps):
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        barrier()
        el = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        out.append(el)
    return out


def rep_stats(times, steps):
    med = statistics.median(times)
    return med, {"reps": len(times), "rep_min_ms": min(times) / steps * 1e3, "rep_max_ms": max(times) / steps * 1e3,
                 "rep_ms": [round(t / steps * 1e3, 5) for t in times],
                 "timing_note": "value / ms_per_step = the median of `reps` repetitions of the bracketed --steps region"}


# ---------------------------------------------------------------------------------------------------
# NLL cross-checks against the CPU oracle at weights where the flow is far from the identity
# ---------------------------------------------------------------------------------------------------
def nll_check_flow(cfg, B, dev, seed=0):
    """mean NLL of the chained fast path (FlowTrainer's launches: the plan variant the timed batch size runs on) against
    oracle/hint_oracle.py's OracleFlow in float64 on the same float32 weights (randn * check_scale: log-dets of a few nats)
    and the same B rows.  north_star bar: 1e-4 relative."""
    import hint_amd
    from oracle import hint_oracle as orc
    d, nb, widths, scale = cfg["d"], cfg["n_blocks"], cfg["c_internal"], cfg["check_scale"]
    ref = orc.OracleFlow(d, nb, widths, seed=seed, init_scale=scale, dtype=torch.float64)
    flow = hint_amd.HintFlow(d, nb, widths)
    for i, blk in enumerate(flow.blocks):
        blk.load_state_dict({k: v.float() for k, v in ref.params[i].items()})
        if ref.perms[i] is not None:
            flow.perms[i].W.copy_(ref.perms[i].float())
        ref.params[i] = {k: v.float().double() for k, v in ref.params[i].items()}      # the oracle sees the float32 weights
    ref.perms = [None if p is None else p.float().double() for p in ref.perms]
    flow = flow.to(dev)
    g = torch.Generator().manual_seed(7)
    x = torch.randn(B, d, generator=g)
    tr = hint_amd.FlowTrainer(flow, noise=0.0, use_graph=False, lr=0.0, weight_decay=0.0)
    l0, l1 = tr.step(x.to(dev))
    const = 0.5 * d * math.log(2 * math.pi)
    nll_gpu = float(l0) + float(l1) + const
    nt = torch.get_num_threads()
    torch.set_num_threads(min(CHECK_THREADS, os.cpu_count() or 1))
    try:
        with torch.no_grad():
            z, J = ref.forward(x.double())
        nll_ref = ref.nll(z, J)
    finally:
        torch.set_num_threads(nt)
    del tr, flow
    return {"nll_gpu": nll_gpu, "nll_cpu_oracle_f64": nll_ref, "nll_rel_err": abs(nll_gpu - nll_ref) / abs(nll_ref),
            "rows": B, "weights": f"{scale} * randn (seed {seed})",
            "path": "FlowTrainer.step (chained launches, lr = 0, no noise) vs OracleFlow.forward in float64"}


def conditional_oracle_losses(model, mods, x, y, ny):
    """(0.5 |z|^2 mean, -log|det J| mean) of the two-lane graph assembled from oracle blocks in float64"""
    from oracle import hint_oracle as orc
    nb = model.n_blocks
    P = [{k: v.detach().cpu().double() for k, v in m.state_dict().items()} for m, _ in mods]
    nodes = [tree_nodes(m.tree, dc) for m, dc in mods]
    xo, yo = x.double(), y.double()
    J = 0
    with torch.no_grad():
        for i in range(nb):
            if i > 0:
                yo = yo @ model.perm_y[i].W.cpu().double(); xo = xo @ model.perm_x[i].W.cpu().double()
            xo, j = orc.block_apply(nodes[3 * i], P[3 * i], xo, [], clamp=mods[3 * i][0].tree.clamp); J = J + j
            xo, j = orc.block_apply(nodes[3 * i + 1], P[3 * i + 1], xo, [yo], clamp=mods[3 * i + 1][0].tree.clamp); J = J + j
            yo, j = orc.block_apply(nodes[3 * i + 2], P[3 * i + 2], yo, [], clamp=mods[3 * i + 2][0].tree.clamp); J = J + j
    return float(0.5 * (torch.cat([xo, yo], -1) ** 2).sum(1).mean()), float(-J.mean())


def conditional_mods(model, cfg):
    mods = []
    for i in range(cfg["n_blocks"]):
        mods += [(model.hac_x[i], 0), (model.ac_y_to_x[i], cfg["ny"]), (model.ac_y[i], 0)]
    return mods


def nll_check_conditional(cfg, B, dev, seed=0):
    """the same for the two-lane conditional model: ConditionalFlowTrainer's fast path (graph replay) against the oracle
    composition in float64"""
    import hint_amd
    torch.manual_seed(seed)
    model = hint_amd.ConditionalHintFlow(cfg["nx"], cfg["ny"], cfg["n_blocks"], cfg["hidden"]).to(dev)
    gw = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in model.parameters():
            p.data = (cfg["check_scale"] * torch.randn(p.shape, generator=gw)).to(dev)
    g = torch.Generator().manual_seed(7)
    x, y = torch.randn(B, cfg["nx"], generator=g), torch.randn(B, cfg["ny"], generator=g)
    tr = hint_amd.ConditionalFlowTrainer(model, noise=0.0, lr=0.0, weight_decay=0.0, use_graph=True)
    l0, l1 = tr.step(x.to(dev), y.to(dev))
    const = 0.5 * (cfg["nx"] + cfg["ny"]) * math.log(2 * math.pi)
    nll_gpu = float(l0) + float(l1) + const
    nt = torch.get_num_threads()
    torch.set_num_threads(min(CHECK_THREADS, os.cpu_count() or 1))
    try:
        r0, r1 = conditional_oracle_losses(model, conditional_mods(model, cfg), x, y, cfg["ny"])
    finally:
        torch.set_num_threads(nt)
    nll_ref = r0 + r1 + const
    return {"nll_gpu": nll_gpu, "nll_cpu_oracle_f64": nll_ref, "nll_rel_err": abs(nll_gpu - nll_ref) / abs(nll_ref),
            "rows": B, "weights": f"{cfg['check_scale']} * randn (seed {seed})",
            "path": "ConditionalFlowTrainer.step (hipGraph replay, lr = 0, no noise) vs the two-lane graph of oracle blocks in float64"}


# ---------------------------------------------------------------------------------------------------
# CPU baselines
# ---------------------------------------------------------------------------------------------------
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


def cpu_baseline_conditional(model, cfg, x, y, budget_s=14.0):
    """the conditional model's training step on the host cores: oracle blocks (plain torch CPU ops + autograd) composed as
    the two-lane graph, at the GPU's batch size (the rows the GPU step ran on), a bounded number of steps"""
    from oracle import hint_oracle as orc
    mods = conditional_mods(model, cfg)
    nb, B = cfg["n_blocks"], x.shape[0]
    P = [{k: v.detach().cpu().clone().requires_grad_(True) for k, v in m.state_dict().items()} for m, _ in mods]
    nodes = [tree_nodes(m.tree, dc) for m, dc in mods]
    Wy = [model.perm_y[i].W.cpu() if i > 0 else None for i in range(nb)]
    Wx = [model.perm_x[i].W.cpu() if i > 0 else None for i in range(nb)]
    opt = torch.optim.Adam([p for d_ in P for p in d_.values()], lr=0.01 * 3e-2, betas=(0.9, 0.95), eps=1e-4, weight_decay=1.86e-5)
    xc, yc = x.cpu(), y.cpu()

    def cpu_step():
        opt.zero_grad()
        xo, yo = xc + 0.01 * torch.randn_like(xc), yc
        J = 0
        for i in range(nb):
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
    for nt in sorted({min(ncores, v) for v in (16, 32, 64)}):
        torch.set_num_threads(nt)
        cpu_step()
        t1 = time.perf_counter(); cpu_step(); t = time.perf_counter() - t1
        if t < best_t:
            best_nt, best_t = nt, t
    torch.set_num_threads(best_nt)
    n = int(max(2, min(50, budget_s / best_t)))
    t1 = time.perf_counter()
    for _ in range(n):
        cpu_step()
    dt = time.perf_counter() - t1
    return dict(value=B * n / dt, unit="samples/s", cores=best_nt, kind="port",
                sample=f"{n} training steps of {B} rows - the GPU's batch - ({dt:.1f} s): oracle blocks composed as the two-lane graph "
                       f"(torch CPU ops + autograd + clamp + Adam), best of 16/32/64 threads = {best_nt}; host has {ncores} cores")


# ---------------------------------------------------------------------------------------------------
# the workloads
# ---------------------------------------------------------------------------------------------------
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


def run_flow(args, name, rank, world, dev, use_dist, dist, headline):
    """one unconditional workload on hint_amd.FlowTrainer; returns rank 0's result dict (None on the other ranks)"""
    import hint_amd
    cfg = WORKLOADS[name]
    d, B = cfg["d"], cfg["batch"]
    if args.batch > 0 and headline:
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
    spg = args.steps_per_graph if (headline and args.steps_per_graph > 1 and not use_dist and not args.no_graph) else 1
    x, _ = trainer.input_buffers(x)
    if spg > 1:                              # K resident batches (K different shards of synthetic data) per replay
        xs = torch.randn(spg, B, d, generator=gx).to(dev)
        xs, _ = trainer.input_buffers_many(xs)
    for _ in range(args.warmup):
        trainer.step(x)
    if spg > 1:
        trainer.step_many(xs)

    if spg > 1:
        def region_step():
            trainer.step_many(xs)            # spg full iterations (re-pack, forward, backward, clamp+Adam each)
        steps_calls = args.steps // spg
        times = timed_reps(region_step, barrier, steps_calls, args.reps, use_dist, dist, dev)
        steps_timed = steps_calls * spg
    else:
        times = timed_reps(lambda: trainer.step(x), barrier, args.steps, args.reps, use_dist, dist, dev)
        steps_timed = args.steps
    elapsed, reps_info = rep_stats(times, steps_timed)

    ms_per_step = elapsed / steps_timed * 1e3
    value = B * world * steps_timed / elapsed
    l0, l1 = trainer.last_losses()
    loss_last = float(l0) + float(l1)

    # the step's launches inside real (un-captured) steps, HIP events between them: what the roofline
    # is priced on.  Every rank runs them (the steps contain the gradient all-reduce).
    in_step = {}
    if trainer._chainable:
        for k in range(3 + 20):
            t = trainer.timed_step(x)
            if k >= 3:
                for n_, v_ in t.items():
                    in_step[n_] = in_step.get(n_, 0.0) + v_ / 20
    # mean NLL of the weights as they are NOW (after every step above), on this rank's resident rows: the same weights the
    # CPU oracle re-evaluates below (`nll_rel_err`)
    nll = trainer.nll(x)

    # the sampling direction: x = f^-1(z) through the whole chain, one launch (hint_chain_inverse), on the same
    # resident rows; HIP events on the launch stream
    inv_us = None
    if trainer._chainable and headline:
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

    if rank != 0:
        return None
    F = flops_per_sample_block(d, cfg["c_internal"])
    nb = cfg["n_blocks"]                           # blocks one launch processes
    legs = kernel_legs(trainer, x) if (args.legs and headline) else {}
    res = {
        "metric": "train_samples_per_sec", "value": value, "unit": "samples/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{name}: d={d}, {nb} recursive coupling blocks, "
                               f"c_internal={cfg['c_internal']}, batch {B} per GPU",
                   "global_batch": B * world, "parallelism": f"dp{world}", "hip_graph": not args.no_graph, "steps_per_graph": spg,
                   "input": "batch resident in the captured step's input buffer (no per-step copy)",
                   **runtime_info(world, use_dist, dist, trainer)},
        "mean_nll_nats": nll, "last_step_loss": loss_last,
    }
    res.update(reps_info)
    # HBM traffic per launch comes from rocprofv3 PMC passes run separately (tools/refresh_profiles.sh); it is NOT measured in
    # this process - the line says where it was read from
    pmc, pmc_file = newest_pmc(name) if B == cfg["batch"] else ({}, None)
    kstats, kstats_file = newest_kernel_stats(name) if B == cfg["batch"] else ({}, None)
    pmc_build = pmc.get("_build", {}) if isinstance(pmc, dict) else {}
    lib_info = res["config"].get("library_build", "")
    lib_stamp = re.search(r"src ([0-9a-f]+)", lib_info)
    lib_stamp = lib_stamp.group(1) if lib_stamp else None

    def pmc_entry(kernel):
        # rocprof's kernel names carry the template arguments: the full name must match (hint_bwd_kernel is a prefix of
        # hint_bwd_kernel_n3 and hint_bwd_kernel_fly); a summary of an older build, whose instances had fewer template
        # arguments, matches when exactly one of its kernels has the same base name and leading arguments
        want = kernel.replace(" ", "")
        keys = {k.replace(" ", ""): v for k, v in pmc.items() if k != "_build"}
        if want in keys:
            return keys[want]
        base, _, targs = want.partition("<")
        cand = [v for k, v in keys.items() if k.partition("<")[0] == base and
                (not targs or not k.partition("<")[2] or targs.startswith(k.partition("<")[2].rstrip(">")))]
        return cand[0] if len(cand) == 1 else None

    def mfma_roofline(kernel, us, flops):
        ach = flops / (us * 1e-6) / 1e12
        e = pmc_entry(kernel) or {}
        issued = e.get("sq", {}).get("SQ_INSTS_MFMA")
        # the same kernel's average launch in the committed rocprofv3 --kernel-trace --stats summary: the second opinion on `frac`
        ns = kstats.get(kernel.replace(" ", ""))
        src = None
        if e:
            src = (f"{pmc_file} (rocprofv3 --pmc passes of tools/steps.py, builder-run, not measured in this run; counters taken on library "
                   f"src {pmc_build.get('src_stamp')}, git {pmc_build.get('git_head')}; this run's library: src {lib_stamp}; "
                   f"same sources: {pmc_build.get('src_stamp') is not None and pmc_build.get('src_stamp') == lib_stamp})")
        lv = live.get(kernel.replace(" ", ""))
        return {"bound": "mfma", "kernel": kernel, "achieved": ach, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": ach / PEAK_F32_MFMA_TFLOPS, "traffic": lv if lv is not None else e.get("hbm_bytes_per_launch"),
                "traffic_source": live_note if lv is not None else src,
                "traffic_committed": e.get("hbm_bytes_per_launch"), "traffic_committed_source": src if lv is not None else None,
                "frac_rocprof": (flops / (ns * 1e-9) / 1e12 / PEAK_F32_MFMA_TFLOPS) if ns else None,
                "frac_rocprof_source": f"{kstats_file}: average launch {ns / 1e3:.1f} us" if ns else None,
                "avg_launch_us": us, "timing": "HIP events around the launch inside 20 un-captured training steps (FlowTrainer.timed_step); "
                                               "`value` is timed on graph replays",
                "algorithmic_flops_per_launch": flops, "blocks_per_launch": nb,
                "mfma_issued_over_algorithmic": (issued * 2048.0 / flops) if issued else None}

    # HBM bytes per launch measured in THIS run (headline, one process): child processes under rocprofv3's counters
    live, live_note = ({}, None)
    if headline and world == 1 and not use_dist and B == cfg["batch"] and not args.no_live_traffic:
        live, live_note = live_hbm_traffic(name)
    if live_note and not live:
        res["live_traffic_note"] = live_note
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
        res["roofline_whole_step"] = {"achieved": 3.0 * F * B * nb / (ms_per_step * 1e-3) / 1e12, "unit": "TFLOP/s",
                                      "frac": 3.0 * F * B * nb / (ms_per_step * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS}
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
    if legs:
        res["kernels_hot_loop_us"] = legs         # back-to-back loops of one kernel (caches hot)
    if world == 1 and not args.no_cpu_baseline:
        if headline:
            res["cpu_baseline"] = cpu_baseline(dict(cfg, batch=B))
            res["speedup_vs_cpu"] = value / res["cpu_baseline"]["value"]
            # NLL of the trained GPU weights (0.005 * randn + the steps above: a flow next to the identity) re-evaluated by
            # the CPU oracle on the same rows - the same weights `mean_nll_nats` was taken at
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
            res["nll_trained_weights"] = {"nll_cpu_oracle": nll_cpu, "nll_gpu_f64_reduction": nll_gpu,
                                          "nll_rel_err": abs(nll_gpu - nll_cpu) / abs(nll_cpu),
                                          "note": "the weights of `mean_nll_nats` (0.005 * randn + this run's steps): a near-identity flow, "
                                                  "a weak check - `nll_check` below is the one that can fail"}
        # the check that can fail: weights far from the identity (log-dets of a few nats), the fast path's own loss sums
        res["nll_check"] = nll_check_flow(cfg, B, dev)
        res["nll_rel_err"] = res["nll_check"]["nll_rel_err"]
    return res


def run_conditional(args, name, rank, world, dev, use_dist, dist, headline):
    """the conditional two-lane model (BASELINE.json configs[3]) on hint_amd.ConditionalFlowTrainer: one step = noise, both
    lanes forward and backward (train_conditional.py:120-150), one all-reduce of the flat gradient arena, fused clamp + Adam"""
    import hint_amd
    cfg = CONDITIONAL[name]
    B = args.batch if (args.batch > 0 and headline) else cfg["batch"]
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

    out = [None]

    def step():
        out[0] = tr.step(x, y)
    for _ in range(args.warmup):
        step()
    times = timed_reps(step, barrier, args.steps, args.reps, use_dist, dist, dev)
    elapsed, reps_info = rep_stats(times, args.steps)
    if rank != 0:
        return None
    l0, l1 = out[0]
    # algorithmic FLOPs of a step: forward MACs of every subnet (SURVEY §8d's F, summed over the three modules of a
    # block) x 3 (forward, dX, dW)
    mods = conditional_mods(model, cfg)
    F = 0
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
                   "global_batch": B * world, "parallelism": f"dp{world}", "hip_graph": not args.no_graph,
                   **runtime_info(world, use_dist, dist, tr)},
        "last_step_loss": float(l0) + float(l1),
        "roofline": {"bound": "mfma", "kernel": "whole step (the x lane's hint_apply / hint_bwd / hint_wgrad launches dominate)",
                     "achieved": ach, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_F32_MFMA_TFLOPS, "traffic": None,
                     "algorithmic_flops_per_step": 3.0 * F * B,
                     "timing": "wall clock of the timed steps (graph replays), all launches of a step"},
    }
    res.update(reps_info)
    if world == 1 and not args.no_cpu_baseline:
        if headline:
            res["cpu_baseline"] = cpu_baseline_conditional(model, cfg, x, y)
            res["speedup_vs_cpu"] = res["value"] / res["cpu_baseline"]["value"]
        res["nll_check"] = nll_check_conditional(cfg, B, dev)
        res["nll_rel_err"] = res["nll_check"]["nll_rel_err"]
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--reps", type=int, default=7, help="repetitions of the timed --steps region (the median is reported)")
    ap.add_argument("--workload", default="power_hint_8", choices=sorted(WORKLOADS) + sorted(CONDITIONAL))
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--steps-per-graph", type=int, default=1,
                    help="training iterations per hipGraph replay (FlowTrainer.step_many; one process only)")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU baseline and the oracle cross-checks")
    ap.add_argument("--no-other-workloads", action="store_true",
                    help="default single-GPU run: do not time BASELINE's other configs after the headline")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="default single-GPU run: do not measure the headline kernels' HBM bytes under rocprofv3 (two child processes); "
                         "roofline.traffic then comes from the committed counter summary")
    ap.add_argument("--no-module-path", action="store_true",
                    help="default single-GPU run: do not time the drop-in module route (extra.module_path) after the headline")
    ap.add_argument("--batch", type=int, default=0,
                    help="rows per GPU instead of the workload's (small-batch measurements; the config line says so)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: the workload's batch per rank; strong: the workload's batch is the global batch")
    ap.add_argument("--legs", action="store_true",
                    help="also time every kernel in a hot back-to-back loop (single-block and chained launches)")
    args = ap.parse_args()
    args.reps = max(1, args.reps)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU fallback")
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {torch.cuda.device_count()} devices are visible")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    use_dist = world > 1 or os.environ.get("HINT_FORCE_DIST") == "1"   # the latter: exercise RCCL on one GPU
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import hint_amd  # noqa: F401
    run = run_conditional if args.workload in CONDITIONAL else run_flow
    res = run(args, args.workload, rank, world, dev, use_dist, dist, True)

    # the other BASELINE configs, same protocol (their own warm-up, the same --steps / --reps), after the headline's timed
    # region: single-GPU default run only - the multi-GPU lines stay what the contract asks for
    if rank == 0 and world == 1 and not use_dist and args.workload == "power_hint_8" and args.batch == 0 \
            and args.scaling == "weak" and not args.no_other_workloads:
        others = {}
        keep = ("value", "unit", "ms_per_step", "reps", "rep_min_ms", "rep_max_ms", "roofline", "roofline_whole_step",
                "kernels_in_step_us", "mean_nll_nats", "nll_check", "nll_rel_err", "last_step_loss")
        for w in OTHER_WORKLOADS:
            t0 = time.perf_counter()
            try:
                torch.cuda.empty_cache()
                r = (run_conditional if w in CONDITIONAL else run_flow)(args, w, rank, world, dev, use_dist, dist, False)
                o = {k: r[k] for k in keep if k in r}
                o["workload"] = r["config"]["workload"]
                o["steps"], o["warmup"] = args.steps, args.warmup
            except Exception as e:      # noqa: BLE001 - the headline line must come out whatever happens here
                o = {"error": f"{type(e).__name__}: {e}"}
            o["bench_wall_s"] = round(time.perf_counter() - t0, 1)
            others[w] = o
        res["other_workloads"] = others
    # the DROP-IN module route (north_star: "drops into train_unconditional.py unchanged"): the reference loop body verbatim on
    # hint_amd's nn.Modules with torch.optim.Adam and the per-parameter clamp - outside the headline's timed region
    if rank == 0 and world == 1 and not use_dist and args.workload == "power_hint_8" and args.batch == 0 \
            and args.scaling == "weak" and not args.no_module_path:
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import module_path
            torch.cuda.empty_cache()
            fused = module_path.run("power_hint_8", steps=60, warmup=10, per_block=False, dev=dev)
            walk = module_path.run("power_hint_8", steps=60, warmup=10, per_block=True, dev=dev)
            fast_ms = sum(v for k, v in res.get("kernels_in_step_us", {}).items() if k.startswith("hint_")) * 1e-3
            for r in (fused, walk):
                r["loop_own_ms"] = r["ms_per_step"] - r["hint_amd_host_ms"]
                if fast_ms > 0:
                    r["hint_amd_share_over_fast_path_kernels"] = max(r["hint_amd_host_ms"], r["hint_amd_device_ms"]) / fast_ms
            res.setdefault("extra", {})["module_path"] = {
                "what": "train_unconditional.py:114-144 verbatim (zero_grad, noise, model(x), log_jacobian, two loss terms + .item(), backward, "
                        "per-parameter clamp_, torch.optim.Adam.step) on hint_amd.HintFlow; hint_amd_host_ms = host clock inside hint_amd's "
                        "entry points, hint_amd_device_ms = HIP events around their launches (first launch to last, host gaps between them included), loop_own_ms = the rest of the step (autograd engine, "
                        "the loop's ATen work: 288 clamp_ calls, foreach Adam); the step is host-bound",
                "fast_path_kernels_ms": fast_ms, "hintflow": fused, "block_walk": walk}
        except Exception as e:      # noqa: BLE001 - the headline line must come out whatever happens here
            res.setdefault("extra", {})["module_path"] = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0:
        print(json.dumps(res), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
