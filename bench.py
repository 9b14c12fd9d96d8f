#!/usr/bin/env python3
"""Benchmark of the tree-expansion hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--queries Q] [--config 2|3|4]

A "step" is one pass of the hot path over one batch of queries that is already resident in HBM
(grid, packed sample streams): re-arm the trees, run the expansion kernel to completion
(sample -> nearest -> line of sight -> choose parent -> insert for all n iterations, then
go2goal), and for N > 1 all-gather the result slabs over RCCL.  Default workload = BASELINE.json
configs[1]: RRT*, 1024x1024 noise grid, n = 50000, r_rewire = 64, one query per GPU.
Metric: nodes expanded per second (inserted tree nodes / wall time), whole job over all ranks.

For N > 1 the driver launches one rank per GPU with torch.distributed.run; queries are sharded
query -> rank with no data-path collective (weak scaling: per-GPU work fixed).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)

CONFIGS = {
    # BASELINE.json configs[1..3]
    2: dict(name="RRT* 1024x1024 n=50000 r_rewire=64", alg=1, grid=1024, n=50000, r_rewire=64, r_goal=None, queries=1),
    3: dict(name="Informed-RRT* 1024x1024 n=50000 r_rewire=64 r_goal=12", alg=2, grid=1024, n=50000, r_rewire=64, r_goal=12, queries=1),
    4: dict(name="batch of independent RRT* queries 1024x1024 n=20000 r_rewire=64 (64 per GPU)", alg=1, grid=1024, n=20000,
            r_rewire=64, r_goal=None, queries=64),
}


def algorithmic_bytes(res):
    """per iteration 8*j_i node-coordinate bytes (fused NN + radius pass) + 1 B per
    line-of-sight cell nearest->new; per accepted iteration 8 B vcost per near-set entry + 1 B per
    choose-parent line-of-sight cell + 20 B insert."""
    return 8 * res.sum_j + res.sum_cells_nn + 8 * res.sum_near + res.sum_cells_cand + 20 * (res.j - 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS))
    ap.add_argument("--queries", type=int, default=None, help="queries per GPU (default: the config's)")
    ap.add_argument("--n", type=int, default=None)
    ap.add_argument("--team", type=int, default=None, choices=[1, 2, 4, 8, 16, 32, 64],
                    help="cap on the CUs per query (default: the largest team for which all teams are resident together)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-batched", action="store_true", help="skip the informational batched leg (configs[3] share of one GPU)")
    args = ap.parse_args()

    # rank 0 prints exactly ONE line on stdout: route everything else that writes to fd 1 (RCCL's banner, library
    # chatter) to stderr until the JSON line is ready
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
        args.gpus = world

    import torch

    torch.cuda.set_device(local_rank)
    dist = None
    use_dist = world > 1 or "RANK" in os.environ  # launched by torch.distributed.run (also with one rank: exercises the gather)
    if use_dist:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")

        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    from rrtplanner_amd import _ffi, hostprep, multi
    from rrtplanner_amd.oggen import perlin_occupancygrid, random_connected_pair

    cfg = dict(CONFIGS[args.config])
    if args.queries:
        cfg["queries"] = args.queries
    if args.n:
        cfg["n"] = args.n
    Q, n, alg = cfg["queries"], cfg["n"], cfg["alg"]

    # ---- synthetic workload: seeded noise grid, start/goal in one free component ----
    og = perlin_occupancygrid(cfg["grid"], cfg["grid"], thresh=0.33, seed=1)
    og8 = hostprep.og_nonzero(og)
    free = np.argwhere(og == 0)
    sg_rng = np.random.default_rng(7)
    pairs = [random_connected_pair(og, sg_rng) for _ in range(Q * world)]  # query g = rank + world*slot
    r2 = hostprep.radius_threshold(cfg["r_rewire"])
    gd2 = hostprep.goal_threshold(cfg["r_goal"]) if cfg["r_goal"] is not None else 0

    ctx = _ffi.Context(local_rank)
    ctx.set_grid(og8)
    batch = _ffi.Batch(ctx, Q, n, team=args.team)
    keep, rngs, states = [], [], []
    for slot in range(Q):
        g = rank + world * slot
        xs, xg = pairs[g]
        rng = np.random.default_rng(g)  # planner seed = global query index
        states.append(rng.bit_generator.state)
        samples = hostprep.draw_free_samples(rng, free, n)
        Cm = hostprep.rotation_to_world_frame(xs, xg) if alg == 2 else None
        qu, k = _ffi.make_query(alg, n, xs, xg, samples, r2_rewire=r2, goal_d2=gd2, Cmat=Cm)
        keep.append(k)
        rngs.append(rng)
        batch.set_query(slot, qu)

    ub_cache = {}

    def one_step():
        """Re-arm, expand to completion (Informed: resume after the host hands over the unit-ball
        stream for the iterations past the switch point), return kernel milliseconds."""
        batch.rearm()
        batch.launch()
        batch.sync()
        ms = batch.elapsed_ms()
        if alg == 2:
            pending = False
            for slot in range(Q):
                r = batch.get_result(slot, arrays=False)
                if r.c.status == _ffi.RRT_NEED_UNITBALL:
                    if slot not in ub_cache:  # same stream every step: draw once
                        rng = rngs[slot]
                        rng.bit_generator.state = states[slot]
                        hostprep.draw_free_samples(rng, free, r.c.i_switch)
                        ub_cache[slot] = (hostprep.draw_unitball(rng, n - r.c.i_switch), r.c.i_switch)
                    batch.set_unitball(slot, ub_cache[slot][0], ub_cache[slot][1])
                    pending = True
            if pending:
                batch.launch()
                batch.sync()
                ms += batch.elapsed_ms()
        return ms

    gather_buf = None
    if use_dist:
        ptr, nbytes = batch.result_block()
        gather_buf = torch.as_tensor(multi.DeviceBlock(ptr, nbytes), device=f"cuda:{local_rank}")

    def sync_all():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    gathered = None
    for _ in range(args.warmup):
        one_step()
        if use_dist:
            gathered = multi.gather_result_blocks(gather_buf)
            torch.cuda.current_stream().synchronize()
    sync_all()
    t0 = time.perf_counter()
    kern_ms = 0.0
    for _ in range(args.steps):
        kern_ms += one_step()
        if use_dist:
            gathered = multi.gather_result_blocks(gather_buf)
            torch.cuda.current_stream().synchronize()  # the next step rewrites the slab the collective is reading
    sync_all()
    dt = time.perf_counter() - t0
    if gathered is not None and rank == 0:  # sanity of the collective: rank 0's own slab must come back unchanged
        own = torch.as_tensor(multi.DeviceBlock(*batch.result_block()), device=f"cuda:{local_rank}")
        if gathered.shape[0] != world or not torch.equal(gathered[0], own):
            raise SystemExit("result gather returned a wrong slab")
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    results = [batch.get_result(slot, arrays=False) for slot in range(Q)]
    nodes_local = sum(r.c.j - 1 for r in results)
    iters_local = Q * n
    bytes_local = sum(algorithmic_bytes(r.c) for r in results)
    bad = [r.c.status for r in results if r.c.status != 0]
    if use_dist:
        agg = torch.tensor([nodes_local, iters_local, len(bad)], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(agg)
        nodes_total, iters_total, nbad = float(agg[0]), float(agg[1]), int(agg[2])
    else:
        nodes_total, iters_total, nbad = float(nodes_local), float(iters_local), len(bad)

    if rank == 0:
        kern_avg_ms = kern_ms / args.steps
        achieved = bytes_local / (kern_avg_ms * 1e-3) / 1e9
        out = {
            "metric": "RRT* nodes-expanded/s on 1024x1024 Perlin grid; achieved HBM GB/s",
            "value": nodes_total * args.steps / dt,
            "unit": "nodes/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int16x2 coordinates / u32 squared distances / f64 costs",
            "data": "synthetic",
            "config": {"workload": f"BASELINE.json configs[{args.config - 1}]: {cfg['name']}", "queries_per_gpu": Q, "n": n,
                       "grid": [cfg["grid"], cfg["grid"]], "free_fraction": float((og == 0).mean()),
                       "iters_per_s": iters_total * args.steps / dt, "unfinished_queries": nbad,
                       "cus_per_query": batch.team()[0] + (1 if batch.pipelined() else 0), "pipelined": batch.pipelined(),
                       "team_fallbacks": batch.team()[1]},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "rrt_expand_block_kernel", "kernel_ms": kern_avg_ms, "algorithmic_bytes_per_launch": int(bytes_local)},
        }
        tr = measured_traffic(args.config, Q, n)
        if tr is not None:
            out["roofline"]["traffic"] = tr
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(og8, cfg, pairs[0], free, states[0], ub_cache.get(0))
        if world == 1 and args.config == 2 and not args.no_batched:
            out["batched"] = batched_leg(ctx, og, free, _ffi, hostprep)
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    batch.close()
    ctx.close()


def measured_traffic(config, Q, n):
    """HBM bytes per launch of rrt_expand_block_kernel from the committed rocprofv3 PMC passes (profiles/), when they
    were taken on this exact workload; None otherwise (the counters cannot be read from inside the bench)."""
    path = os.path.join(ROOT, "profiles", "r01_traffic.json")
    try:
        with open(path) as f:
            t = json.load(f)
        for e in t["entries"]:
            if e["config"] == config and e["queries_per_gpu"] == Q and e["n"] == n:
                return e["hbm_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        pass
    return None


def batched_leg(ctx, og, free, _ffi, hostprep):
    """Informational: one GPU's share of BASELINE.json configs[3] (64 independent RRT* queries, n = 20000), every query on
    its own team of CUs.  Not the headline value."""
    from rrtplanner_amd.oggen import random_connected_pair

    cfg = CONFIGS[4]
    Q, n = cfg["queries"], cfg["n"]
    b = _ffi.Batch(ctx, Q, n)
    sg = np.random.default_rng(7)
    keep = []
    for q in range(Q):
        xs, xg = random_connected_pair(og, sg)
        s = hostprep.draw_free_samples(np.random.default_rng(q), free, n)
        qu, k = _ffi.make_query(cfg["alg"], n, xs, xg, s, r2_rewire=hostprep.radius_threshold(cfg["r_rewire"]))
        keep.append(k)
        b.set_query(q, qu)
    b.launch(); b.sync()
    steps, t0, kms = 3, time.perf_counter(), 0.0
    for _ in range(steps):
        b.rearm(); b.launch(); b.sync()
        kms += b.elapsed_ms()
    dt = time.perf_counter() - t0
    res = [b.get_result(q, arrays=False) for q in range(Q)]
    nodes = sum(r.c.j - 1 for r in res)
    by = sum(algorithmic_bytes(r.c) for r in res)
    cus, fallbacks = b.team()
    cus += 1 if b.pipelined() else 0
    b.close()
    ach = by / (kms / steps * 1e-3) / 1e9
    return {"workload": "BASELINE.json configs[3] share of one GPU: " + cfg["name"], "value": nodes * steps / dt, "unit": "nodes/s",
            "ms_per_step": dt / steps * 1e3, "cus_per_query": cus, "team_fallbacks": fallbacks, "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                          "frac": ach / HBM_PEAK_GBS, "kernel_ms": kms / steps}}


def cpu_baseline(og8, cfg, pair, free, state0, ub):
    """The CPU oracle (oracle/rrt_oracle.c, 1 thread, scalar) on query 0 of the same workload."""
    import oracle
    from rrtplanner_amd import hostprep

    n, alg = cfg["n"], cfg["alg"]
    xs, xg = pair
    rng = np.random.default_rng(0)
    rng.bit_generator.state = state0
    samples = hostprep.draw_free_samples(rng, free, n)
    r2 = hostprep.radius_threshold(cfg["r_rewire"])
    kw = dict(r2_rewire=r2, r_goal=cfg["r_goal"] or 0.0, logs=False)
    if alg == 2:
        kw["Cmat"] = hostprep.rotation_to_world_frame(xs, xg)
        if ub is not None:
            kw.update(unitball=ub[0], ub_offset=ub[1])
    reps, t0, nodes = 0, time.perf_counter(), 0
    while reps < 1 or (time.perf_counter() - t0 < 10.0 and reps < 8):
        st, r = oracle.plan(og8, n, alg, xs, xg, samples, **kw)
        nodes += r.j - 1
        reps += 1
    dt = time.perf_counter() - t0
    out = {"value": nodes / dt, "unit": "nodes/s", "cores": 1, "kind": "port",
           "sample": f"query 0 of the same workload (n={n}), {reps} repetition(s), {dt:.1f} s of one host core",
           "host_cores": os.cpu_count()}
    if alg == 1:
        # informational: a numpy harness with the reference's per-iteration operation mix (full-capacity array passes, argsort,
        # Python near-set loop; oracle/numpy_like.py), first iterations of the same query, bounded to ~8 s
        from oracle import numpy_like

        _, _, _, jl, it, dl = numpy_like.rrtstar_like(og8, n, xs, xg, samples, cfg["r_rewire"], time_limit=8.0)
        out["reference_like"] = {"value": (jl - 1) / dl, "unit": "nodes/s", "cores": 1,
                                 "sample": f"first {it} iterations of the same query at capacity n={n}, {dl:.1f} s of one host core, numpy"}
    return out


if __name__ == "__main__":
    main()
