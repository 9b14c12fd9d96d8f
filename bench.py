#!/usr/bin/env python3
"""Benchmark of the tree-expansion hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--queries Q] [--config 2|3|4]

A "step" is one pass of the hot path over one batch of queries that is already resident in HBM
(grid, packed sample streams): re-arm the trees, run the expansion kernel to completion
(sample -> nearest -> line of sight -> choose parent -> insert for all n iterations, then
go2goal), and for N > 1 all-gather the result slabs over RCCL (rrt_gather, C ABI).  Default
workload = BASELINE.json configs[1]: RRT*, 1024x1024 noise grid, n = 50000, r_rewire = 64, one
query per GPU.  Metric: nodes expanded per second (inserted tree nodes / wall time), whole job
over all ranks.

For N > 1 there is one rank per GPU.  Either a launcher starts them (`python -m torch.distributed.run
--nproc-per-node N bench.py --gpus N ...`: torch is only the process launcher that sets RANK /
LOCAL_RANK / WORLD_SIZE / MASTER_*), or `python bench.py --gpus N` is called plainly and this file
starts its own N rank processes (spawn_ranks, below) BEFORE anything touches the GPU or loads
librrt_hip.so -- the parent never creates a GPU context, it relays rank 0's JSON line and the
children's exit codes.  Queries are sharded query -> rank with no data-path collective (weak
scaling: per-GPU work fixed); the communicator and every collective (barrier, max-over-ranks,
gather) go through librrt_hip.so.  No torch anywhere in this file.
"""
import argparse
import json
import os
import socket
import subprocess
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
    # BASELINE.json configs[4]: no reference code exists for it (README only) -> own semantics (include/rrt_dubins.h), own oracle
    5: dict(name="Dubins-RRT* 2048x2048 n=100000 r_rewire=64 rho=8 64 headings (256 independent queries per GPU, one CU each)", alg=4, grid=2048,
            n=100000, r_rewire=64, r_goal=None, queries=256, rho=8.0, nh=64, grid_seed=3),
}

DUBINS_F64_OPS = 955  # f64 VALU instructions of one dub_shortest() in the gfx950 ISA (static count over all six words, hipcc -O3; 2 043 instructions in all, 12 divisions)
F64_VALU_PEAK_TFLOPS = 78.6  # MI355X vector f64 (MI355X_MICROARCH.md)


def algorithmic_bytes(res):
    """per iteration 8*j_i node-coordinate bytes (fused NN + radius pass) + 1 B per
    line-of-sight cell nearest->new; per accepted iteration 8 B vcost per near-set entry + 1 B per
    choose-parent line-of-sight cell + 20 B insert."""
    return 8 * res.sum_j + res.sum_cells_nn + 8 * res.sum_near + res.sum_cells_cand + 20 * (res.j - 1)


def lib_build_id():
    """sha256 (first 16 hex digits) of the HIP library this process loads: every bench line and every profile taken through
    tools/gpu_round.sh carries it, so that a committed counter pass can be tied to the binary it measured."""
    import hashlib

    from rrtplanner_amd import _ffi

    try:
        with open(_ffi.LIB_PATH, "rb") as f:
            return hashlib.sha256(f.read()).hexdigest()[:16]
    except OSError:
        return None


def committed_json(name):
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


def measured_traffic(config, Q, n, team, pipelined):
    """HBM bytes per launch of rrt_expand_block_kernel from the committed rocprofv3 PMC passes (profiles/*_traffic.json), only
    when a pass was taken on this exact workload AND kernel variant (team size, pipeline); None otherwise (the counters cannot
    be read from inside the bench)."""
    for name in ("r04_traffic.json", "r03_traffic.json", "r02_traffic.json"):
        t = committed_json(name)
        for e in (t or {}).get("entries", []):
            if (e.get("config"), e.get("queries_per_gpu"), e.get("n"), e.get("team"), e.get("pipelined")) == (config, Q, n, team, bool(pipelined)):
                return e["hbm_bytes_per_launch"], "profiles/" + name, e.get("build_sha256")
    return None, None, None


def roofline_block(config, Q, n, team, pipelined, kernel, kernel_ms, model_bytes, fallbacks):
    """The roofline object of one launch.
    achieved / frac: SURVEY.md 8(d) ALGORITHMIC bytes of the launch / its HIP-event duration, against the HBM peak.  The model
        bytes describe the reference's brute force (every live node read for every sample); the kernel serves them from LDS / L2
        or never reads them (nearest from the cell records), so when the model rate exceeds the peak it is no bound on this
        launch: `frac` is then null and `model_exceeds_peak` true.
    measured: what HBM really moved per launch, from the committed rocprofv3 PMC passes ((2 FETCH_SIZE + WRITE_SIZE) KiB, separate
        passes, profiles/*_traffic.json) -- taken on this exact workload and kernel variant in an EARLIER run (`source`), divided
        by THIS run's kernel time; null when no pass matches or the launch fell back to one CU per query.
    bound_observed: what the SQ counters of the same kernel say it waits on (profiles/r03_sq_counters.json)."""
    ach = model_bytes / (kernel_ms * 1e-3) / 1e9
    over = ach > HBM_PEAK_GBS
    traffic, src, tbuild = (None, None, None) if fallbacks else measured_traffic(config, Q, n, team, pipelined)
    build = lib_build_id()
    if traffic is not None and tbuild != build:
        # the counter pass measured another binary (or one of unknown identity): not this kernel's traffic
        src = f"{src} was taken on build {tbuild}, this run is build {build}: not reported"
        traffic = None
    elif src:
        src += f" (an earlier run of this workload and kernel variant on the same build {build}, not this run)"
    r = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None if over else ach / HBM_PEAK_GBS,
         "model_exceeds_peak": over, "traffic": traffic, "traffic_source": src, "traffic_build": tbuild,
         "kernel": kernel, "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": int(model_bytes),
         "achieved_is": "SURVEY 8(d) algorithmic (model) bytes per second, not measured HBM traffic: see `measured`",
         "measured": None, "bound_observed": None}
    if traffic is not None:
        gbs = traffic / (kernel_ms * 1e-3) / 1e9
        r["measured"] = {"hbm_bytes": int(traffic), "hbm_gbs": gbs, "hbm_frac": gbs / HBM_PEAK_GBS}
    sq = committed_json("r04_sq_counters.json") or committed_json("r03_sq_counters.json") or committed_json("r02_sq_counters.json")
    for e in (sq or {}).get("entries", []):
        if (e.get("config"), e.get("queries_per_gpu")) == (config, Q) and e.get("build_sha256") == build:
            r["bound_observed"] = {"kind": e.get("kind", "latency"), "waves_waiting_frac": e.get("waves_waiting_frac"),
                                   "waves_issuing_frac": e.get("waves_issuing_frac"), "valu_busy_frac": e.get("valu_busy_frac"),
                                   "source": e.get("source")}
    return r


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(n_ranks):
    """`python bench.py --gpus N` without a launcher: start N copies of this command as ranks 0 .. N-1 (one per GPU) and
    wait for them.  Runs before this process has imported the HIP binding or made any GPU call, and it never does: the
    children are ordinary child processes (no exec of a GPU-initialised process).  Rank 0's stdout (the one JSON line) is
    relayed; every other stream goes to stderr.  Exit code: 0 only if every rank exits 0; when one rank fails the others are
    given 20 s (they may sit in a collective that can no longer complete) and are then terminated by pid."""
    env0 = dict(os.environ)
    env0.update(WORLD_SIZE=str(n_ranks), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
                RRT_COMM_NONCE=f"{os.getpid()}_{time.time_ns()}")
    env0.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = []
    for r in range(n_ranks):
        env = dict(env0, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr))
    deadline = None
    codes = [None] * n_ranks
    out0 = b""
    while any(c is None for c in codes):
        for r, pr in enumerate(procs):
            if codes[r] is None:
                if r == 0:
                    try:
                        o, _ = pr.communicate(timeout=0.2)
                        out0 += o or b""
                    except subprocess.TimeoutExpired:
                        continue
                    codes[r] = pr.returncode
                else:
                    codes[r] = pr.poll()
        if deadline is None and any(c not in (None, 0) for c in codes):
            deadline = time.monotonic() + 20.0
        if deadline is not None and time.monotonic() > deadline:
            for r, pr in enumerate(procs):
                if codes[r] is None:
                    pr.terminate()
            deadline = time.monotonic() + 1e9
        time.sleep(0.05)
    sys.stdout.write(out0.decode(errors="replace"))
    sys.stdout.flush()
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if bad:
        sys.stderr.write("bench: rank(s) failed: " + ", ".join(f"rank {r} exit {c}" for r, c in bad) + "\n")
        return 1
    return 0


def check_against_oracle(tag, r, ro, st):
    """Outside the timed region: the kernel's result scalars and byte-model statistics must be the oracle's.  (Not compared:
    sum_cells_cand -- the device tests choose-parent candidates cheapest first and stops at the first visible one, the
    reference walks them in index order (rrt.py:515-521), so the device reads fewer line-of-sight cells for the same answer;
    the byte model uses the device's own, smaller count.)"""
    got = (r.c.status, r.c.j, r.c.vgoal, r.c.found, r.c.sum_j, r.c.sum_cells_nn, r.c.sum_near)
    want = (st, ro.j, ro.vgoal, ro.found, ro.sum_j, ro.sum_cells_nn, ro.sum_near)
    if r.c.sum_cells_cand > ro.sum_cells_cand:
        raise SystemExit(f"bench: {tag}: the device read more choose-parent cells ({r.c.sum_cells_cand}) than the reference's walk ({ro.sum_cells_cand})")
    if got != want:
        raise SystemExit(f"bench: {tag}: device result {got} differs from the CPU oracle {want}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS), help="2-4: BASELINE.json configs[1..3]; 5: configs[4] (Dubins-RRT*, no reference parity)")
    ap.add_argument("--queries", type=int, default=None, help="queries per GPU (default: the config's)")
    ap.add_argument("--total-queries", type=int, default=None,
                    help="the whole job (BASELINE.json configs[3]: 512 queries) divided over the ranks, total / N per GPU: strong scaling")
    ap.add_argument("--no-plan-wall", action="store_true", help="skip the plan() wall-time leg (config 2, one GPU)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal of the N > 1 path on a one-GPU box: every rank on device 0 (needs RRT_RCCL_LIB = a stand-in collective "
                         "library such as tests/fake_rccl, RCCL refuses two ranks on one device); the line says so, it is no scaling measurement")
    ap.add_argument("--n", type=int, default=None)
    ap.add_argument("--team", type=int, default=None, choices=[1, 2, 3, 4, 8, 16, 32, 64],
                    help="cap on the CUs per query (default: the largest team for which all teams are resident together)")
    ap.add_argument("--serial", action="store_true", help="the one-sample-per-iteration kernel (RRT_FLAG_SERIAL), for comparison")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-batched", action="store_true", help="skip the informational batched leg (configs[3] share of one GPU)")
    ap.add_argument("--rewire-leg", action="store_true", help="add an informational leg: query 0 with the opt-in true rewire (not the reference's behaviour)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # called plainly: be the launcher (nothing GPU-related has been imported or called in this process)
        raise SystemExit(spawn_ranks(args.gpus))

    # rank 0 prints exactly ONE line on stdout: route everything else that writes to fd 1 (RCCL's banner, library
    # chatter) to stderr until the JSON line is ready
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args.gpus = world  # the launcher's world size is authoritative
    use_comm = world > 1 or "RANK" in os.environ  # launched as a rank (also with one rank: exercises the gather)
    if os.environ.get("RRT_BENCH_ECHO_RANK"):
        print(f"bench rank {rank} of {world} (local rank {local_rank}, pid {os.getpid()})", file=sys.stderr, flush=True)

    from rrtplanner_amd import _ffi, hostprep, multi
    from rrtplanner_amd.oggen import perlin_occupancygrid, random_connected_pairs

    cfg = dict(CONFIGS[args.config])
    if args.queries:
        cfg["queries"] = args.queries
    if args.total_queries:
        if args.total_queries % world:
            raise SystemExit(f"--total-queries {args.total_queries} does not divide over {world} ranks")
        cfg["queries"] = args.total_queries // world
    if args.n:
        cfg["n"] = args.n
    Q, n, alg = cfg["queries"], cfg["n"], cfg["alg"]

    # ---- synthetic workload: seeded noise grid, start/goal in one free component ----
    og = perlin_occupancygrid(cfg["grid"], cfg["grid"], thresh=0.33, seed=cfg.get("grid_seed", 1))
    og8 = hostprep.og_nonzero(og)
    free = np.argwhere(og == 0)
    sg_rng = np.random.default_rng(7)
    pairs = random_connected_pairs(og, sg_rng, Q * world)  # query g = rank + world*slot
    r2 = hostprep.radius_threshold(cfg["r_rewire"])
    gd2 = hostprep.goal_threshold(cfg["r_goal"]) if cfg["r_goal"] is not None else 0

    if args.share_gpu and not os.environ.get("RRT_RCCL_LIB"):
        raise SystemExit("--share-gpu needs RRT_RCCL_LIB (a stand-in collective library, e.g. tests/fake_rccl/libfake_rccl.so)")
    ctx = _ffi.Context(0 if args.share_gpu else local_rank)
    if use_comm:
        multi.init_comm(ctx, rank, world)  # RCCL communicator (ncclCommInitRank), id handed over on local tmpfs
    ctx.set_grid(og8)
    dubins = alg >= _ffi.ALG_DUBINS
    batch = _ffi.Batch(ctx, Q, n, team=args.team, dubins=dubins, serial=args.serial)
    keep, rngs, states, dub_inputs = [], [], [], []
    for slot in range(Q):
        g = rank + world * slot
        xs, xg = pairs[g]
        rng = np.random.default_rng(g)  # planner seed = global query index
        states.append(rng.bit_generator.state)
        samples = hostprep.draw_free_samples(rng, free, n)
        if dubins:  # poses: start / goal headings from the query index, sample headings from the planner's stream
            heads = rng.integers(0, cfg["nh"], size=n)
            ps, pg = (int(xs[0]), int(xs[1]), (7 * g) % cfg["nh"]), (int(xg[0]), int(xg[1]), (13 * g + 5) % cfg["nh"])
            qu, k = _ffi.make_query(alg, n, ps, pg, samples, r2_rewire=r2, headings=heads, rho=cfg["rho"], nh=cfg["nh"])
            if slot == 0:
                dub_inputs = [ps, pg, samples, heads]
        else:
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
        if use_comm:
            batch.gather()  # ncclAllGather of the result slabs, on the context's stream
            ctx.sync()      # the next step rewrites the slab the collective is reading
        return ms

    def sync_all():
        ctx.sync()  # stream + device
        if use_comm:
            ctx.barrier()
            ctx.sync()

    for _ in range(args.warmup):
        one_step()
    sync_all()
    t0 = time.perf_counter()
    kern_ms = 0.0
    for _ in range(args.steps):
        kern_ms += one_step()
    sync_all()
    dt = time.perf_counter() - t0
    if use_comm:
        dt = float(ctx.allreduce([dt], "max")[0])

    results = [batch.get_result(slot, arrays=False) for slot in range(Q)]
    nodes_local = sum(r.c.j - 1 for r in results)
    iters_local = Q * n
    bytes_local = sum(algorithmic_bytes(r.c) for r in results)
    bad = [r.c.status for r in results if r.c.status not in (0, _ffi.RRT_E_GOAL_UNREACHABLE if dubins else 0)]
    if use_comm:
        # sanity of the collective: every rank's slab must describe its own queries; this rank's own slab must come back unchanged.
        # A failure is counted and summed over the ranks, so that every rank leaves the collectives together before anyone exits.
        gather_bad = 0
        own = batch.get_result(0)
        back = batch.gather_fetch(rank, 0)
        live = own.j + (1 if own.found else 0)
        if (back.j, back.vgoal) != (own.j, own.vgoal) or not (np.array_equal(back.pts[:live], own.pts[:live]) and
                                                              np.array_equal(back.parent[:live], own.parent[:live]) and
                                                              np.array_equal(back.vcost[:live], own.vcost[:live])):
            gather_bad += 1
        far = batch.gather_fetch((rank + 1) % world, Q - 1)
        if not (1 <= far.j <= n):
            gather_bad += 1
        agg = ctx.allreduce([nodes_local, iters_local, len(bad), gather_bad], "sum")
        nodes_total, iters_total, nbad = float(agg[0]), float(agg[1]), int(agg[2])
        if int(agg[3]) != 0:
            ctx.barrier()
            raise SystemExit(f"result gather returned a wrong slab on {int(agg[3])} check(s) (this rank: {gather_bad})")
    else:
        nodes_total, iters_total, nbad = float(nodes_local), float(iters_local), len(bad)

    if rank == 0:
        kern_avg_ms = kern_ms / args.steps
        team, fallbacks = batch.team()
        pipelined = batch.pipelined()
        ms_per_step = dt / args.steps * 1e3
        out = {
            "metric": "RRT* nodes-expanded/s on 1024x1024 Perlin grid; achieved HBM GB/s",
            "value": nodes_total * args.steps / dt,
            "unit": "nodes/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            # wall time of a step that is not the expansion kernel: re-arm copy, init kernel, team memset, launch calls, the
            # descriptor read-back and the host's wait (for N > 1 also the gather and its sync)
            "host_gap_ms": ms_per_step - kern_avg_ms,
            "higher_is_better": True,
            "scaling": "strong" if args.total_queries else "weak",
            "vs_baseline": None,
            "dtype": "int16x2 coordinates / u32 squared distances / f64 costs",
            "data": "synthetic",
            "build": {"lib_sha256": lib_build_id(), "lib": os.path.basename(_ffi.LIB_PATH)},
            "config": {"workload": f"BASELINE.json configs[{args.config - 1}]: {cfg['name']}" + (f" -- the whole job of {args.total_queries} queries over {world} GPU(s)" if args.total_queries else ""),
                       "queries_per_gpu": Q, "total_queries": Q * world, "n": n,
                       "grid": [cfg["grid"], cfg["grid"]], "free_fraction": float((og == 0).mean()),
                       "iters_per_s": iters_total * args.steps / dt, "unfinished_queries": nbad,
                       "cus_per_query": team + (1 if pipelined else 0), "pipelined": pipelined,
                       "team_fallbacks": fallbacks, "collective": "rrt_gather (ncclAllGather, C ABI)" if use_comm else None,
                       "rehearsal_shared_gpu": bool(args.share_gpu), "collective_library": os.environ.get("RRT_RCCL_LIB") or "librccl.so.1"},
            "roofline": roofline_block(args.config, Q, n, team, pipelined, batch.kernel_name(), kern_avg_ms, bytes_local, fallbacks),
        }
        if dubins:
            # no reference parity for this workload (the reference has no Dubins code); bound by f64 VALU work (fixed-order
            # sin / atan2 polynomials of include/rrt_dubins.h), not by HBM
            out["metric"] = "Dubins-RRT* nodes-expanded/s on 2048x2048 noise grid; achieved HBM GB/s (model)"
            out["dtype"] = "int16x2 coordinates / u8 headings / f64 Dubins arc lengths and costs"
            out["config"]["reference_parity"] = "none: the reference only advertises Dubins planners (README.md:12,18-19)"
            # word evaluations of the MODEL (what a brute-force pricing of every near-set entry makes): one per near-set entry and
            # one for the nearest node of every iteration; the kernel screens entries by the chord lower bound, so it makes
            # fewer (`dubins_word_evaluations_made`, counted on the device)
            ndub = sum(r.c.sum_near for r in results) + iters_local
            made = sum(r.c.n_words for r in results)  # word evaluations the device really made (rrt_result.n_words)
            fl = made * DUBINS_F64_OPS / (kern_avg_ms * 1e-3) / 1e12
            out["roofline"]["valu_f64_model"] = {"bound": "valu-f64", "achieved": fl, "peak": F64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                                                 "model_frac": fl / F64_VALU_PEAK_TFLOPS,
                                                 "dubins_word_evaluations_model": int(ndub), "dubins_word_evaluations_made": int(made),
                                                 "f64_ops_per_evaluation": DUBINS_F64_OPS,
                                                 "note": "f64 instructions per evaluation: static count in the compiled ISA of dub_shortest (all branches), not a counter pass; a lane-level figure -- most lanes of a word pass are idle, the SIMD is busy: see bound_observed"}
        if world == 1 and not args.no_cpu_baseline:
            if dubins:
                out["cpu_baseline"] = cpu_baseline_dubins(og8, cfg, dub_inputs, results[0])
            else:
                out["cpu_baseline"] = cpu_baseline(og8, cfg, pairs[0], free, states[0], ub_cache.get(0), results[0])
        if world == 1 and args.config == 2 and not args.no_batched:
            out["batched"] = batched_leg(ctx, og, og8, free, _ffi, hostprep)
        if world == 1 and args.config == 2 and not args.no_plan_wall and not args.queries and not args.n and not args.team and not args.serial:
            out["plan_wall"] = plan_wall_leg(og, cfg, pairs[0])
        if world == 1 and args.rewire_leg and alg >= 1:
            out["rewire_correct"] = rewire_leg(ctx, og8, cfg, pairs[0], free, states[0], _ffi, hostprep)
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if use_comm:
        ctx.barrier()
    batch.close()
    ctx.close()


def plan_wall_leg(og, cfg, pair):
    """Informational: the reference's own unit of work (SURVEY.md 8(d) metric: n / wall time of plan(), rrt.py:466-556) through the
    planner class -- sample draw on the host, upload, expansion kernel, read-back, the lazy graph object -- and what the reference's
    consumers then do with the graph (plots.py:27-29 walks every edge): the first and a second walk over `T.edges(data=True)`."""
    from rrtplanner_amd import RRTStar

    xs, xg = pair
    p = RRTStar(og, cfg["n"], cfg["r_rewire"], pbar=False, seed=0)
    p.plan(xs, xg)  # context, grid upload, first launch
    ts, T, nodes = [], None, 0
    for _ in range(7):
        t0 = time.perf_counter()
        T, gv = p.plan(xs, xg)
        ts.append((time.perf_counter() - t0) * 1e3)
        nodes = p.last_stats["j"] - 1
    ts.sort()
    t0 = time.perf_counter()
    k = 0
    for u, v, d in T.edges(data=True):
        k += 1
    t1 = time.perf_counter()
    for u, v, d in T.edges(data=True):
        k += 1
    t2 = time.perf_counter()
    still_arrays = T.lazy_points() is not None
    if p._ctx is not None:
        p._ctx.close()
    return {"what": "RRTStar(og, n=50000, r_rewire=64).plan(xstart, xgoal) wall time: sample draw + upload + kernel + read-back + graph object (7 calls after a warm-up, "
                    "each with the generator's next samples)",
            "ms": ts[len(ts) // 2], "ms_min": ts[0], "ms_max": ts[-1], "nodes_per_s": nodes / (ts[len(ts) // 2] * 1e-3),
            "edges_iter_ms": (t1 - t0) * 1e3, "edges_iter_again_ms": (t2 - t1) * 1e3, "edges": k // 2,
            "graph_still_array_backed_after_the_walks": still_arrays}


def batched_leg(ctx, og, og8, free, _ffi, hostprep):
    """Informational: one GPU's share of BASELINE.json configs[3] (64 independent RRT* queries, n = 20000), every query on
    its own team of CUs.  Not the headline value.  Four of the queries are checked against the CPU oracle afterwards."""
    import oracle
    from rrtplanner_amd.oggen import random_connected_pairs

    cfg = CONFIGS[4]
    Q, n = cfg["queries"], cfg["n"]
    r2 = hostprep.radius_threshold(cfg["r_rewire"])
    b = _ffi.Batch(ctx, Q, n)
    sg_pairs = random_connected_pairs(og, np.random.default_rng(7), Q)
    keep, qs = [], []
    for q in range(Q):
        xs, xg = sg_pairs[q]
        s = hostprep.draw_free_samples(np.random.default_rng(q), free, n)
        qu, k = _ffi.make_query(cfg["alg"], n, xs, xg, s, r2_rewire=r2)
        keep.append(k)
        qs.append((xs, xg, s))
        b.set_query(q, qu)
    b.launch(); b.sync()
    steps, t0, kms = 3, time.perf_counter(), 0.0
    for _ in range(steps):
        b.rearm(); b.launch(); b.sync()
        kms += b.elapsed_ms()
    dt = time.perf_counter() - t0
    res = [b.get_result(q, arrays=False) for q in range(Q)]
    for q in (0, 21, 42, 63):
        xs, xg, s = qs[q]
        st, ro = oracle.plan(og8, n, cfg["alg"], xs, xg, s, r2_rewire=r2, logs=False)
        check_against_oracle(f"batched leg, query {q}", res[q], ro, st)
    nodes = sum(r.c.j - 1 for r in res)
    by = sum(algorithmic_bytes(r.c) for r in res)
    cus, fallbacks = b.team()
    pipelined, kname = b.pipelined(), b.kernel_name()
    b.close()
    return {"workload": "BASELINE.json configs[3] share of one GPU: " + cfg["name"], "value": nodes * steps / dt, "unit": "nodes/s",
            "ms_per_step": dt / steps * 1e3, "host_gap_ms": dt / steps * 1e3 - kms / steps, "cus_per_query": cus + (1 if pipelined else 0),
            "team_fallbacks": fallbacks, "oracle_checked_queries": [0, 21, 42, 63],
            "roofline": roofline_block(4, Q, n, cus, pipelined, kname, kms / steps, by, fallbacks)}


def cpu_baseline_dubins(og8, cfg, inputs, dev0):
    """The Dubins oracle (oracle/dubins_oracle.c, 1 thread) on query 0; its result must equal the device's."""
    import oracle
    from rrtplanner_amd import hostprep

    ps, pg, samples, heads = inputs
    n = cfg["n"]
    t0 = time.perf_counter()
    st, r = oracle.dubins_plan(og8, n, 1, ps, pg, samples, heads, r2_rewire=hostprep.radius_threshold(cfg["r_rewire"]), rho=cfg["rho"], nh=cfg["nh"], logs=False)
    dt = time.perf_counter() - t0
    got = (dev0.c.status, dev0.c.j, dev0.c.vgoal, dev0.c.found, dev0.c.sum_j, dev0.c.sum_cells_nn, dev0.c.sum_near)
    want = (st, r.j, r.vgoal, r.found, r.sum_j, r.sum_cells_nn, r.sum_near)
    if got != want:
        raise SystemExit(f"bench: Dubins query 0: device result {got} differs from the CPU oracle {want}")
    return {"value": (r.j - 1) / dt, "unit": "nodes/s", "cores": 1, "kind": "port",
            "sample": f"query 0 of the same workload (n={n}), 1 repetition, {dt:.1f} s of one host core", "host_cores": os.cpu_count(),
            "device_result_equals_oracle": True}


def rewire_leg(ctx, og8, cfg, pair, free, state0, _ffi, hostprep):
    """Informational: query 0 of the workload with the opt-in TRUE rewire (RRT_FLAG_REWIRE; one-sample-per-iteration kernel, one
    CU).  Outside reference parity by definition; checked against the oracle's restatement of the same semantics."""
    import oracle

    n, alg = cfg["n"], 1
    xs, xg = pair
    rng = np.random.default_rng(0)
    rng.bit_generator.state = state0
    samples = hostprep.draw_free_samples(rng, free, n)
    r2 = hostprep.radius_threshold(cfg["r_rewire"])
    b = _ffi.Batch(ctx, 1, n, rewire=True)
    qu, keep = _ffi.make_query(alg, n, xs, xg, samples, r2_rewire=r2)
    b.set_query(0, qu)
    b.launch(); b.sync()
    steps, kms, t0 = 3, 0.0, time.perf_counter()
    for _ in range(steps):
        b.rearm(); b.launch(); b.sync()
        kms += b.elapsed_ms()
    dt = time.perf_counter() - t0
    r = b.get_result(0, arrays=False)
    t1 = time.perf_counter()
    st, ro = oracle.plan(og8, n, alg, xs, xg, samples, r2_rewire=r2, logs=False, rewire=True)
    t_or = time.perf_counter() - t1
    if (r.c.status, r.c.j, r.c.vgoal, r.c.n_rewired, r.c.n_propagated) != (st, ro.j, ro.vgoal, ro.n_rewired, ro.n_propagated):
        raise SystemExit("bench: rewire leg differs from the oracle")
    b.close()
    return {"workload": f"RRT* with rewire=\"correct\" (RRT_FLAG_REWIRE), query 0, n={n}", "value": (r.c.j - 1) * steps / dt, "unit": "nodes/s",
            "kernel_ms": kms / steps, "kernel": "rrt_expand_kernel<true>", "cus_per_query": 1, "n_rewired": int(r.c.n_rewired),
            "n_propagated": int(r.c.n_propagated), "cpu_oracle_nodes_per_s": (ro.j - 1) / t_or, "reference_parity": "none (not the reference's behaviour)"}


def cpu_baseline(og8, cfg, pair, free, state0, ub, dev0):
    """The CPU oracle (oracle/rrt_oracle.c, 1 thread, scalar) on query 0 of the same workload; its result must equal the
    device's (checked here, outside the timed region)."""
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
    check_against_oracle("query 0", dev0, r, st)
    out = {"value": nodes / dt, "unit": "nodes/s", "cores": 1, "kind": "port",
           "sample": f"query 0 of the same workload (n={n}), {reps} repetition(s), {dt:.1f} s of one host core",
           "host_cores": os.cpu_count(), "device_result_equals_oracle": True}
    if alg == 1:
        # informational: a numpy harness with the reference's per-iteration operation mix (full-capacity array passes, argsort,
        # Python near-set loop; oracle/numpy_like.py) on the first ~20 s of query 0 at the config's own capacity.  How close the
        # harness is to the real reference was timed in the build container: tests/golden/reference_like_pin.json.
        from oracle import numpy_like

        nl = n  # the config's own capacity: the reference's array passes run over all n rows whatever the live node count
        _, _, _, jl, it, dl = numpy_like.rrtstar_like(og8, nl, xs, xg, samples[:nl], cfg["r_rewire"], time_limit=20.0)
        pin = None
        try:
            with open(os.path.join(ROOT, "tests", "golden", "reference_like_pin.json")) as f:
                pin = json.load(f)["bench1024_star_n20000"]["ratio_numpy_like_over_reference"]
        except (OSError, KeyError, ValueError):
            pass
        out["reference_like"] = {"value": (jl - 1) / dl, "unit": "nodes/s", "cores": 1,
                                 "sample": f"the first {it} of {nl} iterations of query 0 at the config's capacity n={nl}, {dl:.1f} s of one host core, numpy "
                                           "(the near-set loop grows with the tree, so the first iterations overstate the whole query's rate; the "
                                           "real reference measured 273 nodes/s on this config in the build container, BASELINE.md)",
                                 "harness_seconds_over_reference_seconds": pin}
    return out


if __name__ == "__main__":
    main()
