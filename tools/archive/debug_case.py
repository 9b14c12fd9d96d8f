#!/usr/bin/env python3
"""Debug helper: run one golden case on the device (block or serial kernel) and diff it against the oracle."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle, orchelp
from rrtplanner_amd import _ffi, hostprep
cid = sys.argv[1]; serial = len(sys.argv) > 2 and sys.argv[2] == "serial"
G = orchelp.golden("plans_A.npz"); m = G.by_id[cid]
og8 = G.grid(m["grid"]); n, alg = m["n"], m["alg"]
rng = np.random.default_rng(m["seed"]); free = np.argwhere(og8 == 0)
samples = hostprep.draw_free_samples(rng, free, n)
r2 = hostprep.radius_threshold(m["r_rewire"]) if m["r_rewire"] is not None else 0
gd2 = hostprep.goal_threshold(m["r_goal"]) if m["r_goal"] is not None else 0
xs, xg = np.array(m["xstart"]), np.array(m["xgoal"])
Cm = hostprep.rotation_to_world_frame(xs, xg) if alg == 2 else None
ctx = _ffi.Context(0); ctx.set_grid(og8)
q, keep = _ffi.make_query(alg, n, xs, xg, samples, r2_rewire=r2, goal_d2=gd2, Cmat=Cm)
rc, res = ctx.plan(q, n, logs=True, serial=serial)
st, ro = oracle.plan(og8, n, alg, xs, xg, samples, r2_rewire=r2, r_goal=m["r_goal"] or 0.0, Cmat=Cm)
if rc == 1:
    ub = hostprep.draw_unitball(rng, n - res.i_switch); rc = ctx.plan_resume(ub, res)
    st, ro = oracle.plan(og8, n, alg, xs, xg, samples, r2_rewire=r2, r_goal=m["r_goal"], unitball=ub, ub_offset=res.i_switch, Cmat=Cm)
print("rc", rc, st, "j", res.j, ro.j, "vgoal", res.vgoal, ro.vgoal, "i_switch", res.i_switch, ro.i_switch)
for name, a, b in (("nearest", res.nearest_log, ro.nearest_log), ("accept", res.accept_log, ro.accept_log), ("jlog", res.j_log, ro.jlog)):
    d = np.flatnonzero(a != b)
    print(name, "first diff at iter", d[:5], "dev", a[d[:5]], "orc", b[d[:5]])
live = min(res.j, ro.j)
for name, a, b in (("parent", res.parent[:live], ro.parent[:live]), ("vcost", res.vcost[:live], ro.vcost[:live]), ("ptsx", res.pts[:live,0], ro.pts[:live,0])):
    d = np.flatnonzero(a != b)
    print(name, "first diff at node", d[:5], "dev", a[d[:5]], "orc", b[d[:5]])
d = np.flatnonzero(res.parent[:live] != ro.parent[:live])
if d.size:
    v = d[0]; it = np.flatnonzero(ro.jlog == v)
    print("node", v, "inserted at iter", it[-1] if it.size else None, "block pos", (it[-1] % 16) if it.size else None, "pt", ro.pts[v], "orc parent", ro.parent[v], ro.pts[ro.parent[v]], "dev parent", res.parent[v], res.pts[res.parent[v]])
    print("costs: orc", ro.vcost[v], "dev", res.vcost[v])
print("stats dev", res.sum_j, res.sum_cells_nn, res.sum_near, "orc", ro.sum_j, ro.sum_cells_nn, ro.sum_near)
if os.environ.get("RRT_HIP_LIB", "").endswith("dbg.so"):
    # per-iteration snapshot+block near counts as the device saw them vs numpy
    pts = ro.pts.astype(np.int64)
    for it in range(n):
        jj = ro.jlog[it]
        d2 = ((pts[:jj] - samples[it]) ** 2).sum(1)
        want = int((d2 < r2).sum())
        got = res.cbest_log[it]
        if ro.accept_log[it] and int(got) != want:
            miss = np.flatnonzero(d2 < r2)
            print("iter", it, "blockpos", it % 16, "j", jj, "j0", ro.jlog[it - it % 16], "want", want, "got", got, "nn", ro.nearest_log[it], "near idx", miss[:12])
            nbad = globals().get("nbad", 0) + 1
            if nbad > 25: break
