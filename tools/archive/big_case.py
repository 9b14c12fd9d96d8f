import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, oracle
from rrtplanner_amd import _ffi, hostprep
from rrtplanner_amd.oggen import perlin_occupancygrid, random_connected_pair
og = perlin_occupancygrid(2048, 2048, seed=3); og8 = oracle.og_u8(og)
xs, xg = random_connected_pair(og, np.random.default_rng(11))
free = np.argwhere(og8 == 0)
ctx = _ffi.Context(0); ctx.set_grid(og8)
for n, alg, kw in ((262143, 1, {}), (262143, 0, {}), (200000, 2, {}), (262143, 1, {"team": 1})):
    rng = np.random.default_rng(5)
    samples = hostprep.draw_free_samples(rng, free, n)
    r2 = hostprep.radius_threshold(64) if alg else 0
    gd2 = hostprep.goal_threshold(40) if alg == 2 else 0
    Cm = hostprep.rotation_to_world_frame(xs, xg) if alg == 2 else None
    q, keep = _ffi.make_query(alg, n, xs, xg, samples, r2_rewire=r2, goal_d2=gd2, Cmat=Cm)
    t = time.time(); rc, res = ctx.plan(q, n, **kw); td = time.time() - t
    t = time.time(); st, ro = oracle.plan(og8, n, alg, xs, xg, samples, r2_rewire=r2, r_goal=40 if alg == 2 else 0.0, Cmat=Cm, logs=False); to = time.time() - t
    if rc == _ffi.RRT_NEED_UNITBALL:
        ub = hostprep.draw_unitball(rng, n - res.i_switch)
        rc = ctx.plan_resume(ub, res)
        st, ro = oracle.plan(og8, n, alg, xs, xg, samples, r2_rewire=r2, r_goal=40, unitball=ub, ub_offset=res.i_switch, Cmat=Cm, logs=False)
    live = ro.j + (1 if ro.found else 0)
    ok = rc == st and res.j == ro.j and res.vgoal == ro.vgoal and np.array_equal(res.pts[:live], ro.pts[:live]) and np.array_equal(res.parent[:live], ro.parent[:live]) and np.array_equal(res.vcost[:live], ro.vcost[:live])
    print(f"n={n} alg={alg} {kw}: device {td:.2f}s oracle {to:.1f}s j={ro.j} equal={ok}", flush=True)
# one over the limit must be refused
try:
    _ffi.Batch(ctx, 1, 262144); print("limit NOT enforced")
except _ffi.RRTError as e:
    print("n=262144 refused:", e.code)
