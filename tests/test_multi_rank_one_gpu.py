"""The world > 1 branches of the C ABI's gather (BASELINE.json configs[3]: shards of queries per GPU, one all-gather of the result
slabs) executed for real on a one-GPU box: two rank PROCESSES on device 0, collectives through the stand-in library tests/fake_rccl
(RCCL refuses two ranks on one device).  The reference has no counterpart: rrt.py is one process."""
import os
import subprocess
import sys
import tempfile

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def test_stand_in_library_builds_and_exports_what_the_engine_binds():
    sys.path.insert(0, os.path.join(HERE, "fake_rccl"))
    import build as fake_build

    lib = fake_build.build()
    out = subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True, check=True).stdout
    for sym in ("ncclGetUniqueId", "ncclCommInitRank", "ncclCommDestroy", "ncclAllGather", "ncclAllReduce", "ncclGetErrorString"):
        assert f" T {sym}" in out, sym


def _run_ranks(world, extra_env=None, script="two_ranks_one_gpu.py", args=()):
    sys.path.insert(0, os.path.join(HERE, "fake_rccl"))
    import build as fake_build

    lib = fake_build.build()
    with tempfile.TemporaryDirectory() as d:
        env = dict(os.environ, RRT_RCCL_LIB=lib, HSA_ENABLE_IPC_MODE_LEGACY="0", **(extra_env or {}))
        # fresh child processes, started before either touches the GPU (this pytest process's own GPU state is not inherited:
        # Popen starts a new interpreter)
        procs = [subprocess.Popen([sys.executable, os.path.join(HERE, script), str(r), str(world), os.path.join(d, "comm.id"), *args], env=env,
                                  stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
        outs = []
        for p in procs:
            try:
                outs.append(p.communicate(timeout=420))
            except subprocess.TimeoutExpired:
                for q in procs:
                    q.kill()
                raise
        return [(p.returncode, o, e) for p, (o, e) in zip(procs, outs)]


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_gather_and_fetch_each_others_trees():
    res = _run_ranks(2)
    for r, (rc, out, err) in enumerate(res):
        assert rc == 0 and f"RANK {r} of 2 OK" in out, f"rank {r}: rc={rc}\n{out}\n{err[-3000:]}"


@pytest.mark.gpu
def test_bench_with_two_ranks_on_one_gpu_prints_one_line():
    """`bench.py --gpus 2 --share-gpu` with the stand-in library: both ranks on device 0, the gather between them, ONE JSON line
    from rank 0 with n_gpus = 2 (a rehearsal of the N > 1 path, marked as such in the line; not a scaling measurement)."""
    import json

    sys.path.insert(0, os.path.join(HERE, "fake_rccl"))
    import build as fake_build

    env = dict(os.environ, RRT_RCCL_LIB=fake_build.build(), HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--steps", "2", "--warmup", "1", "--team", "16",
                        "--no-cpu-baseline", "--no-batched"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["collective"].startswith("rrt_gather") and d["config"]["rehearsal_shared_gpu"] is True
    assert d["config"]["unfinished_queries"] == 0 and d["value"] > 0
