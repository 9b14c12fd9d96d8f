"""CPU, world_size 2: the N > 1 path of bench.py -- round-robin sharding of independent queries, the hand-over of the
communicator id between the ranks, one all-gather of fixed-size self-describing result slabs, max-over-ranks timing.
Here gloo's all_gather stands in for the ncclAllGather of rrt_gather (C ABI, needs GPUs) and the CPU oracle for the
expansion (tests only); the shard map, the slab layout and the id hand-over are the product's (rrtplanner_amd/multi.py)."""
import os
import socket

import numpy as np
import pytest

import oracle
from rrtplanner_amd import hostprep, multi
from rrtplanner_amd.oggen import perlin_occupancygrid, random_connected_pair


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, n, out_dir):
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    og = perlin_occupancygrid(96, 96, seed=2)
    og8 = oracle.og_u8(og)
    free = np.argwhere(og == 0)
    sg = np.random.default_rng(7)
    pairs = [random_connected_pair(og, sg) for _ in range(total)]
    mine = multi.shard_queries(total, world, rank)
    stride = ((n + 1 + 4095) // 4096) * 4096
    Q = len(mine)
    vcost = np.zeros((Q, stride)); nodes = np.zeros((Q, stride), dtype=np.uint32); parent = np.zeros((Q, stride), dtype=np.int32)
    meta = np.zeros((Q, 4), dtype=np.int32)
    for slot, g in enumerate(mine):
        xs, xg = pairs[g]
        s = hostprep.draw_free_samples(np.random.default_rng(g), free, n)
        st, r = oracle.plan(og8, n, 1, xs, xg, s, r2_rewire=hostprep.radius_threshold(16), logs=False)
        live = r.j + (1 if r.found else 0)
        vcost[slot, :live] = r.vcost[:live]
        nodes[slot, :live] = r.pts[:live, 0].astype(np.uint32) | (r.pts[:live, 1].astype(np.uint32) << 16)
        parent[slot, :live] = r.parent[:live]
        meta[slot] = (st, r.j, r.vgoal, r.found)
    slab = np.concatenate([vcost.view(np.uint8).ravel(), nodes.view(np.uint8).ravel(), parent.view(np.uint8).ravel(),
                           meta.view(np.uint8).ravel()])
    assert slab.size == multi.slab_bytes(Q, stride)
    # the id hand-over of multi.init_comm (the id itself is only meaningful to RCCL)
    uid = multi.exchange_unique_id(rank, world, lambda: bytes(range(128)), path=os.path.join(out_dir, "comm.id"))
    assert uid == bytes(range(128))
    local = torch.from_numpy(slab)
    blocks = torch.empty((world, local.numel()), dtype=torch.uint8)
    dist.all_gather_into_tensor(blocks.view(-1), local)
    t = torch.tensor([0.1 * (rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        np.save(os.path.join(out_dir, "blocks.npy"), blocks.numpy())
        np.save(os.path.join(out_dir, "tmax.npy"), t.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_and_gather(tmp_path):
    import torch.multiprocessing as mp

    world, total, n = 2, 6, 150
    mp.spawn(_worker, args=(world, _free_port(), total, n, str(tmp_path)), nprocs=world, join=True)
    blocks = np.load(tmp_path / "blocks.npy")
    assert np.load(tmp_path / "tmax.npy")[0] == pytest.approx(0.2)
    stride = ((n + 1 + 4095) // 4096) * 4096
    Q = total // world
    assert blocks.shape == (world, multi.slab_bytes(Q, stride))
    og = perlin_occupancygrid(96, 96, seed=2)
    og8 = oracle.og_u8(og)
    free = np.argwhere(og == 0)
    sg = np.random.default_rng(7)
    pairs = [random_connected_pair(og, sg) for _ in range(total)]
    for g in range(total):
        v, nd, pa, meta = multi.unpack_slab(blocks[multi.owner_of(g, world)], Q, stride)
        slot = multi.local_slot(g, world)
        xs, xg = pairs[g]
        s = hostprep.draw_free_samples(np.random.default_rng(g), free, n)
        st, r = oracle.plan(og8, n, 1, xs, xg, s, r2_rewire=hostprep.radius_threshold(16), logs=False)
        live = r.j + (1 if r.found else 0)
        assert meta[slot].tolist() == [st, r.j, r.vgoal, r.found]
        assert np.array_equal(nd[slot, :live] & 0xffff, r.pts[:live, 0]) and np.array_equal(nd[slot, :live] >> 16, r.pts[:live, 1])
        assert np.array_equal(pa[slot, :live], r.parent[:live]) and np.array_equal(v[slot, :live], r.vcost[:live])


def _id_worker(rank, world, path, q):
    q.put((rank, multi.exchange_unique_id(rank, world, lambda: os.urandom(128), path=path, timeout=30.0)))


def test_unique_id_hand_over_between_three_ranks(tmp_path):
    """Rank 0 publishes, the others poll (started first, so they really wait); all see the same 128 bytes; the launcher
    key is the same in every child of one parent."""
    import multiprocessing as mp

    ctx = mp.get_context("fork")
    q = ctx.Queue()
    path = str(tmp_path / "x.id")
    ps = [ctx.Process(target=_id_worker, args=(r, 3, path, q)) for r in (2, 1, 0)]
    for p in ps:
        p.start()
    got = dict(q.get(timeout=60) for _ in ps)
    for p in ps:
        p.join(timeout=60)
    assert len(got[0]) == 128 and got[0] == got[1] == got[2]
    multi.release_unique_id(0, path=path)
    assert not os.path.exists(path)
    assert multi._launcher_key().split("_")[0] == str(os.getppid())
    with pytest.raises(TimeoutError):
        multi.exchange_unique_id(1, 2, None, path=str(tmp_path / "never.id"), timeout=0.2)


def test_unique_id_port_keyed_fallback(tmp_path, monkeypatch):
    """Ranks that do not share a launcher process still meet through the file named after the rendezvous port; a stale one is
    ignored."""
    import time

    monkeypatch.setenv("RRT_COMM_DIR", str(tmp_path))
    monkeypatch.setenv("MASTER_PORT", "45678")
    uid = multi.exchange_unique_id(0, 2, lambda: bytes(range(128)))
    main, fallback = multi._id_paths(None)
    assert os.path.exists(main) and os.path.exists(fallback)
    monkeypatch.setattr(multi, "_launcher_key", lambda: "another_parent")  # rank 1 was started by someone else
    assert multi.exchange_unique_id(1, 2, None, timeout=20.0) == uid
    old = time.time() - 3600
    os.utime(fallback, (old, old))  # a file a crashed run left behind an hour ago
    with pytest.raises(TimeoutError):
        multi.exchange_unique_id(1, 2, None, timeout=4.0)
    monkeypatch.undo()
    monkeypatch.setenv("RRT_COMM_DIR", str(tmp_path))
    monkeypatch.setenv("MASTER_PORT", "45678")
    multi.release_unique_id(0)
    assert not os.path.exists(main) and not os.path.exists(fallback)


def test_stale_main_file_is_ignored_and_replaced(tmp_path, monkeypatch):
    """a file an earlier attempt left under the SAME launcher-keyed name (rank 0 died inside ncclCommInitRank) must
    not be handed to the other ranks: an old one is refused by its age, and rank 0 removes it before it makes the new id."""
    import time

    monkeypatch.setenv("RRT_COMM_DIR", str(tmp_path))
    monkeypatch.setenv("MASTER_PORT", "45679")
    main, fallback = multi._id_paths(None)
    with open(main, "wb") as f:
        f.write(b"\xee" * 128)
    old = time.time() - 3600
    os.utime(main, (old, old))
    with pytest.raises(TimeoutError):
        multi.exchange_unique_id(1, 2, None, timeout=0.5)
    seen = {}

    def make():
        seen["stale_gone"] = not os.path.exists(main) and not os.path.exists(fallback)  # nothing readable while the id is being made
        return bytes(range(128))

    with open(main, "wb") as f:  # a FRESH leftover (an attempt that died seconds ago): only the unlink-before-publish protects
        f.write(b"\xee" * 128)
    uid = multi.exchange_unique_id(0, 2, make)
    assert seen["stale_gone"] and uid == bytes(range(128))
    assert multi.exchange_unique_id(1, 2, None, timeout=5.0) == uid
    assert (os.stat(main).st_mode & 0o777) == 0o600
    multi.release_unique_id(0)


def test_attempt_nonce_names_the_file(monkeypatch):
    """Worker restarts under one torchrun agent (same parent, same port) get different file names."""
    monkeypatch.setenv("MASTER_PORT", "45680")
    monkeypatch.setenv("TORCHELASTIC_RUN_ID", "run/1")
    monkeypatch.setenv("TORCHELASTIC_RESTART_COUNT", "0")
    k0 = multi._launcher_key()
    monkeypatch.setenv("TORCHELASTIC_RESTART_COUNT", "1")
    k1 = multi._launcher_key()
    monkeypatch.setenv("RRT_COMM_NONCE", "abc")
    k2 = multi._launcher_key()
    assert len({k0, k1, k2}) == 3 and "/" not in k0 + k1 + k2


def test_planted_symlink_is_not_followed(tmp_path):
    """The id file lives in a world-writable directory: a link planted under its name is neither read nor written through."""
    victim = tmp_path / "victim"
    victim.write_bytes(b"\x01" * 128)
    link = tmp_path / "planted.id"
    os.symlink(victim, link)
    with pytest.raises(TimeoutError):
        multi.exchange_unique_id(1, 2, None, path=str(link), timeout=0.3)
    uid = multi.exchange_unique_id(0, 2, lambda: bytes(range(128)), path=str(link))
    assert victim.read_bytes() == b"\x01" * 128 and not os.path.islink(link) and open(link, "rb").read() == uid


def test_bench_gpus_n_starts_its_own_ranks():
    """`python bench.py --gpus 2` called plainly (no launcher, no RANK) must start two rank processes itself.  On a box
    without a GPU each rank fails at its first HIP call: the failure comes from INSIDE the children (rank environment set,
    library loaded), not from argument checking, and the parent reports the ranks' exit codes."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    env["RRT_BENCH_ECHO_RANK"] = "1"
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "0", "--no-cpu-baseline",
                        "--no-batched", "--n", "500"], env=env, capture_output=True, text=True, timeout=600)
    err = p.stderr
    assert "bench rank 0 of 2" in err and "bench rank 1 of 2" in err, err[-2000:]
    try:
        import ctypes

        have_gpu = ctypes.CDLL("libamdhip64.so").hipGetDeviceCount(ctypes.byref(ctypes.c_int(0))) == 0
    except OSError:
        have_gpu = False
    if not have_gpu:
        assert p.returncode != 0 and "rank(s) failed" in err and "librrt_hip error" in err, err[-2000:]
        assert "launch with" not in err
