"""Many independent queries: sharding across the GPUs of one node and the result gather.

Queries are independent (own tree, own RNG stream, shared read-only grid), so the path shards
with no data-path collective: query q runs on rank q % world_size.  The one
exchange step is the all-gather of the fixed-size result slabs at the end of a batch; it lives in
the C ABI (``rrt_comm_init`` / ``rrt_gather``: ncclAllGather on the context's stream, RCCL over
xGMI).  This module holds the host side around it: the shard map, the hand-over of the
communicator id between the ranks of a node, and the slab layout.  No torch: a launcher such as
``python -m torch.distributed.run`` only starts the rank processes and sets RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_PORT.
"""
import os
import time
from typing import List


def shard_queries(total: int, world_size: int, rank: int) -> List[int]:
    """Indices of the queries that `rank` owns: round-robin, q -> q % world_size."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError(f"bad rank {rank} of {world_size}")
    return list(range(rank, total, world_size))


def owner_of(q: int, world_size: int) -> int:
    return q % world_size


def local_slot(q: int, world_size: int) -> int:
    """Position of query q inside its owner's batch."""
    return q // world_size


# ---------------------------------------------------------------------------------- communicator id hand-over
def _attempt_nonce() -> str:
    """What distinguishes this launch ATTEMPT from an earlier one under the same launcher: bench.py's own spawner sets
    RRT_COMM_NONCE per launch; torch.distributed.run sets TORCHELASTIC_RUN_ID and, per worker restart under one agent
    (--max-restarts), TORCHELASTIC_RESTART_COUNT.  Every rank of one attempt sees the same values."""
    e = os.environ
    return "_".join(x for x in (e.get("RRT_COMM_NONCE", ""), e.get("TORCHELASTIC_RUN_ID", ""), e.get("TORCHELASTIC_RESTART_COUNT", "")) if x) or "0"


def _launcher_key() -> str:
    """A name all ranks of one launch attempt on this node agree on and no other attempt shares: the launcher's pid and
    start time (every rank is a child of the same launcher process), the rendezvous port and the attempt nonce."""
    ppid = os.getppid()
    start = "0"
    try:
        with open(f"/proc/{ppid}/stat") as f:
            start = f.read().rsplit(")", 1)[1].split()[19]  # field 22: start time in clock ticks since boot
    except (OSError, IndexError):
        pass
    nonce = "".join(c if c.isalnum() or c in "-_" else "-" for c in _attempt_nonce())[:96]
    return f"{ppid}_{start}_{os.environ.get('MASTER_PORT', '0')}_{nonce}"


def _id_paths(path):
    """(launcher-keyed file, port-keyed fallback file).  All ranks of a torch.distributed.run launch are children of one agent
    process, so the first name is shared and unique; a launcher that starts the ranks from different parents still agrees on
    the rendezvous port, which names the fallback."""
    if path is not None:
        return path, None
    base = os.environ.get("RRT_COMM_DIR") or ("/dev/shm" if os.path.isdir("/dev/shm") else "/tmp")
    return (os.path.join(base, f"rrt_comm_{_launcher_key()}.id"),
            os.path.join(base, f"rrt_comm_port_{os.environ.get('MASTER_ADDR', 'local')}_{os.environ.get('MASTER_PORT', '0')}.id"))


def _process_start_time() -> float:
    """Epoch seconds at which this process started (not the later moment at which this module was imported: rank 0 may have
    published the id while a slower rank was still importing numpy)."""
    try:
        with open("/proc/self/stat") as f:
            ticks = float(f.read().rsplit(")", 1)[1].split()[19])
        with open("/proc/uptime") as f:
            up = float(f.read().split()[0])
        return time.time() - (up - ticks / os.sysconf("SC_CLK_TCK"))
    except (OSError, IndexError, ValueError):
        return time.time()


_T_START = _process_start_time()
_FRESH_SLACK = 30.0  # ranks of one launch start within this many seconds of each other


def _publish(target: str, uid: bytes):
    """Write `uid` to `target` so that a reader sees all of it or nothing, without following a link somebody else planted
    in the (world-writable) directory: the temporary file is created exclusively, 0600, and renamed over the target."""
    tmp = f"{target}.{os.getpid()}.{time.time_ns()}.tmp"
    fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL | getattr(os, "O_NOFOLLOW", 0), 0o600)
    try:
        with os.fdopen(fd, "wb") as f:
            f.write(uid)
        os.replace(tmp, target)
    except BaseException:
        try:
            os.unlink(tmp)
        except OSError:
            pass
        raise


def _read_fresh(target: str):
    """The id in `target`, or None when the file is absent, not a regular file of ours, or older than this launch (a file a
    crashed earlier attempt left behind)."""
    try:
        fd = os.open(target, os.O_RDONLY | getattr(os, "O_NOFOLLOW", 0))
    except OSError:
        return None
    try:
        st = os.fstat(fd)
        import stat as _stat

        if not _stat.S_ISREG(st.st_mode) or st.st_uid != os.getuid() or st.st_mtime < _T_START - _FRESH_SLACK:
            return None
        with os.fdopen(os.dup(fd), "rb") as f:
            uid = f.read()
        return uid if len(uid) > 0 else None
    except OSError:
        return None
    finally:
        os.close(fd)


def exchange_unique_id(rank: int, world_size: int, make_id, path: str = None, timeout: float = 120.0) -> bytes:
    """Rank 0 calls `make_id()` (-> 128 bytes, ``_ffi.comm_unique_id``) and publishes the result; the other ranks of the
    node wait for it.  Single node, so a file on local tmpfs is the side channel.  Rank 0 first removes whatever an earlier
    attempt left under the same names, publishes, and removes the files again in `release_unique_id` once the communicator
    exists (ncclCommInitRank returns only after every rank has joined, i.e. has read the id).  A reader accepts a file only
    when it is a regular file of the same user written after this launch began (give or take `_FRESH_SLACK` seconds), under
    either name; the name itself carries the launcher's identity and a per-attempt nonce, so a restarted attempt does not
    even look at its predecessor's file."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError(f"bad rank {rank} of {world_size}")
    main, fallback = _id_paths(path)
    if rank == 0:
        for target in (main, fallback):
            if target is not None:
                try:
                    os.unlink(target)
                except OSError:
                    pass
        uid = bytes(make_id())
        for target in (main, fallback):
            if target is not None:
                _publish(target, uid)
        return uid
    t0 = time.monotonic()
    deadline = t0 + timeout
    while True:
        for target, late in ((main, False), (fallback, True)):
            if target is None or (late and time.monotonic() - t0 < 3.0):  # the fallback only after the shared name stayed absent
                continue
            uid = _read_fresh(target)
            if uid is not None:
                return uid
        if time.monotonic() > deadline:
            raise TimeoutError(f"rank {rank}: no communicator id at {main} after {timeout:.0f} s")
        time.sleep(0.01)


def release_unique_id(rank: int, path: str = None):
    if rank != 0:
        return
    for target in _id_paths(path):
        if target is not None:
            try:
                os.unlink(target)
            except OSError:
                pass


def init_comm(ctx, rank: int, world_size: int, path: str = None):
    """Create the RCCL communicator of `ctx` (an ``_ffi.Context`` on this rank's GPU): collective over all ranks."""
    from . import _ffi

    alt = os.environ.get("RRT_RCCL_LIB")  # a stand-in collective library (ranks that share one GPU: tests/fake_rccl); the default is RCCL
    if alt and not getattr(init_comm, "_lib_chosen", False):
        _ffi.comm_use_library(alt)
        init_comm._lib_chosen = True
    uid = exchange_unique_id(rank, world_size, _ffi.comm_unique_id, path=path)
    try:
        ctx.comm_init(rank, world_size, uid)
    finally:
        release_unique_id(rank, path=path)


# ---------------------------------------------------------------------------------- slab layout
def slab_bytes(Q: int, node_stride: int) -> int:
    return Q * node_stride * 16 + Q * 16


def unpack_slab(slab, Q: int, node_stride: int):
    """Split one rank's slab (uint8 numpy array) into (vcost[Q,stride] f64, nodes[Q,stride] u32, parent[Q,stride] i32,
    meta[Q,4] i32 = {status, j, vgoal, found}) -- the layout of rrt_batch_result_block / rrt_gather."""
    import numpy as np

    a = np.asarray(slab, dtype=np.uint8)
    if a.size != slab_bytes(Q, node_stride):
        raise ValueError(f"slab of {a.size} bytes does not hold {Q} queries of stride {node_stride}")
    n8 = Q * node_stride * 8
    n4 = Q * node_stride * 4
    vcost = a[:n8].view(np.float64).reshape(Q, node_stride)
    nodes = a[n8:n8 + n4].view(np.uint32).reshape(Q, node_stride)
    parent = a[n8 + n4:n8 + 2 * n4].view(np.int32).reshape(Q, node_stride)
    meta = a[n8 + 2 * n4:].view(np.int32).reshape(Q, 4)
    return vcost, nodes, parent, meta
