"""One process per GPU: row sharding + RCCL bootstrap through ``torch.distributed``.

The reference has no distributed code; the data-parallel
structure used here is the one its loops imply: every per-sample quantity
(``A[i]``, ``y_pred[i]``, ``y[i]``) belongs to the rank that owns row i, and the
only cross-sample operations are the column sums of ``pcd._update``
(optimizer/pcd.py:54-59), ``pbcd._update`` (pbcd.py:60-72) and
``_cd_linear_epoch`` (cd_linear.py:15-18).  Those partial sums are all-reduced
(sum, f64) once per dependent step inside the C++ engine with RCCL; the prox and
regularizer recurrences then run replicated and bit-identical on every rank.

``torch.distributed`` is used only to learn rank / world size and to ship the
128-byte RCCL unique id; the data path never touches torch.
"""
import numpy as np


def _dist():
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        raise RuntimeError("distributed=True needs an initialised torch.distributed group "
                           "(one process per GPU; backend 'nccl' = RCCL, or 'gloo')")
    return dist


def rank_world():
    d = _dist()
    return d.get_rank(), d.get_world_size()


def row_block(n_samples, rank=None, world=None):
    """Contiguous block of rows owned by `rank`: [lo, hi)."""
    if rank is None:
        rank, world = rank_world()
    base, rem = divmod(int(n_samples), int(world))
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def broadcast_bytes(payload, src=0):
    """Broadcast a bytes object from `src` to all ranks (tiny control message)."""
    d = _dist()
    box = [payload if d.get_rank() == src else None]
    d.broadcast_object_list(box, src=src)
    return box[0]


def init_engine_comm(engine):
    """Create the engine's communicator spanning the torch process group: RCCL (one GPU per
    rank), or -- ``SPFM_COMM=shm``, ranks sharing one GPU on a test box, where RCCL cannot
    form a communicator -- the engine's host shared-memory exchange."""
    import os

    rank, world = rank_world()
    if os.environ.get("SPFM_COMM", "rccl") == "shm":
        name = broadcast_bytes(("/spfm_%d_%s" % (os.getpid(), os.environ.get("MASTER_PORT", "0")))
                               .encode() if rank == 0 else None, src=0).decode()
        engine.comm_init_shm(name, world, rank)
        _dist().barrier()  # every rank has mapped the segment
        if rank == 0:
            try:
                os.unlink("/dev/shm" + name)
            except OSError:
                pass
        return rank, world
    uid = engine.comm_unique_id() if rank == 0 else None
    uid = broadcast_bytes(uid, src=0)
    engine.comm_init(uid, world, rank)
    return rank, world


def connect_peers(engine):
    """Enable the in-kernel cross-GPU exchange of the persistent passes: all-gather the ranks'
    exchange-slab IPC handles over the (control) process group and map them.  Call after
    ``init_engine_comm``; ``SPFM_PEER=0`` keeps the per-step collective instead."""
    import os

    if os.environ.get("SPFM_PEER", "1") == "0":
        return False
    d = _dist()
    rank, world = rank_world()
    if world < 2 or world > 8:
        return False
    try:
        mine = engine.peer_alloc()
    except RuntimeError:
        mine = None
    box = [None] * world
    d.all_gather_object(box, mine)
    ok = all(h is not None for h in box)
    if ok:
        try:
            engine.peer_connect(world, rank, box)
        except RuntimeError:
            ok = False
    # all ranks use the peer exchange or none does (a rank that cannot map its peers' slabs --
    # no IPC / no peer access between two devices -- takes everybody to the collective)
    flags = [None] * world
    d.all_gather_object(flags, bool(ok))
    if not all(flags):
        if ok:
            engine.set_option("peer_exchange", 0)
        return False
    return True
