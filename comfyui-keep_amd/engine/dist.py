"""Multi-GPU: one process per GPU, independent clips sharded data-parallel.

The path has NO exchange step: clips share no state (keep_arch.py:1050,1064,1113), so the only collective
is the one-off broadcast of the packed weight blob (633 MB fp32) from rank 0 -- ``torch.distributed``
backend ``nccl`` is RCCL on ROCm, i.e. one ncclBroadcast over xGMI.  Results are collected by clip index with one fixed-size
uint8 tensor gather to rank 0 (which alone runs the paste-back).  The same code runs on ``gloo`` for the CPU tests.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise the default process group from torchrun's env (RANK/WORLD_SIZE/MASTER_*).  Returns
    (rank, world, local_rank); a no-op (0, 1, 0) when not launched distributed."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world == 1:
        return 0, 1, 0
    rank = int(os.environ['RANK'])
    local = int(os.environ.get('LOCAL_RANK', rank))
    if os.environ.get('KEEP_DIST_DEVICE') is not None:      # single-GPU smoke runs of the N > 1 path: every rank on one device
        local = int(os.environ['KEEP_DIST_DEVICE'])
    if not dist.is_initialized():
        if backend is None:
            # several ranks on ONE device (KEEP_DIST_DEVICE): RCCL refuses that ("Duplicate GPU detected"), the wire is gloo
            shared = os.environ.get('KEEP_DIST_DEVICE') is not None
            backend = os.environ.get('KEEP_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() and not shared else 'gloo')
        if backend == 'nccl':
            torch.cuda.set_device(local)
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def rank_world():
    """(rank, world) of the default process group; (0, 1) when not running distributed."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_clips(n_clips, rank, world):
    """Static round-robin: clip c runs on rank c % world.  Returns this rank's clip indices (ascending)."""
    return list(range(rank, n_clips, world))


def broadcast_packed_weights(index, blob, src=0):
    """Rank ``src`` passes (index, blob tensor); others pass (None, None) and receive both.
    ``blob`` lives on the device the backend moves (cuda for nccl/RCCL, cpu for gloo)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return index, blob
    rank = dist.get_rank()
    nccl = dist.get_backend() == 'nccl'
    dev = torch.device('cuda', torch.cuda.current_device()) if nccl else torch.device('cpu')
    target = dev if (nccl or not torch.cuda.is_available()) else torch.device('cuda', torch.cuda.current_device())
    meta = [index, None if blob is None else int(blob.numel())] if rank == src else [None, None]
    dist.broadcast_object_list(meta, src=src)
    index, numel = meta
    if rank != src:
        blob = torch.empty(numel, dtype=torch.float32, device=dev)
    wire = blob if blob.device.type == dev.type else blob.to(dev)     # gloo moves host memory
    # pieces of at most BCAST_CHUNK_MB (256 MB) of the one flat blob: the first multi-GPU run of a fresh RCCL communicator does not start
    # with a single 633 MB call (registration / staging limits are per call), a failing piece names its byte range, and LAST_BROADCAST
    # records what went over the wire for the bench line.  Views of one allocation: no copy, the same bytes as one call.
    flat = wire.view(-1)
    step = max(1, int(float(os.environ.get('KEEP_BCAST_CHUNK_MB', '256')) * (1 << 20)) // flat.element_size())
    pieces = []
    for a in range(0, flat.numel(), step):
        b = min(flat.numel(), a + step)
        try:
            dist.broadcast(flat[a:b], src=src)
        except Exception as e:
            raise RuntimeError(f"weight broadcast failed on elements [{a}, {b}) of {flat.numel()} "
                               f"(backend {dist.get_backend()}, world {dist.get_world_size()}, rank {rank}): {e}") from e
        pieces.append(int((b - a) * flat.element_size()))
    LAST_BROADCAST.clear()
    LAST_BROADCAST.update(backend=str(dist.get_backend()), world=int(dist.get_world_size()), pieces=len(pieces), bytes=int(sum(pieces)),
                          largest_piece_bytes=int(max(pieces)) if pieces else 0)
    return index, (blob if rank == src else wire.to(target))


LAST_BROADCAST = {}      # what the last broadcast_packed_weights moved (backend, world, pieces, bytes): bench.py copies it into its line


def collective_library():
    """What actually carries the collectives of this process group, for the bench line: backend name, world size as the backend sees
    it, and -- for `nccl` (= RCCL on ROCm) -- the library version torch reports (what NCCL_DEBUG=VERSION would print)."""
    info = {"backend": None, "world": 1, "rccl_version": None}
    if dist.is_available() and dist.is_initialized():
        info["backend"], info["world"] = str(dist.get_backend()), int(dist.get_world_size())
        if info["backend"] == 'nccl':
            try:
                v = torch.cuda.nccl.version()
                info["rccl_version"] = '.'.join(str(x) for x in v) if isinstance(v, (tuple, list)) else str(v)
            except Exception as e:      # pragma: no cover
                info["rccl_version"] = f"unavailable ({e})"
    return info


def gather_by_clip(local_results, n_clips, rank, world, to_all=False, shapes=None):
    """local_results: {clip_index: uint8 numpy array [T,H,W,3]}.  Returns the full list on rank 0 (others: None), or on every
    rank with ``to_all``.

    ONE fixed-size uint8 tensor collective (``dist.gather`` / ``dist.all_gather``; nccl = RCCL moves device buffers over xGMI,
    gloo host buffers): every rank knows every clip's shape (``shapes``: all ranks hold the same clip list and the restored
    clip has the shape of its input), so each rank sends its clips back to back in ascending clip order, padded to the largest
    per-rank byte count -- no pickling, no per-object round trips (round 2 used all_gather_object on numpy arrays)."""
    if not dist.is_initialized() or world == 1:
        return [local_results[i] for i in range(n_clips)]
    import numpy as np
    if shapes is None:                                   # shape exchange (tiny) when the caller could not provide them
        mine = {i: tuple(a.shape) for i, a in local_results.items()}
        buckets = [None] * world
        dist.all_gather_object(buckets, mine)
        shapes = [None] * n_clips
        for b in buckets:
            for i, sh in b.items():
                shapes[i] = sh
    per_rank = [shard_clips(n_clips, r, world) for r in range(world)]
    nbytes = [sum(int(np.prod(shapes[i])) for i in ids) for ids in per_rank]
    cap = max(max(nbytes), 1)
    nccl = dist.get_backend() == 'nccl'
    dev = torch.device('cuda', torch.cuda.current_device()) if nccl else torch.device('cpu')
    send = torch.zeros(cap, dtype=torch.uint8, device=dev)
    off = 0
    for i in per_rank[rank]:
        a = torch.from_numpy(np.ascontiguousarray(local_results[i])).reshape(-1)
        send[off:off + a.numel()].copy_(a, non_blocking=True)
        off += a.numel()
    if to_all:
        recv = [torch.empty(cap, dtype=torch.uint8, device=dev) for _ in range(world)]
        dist.all_gather(recv, send)
    else:
        recv = [torch.empty(cap, dtype=torch.uint8, device=dev) for _ in range(world)] if rank == 0 else None
        dist.gather(send, recv, dst=0)
        if rank != 0:
            return None
    out = [None] * n_clips
    for r in range(world):
        flat = recv[r].cpu().numpy()
        off = 0
        for i in per_rank[r]:
            n = int(np.prod(shapes[i]))
            out[i] = flat[off:off + n].reshape(shapes[i])
            off += n
    return out


def sharded_map(items, local_fn, gather='root', shapes=None):
    """The product's multi-GPU pattern in one place: ``items`` (independent clips) are sharded round-robin over the
    ranks, ``local_fn({index: item})`` restores this rank's share and returns {index: uint8 numpy array}, and the
    results are collected by index with one tensor gather ('root': rank 0 gets the full list, the others None; 'all':
    every rank gets it; 'none': the local dict).  No data-path collective between clips -- they share no state."""
    rank, world = rank_world()
    mine = shard_clips(len(items), rank, world)
    local = local_fn({i: items[i] for i in mine})
    if world == 1:
        return [local[i] for i in range(len(items))]
    if gather == 'none':
        return local
    return gather_by_clip(local, len(items), rank, world, to_all=(gather == 'all'), shapes=shapes)
