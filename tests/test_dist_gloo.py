"""CPU suite: the N>1 path on gloo, world_size 2 -- weight-blob broadcast, clip sharding, host gather by clip index.
(The data path has no collective: clips are independent; the only exchange is the one-off weight broadcast.)"""
import os
import socket
import sys

import numpy as np
import torch
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_clips, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    from __graft_entry__ import load_package
    load_package()
    from comfyui_keep_amd.engine import dist as kdist
    from comfyui_keep_amd.engine.weights import pack_blob, views
    r, w, _ = kdist.init_from_env(backend='gloo')
    assert (r, w) == (rank, world)
    # rank 0 owns the packed blob; everyone else receives index + blob
    if rank == 0:
        tensors = {'a.weight': torch.arange(12, dtype=torch.float32).view(3, 4), 'b.bias': torch.full((5,), 7.0)}
        blob, index = pack_blob(tensors)
        index, blob_t = kdist.broadcast_packed_weights(index, torch.from_numpy(blob), src=0)
    else:
        index, blob_t = kdist.broadcast_packed_weights(None, None, src=0)
    v = views(blob_t, index)
    assert torch.equal(v['a.weight'], torch.arange(12, dtype=torch.float32).view(3, 4)) and float(v['b.bias'][4]) == 7.0
    # independent clips: round-robin shard, "restore" = a function of the clip index only, gather on host by index
    mine = kdist.shard_clips(n_clips, rank, world)
    local = {c: np.full((2, 2), c * 10 + 1, dtype=np.uint8) for c in mine}
    full = kdist.gather_by_clip(local, n_clips, rank, world)
    if rank == 0:
        assert [int(a[0, 0]) for a in full] == [c * 10 + 1 for c in range(n_clips)]
        np.save(os.path.join(out_dir, 'ok.npy'), np.array([len(full)]))
    else:
        assert full is None
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_world2_broadcast_shard_gather(tmp_path):
    n_clips = 5
    mp.spawn(_worker, args=(2, _free_port(), n_clips, str(tmp_path)), nprocs=2, join=True)
    assert int(np.load(tmp_path / 'ok.npy')[0]) == n_clips


def test_shard_is_a_partition():
    sys.path.insert(0, ROOT)
    from comfyui_keep_amd.engine.dist import shard_clips
    for world in (1, 2, 4, 8):
        for n in (0, 1, 7, 15, 45):
            parts = [shard_clips(n, r, world) for r in range(world)]
            assert sorted(sum(parts, [])) == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
