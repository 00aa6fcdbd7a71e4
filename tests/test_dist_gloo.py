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
    # the product's sharding entry point (what KeepNet.run_clips_u8 runs on): every rank ends up with the full list
    clips = [np.full((2, 4, 4, 3), 3 * c, dtype=np.uint8) for c in range(n_clips)]
    seen = []

    def local_fn(mine_d):
        seen.extend(mine_d)
        return {i: 255 - c for i, c in mine_d.items()}
    allr = kdist.sharded_map(clips, local_fn, gather='all')
    assert sorted(seen) == mine and len(allr) == n_clips
    assert all(np.array_equal(allr[c], 255 - clips[c]) for c in range(n_clips))
    root = kdist.sharded_map(clips, local_fn, gather='root')
    assert (root is None) == (rank != 0)
    # ragged clips (different T per clip, what a 41-crop video gives: 20 + 20 + 1) through the fixed-size tensor gather,
    # with and without the caller providing the shapes
    ragged = [np.full((t, 4, 4, 3), 10 + c, dtype=np.uint8) for c, t in enumerate((20, 20, 1, 7, 3))]
    for shapes in ([r.shape for r in ragged], None):
        got = kdist.sharded_map(ragged, lambda d: {i: (c // 2) for i, c in d.items()}, gather='root', shapes=shapes)
        if rank == 0:
            assert all(g.shape == r.shape and np.array_equal(g, r // 2) for g, r in zip(got, ragged))
        else:
            assert got is None
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def _worker_real_blob(rank, world, port, out_dir):
    """The REAL packed weight blob (896 reference tensors -> fused / permuted kernel layouts, 633 MB) through the same
    broadcast + adopt_packed path bench.py and a multi-GPU deployment use; every named view must arrive bit-identical."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    from __graft_entry__ import load_package
    load_package()
    import hashlib
    from comfyui_keep_amd.engine import dist as kdist, synth
    from comfyui_keep_amd.engine.arch import DEFAULT_ARCH
    from comfyui_keep_amd.engine.net import KeepNet
    kdist.init_from_env(backend='gloo')
    net = KeepNet(**DEFAULT_ARCH)
    if rank == 0:
        net.load_state_dict(synth.synth_state_dict(seed=0), strict=True)
        index, blob = kdist.broadcast_packed_weights(net._index, torch.from_numpy(net._blob), src=0)
    else:
        index, blob = kdist.broadcast_packed_weights(None, None, src=0)
    net.adopt_packed(index, blob)
    h = hashlib.blake2b(digest_size=16)
    for name in sorted(net.w):
        h.update(name.encode())
        h.update(str(tuple(net.w[name].shape)).encode())
        h.update(net.w[name].contiguous().numpy().tobytes())
    with open(os.path.join(out_dir, f'digest{rank}.txt'), 'w') as f:
        f.write(f'{len(net.w)} {blob.numel()} {h.hexdigest()}')
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_world2_real_weight_blob_broadcast(tmp_path):
    mp.spawn(_worker_real_blob, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    d0, d1 = (open(tmp_path / f'digest{r}.txt').read() for r in (0, 1))
    assert d0 == d1 and int(d0.split()[0]) > 600 and int(d0.split()[1]) > 150_000_000


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
