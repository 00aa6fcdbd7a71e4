"""Worker pool: the ComfyUI process drives N GPUs of one node (BASELINE configs[4]).

The reference runs in ONE process (nodes.py:111-136; clip loop keep_processor.py:263-270) and so does a ComfyUI node: nobody
launches it under ``torchrun``.  ``KEEP_AMD_GPUS=N`` (or ``KeepNet.start_pool(N)``) makes the process that owns the weights the
root of a pool: it spawns N - 1 worker processes -- one per additional GPU, ``engine/pool_worker.py`` -- joins them in a
``torch.distributed`` group just long enough to BROADCAST the packed weight blob once (backend ``nccl`` = RCCL over xGMI; the only
collective of the path, as ``north_star`` asks), and from then on ``KeepNet.run_clips_u8`` shards its clips round-robin over the
N GPUs: clips are independent (keep_arch.py:1050,1064,1113), nothing is exchanged between GPUs while they run.  uint8 crops travel
to the workers and restored uint8 crops back through POSIX shared memory (one memcpy each way per side; the control messages are
pickled over a loopback ``multiprocessing.connection``).  Rank 0 restores its own share on its own GPU while the workers run.

``KEEP_DIST_DEVICE=<d>`` puts every worker on device d (a 1-GPU box: the broadcast then runs over gloo -- RCCL refuses two ranks
on one device), which is how the GPU tests exercise this file.
"""
import atexit
import os
import secrets
import socket
import subprocess
import sys
import time
from multiprocessing import shared_memory
from multiprocessing.connection import Listener

import numpy as np
import torch

from . import dist as kdist

_WORKER = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'pool_worker.py')


def _free_port():
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        return sk.getsockname()[1]


def wanted_gpus():
    """KEEP_AMD_GPUS: how many GPUs of this node the product should drive (default 1 = no pool)."""
    try:
        return max(1, int(os.environ.get('KEEP_AMD_GPUS', '1')))
    except ValueError:
        return 1


class GpuPool:
    """Root side of the pool.  ``net`` is the root's KeepNet, weights already on its device."""

    def __init__(self, net, n_gpus, timeout=600.0):
        if n_gpus < 2:
            raise ValueError("a pool needs at least 2 GPUs")
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            raise RuntimeError("this process already belongs to a torch.distributed job (torchrun): run_clips_u8 shards over its ranks")
        self.world = int(n_gpus)
        self.net = net
        shared = os.environ.get('KEEP_DIST_DEVICE')
        n_dev = torch.cuda.device_count()
        if shared is None and n_dev < self.world:
            raise RuntimeError(f"KEEP_AMD_GPUS={self.world} but {n_dev} device(s) are visible (KEEP_DIST_DEVICE=<d> shares one device)")
        root_dev = net.device.index if net.device.index is not None else torch.cuda.current_device()
        # worker r drives device (root + r) mod count: the root keeps its own
        self.devices = [int(shared) if shared is not None else (root_dev + r) % n_dev for r in range(self.world)]
        backend = os.environ.get('KEEP_DIST_BACKEND') or ('gloo' if shared is not None else 'nccl')
        port, key = _free_port(), secrets.token_bytes(16)
        self._listener = Listener(('127.0.0.1', 0), authkey=key)
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'), KEEP_AMD_GPUS='1')
        env.pop('KEEP_POOL_AUTHKEY', None)
        self._procs = []
        for r in range(1, self.world):
            cmd = [sys.executable, _WORKER, '--rank', str(r), '--world', str(self.world), '--master-port', str(port),
                   '--device', str(self.devices[r]), '--backend', backend, '--ctl-port', str(self._listener.address[1]),
                   '--precision', net.precision, '--arch', repr(dict(net.cfg))]
            self._procs.append(subprocess.Popen(cmd, env=dict(env, KEEP_POOL_AUTHKEY=key.hex())))
        atexit.register(self.close)
        # control connections (workers connect as soon as they are up), then the process group, then the one collective
        self._conns = {}
        sock = getattr(getattr(self._listener, '_listener', None), '_socket', None)      # (accept() itself has no timeout)
        if sock is not None:
            sock.settimeout(5.0)
        deadline = time.monotonic() + timeout
        while len(self._conns) < self.world - 1:
            dead = [p.args[p.args.index('--rank') + 1] for p in self._procs if p.poll() is not None]
            if dead or time.monotonic() > deadline:
                self.close()
                raise RuntimeError(f"GPU pool: worker(s) {dead or '?'} did not come up (see their stderr above)")
            try:
                c = self._listener.accept()
            except (socket.timeout, TimeoutError, OSError):
                continue
            self._conns[int(c.recv())] = c
        t0 = time.perf_counter()
        torch.distributed.init_process_group(backend=backend, init_method=f'tcp://127.0.0.1:{port}', rank=0, world_size=self.world)
        try:
            if backend == 'nccl':
                torch.cuda.set_device(root_dev)
            kdist.broadcast_packed_weights(net._index, net.packed_blob(), src=0)
            torch.distributed.barrier()
        finally:
            torch.distributed.destroy_process_group()       # the path has no other collective: the group must not linger in ComfyUI
        self.broadcast_ms = (time.perf_counter() - t0) * 1e3
        for r, c in self._conns.items():
            msg = c.recv()
            if msg != ('ready', r):
                raise RuntimeError(f"pool worker {r} failed to start: {msg!r}")

    # ------------------------------------------------------------------ one sharded call
    def run(self, clips_u8, max_b=None):
        """list of uint8 [T_i,H,W,3] tensors (host) -> list of restored uint8 tensors, clip c on rank c % world."""
        n = len(clips_u8)
        jobs = {}
        for r in range(1, self.world):
            ids = list(range(r, n, self.world))
            if not ids:
                continue
            shapes = [tuple(clips_u8[i].shape) for i in ids]
            total = int(sum(int(np.prod(s)) for s in shapes))
            shm_in = shared_memory.SharedMemory(create=True, size=max(total, 1))
            shm_out = shared_memory.SharedMemory(create=True, size=max(total, 1))
            off = 0
            for i, s in zip(ids, shapes):
                k = int(np.prod(s))
                dst = np.ndarray(s, dtype=np.uint8, buffer=shm_in.buf, offset=off)
                c = clips_u8[i]
                if hasattr(c, 'frames'):                     # T separate crops (engine/net.py:_FrameList)
                    for t, fr in enumerate(c.frames):
                        dst[t] = fr
                else:
                    dst[...] = c.cpu().numpy()
                off += k
            self._conns[r].send(('run', shm_in.name, shm_out.name, ids, shapes, max_b))
            jobs[r] = (ids, shapes, shm_in, shm_out)
        mine = {i: clips_u8[i] for i in range(0, n, self.world)}
        out = [None] * n
        err = None
        try:
            local = self.net._run_clips_u8_local(mine, max_b) if mine else {}
            for i, a in local.items():
                out[i] = torch.from_numpy(a)
        except BaseException as e:                           # still drain the workers: their shared memory must be released
            err = e
        for r, (ids, shapes, shm_in, shm_out) in jobs.items():
            try:
                msg = self._conns[r].recv()
                if msg[0] != 'done':
                    raise RuntimeError(f"pool worker {r}: {msg[1]}")
                off = 0
                for i, s in zip(ids, shapes):
                    k = int(np.prod(s))
                    out[i] = torch.from_numpy(np.ndarray(s, dtype=np.uint8, buffer=shm_out.buf, offset=off).copy())
                    off += k
            except BaseException as e:
                err = err or e
            finally:
                for m in (shm_in, shm_out):
                    m.close()
                    m.unlink()
        if err is not None:
            raise err
        return out

    def close(self):
        for c in getattr(self, '_conns', {}).values():
            try:
                c.send(('exit',))
                c.close()
            except Exception:
                pass
        self._conns = {}
        for p in getattr(self, '_procs', []):
            try:
                p.wait(timeout=30)
            except Exception:
                p.kill()
        self._procs = []
        if getattr(self, '_listener', None) is not None:
            self._listener.close()
            self._listener = None
