"""Worker pool: the ComfyUI process drives N GPUs of one node (BASELINE configs[4]).

The reference runs in ONE process (nodes.py:111-136; clip loop keep_processor.py:263-270) and so does a ComfyUI node: nobody
launches it under ``torchrun``.  ``KEEP_AMD_GPUS=N`` (or ``KeepNet.start_pool(N)``) makes the process that owns the weights the
root of a pool: it spawns N - 1 worker processes -- one per additional GPU, ``engine/pool_worker.py`` -- joins them in a
``torch.distributed`` group just long enough to BROADCAST the packed weight blob once (backend ``nccl`` = RCCL over xGMI; the only
collective of the path, as ``north_star`` asks), and from then on ``KeepNet.run_clips_u8`` shards its clips round-robin over the
N GPUs: clips are independent (keep_arch.py:1050,1064,1113), nothing is exchanged between GPUs while they run.  uint8 crops travel
to the workers and restored uint8 crops back through POSIX shared memory (one memcpy each way per side; the control messages are
pickled over a loopback ``multiprocessing.connection``).  Rank 0 restores its own share on its own GPU while the workers run.

Round 5:
  * start-up cannot hang the node: rank 0 binds the rendezvous store BEFORE it spawns anybody (no released-port race), every worker
    reports ``imported`` (engine loaded, device selected) before anyone joins the group, the join itself has a short timeout, and
    every failure path reads the workers' ``failed`` messages / exit codes, closes the pool and raises with the worker's traceback;
  * the workers follow the root: ``run`` compares ``KeepNet.pool_config()`` (precision policy, plan reference batch, kernel
    overrides, graph mode) with what the workers were last told and re-configures them first; a weight change closes the pool
    (``KeepNet.load_state_dict`` / ``adopt_packed`` / ``.to('cpu')``) -- clips never run under two policies or two sets of weights;
  * messages carry sequence numbers; a reply to another request is a protocol error, not a result;
  * ONE pair of shared-memory arenas per worker, grown on demand and reused by every call (a fresh block per call costs its page
    faults again: 236 MB per 300 crops);
  * ``set_parser``: the workers also run ParseNet on the crops they restored (face_restoration_helper.py:418-424) and hand the class
    maps back with them, so that of a single video's ``detect -> restore -> parse -> paste`` chain only the paste-back itself stays
    on the root GPU;
  * no strong reference from the interpreter's exit handlers to the net: ``weakref.finalize`` owns the clean-up, the pool holds no
    reference to the net at all (``run(net, ...)``).

``KEEP_DIST_DEVICE=<d>`` puts every worker on device d (a 1-GPU box: the broadcast then runs over gloo -- RCCL refuses two ranks
on one device), which is how the GPU tests exercise this file; ``KEEP_POOL_FAKE_NET=1`` (tests/test_dist_gloo.py) replaces the
engine inside the workers by a stand-in so that the protocol can be driven on a machine without any GPU.
"""
import datetime
import os
import secrets
import socket
import subprocess
import sys
import time
import weakref
from multiprocessing import shared_memory
from multiprocessing.connection import Listener

import numpy as np
import torch

from . import dist as kdist

_WORKER = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'pool_worker.py')
JOIN_TIMEOUT_S = float(os.environ.get('KEEP_POOL_JOIN_TIMEOUT', '120'))


def wanted_gpus():
    """KEEP_AMD_GPUS: how many GPUs of this node the product should drive (default 1 = no pool)."""
    try:
        return max(1, int(os.environ.get('KEEP_AMD_GPUS', '1')))
    except ValueError:
        return 1


class PoolError(RuntimeError):
    pass


def _reset_default_group_counter():
    """The store keys of a default process group are prefixed with torch's count of groups created so far in THIS process.
    ``destroy_process_group`` resets the count, but an ``init_process_group`` that FAILED (a worker died around the rendezvous) leaves it
    advanced: the root's next pool would wait on '/1/...' keys while its freshly started workers post '/0/...' -- until the join
    timeout, every time.  No group is initialised here (checked by the caller), so the count restarts where a fresh worker's does."""
    try:
        if not torch.distributed.is_initialized():
            torch.distributed.distributed_c10d._world.group_count = 0
    except Exception:
        pass


def _cleanup(procs, conns, listener_box, arenas):
    """Everything a pool owns outside Python's heap (module-level: must not reference the pool or the net)."""
    for c in list(conns.values()):
        try:
            c.send(('exit',))
            c.close()
        except Exception:
            pass
    conns.clear()
    for p in list(procs):
        try:
            p.wait(timeout=30)
        except Exception:
            try:
                p.kill()
                p.wait(timeout=10)
            except Exception:
                pass
    procs.clear()
    if listener_box and listener_box[0] is not None:
        try:
            listener_box[0].close()
        except Exception:
            pass
        listener_box[0] = None
    for pair in list(arenas.values()):
        for m in pair:
            if m is not None:
                try:
                    m.close()
                    m.unlink()
                except Exception:
                    pass
    arenas.clear()


class GpuPool:
    """Root side of the pool.  ``net`` is the root's KeepNet (weights already on its device); the pool keeps no reference to it."""

    def __init__(self, net, n_gpus, timeout=600.0, join_timeout=None):
        if n_gpus < 2:
            raise ValueError("a pool needs at least 2 GPUs")
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            raise RuntimeError("this process already belongs to a torch.distributed job (torchrun): run_clips_u8 shards over its ranks")
        join_timeout = JOIN_TIMEOUT_S if join_timeout is None else float(join_timeout)
        self.world = int(n_gpus)
        shared = os.environ.get('KEEP_DIST_DEVICE')
        fake = os.environ.get('KEEP_POOL_FAKE_NET') == '1'
        n_dev = torch.cuda.device_count() if not fake else self.world
        if shared is None and n_dev < self.world:
            raise RuntimeError(f"KEEP_AMD_GPUS={self.world} but {n_dev} device(s) are visible (KEEP_DIST_DEVICE=<d> shares one device)")
        root_dev = net.device.index if getattr(net.device, 'index', None) is not None else (0 if fake else torch.cuda.current_device())
        # worker r drives device (root + r) mod count: the root keeps its own
        self.devices = [int(shared) if shared is not None else (root_dev + r) % n_dev for r in range(self.world)]
        backend = os.environ.get('KEEP_DIST_BACKEND') or ('gloo' if (shared is not None or fake) else 'nccl')
        self._procs, self._conns, self._arenas, self._listener_box = [], {}, {}, [None]
        self._seq = 0
        self._parser_sent = None
        self.weights_generation = getattr(net, 'weights_generation', 0)
        self._finalizer = weakref.finalize(self, _cleanup, self._procs, self._conns, self._listener_box, self._arenas)
        try:
            self._start(net, backend, root_dev, timeout, join_timeout)
        except BaseException:
            self.close()
            raise
        self.config = dict(net.pool_config())

    # ------------------------------------------------------------------ start-up
    def _start(self, net, backend, root_dev, timeout, join_timeout):
        key = secrets.token_bytes(16)
        listener = self._listener_box[0] = Listener(('127.0.0.1', 0), authkey=key)
        # rank 0 binds the rendezvous store first: the port is never free between "chosen" and "bound"
        store = torch.distributed.TCPStore('127.0.0.1', 0, self.world, True, datetime.timedelta(seconds=join_timeout), wait_for_workers=False)
        port = store.port
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'), KEEP_AMD_GPUS='1')
        env.pop('KEEP_POOL_AUTHKEY', None)
        for r in range(1, self.world):
            cmd = [sys.executable, _WORKER, '--rank', str(r), '--world', str(self.world), '--master-port', str(port),
                   '--device', str(self.devices[r]), '--backend', backend, '--ctl-port', str(listener.address[1]),
                   '--join-timeout', str(join_timeout), '--config', repr(dict(net.pool_config())), '--arch', repr(dict(net.cfg))]
            self._procs.append(subprocess.Popen(cmd, env=dict(env, KEEP_POOL_AUTHKEY=key.hex())))
        # 1. control connections (workers connect as soon as the interpreter is up)
        sock = getattr(getattr(listener, '_listener', None), '_socket', None)      # (accept() itself has no timeout)
        if sock is not None:
            sock.settimeout(2.0)
        deadline = time.monotonic() + timeout
        while len(self._conns) < self.world - 1:
            self._raise_if_dead("did not come up")
            if time.monotonic() > deadline:
                raise PoolError(f"GPU pool: only {len(self._conns)} of {self.world - 1} workers connected within {timeout:.0f} s")
            try:
                c = listener.accept()
            except (socket.timeout, TimeoutError, OSError):
                continue
            self._conns[int(c.recv())] = c
        # 2. every worker has imported the engine and selected its device -- only then does anybody join the group
        self._expect_all('imported', deadline)
        # 3. the process group, for exactly one collective; a worker that dies now costs `join_timeout`, not the c10d default
        t0 = time.perf_counter()
        _reset_default_group_counter()
        try:
            torch.distributed.init_process_group(backend=backend, store=store, rank=0, world_size=self.world,
                                                 timeout=datetime.timedelta(seconds=join_timeout))
        except BaseException as e:
            raise PoolError(f"GPU pool: joining the process group failed ({e!r}); workers: {self._failures() or 'no report'}") from e
        try:
            if backend == 'nccl':
                torch.cuda.set_device(root_dev)
            kdist.broadcast_packed_weights(net._index, net.packed_blob(), src=0)
            torch.distributed.barrier()
        except BaseException as e:
            raise PoolError(f"GPU pool: the weight broadcast failed ({e!r}); workers: {self._failures() or 'no report'}") from e
        finally:
            torch.distributed.destroy_process_group()       # the path has no other collective: the group must not linger in ComfyUI
        self.broadcast_ms = (time.perf_counter() - t0) * 1e3
        self._expect_all('ready', time.monotonic() + timeout)

    def _failures(self):
        """'failed' messages already sitting in the control connections + exit codes of dead workers (never blocks)."""
        out = {}
        for r, c in self._conns.items():
            try:
                while c.poll(0):
                    m = c.recv()
                    if isinstance(m, tuple) and m and m[0] == 'failed':
                        out[r] = m[-1]
            except (EOFError, OSError):
                out.setdefault(r, 'connection closed')
        for p in self._procs:
            if p.poll() is not None:
                r = int(p.args[p.args.index('--rank') + 1])
                out.setdefault(r, f'exited with code {p.returncode}')
        return out

    def _raise_if_dead(self, what):
        dead = [int(p.args[p.args.index('--rank') + 1]) for p in self._procs if p.poll() is not None]
        if dead:
            raise PoolError(f"GPU pool: worker(s) {dead} {what}: {self._failures()}")

    def _expect_all(self, tag, deadline):
        """One ``(tag, rank)`` message from every worker, watching for 'failed' reports and dead processes meanwhile."""
        waiting = set(self._conns)
        while waiting:
            for r in sorted(waiting):
                c = self._conns[r]
                try:
                    if not c.poll(0.2):
                        continue
                    m = c.recv()
                except (EOFError, OSError):
                    raise PoolError(f"GPU pool: worker {r} closed its connection before '{tag}': {self._failures()}")
                if m == (tag, r):
                    waiting.discard(r)
                elif isinstance(m, tuple) and m and m[0] == 'failed':
                    raise PoolError(f"GPU pool: worker {r} failed before '{tag}':\n{m[-1]}")
                else:
                    raise PoolError(f"GPU pool: worker {r} sent {m!r} where '{tag}' was expected")
            if waiting:
                self._raise_if_dead(f"died before '{tag}'")
                if time.monotonic() > deadline:
                    raise PoolError(f"GPU pool: workers {sorted(waiting)} did not report '{tag}' in time")

    # ------------------------------------------------------------------ requests
    def _request(self, ranks, make_msg):
        """Send one request per rank (same sequence number), return {rank: reply payload}; a reply with another number is refused."""
        self._seq += 1
        seq = self._seq
        sent = []
        err = None
        for r in ranks:
            try:
                self._conns[r].send(make_msg(r, seq))
                sent.append(r)
            except BaseException as e:
                err = err or PoolError(f"pool worker {r}: send failed ({e!r})")
                break
        return seq, sent, err

    def _collect(self, r, seq):
        try:
            msg = self._conns[r].recv()
        except (EOFError, OSError) as e:
            raise PoolError(f"pool worker {r} went away ({e!r}); exit code {self._procs[r - 1].poll()}")
        if not isinstance(msg, tuple) or len(msg) < 2 or msg[1] != seq:
            raise PoolError(f"pool worker {r}: reply {msg[:2] if isinstance(msg, tuple) else msg!r} does not answer request {seq}")
        if msg[0] == 'failed':
            raise PoolError(f"pool worker {r}:\n{msg[-1]}")
        return msg

    def _broadcast_request(self, make_msg):
        seq, sent, err = self._request(sorted(self._conns), make_msg)
        for r in sent:
            try:
                self._collect(r, seq)
            except BaseException as e:
                err = err or e
        if err is not None:
            self.close()                     # a half-configured pool must not serve clips
            raise err

    def configure(self, cfg):
        cfg = dict(cfg)
        self._broadcast_request(lambda r, seq: ('configure', seq, cfg))
        self.config = cfg

    def set_parser(self, engine):
        """Give every worker the ParseNet weights -- ``engine``: the root's ``engine/parsenet.py:ParseNetEngine`` (its packed blob is
        what travels) -- so that ``run(parse=True)`` also returns the class maps of the crops a worker restored.  Sent once per engine."""
        if self._parser_sent is engine:
            return
        packed = tuple(engine.packed()) if hasattr(engine, 'packed') else engine
        self._broadcast_request(lambda r, seq: ('parsenet', seq, packed))
        self._parser_sent = engine

    def _arena(self, r, need_in, need_out):
        """The worker's pair of shared-memory arenas, grown (never shrunk) to the sizes this call needs."""
        cur = self._arenas.get(r, (None, None))
        sizes = (need_in, need_out)
        new = list(cur)
        changed = False
        for k in (0, 1):
            if cur[k] is None or cur[k].size < sizes[k]:
                if cur[k] is not None:
                    cur[k].close()
                    cur[k].unlink()
                new[k] = shared_memory.SharedMemory(create=True, size=max(int(sizes[k] * 1.25), 1 << 20))
                changed = True
        self._arenas[r] = tuple(new)
        return self._arenas[r], changed

    # ------------------------------------------------------------------ one sharded call
    def run(self, net, clips_u8, max_b=None, sink=None, parse=False):
        """list of uint8 [T_i,H,W,3] tensors -> list of restored uint8 tensors, clip c on rank c % world.  ``sink`` as in
        ``KeepNet.run_clips_u8``: groups are handed over as they finish (the root's own on its GPU, a worker's in host memory, with
        the class maps when ``parse``) and None is returned."""
        if getattr(net, 'weights_generation', 0) != self.weights_generation:
            raise PoolError("the pool's workers hold other weights than this net (it should have been closed by load_state_dict)")
        cfg = dict(net.pool_config())
        if cfg != self.config:
            self.configure(cfg)
        n = len(clips_u8)
        jobs = {}
        for r in range(1, self.world):
            ids = list(range(r, n, self.world))
            if not ids:
                continue
            shapes = [tuple(int(v) for v in clips_u8[i].shape) for i in ids]
            total = int(sum(int(np.prod(s)) for s in shapes))
            n_cls = int(sum(int(np.prod(s[:3])) for s in shapes)) if parse else 0
            (shm_in, shm_out), changed = self._arena(r, total, total + n_cls)
            off = 0
            for i, s in zip(ids, shapes):
                k = int(np.prod(s))
                dst = np.ndarray(s, dtype=np.uint8, buffer=shm_in.buf, offset=off)
                c = clips_u8[i]
                if hasattr(c, 'frames'):                     # T separate crops (engine/net.py:_FrameList)
                    for t, fr in enumerate(c.frames):
                        dst[t] = fr
                else:
                    torch.from_numpy(dst).copy_(c)           # (host or device tensor)
                off += k
            jobs[r] = (ids, shapes, shm_in, shm_out, changed)
        seq, sent, err = self._request(
            sorted(jobs), lambda r, s: ('run', s, (jobs[r][2].name, jobs[r][3].name) if jobs[r][4] else None, jobs[r][0], jobs[r][1], max_b, bool(parse)))
        mine = {i: clips_u8[i] for i in range(0, n, self.world)}
        out = [None] * n
        if err is None:
            try:
                local = net._run_clips_u8_local(mine, max_b, sink=sink) if mine else {}
                for i, a in local.items():
                    out[i] = torch.from_numpy(a)
            except BaseException as e:                       # still drain the workers that were dispatched: their replies must
                err = e                                      # not be read by the NEXT call
        for r in sent:
            ids, shapes, shm_in, shm_out, _ = jobs[r]
            try:
                self._collect(r, seq)
                if err is not None:
                    continue
                off = 0
                crops, classes = [], []
                for i, s in zip(ids, shapes):
                    k = int(np.prod(s))
                    crops.append(torch.from_numpy(np.ndarray(s, dtype=np.uint8, buffer=shm_out.buf, offset=off).copy()))
                    off += k
                if parse:
                    for s in shapes:
                        k = int(np.prod(s[:3]))
                        classes.append(torch.from_numpy(np.ndarray(s[:3], dtype=np.uint8, buffer=shm_out.buf, offset=off).copy()))
                        off += k
                if sink is not None:
                    sink(list(ids), crops, classes if parse else None)
                else:
                    for i, c in zip(ids, crops):
                        out[i] = c
            except BaseException as e:
                err = err or e
        if err is not None:
            if isinstance(err, PoolError):
                self.close()                                 # protocol / worker failure: nothing of this pool can be trusted
            raise err
        return None if sink is not None else out

    def close(self):
        self._finalizer()                                    # idempotent: runs _cleanup once, also at garbage collection / exit
