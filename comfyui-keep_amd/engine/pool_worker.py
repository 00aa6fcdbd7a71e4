"""One worker of the GPU pool (engine/pool.py): a process that owns ONE additional GPU.

Started by ``GpuPool`` as ``python pool_worker.py --rank r --world N ...``.  It reports ``imported`` once the engine is loaded and
its device selected, joins the root's ``torch.distributed`` group (a store the root bound beforehand, a short join timeout),
receives the packed weight blob by the ONE broadcast of the path (RCCL over xGMI; gloo when several ranks share a device), leaves
the group and then serves numbered requests:

  ('run', seq, arena names | None, ids, shapes, max_b, parse)   uint8 crops in the input arena -> ``KeepNet._run_clips_u8_local`` on
        this GPU -> restored uint8 crops (and, with ``parse``, the ParseNet class maps of those crops) into the output arena
  ('configure', seq, cfg)      ``KeepNet.apply_pool_config``: precision policy, plan reference batch, kernel overrides, graph mode
  ('parsenet', seq, packed)    rebuild ``engine/parsenet.py:ParseNetEngine`` from the root engine's packed blob (``ParseNetEngine.packed()``)

The engine is imported by path, without the ComfyUI node surface (``comfyui-keep_amd/__init__.py`` is never executed here): a
worker needs neither ComfyUI nor the face helper.  ``KEEP_POOL_FAKE_NET=1``: a stand-in engine (restored = 255 - crop, class map =
blue channel mod 19) so that the protocol runs on a machine without a GPU (tests/test_dist_gloo.py).
"""
import argparse
import ast
import datetime
import importlib
import os
import sys
import traceback
import types
from multiprocessing import resource_tracker, shared_memory
from multiprocessing.connection import Client


def _engine(*mods):
    pkg_dir = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    name = 'keep_amd_pool_worker'
    if name not in sys.modules:
        pkg = types.ModuleType(name)
        pkg.__path__ = [pkg_dir]
        sys.modules[name] = pkg
    return [importlib.import_module(f'{name}.engine.{m}') for m in mods]


class _FakeNet:
    """KEEP_POOL_FAKE_NET=1: the pool protocol without a GPU.  Keeps what it was configured with so tests can read it back through
    the results: restored = 255 - crop when precision is 'x3', 254 - crop otherwise."""

    def __init__(self, **arch):
        self.cfg, self.config, self.blob_sum = arch, None, None

    def adopt_packed(self, index, blob):
        self.blob_sum = float(blob.double().sum())

    def apply_pool_config(self, cfg):
        self.config = dict(cfg)

    def _run_clips_u8_local(self, mine, max_b=None, sink=None):
        import numpy as np
        top = 255 if self.config['precision'] == 'x3' else 254
        return {i: (top - c.numpy().astype(np.int16)).clip(0, 255).astype(np.uint8) for i, c in mine.items()}


def _attach(name):
    m = shared_memory.SharedMemory(name=name)
    try:                                   # the ROOT owns (and unlinks) the arenas: python 3.10 registers attachments with this
        resource_tracker.unregister(m._name, 'shared_memory')      # process's tracker too, which would unlink them again at exit
    except Exception:
        pass
    return m


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rank', type=int, required=True)
    ap.add_argument('--world', type=int, required=True)
    ap.add_argument('--master-port', type=int, required=True)
    ap.add_argument('--ctl-port', type=int, required=True)
    ap.add_argument('--device', type=int, required=True)
    ap.add_argument('--backend', required=True)
    ap.add_argument('--join-timeout', type=float, default=120.0)
    ap.add_argument('--config', required=True)
    ap.add_argument('--arch', required=True)
    a = ap.parse_args()
    conn = Client(('127.0.0.1', a.ctl_port), authkey=bytes.fromhex(os.environ['KEEP_POOL_AUTHKEY']))
    conn.send(a.rank)
    fake = os.environ.get('KEEP_POOL_FAKE_NET') == '1'
    # failure injection of the CPU tests ('<rank>:<stage>' raises at that stage of that worker): only with the stand-in engine
    fail_at = os.environ.get('KEEP_POOL_TEST_FAIL', '') if fake else ''
    try:
        import numpy as np
        import torch
        if fake:
            (kdist,) = _engine('dist')
            make_net = _FakeNet
            device = torch.device('cpu')
        else:
            knet, kdist = _engine('net', 'dist')
            make_net = knet.KeepNet
            torch.cuda.set_device(a.device)
            device = torch.device('cuda', a.device)
        if fail_at == f'{a.rank}:import':
            raise RuntimeError('injected failure before the group is joined')
        conn.send(('imported', a.rank))
        timeout = datetime.timedelta(seconds=a.join_timeout)
        store = torch.distributed.TCPStore('127.0.0.1', a.master_port, a.world, False, timeout)
        if fail_at == f'{a.rank}:join':
            raise RuntimeError('injected failure instead of joining the group')
        torch.distributed.init_process_group(backend=a.backend, store=store, rank=a.rank, world_size=a.world, timeout=timeout)
        try:
            index, blob = kdist.broadcast_packed_weights(None, None, src=0)
            torch.distributed.barrier()
        finally:
            torch.distributed.destroy_process_group()
        net = make_net(**ast.literal_eval(a.arch))
        net.adopt_packed(index, blob.to(device))
        if not fake:
            net.eval()
        net.apply_pool_config(ast.literal_eval(a.config))
        conn.send(('ready', a.rank))
    except BaseException:
        try:
            conn.send(('failed', 0, traceback.format_exc()))
        except Exception:
            pass
        raise
    parser = None
    arenas = [None, None]
    while True:
        try:
            msg = conn.recv()
        except EOFError:
            break
        if msg[0] == 'exit':
            break
        seq = msg[1]
        try:
            if msg[0] == 'configure':
                net.apply_pool_config(msg[2])
                conn.send(('ok', seq))
                continue
            if msg[0] == 'parsenet':
                if fake:
                    parser = 'fake'
                else:
                    (PN,) = _engine('parsenet')
                    parser = PN.ParseNetEngine.from_packed(*msg[2]).to(device)
                conn.send(('ok', seq))
                continue
            _, _, names, ids, shapes, max_b, parse = msg
            if names is not None:                        # the root (re)allocated this worker's arenas
                for m in arenas:
                    if m is not None:
                        m.close()
                arenas = [_attach(names[0]), _attach(names[1])]
            shm_in, shm_out = arenas
            mine, off = {}, 0
            for i, s in zip(ids, shapes):
                k = int(np.prod(s))
                mine[i] = torch.from_numpy(np.ndarray(s, dtype=np.uint8, buffer=shm_in.buf, offset=off))
                off += k
            if parse and parser is None:
                raise RuntimeError("run(parse=True) before the ParseNet weights were sent (GpuPool.set_parser)")
            cls = {}
            if parse and not fake:
                (L,) = _engine('hiplib')

                def keep_and_parse(gids, crops, _):
                    # crops: restored uint8 [T,H,W,3] on this GPU -- ParseNet input like face_restoration_helper.py:418-424
                    # (BGR uint8 -> RGB float (x / 255 - 0.5) / 0.5 = keep_img2tensor), in batches of <= 32 faces
                    for i, c in zip(gids, crops):
                        out_c = torch.empty(c.shape[:3], dtype=torch.uint8, device=c.device)
                        for s0 in range(0, c.shape[0], 32):
                            part = c[s0:s0 + 32].contiguous()
                            x = torch.empty(part.shape, dtype=torch.float32, device=c.device)
                            L.call('keep_img2tensor', part, x, part.numel() // 3)
                            out_c[s0:s0 + 32] = parser.classes(x)
                        local[i] = c
                        cls[i] = out_c
                local = {}
                with torch.cuda.device(device):
                    net._run_clips_u8_local(mine, max_b, sink=keep_and_parse)
                    local = {i: v.cpu().numpy() for i, v in local.items()}
                    cls = {i: v.cpu().numpy() for i, v in cls.items()}
            else:
                local = net._run_clips_u8_local(mine, max_b)
                if parse:
                    cls = {i: (local[i][..., 0] % 19).astype(np.uint8) for i in ids}
            off = 0
            for i, s in zip(ids, shapes):
                k = int(np.prod(s))
                np.ndarray(s, dtype=np.uint8, buffer=shm_out.buf, offset=off)[...] = local[i]
                off += k
            if parse:
                for i, s in zip(ids, shapes):
                    k = int(np.prod(s[:3]))
                    np.ndarray(s[:3], dtype=np.uint8, buffer=shm_out.buf, offset=off)[...] = cls[i]
                    off += k
            del mine, local, cls
            conn.send(('done', seq))
        except BaseException:
            conn.send(('failed', seq, traceback.format_exc()))
    for m in arenas:
        if m is not None:
            m.close()


if __name__ == '__main__':
    main()
