"""One worker of the GPU pool (engine/pool.py): a process that owns ONE additional GPU.

Started by ``GpuPool`` as ``python pool_worker.py --rank r --world N ...``.  It joins the root's ``torch.distributed`` group,
receives the packed weight blob by the ONE broadcast of the path (RCCL over xGMI; gloo when several ranks share a device),
leaves the group and then serves ``run`` requests: uint8 crops in a shared-memory block -> ``KeepNet._run_clips_u8_local`` on its
GPU -> restored uint8 crops into a second shared-memory block.  The engine is imported by path, without the ComfyUI node surface
(``comfyui-keep_amd/__init__.py`` is never executed here): a worker needs neither ComfyUI nor the face helper.
"""
import argparse
import ast
import importlib
import os
import sys
import traceback
import types
from multiprocessing import resource_tracker, shared_memory
from multiprocessing.connection import Client


def _engine():
    pkg_dir = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    name = 'keep_amd_pool_worker'
    pkg = types.ModuleType(name)
    pkg.__path__ = [pkg_dir]
    sys.modules[name] = pkg
    return importlib.import_module(name + '.engine.net'), importlib.import_module(name + '.engine.dist')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rank', type=int, required=True)
    ap.add_argument('--world', type=int, required=True)
    ap.add_argument('--master-port', type=int, required=True)
    ap.add_argument('--ctl-port', type=int, required=True)
    ap.add_argument('--device', type=int, required=True)
    ap.add_argument('--backend', required=True)
    ap.add_argument('--precision', default='x3')
    ap.add_argument('--arch', required=True)
    a = ap.parse_args()
    conn = Client(('127.0.0.1', a.ctl_port), authkey=bytes.fromhex(os.environ['KEEP_POOL_AUTHKEY']))
    conn.send(a.rank)
    try:
        import numpy as np
        import torch
        knet, kdist = _engine()
        torch.cuda.set_device(a.device)
        torch.distributed.init_process_group(backend=a.backend, init_method=f'tcp://127.0.0.1:{a.master_port}', rank=a.rank,
                                             world_size=a.world)
        try:
            index, blob = kdist.broadcast_packed_weights(None, None, src=0)
            torch.distributed.barrier()
        finally:
            torch.distributed.destroy_process_group()
        net = knet.KeepNet(**ast.literal_eval(a.arch))
        net.adopt_packed(index, blob.to(torch.device('cuda', a.device)))
        net.eval().set_precision(a.precision)
        conn.send(('ready', a.rank))
    except BaseException:
        conn.send(('failed', traceback.format_exc()))
        raise
    while True:
        try:
            msg = conn.recv()
        except EOFError:
            break
        if msg[0] == 'exit':
            break
        _, name_in, name_out, ids, shapes, max_b = msg
        try:
            shm_in, shm_out = shared_memory.SharedMemory(name=name_in), shared_memory.SharedMemory(name=name_out)
            for m in (shm_in, shm_out):            # the ROOT owns (and unlinks) the blocks: python 3.10 registers attachments with this
                try:                               # process's resource tracker as well, which would unlink them a second time at exit
                    resource_tracker.unregister(m._name, 'shared_memory')
                except Exception:
                    pass
            try:
                mine, off = {}, 0
                for i, s in zip(ids, shapes):
                    k = int(np.prod(s))
                    mine[i] = torch.from_numpy(np.ndarray(s, dtype=np.uint8, buffer=shm_in.buf, offset=off))
                    off += k
                local = net._run_clips_u8_local(mine, max_b)
                off = 0
                for i, s in zip(ids, shapes):
                    k = int(np.prod(s))
                    np.ndarray(s, dtype=np.uint8, buffer=shm_out.buf, offset=off)[...] = local[i]
                    off += k
                del mine
            finally:
                shm_in.close()
                shm_out.close()
            conn.send(('done',))
        except BaseException:
            conn.send(('failed', traceback.format_exc()))


if __name__ == '__main__':
    main()
